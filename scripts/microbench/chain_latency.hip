// What one k-tile of a small product can cost at best: the dependent-issue latency of v_mfma_f32_32x32x16_f16, an LDS fragment read, a
// workgroup barrier of four waves, and an LDS-DMA round trip from L2 - each as a serial chain in ONE workgroup (the state a B = 1 forward
// runs in: one round of workgroups, nothing to overlap with), with the shader clock the chip runs at in that state.
//   hipcc -O3 --offload-arch=gfx950 scripts/microbench/chain_latency.hip -o /tmp/chain_latency && /tmp/chain_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long wall() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;  // 100 MHz
}

// mode 0: N dependent MFMAs on one accumulator; 1: two interleaved accumulators; 2: ds_read_b128 -> address of the next (dependent chain);
// 3: s_barrier x N (4 waves); 4: LDS-DMA of 1 KB per wave -> wait -> next (round trip from L2); 5: global_load_dwordx4 dependent chain
__global__ __launch_bounds__(256) void chain(int mode, int n, const float* __restrict__ src, unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 256) lds[i] = (float)((i * 7) & 1023) * 0;  // zeros: the pointer chase stays at offset 0
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0};
  f16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(0.001f * (lane + e)); y[e] = (_Float16)(0.002f * (lane - e)); }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall();
  float acc = 0.f;
  if (mode == 0) {
    for (int i = 0; i < n; ++i) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
  } else if (mode == 1) {
    for (int i = 0; i < n; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
    }
  } else if (mode == 2) {
    int off = lane * 4;
    for (int i = 0; i < n; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(lds + off);
      off = lane * 4 + (int)v.x;  // 0
      acc += v.y;
    }
  } else if (mode == 3) {
    for (int i = 0; i < n; ++i) __syncthreads();
  } else if (mode == 4) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0xffffffff, 0x00020000);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds + (tid >> 6) * 1024);
    for (int i = 0; i < n; ++i) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                   : "=&s"(keep)
                   : "v"((unsigned)(lane * 16)), "s"(rsrc), "s"(base), "s"((i & 63) * 4096)
                   : "memory");
    }
  } else {
    const f32x4* p = reinterpret_cast<const f32x4*>(src);
    int off = lane;
    for (int i = 0; i < n; ++i) {
      const f32x4 v = p[off + (i & 63) * 256];
      off = lane + (int)v.x;  // src holds zeros
      acc += v.y;
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall();
  if (tid == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  float s = acc;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  float* src; unsigned long long* out; float* sink;
  hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20);
  hipMalloc(&out, 16); hipMalloc(&sink, 4);
  const char* names[] = {"dependent v_mfma_f32_32x32x16_f16 (one accumulator)", "two accumulators interleaved (per pair)", "ds_read_b128 dependent chain",
                         "s_barrier, 4 waves", "LDS-DMA 1 KB per wave, issue -> landed (L2-resident source)", "global_load_dwordx4 dependent chain (L2)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
      const int n = 2000;
      unsigned long long h[2];
      hipLaunchKernelGGL(chain, dim3(1), dim3(256), 0, 0, mode, n, src, out, sink);
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      if (rep) printf("%-70s %7.1f shader cycles, %6.1f ns each   (clock %.2f GHz)\n", names[mode], (double)h[0] / n, (double)h[1] * 10.0 / n,
                      (double)h[0] / ((double)h[1] * 10.0));
    }
  // the same chains with 256 workgroups (every CU busy with the same thing)
  for (int mode = 0; mode < 6; mode += 4) {
    unsigned long long h[2];
    hipLaunchKernelGGL(chain, dim3(256), dim3(256), 0, 0, mode, 2000, src, out, sink);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("256 workgroups: %-54s %7.1f shader cycles, %6.1f ns each   (clock %.2f GHz)\n", names[mode], (double)h[0] / 2000, (double)h[1] * 10.0 / 2000,
           (double)h[0] / ((double)h[1] * 10.0));
  }
  return 0;
}
