// Microbenchmark (gfx950): round trip of a hand-off through LDS progress words between two waves of a 16-wave workgroup
// (ds_write_b32 by lane 0 -> spin on ds_read_b32 + ballot, optional s_sleep), while the other 14 waves idle, read LDS
// (8 x ds_read_b128 per iteration) or issue matrix instructions.
//   hipcc --offload-arch=gfx950 -O3 lds_pingpong.hip -o /tmp/lds_pingpong && /tmp/lds_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void post(unsigned addr, unsigned value, int lane) {
  if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(value) : "memory");
}
template <int SLEEP>
__device__ __forceinline__ void wait_ge(unsigned addr, unsigned need) {
  for (int spin = 0; spin < (1 << 20); ++spin) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    if (__builtin_amdgcn_ballot_w64((int)(v - need) < 0) == 0ull) return;
    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
  }
}

// BG: 0 other waves exit, 1 other waves read LDS, 2 other waves issue matrix instructions, 3 both
template <int BG, int SLEEP>
__global__ __launch_bounds__(1024) void pp_kernel(float* __restrict__ sink, int iters, int bg_iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds;
  if (tid < 64) reinterpret_cast<unsigned*>(lds)[32768 + tid] = 0u;
  for (int i = tid; i < 32768; i += 1024) lds[i] = 1.0f;
  __syncthreads();
  const unsigned fa = lds0 + 32768 * 4, fb = fa + 64;
  if (wave == 0) {
    const long long t0 = wall_clock64(), c0 = clock64();
    for (int i = 1; i <= iters; ++i) {
      post(fa, i, lane);
      wait_ge<SLEEP>(fb, i);
    }
    const long long t1 = wall_clock64(), c1 = clock64();
    if (lane == 0 && blockIdx.x < 256) {
      reinterpret_cast<long long*>(sink)[2 * blockIdx.x] = t1 - t0;      // 100 MHz ticks
      reinterpret_cast<long long*>(sink)[2 * blockIdx.x + 1] = c1 - c0;  // shader clocks
    }
    return;
  }
  if (wave == 12) {
    for (int i = 1; i <= iters; ++i) {
      wait_ge<SLEEP>(fa, i);
      post(fb, i, lane);
    }
    return;
  }
  if (BG == 0) return;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int n0 = lane & 31, hb = lane >> 5, swz = (n0 >> 2) & 3;
  for (int it = 0; it < bg_iters; ++it) {
    f32x4 f[8];
    if (BG & 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        f[j] = *reinterpret_cast<const f32x4*>(lds + ((wave & 3) * 1024 + n0 * 16 + 4 * ((2 * ((j >> 2) & 1) + hb) ^ swz) + (j & 3) * 512 + (it & 7) * 4096));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = f32x4{1.f, 2.f, 3.f, (float)it};
    }
    if (BG & 2) {
#pragma unroll
      for (int m = 0; m < 12; ++m)
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[m & 7]), __builtin_bit_cast(f16x8, f[(m + 1) & 7]), acc[m & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(f[j]));
    }
  }
  float k = 0.f;
  for (int a = 0; a < 4; ++a) k += acc[a][0];
  if (k == 123.456f) sink[4096 + tid] = k;
}

template <int BG, int SLEEP>
static void run(const char* what, float* sink) {
  const int iters = 2000;
  auto k = pp_kernel<BG, SLEEP>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  // background work sized to outlast the ping-pong; time = the ping-pong only if it is the longer part, so run both lengths
  for (int bg_iters : {0, 1 << 30}) {
    if (BG == 0 && bg_iters) continue;
    const int bgi = bg_iters ? 6000 : 0;
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 140 * 1024, 0, sink, 10, bgi ? 10 : 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 140 * 1024, 0, sink, iters, bgi);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    long long h[512];
    CK(hipMemcpy(h, sink, sizeof(h), hipMemcpyDeviceToHost));
    double tw = 0, tc = 0;
    for (int i = 0; i < 256; ++i) { tw += h[2 * i]; tc += h[2 * i + 1]; }
    printf("%-40s sleep %d, background %s: kernel %8.1f us; round trip %7.1f ns = %6.0f shader clocks (mean over 256 CUs)\n", what, SLEEP, bgi ? "busy" : "none",
           ms * 1e3, tw / 256 * 10.0 / iters, tc / 256 / iters);
  }
}

int main() {
  float* sink;
  CK(hipMalloc(&sink, 1 << 20));
  run<0, 1>("others idle", sink);
  run<0, 0>("others idle", sink);
  run<1, 1>("others read LDS", sink);
  run<1, 0>("others read LDS", sink);
  run<2, 1>("others issue matrix instructions", sink);
  run<3, 1>("others read LDS + matrix instructions", sink);
  run<3, 0>("others read LDS + matrix instructions", sink);
  return 0;
}
