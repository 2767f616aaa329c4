// Microbenchmark (gfx950): ds_read_b128 throughput of a CU in the fragment-read pattern of the split-f16 GEMMs (rows of 64 bytes,
// 16-byte chunks XOR-swizzled by (row >> 2) & 3, lane = (row n0, k-half hb)), against a lane-linear pattern, for 4 / 8 / 12 / 16 waves
// per CU and 8 or 16 reads per s_waitcnt, with and without matrix instructions between the waits.
//   hipcc --offload-arch=gfx950 -O3 lds_read_bw.hip -o /tmp/lds_read_bw && /tmp/lds_read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// PATTERN 0: GEMM fragment pattern, 1: lane-linear (lane * 16 bytes), 2: GEMM pattern without the swizzle
template <int PATTERN, int READS, int MFMAS>
__global__ __launch_bounds__(1024) void read_kernel(float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * 1024; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const int n0 = lane & 31, hb = lane >> 5;
  const int swz = PATTERN == 0 ? (n0 >> 2) & 3 : 0;
  const int wbase = (wave & 3) * 1024;  // floats
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f32x4 keep = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x4 f[READS];
#pragma unroll
    for (int j = 0; j < READS; ++j) {
      // fragment j: 32-row block j & 3, plane (j >> 2) & 1 (hi / lo): row n0, chunk (2 plane + hb) ^ swz
      const int off = PATTERN == 1 ? lane * 4 + j * 256 : n0 * 16 + 4 * ((2 * ((j >> 2) & 1) + hb) ^ swz) + (j & 3) * 512 + (j >> 3) * 2048;
      f[j] = *reinterpret_cast<const f32x4*>(lds + ((wbase + off + (it & 7) * 4096) & (32 * 1024 - 1)));
    }
    if (MFMAS > 0) {
#pragma unroll
      for (int m = 0; m < MFMAS; ++m) {
        const f16x8 a = __builtin_bit_cast(f16x8, f[m % READS]);
        const f16x8 b = __builtin_bit_cast(f16x8, f[(m + 1) % READS]);
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < READS; ++j) asm volatile("" ::"v"(f[j]));
    }
  }
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) keep[r] += acc[a][r];
  if (keep[0] == 123.456f) sink[tid] = keep[0] + keep[1] + keep[2] + keep[3];
}

// One wave per SIMD, wave tile 64 x 128 of the split-f16 GEMM: 12 fragment reads + 24 matrix instructions per k-tile, the reads of
// k-tile t + 1 issued AHEAD of the matrix instructions of k-tile t (two register sets).  clk[0] / clk[1]: shader clocks / 100 MHz ticks.
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void pipelined_kernel(float* __restrict__ sink, unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * 1024; i += blockDim.x) lds[i] = (float)(i & 1023) * 0.001f;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  const int n0 = lane & 31, hb = lane >> 5, swz = (n0 >> 2) & 3;
  const int wbase = (wave & 3) * 1024;
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  auto rd = [&](int it, f16x8 (&f)[12]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int off = n0 * 16 + 4 * ((2 * ((j >> 2) & 1) + hb) ^ swz) + (j & 3) * 512 + (j >> 3) * 2048;
      f[j] = *reinterpret_cast<const f16x8*>(lds + ((wbase + off + (it & 7) * 4096) & (32 * 1024 - 1)));
    }
  };
  auto mm = [&](f16x8 (&f)[12]) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 24; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m % 12], f[(m + 5) % 12], acc[m & 7], 0, 0, 0);
  };
  f16x8 fa[12], fb[12];
  rd(0, fa);
  for (int it = 0; it < iters; it += 2) {
    rd(it + 1, fb);
    mm(fa);
    rd(it + 2, fa);
    mm(fb);
  }
  float k = 0.f;
  for (int a = 0; a < 8; ++a) k += acc[a][0];
  if (k == 123.456f) sink[tid] = k;
  if (tid == 0) {
    atomicAdd(&clk[0], (unsigned long long)(clock64() - c0));
    atomicAdd(&clk[1], (unsigned long long)(wall_clock64() - w0));
  }
}
// The same loop with the other full-rate f16 shape, v_mfma_f32_16x16x32_f16: the same 12 fragment reads, 48 instructions of 16 cycles
// instead of 24 of 32 (32 accumulator tiles of 4 registers).  mfma_peak.hip: from registers, on random data, this shape sustains
// 1.76 PFLOP/s at 1.72 GHz where 32x32x16 sustains 1.53 at 1.48.
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void pipelined16_kernel(float* __restrict__ sink, unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * 1024; i += blockDim.x) lds[i] = (float)(i & 1023) * 0.001f;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  const int n0 = lane & 31, hb = lane >> 5, swz = (n0 >> 2) & 3;
  const int wbase = (wave & 3) * 1024;
  f32x4 acc[32];
  for (int a = 0; a < 32; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
  auto rd = [&](int it, f16x8 (&f)[12]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int off = n0 * 16 + 4 * ((2 * ((j >> 2) & 1) + hb) ^ swz) + (j & 3) * 512 + (j >> 3) * 2048;
      f[j] = *reinterpret_cast<const f16x8*>(lds + ((wbase + off + (it & 7) * 4096) & (32 * 1024 - 1)));
    }
  };
  auto mm = [&](f16x8 (&f)[12]) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 48; ++m) acc[m & 31] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[m % 12], f[(m + 5) % 12], acc[m & 31], 0, 0, 0);
  };
  f16x8 fa[12], fb[12];
  rd(0, fa);
  for (int it = 0; it < iters; it += 2) {
    rd(it + 1, fb);
    mm(fa);
    rd(it + 2, fa);
    mm(fb);
  }
  float k = 0.f;
  for (int a = 0; a < 32; ++a) k += acc[a][0];
  if (k == 123.456f) sink[tid] = k;
  if (tid == 0) {
    atomicAdd(&clk[0], (unsigned long long)(clock64() - c0));
    atomicAdd(&clk[1], (unsigned long long)(wall_clock64() - w0));
  }
}
template <int WAVES_PER_SIMD>
static void run_pipelined16(float* sink) {
  const int iters = 20000;
  auto k = pipelined16_kernel<WAVES_PER_SIMD>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  unsigned long long* clk;
  CK(hipMalloc(&clk, 16));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(256), dim3(256 * WAVES_PER_SIMD), 128 * 1024, 0, sink, clk, 100);
  CK(hipDeviceSynchronize());
  CK(hipMemset(clk, 0, 16));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(256), dim3(256 * WAVES_PER_SIMD), 128 * 1024, 0, sink, clk, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  unsigned long long h[2];
  CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  const double ns = ms * 1e6 / iters;
  printf("pipelined, 16x16x32 shape, %d wave(s)/SIMD: %7.1f ns per k-tile of 48 mfma per wave = %5.1f %% of the matrix peak at 2.4 GHz; shader clock %.2f GHz -> %5.1f %% at that clock\n",
         WAVES_PER_SIMD, ns, 100.0 * WAVES_PER_SIMD * 48 * 16 / 2.4 / ns, ghz, 100.0 * WAVES_PER_SIMD * 48 * 16 / ghz / ns);
}

template <int WAVES_PER_SIMD>
static void run_pipelined(float* sink) {
  const int iters = 20000;
  auto k = pipelined_kernel<WAVES_PER_SIMD>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  unsigned long long* clk;
  CK(hipMalloc(&clk, 16));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(256), dim3(256 * WAVES_PER_SIMD), 128 * 1024, 0, sink, clk, 100);
  CK(hipDeviceSynchronize());
  CK(hipMemset(clk, 0, 16));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(256), dim3(256 * WAVES_PER_SIMD), 128 * 1024, 0, sink, clk, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  unsigned long long h[2];
  CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  const double ns = ms * 1e6 / iters;
  printf("pipelined 64x128 wave tile, %d wave(s)/SIMD: %7.1f ns per k-tile of 24 mfma per wave = %5.1f %% of the matrix peak at 2.4 GHz; shader clock %.2f GHz -> %5.1f %% at that clock\n",
         WAVES_PER_SIMD, ns, 100.0 * WAVES_PER_SIMD * 24 * 32 / 2.4 / ns, ghz, 100.0 * WAVES_PER_SIMD * 24 * 32 / ghz / ns);
}

template <int PATTERN, int READS, int MFMAS>
static void run(const char* what, int waves, float* sink) {
  const int iters = 20000;
  auto k = read_kernel<PATTERN, READS, MFMAS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 128 * 1024, 0, sink, 100);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 128 * 1024, 0, sink, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)iters * waves * READS * 1024.0;  // per CU
  const double ns_per_iter = ms * 1e6 / iters;
  printf("%-44s %2d waves/CU x %2d reads, %2d mfma: %8.1f ns/iter  %6.1f B/ns/CU (%5.1f B/clk at 2.4 GHz)  mfma %5.1f %% of 32 clk each at 2.4 GHz\n", what, waves, READS, MFMAS,
         ns_per_iter, bytes / (ms * 1e6), bytes / (ms * 1e6) / 2.4, MFMAS ? 100.0 * (waves / 4.0) * MFMAS * 32 / 2.4 / ns_per_iter : 0.0);
}

int main(int argc, char** argv) {
  float* sink;
  CK(hipMalloc(&sink, 1 << 20));
  const bool only_pipelined = argc > 1;
  if (!only_pipelined)
  for (int waves : {4, 8, 12, 16}) {
    run<0, 8, 0>("fragment pattern (swizzled)", waves, sink);
    run<0, 16, 0>("fragment pattern (swizzled)", waves, sink);
    run<1, 8, 0>("lane-linear", waves, sink);
    run<2, 8, 0>("fragment pattern, no swizzle", waves, sink);
  }
  if (!only_pipelined)
  for (int waves : {4, 8, 12}) {
    run<0, 8, 12>("fragment pattern + matrix instructions", waves, sink);
    run<0, 12, 24>("fragment pattern + matrix instructions", waves, sink);
    run<0, 8, 24>("fragment pattern + matrix instructions", waves, sink);
  }
  run_pipelined<1>(sink);
  run_pipelined<2>(sink);
  run_pipelined16<1>(sink);
  run_pipelined16<2>(sink);
  return 0;
}
