// Microbenchmark (gfx950): what v_mfma_f32_32x32x16_f16 sustains from REGISTER operands - no LDS, no memory traffic at all - with
// operands that are zero, constant, or random f16 bit patterns, at 1 / 2 / 4 waves per SIMD, and the shader clock the chip holds
// meanwhile (s_memtime against the 100 MHz wall clock).  The 2.5 PFLOP/s figure is 1024 SIMDs x 512 f16 multiply-adds per clock
// at 2.4 GHz; this shows how much of it survives the power limit, before a GEMM adds operand traffic.
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// (4 waves per SIMD leave 128 registers per lane: 4 accumulator tiles there, 8 otherwise - enough independent chains either way)
template <int WPS>
__global__ __launch_bounds__(256 * WPS) void spin(const unsigned* __restrict__ seed, float* __restrict__ sink, unsigned long long* clk, int iters) {
  constexpr int NACC = WPS == 4 ? 4 : 8;
  const int tid = threadIdx.x;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    unsigned w[4], v[4];
    for (int k = 0; k < 4; ++k) {
      w[k] = seed[(blockIdx.x * 1024 + tid) * 32 + i * 8 + k];
      v[k] = seed[(blockIdx.x * 1024 + tid) * 32 + i * 8 + 4 + k];
    }
    a[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<uint4*>(w));
    b[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<uint4*>(v));
  }
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[(j + 1 + u) & 3], acc[j], 0, 0, 0);
  }
  if (tid == 0) {
    atomicAdd(&clk[0], (unsigned long long)(clock64() - c0));
    atomicAdd(&clk[1], (unsigned long long)(wall_clock64() - w0));
  }
  float k = 0.f;
  for (int j = 0; j < NACC; ++j) k += acc[j][0] + acc[j][7];
  if (k == 123.456f) sink[tid] = k;
}

// the other full-rate f16 shape: 16x16x32 (8,192 multiply-adds in 16 cycles; 4 accumulator registers per chain)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPS>
__global__ __launch_bounds__(256 * WPS) void spin16(const unsigned* __restrict__ seed, float* __restrict__ sink, unsigned long long* clk, int iters) {
  const int tid = threadIdx.x;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    unsigned w[4], v[4];
    for (int k = 0; k < 4; ++k) {
      w[k] = seed[(blockIdx.x * 1024 + tid) * 32 + i * 8 + k];
      v[k] = seed[(blockIdx.x * 1024 + tid) * 32 + i * 8 + 4 + k];
    }
    a[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<uint4*>(w));
    b[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<uint4*>(v));
  }
  f32x4 acc[16];
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);
  }
  if (tid == 0) {
    atomicAdd(&clk[0], (unsigned long long)(clock64() - c0));
    atomicAdd(&clk[1], (unsigned long long)(wall_clock64() - w0));
  }
  float k = 0.f;
  for (int j = 0; j < 16; ++j) k += acc[j][0] + acc[j][3];
  if (k == 123.456f) sink[tid] = k;
}

int main() {
  const int iters = 40000;
  unsigned* seed;
  float* sink;
  unsigned long long* clk;
  const size_t n = (size_t)256 * 1024 * 32;
  CK(hipMalloc(&seed, n * 4));
  CK(hipMalloc(&sink, 1 << 16));
  CK(hipMalloc(&clk, 16));
  unsigned* h = (unsigned*)malloc(n * 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int data = 0; data < 3; ++data) {
    for (size_t i = 0; i < n; ++i) {
      if (data == 0) h[i] = 0u;                       // zeros
      else if (data == 1) h[i] = 0x3c003c00u;         // 1.0, 1.0
      else {                                          // random finite f16 pairs in [-2, 2): sign, exponent 0..15 (of bias 15), mantissa
        const unsigned r = (unsigned)rand() * 2654435761u + (unsigned)rand();
        const unsigned lo = (r & 0x83ffu) | (((r >> 10) & 0xf) << 10), hi = ((r >> 16) & 0x83ffu) | (((r >> 26) & 0xf) << 10);
        h[i] = lo | (hi << 16);
      }
    }
    CK(hipMemcpy(seed, h, n * 4, hipMemcpyHostToDevice));
    for (int wps : {1, 2, 4}) {
      auto launch = [&](int n) {
        if (wps == 1) hipLaunchKernelGGL(spin<1>, dim3(256), dim3(256), 0, 0, seed, sink, clk, n);
        else if (wps == 2) hipLaunchKernelGGL(spin<2>, dim3(256), dim3(512), 0, 0, seed, sink, clk, n);
        else hipLaunchKernelGGL(spin<4>, dim3(256), dim3(1024), 0, 0, seed, sink, clk, n);
      };
      launch(200);
      CK(hipDeviceSynchronize());
      CK(hipMemset(clk, 0, 16));
      CK(hipEventRecord(e0));
      launch(iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c[2];
      CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
      const double ghz = (double)c[0] / (double)c[1] * 0.1;
      const double flops = 2.0 * 32 * 32 * 16 * 8.0 * iters * 1024.0 * wps;   // per launch
      const double tf = flops / (ms * 1e-3) / 1e12;
      printf("32x32x16 %-18s %d wave(s)/SIMD: %7.1f TFLOP/s = %.3f of 2500; shader clock %.2f GHz -> %.3f of the rate at that clock\n",
             data == 0 ? "zero operands" : data == 1 ? "constant 1.0" : "random f16", wps, tf, tf / 2500.0, ghz, tf / (2500.0 * ghz / 2.4));
    }
    for (int wps : {1, 2}) {
      auto launch = [&](int n) {
        if (wps == 1) hipLaunchKernelGGL(spin16<1>, dim3(256), dim3(256), 0, 0, seed, sink, clk, n);
        else hipLaunchKernelGGL(spin16<2>, dim3(256), dim3(512), 0, 0, seed, sink, clk, n);
      };
      launch(200);
      CK(hipDeviceSynchronize());
      CK(hipMemset(clk, 0, 16));
      CK(hipEventRecord(e0));
      launch(iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c[2];
      CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
      const double ghz = (double)c[0] / (double)c[1] * 0.1;
      const double tf = 2.0 * 16 * 16 * 32 * 16.0 * iters * 1024.0 * wps / (ms * 1e-3) / 1e12;
      printf("16x16x32 %-18s %d wave(s)/SIMD: %7.1f TFLOP/s = %.3f of 2500; shader clock %.2f GHz -> %.3f of the rate at that clock\n",
             data == 0 ? "zero operands" : data == 1 ? "constant 1.0" : "random f16", wps, tf, tf / 2500.0, ghz, tf / (2500.0 * ghz / 2.4));
    }
  }
  return 0;
}
