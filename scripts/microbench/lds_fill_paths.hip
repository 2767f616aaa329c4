// Microbenchmark (gfx950): how fast can a CU fill LDS from an L2-resident source, and what do concurrent fragment reads cost?
//   mode 0: LDS-DMA  (buffer_load_dwordx4 ... lds, 64 lanes x 16 B per instruction, the path of the GEMMs; 39 B/clk/CU measured
//           inside gemm_split_kernel's skeleton, profiles/r02_f_gemm_split_ablation.txt)
//   mode 1: through registers (global_load_dwordx4 -> ds_write_b128, the loads of batch it+1 in flight while batch it is written)
// each with 0 or 12 ds_read_b128 per wave and iteration (the fragment reads of a 128x256 GEMM tile) and 1 or 2 workgroups per CU.
// One "iteration" moves 24 KB per workgroup (a k-tile of the 128x256 split-f16 GEMM), barrier per iteration as in the GEMM.
//   hipcc --offload-arch=gfx950 -O3 lds_fill_paths.hip -o /tmp/lds_fill && /tmp/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int STAGE = 24 * 1024, NS = 3, SRC_BYTES = 8 << 20;

__device__ __forceinline__ void sdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(lds_base), "s"(soff)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, bool READS>
__global__ __launch_bounds__(256, 2) void fill_kernel(const float* __restrict__ src, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, SRC_BYTES, 0x00020000);
  f32x4 keepalive = {0.f, 0.f, 0.f, 0.f};
  unsigned blk = (blockIdx.x * 977u) % (SRC_BYTES / STAGE);  // this workgroup's current 24 KB block of the source
  auto next_blk = [&]() { blk = blk + 1 == SRC_BYTES / STAGE ? 0 : blk + 1; };
  auto reads = [&](int stage) {
    if constexpr (READS) {
#pragma unroll
      for (int j = 0; j < 12; ++j)
        keepalive += *reinterpret_cast<const f32x4*>(lds + stage * (STAGE / 4) + ((wave * 12 + j) % 24) * 256 + lane * 4);
    }
  };
  if constexpr (MODE == 0) {
    auto issue = [&](int stage) {
#pragma unroll
      for (int q = 0; q < 6; ++q)
        sdma16(rsrc, (unsigned)lane * 16u, (int)(blk * STAGE + (wave * 6 + q) * 1024), lds0 + stage * STAGE + (wave * 6 + q) * 1024);
      next_blk();
    };
    issue(0);
    issue(1);
    int stage = 0, istage = 2;
    for (int it = 0; it < iters; ++it) {
      wait_vm<6>();  // two batches in flight: the older one has landed
      __syncthreads();
      issue(istage);
      reads(stage);
      stage = stage + 1 == NS ? 0 : stage + 1;
      istage = istage + 1 == NS ? 0 : istage + 1;
    }
    wait_vm<0>();
  } else {
    f32x4 r[6];
    auto gload = [&]() {
#pragma unroll
      for (int q = 0; q < 6; ++q) r[q] = *reinterpret_cast<const f32x4*>(src + (size_t)blk * (STAGE / 4) + (wave * 6 + q) * 256 + lane * 4);
      next_blk();
    };
    gload();
    int stage = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(lds + stage * (STAGE / 4) + (wave * 6 + q) * 256 + lane * 4) = r[q];
      gload();  // the next batch is in flight while this one is consumed
      __syncthreads();
      reads(stage);
      stage = stage + 1 == NS ? 0 : stage + 1;
    }
    keepalive += r[0];
  }
  if (keepalive.x == 12345.678f) sink[tid] = keepalive.x + keepalive.y + keepalive.z + keepalive.w;
}

template <int MODE, bool READS>
static void run(const float* src, float* sink, int wg_per_cu, int iters, double clk_ghz, int cus) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = cus * wg_per_cu;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<MODE, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((fill_kernel<MODE, READS>), dim3(grid), dim3(256), NS * STAGE, 0, src, sink, iters);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
  }
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)grid * iters * STAGE;
  printf("mode %d (%s)%s, %d workgroup(s)/CU: %7.1f us, %6.2f TB/s chip, %5.1f B/clk/CU at %.2f GHz\n", MODE,
         MODE == 0 ? "LDS-DMA" : "registers", READS ? " + 12 ds_read_b128/wave" : "", wg_per_cu, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
         bytes / cus / (ms * 1e-3 * clk_ghz * 1e9), clk_ghz);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e-6;  // kHz -> GHz (peak engine clock: the achieved clock under load is lower)
  float *src, *sink;
  CK(hipMalloc(&src, SRC_BYTES));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(src, 0, SRC_BYTES));
  const int iters = 2000;
  for (int wg = 1; wg <= 2; ++wg) {
    run<0, false>(src, sink, wg, iters, clk, cus);
    run<1, false>(src, sink, wg, iters, clk, cus);
    run<0, true>(src, sink, wg, iters, clk, cus);
    run<1, true>(src, sink, wg, iters, clk, cus);
  }
  return 0;
}
