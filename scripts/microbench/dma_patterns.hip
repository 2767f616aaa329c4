// What does an LDS-DMA instruction (buffer_load_dwordx4 ... lds: 64 lanes x 16 B) cost a CU, as a function of HOW MANY CACHE LINES it touches?
// The split GEMM's A operand and the GRU step's operands are fetched as 16 rows x 64 B per instruction, the rows one operand row apart
// (K x 4 bytes): 16 half lines.  A blocked layout makes the same 1 KB contiguous: 8 full lines.  This streams stages of Q instructions per
// wave (4 waves, F stages in flight, optional barrier per stage, no compute) from an L2-resident window and prints cycles per stage:
//   pattern c = contiguous 1 KB | h = 16 rows x 64 B, rows 2 KB apart (K = 512) | H = rows 8 KB apart (K = 2048) | p = 8 rows x 128 B, rows 8 KB apart
//   hipcc -O3 --offload-arch=gfx950 scripts/microbench/dma_patterns.hip -o /tmp/dma_patterns && /tmp/dma_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(lds_base), "s"(soff)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Q instructions per wave per stage, F stages in flight
template <int Q, int F, bool BARRIER>
__global__ __launch_bounds__(256) void stream(const float* __restrict__ src, int pattern, int n, long long wg_stride_bytes, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) float lds[(F + 1) * 4 * Q * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = reinterpret_cast<const char*>(src) + (long long)blockIdx.x * wg_stride_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xffffffff, 0x00020000);
  // per-instruction lane offsets and the per-stage advance
  unsigned voff[Q];
  int adv, wrap;
  for (int q = 0; q < Q; ++q) {
    const int g = wave * Q + q;  // row group of the stage
    if (pattern == 'c') { voff[q] = (unsigned)(g * 1024 + lane * 16); }
    else if (pattern == 'h') { voff[q] = (unsigned)((g * 16 + (lane >> 2)) * 2048 + (lane & 3) * 16); }
    else if (pattern == 'H') { voff[q] = (unsigned)((g * 16 + (lane >> 2)) * 8192 + (lane & 3) * 16); }
    else { voff[q] = (unsigned)((g * 8 + (lane >> 3)) * 8192 + (lane & 7) * 16); }
  }
  if (pattern == 'c') { adv = 4 * Q * 1024; wrap = 64; }
  else if (pattern == 'h') { adv = 64; wrap = 32; }
  else if (pattern == 'H') { adv = 64; wrap = 128; }
  else { adv = 128; wrap = 64; }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds) + wave * Q * 1024;
  auto issue = [&](int it) {
    const int soff = (it % wrap) * adv;
    const unsigned dst = lds0 + (it % (F + 1)) * (4 * Q * 1024);
#pragma unroll
    for (int q = 0; q < Q; ++q) dma16(rsrc, voff[q], soff, dst + q * 1024);
  };
  for (int it = 0; it < F; ++it) issue(it);
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
    wait_vm<(F - 1) * Q>();
    if (BARRIER) __syncthreads();
    issue(it + F);
  }
  wait_vm<0>();
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (tid == 0 && blockIdx.x == 0) out[0] = c1 - c0;
}

template <int Q, int F, bool BARRIER>
static void run(const float* src, unsigned long long* out, int pattern, int grid, const char* what) {
  const int n = 4000;
    unsigned long long h = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((stream<Q, F, BARRIER>), dim3(grid), dim3(256), 0, 0, src, pattern, n, 0ll, out);  // every workgroup reads the same L2-resident window
    (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  }
  const double cyc = (double)h / n;
  printf("%-44s Q=%d F=%d %s grid %3d: %7.1f cycles per stage of %2d KB -> %5.1f cycles per instruction, %5.1f B/clk per CU\n", what, Q, F,
         BARRIER ? "barrier" : "free   ", grid, cyc, 4 * Q, cyc / (4 * Q), 4.0 * Q * 1024 / cyc);
}

int main() {
  float* src; unsigned long long* out;
  (void)hipMalloc(&src, 1ll << 30); (void)hipMemset(src, 0, 1ll << 30);
  (void)hipMalloc(&out, 8);
  const struct { int p; const char* what; } pats[] = {{'c', "contiguous 1 KB (8 lines)"}, {'h', "16 rows x 64 B, 2 KB apart (16 half lines)"},
                                                     {'H', "16 rows x 64 B, 8 KB apart (16 half lines)"}, {'p', "8 rows x 128 B, 8 KB apart (8 lines)"}};
  for (int grid : {1, 24, 256, 512})
    for (auto& pt : pats) {
      run<3, 2, true>(src, out, pt.p, grid, pt.what);    // the 64x128 tile, one k-tile per stage, two in flight
      run<6, 2, true>(src, out, pt.p, grid, pt.what);    // the 128x256 tile / the 64x128 tile with two k-tiles per stage
      run<4, 7, false>(src, out, pt.p, grid, pt.what);   // the small-batch GRU step's private rings
      if (grid == 1) run<6, 5, true>(src, out, pt.p, grid, pt.what);
    }
  return 0;
}
