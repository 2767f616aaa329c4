// Shared device/host helpers for the PMCE hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PMCE_OK 0
#define PMCE_ERR_ARG (-1)
#define PMCE_ERR_LAUNCH (-2)
#define PMCE_ERR_WORKSPACE (-3)
#define PMCE_ERR_OVERFLOW (-4)

// host-side error slot (thread-local; pmce_last_error_string() returns it)
void pmce_set_error(const char* fmt, ...);
int pmce_check_launch(const char* what);
// Where the launchers of the calling thread point their kernels' "non-finite result" reports (a device-visible word, or null):
// set by the model entry points for the duration of a call (model.cpp), null for stand-alone operator calls.
unsigned* pmce_overflow_sink(void);
void pmce_set_overflow_sink(unsigned* device_visible_word);
// Likewise the clock probe of a model (pmce_model_set_clock_probe): two device words the split GEMM's workgroups add their residence
// to (shader clocks, 100 MHz ticks) while one of that model's entry points runs on this thread; null = off.
unsigned long long* pmce_clock_sink(void);
void pmce_set_clock_sink(unsigned long long* device_two_words);

// opt a kernel into > 64 KB of dynamic LDS, once per device (the attribute is per device: a process-wide flag would leave
// the second GPU of a process without it).  `done` is the call site's static bit mask of devices already set.
#ifdef __cplusplus
#include <atomic>
static inline int pmce_opt_in_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done, const char* what) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) d = 0;
  const unsigned long long bit = 1ull << d;
  if (done.load(std::memory_order_relaxed) & bit) return PMCE_OK;
  const hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (rc != hipSuccess) {
    pmce_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed: %s", what, bytes, hipGetErrorString(rc));
    return PMCE_ERR_LAUNCH;
  }
  done.fetch_or(bit, std::memory_order_relaxed);
  return PMCE_OK;
}
// integer environment knob, read once per process by its (function-local static) caller
int pmce_env_int(const char* name, int dflt);
#endif

#define PMCE_REQUIRE(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      pmce_set_error(__VA_ARGS__);             \
      return PMCE_ERR_ARG;                     \
    }                                          \
  } while (0)

#define PMCE_TRY(expr)              \
  do {                              \
    int _rc = (expr);               \
    if (_rc != PMCE_OK) return _rc; \
  } while (0)

// ---------------------------------------------------------------------------------------------
// "slot layout" of a 64-channel token held by a lane PAIR (l, l+32) of one wavefront
// ---------------------------------------------------------------------------------------------
// lane = (tok = lane & 31, hb = lane >> 5).  Register slot s in [0,32) of lane (tok,hb) holds channel
//     c = 8*(s>>2) + 4*hb + (s&3)
// i.e. the lane owns the eight 16-byte chunks {8q+4hb .. 8q+4hb+3}, q = 0..7, of the token's 256-byte
// row; the pair covers the row.  This is at once
//   * the B-operand order of v_mfma_f32_32x32x2_f32 for a K=64 contraction (step s uses slot s of both
//     halves: k-pair = {slot s of hb=0, slot s of hb=1}), and
//   * the C/D layout of a 32-row output tile whose rows are channels and whose columns are tokens
//     (row = (r&3) + 8*(r>>2) + 4*hb for accumulator register r; tile nt, register r == slot 16*nt+r),
// so the output of one token-local GEMM is directly the B operand of the next: no LDS round trip.
__device__ __forceinline__ int slot_channel(int s, int hb) { return 8 * (s >> 2) + 4 * hb + (s & 3); }

// The erf-GELU of the path (reference: nn.GELU() = x * Phi(x), exact-erf form), ONE formula for every kernel and every template
// instantiation.  fp32 VALU work costs matrix throughput on gfx950 (fp32 MFMA and fp32 VALU share the FMA lanes:
// scripts/microbench/mfma_valu.hip), and the decoder's FFN kernels are bound by their vector instructions, so the formula is counted
// in issue cycles:
//     gelu(x) = max(x, 0) - |x| * (erfc(|x| / sqrt 2) / 2),      erfc(|x| / sqrt 2) / 2 = 2^-(|x| * R(|x|) + 1)
// with R a degree-7 polynomial fitted to -log2(erfc(a / sqrt 2)) / a on [0, 6] (weighted so that the error of the GELU, not of R, is
// minimised; its leading coefficient is positive and a * R(a) keeps growing beyond the fit, so the tail term underflows to 0 for
// every larger |x| and no clamp is needed).  9 FMAs + v_max + one transcendental (v_exp_f32) = 56 issue cycles per wave instruction
// against the 84 of the form used until round 4 (Abramowitz-Stegun 7.1.26 on v_rcp + v_exp: 13 regular + 2 quarter-rate
// instructions), and closer to the exact function: max abs error 2.5e-7 over [-12, 12] = half an ulp of the result at |x| = 4.5
// (A-S: 3.8e-7; torch's own fp32 GELU sits 4.5e-7 from fp64), relative error of the negative tail 1.1e-5 instead of 2.2e-4 (no
// 1 - erf cancellation: the tail is computed directly).  +-inf gives NaN (inf * 0), where torch gives inf / NaN: non-finite in,
// non-finite out.  Fit and fp32 emulation: scripts/microbench/gelu_fit.py.
// Each fused multiply-add is spelled out and contraction is off, because hipcc's default (-ffp-contract=fast) fuses the SAME source
// differently from one instantiation to the next - the split GEMM's packed-output form once came out a 1-ulp different GELU than its
// fp32-output form, which shows as different (hi, lo) planes and breaks "the result does not depend on which kernel a batch size
// selects".
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
  const float a = fabsf(x);
  float p = fmaf(2.05025208e-06f, a, -3.00662196e-05f);
  p = fmaf(p, a, 0.000142456265f);
  p = fmaf(p, a, 0.000240916706f);
  p = fmaf(p, a, -0.00719628949f);
  p = fmaf(p, a, 0.0525850132f);
  p = fmaf(p, a, 0.459180683f);
  p = fmaf(p, a, 1.15110803f);
  const float t = fmaf(p, a, 1.0f);
  const float e = __builtin_amdgcn_exp2f(-t);  // Phi(-|x|)
  return fmaf(-a, e, fmaxf(x, 0.0f));
}
// two elements (the packed-fp32 form this once was is gone with -packed-fp32-ops)
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) { return f32x2{gelu_erf(x.x), gelu_erf(x.y)}; }

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float pair_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }

// One token-local GEMM step group: acc[nt] += W[nt*32 + n0][k-slots] * x[slots], K = 8*KQ.
// Wl: LDS, row-major [rows][LDW] floats with LDW = K + 4 (conflict-free ds_read_b128).
template <int KQ, int NT, int LDW>
__device__ __forceinline__ void tl_gemm(const float* __restrict__ Wl, const float* x, f32x16* acc, int n0, int hb) {
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(Wl + (nt * 32 + n0) * LDW + 8 * q + 4 * hb);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, x[4 * q + 0], acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, x[4 * q + 1], acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, x[4 * q + 2], acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, x[4 * q + 3], acc[nt], 0, 0, 0);
    }
  }
}

// Cooperative copy of a row-major [rows][K] global matrix into LDS [rows][K+4].  Four loads per thread are in flight before the first
// LDS write: left to itself hipcc emits load -> wait -> store per iteration, one exposed L2 round trip each (round 5: the un-batched
// copy of a 137 KB FFN image cost a workgroup 17 serial round trips, a quarter of the kernel).
template <int K>
__device__ __forceinline__ void stage_weight(float* __restrict__ dst, const float* __restrict__ src, int rows, int tid,
                                             int nthreads) {
  constexpr int C4 = K / 4;
  const int total = rows * C4;
  for (int i0 = tid; i0 < total; i0 += 4 * nthreads) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * nthreads;
      if (i < total) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(i / C4) * K + 4 * (i % C4));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * nthreads;
      if (i < total) *reinterpret_cast<f32x4*>(dst + (i / C4) * (K + 4) + 4 * (i % C4)) = v[u];
    }
  }
}

// LDS-DMA (buffer_load_dwordx4 ... lds): 64 lanes x 16 B from per-lane buffer offsets straight into LDS at M0 + lane * 16 - no VGPR round
// trip, no ds_write.  hipcc does not count these loads: whoever reads the destination first issues lds_dma_wait() (s_waitcnt vmcnt(0))
// and, across waves, a barrier.  (M0 is compiler-reserved: saved and restored inside the statement.)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_wave_base), "s"(soff)
      : "memory");
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Linear copy of `kib` KiB (16-byte aligned source, LDS destination) by LDS-DMA: every wave instruction moves 1 KiB, the waves take the
// 1 KiB pieces round-robin and issue all of theirs back to back (<= 63 per wave: the counter's range).  lds_dma_wait + barrier before use.
__device__ __forceinline__ void lds_dma_copy(float* lds_dst, const float* src, int kib, int wave, int nwaves, int lane) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0xffffffff, 0x00020000);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds_dst);
  const int w0 = __builtin_amdgcn_readfirstlane(wave);  // (wave-uniform by construction; the M0 operand must be scalar)
  for (int c = w0; c < kib; c += nwaves) lds_dma16(rsrc, (unsigned)(c * 1024 + lane * 16), 0, base + (unsigned)c * 1024u);
}

// Four consecutive channels c..c+3 (c % 4 == 0) of a row, written in the pre-split operand layout of the three-product f16 GEMM
// (gemm_split_f16.hip): per 16-wide k-tile 16 f16 "hi" then 16 f16 "lo * 2^11".  `row` is the row's start (the packed row takes
// the bytes of the fp32 row).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// A COMPUTED value that is about to be split into its (hi, lo) f16 planes goes through this first.  Under -ffp-contract=fast hipcc
// folds the arithmetic that produced v into each conversion on its own: hi = v_fma_mixlo_f16(a, b) (ONE rounding, of the exact
// product) in one place and v_cvt_f16_f32 of the fp32-rounded a*b in another.  When rne32(a*b) is an f16 tie the two disagree by
// an ulp, and a lo computed against the one is stored next to the other: an element off by 2^-10 relative, about one in 10^5
// (found by the matrix-pipe attention's test).  Pinned, v is ONE fp32 value for both planes.
__device__ __forceinline__ float pinned(float v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ void store4_split_f16(float* __restrict__ row, int c, const f32x4& v_) {
  f16x4 hi, lo;
  const f32x4 v = {pinned(v_[0]), pinned(v_[1]), pinned(v_[2]), pinned(v_[3])};
#pragma unroll
  for (int e = 0; e < 4; ++e) hi[e] = (_Float16)v[e];
#pragma unroll
  for (int e = 0; e < 4; ++e) lo[e] = (_Float16)((v[e] - (float)hi[e]) * 2048.0f);
  _Float16* p = reinterpret_cast<_Float16*>(row) + (c >> 4) * 32 + (c & 15);
  *reinterpret_cast<f16x4*>(p) = hi;
  *reinterpret_cast<f16x4*>(p + 16) = lo;
}

// AdaLayerNorm on a slot-layout token (reference CoevoDecoder.py:23-29): unbiased std, eps on the std.
// gb points at this clip's [gamma(64) | beta(64)] for the instance.
__device__ __forceinline__ void adaln_slots(const float* x, float* y, const float* __restrict__ gb, int hb) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
  const float mean = pair_sum(s) * (1.0f / 64.0f);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const float d = x[i] - mean;
    ss += d * d;
  }
  const float var = pair_sum(ss) * (1.0f / 63.0f);
  const float inv = 1.0f / (sqrtf(var) + 1e-6f);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gb + 8 * q + 4 * hb);
    const f32x4 b = *reinterpret_cast<const f32x4*>(gb + 64 + 8 * q + 4 * hb);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[4 * q + i] = g[i] * (x[4 * q + i] - mean) * inv + b[i];
  }
}

// load / store a 64-channel row in slot layout (8 x 16-byte chunks per lane)
__device__ __forceinline__ void load_slots(const float* __restrict__ row, float* x, int hb) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 8 * q + 4 * hb);
    x[4 * q + 0] = v.x;
    x[4 * q + 1] = v.y;
    x[4 * q + 2] = v.z;
    x[4 * q + 3] = v.w;
  }
}
__device__ __forceinline__ void store_slots(float* __restrict__ row, const float* x, int hb) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f32x4 v;
    v.x = x[4 * q + 0];
    v.y = x[4 * q + 1];
    v.z = x[4 * q + 2];
    v.w = x[4 * q + 3];
    *reinterpret_cast<f32x4*>(row + 8 * q + 4 * hb) = v;
  }
}
