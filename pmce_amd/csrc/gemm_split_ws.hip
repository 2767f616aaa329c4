// Wave-specialised three-product f16 "NT" GEMM for gfx950 (arithmetic, operand layout and the per-accumulator order of the
// three products are those of gemm_split_f16.hip - results are bitwise equal to its tiles; reference: the pose lifter's
// Linear layers, PoseEstimation.py:13-29 via timm Attention / Mlp).
//
// Why a second kernel.  The lifter's products (M = 69,632 rows, K = 512 / 1024) are bound three ways at once: f16 matrix time,
// operand fill from L2 (a CU receives 35-39 B/clk) and the result stream to HBM (a 256-column tile row is stored at about the
// rate it is computed).  In the 4-wave kernel every wave issues LDS-DMA, matrix instructions AND the epilogue's stores; gfx950
// counts loads and stores in ONE in-order counter (vmcnt), so a wave's wait for the next k-tile also waits for the stores in
// front of it, and its DMA issue (60-180 cycles per instruction) sits in front of its own matrix instructions: fill, math and
// stores ran one after the other (profiles/r02_f_gemm_split_ablation.txt).  Here the roles are separate waves:
//   * 16 waves per workgroup, one workgroup per CU, <= 128 registers: waves 0-11 compute (3 x 4 waves of 64 x 64 -> a
//     192 x 256 tile: 25 B/clk of operand fill at full matrix rate instead of 31), waves 12-15 only load (one per SIMD).
//   * LOADER waves stream k-tiles (16 wide: 192 + 256 rows of 64 bytes = 28 pieces of 1 KB, 7 per loader) by LDS-DMA into a ring
//     of 5 stages, across tile boundaries, three batches in flight each; their vmcnt counts nothing but fills.
//   * COMPUTE waves never load in the k-loop and never wait on vmcnt there: a tile's 64 result stores per wave (or the residual
//     loads + stores) drain in the background of the next tile's k-loop.  Three compute waves share a SIMD's matrix pipe and
//     drift apart freely (one wave's epilogue under the other two's matrix instructions): there is NO workgroup barrier after
//     kernel start.  Hand-off is two LDS counters per stage: land[s] (+1 per loader once its pieces of the k-tile are in LDS;
//     a compute wave reads the stage at 4 x use) and rel[s] (+1 per compute wave once its fragments are in registers; a loader
//     refills the stage at 12 x use).  Counters only grow; every spin is bounded (a timed-out wave raises a device flag, read
//     by pmce_gemm_ws_timeouts, and leaves).
#include <atomic>

#include "gemm_split_common.hpp"

namespace {
constexpr int WS_BM = 192, WS_BN = 256;
constexpr int WS_NCW = 12, WS_NLW = 4;  // compute / loader waves
constexpr int WS_NS = 5;                // ring stages
constexpr int WS_STAGE_FLOATS = (WS_BM + WS_BN) * 16;
constexpr int WS_STAGE_BYTES = WS_STAGE_FLOATS * 4;
constexpr int WS_PPL = (WS_BM + WS_BN) / 16 / WS_NLW;  // DMA pieces (16 rows x 64 B) per loader per k-tile: 7
constexpr int WS_APL = WS_BM / 16 / WS_NLW;            // of which A's: 3
constexpr int WS_SLICE_OFF = WS_NS * WS_STAGE_FLOATS;  // two bias slices of 256 floats (this tile's, the next one's)
constexpr int WS_FLAG_OFF = WS_SLICE_OFF + 2 * 256;    // land[8], rel[8]
constexpr int WS_LDS_BYTES = (WS_FLAG_OFF + 16) * 4;
constexpr int WS_SPIN_LIMIT = 1 << 18;
}  // namespace

__device__ unsigned g_ws_timeouts;

// spin (bounded) until the LDS counter at byte address `addr` reaches `need`; wave-uniform
__device__ __forceinline__ bool ws_wait_ge(unsigned addr, unsigned need) {
  for (int spin = 0; spin < WS_SPIN_LIMIT; ++spin) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - need) >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
__device__ __forceinline__ void ws_signal(unsigned addr, int lane) {
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
}

template <int ACT, bool RES, bool OPACK>
__global__ __launch_bounds__(1024) void gemm_split_ws_kernel(SplitParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds;
  const unsigned land0 = lds0 + WS_FLAG_OFF * 4, rel0 = land0 + 32;
  if (tid < 16) reinterpret_cast<unsigned*>(lds)[WS_FLAG_OFF + tid] = 0u;
  __syncthreads();  // the only workgroup barrier

  // ---- persistent workgroups on an XCD-local chunk of the grouped tile order (as gemm_split_f16.hip) ----
  const int nblk = p.ntm * p.ntn;
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int cq = nblk >> 3, cr = nblk & 7;
  const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int chunk_len = cq + (xcd < cr ? 1 : 0);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.ntn;
  auto tile_coords = [&](int bid, int& mb, int& nb) {
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.ntm - first_m, GROUP_M);
    mb = (first_m + (bid % per_group) % gsz) * WS_BM;
    nb = ((bid % per_group) / gsz) * WS_BN;
  };
  const int nk = p.K / 16;

  if (wave >= WS_NCW) {
    // =========================================== loader waves ===========================================
    const int l = wave - WS_NCW;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =  // bounded: lanes past bias[N-1] read zeros
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    // piece g = l + 4 q of a stage holds rows 16 g .. 16 g + 15 (A rows first): lane L -> row 16 g + (L >> 2), PHYSICAL chunk
    // L & 3, which holds logical chunk (L & 3) ^ ((row >> 2) & 3) = (L & 3) ^ ((L >> 4) & 3)
    const int drow = lane >> 2;
    const unsigned dchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);  // bytes
    const unsigned lds_l = lds0 + l * 1024;
    unsigned doff[WS_PPL];
    int s = 0, par = 0;
    unsigned u = 0, g = 0;
    for (int li = bx; li < chunk_len; li += gx) {
      int mb, nb;
      tile_coords(chunk_start + li, mb, nb);
#pragma unroll
      for (int q = 0; q < WS_PPL; ++q) {
        const int gq = l + 4 * q;
        if (q < WS_APL)
          doff[q] = (unsigned)min(mb + 16 * gq + drow, p.M - 1) * (p.lda * 4u) + dchunk;
        else
          doff[q] = (unsigned)min(nb + 16 * (gq - WS_BM / 16) + drow, p.N - 1) * ((unsigned)p.K * 4u) + dchunk;
      }
      for (int kt = 0; kt < nk; ++kt) {
        if (u > 0 && !ws_wait_ge(rel0 + 4 * s, WS_NCW * u)) {  // every compute wave is done with the stage's previous k-tile
          atomicAdd(&g_ws_timeouts, 1u);
          return;
        }
        if (kt == 0 && l == 0) {  // the tile's bias slice rides in ahead of its first k-tile
          if (p.bias) sdma16(rsrc_b, (unsigned)lane * 16u, nb * 4, lds0 + (WS_SLICE_OFF + par * 256) * 4);
          par ^= 1;
        }
#pragma unroll
        for (int q = 0; q < WS_PPL; ++q)
          sdma16(q < WS_APL ? rsrc_a : rsrc_w, doff[q], kt * 64, lds_l + s * WS_STAGE_BYTES + q * 4096);
        if (g >= 2) {  // in-order completion: all but the youngest two batches have landed
          wait_vm<2 * WS_PPL>();
          ws_signal(land0 + 4 * (s >= 2 ? s - 2 : s + WS_NS - 2), lane);
        }
        ++g;
        if (++s == WS_NS) { s = 0; ++u; }
      }
    }
    if (g >= 2) {
      wait_vm<WS_PPL>();
      ws_signal(land0 + 4 * (s >= 2 ? s - 2 : s + WS_NS - 2), lane);
    }
    if (g >= 1) {
      wait_vm<0>();
      ws_signal(land0 + 4 * (s >= 1 ? s - 1 : WS_NS - 1), lane);
    }
    return;
  }

  // =========================================== compute waves ===========================================
  const int n0 = lane & 31, hb = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;  // 3 x 4 waves of 64 x 64
  const float w_up = p.wscale[0], w_down = p.wscale[1];
  const int swz = (n0 >> 2) & 3;
  const int a_row = (wm * 64 + n0) * 16, w_row = (WS_BM + wn * 64 + n0) * 16;  // floats inside a stage
  const int ch = 4 * (hb ^ swz), cl = 4 * ((2 + hb) ^ swz);                    // hi / lo plane, k = 8 hb + [0,8)
  int s = 0, par = 0;
  unsigned u = 0;
  f32x16 acc[2][2];
  for (int li = bx; li < chunk_len; li += gx) {
    int m_base, n_base;
    tile_coords(chunk_start + li, m_base, n_base);
    for (int kt = 0; kt < nk; ++kt) {
      if (!ws_wait_ge(land0 + 4 * s, WS_NLW * (u + 1))) {  // all four loaders' pieces of this k-tile are in LDS
        atomicAdd(&g_ws_timeouts, 1u);
        return;
      }
      const float* sA = lds + s * WS_STAGE_FLOATS;
      if (kt == 0) {  // the bias slice (scaled like W) is the accumulators' initial value
        const float* sB = lds + WS_SLICE_OFF + par * 256 + wn * 64 + n0;
        par ^= 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float bv = p.bias ? sB[j * 32] * w_up : 0.f;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
        }
      }
      f16x8 ahi[2], alo[2], whi[2], wlo[2], wh2[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ahi[i] = *reinterpret_cast<const f16x8*>(sA + a_row + i * 512 + ch);
        alo[i] = *reinterpret_cast<const f16x8*>(sA + a_row + i * 512 + cl);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        whi[j] = *reinterpret_cast<const f16x8*>(sA + w_row + j * 512 + ch);
        wlo[j] = *reinterpret_cast<const f16x8*>(sA + w_row + j * 512 + cl);
      }
      ws_signal(rel0 + 4 * s, lane);  // LDS serves a wave's requests in order: the fragments are read before this add lands
#pragma unroll
      for (int j = 0; j < 2; ++j) wh2[j] = whi[j] * (_Float16)0.00048828125f;  // 2^-11: undoes the scale of alo
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], whi[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], wlo[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], wh2[j], acc[i][j], 0, 0, 0);
      if (++s == WS_NS) { s = 0; ++u; }
    }

    // ---- epilogue of the wave's 64 x 64 part, straight from the accumulators (the arithmetic of gemm_split_f16.hip) ----
    const int wm0 = m_base + wm * 64, wn0 = n_base + wn * 64;
    // lane coordinates re-made opaque per tile: hipcc would otherwise hoist the epilogue's 16 lane offsets (and the row pointers
    // of the edge path) out of the tile loop and spill them across the k-loop
    int n0e = n0, hbe = hb;
    asm volatile("" : "+v"(n0e), "+v"(hbe));
    if constexpr (OPACK) {
      // The result is the A operand of the next product (fc1 -> fc2): written pre-split, [row][K/16][16 hi | 16 lo*2^11] f16 in
      // the bytes of the fp32 row; adjacent lanes pair up (DPP) so that every lane stores one dword per element.
      const bool odd = n0e & 1;
      const int colf = (n0e >> 4) * 32 + (odd ? 16 + ((n0e - 1) & 15) : (n0e & 15));  // f16 index inside the 32-column group
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb = wn0 + j * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mrow = wm0 + i * 32 + 4 * hbe;
          _Float16* __restrict__ Cp = reinterpret_cast<_Float16*>(p.C) + (size_t)mrow * (2 * p.ldc) + 2 * cb + colf;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 v = {acc[i][j][r] * w_down, acc[i][j][r + 1] * w_down};
            if (ACT == 1) v = gelu_erf2(v);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float x = e ? v.y : v.x;
              const _Float16 h = (_Float16)x;
              const _Float16 lo = (_Float16)((x - (float)h) * 2048.0f);
              const unsigned w = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
              const unsigned nbr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, true);  // lane ^ 1
              const unsigned outw = odd ? ((nbr >> 16) | (w & 0xffff0000u)) : ((w & 0xffffu) | (nbr << 16));
              const int rr = ((r + e) & 3) + 8 * ((r + e) >> 2);
              if (mrow + rr < p.M && cb < p.N) __builtin_nontemporal_store(outw, reinterpret_cast<unsigned*>(Cp + (size_t)rr * (2 * p.ldc)));
            }
          }
        }
      }
      continue;
    }
    const bool full = (wm0 + 64 <= p.M) && (wn0 + 64 <= p.N);
    if (full) {
      // buffer-form accesses: one set of 16 lane offsets serves the four 32x32 blocks and both R and C (the block's position
      // is the scalar offset); the residual of block b + 1 is requested before block b is stored
      const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(p.C + (size_t)wm0 * p.ldc, 0, 0xffffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrc_r =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.R + (size_t)wm0 * p.ldc : p.C), 0, 0xffffffff, 0x00020000);
      unsigned voff[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) voff[r] = ((unsigned)(4 * hbe + (r & 3) + 8 * (r >> 2)) * p.ldc + (unsigned)n0e) * 4u;
      auto blk_off = [&](int b) __attribute__((always_inline)) {  // wave-uniform
        return ((unsigned)((b % 2) * 32) * p.ldc + (unsigned)(wn0 + (b / 2) * 32)) * 4u;
      };
      float rv[2][16];
      auto res_load = [&](int b, float (&dst)[16]) __attribute__((always_inline)) {
        const unsigned so = blk_off(b);
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, voff[r], so, 0));
      };
      if (RES) res_load(0, rv[0]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = b / 2, i = b % 2;
        if (RES && b + 1 < 4) res_load(b + 1, rv[(b + 1) & 1]);
        const unsigned so = blk_off(b);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 v = {acc[i][j][r] * w_down, acc[i][j][r + 1] * w_down};
          if (ACT == 1) v = gelu_erf2(v);
          if (RES) v += f32x2{rv[b & 1][r], rv[b & 1][r + 1]};
          const float vx = v.x, vy = v.y;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), rsrc_c, voff[r], so, 2);  // aux 2 = nt
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), rsrc_c, voff[r + 1], so, 2);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = wn0 + j * 32 + n0e;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mrow = wm0 + i * 32 + 4 * hbe;
          float* __restrict__ Cp = p.C + (size_t)mrow * p.ldc + n;
          const float* __restrict__ Rp = RES ? p.R + (size_t)mrow * p.ldc + n : nullptr;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {  // the same arithmetic as the full path (results do not depend on the tile shape)
            f32x2 v = {acc[i][j][r] * w_down, acc[i][j][r + 1] * w_down};
            if (ACT == 1) v = gelu_erf2(v);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int rr = ((r + e) & 3) + 8 * ((r + e) >> 2);
              if (n < p.N && mrow + rr < p.M) {
                float o = e ? v.y : v.x;
                if (RES) o += Rp[rr * p.ldc];
                Cp[rr * p.ldc] = o;
              }
            }
          }
        }
      }
    }
  }
}

// ---- launch ---------------------------------------------------------------------------------------------------------------
template <int ACT, bool RES, bool OPACK>
static int ws_launch_one(const SplitParams& p, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> done{0};
  PMCE_TRY(pmce_opt_in_lds(reinterpret_cast<const void*>(&gemm_split_ws_kernel<ACT, RES, OPACK>), WS_LDS_BYTES, done, "gemm_split_ws"));
  hipLaunchKernelGGL((gemm_split_ws_kernel<ACT, RES, OPACK>), dim3(grid), dim3(1024), WS_LDS_BYTES, stream, p);
  return PMCE_OK;
}

// Is the wave-specialised kernel applicable to (and worth it for) this product?  Pre-split A only (the lifter blocks' products),
// no row map, and enough 192 x 256 tiles to put one workgroup on most CUs.
bool pmce_gemm_split_ws_wants(int M, int N, int K, int a_packed, int c_div) {
  if (!a_packed || c_div != 0 || K < 128) return false;
  const long long tiles = (long long)((M + WS_BM - 1) / WS_BM) * ((N + WS_BN - 1) / WS_BN);
  return tiles >= 192;
}

int pmce_gemm_split_ws_launch(SplitParams& p, int act, int c_packed, hipStream_t stream) {
  p.ntm = (p.M + WS_BM - 1) / WS_BM;
  p.ntn = (p.N + WS_BN - 1) / WS_BN;
  int g = p.ntm * p.ntn;
  if (g > 256) g = 256;
  g = (g + 7) & ~7;
  const bool res = p.R != nullptr;
  if (c_packed) return ws_launch_one<1, false, true>(p, g, stream);
  if (act == 1) return res ? ws_launch_one<1, true, false>(p, g, stream) : ws_launch_one<1, false, false>(p, g, stream);
  return res ? ws_launch_one<0, true, false>(p, g, stream) : ws_launch_one<0, false, false>(p, g, stream);
}

// number of waves that gave up on a hand-off since the library was loaded (0 in a healthy process); synchronises the device
extern "C" int pmce_gemm_ws_timeouts(void) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ws_timeouts), sizeof(v)) != hipSuccess) return -1;
  return (int)v;
}
