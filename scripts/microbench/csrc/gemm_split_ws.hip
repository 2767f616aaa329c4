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
#include <type_traits>

#include "gemm_split_common.hpp"

namespace {
constexpr int WS_BM = 192, WS_BN = 256;
constexpr int WS_NCW = 12, WS_NLW = 4;  // compute / loader waves
#ifndef WS_NS_OVERRIDE
#define WS_NS_OVERRIDE 5
#endif
constexpr int WS_NS = WS_NS_OVERRIDE;  // ring stages
constexpr int WS_STAGE_FLOATS = (WS_BM + WS_BN) * 16;
constexpr int WS_STAGE_BYTES = WS_STAGE_FLOATS * 4;
constexpr int WS_PPL = (WS_BM + WS_BN) / 16 / WS_NLW;  // DMA pieces (16 rows x 64 B) per loader per k-tile: 7
constexpr int WS_APL = WS_BM / 16 / WS_NLW;            // of which A's: 3
constexpr int WS_SLICE_OFF = WS_NS * WS_STAGE_FLOATS;  // two {bias, 2^-s} slice pairs of 256 floats each (this tile's, the next one's)
constexpr int WS_FLAG_OFF = WS_SLICE_OFF + 4 * 256;    // land[4] (16 words), prog[12] (16 words), then a 256-byte dump slot
constexpr int WS_LDS_BYTES = (WS_FLAG_OFF + 32 + 64) * 4;
constexpr int WS_SPIN_LIMIT = 1 << 18;
constexpr int WS_DEFAULT_OPT = 9;  // measured best: no dynamic priorities, loaders at normal priority (profiles/r03_a_*)
}  // namespace

__device__ unsigned g_ws_timeouts;
#ifdef PMCE_WS_ABLATE
__device__ unsigned long long g_ws_stat[4];  // failed polls of compute waves / of loaders, compute k-tiles that had to poll at all
#define WS_STAT(i, n) atomicAdd(&g_ws_stat[i], (unsigned long long)(n))
extern "C" int pmce_gemm_ws_stats(unsigned long long* out4, int reset) {
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_ws_stat), 32) != hipSuccess) return -1;
  if (reset) {
    const unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stat), z, 32) != hipSuccess) return -1;
  }
  return 0;
}
__device__ unsigned long long g_ws_clk[2];  // shader clocks / 100 MHz ticks summed over the workgroups' first waves
extern "C" int pmce_gemm_ws_clk(unsigned long long* out2, int reset) {
  if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_ws_clk), 16) != hipSuccess) return -1;
  if (reset) {
    const unsigned long long z[2] = {0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ws_clk), z, 16) != hipSuccess) return -1;
  }
  return 0;
}
__device__ unsigned long long g_ws_prof[16][4];  // DBG & 128: shader clocks per wave role: see the kernel
extern "C" int pmce_gemm_ws_prof(unsigned long long* out64, int reset) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ws_prof), 512) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[64] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ws_prof), z, 512) != hipSuccess) return -1;
  }
  return 0;
}
#define WS_CLK() ((DBG & 128) ? (long long)__builtin_readcyclecounter() : 0ll)
#else
#define WS_STAT(i, n) ((void)0)
#define WS_CLK() 0ll
#endif

// Hand-off words in LDS: every wave owns ONE monotonic progress word (no atomics): loader l writes land[l] = number of k-tiles
// whose pieces from l are in LDS, compute wave c writes prog[c] = number of k-tiles whose fragments it holds in registers.  A
// reader fetches all words of a kind with one ds_read_b32 (lane i reads word i % n) and proceeds when every lane sees >= need.
__device__ __forceinline__ void ws_post(unsigned addr, unsigned value, int lane) {
  if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(value) : "memory");
}
__device__ __forceinline__ bool ws_all_ge(unsigned v, unsigned need) { return __builtin_amdgcn_ballot_w64((int)(v - need) < 0) == 0ull; }
// A wave-uniform pointer the compiler can SEE is uniform (else every buffer access on a descriptor built from it is wrapped in a
// waterfall loop)
template <typename T>
__device__ __forceinline__ T* ws_uniform(T* q) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
// Spin (bounded) until every word is >= need; wave-uniform.  A wave that gives up raises the device flag, marks itself dead and
// stops waiting for anything: the launch then ends with wrong results (and a non-zero pmce_gemm_ws_timeouts) instead of hanging.
__device__ __forceinline__ void ws_wait_all(unsigned lane_addr, unsigned need, int& dead, int& fails) {
  if (dead) return;
  for (int spin = 0; spin < WS_SPIN_LIMIT; ++spin) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lane_addr) : "memory");
    if (ws_all_ge(v, need)) return;
    ++fails;
    __builtin_amdgcn_s_sleep(1);
  }
  dead = 1;
  atomicAdd(&g_ws_timeouts, 1u);
}

// DBG (timing-only ablations, results invalid; scripts/microbench/gemm_ws.py --ablate): 1 no result stores, 2 loaders issue no
// DMA, 4 no matrix instructions, 8 DMA sources folded into a 4 KB window (L1 hits), 16 compute waves do not wait for land[],
// 32 / 64 only A's / only W's sources folded.  QD = DMA batches (k-tiles) a loader keeps in flight.
// OPT (schedule options for A/B runs, results unaffected): 1 no dynamic priority between the compute waves of a SIMD, 8 loaders not
// at priority 3.
template <int ACT, bool RES, bool OPACK, int DBG = 0, int QD = 3, int OPT = WS_DEFAULT_OPT>
__global__ __launch_bounds__(1024) void gemm_split_ws_kernel(SplitParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds;
  const unsigned land0 = lds0 + WS_FLAG_OFF * 4, prog0 = land0 + 64;  // land[4] | prog[12]
  if (tid < 32) reinterpret_cast<unsigned*>(lds)[WS_FLAG_OFF + tid] = 0u;
  __syncthreads();  // the only workgroup barrier
#ifdef PMCE_WS_ABLATE
  const long long k_c0 = (long long)__builtin_readcyclecounter(), k_w0 = (long long)wall_clock64();  // shader clocks, 100 MHz ticks
#endif

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
  int dead = 0, fails = 0, polled = 0;
  long long pt[4] = {0, 0, 0, 0};  // DBG & 128.  loader: wait for a free stage | DMA issue | wait landed + post | k-tiles
                                   // compute: wait for land | fragment reads until in registers | matrix issue | epilogue
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0xffffffff, 0x00020000);

  if (wave >= WS_NCW) {
    // =========================================== loader waves ===========================================
    const int l = wave - WS_NCW;
    if constexpr ((OPT & 8) == 0) __builtin_amdgcn_s_setprio(3);  // (A/B option: DMA issue ahead of everything - no gain)
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =  // bounded: lanes past bias[N-1] read zeros
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wscale), 0, p.N * 4, 0x00020000);
    // piece g = l + 4 q of a stage holds rows 16 g .. 16 g + 15 (A rows first): lane L -> row 16 g + (L >> 2), PHYSICAL chunk
    // L & 3, which holds logical chunk (L & 3) ^ ((row >> 2) & 3) = (L & 3) ^ ((L >> 4) & 3)
    const int drow = lane >> 2;
    const unsigned dchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);  // bytes
    const unsigned lds_l = lds0 + l * 1024;
    const unsigned prog_lane = prog0 + 4 * (lane < WS_NCW ? lane : 0);
    const unsigned my_land = land0 + 4 * l;
    unsigned doff[WS_PPL];
    int s = 0, par = 0;
    unsigned g = 0;  // k-tiles issued
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
        if constexpr ((DBG & 8) != 0) doff[q] &= 0xff0u;
        if constexpr ((DBG & 32) != 0) if (q < WS_APL) doff[q] &= 0xff0u;
        if constexpr ((DBG & 64) != 0) if (q >= WS_APL) doff[q] &= 0xff0u;
      }
      for (int kt = 0; kt < nk; ++kt) {
        const long long c0 = WS_CLK();
        if (g >= (unsigned)WS_NS) ws_wait_all(prog_lane, g - WS_NS + 1, dead, fails);  // every compute wave holds the stage's previous k-tile
        const long long c1 = WS_CLK();
        if (kt == 0 && l == 0) {  // the tile's bias and 2^-s slices ride in ahead of its first k-tile
          if (p.bias) sdma16(rsrc_b, (unsigned)lane * 16u, nb * 4, lds0 + (WS_SLICE_OFF + par * 512) * 4);
          sdma16(rsrc_s, (unsigned)lane * 16u, nb * 4, lds0 + (WS_SLICE_OFF + par * 512 + 256) * 4);
          par ^= 1;
        }
        if constexpr ((DBG & 2) == 0) {
#pragma unroll
          for (int q = 0; q < WS_PPL; ++q) {
            const bool fold = (DBG & 8) || ((DBG & 32) && q < WS_APL) || ((DBG & 64) && q >= WS_APL);
            sdma16(q < WS_APL ? rsrc_a : rsrc_w, doff[q], fold ? 0 : kt * 64, lds_l + s * WS_STAGE_BYTES + q * 4096);
          }
        }
        ++g;
        const long long c2 = WS_CLK();
        if (g >= (unsigned)QD) {  // in-order completion: all but the youngest QD - 1 batches have landed
          wait_vm<(QD - 1) * WS_PPL>();
          ws_post(my_land, g - (QD - 1), lane);
        }
        if constexpr ((DBG & 128) != 0) {
          const long long c3 = WS_CLK();
          pt[0] += c1 - c0; pt[1] += c2 - c1; pt[2] += c3 - c2; pt[3] += 1;
        }
        if (++s == WS_NS) s = 0;
      }
    }
    wait_vm<0>();
    ws_post(my_land, g, lane);
    if (lane == 0) WS_STAT(1, fails);
#ifdef PMCE_WS_ABLATE
    if constexpr ((DBG & 128) != 0)
      if (lane == 0) for (int i = 0; i < 4; ++i) atomicAdd(&g_ws_prof[wave][i], (unsigned long long)pt[i]);
#endif
    return;
  }

  // =========================================== compute waves ===========================================
  // Per k-tile: the eight fragment reads (hi planes first), the progress word, the read of land[] for the NEXT k-tile - then the
  // twelve matrix instructions, which hipcc releases by counted waits as their operands arrive (hi x hi after the first four
  // reads).  The reads stay compiler-visible loads: fragments read by inline asm would be waited for correctly, but hipcc is
  // free to COPY a register an asm load has not yet filled (it did, at a join of two paths) - an asm load belongs in one
  // statement with its wait.  Nothing in the k-tile is a branch except the rare wait for a late k-tile, so that the loads and
  // their uses share a basic block (across blocks hipcc falls back to lgkmcnt(0)).
  const int n0 = lane & 31, hb = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;  // 3 x 4 waves of 64 x 64
  float w_down[2];  // 2^-s of this lane's column in the wave's two 32-column blocks
  const int swz = (n0 >> 2) & 3;
  const int a_row = (wm * 64 + n0) * 16, w_row = (WS_BM + wn * 64 + n0) * 16;  // floats inside a stage
  const int ch = 4 * (hb ^ swz);  // the hi plane's chunk of k = 8 hb + [0,8); the lo plane's is ch ^ 8, i.e. byte address ^ 32
  const unsigned land_lane = land0 + 4 * (lane & 3);
  const unsigned my_prog = prog0 + 4 * wave;
  // The word every lane reads beside the fragments: land[lane & 3] - except lanes 4..6, which read the progress words of the three
  // compute waves that share this wave's SIMD (waves w, w + 4, w + 8 of a workgroup land on one SIMD).  The arbiter serves the
  // oldest wave first: left alone, a SIMD's oldest compute wave runs 3 k-tiles ahead and then spins on land[], while the youngest
  // - behind in matrix issue, LDS and store issue alike - sets the pace of the tile (k-tile 1,880 cycles for it against 710 of
  // work for the oldest).  So each wave raises its priority by one for every SIMD mate that is AHEAD of it (as of the previous
  // k-tile): the three stay within a k-tile of each other and nobody waits at the ring's end.
  const unsigned peek_addr = (lane >= 4 && lane <= 6) ? prog0 + 4 * ((wave & 3) + 4 * (lane - 4)) : land_lane;
  constexpr unsigned long long LAND_LANES = ~0x70ull;
  int s = 0, par = 0, prio = 0;
  unsigned g = 0;     // k-tiles consumed
  unsigned peek = 0;  // land[] (and the mates' progress) as read during the previous k-tile
  f32x16 acc[2][2];
  for (int li = bx; li < chunk_len; li += gx) {
    int m_base, n_base;
    tile_coords(chunk_start + li, m_base, n_base);
    for (int kt = 0; kt < nk; ++kt) {
      const long long c0 = WS_CLK();
      if constexpr ((DBG & 16) == 0) {  // all four loaders' pieces of this k-tile are in LDS
        if ((__builtin_amdgcn_ballot_w64((int)(peek - (g + 1)) < 0) & LAND_LANES) != 0ull) {  // (peek = 0 before the first k-tile)
          polled += 1;
          ws_wait_all(land_lane, g + 1, dead, fails);
        }
        if constexpr ((OPT & 1) == 0) {
          const int mine = __builtin_amdgcn_readlane((int)peek, 4 + (wave >> 2));
          const int behind = (int)(__builtin_amdgcn_readlane((int)peek, 4 + ((wave >> 2) + 1) % 3) - mine > 0) +
                             (int)(__builtin_amdgcn_readlane((int)peek, 4 + ((wave >> 2) + 2) % 3) - mine > 0);
          if (behind != prio) {
            prio = behind;
            if (behind == 0) __builtin_amdgcn_s_setprio(0);
            else if (behind == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
          }
        }
      }
      const long long c1 = WS_CLK();
      if (kt == 0) {  // the bias slice (scaled like W) is the accumulators' initial value
        const float* sB = lds + WS_SLICE_OFF + par * 512 + wn * 64 + n0;
        par ^= 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          w_down[j] = sB[256 + j * 32];
          const float bv = p.bias ? sB[j * 32] * pow2_recip(w_down[j]) : 0.f;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
        }
      }
      f16x8 ahi[2], alo[2], whi[2], wlo[2];
      ++g;
      {
        // ONE statement: the eight fragment reads (hi planes first), the progress word (lane 0 only, EXEC narrowed inside the
        // statement), land[] for the next k-tile, and the wait for the hi fragments.  LDS serves a wave's requests in order, so
        // the fragment reads are performed before the progress word changes, and lgkmcnt(6) = "all but the six youngest of these
        // ten operations are done".  The lo fragments and land[] are still in flight when the statement ends: until the second
        // statement below nothing but the four hi x hi matrix instructions may sit between (hipcc copies or spills a register
        // whenever it likes - an asm load it cannot see is safe only inside such a fenced straight line; the device code is
        // checked for exactly that by tests/test_host_logic.py).
        const unsigned pa = lds0 + (unsigned)(s * WS_STAGE_FLOATS + a_row + ch) * 4u, pw = lds0 + (unsigned)(s * WS_STAGE_FLOATS + w_row + ch) * 4u;
        unsigned long long keep;
        asm volatile(
            "ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:2048\n\tds_read_b128 %2, %11\n\tds_read_b128 %3, %11 offset:2048\n\t"
            "ds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:2048\n\tds_read_b128 %6, %12\n\tds_read_b128 %7, %12 offset:2048\n\t"
            "s_mov_b64 %8, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %14, %15\n\ts_mov_b64 exec, %8\n\t"
            "ds_read_b32 %9, %16\n\ts_waitcnt lgkmcnt(6)"
            : "=&v"(ahi[0]), "=&v"(ahi[1]), "=&v"(whi[0]), "=&v"(whi[1]), "=&v"(wlo[0]), "=&v"(wlo[1]), "=&v"(alo[0]), "=&v"(alo[1]),
              "=&s"(keep), "=&v"(peek)
            : "v"(pa), "v"(pw), "v"(pa ^ 32u), "v"(pw ^ 32u), "v"(my_prog), "v"(g), "v"(peek_addr)
            : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((DBG & 4) == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], whi[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wlo[0]), "+v"(wlo[1]), "+v"(alo[0]), "+v"(alo[1]), "+v"(peek)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((DBG & 4) == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], wlo[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) whi[j] = whi[j] * (_Float16)0.00048828125f;  // 2^-11: undoes the scale of alo
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], whi[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(ahi[i]), "v"(alo[i]), "v"(whi[i]), "v"(wlo[i]));
      }
      if constexpr ((DBG & 128) != 0) {
        __builtin_amdgcn_sched_barrier(0);
        const long long c3 = WS_CLK();
        pt[0] += c1 - c0; pt[2] += c3 - c1;
      }
      if (++s == WS_NS) s = 0;
    }
    const long long ce0 = WS_CLK();

    // ---- epilogue of the wave's 64 x 64 part, straight from the accumulators (the arithmetic of gemm_split_f16.hip): a lane
    // holds one column, a store instruction writes two full 128-byte lines.  Measured and dropped here (profiles/r03_*): a 4 x 4
    // transpose inside lane quads (DPP) for dwordx4 stores of full lines - its ~320 vector instructions per tile starve on the
    // younger waves (epilogue 4.8 - 19 k cycles per tile instead of 3.2 - 7.7 k); the matrix instructions with W as first
    // operand (a lane then holds a row of C: dwordx4 stores with no transpose, but of 32-byte pieces of 32 different lines: 2 x
    // slower overall).  NOTE for dwordx4 buffer stores on gfx950: a vector instruction overwriting the data registers right
    // behind the store corrupts it also when the SCALAR offset is a register - hipcc pads that hazard only for a literal offset. ----
    if constexpr ((DBG & 1) != 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
      continue;
    }
    const int wm0 = m_base + wm * 64, wn0 = n_base + wn * 64;
    if (wn0 >= p.N || wm0 >= p.M) continue;  // (N % 64 == 0: a wave's 64 columns are all valid or all past the matrix)
    // lane coordinates re-made opaque per tile: hipcc would otherwise hoist the epilogue's 16 lane offsets out of the tile loop and
    // spill them across the k-loop
    int n0e = n0, hbe = hb;
    asm volatile("" : "+v"(n0e), "+v"(hbe));
    const int rows_left = p.M - wm0 - 4 * hbe;  // row (r & 3) + 8 (r >> 2) of block i is valid iff 32 i + that < rows_left
    const bool full = wm0 + 64 <= p.M;           // wave-uniform
    float* const c_base = ws_uniform(p.C + (size_t)wm0 * p.ldc);
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(c_base, 0, 0xffffffff, 0x00020000);
    // (FULL = the wave's 64 rows all exist; otherwise - the matrix's last row tile only - every access is predicated on its row)
    bool bad = false;  // this lane produced a non-finite value (an operand beyond the f16 range, or fp32 overflow)
    auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_tag)::value;
      auto row_ok = [&](int i, int r) __attribute__((always_inline)) { return FULL || 32 * i + (r & 3) + 8 * (r >> 2) < rows_left; };
      if constexpr (OPACK) {
        // The result is the A operand of the next product (fc1 -> fc2): written pre-split, [row][K/16][16 hi | 16 lo*2^11] f16 in
        // the bytes of the fp32 row; adjacent lanes pair up (DPP) so that every lane stores one dword per element: even lanes
        // {hi(n), hi(n+1)}, odd lanes {lo(n-1), lo(n)}.
        const bool odd = n0e & 1;
        const int colf = (n0e >> 4) * 32 + (odd ? 16 + ((n0e - 1) & 15) : (n0e & 15));  // f16 index inside the 32-column group
        unsigned voff[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) voff[r] = (unsigned)(4 * hbe + (r & 3) + 8 * (r >> 2)) * p.ldc * 4u + (unsigned)colf * 2u;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j = b / 2, i = b % 2;
          const unsigned so = ((unsigned)(i * 32) * p.ldc + (unsigned)(wn0 + j * 32)) * 4u;  // (a packed row takes the fp32 row's bytes)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 v = {acc[i][j][r] * w_down[j], acc[i][j][r + 1] * w_down[j]};
            if (ACT == 1) v = gelu_erf2(v);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float x = pinned(e ? v.y : v.x);
              const _Float16 h = (_Float16)x;
              bad = bad || nonfinite((float)h);  // also a finite x beyond f16's 65504: the next product could not read it
              const _Float16 lo = (_Float16)((x - (float)h) * 2048.0f);
              const unsigned w = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
              const unsigned nbr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, true);  // lane ^ 1
              const unsigned outw = odd ? ((nbr >> 16) | (w & 0xffff0000u)) : ((w & 0xffffu) | (nbr << 16));
              if (row_ok(i, r + e)) __builtin_amdgcn_raw_buffer_store_b32(outw, rsrc_c, voff[r + e], so, 2);
            }
          }
        }
      } else {
        // one set of 16 lane offsets serves the four 32x32 blocks and both R and C (the block's position is the scalar offset);
        // the residual of block b + 1 is requested before block b is stored
        const float* const r_base = ws_uniform(RES ? p.R + (size_t)wm0 * p.ldc : p.C);
        const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(r_base), 0, 0xffffffff, 0x00020000);
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
          for (int r = 0; r < 16; ++r) {
            if (!FULL) dst[r] = 0.f;
            if (row_ok(b % 2, r)) dst[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, voff[r], so, 0));
          }
        };
        if (RES) res_load(0, rv[0]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j = b / 2, i = b % 2;
          if (RES && b + 1 < 4) res_load(b + 1, rv[(b + 1) & 1]);
          const unsigned so = blk_off(b);
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 v = {acc[i][j][r] * w_down[j], acc[i][j][r + 1] * w_down[j]};
            if (ACT == 1) v = gelu_erf2(v);
            if (RES) v += f32x2{rv[b & 1][r], rv[b & 1][r + 1]};
            const float vx = v.x, vy = v.y;
            bad = bad || nonfinite(vx) || nonfinite(vy);
            if constexpr ((DBG & 256) != 0) if (b == 3 && r >= 14) continue;  // timing experiment: 62 stores per tile
            if constexpr ((DBG & 512) != 0) if (b == 3 && r >= 8) continue;   // timing experiment: 56 stores per tile
            if (row_ok(i, r)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), rsrc_c, voff[r], so, 2);  // aux 2 = nt
            if (row_ok(i, r + 1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), rsrc_c, voff[r + 1], so, 2);
          }
        }
      }
    };
    if (full) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    report_nonfinite(p.oflow, bad);
    if constexpr ((DBG & 128) != 0) pt[3] += WS_CLK() - ce0;
  }
  (void)polled;
  if (lane == 0) {
    WS_STAT(0, fails);
    WS_STAT(2, polled);
  }
#ifdef PMCE_WS_ABLATE
  if (tid == 0) {  // effective shader clock of this workgroup's CU over the launch
    atomicAdd(&g_ws_clk[0], (unsigned long long)((long long)__builtin_readcyclecounter() - k_c0));
    atomicAdd(&g_ws_clk[1], (unsigned long long)((long long)wall_clock64() - k_w0));
  }
#endif
#ifdef PMCE_WS_ABLATE
  if constexpr ((DBG & 128) != 0)
    if (lane == 0) for (int i = 0; i < 4; ++i) atomicAdd(&g_ws_prof[wave][i], (unsigned long long)pt[i]);
#endif
}

// ---- launch ---------------------------------------------------------------------------------------------------------------
template <int ACT, bool RES, bool OPACK, int DBG = 0, int QD = 3, int OPT = WS_DEFAULT_OPT>
static int ws_launch_one(const SplitParams& p, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> done{0};
  PMCE_TRY(pmce_opt_in_lds(reinterpret_cast<const void*>(&gemm_split_ws_kernel<ACT, RES, OPACK, DBG, QD, OPT>), WS_LDS_BYTES, done, "gemm_split_ws"));
  hipLaunchKernelGGL((gemm_split_ws_kernel<ACT, RES, OPACK, DBG, QD, OPT>), dim3(grid), dim3(1024), WS_LDS_BYTES, stream, p);
  return PMCE_OK;
}
#ifdef PMCE_WS_ABLATE
static std::atomic<int> g_ws_dbg{0};
extern "C" int pmce_gemm_ws_set_dbg(int v) {
  g_ws_dbg.store(v, std::memory_order_relaxed);
  return PMCE_OK;
}
#endif

// Is the wave-specialised kernel applicable to (and worth it for) this product?  Pre-split A only (the lifter blocks' products),
// no row map, and enough 192 x 256 tiles to put one workgroup on most CUs.
bool pmce_gemm_split_ws_wants(int M, int N, int K, int a_packed, int c_div) {
  if (!a_packed || c_div != 0 || K < 128 || K % 32 != 0 || N % 64 != 0) return false;
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
#ifdef PMCE_WS_ABLATE
  if (!c_packed && act == 0 && !res) switch (g_ws_dbg.load(std::memory_order_relaxed)) {
      case 1: return ws_launch_one<0, false, false, 1>(p, g, stream);
      case 2: return ws_launch_one<0, false, false, 2>(p, g, stream);
      case 3: return ws_launch_one<0, false, false, 3>(p, g, stream);
      case 4: return ws_launch_one<0, false, false, 4>(p, g, stream);
      case 5: return ws_launch_one<0, false, false, 5>(p, g, stream);
      case 6: return ws_launch_one<0, false, false, 6>(p, g, stream);
      case 8: return ws_launch_one<0, false, false, 8>(p, g, stream);
      case 9: return ws_launch_one<0, false, false, 9>(p, g, stream);
      case 18: return ws_launch_one<0, false, false, 18>(p, g, stream);
      case 19: return ws_launch_one<0, false, false, 19>(p, g, stream);
      case 7: return ws_launch_one<0, false, false, 7>(p, g, stream);
      case 23: return ws_launch_one<0, false, false, 23>(p, g, stream);
      case 128: return ws_launch_one<0, false, false, 128>(p, g, stream);
      case 129: return ws_launch_one<0, false, false, 129>(p, g, stream);
      case 256: return ws_launch_one<0, false, false, 256>(p, g, stream);
      case 512: return ws_launch_one<0, false, false, 512>(p, g, stream);
      case 384: return ws_launch_one<0, false, false, 384>(p, g, stream);
      case 640: return ws_launch_one<0, false, false, 640>(p, g, stream);
      case 130: return ws_launch_one<0, false, false, 130>(p, g, stream);
      case 131: return ws_launch_one<0, false, false, 131>(p, g, stream);
      case 32: return ws_launch_one<0, false, false, 32>(p, g, stream);
      case 64: return ws_launch_one<0, false, false, 64>(p, g, stream);
      case 33: return ws_launch_one<0, false, false, 33>(p, g, stream);
      case 65: return ws_launch_one<0, false, false, 65>(p, g, stream);
#define WS_OPT_CASE(o) case 200 + o: return ws_launch_one<0, false, false, 0, 3, o>(p, g, stream); \
                       case 300 + o: return ws_launch_one<0, false, false, 3, 3, o>(p, g, stream);
      WS_OPT_CASE(0) WS_OPT_CASE(1) WS_OPT_CASE(8) WS_OPT_CASE(9)
#undef WS_OPT_CASE
      default: break;
    }
#endif
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
