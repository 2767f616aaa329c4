// Short-sequence multi-head attention of the pose lifter (timm Attention == reference PoseEstimation.py:78-104 via
// vision_transformer.Attention; the in-tree copy of the same arithmetic is CoevoDecoder.py:118-131) on the f16 MATRIX pipe, for
// the split-f16 product mode: q, k, v arrive as the fp32 rows the qkv product wrote, are split into (hi, lo) f16 planes on the way
// to the matrix pipe, and the result leaves pre-split - [16 f16 hi | 16 f16 lo*2^11] per 16 channels, in the bytes of the fp32
// values - as the A operand of `proj`.
//
// Why a second attention kernel.  seq_attention_pair_kernel (lifter.hip) does the 2 x N x N x HD multiply-adds of a head on the
// vector pipe, one query pair per lane: at C = 512 it spends 4,352 v_fma per lane and sequence with 72 of 128 lanes active, holds
// the whole q / k / v of a sequence in registers on its way to LDS (256 VGPRs, 6-20 spilled), and its phases (fetch, spread,
// pass 1, softmax, pass 2, store) run one after the other in each of the 4 workgroups a CU holds: 175-186 / 150-170 us per launch
// at B = 256 against 91-113 us of HBM time for the 571-713 MB it moves (6.3 TB/s achievable).  Here the arithmetic is 24 matrix
// instructions per head, K and V go global -> LDS without passing through registers (LDS-DMA), and the bytes in flight do not
// depend on how many registers a wave can spare.
//
// Arithmetic.  With x = xh + xl*2^-11 (xh = rne16(x), xl = rne16((x - xh) * 2^11)):
//     q.k = sum qh*kh + 2^-11 * sum (qh*kl + ql*kh)          (dropped: ql*kl, 2^-22 relative)
// the two sums in separate fp32 accumulators of v_mfma_f32_32x32x16_f16.  Computed TRANSPOSED, S^T = K Q^T (rows = keys, columns =
// queries): a lane then holds 16 of the 32 keys of ONE query, the softmax over keys is in-lane plus one exchange with lane ^ 32,
// and P^T in the accumulators already IS the B operand (k = keys, n = queries) of out^T = V^T P^T - in the register order
// key(s, hb, e) = 16 s + 4 hb + (e & 3) + 8 (e >> 2), which the A operand (V^T, gathered from the row-major V with 16-bit LDS
// reads) simply follows.  P is split like every other operand (p in [0, 1]), the softmax is the exact two-pass form on the hardware
// 2^x with log2-scaled scores, normalisation by 1 / sum once at the end.  Keys >= N (a sequence is 16 frames or 17 / 19 joints;
// the matrix tile is 32 x 32) read a clamped row and get a score of -inf.
//
// Work split.  A unit = one sequence x one 1 KB column chunk (4 heads at HD = 64, all 8 at HD = 32).  Four waves, each owning the
// heads in its 256 B of the chunk:
//   * K and V rows of the unit: 2 N DMA instructions of 64 lanes x 16 B into a ring of two slots, rows at a stride of 1040 B (65
//     sixteen-byte slots: the 16 rows a ds_read_b128 service group touches fall on 16 different ones).  Each wave converts ITS
//     strip of every row in place, fp32 -> [16 hi | 16 lo*2^11] per 16 channels (a quad of lanes reads the 64 bytes of a group in
//     one instruction and writes them back afterwards; all reads of the strip are issued before the first value is used);
//   * Q is the B operand - lane (query, k-half) needs 8 consecutive channels of its own row - so it goes global -> registers
//     directly (32 B per lane and k-step, every fetched line fully used by the wave) and is split in registers; the loads of the
//     NEXT unit are issued before this unit's arithmetic;
//   * the result is staged through the wave's own strip of the (dead) K rows and leaves as full 256-byte row segments.
// One barrier per unit: wait for the unit's DMAs (counted s_waitcnt: vmcnt counts a wave's loads, DMAs and stores in issue order),
// barrier, issue the next unit's Q loads and DMAs into the other slot, compute.  Persistent workgroups, TWO per CU (71-79 KB of
// LDS each): with one wave per SIMD nothing hides the dependent chain scores -> softmax -> P V -> split -> store of a unit;
// with two, one workgroup's chain runs under the other's.  (Measured on the way: q, k, v pre-split by the qkv product's epilogue
// and all three through LDS, one workgroup per CU with a ring of three: 117 / 110 us per launch at C = 512, but +37 us on each
// qkv product, which runs at the chip's power limit; the same kernel splitting all three strips in LDS: 149 / 131 us.)
#include "gemm_split_common.hpp"

namespace {

// lo plane of x given hi = rne16(x): rne16((x - hi) * 2^11), written as ONE fused multiply-add on the f16 itself (v_fma_mix_f32: no separate
// f16 -> f32 conversion, no separate subtraction).  x - hi is exact in fp32 (hi is x rounded to 11 bits), so is every product by 2^11, so
// the fused form has the same bits as the three-instruction form - the kernel is bound by vector issue (980 vector instructions per head and
// unit against 24 matrix instructions, 66 % of its time by the PMC counts), and the splits are a third of them.
__device__ __forceinline__ _Float16 lo_plane(float x, _Float16 hi) { return (_Float16)fmaf((float)hi, -2048.0f, x * 2048.0f); }
__device__ __forceinline__ void split8_fused(const f32x4& x0, const f32x4& x1, f16x8& hi, f16x8& lo) {
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) hi[e] = (_Float16)v[e];
#pragma unroll
  for (int e = 0; e < 8; ++e) lo[e] = lo_plane(v[e], hi[e]);
}

template <int HD, int N>
struct AttnCfg {
  static constexpr int C = 8 * HD;
  static constexpr int UPS = C * 4 / 1024;         // units (1 KB column chunks) per sequence
  static constexpr int HPW = 1024 / (HD * 4) / 4;  // heads per wave
  static constexpr int RS = 1040;                  // LDS row stride in bytes
  static constexpr int UNIT = 2 * N * RS;          // one ring slot: K rows, V rows
  static constexpr int NS = 2;
  static constexpr int DPW = (2 * N + 3) / 4;      // DMA instructions per wave and unit (the last ones may repeat a row)
  static constexpr int ST = (N + 3) / 4;           // result store instructions per wave and unit
  static constexpr int LDS_BYTES = NS * UNIT;
  static constexpr int KSTEPS = HD / 16, DBLK = HD / 32, PSTEPS = N > 16 ? 2 : 1;
  static constexpr int QL = 2 * KSTEPS * HPW;      // 16-byte Q loads per lane and unit
};

template <int HD, int N>
__global__ __launch_bounds__(256, 2) void seq_attention_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int nseq,
                                                                   int seq_div, long long seq_lo, long long seq_hi,
                                                                   long long tok_stride, unsigned* oflow) {
  using Cfg = AttnCfg<HD, N>;
  constexpr int C = Cfg::C, UPS = Cfg::UPS, HPW = Cfg::HPW, RS = Cfg::RS, UNIT = Cfg::UNIT, DPW = Cfg::DPW, ST = Cfg::ST, QL = Cfg::QL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hb = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  const int total = nseq * UPS;
  const int nmine = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const unsigned tok_bytes = (unsigned)tok_stride * (unsigned)(3 * C * 4);  // (a sequence spans < 4 GiB: checked by the launcher)
  const int rowc = min(l31, N - 1);  // this lane's key row (A operand of K Q^T) and query row (B operand), clamped into the sequence

  auto unit_coords = [&](int k, long long& tokbase, int& g) {
    const int u = (int)blockIdx.x + k * (int)gridDim.x;
    const int seq = u / UPS;
    g = u - seq * UPS;
    tokbase = (long long)(seq % seq_div) * seq_lo + (long long)(seq / seq_div) * seq_hi;
  };
  // K and V rows of unit k -> ring slot (arrays 1 and 2 of the row: byte offsets C*4 and 2*C*4)
  auto issue = [&](int k, int slot) {
    long long tb;
    int g;
    unit_coords(k, tb, g);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qkv) + tb * (3 * C), 0, 0xffffffff, 0x00020000);
    const unsigned slot_base = lds0 + (unsigned)slot * UNIT;
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
      const int i = min(wave + 4 * j, 2 * N - 1);  // (a repeated instruction rewrites the same row with the same bytes)
      const int a = i / N, r = i - a * N;
      sdma16(rsrc, (unsigned)lane * 16u, (int)((unsigned)r * tok_bytes + (unsigned)((a + 1) * C * 4 + g * 1024)), slot_base + (unsigned)(i * RS));
    }
  };
  // this lane's part of Q of unit k: row rowc, per head and k-step the 8 channels 16 ks + 8 hb + [0, 8)
  auto load_q = [&](int k, f32x4 (&q)[QL]) {
    long long tb;
    int g;
    unit_coords(k, tb, g);
    const float* row = qkv + (tb + (long long)rowc * tok_stride) * (3 * C) + g * 256 + wave * 64 + hb * 8;
#pragma unroll
    for (int i = 0; i < QL / 2; ++i) {  // i = head-in-wave * KSTEPS + ks: consecutive 16-channel groups of the wave's 64 floats
      q[2 * i] = *reinterpret_cast<const f32x4*>(row + i * 16);
      q[2 * i + 1] = *reinterpret_cast<const f32x4*>(row + i * 16 + 4);
    }
  };

  // hd^-0.5 * log2(e): scores in log2 units, softmax on the hardware 2^x
  constexpr float scale = (HD == 32 ? 0.17677669529663688110f : 0.125f) * 1.44269504088896340736f;
  constexpr float two_m11 = 0.00048828125f;
  bool bad = false;
  f32x4 qraw[QL];
  if (nmine > 0) {
    load_q(0, qraw);
    issue(0, 0);
  }
  int slot = 0;
  for (int it = 0; it < nmine; ++it) {
    // the DMAs of unit `it` were followed by the stores of unit it - 1 only (the Q loads of `it` went out before them)
    if (it > 0) wait_vm<ST>();
    else wait_vm<0>();
    __syncthreads();  // every wave's rows of unit `it` are in LDS; every wave is done with the other slot
    f16x8 qh[QL / 2], ql[QL / 2];
#pragma unroll
    for (int i = 0; i < QL / 2; ++i) split8_fused(qraw[2 * i], qraw[2 * i + 1], qh[i], ql[i]);
    if (it + 1 < nmine) {
      load_q(it + 1, qraw);
      issue(it + 1, slot ^ 1);
    }
    unsigned char* const U = smem + slot * UNIT;
    {
      // this wave's 256-byte strip of every K and V row, fp32 -> planes in place; every read is issued before the first value is
      // used; the rows past 2 N - 1 of the last instruction are clamped (their lanes write the same bytes a second time)
      constexpr int NI = (2 * N + 3) / 4;
      unsigned char* grp[NI];
      f32x4 x[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        grp[i] = U + min(4 * i + (lane >> 4), 2 * N - 1) * RS + wave * 256 + ((lane & 15) >> 2) * 64;
        x[i] = *reinterpret_cast<const f32x4*>(grp[i] + (lane & 3) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        f16x4 xh, xl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[e] = (_Float16)x[i][e];
          xl[e] = lo_plane(x[i][e], xh[e]);
        }
        // 16 consecutive lanes (a ds_write_b64 service group) hold one row's four groups: groups 0, 1 write hi first and groups 2, 3
        // lo first, so that the 16 eight-byte stores of an instruction fall on 16 different bank pairs (hi and lo areas of groups g
        // and g + 2 are 128 B apart - the same banks for a store)
        const bool lo_first = (lane & 8) != 0;
        *reinterpret_cast<f16x4*>(grp[i] + (lo_first ? 32 : 0) + (lane & 3) * 8) = lo_first ? xl : xh;
        *reinterpret_cast<f16x4*>(grp[i] + (lo_first ? 0 : 32) + (lane & 3) * 8) = lo_first ? xh : xl;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
      const int hoff = (wave * HPW + h) * (HD * 4);  // this head's bytes inside a row of the unit
      // Every LDS read of the head - the K fragments and the gathered V^T fragments - is issued here, ahead of the dependent
      // chain (scores -> softmax -> P V).
      const unsigned char* pk = U + rowc * RS + hoff + hb * 16;
      f16x8 kh[Cfg::KSTEPS], kl[Cfg::KSTEPS];
#pragma unroll
      for (int ks = 0; ks < Cfg::KSTEPS; ++ks) {
        kh[ks] = *reinterpret_cast<const f16x8*>(pk + ks * 64);
        kl[ks] = *reinterpret_cast<const f16x8*>(pk + ks * 64 + 32);
      }
      f16x8 vh[Cfg::DBLK][Cfg::PSTEPS], vl[Cfg::DBLK][Cfg::PSTEPS];
#pragma unroll
      for (int blk = 0; blk < Cfg::DBLK; ++blk) {
        // f16 index of channel 32 blk + l31 inside the row's planes: group (2 blk + (l31 >> 4)) of 32 halves, hi at (l31 & 15), lo 16 further
        const _Float16* pv = reinterpret_cast<const _Float16*>(U + N * RS + hoff) + (2 * blk + (l31 >> 4)) * 32 + (l31 & 15);
#pragma unroll
        for (int s = 0; s < Cfg::PSTEPS; ++s) {
          // keys 16.. exist only up to N - 1 (and only in the hb = 0 half): the other slots carry p = 0 and may hold any finite v
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int key = 16 * s + (e & 3) + 8 * (e >> 2);  // + 4 hb
            if (s == 0 || key < N) {
              const _Float16* pr = pv + (size_t)min(key + 4 * hb, N - 1) * (RS / 2);
              vh[blk][s][e] = pr[0];
              vl[blk][s][e] = pr[16];
            } else {
              vh[blk][s][e] = vh[blk][s][0];
              vl[blk][s][e] = vl[blk][s][0];
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- S^T = K Q^T ----
      f32x16 s_main, s_cross;
#pragma unroll
      for (int r = 0; r < 16; ++r) s_main[r] = s_cross[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < Cfg::KSTEPS; ++ks) {
        s_main = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[ks], qh[h * Cfg::KSTEPS + ks], s_main, 0, 0, 0);
        s_cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[ks], ql[h * Cfg::KSTEPS + ks], s_cross, 0, 0, 0);
        s_cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[ks], qh[h * Cfg::KSTEPS + ks], s_cross, 0, 0, 0);
      }
      // ---- softmax over the keys of this lane's query: register r is key 4 hb + (r & 3) + 8 (r >> 2) ----
      float p[16];
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 4 * hb + (r & 3) + 8 * (r >> 2);
        const float sv = fmaf(s_cross[r], two_m11, s_main[r]) * scale;
        p[r] = (r < 8 || N > 16) ? (key < N ? sv : -INFINITY) : -INFINITY;  // (N <= 16: registers 8..15 are keys >= 16)
        mx = fmaxf(mx, p[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(p[r] - mx);
        sum += p[r];
      }
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
      // ---- out^T = V^T P^T, one 32-channel block at a time; P split ONCE for the head (it was split again for every block) ----
      f16x8 ph[Cfg::PSTEPS], pl[Cfg::PSTEPS];
#pragma unroll
      for (int s = 0; s < Cfg::PSTEPS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pv32 = pinned(p[8 * s + e]);
          ph[s][e] = (_Float16)pv32;
          pl[s][e] = lo_plane(pv32, ph[s][e]);
        }
#pragma unroll
      for (int blk = 0; blk < Cfg::DBLK; ++blk) {
        f32x16 o_main, o_cross;
#pragma unroll
        for (int r = 0; r < 16; ++r) o_main[r] = o_cross[r] = 0.f;
#pragma unroll
        for (int s = 0; s < Cfg::PSTEPS; ++s) {
          o_main = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[blk][s], ph[s], o_main, 0, 0, 0);
          o_cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[blk][s], pl[s], o_cross, 0, 0, 0);
          o_cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[blk][s], ph[s], o_cross, 0, 0, 0);
        }
        // register r is channel 32 blk + 4 hb + (r & 3) + 8 (r >> 2) of query l31: four consecutive channels per r >> 2, written
        // pre-split into this head's bytes of the (dead) K row of that query
        if (l31 < N) {
          unsigned char* po = U + l31 * RS + hoff;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            f16x4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = pinned(fmaf(o_cross[4 * rq + e], two_m11, o_main[4 * rq + e]) * inv);
              bad = bad || nonfinite(v);
              oh[e] = (_Float16)v;
              ol[e] = lo_plane(v, oh[e]);
            }
            const int g16 = 2 * blk + (rq >> 1), idx = 8 * (rq & 1) + 4 * hb;
            *reinterpret_cast<f16x4*>(po + g16 * 64 + idx * 2) = oh;
            *reinterpret_cast<f16x4*>(po + g16 * 64 + 32 + idx * 2) = ol;
          }
        }
      }
    }
    // ---- this wave's 256 B of every result row: LDS (written by other lanes of the wave) -> global, full segments ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      long long tb;
      int g;
      unit_coords(it, tb, g);
#pragma unroll
      for (int i = 0; i < ST; ++i) {
        const int row = 4 * i + (lane >> 4), piece = lane & 15;
        if (row < N) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(U + row * RS + wave * 256 + piece * 16);
          *reinterpret_cast<f32x4*>(out + (tb + row * tok_stride) * C + g * 256 + wave * 64 + piece * 4) = t;
        }
      }
    }
    slot ^= 1;
  }
  report_nonfinite(oflow, bad);
}

template <int HD, int N>
int launch(const float* qkv, float* out, int nseq, int seq_div, long long seq_lo, long long seq_hi, long long tok_stride, hipStream_t stream) {
  using Cfg = AttnCfg<HD, N>;
  static std::atomic<unsigned long long> done{0};
  PMCE_TRY(pmce_opt_in_lds(reinterpret_cast<const void*>(&seq_attention_mfma_kernel<HD, N>), Cfg::LDS_BYTES, done, "seq_attention_split_f16"));
  const int units = nseq * Cfg::UPS;
  const int grid = units < 512 ? units : 512;
  hipLaunchKernelGGL((seq_attention_mfma_kernel<HD, N>), dim3(grid), dim3(256), Cfg::LDS_BYTES, stream, qkv, out, nseq, seq_div, seq_lo, seq_hi,
                     tok_stride, pmce_overflow_sink());
  return pmce_check_launch("seq_attention_split_f16");
}

}  // namespace

extern "C" int pmce_seq_attention_split_supported(int N, int C) { return (C == 256 || C == 512) && (N == 16 || N == 17 || N == 19) ? 1 : 0; }

extern "C" int pmce_seq_attention_split_f16(const float* qkv, float* out, int nseq, int N, int C, int seq_div, long long seq_lo,
                                            long long seq_hi, long long tok_stride, hipStream_t stream) {
  PMCE_REQUIRE(qkv && out && nseq > 0, "seq_attention_split_f16: bad arguments");
  if (seq_div <= 0) seq_div = 0x7fffffff;  // (as pmce_seq_attention_f32: sequence s starts at token s * seq_lo)
  PMCE_REQUIRE(pmce_seq_attention_split_supported(N, C), "seq_attention_split_f16: C must be 256 or 512 and N 16, 17 or 19 (got N=%d C=%d)", N, C);
  PMCE_REQUIRE(tok_stride > 0 && (long long)N * tok_stride * 3 * C * 4 < (1ll << 32), "seq_attention_split_f16: a sequence spans 4 GiB or more");
#define PMCE_ATTN_CASE(HD_, N_) \
  if (C == 8 * HD_ && N == N_) return launch<HD_, N_>(qkv, out, nseq, seq_div, seq_lo, seq_hi, tok_stride, stream)
  PMCE_ATTN_CASE(64, 16);
  PMCE_ATTN_CASE(64, 17);
  PMCE_ATTN_CASE(64, 19);
  PMCE_ATTN_CASE(32, 16);
  PMCE_ATTN_CASE(32, 17);
  PMCE_ATTN_CASE(32, 19);
#undef PMCE_ATTN_CASE
  return PMCE_OK;
}
