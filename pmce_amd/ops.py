"""Per-operator Python wrappers over the C ABI (include/pmce_hip.h).  Used by the parity tests and by callers that
want a single stage; the whole-path modules in pmce_amd/models call pmce_forward instead.  All tensors are
contiguous fp32 CUDA tensors; weights are passed the way the reference's state_dict stores them."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .packing import N_ADA

P = _lib.ptr


def _st():
    return _lib.current_stream()


def _c(t):
    return t.to(torch.float32).contiguous()


def gemm_nt(A, W, bias=None, residual=None, act=0, out=None):
    """out = act(A @ W^T + bias) + residual  (nn.Linear)."""
    lib = _lib.load()
    A, W = _c(A), _c(W)
    M, K = A.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    _lib.check(lib.pmce_gemm_nt_f32(P(A), P(W), P(bias), P(residual), P(out), M, N, K, K, K, N, act, 0, 0, 0, 0, 0, 0, 1,
                                    0, 0, 0, 0, _st()), "gemm_nt")
    return out


def pack_split_f16(W):
    """W[N,K] fp32 -> (Wp[N,K] storage of the f16 hi/lo planes, wscale[N] = 2^-s per output row) for :func:`gemm_nt_split`."""
    lib = _lib.load()
    W = _c(W)
    N, K = W.shape
    Wp = torch.empty(N, K, device=W.device, dtype=torch.float32)
    wscale = torch.empty(N, device=W.device, dtype=torch.float32)
    _lib.check(lib.pmce_gemm_pack_split_f16(P(W), N, K, K, P(Wp), P(wscale), _st()), "gemm_pack_split_f16")
    return Wp, wscale


def pack_split_f16_blk(W):
    """The same planes in the blocked layout [ceil(N/64)][K/16][64][16 hi | 16 lo] (what the model packs); -> (Wblk, wscale, N)."""
    lib = _lib.load()
    W = _c(W)
    N, K = W.shape
    Np = (N + 63) // 64 * 64
    Wp = torch.zeros(Np, K, device=W.device, dtype=torch.float32)
    ws = torch.empty(N, device=W.device, dtype=torch.float32)
    _lib.check(lib.pmce_gemm_pack_split_f16_blk(P(W), N, K, K, P(Wp), P(ws), _st()), "gemm_pack_split_f16_blk")
    return Wp, ws, N


def gemm_nt_split_blk(A, Wblk, wscale, N, bias=None, residual=None, act=0, a_packed=False, c_packed=False, rscale=None, rowmap=None, out=None):
    """Every form of the three-product GEMM on a blocked weight (pmce_gemm_nt_split_f16_blk)."""
    lib = _lib.load()
    M, K = A.shape
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    cd, lo, hi = rowmap if rowmap else (0, 0, 0)
    _lib.check(lib.pmce_gemm_nt_split_f16_blk(P(A), P(rscale), P(Wblk), P(wscale), P(bias), P(residual), P(out), M, N, K, K, N, act,
                                              1 if a_packed else 0, 1 if c_packed else 0, cd, lo, hi, _st()), "gemm_nt_split_blk")
    return out


def gemm_nt_split_ln(Ap, Wp, wscale, bias, residual, ln1=None, ln2=None, blocked=True, want_out1=True):
    """pmce_gemm_nt_split_f16_ln: x = Ap W^T + bias + residual (N = 256); y1 = LN(x; ln1) or x; returns (y1 fp32 or None,
    LN(y1; ln2) pre-split or None).  ln1 / ln2 = (weight, bias, eps)."""
    lib = _lib.load()
    M, K = Ap.shape
    out1 = torch.empty(M, 256, device=Ap.device) if want_out1 else None
    out2 = torch.empty(M, 256, device=Ap.device) if ln2 is not None else None
    w1, b1, e1 = (_c(ln1[0]), _c(ln1[1]), float(ln1[2])) if ln1 is not None else (None, None, 0.0)
    w2, b2, e2 = (_c(ln2[0]), _c(ln2[1]), float(ln2[2])) if ln2 is not None else (None, None, 0.0)
    _lib.check(lib.pmce_gemm_nt_split_f16_ln(P(Ap), P(Wp), 1 if blocked else 0, P(wscale), P(bias), P(residual), M, K, P(w1), P(b1), e1, P(out1),
                                             P(w2), P(b2), e2, P(out2), _st()), "gemm_nt_split_ln")
    return out1, out2


def split_rows_f16(A):
    """A[M,K] fp32 -> the packed (hi | lo*2^11) f16 planes, as an [M,K] float32-typed buffer."""
    lib = _lib.load()
    A = _c(A)
    M, K = A.shape
    Ap = torch.empty(M, K, device=A.device, dtype=torch.float32)
    _lib.check(lib.pmce_split_rows_f16(P(A), M, K, K, P(Ap), _st()), "split_rows_f16")
    return Ap


def gemm_nt_split(A, Wp, wscale, bias=None, residual=None, act=0, out=None, a_packed=False, c_packed=False):
    """gemm_nt on the f16 matrix pipe (three-product split, fp32 accumulate): fp32 in, fp32 out, fp32 accuracy."""
    lib = _lib.load()
    A = _c(A)
    M, K = A.shape
    N = Wp.shape[0]
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    _lib.check(lib.pmce_gemm_nt_split_f16_ex(P(A), P(Wp), P(wscale), P(bias), P(residual), P(out), M, N, K, K, N, act,
                                             1 if a_packed else 0, 1 if c_packed else 0, _st()), "gemm_nt_split")
    return out


def split_rows_scaled_f16(A):
    """fp32 rows of any finite magnitude -> (planes of A[m] * 2^-e(m) in the packed-A layout, rscale[m] = 2^e(m))."""
    lib = _lib.load()
    A = _c(A)
    M, K = A.shape
    Ap = torch.empty(M, K, device=A.device, dtype=torch.float32)
    rs = torch.empty(M, device=A.device, dtype=torch.float32)
    _lib.check(lib.pmce_split_rows_scaled_f16(P(A), M, K, K, P(Ap), P(rs), _st()), "split_rows_scaled_f16")
    return Ap, rs


def gemm_nt_split_rs(Ap, rscale, Wp, wscale, bias=None, out=None, rowmap=None):
    """The three-product f16 GEMM on a row-scaled packed A (raw inputs: img_feat).  rowmap = (c_div, c_lo, c_hi) maps output rows.
    K >= 128 (include/pmce_hip.h at pmce_gemm_nt_split_f16_rs)."""
    lib = _lib.load()
    M, K = Ap.shape
    N = Wp.shape[0]
    if out is None:
        out = torch.empty(M, N, device=Ap.device, dtype=torch.float32)
    cd, lo, hi = rowmap if rowmap else (0, 0, 0)
    _lib.check(lib.pmce_gemm_nt_split_f16_rs(P(Ap), P(rscale), P(Wp), P(wscale), P(bias), P(out), M, N, K, N, cd, lo, hi, _st()),
               "gemm_nt_split_rs")
    return out


def gru_step_split(gi, whh, bhh, h_prev, blocked=True):
    """One GRU time step of ONE direction in the three-product f16 form (nn.GRU gate order r, z, n; CoevoDecoder.py:216-221): gi [B, 3H] =
    W_ih x + b_ih, whh [3H, H], bhh [3H], h_prev [B, H] or None (h = 0) -> h' [B, H].  B <= 64 runs the small-batch kernel, larger batches
    gru_step_v2 - same numbers, bit for bit."""
    lib = _lib.load()
    gi, whh, bhh = _c(gi), _c(whh), _c(bhh)
    B, H = gi.shape[0], whh.shape[1]
    if blocked:     # the layout the model packs W_hh in
        Wp, wscale, _ = pack_split_f16_blk(whh)
    else:
        Wp, wscale = pack_split_f16(whh)
    hp = None if h_prev is None else _c(h_prev)
    out = torch.empty(B, H, device=gi.device, dtype=torch.float32)
    _lib.check((lib.pmce_gru_step_split_blk_f32 if blocked else lib.pmce_gru_step_split_f32)(P(gi), None, P(Wp), None, P(wscale), P(bhh), None, P(hp), None, P(out), None, 3 * H, H, B, H, 1,
                                           _st()), "gru_step_split")
    return out


def ln_chain(x, w1=None, b1=None, eps1=1e-6, add=None, add_div=1, add_mod=1, want_out1=True, w2=None, b2=None, eps2=1e-6,
             out2_split=False):
    lib = _lib.load()
    x = _c(x)
    rows, Cc = x.shape
    out1 = torch.empty_like(x) if want_out1 else None
    out2 = torch.empty_like(x) if w2 is not None else None
    _lib.check(lib.pmce_ln_chain_ex_f32(P(x), rows, Cc, P(w1), P(b1), eps1, P(add), add_div, add_mod, P(out1), P(w2), P(b2), eps2,
                                        P(out2), 1 if out2_split else 0, _st()), "ln_chain")
    return out1, out2


def embed_tokens(pose2d, E, Wje, bje, spos):
    """Token embedding (PoseEstimation.py:78-81): pose2d [BT, J, 2], E = imgfeat_embed(img_feat) [BT, C] -> tokens [BT * J, C]."""
    lib = _lib.load()
    BT, J, _ = pose2d.shape
    Cc = E.shape[1]
    x = torch.empty(BT * J, Cc, device=E.device, dtype=torch.float32)
    _lib.check(lib.pmce_embed_tokens_f32(P(_c(pose2d)), P(_c(E)), P(_c(Wje)), P(_c(bje)), P(_c(spos)), P(x), BT * J, J, Cc, _st()), "embed_tokens")
    return x


def embed_ln(pose2d, E, Wje, bje, spos, w2, b2, eps2=1e-6, xn_split=False):
    """embed_tokens + LayerNorm(w2, b2) of every row in one launch: (tokens, their LayerNorm - pre-split planes when xn_split)."""
    lib = _lib.load()
    BT, J, _ = pose2d.shape
    Cc = E.shape[1]
    x = torch.empty(BT * J, Cc, device=E.device, dtype=torch.float32)
    xn = torch.empty_like(x)
    _lib.check(lib.pmce_embed_ln_f32(P(_c(pose2d)), P(_c(E)), P(_c(Wje)), P(_c(bje)), P(_c(spos)), P(x), BT * J, J, Cc, P(_c(w2)), P(_c(b2)), eps2,
                                     P(xn), 1 if xn_split else 0, _st()), "embed_ln")
    return x, xn


def lifter_head(x, lnw, lnb, Wr, br, wf, bf, B, T, J, pre=None):
    """Regression head + frame fusion (PoseEstimation.py:62-66,109-113): x [B*T*J, C] -> pose3d [B, J, 3].  pre = (weight, bias, eps): the rows
    pass through that LayerNorm first (the last block's norm_t folded into the head)."""
    lib = _lib.load()
    Cc = x.shape[1]
    out = torch.empty(B, J, 3, device=x.device, dtype=torch.float32)
    pw, pb, pe = (_c(pre[0]), _c(pre[1]), float(pre[2])) if pre is not None else (None, None, 0.0)
    _lib.check(lib.pmce_lifter_head_ex_f32(P(_c(x)), P(pw), P(pb), pe, P(_c(lnw)), P(_c(lnb)), P(_c(Wr)), P(_c(br)), P(_c(wf)), P(_c(bf)), P(out),
                                           B, T, J, Cc, _st()), "lifter_head")
    return out


def seq_attention(qkv, nseq, N, Cc, seq_div, seq_lo, seq_hi, tok_stride, out_split=False):
    lib = _lib.load()
    qkv = _c(qkv)
    out = torch.empty(qkv.shape[0], Cc, device=qkv.device, dtype=torch.float32)
    _lib.check(lib.pmce_seq_attention_ex_f32(P(qkv), P(out), nseq, N, Cc, seq_div, seq_lo, seq_hi, tok_stride,
                                             1 if out_split else 0, _st()), "seq_attention")
    return out


def seq_attention_split(qkv, nseq, N, Cc, seq_div, seq_lo, seq_hi, tok_stride):
    """The matrix-pipe attention of the split-f16 mode: fp32 q, k, v in, pre-split result out (see :func:`unsplit_rows_f16`)."""
    lib = _lib.load()
    qkv = _c(qkv)
    out = torch.empty(qkv.shape[0], Cc, device=qkv.device, dtype=torch.float32)
    _lib.check(lib.pmce_seq_attention_split_f16(P(qkv), P(out), nseq, N, Cc, seq_div, seq_lo, seq_hi, tok_stride, _st()),
               "seq_attention_split_f16")
    return out


def unsplit_rows_f16(Ap):
    """The fp32 values a packed (hi | lo*2^11) f16 buffer stands for (float64, exact): inverse of :func:`split_rows_f16`."""
    M, K = Ap.shape
    h = Ap.contiguous().view(torch.float16).view(M, K // 16, 2, 16).double()
    return (h[:, :, 0] + h[:, :, 1] / 2048.0).reshape(M, K)


def vertex_init_gather(joints, vj_relation):
    lib = _lib.load()
    joints = _c(joints)
    B, J, _ = joints.shape
    vj = torch.as_tensor(np.asarray(vj_relation).astype(np.int32), device=joints.device)
    out = torch.empty(B, 431, 3, device=joints.device, dtype=torch.float32)
    _lib.check(lib.pmce_vertex_init_gather_f32(P(joints), P(vj), P(out), B, J, _st()), "vertex_init_gather")
    return out


def adaln_params(g, sd, names):
    """GB[B, len(names)*128]: [gamma(64)|beta(64)] per AdaLN instance, via the packed GEMM (CoevoDecoder.py:19-20,27-28)."""
    Wt = torch.cat([torch.cat([sd[n + ".mlp_gamma.weight"], sd[n + ".mlp_beta.weight"]], 0) for n in names], 0)
    bt = torch.cat([torch.cat([sd[n + ".mlp_gamma.bias"], sd[n + ".mlp_beta.bias"]], 0) for n in names], 0)
    return gemm_nt(g, _c(Wt), _c(bt))


def cross_attn_vertex(xq, xk, xv, g, sd, p):
    """Fused AdaLN + vertex<-joint cross-attention + residual: xq + CA(AdaLN_q(xq), AdaLN_k(xk), AdaLN_v(xv))
    (first line of CrossAttentionBlock.forward, CoevoDecoder.py:83) with p = '...vertx_CA_FFN'."""
    lib = _lib.load()
    xq, xk, xv = _c(xq), _c(xk), _c(xv)
    B, Nq, _ = xq.shape
    J = xk.shape[1]
    assert Nq == 431
    GB = adaln_params(g, sd, [p + ".normq", p + ".normk", p + ".normv"])
    dev = xq.device
    Kf = torch.empty(B, 64, 64, device=dev)
    s0 = torch.empty(B, 64, device=dev)
    Vf = torch.empty(B, 64, 64, device=dev)
    w = {k: _c(sd[f"{p}.attn.{k}"]) for k in ("wq.weight", "wq.bias", "wk.weight", "wk.bias", "wv.weight", "wv.bias",
                                              "proj.weight", "proj.bias")}
    _lib.check(lib.pmce_ca_fold_f32(P(xk), P(xv), P(GB), GB.shape[1], 0, 1, 2, P(w["wq.weight"]), P(w["wq.bias"]),
                                    P(w["wk.weight"]), P(w["wk.bias"]), P(w["wv.weight"]), P(w["wv.bias"]),
                                    P(w["proj.weight"]), P(Kf), P(s0), P(Vf), B, J, _st()), "ca_fold")
    out = torch.empty_like(xq)
    _lib.check(lib.pmce_vertex_ca_f32(P(xq), None, None, None, P(Kf), P(s0), P(Vf), P(w["proj.bias"]), P(out), B, J, _st()),
               "vertex_ca")
    return out


def ffn_image(fc1_w, fc2_w):
    """The LDS image of a 64->256->64 FFN's f16 form (pmce_ffn_pack_f16): what pmce_model_finalize makes per decoder FFN."""
    lib = _lib.load()
    fc1_w, fc2_w = _c(fc1_w), _c(fc2_w)
    assert tuple(fc1_w.shape) == (256, 64) and tuple(fc2_w.shape) == (64, 256)
    img = torch.empty(lib.pmce_ffn_image_floats(), device=fc1_w.device)
    _lib.check(lib.pmce_ffn_pack_f16(P(fc1_w), P(fc2_w), P(img), _st()), "ffn_pack_f16")
    return img


def cross_attn_block_vertex(xq, xk, xv, g, sd, p, split_f16=False, packed=False):
    """The whole vertex-stream CrossAttentionBlock in one launch (CoevoDecoder.py:82-87), p = '...vertx_CA_FFN':
    xq + CA(...) then + Mlp(AdaLN_2(.)).  fp32 form: bit-identical to cross_attn_vertex followed by adaln_mlp.  split_f16: the FFN and
    (round 5, J <= 23) the attention's two contractions in the three-product f16 form - the folded key / value operands reach the kernel as
    the f16 image ca_fold writes.  packed (with split_f16): the FFN's planes from a pre-made image (ffn_image) instead of converted per
    workgroup."""
    lib = _lib.load()
    xq, xk, xv = _c(xq), _c(xk), _c(xv)
    B, Nq, _ = xq.shape
    J = xk.shape[1]
    assert Nq == 431
    GB = adaln_params(g, sd, [p + ".normq", p + ".normk", p + ".normv", p + ".norm2"])
    dev = xq.device
    Kf = torch.empty(B, 64, 64, device=dev)
    s0 = torch.empty(B, 64, device=dev)
    Vf = torch.empty(B, 64, 64, device=dev)
    img = torch.empty(B, lib.pmce_ca_image_floats(), device=dev) if (split_f16 and J <= 23) else None
    w = {k: _c(sd[f"{p}.attn.{k}"]) for k in ("wq.weight", "wq.bias", "wk.weight", "wk.bias", "wv.weight", "wv.bias",
                                              "proj.weight", "proj.bias")}
    _lib.check(lib.pmce_ca_fold_img_f32(P(xk), P(xv), P(GB), GB.shape[1], 0, 1, 2, P(w["wq.weight"]), P(w["wq.bias"]),
                                        P(w["wk.weight"]), P(w["wk.bias"]), P(w["wv.weight"]), P(w["wv.bias"]),
                                        P(w["proj.weight"]), P(Kf), P(s0), P(Vf), P(img), B, J, _st()), "ca_fold")
    m = [_c(sd[p + k]) for k in (".mlp.fc1.weight", ".mlp.fc1.bias", ".mlp.fc2.weight", ".mlp.fc2.bias")]
    out = torch.empty_like(xq)
    scratch = torch.empty_like(xq) if J > 23 else None
    fimg = ffn_image(m[0], m[2]) if packed else None
    _lib.check(lib.pmce_vertex_ca_mlp_pk_f32(P(xq), None, None, None, P(Kf), P(s0), P(Vf), P(w["proj.bias"]), P(GB), GB.shape[1],
                                             3, P(m[0]), P(m[1]), P(m[2]), P(m[3]), P(out), P(scratch), B, J,
                                             1 if split_f16 else 0, P(fimg), P(img), _st()), "vertex_ca_mlp")
    return out


def adaln_mlp(x, g, sd, p_norm, p_mlp, coor=None, vt_in=None, want_features=True, split_f16=False, packed=False):
    """x + Mlp(AdaLN(x)) on [B,431,64]; optional coordinate head (Wc[3,64], bc[3]) + vt_in residual.  packed: see cross_attn_block_vertex."""
    lib = _lib.load()
    x = _c(x)
    B = x.shape[0]
    GB = adaln_params(g, sd, [p_norm])
    w = [_c(sd[p_mlp + k]) for k in (".fc1.weight", ".fc1.bias", ".fc2.weight", ".fc2.bias")]
    y = torch.empty_like(x) if want_features else None
    vt_out = None
    Wc = bc = None
    if coor is not None:
        Wc, bc = _c(coor[0]), _c(coor[1])
        vt_in = _c(vt_in)
        vt_out = torch.empty_like(vt_in)
    img = ffn_image(w[0], w[2]) if packed else None
    _lib.check(lib.pmce_adaln_mlp_pk_f32(P(x), P(GB), GB.shape[1], 0, P(w[0]), P(w[1]), P(w[2]), P(w[3]), P(y), P(Wc), P(bc),
                                         P(vt_in), P(vt_out), B, 1 if split_f16 else 0, P(img), _st()), "adaln_mlp")
    return y, vt_out


def vertex_self_attn(x, g, sd, p, split_f16=False):
    """x + SA(AdaLN(x)) on [B,431,64], 2 heads (first line of Block.forward, CoevoDecoder.py:103); p = '...vertx_SA_FFN'.
    fp32 pipe: two launches (adaln_qkv, vertex_sa), returns (y, qkv).  split_f16: ONE launch (pmce_vertex_sab_split_f32: AdaLN + qkv
    product + attention + proj + residual, every contraction but the projection as three f16 matrix products; what a model in split_f16
    mode runs), returns (y, None) - no fp32 qkv exists."""
    lib = _lib.load()
    x = _c(x)
    B = x.shape[0]
    GB = adaln_params(g, sd, [p + ".norm1"])
    Wqkv = _c(sd[p + ".attn.qkv.weight"])
    y = torch.empty_like(x)
    if split_f16:
        img = torch.empty(lib.pmce_qkv_image_floats(), device=x.device)
        _lib.check(lib.pmce_qkv_pack_f16(P(Wqkv), P(img), _st()), "qkv_pack_f16")
        scratch = torch.empty(lib.pmce_vertex_sab_scratch_floats(B), device=x.device)
        _lib.check(lib.pmce_vertex_sab_split_f32(P(x), P(GB), GB.shape[1], 0, P(img), P(_c(sd[p + ".attn.qkv.bias"])),
                                                 P(_c(sd[p + ".attn.proj.weight"])), P(_c(sd[p + ".attn.proj.bias"])), P(scratch), P(y), B,
                                                 _st()), "vertex_sab")
        return y, None
    qkv = torch.empty(B, 431, 192, device=x.device)
    _lib.check(lib.pmce_adaln_qkv_f32(P(x), P(GB), GB.shape[1], 0, P(Wqkv), P(_c(sd[p + ".attn.qkv.bias"])), P(qkv), B, _st()), "adaln_qkv")
    _lib.check(lib.pmce_vertex_sa_f32(P(x), P(qkv), P(_c(sd[p + ".attn.proj.weight"])), P(_c(sd[p + ".attn.proj.bias"])), P(y), B, _st()),
               "vertex_sa")
    return y, qkv


def joint_stream(xq, xk, xv, g, sd, blk, stage, jt=None, split_f16=False):
    """Joint stream of a CoevoBlock given explicit q/k/v token sets (xq[B,J,64], xk/xv[B,431,64]):
    stage 1 = xq + CA (CoevoDecoder.py:83), 2 = + FFN (:85-86), 3 = + joint_SA_FFN (:187) (+ coords if jt).
    split_f16: the k / v products over the 431 vertex tokens (tokens_kv) in the three-product f16 form from the weights' image."""
    lib = _lib.load()
    xq, xk, xv = _c(xq), _c(xk), _c(xv)
    B, J, _ = xq.shape
    ca, sa = blk + ".joint_CA_FFN", blk + ".joint_SA_FFN"
    GB = adaln_params(g, sd, [ca + ".normq", ca + ".normk", ca + ".normv", ca + ".norm2", sa + ".norm1", sa + ".norm2"])
    kv = torch.empty(B, 431, 128, device=xq.device)
    Wk, Wv = _c(sd[ca + ".attn.wk.weight"]), _c(sd[ca + ".attn.wv.weight"])
    img = None
    if split_f16:   # (the generic form does not use proj_v2j_dim: any 64 x 64 weight fills the image's first slot)
        img = torch.empty(lib.pmce_tkv_image_floats(), device=xq.device)
        _lib.check(lib.pmce_tkv_pack_f16(P(Wk), P(Wk), P(Wv), P(img), _st()), "tkv_pack_f16")
    _lib.check(lib.pmce_tokens_kv_pk_f32(P(xk), P(xv), None, None, None, None, None, P(GB), GB.shape[1], 1, 2,
                                         P(Wk), P(_c(sd[ca + ".attn.wk.bias"])), P(Wv), P(_c(sd[ca + ".attn.wv.bias"])), P(kv), B,
                                         P(img), _st()), "tokens_kv")
    names = [ca + ".attn.wq.weight", ca + ".attn.wq.bias", ca + ".attn.proj.weight", ca + ".attn.proj.bias",
             ca + ".mlp.fc1.weight", ca + ".mlp.fc1.bias", ca + ".mlp.fc2.weight", ca + ".mlp.fc2.bias",
             sa + ".attn.qkv.weight", sa + ".attn.qkv.bias", sa + ".attn.proj.weight", sa + ".attn.proj.bias",
             sa + ".mlp.fc1.weight", sa + ".mlp.fc1.bias", sa + ".mlp.fc2.weight", sa + ".mlp.fc2.bias",
             blk + ".proj_joint_feat2coor.weight", blk + ".proj_joint_feat2coor.bias"]
    ws = [_c(sd[n]) for n in names]
    wptr = (C.c_void_p * 18)(*[w.data_ptr() for w in ws])
    inst = (C.c_int * 4)(0, 3, 4, 5)
    y = torch.empty_like(xq)
    pose = None
    if jt is not None:
        jt = _c(jt)
        pose = torch.empty_like(jt)
    _lib.check(lib.pmce_joint_stream_f32(P(xq), None, P(kv), P(GB), GB.shape[1], wptr, inst, P(jt), P(y), P(pose), B, J,
                                         stage, _st()), "joint_stream")
    return y, pose, kv


def joint_self_attn_block(x, g, sd, blk):
    """joint_SA_FFN alone: x + SA(AdaLN(x)); x + Mlp(AdaLN(x)) on [B,J,64], 8 heads (the reference's Block module,
    CoevoDecoder.py:102-105) - joint_stream's stage 4."""
    lib = _lib.load()
    x = _c(x)
    B, J, _ = x.shape
    sa = blk + ".joint_SA_FFN"
    GB = adaln_params(g, sd, [sa + ".norm1", sa + ".norm2"])
    names = [sa + ".attn.qkv.weight"] * 8 + [sa + ".attn.qkv.weight", sa + ".attn.qkv.bias", sa + ".attn.proj.weight",
                                             sa + ".attn.proj.bias", sa + ".mlp.fc1.weight", sa + ".mlp.fc1.bias",
                                             sa + ".mlp.fc2.weight", sa + ".mlp.fc2.bias"] + [sa + ".attn.qkv.weight"] * 2
    ws = [_c(sd[n]) for n in names]            # slots 0-7 (cross-attention block) and 16-17 (coordinate head) are unused
    wptr = (C.c_void_p * 18)(*[w.data_ptr() for w in ws])
    inst = (C.c_int * 4)(0, 0, 0, 1)
    y = torch.empty_like(x)
    _lib.check(lib.pmce_joint_stream_f32(P(x), None, None, P(GB), GB.shape[1], wptr, inst, None, P(y), None, B, J, 4, _st()),
               "joint_stream(stage 4)")
    return y


def coevo_block(model, k, joints, vt_in, g):
    """CoevoBlock k of a pmce_amd.models.{PMCE,CoevoDecoder} instance on explicit inputs (CoevoDecoder.py:175-191):
    -> (vt_out[B,431,3], joint_out[B,J,3] or None for k < 3)."""
    eng = model._ensure_packed()
    joints, vt_in, g = _c(joints), _c(vt_in), _c(g)
    B = joints.shape[0]
    vt_out = torch.empty_like(vt_in)
    j_out = torch.empty_like(joints) if k == 3 else None
    ws = eng.workspace(B)
    _lib.check(eng.lib.pmce_coevo_block_forward(eng.handle, k, P(joints), P(vt_in), P(g), P(vt_out), P(j_out), B,
                                                C.c_void_p(ws.data_ptr()), ws.numel(), _st()), "coevo_block_forward")
    return vt_out, j_out


def final_product(model, vt, g):
    """cam_mesh[B,6890,3] = upsample_conv(vt) + cat(linear_cur1..3(relu(g)))  (CoevoDecoder.py:238-244) through the packed
    [20670, 3360] operand of a pmce_amd model."""
    lib = _lib.load()
    eng = model._ensure_packed()
    vt, g = _c(vt), _c(g)
    B = vt.shape[0]
    KP = eng.packed["dec.final.weight"].shape[1]
    A = torch.empty(B, KP, device=vt.device, dtype=torch.float32)
    _lib.check(lib.pmce_build_final_operand_f32(P(g), P(vt), P(A), B, KP, _st()), "build_final_operand")
    return gemm_nt(A, eng.packed["dec.final.weight"], eng.packed["dec.final.bias"]).reshape(B, 6890, 3)


def j_regress(cam_mesh_m, j_regressor, scale=1000.0):
    """J_regressor[None] @ (cam_mesh*1000) (lib/core/base.py:223-225); j_regressor dense [R,6890] (numpy or tensor)."""
    from .assets import regressor_to_csr
    lib = _lib.load()
    mesh = _c(cam_mesh_m)
    B = mesh.shape[0]
    jr = j_regressor.detach().cpu().numpy() if isinstance(j_regressor, torch.Tensor) else np.asarray(j_regressor)
    indptr, indices, data = regressor_to_csr(jr)
    dev = mesh.device
    ip, ix, dt = (torch.from_numpy(a).to(dev) for a in (indptr, indices, data))
    out = torch.empty(B, jr.shape[0], 3, device=dev, dtype=torch.float32)
    _lib.check(lib.pmce_j_regress_f32(P(mesh), P(ip), P(ix), P(dt), P(out), B, jr.shape[0], mesh.shape[1], scale, _st()),
               "j_regress")
    return out
