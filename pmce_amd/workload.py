"""Analytical work of the path per clip: what bench.py divides measured times into.  Pure arithmetic on the
architecture's dimensions; tests check it against the oracle's own count."""
from __future__ import annotations


def flops_per_clip(num_joint=17, embed_dim=256, depth=3):
    """Reference-equivalent FLOPs of one clip (2*MACs of every matmul as the reference computes them,
    dead code included) — SURVEY §8d; J=17,C=256 -> 3.55e9."""
    T, J, C, F_, D, V, VF, H = 16, num_joint, embed_dim, 2048, 64, 431, 6890, 1024
    tok = T * J
    lifter = 2 * T * F_ * C + 2 * tok * 2 * C
    per_blk = 2 * tok * C * 3 * C + 2 * tok * C * C + 2 * 2 * tok * C * 2 * C
    attn_s = T * 8 * 2 * 2 * J * J * (C // 8)
    attn_t = J * 8 * 2 * 2 * T * T * (C // 8)
    lifter += depth * (2 * per_blk + attn_s + attn_t) + 2 * tok * C * 3
    gru = 2 * (2 * T * 3 * H * F_ + 2 * T * 3 * H * H) * 2
    ada = 72 * 2 * F_ * D

    def ca(nq, nk):
        return 2 * nq * D * D * 2 + 2 * nk * D * D * 2 + 2 * 2 * nq * nk * D + 2 * 2 * nq * D * 4 * D

    def sa(n):
        return 2 * n * D * 3 * D + 2 * 2 * n * n * D + 2 * n * D * D + 2 * 2 * n * D * 4 * D

    blk = ca(J, V) + ca(V, J) + sa(J) + sa(V) + 2 * V * D * D + 2 * J * D * D + 2 * (J + V) * 3 * D * 2
    up = 2 * 3 * V * 3 * VF + 3 * 2 * 2 * H * VF
    return dict(lifter=lifter, gru=gru, coevo=3 * blk + ada, upsample=up, total=lifter + gru + 3 * blk + ada + up)
