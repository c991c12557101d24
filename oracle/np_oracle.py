"""Independent fp64 NumPy evaluation of the same path — cross-check for nfai_oracle.c.

Written from the op definitions in SURVEY.md §2.1 / §8a (which cite the reference GLSL), not from
the C file: vectorised, float64 throughout, spec-correct attention (causal over t < S).  The C
oracle (fp32, sequential order) must agree with this to fp32 rounding; tests/test_oracle.py states
the bounds.  Test infrastructure only.
"""
from __future__ import annotations

import numpy as np


def rmsnorm(x, g, eps):
    x = np.asarray(x, np.float64)
    return x / np.sqrt(np.mean(x * x) + eps) * np.asarray(g, np.float64)


def rope_freqs(rope_dims, base=500000.0, n_valid=None):
    i = np.arange(rope_dims // 2, dtype=np.float64)
    f = base ** (-(i / (rope_dims / 2)))
    if n_valid is not None:
        f[n_valid:] = 0.0
    return f


def rope(x, freqs, rope_dims, n_heads, head_dim, pos):
    x = np.asarray(x, np.float64).reshape(n_heads, head_dim // 2, 2).copy()
    npair = min(rope_dims, head_dim) // 2
    th = np.asarray(freqs, np.float64)[:npair] * pos
    c, s = np.cos(th), np.sin(th)
    a, b = x[:, :npair, 0].copy(), x[:, :npair, 1].copy()
    x[:, :npair, 0] = c * a - s * b
    x[:, :npair, 1] = s * a + c * b
    return x.reshape(-1)


def attention(q, Kc, Vc, H, Hkv, D, S):
    """scores -> softmax -> weighted V for one query vector; K/V caches [C][Hkv*D]."""
    q = np.asarray(q, np.float64).reshape(H, D)
    K = np.asarray(Kc, np.float64)[:S].reshape(S, Hkv, D)
    V = np.asarray(Vc, np.float64)[:S].reshape(S, Hkv, D)
    g = H // Hkv
    out = np.empty((H, D))
    for h in range(H):
        s = K[:, h // g, :] @ q[h] / np.sqrt(D)
        e = np.exp(np.clip(s - s.max(), -80.0, 80.0))
        out[h] = (e / e.sum()) @ V[:, h // g, :]
    return out.reshape(-1)


def silu(x):
    x = np.asarray(x, np.float64)
    return x / (1.0 + np.exp(-x))


def q4k_dequant(blocks, n):
    b = np.asarray(blocks, np.uint8).reshape(-1, 144)
    nb = b.shape[0]
    d = b[:, 0:2].copy().view(np.float16).astype(np.float64).reshape(nb)
    dmin = b[:, 2:4].copy().view(np.float16).astype(np.float64).reshape(nb)
    sc = b[:, 4:16].astype(np.int64)
    qs = b[:, 16:144].astype(np.int64).reshape(nb, 4, 32)
    scale = np.empty((nb, 8), np.int64)
    mn = np.empty((nb, 8), np.int64)
    for j in range(8):
        if j < 4:
            scale[:, j] = sc[:, j] & 63
            mn[:, j] = sc[:, j + 4] & 63
        else:
            scale[:, j] = (sc[:, j + 4] & 0xF) | ((sc[:, j - 4] >> 6) << 4)
            mn[:, j] = (sc[:, j + 4] >> 4) | ((sc[:, j] >> 6) << 4)
    out = np.empty((nb, 8, 32))
    out[:, 0::2, :] = qs & 0xF
    out[:, 1::2, :] = qs >> 4
    out = out * (d[:, None] * scale)[:, :, None] - (dmin[:, None] * mn)[:, :, None]
    return out.reshape(-1)[:n]


def q6k_dequant(blocks, n):
    b = np.asarray(blocks, np.uint8).reshape(-1, 210)
    nb = b.shape[0]
    ql = b[:, 0:128].astype(np.int64).reshape(nb, 2, 64)
    qh = b[:, 128:192].astype(np.int64).reshape(nb, 2, 32)
    sc = b[:, 192:208].copy().view(np.int8).astype(np.float64).reshape(nb, 2, 8)
    d = b[:, 208:210].copy().view(np.float16).astype(np.float64).reshape(nb)
    out = np.empty((nb, 2, 4, 32))
    out[:, :, 0, :] = (ql[:, :, 0:32] & 0xF) | (((qh >> 0) & 3) << 4)
    out[:, :, 1, :] = (ql[:, :, 32:64] & 0xF) | (((qh >> 2) & 3) << 4)
    out[:, :, 2, :] = (ql[:, :, 0:32] >> 4) | (((qh >> 4) & 3) << 4)
    out[:, :, 3, :] = (ql[:, :, 32:64] >> 4) | (((qh >> 6) & 3) << 4)
    out -= 32
    # scale index: is = l // 16 (+0, +2, +4, +6 for the four quarters)
    scq = np.empty((nb, 2, 4, 32))
    for quarter in range(4):
        for l in range(32):
            scq[:, :, quarter, l] = sc[:, :, l // 16 + 2 * quarter]
    out = out * scq * d[:, None, None, None]
    return out.reshape(-1)[:n]


class NpLlama:
    """fp64 whole-model evaluation with a growing KV cache; weights dict as OracleLlama."""

    def __init__(self, desc, weights):
        self.d = desc
        self.w = {k: np.asarray(v, np.float64) for k, v in weights.items()}
        rd, nf = desc.resolved()
        self.rd = rd
        self.freqs = rope_freqs(rd, desc.rope_base, nf)
        self.K = [np.zeros((desc.C, desc.Hkv * desc.D)) for _ in range(desc.L)]
        self.V = [np.zeros((desc.C, desc.Hkv * desc.D)) for _ in range(desc.L)]
        self.pos = 0

    def step(self, tok):
        d, w, p = self.d, self.w, self.pos
        x = w["token_embd.weight"][tok].copy()
        for l in range(d.L):
            b = f"blk.{l}."
            xn = rmsnorm(x, w[b + "attn_norm.weight"], d.eps)
            q = w[b + "attn_q.weight"] @ xn
            k = w[b + "attn_k.weight"] @ xn
            v = w[b + "attn_v.weight"] @ xn
            q = rope(q, self.freqs, self.rd, d.H, d.D, p)
            k = rope(k, self.freqs, self.rd, d.Hkv, d.D, p)
            self.K[l][p] = k
            self.V[l][p] = v
            att = attention(q, self.K[l], self.V[l], d.H, d.Hkv, d.D, p + 1)
            h = x + w[b + "attn_output.weight"] @ att
            hn = rmsnorm(h, w[b + "ffn_norm.weight"], d.eps)
            act = (w[b + "ffn_up.weight"] @ hn) * silu(w[b + "ffn_gate.weight"] @ hn)
            x = h + w[b + "ffn_down.weight"] @ act
        xn = rmsnorm(x, w["output_norm.weight"], d.eps)
        head = w.get("output.weight", w["token_embd.weight"])
        self.pos += 1
        return head @ xn
