#!/usr/bin/env python3
"""Generates tests/golden/*.npz — seeded inputs and the ORACLE's outputs for the hot path.

The reference (NFAI) ships no golden vectors and cannot run here (C#/.NET + Vulkan), so these
fixtures are produced by this repository's CPU restatement (oracle/nfai_oracle.c, fp32, reference
summation order) — "parity unpinned" by the reference, see DESIGN.md §2.  They pin the oracle
against silent drift (tests/test_golden.py re-derives them on CPU) and give the GPU tests a fixed
target that does not depend on the oracle being built on the GPU box.

    python tests/golden/make_golden.py        # rewrites the .npz files (deterministic)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle as orc  # noqa: E402
from nfai_amd import synth  # noqa: E402


def ops_fixture():
    r = np.random.Generator(np.random.PCG64(2024))
    H, Hkv, D, C, S, E, F, N = 8, 2, 64, 24, 19, 512, 768, 96
    out = {}
    x = r.standard_normal(E).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(E)).astype(np.float32)
    W = (0.03 * r.standard_normal((N, E))).astype(np.float16)
    out.update(x=x, gamma=g, W=W, rmsnorm=orc.rmsnorm(x, g, 1e-5), gemv=orc.gemv_f16w(W, x))
    q = r.standard_normal(H * D).astype(np.float32)
    Kc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    Vc = r.standard_normal((C, Hkv * D)).astype(np.float32)
    freqs = orc.rope_freqs(D)
    sc = orc.attn_scores(q, Kc, H, Hkv, D, S)
    w = orc.attn_softmax(sc)
    out.update(q=q, Kc=Kc, Vc=Vc, freqs=freqs, rope_q_pos7=orc.rope(q, freqs, D, H, D, 7), scores=sc, softmax=w,
               wsum=orc.attn_wsum(w, Vc, H, Hkv, D, S), dims=np.array([H, Hkv, D, C, S], np.int32))
    a = (3 * r.standard_normal(F)).astype(np.float32)
    b = r.standard_normal(F).astype(np.float32)
    out.update(a=a, b=b, silu=orc.silu(a), mul=orc.mul(a, b), add=orc.add(a, b), argmax=np.array([orc.argmax(a)], np.int64))
    # K-quant codecs: blocks + their dequantisation (ggml layout)
    wq = (0.02 * r.standard_normal((4, 512))).astype(np.float32)
    b4, b6 = orc.quantize_q4k(wq), orc.quantize_q6k(wq)
    out.update(kq_src=wq, q4k_blocks=b4, q4k_dequant=orc.dequant_q4k(b4, wq.size), q6k_blocks=b6,
               q6k_dequant=orc.dequant_q6k(b6, wq.size))
    return out


def model_fixture(dims, seed, n_prompt, n_new, C):
    """Weights are regenerated from (dims, seed) by synth.make_weights, so only tokens and outputs are stored."""
    w = synth.make_weights(dims, seed=seed, std=0.05)
    m = orc.OracleLlama(orc.LlamaDesc(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, C=C), w)
    prompt = synth.make_tokens(dims, n_prompt, seed=seed + 1)
    logits = []
    for t in prompt:
        logits.append(m.step(int(t)))
    toks = []
    lg = logits[-1]
    for _ in range(n_new):  # greedy: ArgMax in place of the stochastic TopP (SamplingUtils.cs:43-57)
        t = orc.argmax(lg)
        toks.append(t)
        lg = m.step(t)
        logits.append(lg)
    return dict(seed=np.array([seed]), prompt=prompt, greedy=np.array(toks, np.uint32), logits=np.stack(logits).astype(np.float32),
                hidden_last=m.hidden(), k_last_l0=m.kcache(0)[n_prompt + n_new - 1].copy(), C=np.array([C]))


def main():
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops_fixture())
    np.savez_compressed(os.path.join(HERE, "tiny_llama.npz"), **model_fixture(synth.TINY, 101, 5, 8, 16))
    np.savez_compressed(os.path.join(HERE, "tiny_llama_d128.npz"), **model_fixture(synth.TINY_D128, 202, 4, 8, 16))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
