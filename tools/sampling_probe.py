#!/usr/bin/env python3
"""Per-token wall time of the host loops of bench.py's `sampling_path` on a synthetic model (default 1B): blocking greedy,
decode_topk feeding ids[0] back, decode_topk + TopPFromCandidates (vectorised / loop form).  Prints mean, median and max per loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench as B  # noqa: E402
from nfai_amd import _lib, synth  # noqa: E402
from nfai_amd.hip import HipBufferManager  # noqa: E402
from nfai_amd.llama_model import LlamaModel, SamplingUtils  # noqa: E402

dims = synth.BY_NAME[sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"]
torch.cuda.set_device(0)
w = B.gen_weights_hbm(torch, dims, (0, dims.L), True, True)
mgr = HipBufferManager(0)
m = LlamaModel(mgr, synth.make_metadata(dims), B.as_model_tensors(_lib, w), 600,
               dims=dict(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5, rope_dims=dims.D, rope_base=500000.0))
m.SetToken(5)
m.Enqueue(300)
mgr.Synchronize()
n = 128
rng = np.random.default_rng(7)


def loop(name, step):
    m.SetPos(300)
    tok = 77
    step(tok)
    m.SetPos(300)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        tok = step(tok)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"{name:28s} mean {ts.mean():8.1f} us  median {np.median(ts):8.1f}  max {ts.max():8.1f}  -> {1e6 / ts.mean():7.1f} tokens/s", flush=True)


def topk_vec(t):
    ids, probs = m.StepTopK(t)
    return SamplingUtils.TopPFromCandidates(ids, probs, rng=rng)


def topk_loop(t):
    ids, probs = m.StepTopK(t)
    return SamplingUtils.TopPFromCandidatesLoop(ids, probs, 0.95, float(rng.random(dtype=np.float32)))


for rep in range(2):
    loop("blocking greedy", lambda t: m.Step(t, want_logits=False)[1])
    loop("decode_topk, ids[0]", lambda t: int(m.StepTopK(t)[0][0]))
    loop("decode_topk + nucleus (vec)", topk_vec)
    loop("decode_topk + nucleus (loop)", topk_loop)
