#!/usr/bin/env python3
"""Latency of the prompt phase as the provider path runs it (nfai_hip_llama_ingest: T prompt tokens -> K / V rows, MFMA prefill) against
the reference's way (T blocking decode steps), Llama-3.2-3B fp16 synthetic weights: median of 5 calls per T, from an empty cache."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench as B  # noqa: E402
from nfai_amd import _lib, synth  # noqa: E402
from nfai_amd.hip import HipBufferManager  # noqa: E402
from nfai_amd.llama_model import LlamaModel  # noqa: E402

dims = synth.BY_NAME[sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-3b"]
torch.cuda.set_device(0)
w = B.gen_weights_hbm(torch, dims, (0, dims.L), True, True)
mgr = HipBufferManager(0)
m = LlamaModel(mgr, synth.make_metadata(dims), B.as_model_tensors(_lib, w), 1100, max_batch=512,
               dims=dict(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5, rope_dims=dims.D, rope_base=500000.0))
toks = synth.make_tokens(dims, 1024, seed=99)
m.Ingest(toks[:64])
for T in (1, 8, 16, 39, 64, 128, 256, 512, 1024):
    ts = []
    for _ in range(5):
        m.Reset()
        mgr.Synchronize()
        t0 = time.perf_counter()
        m.Ingest(toks[:T])
        ts.append(time.perf_counter() - t0)
    ing = float(np.median(ts)) * 1e3
    n = min(T, 64)
    m.Reset()
    mgr.Synchronize()
    t0 = time.perf_counter()
    for t in toks[:n]:
        m.Step(int(t), want_logits=False)
    tbt = (time.perf_counter() - t0) * 1e3 * T / n
    print(f"T={T:5d}: ingest {ing:8.3f} ms ({T / ing:8.1f} tokens/ms)   token by token {tbt:9.2f} ms   x{tbt / ing:6.1f}", flush=True)
