"""Profile helper (run under rocprofv3): one warm + one timed T-token MFMA prefill of a synthetic model.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/prefill_prof.py [model] [T]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench as B  # noqa: E402
from nfai_amd import _lib, synth  # noqa: E402
from nfai_amd.hip import HipBufferManager  # noqa: E402
from nfai_amd.llama_model import LlamaModel  # noqa: E402

dims = synth.BY_NAME[sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-3b"]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w = B.gen_weights_hbm(torch, dims, (0, dims.L), True, True)
mgr = HipBufferManager(0)
m = LlamaModel(mgr, synth.make_metadata(dims), B.as_model_tensors(_lib, w), T + 8, max_batch=T,
               dims=dict(E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5,
                         rope_dims=dims.D, rope_base=500000.0))
toks = synth.make_tokens(dims, T, seed=99)
m.Prefill(toks, want_logits=False)
m.Reset()
mgr.Synchronize()
mgr.TimerBegin()
m.Prefill(toks, want_logits=False)
print("prefill ms", mgr.TimerEnd())
