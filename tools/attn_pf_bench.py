#!/usr/bin/env python3
"""Times nfai_hip_attn_prefill (k_attn_prefill) at the head shapes of Llama-3.2-3B / 3.1-8B / 3.2-1B, T = 512 rows, first chunk.
NFAI_PREFILL_ATTN_KG=1|2|4 selects the key groups per query tile."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nfai_amd._lib import call  # noqa: E402
from nfai_amd.hip import HipBufferManager, ShaderProperty  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mgr = HipBufferManager(0)
for name, (H, Hkv, D) in {"3b": (24, 8, 128), "8b": (32, 8, 128), "1b": (32, 8, 64)}.items():
    Spad = (T + 127) // 128 * 128
    r = np.random.Generator(np.random.PCG64(3))
    q = ShaderProperty(mgr, T * H * D, np.float16)
    k = ShaderProperty(mgr, Hkv * Spad * D, np.float16)
    v = ShaderProperty(mgr, Hkv * Spad * D, np.float16)
    o = ShaderProperty(mgr, T * H * D, np.float16)
    q.SetValue(r.standard_normal(T * H * D).astype(np.float16))
    k.SetValue(r.standard_normal(Hkv * Spad * D).astype(np.float16))
    v.SetValue(r.standard_normal(Hkv * Spad * D).astype(np.float16))
    for _ in range(5):
        call("nfai_hip_attn_prefill", mgr.handle, q.handle, k.handle, v.handle, o.handle, T, H, Hkv, D, Spad, 0)
    mgr.Synchronize()
    mgr.TimerBegin()
    for _ in range(50):
        call("nfai_hip_attn_prefill", mgr.handle, q.handle, k.handle, v.handle, o.handle, T, H, Hkv, D, Spad, 0)
    print(f"{name} H={H} Hkv={Hkv} D={D} T={T}: {mgr.TimerEnd() * 1e3 / 50:6.2f} us per launch (KG={os.environ.get('NFAI_PREFILL_ATTN_KG', '2')})", flush=True)
