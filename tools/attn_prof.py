#!/usr/bin/env python3
"""Times / profiles the fused decode attention op alone: tools/attn_prof.py [S] [H] [Hkv] [D] (defaults: 3B at 8192 positions)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nfai_amd import _lib  # noqa: E402
from nfai_amd._lib import call  # noqa: E402
from nfai_amd.hip import HipBufferManager, ShaderProperty  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
H, Hkv, D = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((2, 24), (3, 8), (4, 128)))
C = S
mgr = HipBufferManager(0)
r = np.random.Generator(np.random.PCG64(5))
pq, po = ShaderProperty(mgr, H * D), ShaderProperty(mgr, H * D)
pk, pv = ShaderProperty(mgr, C * Hkv * D), ShaderProperty(mgr, C * Hkv * D)
pq.SetValue(r.standard_normal(H * D).astype(np.float32))
pk.SetValue(r.standard_normal(C * Hkv * D).astype(np.float32))
pv.SetValue(r.standard_normal(C * Hkv * D).astype(np.float32))
for _ in range(5):
    call("nfai_hip_attn_decode", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, H, Hkv, D, S, C, _lib.F32)
mgr.Synchronize()
mgr.TimerBegin()
N = 100
for _ in range(N):
    call("nfai_hip_attn_decode", mgr.handle, pq.handle, pk.handle, pv.handle, po.handle, H, Hkv, D, S, C, _lib.F32)
us = mgr.TimerEnd() * 1e3 / N
print(f"S={S} H={H} Hkv={Hkv} D={D}: {us:.2f} us per call, {2.0 * S * Hkv * D * 4 / us / 1e6:.2f} TB/s of K+V bytes")
