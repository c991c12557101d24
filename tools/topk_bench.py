#!/usr/bin/env python3
"""Time nfai_hip_topk (k_topk: the device half of SamplingUtils.TopP) over V = 128,256 logits for several k: wall time of the blocking
call (launch + 528-byte read-back + synchronise), median of 200.  The k = 1 figure is the fixed cost of the call."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nfai_amd._lib import call  # noqa: E402
from nfai_amd.hip import HipBufferManager, ShaderProperty  # noqa: E402

mgr = HipBufferManager(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128256
v = (3.0 * np.random.default_rng(1).standard_normal(n)).astype(np.float32)
pv = ShaderProperty(mgr, n)
pv.SetValue(v)
for k in (1, 2, 8, 20, 40, 64):
    ids, probs = np.empty(k, np.uint32), np.empty(k, np.float32)
    ts = []
    for _ in range(220):
        t0 = time.perf_counter()
        call("nfai_hip_topk", mgr.handle, pv.handle, n, 0.5, k, ids.ctypes.data_as(C.POINTER(C.c_uint32)), probs.ctypes.data_as(C.POINTER(C.c_float)))
        ts.append(time.perf_counter() - t0)
    print(f"n={n} k={k:3d}: median {1e6 * float(np.median(ts[20:])):8.1f} us per blocking call")
