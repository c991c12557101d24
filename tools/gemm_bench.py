#!/usr/bin/env python3
"""Times nfai_hip_gemm_f16 (the prefill GEMM of kernels_prefill.hip) in each tile / staging configuration on the
projection shapes of Llama-3.2-3B at T = 512 (fp32 output, no residual).  Companion of tools/gemm_ref_bench.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nfai_amd._lib import call  # noqa: E402
from nfai_amd.hip import HipBufferManager, ShaderProperty  # noqa: E402

NAMES = {0: "auto", 1: "128x64 reg", 2: "128x128 reg", 3: "128x128 glds2", 4: "128x128 glds3", 5: "128x64 glds2", 6: "128x64 glds3", 7: "128x64 glds4", 8: "128x64 glds3 pipe", 9: "128x64 glds3 bk128 pipe", 10: "128x64 glds2 bk128 pipe", 11: "128x128 glds2 pipe"}


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    mgr = HipBufferManager(0)
    for (N, K) in ((5120, 3072), (3072, 3072), (16384, 3072), (3072, 8192)):
        pa, pw = ShaderProperty(mgr, T * K, np.float16), ShaderProperty(mgr, N * K, np.float16)
        pc = ShaderProperty(mgr, T * N, np.float32)
        r = np.random.Generator(np.random.PCG64(1))
        pa.SetValue(r.standard_normal(T * K).astype(np.float16))
        pw.SetValue((0.02 * r.standard_normal(N * K)).astype(np.float16))
        line = f"M={T} N={N} K={K}:"
        for v in (1, 6, 8, 9, 10, 3, 11, 0, 6):
            for _ in range(3):
                call("nfai_hip_gemm_f16", mgr.handle, pa.handle, pw.handle, 0, pc.handle, T, N, K, v)
            mgr.Synchronize()
            mgr.TimerBegin()
            for _ in range(20):
                call("nfai_hip_gemm_f16", mgr.handle, pa.handle, pw.handle, 0, pc.handle, T, N, K, v)
            us = mgr.TimerEnd() * 1e3 / 20
            line += f"  {NAMES[v]} {us:6.1f} us ({2.0 * T * N * K / us / 1e6:5.0f} TF)"
        print(line, flush=True)
        for p in (pa, pw, pc):
            mgr.DestoryBuffer(p.buffer)


if __name__ == "__main__":
    main()
