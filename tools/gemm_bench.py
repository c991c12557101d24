#!/usr/bin/env python3
"""Times nfai_hip_gemm_f16 (the prefill GEMM of kernels_prefill.hip) in each tile / staging configuration on the
projection shapes of Llama-3.2-3B at T = 512 (fp32 output, no residual).  Companion of tools/gemm_ref_bench.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nfai_amd._lib import call  # noqa: E402
from nfai_amd.hip import HipBufferManager, ShaderProperty  # noqa: E402

NAMES = {0: "auto", 1: "128x64 reg", 2: "128x128 reg", 3: "128x128 glds2", 4: "128x128 glds3", 5: "128x64 glds2", 6: "128x64 glds3", 7: "128x64 glds4", 8: "128x64 glds3 pipe", 9: "128x64 glds3 bk128 pipe", 10: "128x64 glds2 bk128 pipe", 11: "128x128 glds2 pipe", 12: "128x80 glds3", 13: "128x48 glds3", 14: "128x80 glds4", 15: "128x48 glds4", 16: "128x64 roles a3 b6", 17: "128x64 roles a3 b9", 18: "128x128 roles a3 b4", 19: "128x128 roles a3 b6", 20: "128x80 roles a3 b6", 21: "256x128 w4 glds2", 22: "256x128 w4 glds3", 23: "256x128 w8 glds2", 24: "256x128 w8 glds3", 25: "128x128 ks2 glds3", 26: "128x64 ks2 glds3", 27: "128x80 ks2 glds3", 28: "128x48 ks2 glds3", 29: "128x48 ks2 glds4", 30: "128x96 w2x2 ks2 glds3", 31: "128x48 ks2 glds5", 32: "128x48 bk128 ks2 glds3", 33: "128x48 bk128 ks4 glds3", 34: "128x80 ks2 glds4", 35: "256x128 w8 ks2 glds3", 36: "128x48 ks2 glds6", 37: "128x80 ks2 glds5", 38: "128x128 ks2 glds4", 39: "128x64 bk128 ks2 glds3", 40: "128x80 bk128 ks2 glds3", 41: "128x96 ks2 glds4", 42: "128x64 ks2 glds4", 43: "256x128 w8 glds3 pipe", 44: "256x128 w8 glds2 pipe", 45: "128x48 bk128 ks2 pipe", 46: "128x80 bk128 ks2 pipe", 47: "64x96 2x2 bk128 ks2", 48: "64x96 2x2 ks2 glds4"}


def main():
    """--cold: every call reads a different copy of the weights (640 MB of copies per shape: beyond the 256 MB Infinity Cache), as
    in a prefill, where a projection's weights are touched once per 512 tokens; without it the 20 calls re-read one matrix that
    stays in the Infinity Cache."""
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cold = "--cold" in sys.argv
    T = int(args[0]) if args else 512
    mgr = HipBufferManager(0)
    shapes = {"3b": ((5120, 3072), (3072, 3072), (16384, 3072), (3072, 8192)), "8b": ((6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)),
              "1b": ((3072, 2048), (2048, 2048), (16384, 2048), (2048, 8192)),
              # split-K proxies of 3B's Wdown / Wo: the same number of 128 x 96 tiles as two K halves would have, half the K each
              "splitk-proxy": ((6144, 4096), (6144, 1536)),
              # round 4: Wdown (K = 8192) as FOUR K quarters on 256 x 128 tiles (2 x 24 tiles x 4 = 192 workgroups, half the operand bytes per
              # workgroup of the 128 x 48 tiling), Wo (K = 3072) as three K thirds; the same workgroup count as columns of one launch
              "splitk4-proxy": ((12288, 2048), (9216, 1024), (3072, 8192), (3072, 3072))}[os.environ.get("GEMM_MODEL", "3b")]
    for (N, K) in shapes:
        nw = max(2, -(-640 * 2**20 // (N * K * 2))) if cold else 1
        pa = ShaderProperty(mgr, T * K, np.float16)
        pws = [ShaderProperty(mgr, N * K, np.float16) for _ in range(nw)]
        pc = ShaderProperty(mgr, T * N, np.float32)
        r = np.random.Generator(np.random.PCG64(1))
        pa.SetValue(r.standard_normal(T * K).astype(np.float16))
        w = (0.02 * r.standard_normal(N * K)).astype(np.float16)
        for pw in pws:
            pw.SetValue(w)
        line = f"M={T} N={N} K={K}{' cold' if cold else ''}:"
        variants = [int(x) for x in os.environ["GEMM_VARIANTS"].split(",")] if os.environ.get("GEMM_VARIANTS") else (6, 7, 16, 17, 12, 20, 13, 3, 4, 18, 19, 0)
        ref = None
        for v in variants:
            if (v in (12, 14, 20, 27, 34, 37, 40, 46) and N % 80) or (v in (41, 47, 48) and N % 96) or (v in (13, 15, 28, 29, 31, 32, 33, 36, 45) and N % 48) or (v == 30 and N % 96):
                continue
            reps = max(20, nw)
            for i in range(3):
                call("nfai_hip_gemm_f16", mgr.handle, pa.handle, pws[i % nw].handle, 0, pc.handle, T, N, K, v)
            mgr.Synchronize()
            mgr.TimerBegin()
            for i in range(reps):
                call("nfai_hip_gemm_f16", mgr.handle, pa.handle, pws[(i + 3) % nw].handle, 0, pc.handle, T, N, K, v)
            us = mgr.TimerEnd() * 1e3 / reps
            out = pc.GetValue().astype(np.float64)
            if ref is None:
                ref = out
            elif np.abs(out - ref).max() > 1e-3 * np.abs(ref).max():
                line += " MISMATCH"
            line += f"  {NAMES[v]} {us:6.1f} us ({2.0 * T * N * K / us / 1e6:5.0f} TF)"
        print(line, flush=True)
        for p in [pa, pc] + pws:
            mgr.DestoryBuffer(p.buffer)


if __name__ == "__main__":
    main()
