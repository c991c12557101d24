#!/usr/bin/env python3
"""Where a wave of the prefill GEMM (k_gemm_f16_glds) spends its cycles: waiting for its K tile (s_waitcnt vmcnt), at the
workgroup barrier, issuing the next tile's LDS-DMA, reading fragments + MFMA.  Diagnostic build only
(`python -m nfai_amd.build --stamps`; shader-clock deltas accumulated per wave, written at the end of the launch).

    python3 tools/gemm_stamps.py [variant] > profiles/round2_gemm_stamps.json
Cold weights: every launch reads its own copy (640 MB of copies per shape), as in a prefill."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NFAI_HIP_LIB"] = os.path.join(ROOT, "nfai_amd", "csrc", "libnfai_hip_stamps.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nfai_amd import _lib  # noqa: E402
from nfai_amd.hip import HipBufferManager, ShaderProperty  # noqa: E402


def main():
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    torch.cuda.init()
    lib = _lib.load()
    mgr = HipBufferManager(0)
    WAVES, WORDS = 4096, 8
    n_slots = 64
    buf = torch.zeros(n_slots * WAVES * WORDS, device="cuda", dtype=torch.int64)
    lib.nfai_hip_debug_stamps_install.argtypes = [C.c_void_p, C.c_uint32]
    lib.nfai_hip_debug_stamps_install(C.c_void_p(buf.data_ptr()), n_slots)
    out = {"what": __doc__.split("\n\n")[0], "variant": variant, "unit": "shader-clock cycles per wave (sum over the K loop), median over the waves of the last launch",
           "shapes": {}}
    T = 512
    for name, (N, K) in {"q|k|v": (5120, 3072), "Wo": (3072, 3072), "gate|up as plain GEMM": (16384, 3072), "Wdown": (3072, 8192)}.items():
        nw = max(2, -(-640 * 2**20 // (N * K * 2)))
        pa = ShaderProperty(mgr, T * K, np.float16)
        pws = [ShaderProperty(mgr, N * K, np.float16) for _ in range(nw)]
        pc = ShaderProperty(mgr, T * N, np.float32)
        r = np.random.Generator(np.random.PCG64(1))
        pa.SetValue(r.standard_normal(T * K).astype(np.float16))
        w = (0.02 * r.standard_normal(N * K)).astype(np.float16)
        for pw in pws:
            pw.SetValue(w)
        for i in range(n_slots - 2):  # the slot table has n_slots entries: the last launches overwrite nothing of interest
            _lib.call("nfai_hip_gemm_f16", mgr.handle, pa.handle, pws[i % nw].handle, 0, pc.handle, T, N, K, variant)
        mgr.Synchronize()
        torch.cuda.synchronize()
        st = buf.cpu().numpy().reshape(n_slots, WAVES, WORDS)
        last = max(s for s in range(n_slots) if st[s, :, 5].any())
        t = st[last]
        t = t[t[:, 5] > 0].astype(np.float64)
        total = t[:, 5] - t[:, 0]
        kt = float(np.median(t[:, 6]))
        med = lambda a: float(np.median(a))  # noqa: E731
        out["shapes"][f"{name} M={T} N={N} K={K}"] = {
            "waves": int(t.shape[0]), "k_tiles": kt, "total": med(total), "wait_for_tile_vmcnt": med(t[:, 1]), "barrier": med(t[:, 2]),
            "issue_dma": med(t[:, 3]), "fragment_reads_and_mfma": med(t[:, 4]), "epilogue_until_stores_acknowledged": med(t[:, 7]),
            "per_k_tile": {"wait": round(med(t[:, 1]) / kt, 1), "barrier": round(med(t[:, 2]) / kt, 1), "issue": round(med(t[:, 3]) / kt, 1),
                           "reads_and_mfma": round(med(t[:, 4]) / kt, 1), "total": round(med(total) / kt, 1)}}
        buf.zero_()
        lib.nfai_hip_debug_stamps_install(C.c_void_p(buf.data_ptr()), n_slots)
        for p in [pa, pc] + pws:
            mgr.DestoryBuffer(p.buffer)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
