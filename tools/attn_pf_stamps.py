#!/usr/bin/env python3
"""Where a wave of the prefill attention (k_attn_prefill) spends its cycles: waiting for its K / V^T tile (s_waitcnt vmcnt), at the
workgroup barrier, issuing the next tile's LDS-DMA, in scores + softmax, in P.V.  Diagnostic build only
(`python -m nfai_amd.build --stamps`; shader-clock deltas accumulated per wave, written at the end of the launch).

    python3 tools/attn_pf_stamps.py [T] > profiles/round3_attn_prefill_stamps.json
Shapes of Llama-3.2-3B: 24 query heads, 8 kv heads of 128, first chunk (pos0 = 0)."""
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
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    H, Hkv, D = 24, 8, 128
    Spad = (T + 63) // 64 * 64
    torch.cuda.init()
    lib = _lib.load()
    mgr = HipBufferManager(0)
    WAVES, WORDS, n_slots = 4096, 8, 16
    buf = torch.zeros(n_slots * WAVES * WORDS, device="cuda", dtype=torch.int64)
    lib.nfai_hip_debug_stamps_install.argtypes = [C.c_void_p, C.c_uint32]
    lib.nfai_hip_debug_stamps_install(C.c_void_p(buf.data_ptr()), n_slots)
    r = np.random.Generator(np.random.PCG64(3))
    q = ShaderProperty(mgr, T * H * D, np.float16)
    k = ShaderProperty(mgr, Hkv * Spad * D, np.float16)
    v = ShaderProperty(mgr, Hkv * Spad * D, np.float16)
    o = ShaderProperty(mgr, T * H * D, np.float16)
    q.SetValue(r.standard_normal(T * H * D).astype(np.float16))
    k.SetValue(r.standard_normal(Hkv * Spad * D).astype(np.float16))
    v.SetValue(r.standard_normal(Hkv * Spad * D).astype(np.float16))
    for _ in range(n_slots - 2):
        _lib.call("nfai_hip_attn_prefill", mgr.handle, q.handle, k.handle, v.handle, o.handle, T, H, Hkv, D, Spad, 0)
    mgr.Synchronize()
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(n_slots, WAVES, WORDS)
    last = max(s for s in range(n_slots) if st[s, :, 6].any())
    t = st[last]
    t = t[t[:, 6] > 0].astype(np.float64)
    qb = (t[:, 7].astype(np.int64) & 0xFFFFFFFF)
    nsteps = (t[:, 7].astype(np.int64) >> 32)
    t0 = t[:, 0].min()
    out = {"what": __doc__.split("\n\n")[0], "T": T, "unit": "shader-clock cycles per wave; medians over the waves of one launch, by query block (64 rows; the last sees the most keys)",
           "launch_cycles_first_start_to_last_end": float(t[:, 6].max() - t0), "by_query_block": {}}
    med = lambda a: float(np.median(a))  # noqa: E731
    for b in sorted(set(qb.tolist())):
        m = qb == b
        a = t[m]
        n = float(np.median(nsteps[m]))
        out["by_query_block"][str(b)] = {
            "waves": int(m.sum()), "steps": n, "start_after_launch": med(a[:, 0] - t0), "total": med(a[:, 6] - a[:, 0]),
            "wait_for_tile": med(a[:, 1]), "barrier": med(a[:, 2]), "issue_dma_and_q_loads": med(a[:, 3]), "scores_softmax": med(a[:, 4]), "pv": med(a[:, 5]),
            "per_step": {"wait": round(med(a[:, 1]) / n, 1), "barrier": round(med(a[:, 2]) / n, 1), "issue": round(med(a[:, 3]) / n, 1),
                         "scores_softmax": round(med(a[:, 4]) / n, 1), "pv": round(med(a[:, 5]) / n, 1)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
