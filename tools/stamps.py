#!/usr/bin/env python3
"""Where the fixed cost of the short decode kernels goes: in-kernel time stamps (s_memrealtime, 100 MHz, one clock for the
whole chip) of every wave of every launch of one decode token, on the DIAGNOSTIC build libnfai_hip_stamps.so
(`python -m nfai_amd.build --stamps`; the product library holds no stamp).

    python3 tools/stamps.py [--quant f16] [--context 512] [--tokens 6] > profiles/round2_stamps_f16.json

Per kernel class (all 28 blocks of the last token pooled; microseconds; med [p10, p90] over waves unless stated):
  span            last store acknowledged (any wave) - first wave start                       (~ rocprof's kernel duration)
  start_skew      a wave's start - the launch's first wave start                              (dispatch ramp)
  issued          loads issued (activations + first two weight steps) - wave start
  x_ready         x normalised and in LDS - wave start                                         (dependency on the previous kernel)
  first_consumed  first weight step multiplied - wave start                                    (first-byte latency under load)
  stream          last FMA + epilogue stores issued - first step consumed                      (the streaming part)
  drain           stores acknowledged - stores issued
  tail            launch's last acknowledgement - this wave's acknowledgement                  (imbalance: idle at the end)
  gap_to_next     next launch's first wave start - this launch's last acknowledgement          (kernel boundary)
The diagnostic build's fences forbid overlaps the product kernel has: read SHARES, not lengths (MI355X guide)."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NFAI_HIP_LIB"] = os.path.join(ROOT, "nfai_amd", "csrc", "libnfai_hip_stamps.so")

import numpy as np  # noqa: E402


def pct(a):
    a = np.asarray(a, np.float64)
    return [round(float(np.median(a)), 3), round(float(np.percentile(a, 10)), 3), round(float(np.percentile(a, 90)), 3)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quant", default="f16")
    ap.add_argument("--model", default="llama-3.2-3b")
    ap.add_argument("--context", type=int, default=512)
    ap.add_argument("--tokens", type=int, default=6)
    ap.add_argument("--engine", type=int, default=0, help="1: one engine launch per block (kernels_engine.hip), stamps per wave role")
    a = ap.parse_args()
    import torch
    import bench as B
    from nfai_amd import _lib, synth
    from nfai_amd.hip import HipBufferManager
    from nfai_amd.llama_model import LlamaModel
    assert os.path.exists(os.environ["NFAI_HIP_LIB"]), "build it first: python -m nfai_amd.build --stamps"
    lib = _lib.load()
    dims = synth.BY_NAME[a.model]
    torch.cuda.set_device(0)
    weights = B.gen_weights_hbm(torch, dims, (0, dims.L), True, True, quant=a.quant)
    n_slots = dims.L * 6 + 8
    WAVES, WORDS = 4096, 8
    buf = torch.zeros(n_slots * WAVES * WORDS, device="cuda", dtype=torch.int64)
    lib.nfai_hip_debug_stamps_install.argtypes = [C.c_void_p, C.c_uint32]
    lib.nfai_hip_debug_stamps_info.argtypes = [C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.nfai_hip_debug_stamps_install(C.c_void_p(buf.data_ptr()), n_slots)
    mgr = HipBufferManager(0)
    Cc = a.context + a.tokens + 2
    m = LlamaModel(mgr, synth.make_metadata(dims), B.as_model_tensors(_lib, weights), Cc, engine=bool(a.engine), dims=dict(
        E=dims.E, L=dims.L, H=dims.H, Hkv=dims.Hkv, D=dims.D, F=dims.F, V=dims.V, eps=1e-5, rope_dims=dims.D, rope_base=500000.0))
    m.SetToken(128000 % dims.V)
    m.Enqueue(a.context + a.tokens)   # the graph is captured at the first token: slots = launch order within a token
    mgr.Synchronize()
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(n_slots, WAVES, WORDS).astype(np.float64) * 0.01  # 100 MHz ticks -> us
    infos = []
    used = C.c_uint32()
    for s in range(n_slots):
        name = C.create_string_buffer(48)
        g, b = C.c_uint32(), C.c_uint32()
        if lib.nfai_hip_debug_stamps_info(s, name, C.byref(g), C.byref(b), C.byref(used)) != 0:
            break
        infos.append((name.value.decode(), g.value, b.value))
    # one token = the slots captured in the graph, in launch order
    per_cls = {}
    launches = []
    eng = {"stream": [], "control": []}
    eng_span = []
    NS = 8  # stream waves per workgroup (kernels_engine.hip: ENG_NS), the rest are control waves
    for s, (name, grid, block) in enumerate(infos):
        if name != "engine":
            continue
        nw = block // 64
        t = st[s][: grid * nw].reshape(grid, nw, WORDS)
        t0 = t[t > 0].min()
        rel = np.where(t > 0, t - t0, np.nan)
        eng["stream"].append(rel[:, :NS, :].reshape(-1, 8))
        eng["control"].append(rel[:, NS:, :].reshape(-1, 8))
        eng_span.append(float(np.nanmax(rel)))
    if eng_span:
        def med(rows, k):
            a_ = np.concatenate(rows)[:, k]
            a_ = a_[~np.isnan(a_)]
            return (pct(a_) + [round(float(a_.max()), 3)]) if a_.size else None
        names = {"stream": ["x of op0 (Wo) seen", "op0 done", "x of op1 (gate|up) seen", "op1 done", "x of op2 (Wdown) seen", "op2 done",
                            "x of op3 (next q|k|v) seen", "op3 done"],
                 "control": ["start", "att / x in LDS", "local stream waves done op0", "h gathered + normalised", "local done op1", "act gathered",
                             "local done op2", "x' gathered + normalised"]}
        print(json.dumps({"what": "engine launch (kernels_engine.hip): time of each event since the launch's first stamp, us, [median, p10, p90, max] "
                                  "over the waves of that role of all workgroups of all blocks' launches in one token", "model": a.model,
                          "launches": len(eng_span), "span_us": pct(eng_span),
                          "roles": {r: {names[r][k]: med(v, k) for k in range(len(names[r]))} for r, v in eng.items()}}, indent=1))
        return
    aw = {"attention waves": [], "wo waves": [], "span": []}
    for s, (name, grid, block) in enumerate(infos):
        if name != "attn_wo":
            continue
        t = st[s]
        t0 = t[t[:, 0] > 0][:, 0].min()
        rel = np.where(t > 0, t - t0, np.nan)
        aw["attention waves"].append(rel[:1024])
        aw["wo waves"].append(rel[1024:1024 + grid * 4])
        aw["span"].append(float(np.nanmax(rel)))
    for s, (name, grid, block) in enumerate(infos):
        t = st[s]
        live = t[:, 0] > 0
        if not live.any():
            continue
        t = t[live]
        n_last = 8 if name == "attn_decode" else 6
        end = np.where(t[:, n_last - 1] > 0, t[:, n_last - 1], t[:, :n_last].max(axis=1))
        if name == "attn_decode":  # non-merging blocks stop at stamp 6
            end = t[:, :8].max(axis=1)
        launches.append(dict(slot=s, name=name, grid=grid, block=block, t0=float(t[:, 0].min()), t1=float(end.max()), t=t, end=end))
    for i, L in enumerate(launches):
        t, end = L["t"], L["end"]
        d = per_cls.setdefault(L["name"] + f" grid={L['grid']} block={L['block']}", {k: [] for k in
                               ("span", "start_skew", "issued", "x_ready", "first_consumed", "stream", "drain", "tail", "gap_to_next",
                                "scores", "softmax", "v_and_partials", "partials_acked", "ticket", "merge")})
        d["span"].append(L["t1"] - L["t0"])
        d["start_skew"].extend(t[:, 0] - L["t0"])
        d["tail"].extend(L["t1"] - end)
        if i + 1 < len(launches) and launches[i + 1]["slot"] == L["slot"] + 1:
            d["gap_to_next"].append(launches[i + 1]["t0"] - L["t1"])
        if L["name"] == "attn_decode":
            d["issued"].extend(t[:, 1] - t[:, 0])
            d["scores"].extend(t[:, 2] - t[:, 0])
            d["softmax"].extend(t[:, 3] - t[:, 2])
            d["v_and_partials"].extend(t[:, 4] - t[:, 3])
            mg = t[t[:, 7] > 0]
            if (t[:, 6] > 0).any():   # ticket hand-off (NFAI_ATTN_POLL=0)
                d["partials_acked"].extend(t[:, 5] - t[:, 4])
                d["ticket"].extend(t[:, 6] - t[:, 5])
                d["merge"].extend(mg[:, 7] - mg[:, 6])
            else:                     # granule hand-off: every block merges its share of the outputs after stamp 4
                # the launch ends with its SLOWEST wave: per launch, the latest time any wave reaches each stamp (since the launch's first)
                for kk, nm in ((1, "requested"), (2, "scores"), (3, "softmax"), (4, "published"), (5, "all_seen"), (7, "stored")):
                    col_ = t[:, kk][t[:, kk] > 0]
                    if col_.size:
                        d.setdefault("slowest_wave_" + nm, []).append(float(col_.max() - L["t0"]))
                        d.setdefault("fastest_wave_" + nm, []).append(float(col_.min() - L["t0"]))
                d.setdefault("own_slice_published", []).extend(t[:, 4] - L["t0"])
                d.setdefault("all_granules_seen_after_own_publish", []).extend(mg[:, 5] - mg[:, 4])
                d.setdefault("merge_and_store", []).extend(mg[:, 7] - mg[:, 5])
        else:
            # who finishes late?  waves are flushed in order (workgroup * waves per workgroup + wave): by XCD (workgroup id mod 8: the
            # dispatcher deals workgroups round-robin over the eight XCDs) and by workgroup
            wpb_ = max(1, L["block"] // 64)
            if L["t"].shape[0] == L["grid"] * wpb_:
                rel_end = (end - L["t0"]).reshape(L["grid"], wpb_)
                wg_end = rel_end.max(axis=1)
                xcd_mean = np.array([wg_end[x::8].mean() for x in range(8)])
                d.setdefault("xcd_mean_end", []).append(xcd_mean)
                d.setdefault("xcd_spread_of_means", []).append(float(xcd_mean.max() - xcd_mean.min()))
                d.setdefault("wg_end_p50", []).append(float(np.median(wg_end)))
                d.setdefault("wg_end_p90", []).append(float(np.percentile(wg_end, 90)))
                d.setdefault("within_wg_wave_spread", []).append(float(np.median(rel_end.max(axis=1) - rel_end.min(axis=1))))
                lo = np.argsort(wg_end)[-8:]
                d.setdefault("slowest8_wg_mod8", []).append([int(v) % 8 for v in lo])
            d["issued"].extend(t[:, 1] - t[:, 0])
            if (t[:, 6] > 0).any():  # fp16 GEMV with RMSNorm: own x arrived / workgroup sum known
                ok_ = t[:, 6] > 0
                d.setdefault("own_x_arrived", []).extend(t[ok_, 6] - t[ok_, 0])
                d.setdefault("norm_sum_known", []).extend(t[ok_, 7] - t[ok_, 0])
            d["x_ready"].extend(t[:, 2] - t[:, 0])
            d["first_consumed"].extend(t[:, 3] - t[:, 0])
            d["stream"].extend(t[:, 4] - t[:, 3])
            d["drain"].extend(t[:, 5] - t[:, 4])
    out = {"what": __doc__.split("\n\n")[0], "model": a.model, "quant": a.quant, "positions": [a.context + a.tokens - 1],
           "unit": "us; [median, p10, p90] over the waves of all launches of the class in one token (span / gap_to_next: over launches)",
           "clock": "s_memrealtime, 100 MHz (10 ns resolution)", "launches_in_token": len(launches), "classes": {}}
    for k, d in per_cls.items():
        special = {}
        if "xcd_mean_end" in d:  # mean end of a workgroup (us since the launch's first stamp) by XCD, and where the eight slowest workgroups of a launch sit
            special["xcd_mean_end_us_by_xcd"] = [round(float(v), 2) for v in np.mean(np.stack(d.pop("xcd_mean_end")), axis=0)]
            special["slowest8_workgroups_xcd_histogram"] = np.bincount(np.array(d.pop("slowest8_wg_mod8")).ravel(), minlength=8).tolist()
        out["classes"][k] = {kk: pct(v) for kk, v in d.items() if len(v)}
        out["classes"][k].update(special)
        out["classes"][k]["launches"] = len(d["span"])
    if aw["span"]:
        def col(rows, k):
            a_ = np.concatenate(rows)[:, k]
            a_ = a_[~np.isnan(a_)]
            return (pct(a_) + [round(float(a_.max()), 3)]) if a_.size else None
        an = ["start", "K, V, q requested", "scores in LDS", "softmax done", "own granules issued", "all granules of the share seen, merge weights in LDS", "-", "share of the merged output published"]
        wn = ["start", "weights requested", "attention waves of the workgroup done with barriers", "quarter of the attention output seen",
              "all four quarters in LDS", "multiplied, reduced, stores issued", "stores acknowledged"]
        out["attn_wo"] = {"what": "fused attention + Wo launch: time of each event since the launch's first stamp, us, [median, p10, p90, max] over the waves of the role, all blocks' launches of one token",
                          "span_us": pct(aw["span"]),
                          "attention waves": {an[k]: col(aw["attention waves"], k) for k in range(8) if an[k] != "-"},
                          "wo waves": {wn[k]: col(aw["wo waves"], k) for k in range(7)}}
    if launches:
        out["token_span_us"] = round(launches[-1]["t1"] - launches[0]["t0"], 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
