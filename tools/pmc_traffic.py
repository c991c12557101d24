#!/usr/bin/env python3
"""HBM traffic per kernel launch from the PMC counters, as MI355X_MICROARCH.md prescribes: two SEPARATE rocprofv3 passes
(--pmc FETCH_SIZE, --pmc WRITE_SIZE; with --kernel-trace only), hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (on gfx950
FETCH_SIZE reports half of a 16 B/lane coalesced stream; WRITE_SIZE is exact).

    tools/pmc_traffic.py f16     -> gpurun_out/round4_pmc_traffic.json       (copied to profiles/; bench.py reads `traffic` from it)
    tools/pmc_traffic.py q4_k_m  -> gpurun_out/round4_pmc_traffic_q4km.json

The profiled command is bench.py at ITS OWN context (512 prompt tokens through the MFMA prefill, then decode steps at
positions 512...), so the counters belong to the launches the benchmark times.  Runs on the GPU box."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
quant = sys.argv[1] if len(sys.argv) > 1 else "f16"
out_json = os.path.join(ROOT, "gpurun_out", "round4_pmc_traffic.json" if quant == "f16" else "round4_pmc_traffic_q4km.json")
bench = ["python3", os.path.join(ROOT, "bench.py"), "--quant", quant, "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--profile-steps", "0", "--configs", "none", "--sample-tokens", "0"]
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(ROOT, "gpurun_out", f"pmc_{ctr}_{quant}")
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + bench,
                   cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    vals[ctr] = {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}
    subprocess.run(["rm", "-rf", d])  # the raw traces are tens of MB; gpurun copies back at most 64 MiB

# algorithmic bytes per launch of the GEMV kernels at 3B: rows x K x bytes per weight.  The kernel names carry the
# launch geometry (k_gemv<WT, MODE, UPW, U, GUARD, NORM>), which tuning changes: match on weight type and epilogue mode instead.
E, F, HD, KD = 3072, 8192, 3072, 1024
V = 128256


def algorithmic_bytes(name):
    import re
    sk = re.match(r"nfai::k_gemv_sk<(\d+), (\d+), (\d+), (\d+), (\w+)>", name)  # split-K form <WT, MODE, CH, RW, NORM>
    if sk and sk.group(1) == "1" and int(sk.group(2)) == 1:
        return E * F * 2   # the only fp16 RESIDUAL launch of its own at 3B is Wdown (Wo rides in k_attn_wo); K = 8192 = 16 waves x 1 chunk
    m = re.match(r"nfai::k_gemv<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+)(?:, \w+)?>", name)
    if m and m.group(1) == "1":
        mode, u = int(m.group(2)), int(m.group(4))
        if mode == 3:
            return 2 * F * E * 2
        if mode == 2:
            return (HD + 2 * KD) * E * 2
        if mode == 1:
            return E * F * 2 if u == 4 else E * HD * 2   # Wdown streams K = 8192 (four 1-KiB loads per row and step), Wo K = 3072
        if mode == 0 and m.group(6) == "true":
            return V * E * 2
    kq = re.match(r"nfai::k_gemv_kqt<(\d+), (\d+), ", name)   # <weight layout (112 Q4_K, 114 Q6_K, 115 q|k Q4_K + v Q6_K), epilogue mode, ...>
    if kq:
        wt, mode = int(kq.group(1)), int(kq.group(2))
        bpw = {112: 144 / 256, 114: 210 / 256}
        if mode == 3:
            return int(2 * F * E * bpw[wt])
        if mode == 2:
            return int((HD + 2 * KD) * E * bpw[112]) if wt == 112 else int((HD + KD) * E * bpw[112] + KD * E * bpw[114])
        if mode == 1:   # Wo (K = 3072) and Wdown (K = 8192) differ in the waves-per-tile parameter
            return int(E * HD * bpw[wt]) if re.match(r"nfai::k_gemv_kqt<\d+, 1, 1,", name) else int(E * F * bpw[wt])
        if mode == 0:
            return int(V * E * bpw[wt])
    return None


out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --quant %s --steps 8 --warmup 2 "
                  "--no-cpu-baseline --profile-steps 0 --configs none --sample-tokens 0   (second pass with --pmc WRITE_SIZE; the benchmark's own context of 512 tokens); tools/pmc_traffic.py" % quant,
       "engine": False, "dominant_kernel": None,
       "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE reads exactly 1/2 of a 16 B/lane coalesced stream on gfx950 "
                     "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact",
       "model": "llama-3.2-3b " + ("fp16" if quant == "f16" else "Q4_K_M"), "kernels": {}}
for k, (fv, n) in vals["FETCH_SIZE"].items():
    if not k.startswith("nfai::") or "k_gemm" in k or "_rows" in k:
        continue
    wv = vals["WRITE_SIZE"].get(k, (0.0, 0))[0]
    e = {"FETCH_SIZE_KB_avg": fv, "launches": n, "WRITE_SIZE_KB_avg": wv, "hbm_bytes_per_launch": (2 * fv + wv) * 1024}
    alg = algorithmic_bytes(k)
    if alg:
        e["algorithmic_bytes_per_launch"] = alg
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / alg
        if alg == (2 * F * E * 2 if quant == "f16" else 2 * F * E * 144 // 256):
            out["dominant_kernel"] = k
    out["kernels"][k] = e
json.dump(out, open(out_json, "w"), indent=1)
for k, e in out["kernels"].items():
    if "traffic_over_algorithmic" in e:
        print(k, round(e["hbm_bytes_per_launch"]), e["algorithmic_bytes_per_launch"], round(e["traffic_over_algorithmic"], 4))
