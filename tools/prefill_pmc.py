#!/usr/bin/env python3
"""Hardware counters of the prefill kernels (one 512-token MFMA prefill of Llama-3.2-3B fp16, tools/prefill_prof.py), one
rocprofv3 pass per counter group (--pmc with --kernel-trace only), summarised per kernel into profiles/<tag>.json.

    python3 tools/prefill_pmc.py round2_prefill_pmc          (on the GPU box, from the repository root)

mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs that ran waves) / (duration x 2.4 GHz)   — fraction of the dense fp16 MFMA peak
l2_read_tbps = TCP_TCC_READ_REQ_sum x 64 B / duration                                  — what the CUs pulled out of the L2s
l2_hit     = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
lds_*      = fractions of the waves' lifetime (the SQ_*_LDS counters count in units of 4 cycles summed over waves)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_WAVES"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum"],
]


def short(name):
    if "k_gemm_f16" in name or "k_gemm_kq" in name:
        i = name.index("k_gemm")
        return name[i:].split("(")[0]
    for k in ("k_rmsnorm_rows", "k_rope_store_rows", "k_softmax_causal_rows", "k_silu_mul_rows", "k_f32_to_f16", "k_kv_to_f16"):
        if k in name:
            return k
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "round2_prefill_pmc"
    extra = sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for gi, grp in enumerate(GROUPS):
        d = os.path.join(ROOT, "gpurun_out", f"pmc_prefill_{gi}")
        subprocess.run(["rm", "-rf", d])
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                            "python3", os.path.join(ROOT, "tools", "prefill_prof.py")] + extra, cwd="/tmp", env=env, capture_output=True, text=True)
        print(f"pass {gi}: rc={r.returncode} {r.stdout.strip().splitlines()[-1:] }", flush=True)
        if r.returncode != 0:
            print(r.stderr[-2000:], flush=True)
            sys.exit(1)
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        rows = list(csv.DictReader(open(f[0])))
        # one row per (dispatch, counter)
        for row in rows:
            k = short(row["Kernel_Name"])
            if not k:
                continue
            key = (k, row["Grid_Size"])
            agg[key][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            meta[key] = dict(wg=row.get("Workgroup_Size"), lds=row.get("LDS_Block_Size"), vgpr=row.get("VGPR_Count"))
        kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
        if kt:
            for row in csv.DictReader(open(kt[0])):
                k = short(row["Kernel_Name"])
                if k:
                    # (the counter file's Grid_Size is the launch's total thread count; a batched GEMM has a z dimension)
                    gsz = (str(int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1) or 1) * int(row.get("Grid_Size_Z", 1) or 1))
                           if "Grid_Size_X" in row else row.get("Grid_Size"))
                    agg[(k, gsz)]["_dur_us_%d" % gi].append(
                        (int(row["Dispatch_Id"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
        subprocess.run(["rm", "-rf", d])  # raw traces: tens of MB (gpurun copies back at most 64 MiB)
    out = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 tools/prefill_prof.py   (one pass per group: "
                      + " | ".join(" ".join(g) for g in GROUPS) + "); tools/prefill_pmc.py",
           "note": __doc__.split("\n\n")[2], "kernels": {}}
    for key, ctrs in sorted(agg.items(), key=lambda kv: kv[0]):
        k, grid = key
        e = {"grid_threads": grid, **meta.get(key, {})}
        vals = {}
        for c, lst in ctrs.items():
            # a counter that is summed over XCDs shows up once per dispatch; average over dispatches
            per = collections.defaultdict(float)
            for did, v in lst:
                per[did] += v
            vals[c] = sum(per.values()) / max(len(per), 1)
            e["launches"] = len(per)
        durs = [v for c, v in vals.items() if c.startswith("_dur_us_")]
        dur = durs[0] if durs else None
        e["avg_us_by_pass"] = [round(v, 2) for v in durs]
        e["counters"] = {c: round(v, 1) for c, v in vals.items() if not c.startswith("_")}
        if dur:
            def d_of(gi):
                return vals.get("_dur_us_%d" % gi, dur)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
                waves = vals.get("SQ_WAVES", 0)
                simds = min(1024.0, float(int(grid)) / 64.0) if grid else 1024.0
                e["mfma_util"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / (d_of(0) * 2400.0), 4)
            if "SQ_WAVE_CYCLES" in vals and vals["SQ_WAVE_CYCLES"] > 0:
                e["lds_active_frac_of_wave"] = round(vals.get("SQ_ACTIVE_INST_LDS", 0) / vals["SQ_WAVE_CYCLES"], 4)
                e["lds_wait_frac_of_wave"] = round(vals.get("SQ_WAIT_INST_LDS", 0) / vals["SQ_WAVE_CYCLES"], 4)
                e["any_wait_frac_of_wave"] = round(vals.get("SQ_WAIT_INST_ANY", 0) / vals["SQ_WAVE_CYCLES"], 4)
            if "SQ_LDS_IDX_ACTIVE" in vals and vals["SQ_LDS_IDX_ACTIVE"] > 0:
                e["lds_bank_conflict_frac"] = round(vals.get("SQ_LDS_BANK_CONFLICT", 0) / vals["SQ_LDS_IDX_ACTIVE"], 4)
            if "TCC_HIT_sum" in vals:
                e["l2_hit"] = round(vals["TCC_HIT_sum"] / max(vals["TCC_HIT_sum"] + vals.get("TCC_MISS_sum", 0), 1), 4)
                e["l2_req_tbps_128B"] = round(vals.get("TCC_REQ_sum", 0) * 128 / (d_of(2) * 1e6), 2)
                e["hbm_side_read_tbps_64B"] = round(vals.get("TCC_EA0_RDREQ_sum", 0) * 64 / (d_of(2) * 1e6), 2)
            if "TCP_TCC_READ_REQ_sum" in vals:
                e["l1_to_l2_read_tbps_64B"] = round(vals["TCP_TCC_READ_REQ_sum"] * 64 / (d_of(3) * 1e6), 2)
        out["kernels"][f"{k} grid={grid}"] = e
    # what bench.py's prefill.mfma_util cites: MFMA-busy fraction of the GEMM launches, time-weighted, and per kernel
    gemms = {k: e for k, e in out["kernels"].items() if k.startswith("k_gemm") and "mfma_util" in e and e.get("avg_us_by_pass")}
    tot = sum(e["avg_us_by_pass"][0] * e["launches"] for e in gemms.values())
    out["mfma_busy_frac"] = round(sum(e["mfma_util"] * e["avg_us_by_pass"][0] * e["launches"] for e in gemms.values()) / tot, 4) if tot else None
    out["mfma_busy_frac_by_kernel"] = {k: {"mfma_busy_frac": e["mfma_util"], "us": e["avg_us_by_pass"][0], "launches": e["launches"],
                                           "l2_hit": e.get("l2_hit"), "lds_wait_frac_of_wave": e.get("lds_wait_frac_of_wave")} for k, e in gemms.items()}
    path = os.path.join(ROOT, "gpurun_out", tag + ".json")  # copy into profiles/ afterwards (gpurun merges only gpurun_out/)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
