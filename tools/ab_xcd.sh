# A/B of the XCD-weighted row dealing (NFAI_XCD_DEAL, NFAI_XCD_GAIN) on one box: tokens/s and per-kernel us of bench.py's live replay
for rep in 1 2; do for v in "0 1.0" "1 1.0" "1 1.5" "1 2.0" "1 3.0"; do set -- $v; NFAI_XCD_DEAL=$1 NFAI_XCD_GAIN=$2 python bench.py --steps 128 --warmup 8 --configs none --no-cpu-baseline --sample-tokens 0 > gpurun_out/ab_xd.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/ab_xd.json").read().strip().splitlines()[-1])
k={x["class"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print("XD=$1 gain=$2 rep $rep", round(d["value"],1), k, d["config"]["xcd_row_shares"]["shares"])
PY
done; done
