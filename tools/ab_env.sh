# A/B of one environment knob on the 3B decode bench (fp16 and/or Q4_K_M), alternating on one box:  bash tools/ab_env.sh NFAI_KV_WARM "0 1 0 1" "f16 q4_k_m"
knob=$1; vals=${2:-"0 1 0 1"}; quants=${3:-"f16"}
run() { q=$1; shift; env "$@" timeout -k 10 200 python bench.py --quant $q --steps 128 --warmup 8 --configs none --no-cpu-baseline --sample-tokens 0 --profile-steps 2 > gpurun_out/ab.json 2>gpurun_out/ab.err || { echo "FAILED $q $*"; tail -3 gpurun_out/ab.err; return 1; }; python - "$q $*" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
k={x["class"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print(f"{sys.argv[1]:40s} long {d['value']:7.1f}  short {d['short_context']['tokens_per_s']:7.1f}  {k}", flush=True)
PY
}
for q in $quants; do for v in $vals; do run $q $knob=$v || exit 1; done; done
