# 3B fp16 decode: a few launch-geometry knobs A/B on one box (tokens/s long / short context, per-kernel us)
run() { env "$@" python bench.py --steps 128 --warmup 8 --configs none --no-cpu-baseline --sample-tokens 0 --profile-steps 2 > gpurun_out/sw3b.json 2>/dev/null; python - "$*" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sw3b.json").read().strip().splitlines()[-1])
k={x["class"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print(f"{sys.argv[1]:44s} long {d['value']:7.1f}  short {d['short_context']['tokens_per_s']:7.1f}  {k}", flush=True)
PY
}
run A=0
run NFAI_GEMV_SK_RW=8
run NFAI_GEMV_SK_RW=16
run NFAI_GEMV_SK_MODES=1
run NFAI_GEMV_SK_MODES=0
run NFAI_XCD_DEAL=0
run A=0
