# 1B decode: tokens/s at long (512+) and short (16+) context for a few GEMV launch geometries (environment knobs of DESIGN 9)
run() { env "$@" python bench.py --model llama-3.2-1b --steps 256 --warmup 8 --configs none --no-cpu-baseline --sample-tokens 0 --profile-steps 2 > gpurun_out/sw1b.json 2>/dev/null; python - "$*" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sw1b.json").read().strip().splitlines()[-1])
k={x["class"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print(f"{sys.argv[1]:60s} long {d['value']:7.1f}  short {d['short_context']['tokens_per_s']:7.1f}  {k}", flush=True)
PY
}
run A=0
run NFAI_GEMV_SK_MODES=7
run NFAI_GEMV_SK_MODES=11
run NFAI_GEMV_SK_MODES=15
run NFAI_GEMV_SK_MODES=1
run NFAI_GEMV_BPC_QKV=1
run NFAI_GEMV_BPC_GATEUP=1
run NFAI_GEMV_BPC_QKV=1 NFAI_GEMV_BPC_GATEUP=1
run NFAI_GEMV_BPC_GATEUP=3
run NFAI_GEMV_WPB_GATEUP=8
run NFAI_GEMV_WPB_QKV=8
run NFAI_ATTN_WO=0
run A=0
