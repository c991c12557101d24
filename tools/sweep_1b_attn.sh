# 1B decode: the attention + Wo launch's knobs (tuned at 3B) re-swept at 1B
run() { env "$@" python bench.py --model llama-3.2-1b --steps 256 --warmup 8 --configs none --no-cpu-baseline --sample-tokens 0 --profile-steps 2 > gpurun_out/sw1b.json 2>/dev/null; python - "$*" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sw1b.json").read().strip().splitlines()[-1])
k={x["class"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print(f"{sys.argv[1]:44s} long {d['value']:7.1f}  short {d['short_context']['tokens_per_s']:7.1f}  attn+wo {k.get('attn+wo')}", flush=True)
PY
}
run A=0
run NFAI_ATTN_WO_DELAY=0
run NFAI_ATTN_WO_DELAY=20
run NFAI_ATTN_WO_DELAY=36
run NFAI_ATTN_WO_DELAY=80
run NFAI_ATTN_WO_DELAY=120
run NFAI_ATTN_MIN_CHUNK=48
run NFAI_ATTN_MIN_CHUNK=64
run NFAI_ATTN_MIN_CHUNK=16
run A=0
