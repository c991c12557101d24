# Heralds (kernels_attn.hip): KiB per slice-less attention workgroup requested ahead of the following launches, A/B on one box
run() { q=$1; shift; env "$@" timeout -k 10 200 python bench.py --quant $q --steps 128 --warmup 8 --configs none --no-cpu-baseline --sample-tokens 0 --profile-steps 2 > gpurun_out/swh.json 2>gpurun_out/swh.err || { echo "FAILED $q $*"; tail -3 gpurun_out/swh.err; return 1; }; python - "$q $*" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/swh.json").read().strip().splitlines()[-1])
k={x["class"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print(f"{sys.argv[1]:44s} long {d['value']:7.1f}  short {d['short_context']['tokens_per_s']:7.1f}  {k}", flush=True)
PY
}
for q in ${QUANTS:-q4_k_m f16}; do
  for kb in ${KBS:-0 128 256 384 0}; do run $q NFAI_HERALD_KB=$kb || exit 1; done
  run $q NFAI_HERALD_KB=256 NFAI_HERALD_DOWN=0 || exit 1
done
