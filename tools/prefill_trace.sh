#!/bin/bash
# usage (on the GPU box): tools/prefill_trace.sh <tag>  -> prints per-kernel averages of one MFMA prefill (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pf_$1
rm -rf $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o pf -- python3 $GRAFT_REPO_ROOT/tools/prefill_prof.py > $OUT.log 2>&1
python3 - "$OUT" <<PY
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
g=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "at::native" in n or "rocclr" in n: continue
    key=(n[n.index("k_gemm_f16"):][:44] if "k_gemm_f16" in n else n[:36], r["Grid_Size_X"], r["Grid_Size_Y"])
    g[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(g.items(), key=lambda kv:-sum(kv[1]))[:9]:
    print(k, len(v), round(sum(v)/len(v),1), "us avg", round(sum(v)/2/1e3,2), "ms/prefill")
PY
