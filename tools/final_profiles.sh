# The round's committed evidence, produced in one call on the GPU box: the driver's command, a longer timed region, and the
# rocprofv3 --kernel-trace --stats summaries of the headline workload and of the three configs[] workloads.
set -x
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round4_bench_n1.json 2> gpurun_out/round4_bench_n1.err
python bench.py --gpus 1 --steps 128 --warmup 8 --configs none > gpurun_out/round4_bench_n1_128.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
R=/root/repo
prof() { tag=$1; shift; rm -rf $R/gpurun_out/prof_$tag; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1; cp $(ls $R/gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1) $R/gpurun_out/round4_bench_${tag}_kernel_stats.csv; }
prof f16 --gpus 1 --steps 20 --warmup 5 --configs none
prof q4km --child --configs none --gpus 1 --steps 64 --warmup 8 --context 512 --sample-tokens 0 --profile-steps 3 --model llama-3.2-3b --quant q4_k_m --cpu-tokens 4
prof 1b --child --configs none --gpus 1 --steps 64 --warmup 8 --context 512 --sample-tokens 0 --profile-steps 3 --model llama-3.2-1b --quant f16 --cpu-tokens 16
prof 8bstage --child --configs none --gpus 1 --steps 64 --warmup 8 --context 512 --sample-tokens 0 --profile-steps 3 --model llama-3.1-8b --quant q4_k_m --stage-blocks 28:32 --cpu-tokens 8
cd $R; rm -rf gpurun_out/prof_f16 gpurun_out/prof_q4km gpurun_out/prof_1b gpurun_out/prof_8bstage
ls -la gpurun_out/round4_bench_*
