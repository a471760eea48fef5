mkdir -p gpurun_out/prof
export TMPDIR=/tmp
(ESR_BENCH_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/sh_glove -o g -- python bench.py --workload glove --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/prof_sh_glove.log 2>&1
(ESR_BENCH_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/sh_inbatch -o g -- python bench.py --workload inbatch --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/prof_sh_inbatch.log 2>&1
find gpurun_out/prof -name "*.db" -delete; find gpurun_out/prof -name "*kernel_trace.csv" -delete
