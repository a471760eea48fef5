mkdir -p gpurun_out/prof
export TMPDIR=/tmp
(timeout 600 python benchmarks/retrieve_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/retrieve_bench.jsonl
cat gpurun_out/retrieve_bench.jsonl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/retr -o retr -- python benchmarks/retrieve_bench.py --reps 2 --ks 500 > gpurun_out/prof_retr.log 2>&1
find gpurun_out/prof -name "*.db" -delete; find gpurun_out/prof -name "*kernel_trace.csv" -delete
find gpurun_out/prof/retr -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-8 {} | head -14'
