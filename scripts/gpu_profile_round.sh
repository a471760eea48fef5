# Round-end measurement pass: bench lines for the three workloads (+ exact-f32 in-batch), HBM micro-bench,
# rocprofv3 kernel stats, PMC traffic for the dominant kernels.  Summaries are copied into profiles/rNN/.
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
for w in inbatch triplet glove; do
  (timeout 300 python bench.py --workload $w 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_$w.json
done
(timeout 300 python bench.py --precision f32 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_inbatch_f32.json
(timeout 600 python benchmarks/hbm_micro.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/hbm_micro.jsonl
for w in inbatch triplet glove; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats_$w -o $w -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_stats_$w.log 2>&1
done
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_fetch -o inbatch -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_write -o inbatch -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_write.log 2>&1
find gpurun_out/prof -name "*.db" -delete; find gpurun_out/prof -name "*kernel_trace.csv" -delete
du -sh gpurun_out
