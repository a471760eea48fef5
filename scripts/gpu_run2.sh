# One GPU-box pass: RCCL world-1 sharded test, the three bench workloads, HBM micro-bench, rocprofv3
# kernel stats and the two PMC passes.  Outputs under gpurun_out/ (scratch); summaries are copied to
# profiles/ by hand afterwards.
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/t_sharded.log 2>&1
for w in inbatch triplet glove; do (timeout 300 python bench.py --workload $w --steps 100 --warmup 10 2>&1 | tail -3) > gpurun_out/bench_$w.log 2>&1; done
(ESR_BENCH_SHARDED=1 timeout 300 python bench.py --steps 50 --warmup 5 2>&1 | tail -3) > gpurun_out/bench_sharded1.log 2>&1
(timeout 600 python benchmarks/hbm_micro.py 2>&1 | tail -60) > gpurun_out/hbm_micro.log 2>&1
for w in inbatch triplet glove; do
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats_$w -o $w -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -5) > gpurun_out/prof_stats_$w.log 2>&1
done
(timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_fetch -o inbatch -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -5) > gpurun_out/prof_fetch.log 2>&1
(timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_write -o inbatch -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -5) > gpurun_out/prof_write.log 2>&1
find gpurun_out/prof -name "*.db" -delete
# the raw kernel traces are large; keep the stats + counter CSVs, and a trimmed trace
for f in $(find gpurun_out/prof -name "*kernel_trace.csv"); do head -400 $f > $f.head; rm $f; done
find gpurun_out/prof -type f | head -50
du -sh gpurun_out
