mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -30) > gpurun_out/t_all.log 2>&1
for w in inbatch triplet glove; do
(ESR_BENCH_SHARDED=1 timeout 300 python bench.py --workload $w --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_sharded1_$w.log 2>&1
done
grep -E "passed|failed" gpurun_out/t_all.log
