mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -30) > gpurun_out/t_all.log 2>&1
for w in inbatch triplet glove; do
  (timeout 300 python bench.py --workload $w --steps 200 --warmup 20 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_$w.log 2>&1
done
(timeout 300 python bench.py --precision f32 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_inbatch_f32.log 2>&1
(ESR_BENCH_SHARDED=1 timeout 300 python bench.py --steps 50 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_sharded1.log 2>&1
tail -4 gpurun_out/t_all.log
