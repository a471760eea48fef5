mkdir -p gpurun_out
(timeout 1800 python -m pytest tests/test_gpu_retrieve.py tests/test_gpu_ivf.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^$" | tail -30) > gpurun_out/r6_retr_tests.log 2>&1
tail -12 gpurun_out/r6_retr_tests.log
for m in f16r f16x2; do echo "== $m"; timeout 300 python scripts/retr_ktime.py $m 2>&1 | grep -v amdgpu.ids; done
