mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stl_loop.py tests/test_gpu_api.py -m gpu -q -k "inbatch or stl or in_batch" -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40) > gpurun_out/r6_pct_tests.log 2>&1
tail -8 gpurun_out/r6_pct_tests.log
cp esrecsys_amd/libesr_hip.so scripts/libib2h_BASE.so
IB2H_ROUNDS=2 IB2H_VARIANTS="R5 BASE" bash scripts/gpu_r6_probe.sh
