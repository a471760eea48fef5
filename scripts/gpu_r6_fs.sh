mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stl_loop.py tests/test_gpu_api.py tests/test_gpu_config_size_oracle.py -m gpu -q -k "inbatch or stl or in_batch" -p no:cacheprovider 2>&1 | grep -v "^$" | tail -30) > gpurun_out/r6_fs_tests.log 2>&1
tail -6 gpurun_out/r6_fs_tests.log
AB_LIBS="scripts/libib2h_R5.so scripts/libib2h_PRE.so esrecsys_amd/libesr_hip.so" bash scripts/gpu_r6_ab.sh
