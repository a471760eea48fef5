mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_config_size_oracle.py -m gpu -q -s -k bf16 -p no:cacheprovider > gpurun_out/r6_bf16_tests.log 2>&1
grep -n "touched rows; elements\|accumulators\|accumulator:\|^E  \|passed\|failed" gpurun_out/r6_bf16_tests.log | cut -c1-330
