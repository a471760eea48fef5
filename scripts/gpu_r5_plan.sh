mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_ivf.py tests/test_gpu_retrieve.py tests/test_gpu_wire_world.py -m gpu -q -p no:cacheprovider --durations=8 2>&1 | grep -v "^$" | tail -40 | cut -c1-400) > gpurun_out/t_plan.log 2>&1
tail -30 gpurun_out/t_plan.log
