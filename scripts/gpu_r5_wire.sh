mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_wire_world.py -k "world2_loop or world2_overlapped or async_wire" tests/test_gpu_retrieve.py tests/test_gpu_ivf.py -m gpu -q -p no:cacheprovider --durations=10 2>&1 | grep -v "^$" | tail -60 | cut -c1-400) > gpurun_out/t_wire.log 2>&1
tail -40 gpurun_out/t_wire.log
