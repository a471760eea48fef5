mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_spotify.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40) > gpurun_out/t_sp.log 2>&1
tail -25 gpurun_out/t_sp.log
