mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_retrieve.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40) > gpurun_out/t_retr.log 2>&1
tail -5 gpurun_out/t_retr.log
