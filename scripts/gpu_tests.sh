mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40) > gpurun_out/t_all.log 2>&1
grep -E "passed|failed" gpurun_out/t_all.log
