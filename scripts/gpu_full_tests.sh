mkdir -p gpurun_out/r6
(timeout 5400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r6/gpu_tests_full.log 2>&1
tail -6 gpurun_out/r6/gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
