mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|rror" | tail -10) > gpurun_out/t_all.log 2>&1
cat gpurun_out/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
(timeout 300 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1200)
