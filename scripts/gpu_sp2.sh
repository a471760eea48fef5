mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_spotify.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -15) > gpurun_out/t_sp.log 2>&1
tail -4 gpurun_out/t_sp.log
python benchmarks/spotify_step.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/spotify_step.json
for w in triplet glove; do (timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400); done
