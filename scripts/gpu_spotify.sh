export TMPDIR=/tmp
mkdir -p gpurun_out/spotify
timeout 600 python -m pytest tests/test_gpu_spotify.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
python benchmarks/spotify_step.py 2>&1 | grep "^{" | tail -1 | tee gpurun_out/spotify/spotify_step.json | cut -c1-500
rm -rf /tmp/sp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o t -- python benchmarks/spotify_step.py > /dev/null 2>&1
cp $(find /tmp/sp -name "*kernel_stats.csv" | head -1) gpurun_out/spotify/spotify_kernel_stats.csv
python3 scripts/prof_stats.py /tmp/sp 14
