# Round-end measurement pass: bench lines for every workload, micro-benchmarks, rocprofv3 kernel stats.
# Summaries are copied into profiles/rNN/ by hand afterwards (gpurun_out/ is scratch).
mkdir -p gpurun_out/prof gpurun_out/round
export TMPDIR=/tmp
R=gpurun_out/round
for w in inbatch triplet glove retrieve; do
  (timeout 400 python bench.py --workload $w 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_$w.json
done
(timeout 300 python bench.py --precision f32 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_inbatch_f32.json
(timeout 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_inbatch_bf16x3.json
(ESR_IB2H_REF=rowmax timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_inbatch_f16x2_rowmax_pass.json
for w in inbatch triplet glove; do
  (timeout 300 python bench.py --workload $w --ids zipf --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_${w}_zipf.json
done
(timeout 300 python bench.py --workload triplet --graph --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_triplet_hipgraph.json
(timeout 300 python bench.py --table-dtype bf16 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_inbatch_bf16_tables.json
for w in inbatch triplet glove; do
  (ESR_BENCH_SHARDED=1 timeout 300 python bench.py --workload $w --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_sharded_world1_$w.json
done
(ESR_BENCH_SHARDED=1 timeout 600 python bench.py --rows 12500000 --table-dtype bf16 --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1) > $R/bench_sharded_world1_config4_share.json
(timeout 600 python benchmarks/hbm_micro.py 2>&1 | grep -v amdgpu.ids) > $R/hbm_micro.jsonl
(timeout 600 python benchmarks/retrieve_bench.py 2>&1 | grep -v amdgpu.ids) > $R/retrieve_bench.jsonl
(timeout 300 python benchmarks/mfma_peak.py 2>&1 | grep -v amdgpu.ids) > $R/mfma_peak.jsonl
(timeout 300 python benchmarks/reader_bench.py 2>&1 | grep -v amdgpu.ids) > $R/reader_bench.jsonl
(timeout 600 python benchmarks/spotify_step.py 2>&1 | grep -v amdgpu.ids | tail -1) > $R/spotify_step.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats_inbatch_bf16x3 -o x -- python bench.py --precision bf16x3 --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_stats_inbatch_bf16x3.log 2>&1
f=$(find gpurun_out/prof/stats_inbatch_bf16x3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/inbatch_bf16x3_kernel_stats.csv
for w in inbatch triplet glove retrieve; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats_$w -o $w -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_stats_$w.log 2>&1
  f=$(find gpurun_out/prof/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${w}_kernel_stats.csv
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats_spotify -o sp -- python benchmarks/spotify_step.py > gpurun_out/prof_stats_spotify.log 2>&1
f=$(find gpurun_out/prof/stats_spotify -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/spotify_kernel_stats.csv
find gpurun_out/prof -name "*.db" -delete; find gpurun_out/prof -name "*kernel_trace.csv" -delete
du -sh gpurun_out; ls $R
