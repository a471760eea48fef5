# Round-4 measurement pass: bench lines, kernel statistics (rocprofv3 --kernel-trace --stats) and HBM traffic (separate
# --pmc FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only).  Everything lands in gpurun_out/r4/; copied to profiles/r4/.
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
R=gpurun_out/r4
line() { out=$1; shift; (timeout 600 env "$@" 2>&1 | grep '^{' | tail -1) > $R/$out; }
# --- the driver's command: every line it prints
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{') > $R/bench_driver_cmd_lines.jsonl
tail -1 $R/bench_driver_cmd_lines.jsonl > $R/bench_default_line.json
# --- single legs
line bench_inbatch_200.json python bench.py --no-secondary --no-cpu-baseline
line bench_inbatch_zipf.json python bench.py --ids zipf --no-secondary --no-cpu-baseline
line bench_inbatch_bf16_tables.json python bench.py --table-dtype bf16 --no-secondary --no-cpu-baseline
line bench_triplet.json python bench.py --workload triplet --steps 400 --warmup 20 --no-cpu-baseline
line bench_triplet_reference_loop_shape.json ESR_STL_LOOP=presorted python bench.py --workload triplet --steps 400 --warmup 20 --no-cpu-baseline --no-kernel-timing
line bench_triplet_zipf.json python bench.py --workload triplet --ids zipf --steps 400 --warmup 20 --no-cpu-baseline
line bench_glove.json python bench.py --workload glove --no-cpu-baseline
line bench_glove_zipf.json python bench.py --workload glove --ids zipf --no-cpu-baseline
line bench_glove_b2048.json python bench.py --workload glove --batch 2048 --steps 800 --warmup 32 --no-cpu-baseline
line bench_glove_b2048_finalize_launch.json ESR_GLOVE_FIN_FUSED=0 python bench.py --workload glove --batch 2048 --steps 800 --warmup 32 --no-cpu-baseline
line bench_retrieve_exact.json python bench.py --workload retrieve --rows 1048576 --precision f32 --steps 3 --warmup 1
line bench_retrieve_f16x2.json python bench.py --workload retrieve --rows 1048576 --steps 3 --warmup 1
for w in inbatch triplet glove; do line bench_replicated_world1_$w.json ESR_BENCH_SHARDED=1 ESR_BENCH_PARALLELISM=replicated python bench.py --workload $w --steps 200 --warmup 24 --no-cpu-baseline; done
line bench_sharded_world1_config4_share.json ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 python bench.py --rows 12500000 --table-dtype bf16 --steps 100 --warmup 16 --no-cpu-baseline
(timeout 600 python benchmarks/hbm_micro.py 2>&1 | grep '^{') > $R/hbm_micro.jsonl
(timeout 600 python benchmarks/spotify_step.py 2>&1 | grep '^{' | tail -1) > $R/spotify_step.json
# --- kernel statistics
stats() { name=$1; shift; rm -rf /tmp/st_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o x -- "$@" > /tmp/st_$name.log 2>&1
  f=$(find /tmp/st_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${name}_kernel_stats.csv; }
stats inbatch python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats triplet python bench.py --workload triplet --steps 200 --warmup 24 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats triplet_b262144 python bench.py --workload triplet --batch 262144 --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats glove python bench.py --workload glove --steps 100 --warmup 16 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats glove_b2048 python bench.py --workload glove --batch 2048 --steps 800 --warmup 32 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats retrieve_n1m_f16x2 python bench.py --workload retrieve --rows 1048576 --steps 3 --warmup 1 --no-cpu-baseline
stats retrieve_n1m_exact python bench.py --workload retrieve --rows 1048576 --precision f32 --steps 3 --warmup 1 --no-cpu-baseline
for w in inbatch triplet glove; do
  stats sharded_world1_machinery_$w env ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0 python bench.py --workload $w --steps 64 --warmup 16 --no-cpu-baseline --no-kernel-timing
done
# --- HBM traffic (PMC): separate passes per counter
for spec in "inbatch:--steps 8 --warmup 8" "triplet:--workload triplet --steps 8 --warmup 8" "triplet_b262144:--workload triplet --batch 262144 --steps 8 --warmup 8" "glove:--workload glove --steps 8 --warmup 8" "glove_b2048:--workload glove --batch 2048 --steps 8 --warmup 8" "retrieve_n1m_f16x2:--workload retrieve --rows 1048576 --steps 2 --warmup 1" "retrieve_n1m_exact:--workload retrieve --rows 1048576 --precision f32 --steps 2 --warmup 1"; do
  w=${spec%%:*}; a=${spec#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${w}_$c -o x -- python bench.py $a --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/pmc_${w}_$c.log 2>&1
  done
  python scripts/pmc_summarize.py /tmp/pmc_${w}_FETCH_SIZE /tmp/pmc_${w}_WRITE_SIZE $R/pmc_raw_$w.json | head -5
done
ls $R | wc -l; du -sh gpurun_out
