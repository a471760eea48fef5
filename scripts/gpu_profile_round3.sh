# Round-3 measurement pass: bench lines, kernel statistics (rocprofv3 --kernel-trace --stats) and HBM traffic (separate
# --pmc FETCH_SIZE / WRITE_SIZE passes).  Everything lands in gpurun_out/r3/; the summaries are copied to profiles/r3/.
mkdir -p gpurun_out/r3 gpurun_out/r3/pmc
export TMPDIR=/tmp
R=gpurun_out/r3
line() { out=$1; shift; (timeout 600 env "$@" 2>&1 | grep '^{' | tail -1) > $R/$out; }
# --- bench lines
line bench_default_line.json python bench.py --gpus 1 --steps 20 --warmup 5
line bench_inbatch_200.json python bench.py --no-secondary --no-cpu-baseline
line bench_inbatch_f32.json python bench.py --precision f32 --no-secondary --no-cpu-baseline
line bench_inbatch_bf16x3.json python bench.py --precision bf16x3 --no-secondary --no-cpu-baseline
line bench_inbatch_bf16_tables.json python bench.py --table-dtype bf16 --no-secondary --no-cpu-baseline
for w in inbatch triplet glove; do line bench_${w}_zipf.json python bench.py --workload $w --ids zipf --no-secondary --no-cpu-baseline; done
line bench_triplet.json python bench.py --workload triplet --steps 400 --warmup 20 --no-cpu-baseline
line bench_glove.json python bench.py --workload glove --no-cpu-baseline
line bench_glove_b2048.json python bench.py --workload glove --batch 2048 --steps 400 --warmup 20 --no-cpu-baseline
line bench_retrieve.json python bench.py --workload retrieve
for w in inbatch triplet glove; do
  line bench_sharded_world1_$w.json ESR_BENCH_SHARDED=1 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline
  line bench_sharded_world1_machinery_$w.json ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline
  line bench_sharded_world1_machinery_unique_$w.json ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=1 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline
done
for w in inbatch triplet; do line bench_replicated_world1_$w.json ESR_BENCH_SHARDED=1 ESR_BENCH_PARALLELISM=replicated python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline; done
line bench_sharded_world1_config4_share.json ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 python bench.py --rows 12500000 --table-dtype bf16 --steps 100 --warmup 10 --no-cpu-baseline
(timeout 600 python benchmarks/hbm_micro.py 2>&1 | grep '^{') > $R/hbm_micro.jsonl
(timeout 300 python benchmarks/mfma_peak.py 2>&1 | grep '^{') > $R/mfma_peak.jsonl
(timeout 600 python benchmarks/spotify_step.py 2>&1 | grep '^{' | tail -1) > $R/spotify_step.json
python3 - > $R/ivf_bench.jsonl 2>/dev/null <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
from bench_retrieve import measure_ivf
for corpus in ("clustered", "iid"):
    print(json.dumps(measure_ivf(torch.device('cuda', 0), corpus=corpus)), flush=True)
PY
# --- kernel statistics
stats() { name=$1; shift; rm -rf /tmp/st_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o x -- "$@" > /tmp/st_$name.log 2>&1
  f=$(find /tmp/st_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${name}_kernel_stats.csv; }
stats inbatch python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats inbatch_f32 python bench.py --precision f32 --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats triplet python bench.py --workload triplet --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats triplet_b262144 python bench.py --workload triplet --batch 262144 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats glove python bench.py --workload glove --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats glove_b2048 python bench.py --workload glove --batch 2048 --steps 400 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats retrieve_n1m python bench.py --workload retrieve --rows 1048576 --steps 3 --warmup 1 --no-cpu-baseline
stats spotify python benchmarks/spotify_step.py
stats sharded_world1_machinery_inbatch env ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing
# --- HBM traffic (PMC): separate passes per counter, --kernel-trace only
for spec in "inbatch:--steps 6 --warmup 2" "triplet:--workload triplet --steps 6 --warmup 2" "glove:--workload glove --steps 6 --warmup 2" "retrieve_n1m:--workload retrieve --rows 1048576 --steps 2 --warmup 1"; do
  w=${spec%%:*}; a=${spec#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${w}_$c -o x -- python bench.py $a --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/pmc_${w}_$c.log 2>&1
  done
  python scripts/pmc_summarize.py /tmp/pmc_${w}_FETCH_SIZE /tmp/pmc_${w}_WRITE_SIZE $R/pmc_raw_$w.json | head -6
done
ls $R | head -80; du -sh gpurun_out
