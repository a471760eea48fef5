# Round-6 measurement pass: the driver's command, rocprofv3 kernel statistics of the changed legs, HBM traffic and MFMA-busy
# counters (separate --pmc passes, --kernel-trace only).  Everything lands in gpurun_out/r6/; copied to profiles/r6/.
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
R=gpurun_out/r6
line() { out=$1; shift; (timeout 600 env "$@" 2>/dev/null | grep '^{' | tail -1) > $R/$out; }
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{') > $R/bench_driver_cmd_lines.jsonl
tail -1 $R/bench_driver_cmd_lines.jsonl > $R/bench_default_line.json
line bench_inbatch_200.json python bench.py --no-secondary --no-cpu-baseline
line bench_triplet_b262144_bf16.json python bench.py --workload triplet --batch 262144 --table-dtype bf16 --steps 64 --warmup 16 --no-cpu-baseline
line bench_glove_bf16.json python bench.py --workload glove --table-dtype bf16 --no-cpu-baseline
line bench_retrieve_f16r.json ESR_BENCH_RETRIEVE_MODE=f16r python bench.py --workload retrieve --rows 1048576 --steps 3 --warmup 1
stats() { name=$1; shift; rm -rf /tmp/st_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o x -- "$@" > /tmp/st_$name.log 2>&1
  f=$(find /tmp/st_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${name}_kernel_stats.csv; }
stats inbatch python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats triplet_b262144_bf16 python bench.py --workload triplet --batch 262144 --table-dtype bf16 --steps 64 --warmup 16 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats glove_bf16 python bench.py --workload glove --table-dtype bf16 --steps 100 --warmup 16 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats retrieve_f16r python scripts/retr_ktime.py f16r
for spec in "inbatch:--steps 8 --warmup 8" "triplet_b262144_bf16:--workload triplet --batch 262144 --table-dtype bf16 --steps 6 --warmup 4" "glove_bf16:--workload glove --table-dtype bf16 --steps 8 --warmup 8"; do
  w=${spec%%:*}; a=${spec#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${w}_$c -o x -- python bench.py $a --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/pmc_${w}_$c.log 2>&1
  done
  python scripts/pmc_summarize.py /tmp/pmc_${w}_FETCH_SIZE /tmp/pmc_${w}_WRITE_SIZE $R/pmc_raw_$w.json | head -10 | cut -c1-220
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_retr_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_retr_$c -o x -- python scripts/retr_ktime.py f16r > /tmp/pmc_retr_$c.log 2>&1
done
python scripts/pmc_summarize.py /tmp/pmc_retr_FETCH_SIZE /tmp/pmc_retr_WRITE_SIZE $R/pmc_raw_retrieve_f16r.json | head -8 | cut -c1-220
rm -rf /tmp/mf_inbatch
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/mf_inbatch -o x -- python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/mf_inbatch.log 2>&1
python scripts/pmc_mfma_summarize.py /tmp/mf_inbatch $R/pmc_mfma_inbatch.json | head -6 | cut -c1-300
ls $R
