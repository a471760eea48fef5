# Round-5 measurement pass: the driver's command, single legs, rocprofv3 kernel statistics, HBM traffic and MFMA-busy
# counters (separate --pmc passes, --kernel-trace only).  Everything lands in gpurun_out/r5/; copied to profiles/r5/.
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
R=gpurun_out/r5
line() { out=$1; shift; (timeout 600 env "$@" 2>/dev/null | grep '^{' | tail -1) > $R/$out; }
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{') > $R/bench_driver_cmd_lines.jsonl
tail -1 $R/bench_driver_cmd_lines.jsonl > $R/bench_default_line.json
line bench_inbatch_200.json python bench.py --no-secondary --no-cpu-baseline
line bench_inbatch_bf16_tables.json python bench.py --table-dtype bf16 --no-secondary --no-cpu-baseline
line bench_inbatch_bf16_tables_bf16x3.json ESR_INBATCH_BF16_TABLES=bf16x3 python bench.py --table-dtype bf16 --no-secondary --no-cpu-baseline
line bench_triplet.json python bench.py --workload triplet --steps 400 --warmup 20 --no-cpu-baseline
line bench_triplet_plan_on_main_stream.json ESR_STL_PLAN_STREAM=main python bench.py --workload triplet --steps 400 --warmup 20 --no-cpu-baseline
line bench_triplet_b65536.json python bench.py --workload triplet --batch 65536 --steps 100 --warmup 16 --no-cpu-baseline
line bench_triplet_b262144.json python bench.py --workload triplet --batch 262144 --steps 64 --warmup 16 --no-cpu-baseline
line bench_glove.json python bench.py --workload glove --no-cpu-baseline
line bench_glove_b2048.json python bench.py --workload glove --batch 2048 --steps 800 --warmup 32 --no-cpu-baseline
line bench_retrieve_f16x2.json python bench.py --workload retrieve --rows 1048576 --steps 3 --warmup 1
stats() { name=$1; shift; rm -rf /tmp/st_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o x -- "$@" > /tmp/st_$name.log 2>&1
  f=$(find /tmp/st_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${name}_kernel_stats.csv; }
stats inbatch python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats inbatch_bf16_tables python bench.py --table-dtype bf16 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats triplet python bench.py --workload triplet --steps 200 --warmup 24 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
stats glove python bench.py --workload glove --steps 100 --warmup 16 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady
for spec in "inbatch:--steps 8 --warmup 8" "inbatch_bf16_tables:--table-dtype bf16 --steps 8 --warmup 8"; do
  w=${spec%%:*}; a=${spec#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${w}_$c -o x -- python bench.py $a --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/pmc_${w}_$c.log 2>&1
  done
  python scripts/pmc_summarize.py /tmp/pmc_${w}_FETCH_SIZE /tmp/pmc_${w}_WRITE_SIZE $R/pmc_raw_$w.json | head -8 | cut -c1-200
done
for spec in "inbatch:--steps 400 --warmup 100" "inbatch_bf16_tables:--table-dtype bf16 --steps 400 --warmup 100"; do
  w=${spec%%:*}; a=${spec#*:}
  rm -rf /tmp/mf_$w
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/mf_$w -o x -- python bench.py $a --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/mf_$w.log 2>&1
  python scripts/pmc_mfma_summarize.py /tmp/mf_$w $R/pmc_mfma_$w.json | head -4 | cut -c1-300
done
# roctx markers: one short marker + kernel trace of the in-batch step (the tracing row of SURVEY section 5)
rm -rf /tmp/mk; ESR_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/mk -o x -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/mk.log 2>&1
f=$(find /tmp/mk -name "*marker_api_stats.csv" -o -name "*marker*stats*.csv" | head -1); [ -n "$f" ] && cp $f $R/inbatch_marker_stats.csv
ls /tmp/mk/* | head -5; ls $R | wc -l
