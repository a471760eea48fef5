mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rm -rf gpurun_out/prof/shard
ESR_BENCH_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/shard -o s -- python bench.py --workload ${W:-inbatch} --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_shard.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof/shard/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in list(csv.DictReader(open(f)))[:16]:
    print(r['Name'][:78].ljust(78), r['Calls'], r['AverageNs'], r['Percentage'])
t = glob.glob('gpurun_out/prof/shard/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
# one steady-state step: find the 30th inbatch3_kernel<true> occurrences
idx = [i for i, r in enumerate(rows) if 'inbatch3_rowmax' in r['Kernel_Name']]
a, b = idx[30], idx[31]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    print("%8.1f us +%7.1f  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:70]))
PY
find gpurun_out/prof/shard -name "*.db" -delete; find gpurun_out/prof/shard -name "*kernel_trace.csv" -delete
