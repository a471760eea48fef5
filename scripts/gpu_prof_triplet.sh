mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rm -rf gpurun_out/prof/trip
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trip -o t -- python bench.py --workload ${W:-triplet} ${EXTRA:-} --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_trip.log 2>&1
find gpurun_out/prof/trip -name "*.db" -delete; find gpurun_out/prof/trip -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof/trip/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:70].ljust(70), r['Calls'], r['AverageNs'], r['Percentage'])
PY
