mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rm -rf gpurun_out/prof/rtrace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/rtrace -o rt -- python benchmarks/retrieve_bench.py --reps 1 --ks ${KS:-500} --modes ${MODES:-exact} > gpurun_out/prof_rtrace.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof/rtrace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = open('gpurun_out/rtrace_summary.txt', 'w')
for r in rows:
    name = r['Kernel_Name']
    if 'esr::' not in name: continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    out.write("%-60s %10.1f us grid=%s\n" % (name[:60], dur, r.get('Grid_Size', '?')))
out.close()
PY
find gpurun_out/prof/rtrace -name "*.db" -delete; find gpurun_out/prof/rtrace -name "*kernel_trace.csv" -delete
tail -30 gpurun_out/rtrace_summary.txt
