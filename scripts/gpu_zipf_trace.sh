mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rm -rf gpurun_out/prof/zp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/zp -o z -- python scripts/zipf_probe.py > gpurun_out/prof_zp.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof/zp/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'][:40], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'segment_' in r['Kernel_Name'] and 'sort' not in r['Kernel_Name']]
# 13 launches of (update, long) per case: print the last pair of each case
per_case = 26
for c in range(len(seq) // per_case):
    u, l = seq[c * per_case + 24], seq[c * per_case + 25]
    print("case %d: %s %.1f us | %s %.1f us" % (c, u[0], u[1], l[0], l[1]))
PY
find gpurun_out/prof/zp -name "*.db" -delete; find gpurun_out/prof/zp -name "*kernel_trace.csv" -delete
