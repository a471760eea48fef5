export TMPDIR=/tmp
rm -rf /tmp/tzs; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tzs -o x -- python bench.py --workload triplet --ids zipf --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/tzs.log 2>&1
f=$(find /tmp/tzs -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('%-64s calls %5s avg_us %10.1f pct %s'%(r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
