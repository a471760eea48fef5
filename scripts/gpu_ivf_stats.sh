export TMPDIR=/tmp
cat > /tmp/ivf_run.py <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
from bench_retrieve import measure_ivf
print(json.dumps(measure_ivf(torch.device('cuda', 0), corpus="clustered"))[:1500])
PY
rm -rf /tmp/ivfst; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ivfst -o x -- python /tmp/ivf_run.py > /tmp/ivfst.log 2>&1
tail -1 /tmp/ivfst.log | cut -c1-600
f=$(find /tmp/ivfst -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-64s calls %5s avg_us %10.1f tot_ms %8.2f pct %s'%(r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
