"""Kernel durations and the gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (last `tail` kernels)."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 3:  # window: from the 100th to the `tail`-th launch of the named kernel
    occ = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]]
    rows = rows[occ[100]:occ[min(tail, len(occ) - 1)]]
else:
    rows = rows[-tail:]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("esr::", "")[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n] += e - s
    cnt[n] += 1
    if prev_end is not None:
        gap[n] += s - prev_end
    prev_end = e
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span %.1f us over %d kernels" % (span / 1e3, len(rows)))
for n in sorted(dur, key=lambda k: -dur[k]):
    print("%-42s n %5d  avg %8.2f us  gap-before avg %7.2f us" % (n, cnt[n], dur[n] / cnt[n] / 1e3, gap[n] / cnt[n] / 1e3))
