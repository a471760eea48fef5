"""Timeline of a rocprofv3 --kernel-trace run: python scripts/trace_gaps.py <dir> <kernel substring> [skip] [count]
-- from the `skip`-th launch of the first kernel whose name contains the substring, `count` kernels: start (us,
relative), duration, idle gap since the end of the latest kernel before it, queue, name."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 60
count = int(sys.argv[4]) if len(sys.argv) > 4 else 40
hits = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
first = hits[min(skip, len(hits) - 1)]
t0 = int(rows[first]["Start_Timestamp"])
prev_end = None
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f  dur %7.1f  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
    prev_end = max(prev_end or 0, e)
