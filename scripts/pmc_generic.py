"""Mean per-launch value of every counter of a rocprofv3 --pmc run, per kernel (substring filter).
usage: pmc_generic.py DIR FILTER"""
import csv, glob, sys, collections
d, flt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        k = k[:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(disp[k])
    print(k, "launches", n, {c: round(v / n, 1) for c, v in sorted(acc[k].items())})
