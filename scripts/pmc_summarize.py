"""Per-kernel mean of FETCH_SIZE / WRITE_SIZE (KB) from two rocprofv3 --pmc passes -> JSON.
usage: python scripts/pmc_summarize.py <fetch_dir> <write_dir> <out.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].strip()
            acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    res = {}
    for counter, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        for name, vals in collect(d, counter).items():
            if "esr" not in name:
                continue
            e = res.setdefault(name, {})
            e[counter + "_KB_mean"] = sum(vals) / len(vals)
            e["launches"] = len(vals)
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KB_mean", 0))[:12]:
        print(k[:60].ljust(60), v)


if __name__ == "__main__":
    main()
