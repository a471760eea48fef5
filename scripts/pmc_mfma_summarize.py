"""Per-kernel MFMA-pipe busy cycles, GPU-active cycles and launch durations from ONE rocprofv3 pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE [...] --kernel-trace) -> JSON: the matrix pipes' measured utilisation and
the clock the kernel actually ran at.  usage: python scripts/pmc_mfma_summarize.py <dir> <out.json>

SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the device (32 cycles per v_mfma_f32_32x32x16 wave instruction);
GRBM_GUI_ACTIVE over the XCDs.  utilisation = MFMA busy / (SIMDs x active cycles per XCD); clock = active cycles per XCD /
launch duration.  The instance count GRBM_GUI_ACTIVE was summed over is not in the CSV: both readings are written
(`xcd_sum` = 8 instances, `single` = 1) and the one whose clock lands inside the part's 0.5 - 2.4 GHz range is marked."""
import csv
import glob
import json
import sys
from collections import defaultdict

SIMDS, XCDS = 1024, 8


def main():
    d, out = sys.argv[1:3]
    per = defaultdict(lambda: defaultdict(dict))  # kernel -> dispatch -> counter -> value
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].strip()
            if "esr" not in name:
                continue
            rec = per[name][r["Dispatch_Id"]]
            rec[r["Counter_Name"]] = rec.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                rec["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    res = {}
    for name, disp in per.items():
        n = len(disp)
        mean = defaultdict(float)
        for did, rec in disp.items():
            for k, v in rec.items():
                mean[k] += v / n
            if did in dur:
                mean["_ns_trace"] += dur[did] / n
        ns = mean.get("_ns_trace") or mean.get("_ns") or 0.0
        e = {"launches": n, "duration_us_profiled": ns / 1e3}
        for k, v in mean.items():
            if not k.startswith("_"):
                e[k] = v
        busy, act = mean.get("SQ_VALU_MFMA_BUSY_CYCLES"), mean.get("GRBM_GUI_ACTIVE")
        if busy is not None and act and ns:
            for label, inst in (("xcd_sum", XCDS), ("single", 1)):
                cyc = act / inst
                ghz = cyc / ns
                e[label] = {"active_cycles_per_xcd": cyc, "clock_GHz": ghz, "mfma_pipe_utilisation": busy / (SIMDS * cyc),
                            "plausible": 0.5 <= ghz <= 2.45}
        res[name] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:10]:
        print(k[:48].ljust(48), {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a not in ("xcd_sum", "single")},
              {lab: {a: round(b, 3) if isinstance(b, float) else b for a, b in v[lab].items()} for lab in ("xcd_sum", "single") if lab in v})


if __name__ == "__main__":
    main()
