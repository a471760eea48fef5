"""Per-kernel roofline fractions of the bench workloads from the committed rocprofv3 kernel stats (not only per op):
python scripts/per_kernel_roofline.py profiles/r2 > profiles/r2/per_kernel_roofline.json

MFMA kernels: executed flops of the kernel / its average duration against the 2.5 PFLOP/s dense fp16 / bf16 peak.
HBM kernels: the bytes the kernel moved (PMC: FETCH_SIZE x 2 + WRITE_SIZE, profiles/r2/pmc_raw_*.json) / its average
duration against 8 TB/s, next to its algorithmic bytes."""
import csv, json, os, sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r2"
B, D = 8192, 128
unit = 2.0 * B * B * D  # one cross-term GEMM
PEAK_MFMA, PEAK_HBM = 2500e12, 8000e9


def stats(name):
    rows = list(csv.DictReader(open(os.path.join(d, name))))
    return {r["Name"]: (float(r["AverageNs"]) * 1e-9, int(r["Calls"])) for r in rows}


def find(st, key, min_calls=50):
    for k, (t, c) in st.items():
        if key in k and c >= min_calls:
            return t
    return None


def pmc(workload, key):
    try:
        raw = json.load(open(os.path.join(d, "pmc_raw_%s.json" % workload)))
    except OSError:
        return None
    for k, v in raw.items():
        if key in k:
            return (v.get("FETCH_SIZE_KB_mean", 0) * 2 + v.get("WRITE_SIZE_KB_mean", 0)) * 1024
    return None


out = {"_note": __doc__.strip().splitlines()[0]}
st = stats("inbatch_kernel_stats.csv")
rows = []
for key, terms, what in (("inbatch2h_q_kernelILb0", 6, "pass Q: S^T and O^T, three fp16 terms each"),
                         ("inbatch2h_pct_kernel", 3, "pass C: O^T from the stored probabilities")):
    t = find(st, key)
    if t:
        traffic = pmc("inbatch", key)
        rows.append({"kernel": key, "what": what, "avg_us": t * 1e6, "executed_TFLOPs": terms * unit / t / 1e12,
                     "frac_of_2.5PF": terms * unit / t / PEAK_MFMA,
                     "hbm_bytes_pmc": traffic, "hbm_GBps_pmc": traffic / t / 1e9 if traffic else None})
out["inbatch_f16x2"] = rows
try:
    st = stats("inbatch_bf16x3_kernel_stats.csv")
    rows = []
    for key, terms in (("inbatch3_kernel<true, false, 1>", 12), ("inbatch3_pc_kernel<false>", 6)):
        t = find(st, key)
        if t:
            rows.append({"kernel": key, "avg_us": t * 1e6, "executed_TFLOPs": terms * unit / t / 1e12,
                         "frac_of_2.5PF": terms * unit / t / PEAK_MFMA})
    out["inbatch_bf16x3"] = rows
except OSError:
    pass
for workload, keys, alg in (("glove", ["glove_step_kernel", "glove_plan_kernel", "radix_scatter_kernel<11, true>",
                                       "radix_tile_kernel<11, true>"], 10292 * 65536),
                            ("triplet", ["triplet_step_kernel", "tile_sort_kernel", "tile_rank_kernel",
                                         "triplet_plan_kernel", "triplet_step_long_kernel"], 7680 * 8192)):
    st = stats("%s_kernel_stats.csv" % workload)
    rows = []
    for key in keys:
        t = find(st, key)
        traffic = pmc(workload, key)
        if t:
            rows.append({"kernel": key, "avg_us": t * 1e6, "hbm_bytes_pmc": traffic,
                         "hbm_GBps_pmc": traffic / t / 1e9 if traffic else None,
                         "frac_of_8TBps": traffic / t / PEAK_HBM if traffic else None})
    out[workload] = {"step_algorithmic_bytes": alg, "kernels": rows}
print(json.dumps(out, indent=1))
