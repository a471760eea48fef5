# LDS bank conflicts / LDS-busy cycles of the in-batch kernels (one --pmc pass, kernel trace only)
mkdir -p gpurun_out/ldspmc
export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/ldspmc -o x -- python bench.py --workload inbatch --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/ldspmc/run.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ldspmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    if "inbatch3" in k:
        print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
find gpurun_out/ldspmc -name "*.db" -delete
