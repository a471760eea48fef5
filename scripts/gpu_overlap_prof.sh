# kernel stats of the world-1 machinery triplet loop with overlapped lookups (which launches the scheme adds)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4o
for w in ${WORKLOADS:-triplet}; do
env ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0 ESR_SHARDED_OVERLAP=1 ESR_TRACE_HOST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4o/prof_$w -o p -- python $R/bench.py --workload $w --steps 200 --warmup 24 --no-cpu-baseline > $R/gpurun_out/r4o/prof_$w.log 2>&1
grep "sharded loop" $R/gpurun_out/r4o/prof_$w.log
f=$(find $R/gpurun_out/r4o/prof_$w -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:28]:
    print("%-70s calls %6s  avg %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
find $R/gpurun_out/r4o/prof_$w -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
