export TMPDIR=/tmp
mkdir -p gpurun_out/prof_sharded
for w in ${WL:-glove triplet inbatch}; do
rm -rf /tmp/ps_$w
ESR_BENCH_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/ps_$w -o t -- python bench.py --workload $w --steps 60 --warmup 10 --no-kernel-timing > gpurun_out/prof_sharded/$w.log 2>&1
echo "== $w $(grep '^{' gpurun_out/prof_sharded/$w.log | tail -1 | cut -c1-140)"
python3 scripts/prof_stats.py /tmp/ps_$w 14
f=$(find /tmp/ps_$w -name "*memory_copy_stats.csv" | head -1); [ -n "$f" ] && cat $f | head -5
cp $(find /tmp/ps_$w -name "*kernel_stats.csv" | head -1) gpurun_out/prof_sharded/${w}_kernel_stats.csv
done
