export TMPDIR=/tmp
W=${W:-triplet}
rm -rf /tmp/tls; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tls -o t -- env ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=${UNIQUE:-0} python bench.py --workload $W --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing > /tmp/tls.log 2>&1
tail -3 /tmp/tls.log | cut -c1-300
python3 scripts/trace_gaps.py /tmp/tls ${KEY:-triplet_fwd} 30 ${COUNT:-60} | cut -c1-130
