mkdir -p gpurun_out/shard
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in triplet glove; do
  ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 python $R/bench.py --workload $w --steps 400 --warmup 24 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | tail -1 > $R/gpurun_out/shard/bench_sharded_world1_machinery_$w.json
  python -c "import json; d=json.load(open('$R/gpurun_out/shard/bench_sharded_world1_machinery_$w.json')); print('$w', d['value'], d['ms_per_step'], d.get('kernels'))"
  rm -rf /tmp/tr_$w
  ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 rocprofv3 --kernel-trace -d /tmp/tr_$w -o t --output-format csv -- python $R/bench.py --workload $w --steps 400 --warmup 24 --no-cpu-baseline --no-secondary --no-steady --no-kernel-timing > /dev/null 2>&1
  python $R/scripts/trace_gaps.py /tmp/tr_$w 3000 | head -30 | tee $R/gpurun_out/shard/gaps_$w.txt
done
