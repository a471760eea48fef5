# world-1 sharded legs: the default (direct single-GPU steps), the exchange machinery per occurrence, and per distinct row
mkdir -p gpurun_out/sharded_w1
for mode in direct machinery unique; do
for w in inbatch triplet glove; do
  case $mode in direct) envs="";; machinery) envs="ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0";; unique) envs="ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=1";; esac
  env $envs ESR_BENCH_SHARDED=1 timeout 300 python bench.py --workload $w --steps ${STEPS:-200} --warmup 20 --no-kernel-timing --no-cpu-baseline 2>&1 | grep "^{" | tail -1 > gpurun_out/sharded_w1/${mode}_$w.json
  python3 -c "
import json; d=json.load(open('gpurun_out/sharded_w1/${mode}_$w.json')); print('$mode', '$w', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],4), 'ms |', d['config']['parallelism'][:70])"
done; done
for w in inbatch triplet; do ESR_BENCH_PARALLELISM=replicated ESR_BENCH_SHARDED=1 timeout 300 python bench.py --workload $w --steps ${STEPS:-200} --warmup 20 --no-kernel-timing --no-cpu-baseline 2>&1 | grep "^{" | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('replicated $w', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],4), 'ms |', d['config']['parallelism'][:60])"; done
