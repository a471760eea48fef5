# A/B: the tree (new) against _r2tree (HEAD = round-2 kernels), alternating, same box
run() { (cd $1 && timeout 300 python bench.py "${@:2}" --no-cpu-baseline --no-secondary --no-kernel-timing 2>&1 | grep '^{' | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f M' % (d['ms_per_step'], d['value']/1e6))"); }
for rep in 1 2; do
for cfg in "--workload glove --steps 200 --warmup 20" "--workload glove --batch 2048 --steps 400 --warmup 20" "--workload triplet --steps 400 --warmup 20" "--workload triplet --batch 262144 --steps 30 --warmup 5" "--workload glove --ids zipf --steps 200 --warmup 20" "--workload triplet --ids zipf --steps 400 --warmup 20"; do
  echo "$cfg | r2: $(run _r2tree $cfg) | new: $(run . $cfg)"
done; done
