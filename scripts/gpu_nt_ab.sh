# A/B of a library variant against the default build: VARIANT=nt0 bash scripts/gpu_nt_ab.sh  (alternating runs, one box)
mkdir -p gpurun_out
V=${VARIANT:-nt0}
cp esrecsys_amd/libesr_hip.so /tmp/libesr_hip_default.so
run() { timeout 600 python bench.py "$@" --no-secondary --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   %-46s %.5f ms  %8.2f M' % (d['config']['workload'][:46], d['ms_per_step'], d['value']/1e6))"; }
for rep in 1 2; do
for lib in default $V; do
  [ $lib = default ] && cp /tmp/libesr_hip_default.so esrecsys_amd/libesr_hip.so || cp esrecsys_amd/libesr_hip_$lib.so esrecsys_amd/libesr_hip.so
  echo "== $lib (rep $rep)"
  run --workload triplet --steps 400 --warmup 20
  run --workload glove --batch 2048 --steps 800 --warmup 32
  run --steps 200 --warmup 20
  run --workload glove --steps 100 --warmup 16
  run --workload triplet --batch 262144 --steps 64 --warmup 16
done; done 2>&1 | tee gpurun_out/ab_$V.log
cp /tmp/libesr_hip_default.so esrecsys_amd/libesr_hip.so
