# round 4: the direct (by-triplet, in-place) triplet step against the stamped walk over sorted occurrences
python -m pytest tests/test_gpu_triplet_step.py tests/test_gpu_stl_loop.py tests/test_gpu_api.py tests/test_gpu_sharded.py -q 2>&1 | tail -6
for m in "ESR_TRIPLET_STEP=direct" "ESR_TRIPLET_STEP=direct ESR_TRIPLET_DIRECT_LANES=few" "ESR_TRIPLET_STEP=stamped"; do for bsz in 8192 65536 262144; do
  st=400; [ $bsz -gt 10000 ] && st=100; [ $bsz -gt 100000 ] && st=48
  echo "== $m B=$bsz: $(env $m python bench.py --workload triplet --batch $bsz --steps $st --warmup 24 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], "step frac", r.get("step",{}).get("frac"), r.get("per_kernel_us_in_run"))')"
done; done
for m in direct stamped; do echo "== zipf ESR_TRIPLET_STEP=$m: $(ESR_TRIPLET_STEP=$m python bench.py --workload triplet --ids zipf --steps 200 --warmup 24 --no-cpu-baseline --no-secondary --no-steady --no-kernel-timing 2>/dev/null | grep '^{' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"; done
