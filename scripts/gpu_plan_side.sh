# sort + plan of the coming group on the second stream (ESR_STL_PLAN_STREAM=side) against the main stream, triplet legs
for B in 262144 65536 8192; do
  for mode in main side; do
    steps=64; [ $B = 8192 ] && steps=400
    echo "B=$B plan stream $mode: $(ESR_STL_PLAN_STREAM=$mode timeout 600 python bench.py --workload triplet --batch $B --steps $steps --warmup 16 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["value"]/1e6,1), d["roofline"].get("frac"))')"
  done
done
