for B in 65536 262144 2048; do
for mode in main side; do
  ESR_STL_PLAN_STREAM=$mode timeout 600 python bench.py --workload triplet --batch $B --steps 120 --warmup 24 --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('triplet B=$B plan stream=$mode', round(d['ms_per_step'],5), round(d['value']/1e6,2))"
done; done
ESR_STL_LOOP=presorted ESR_STL_PLAN_STREAM=main python bench.py --workload triplet --steps 800 --warmup 40 --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('reference loop shape main', round(d['ms_per_step'],5), round(d['value']/1e6,2))"
ESR_STL_LOOP=presorted ESR_STL_PLAN_STREAM=side python bench.py --workload triplet --steps 800 --warmup 40 --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('reference loop shape side', round(d['ms_per_step'],5), round(d['value']/1e6,2))"
