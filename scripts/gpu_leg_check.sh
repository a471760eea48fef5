for i in 1 2 3; do
python bench.py --workload triplet --batch 262144 --steps 64 --warmup 16 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("triplet262144 64 steps:", d["ms_per_step"], d["value"])'
python bench.py --workload triplet --batch 262144 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("triplet262144 20 steps:", d["ms_per_step"], d["value"])'
done
python bench.py --workload glove --steps 200 --warmup 10 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("glove 200:", d["ms_per_step"], d["value"])'
