# A/B: host-side wait for the side-stream sort (default) against the stream wait (ESR_GLOVE_HOST_WAIT_US=0), C3 B = 65 536
for i in 1 2; do
for w in 2000 0; do
  echo "host_wait_us=$w: $(ESR_GLOVE_HOST_WAIT_US=$w python bench.py --workload glove --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])')"
done; done
bash scripts/gpu_timeline_glove.sh
