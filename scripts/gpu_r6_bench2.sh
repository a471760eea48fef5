# round 6: the driver's command (all legs), line kept under gpurun_out/
mkdir -p gpurun_out
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver_cmd.jsonl 2> gpurun_out/r6_bench_driver_cmd.err
tail -1 gpurun_out/r6_bench_driver_cmd.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['value'], d['ms_per_step'], len(json.dumps(d)))
print({k:v for k,v in r.items() if not isinstance(v,(dict,list))})
print(r.get('legs'))
"
tail -3 gpurun_out/r6_bench_driver_cmd.err
