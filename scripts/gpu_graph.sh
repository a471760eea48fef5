for w in triplet inbatch glove; do for g in "" "--graph"; do (timeout 300 python bench.py --workload $w --no-cpu-baseline --no-kernel-timing $g 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:10], d['config']['launch'][:40], d['value'], d['ms_per_step'])"); done; done
