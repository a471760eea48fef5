for v in 1 0 1 0; do (ESR_INBATCH_PRESORT=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('presort=$v', d['value'], d['ms_per_step'])"); done
