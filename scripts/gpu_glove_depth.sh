run() { (cd $1 && shift && env "$@" timeout 300 python bench.py --workload glove --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-kernel-timing 2>&1 | grep '^{' | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'])"); }
for rep in 1 2; do
echo "r2 depth2: $(run _r2tree A=1)  new depth1: $(run . ESR_GLOVE_PRESORT_DEPTH=1)  new depth2: $(run . ESR_GLOVE_PRESORT_DEPTH=2)  new depth3: $(run . ESR_GLOVE_PRESORT_DEPTH=3)  r2 depth3: $(run _r2tree ESR_GLOVE_PRESORT_DEPTH=3)"
done
