# round 5: the one-call in-batch step (side-stream overlap) -- tests, then A/B of the headline leg alone
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_stl_loop.py -k "one_call" tests/test_gpu_kernels.py -k "one_call or inbatch_trajectory" -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -30) > gpurun_out/t_onecall.log 2>&1
tail -5 gpurun_out/t_onecall.log
for rep in 1 2; do
for mode in "1 1" "1 0" "0 0"; do
  set -- $mode
  ESR_INBATCH_ONECALL=$1 ESR_INBATCH_OVERLAP=$2 timeout 600 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('onecall=$1 overlap=$2 rep=$rep', round(d['ms_per_step'],5), round(d['value']/1e6,2), d['roofline'].get('per_kernel_us_in_run'))" | tee -a gpurun_out/inbatch_ab.log
done; done
