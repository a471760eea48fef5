# A/B of workgroups per CU for the two main kernels of the fp16 x 2 in-batch path (kernel averages from rocprofv3)
export TMPDIR=/tmp
for cfg in "1 2" "2 2" "2 1" "1 3" "2 4"; do
  set -- $cfg
  rm -rf gpurun_out/prof/ab
  ESR_IB2H_Q_PER_CU=$1 ESR_IB2H_PC_PER_CU=$2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ab -o t -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q/cu=$1 pc/cu=$2 ms_per_step %.4f' % d['ms_per_step'])"
  python scripts/prof_stats.py gpurun_out/prof/ab | grep -E "inbatch2h|rowmax2h" | cut -c1-40,100-140
done
