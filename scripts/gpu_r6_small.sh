mkdir -p gpurun_out/small
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for leg in "glove 2048 glove_step" "triplet 8192 triplet_direct_kernel"; do
  set -- $leg
  python $R/bench.py --workload $1 --batch $2 --steps 800 --warmup 32 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"
  rm -rf /tmp/tr_$1
  rocprofv3 --kernel-trace -d /tmp/tr_$1 -o t --output-format csv -- python $R/bench.py --workload $1 --batch $2 --steps 800 --warmup 32 --no-cpu-baseline --no-secondary --no-steady --no-kernel-timing > /dev/null 2>&1
  python $R/scripts/trace_gaps.py /tmp/tr_$1 700 $3 | tee $R/gpurun_out/small/gaps_$1.txt
done
