mkdir -p gpurun_out/trace
export TMPDIR=/tmp
tr() { name=$1; key=$2; shift; shift; rm -rf /tmp/tr_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$name -o t -- python bench.py "$@" --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > gpurun_out/trace/$name.log 2>&1
  echo "== $name $(grep -v amdgpu.ids gpurun_out/trace/$name.log | grep '^{' | tail -1 | cut -c1-120)"
  python3 scripts/trace_gaps.py /tmp/tr_$name $key ${SKIP:-60} ${COUNT:-40} | tee gpurun_out/trace/$name.txt
  python3 scripts/prof_stats.py /tmp/tr_$name 9
}
SKIP=30 tr glove glove_step_kernel --workload glove --steps 60 --warmup 10
ESR_GLOVE_PRESORT=0 SKIP=30 tr glove_inline glove_step_kernel --workload glove --steps 60 --warmup 10
