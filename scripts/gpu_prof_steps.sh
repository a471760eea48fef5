# kernel statistics of the HBM-bound legs (rocprofv3 --kernel-trace --stats), summaries under gpurun_out/prof_steps/
mkdir -p gpurun_out/prof_steps
export TMPDIR=/tmp
prof() {  # name, bench args
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- python bench.py "$@" --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > gpurun_out/prof_steps/$name.log 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/prof_steps/${name}_kernel_stats.csv
  echo "== $name: $(grep -v amdgpu.ids gpurun_out/prof_steps/$name.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  python3 scripts/prof_stats.py /tmp/prof_$name ${ROWS:-10}
}
prof triplet --workload triplet --steps 200 --warmup 20
prof glove --workload glove --steps 100 --warmup 10
prof glove2048 --workload glove --batch 2048 --steps 400 --warmup 20
prof triplet262144 --workload triplet --batch 262144 --steps 20 --warmup 3
