# kernel stats of the GloVe B = 2048 loop, in-launch finalize on / off
export TMPDIR=/tmp
for f in 1 0; do
  rm -rf gpurun_out/prof/gf$f
  ESR_GLOVE_FIN_FUSED=$f timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/gf$f -o t -- python bench.py --workload glove --batch 2048 --steps 800 --warmup 32 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > gpurun_out/prof_gf$f.log 2>&1
  find gpurun_out/prof/gf$f -name "*.db" -delete; find gpurun_out/prof/gf$f -name "*kernel_trace.csv" -delete
  echo "== ESR_GLOVE_FIN_FUSED=$f"; grep '^{' gpurun_out/prof_gf$f.log | tail -1 | cut -c1-200
  python scripts/prof_stats.py gpurun_out/prof/gf$f 8
done
