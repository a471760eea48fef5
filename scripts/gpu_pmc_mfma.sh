# MFMA-pipe utilisation and effective clock of the MFMA-bound kernels, measured by counters (one --pmc pass per workload,
# --kernel-trace only; no --stats / sys-trace beside --pmc).  Long runs: the power manager needs ~10 ms to settle.
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for spec in "inbatch:--steps 400 --warmup 100" "inbatch_f32:--precision f32 --steps 200 --warmup 50" "inbatch_bf16x3:--precision bf16x3 --steps 300 --warmup 100" "retrieve_n1m_f16x2:--workload retrieve --rows 1048576 --steps 4 --warmup 2" "retrieve_n1m_exact:--workload retrieve --rows 1048576 --precision f32 --steps 4 --warmup 2"; do
  w=${spec%%:*}; a=${spec#*:}
  rm -rf /tmp/mf_$w
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/mf_$w -o x -- python bench.py $a --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/mf_$w.log 2>&1
  echo "== $w rc=$? $(grep '^{' /tmp/mf_$w.log | tail -1 | cut -c1-160)"
  python scripts/pmc_mfma_summarize.py /tmp/mf_$w gpurun_out/r4/pmc_mfma_$w.json | head -4 | cut -c1-420
done
