export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU"; do
  i=$((i+1)); rm -rf /tmp/p1h_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p1h_$i -o x -- python bench.py ${PMC_ARGS:---table-dtype bf16} --steps 8 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/p1h_$i.log 2>&1
  python scripts/pmc_generic.py /tmp/p1h_$i "${PMC_FILTER:-inbatch1h}" | cut -c1-400
done 2>&1 | tee gpurun_out/pmc_1h.log
