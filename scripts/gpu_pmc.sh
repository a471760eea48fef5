mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
for p in bf16x3 f32; do
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc/a_$p -o x -- python scripts/prof_inbatch.py $p 4 > gpurun_out/pmc/a_$p.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc/b_$p -o x -- python scripts/prof_inbatch.py $p 4 > gpurun_out/pmc/b_$p.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc/c_$p -o x -- python scripts/prof_inbatch.py $p 4 > gpurun_out/pmc/c_$p.log 2>&1
done
find gpurun_out/pmc -name "*.db" -delete; find gpurun_out/pmc -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/pmc/*.log
