# LDS bank conflicts of every kernel of a bench leg (one --pmc pass): PMC_ARGS = the leg's bench.py arguments
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/plds
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/plds -o x -- python bench.py $PMC_ARGS --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/plds.log 2>&1
python scripts/pmc_generic.py /tmp/plds "" | grep -v "at::native" | cut -c1-330
