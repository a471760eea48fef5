mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc2/f -o x -- python scripts/prof_inbatch.py bf16x3 3 > gpurun_out/pmc2/f.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc2/h -o x -- python scripts/prof_inbatch.py bf16x3 3 > gpurun_out/pmc2/h.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d gpurun_out/pmc2/t -o x -- python scripts/prof_inbatch.py bf16x3 3 > gpurun_out/pmc2/t.log 2>&1
find gpurun_out/pmc2 -name "*.db" -delete; find gpurun_out/pmc2 -name "*kernel_trace.csv" -delete
ls gpurun_out/pmc2/*
