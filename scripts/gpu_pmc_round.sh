# HBM traffic of the dominant kernels of every workload: separate --pmc FETCH_SIZE / WRITE_SIZE passes
mkdir -p gpurun_out/pmcr
export TMPDIR=/tmp
for w in inbatch triplet glove retrieve; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcr/${w}_$c -o x -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/pmcr/${w}_$c.log 2>&1
  done
  python scripts/pmc_summarize.py gpurun_out/pmcr/${w}_FETCH_SIZE gpurun_out/pmcr/${w}_WRITE_SIZE gpurun_out/pmc_raw_$w.json | head -8
done
find gpurun_out/pmcr -name "*.db" -delete; find gpurun_out/pmcr -name "*kernel_trace.csv" -delete; find gpurun_out/pmcr -name "*counter_collection.csv" -delete
