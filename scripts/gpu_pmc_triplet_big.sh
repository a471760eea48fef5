export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_tb_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_tb_$c -o x -- python bench.py --workload triplet --batch 262144 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/pmc_tb_$c.log 2>&1
done
mkdir -p gpurun_out/r3
python scripts/pmc_summarize.py /tmp/pmc_tb_FETCH_SIZE /tmp/pmc_tb_WRITE_SIZE gpurun_out/r3/pmc_raw_triplet_b262144.json | head -8
