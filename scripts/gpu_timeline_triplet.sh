export TMPDIR=/tmp
rm -rf /tmp/tlt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlt -o t -- python bench.py --workload triplet --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/tlt.log 2>&1
python3 scripts/trace_gaps.py /tmp/tlt triplet_step_kernel 100 ${COUNT:-30} | cut -c1-120
