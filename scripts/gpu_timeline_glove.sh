export TMPDIR=/tmp
rm -rf /tmp/tlg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlg -o t -- python bench.py --workload glove --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/tlg.log 2>&1
python3 scripts/trace_gaps.py /tmp/tlg glove_step_resolved 30 ${COUNT:-22} | grep -v "at::native" | cut -c1-110
