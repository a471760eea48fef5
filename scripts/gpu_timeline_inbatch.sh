export TMPDIR=/tmp
rm -rf /tmp/tli; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tli -o t -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/tli.log 2>&1
python3 scripts/trace_gaps.py /tmp/tli prep2h 40 ${COUNT:-34} | cut -c1-120
