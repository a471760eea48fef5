export TMPDIR=/tmp
rm -rf /tmp/tlr; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlr -o t -- python bench.py --workload retrieve --rows 1048576 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/tlr.log 2>&1
python3 scripts/trace_gaps.py /tmp/tlr "score_gemm_kernel<2, false>" 40 ${COUNT:-24} | cut -c1-120
