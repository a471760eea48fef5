export TMPDIR=/tmp
rm -rf /tmp/tlsp; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tlsp -o t -- python benchmarks/spotify_step.py > /tmp/tlsp.log 2>&1
tail -2 /tmp/tlsp.log | cut -c1-400
python3 scripts/trace_gaps.py /tmp/tlsp momentum_catchup 200 ${COUNT:-30} | cut -c1-130
ls /tmp/tlsp/*/ 2>/dev/null | head
