# Board power and clocks while the headline bench loops (evidence for "the kernels run at the power limit"):
# rocm-smi is sampled every 0.5 s beside a 3000-step run.
export TMPDIR=/tmp
(timeout 200 python bench.py --steps 60000 --warmup 50 --no-cpu-baseline --no-secondary --no-kernel-timing > gpurun_out/power_bench.log 2>&1) &
BP=$!
sleep 9
for i in $(seq 1 24); do
  timeout 10 rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|busy" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
tail -1 gpurun_out/power_bench.log | cut -c1-160
timeout 10 rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s ' ' | tr '\n' ';'; echo " (idle)"
timeout 10 rocm-smi --showmaxpower 2>/dev/null | grep -i power | tr -s ' '
