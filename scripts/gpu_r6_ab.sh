# same-box A/B of the whole in-batch train step: other builds of the library (scripts/libib2h_*.so) against this tree's
mkdir -p gpurun_out
for r in 1 2; do
for lib in ${AB_LIBS:-scripts/libib2h_R5.so esrecsys_amd/libesr_hip.so}; do
  ESR_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-kernel-timing 2>gpurun_out/r6_ab.err | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],5), round(d['value']/1e6,2))"
done
done
