# probe builds of the one-plane in-batch kernel: what does an iteration spend its time on?
# (build here or on the box: `bash scripts/gpu_ib1h_probe.sh build`, then on the GPU `bash scripts/gpu_ib1h_probe.sh run`)
export TMPDIR=/tmp
VARIANTS="${IB1H_VARIANTS:-BASE H1_PROBE_NO_DMA H1_PROBE_NO_TR H1_PROBE_NO_EXP H1_PROBE_NO_SLD H1_PROBE_NO_DMA+H1_PROBE_NO_TR+H1_PROBE_NO_SLD H1_PROBE_NO_DMA+H1_PROBE_NO_TR+H1_PROBE_NO_SLD+H1_PROBE_NO_EXP}"
if [ "$1" != "run" ]; then
  cd esrecsys_amd/csrc
  OTHERS=$(ls build/*.o | grep -v esr_inbatch2h.o | grep -v esr_probe.o)
  for v in $VARIANTS; do
    flags=$(echo $v | sed 's/+/ -D/g')
    timeout 300 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -D$flags -I../../include -c esr_inbatch2h.hip -o /tmp/ib1h_$v.o 2>/tmp/cc_$v.log || { tail -5 /tmp/cc_$v.log; exit 1; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/libib1h_$v.so /tmp/ib1h_$v.o $OTHERS -ldl 2>/tmp/ld_$v.log || { tail -5 /tmp/ld_$v.log; exit 1; }
  done
  cd ../..
fi
[ "$1" = "build" ] && exit 0
mkdir -p gpurun_out
for v in $VARIANTS; do
  rm -rf gpurun_out/prof/pr
  env ${IB1H_ENV-ESR_IB2H_BF16=force} IB2H_LIB=libib1h_$v.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/pr -o t -- python scripts/ib2h_probe.py 2>&1 | grep "op "
  echo "== $v"; python scripts/prof_stats.py gpurun_out/prof/pr | grep -E "${IB1H_GREP:-1h}" | cut -c1-44,100-140
done 2>&1 | tee gpurun_out/ib1h_probe.log
