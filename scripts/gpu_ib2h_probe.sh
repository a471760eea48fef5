# probe builds of the fp16 x 2 in-batch path: which part of each main kernel's time is memory traffic of which kind
export TMPDIR=/tmp
VARIANTS="${IB2H_VARIANTS:-BASE H_PROBE_PC_LINEAR H_PROBE_Q_NOSTORE}"
cd esrecsys_amd/csrc
for v in $VARIANTS; do
  flags=$(echo $v | sed 's/+/ -D/g')
  timeout 300 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -D$flags -I../../include esr_inbatch2h.hip esr_core.hip -o ../../scripts/libib2h_$v.so || exit 1
done
cd ../..
for v in $VARIANTS; do
  rm -rf gpurun_out/prof/pr
  IB2H_LIB=libib2h_$v.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/pr -o t -- python scripts/ib2h_probe.py 2>&1 | grep "op "
  python scripts/prof_stats.py gpurun_out/prof/pr | grep -E "2h|merge" | cut -c1-44,100-140
done
