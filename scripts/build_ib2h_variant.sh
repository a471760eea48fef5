# probe build of the fp16 x 2 in-batch kernels: esr_inbatch2h.hip with -D<flag> ("+"-separated), every other unit from the
# regular build (python -m esrecsys_amd.build first) -> scripts/libib2h_<variant>.so.  bash scripts/build_ib2h_variant.sh V...
for v in "$@"; do
  flags=$(echo $v | sed 's/+/ -D/g')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -D$flags -Iinclude -c esrecsys_amd/csrc/esr_inbatch2h.hip -o /tmp/ib2h_$v.o || exit 1
  objs=$(ls esrecsys_amd/csrc/build/*.o | grep -v "esr_inbatch2h.o\|esr_probe.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/libib2h_$v.so /tmp/ib2h_$v.o $objs -ldl && echo built scripts/libib2h_$v.so
done
