cd esrecsys_amd/csrc
OTHERS=$(ls build/*.o | grep -v esr_inbatch2h.o | grep -v esr_probe.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH_TIMING=2 -I../../include -c esr_inbatch2h.hip -o /tmp/ib2h_t2.o 2>/tmp/cc.log || { tail -20 /tmp/cc.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/libib2h_t2.so /tmp/ib2h_t2.o $OTHERS -ldl || exit 1
cd ../..
IB2H_BY_BLOCK=8 IB2H_LIB=libib2h_t2.so IB2H_ITERS=30 timeout 120 python scripts/ib2h_timing.py 2>&1 | grep -v amdgpu.ids
