cd esrecsys_amd/csrc
OTHERS=$(ls build/*.o | grep -v esr_inbatch2h.o | grep -v esr_probe.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH_TIMING=3 -I../../include -c esr_inbatch2h.hip -o /tmp/ib1h_t.o 2>/tmp/cc.log || { tail -20 /tmp/cc.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/libib1h_t3.so /tmp/ib1h_t.o $OTHERS -ldl || exit 1
cd ../..
echo "1h pass Q (phases: barrier wait / S phase incl. exp / O phase)"; ESR_IB2H_BF16=force IB2H_LIB=libib1h_t3.so IB2H_ITERS=28 timeout 120 python scripts/ib2h_timing.py 2>&1 | grep -v amdgpu.ids
