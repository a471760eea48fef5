cd esrecsys_amd/csrc
for v in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DH_TIMING=$v -I../../include esr_inbatch2h.hip esr_core.hip -o ../../scripts/libib2h_t$v.so || exit 1
done
cd ../..
echo "pass Q, 1 WG/CU"; ESR_IB2H_Q_PER_CU=1 IB2H_LIB=libib2h_t1.so IB2H_ITERS=62 timeout 120 python scripts/ib2h_timing.py 2>&1 | grep -v amdgpu.ids
echo "pass Q, 2 WG/CU"; ESR_IB2H_Q_PER_CU=2 IB2H_LIB=libib2h_t1.so IB2H_ITERS=30 timeout 120 python scripts/ib2h_timing.py 2>&1 | grep -v amdgpu.ids
echo "pass C (phases: barrier wait / top of the iteration up to the first LDS wait / from there to the second; the rest of the sum is the second half)"; IB2H_LIB=libib2h_t2.so IB2H_ITERS=30 timeout 120 python scripts/ib2h_timing.py 2>&1 | grep -v amdgpu.ids
