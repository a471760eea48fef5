# Phase stamps of inbatch3_kernel: debug builds with -DESR_IB3_TIMING, transposing-read variant and the
# transposed-image variant (-DESR_IB3_USE_TR=0) side by side
(cd esrecsys_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DESR_IB3_TIMING -I../../include esr_inbatch3.hip esr_core.hip -o ../../scripts/libib3dbg.so) && echo "bf16 tables (one-plane kernels)" && IB3_BF16=1 python scripts/ib3_timing.py 2>&1 | grep -v amdgpu.ids
for tr in 1 0; do
  (cd esrecsys_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DESR_IB3_TIMING -DESR_IB3_USE_TR=$tr -I../../include esr_inbatch3.hip esr_core.hip -o ../../scripts/libib3dbg.so) && echo "USE_TR=$tr" && python scripts/ib3_timing.py 2>&1 | grep -v amdgpu.ids
done
