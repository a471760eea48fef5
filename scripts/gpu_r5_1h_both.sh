bash scripts/gpu_r5_1h.sh
bash scripts/gpu_ib1h_timing.sh 2>&1 | tail -8 | tee gpurun_out/ib1h_timing.log
