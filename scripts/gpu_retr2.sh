bash scripts/gpu_retr.sh
bash scripts/gpu_retr_bench.sh
