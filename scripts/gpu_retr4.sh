bash scripts/gpu_retr.sh
(timeout 600 python benchmarks/retrieve_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/retrieve_bench.jsonl
cut -c1-200 gpurun_out/retrieve_bench.jsonl
bash scripts/gpu_gemm_timing.sh
