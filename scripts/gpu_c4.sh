# config-4 shape on one GPU: this rank's 12.5 M-row share of a 100 M x 128 bf16 table pair (+ fp32 accumulators)
(ESR_BENCH_SHARDED=1 timeout 600 python bench.py --rows 12500000 --table-dtype bf16 --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-900)
(timeout 600 python bench.py --table-dtype bf16 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500)
(ESR_BENCH_SHARDED=1 timeout 600 python bench.py --workload triplet --table-dtype bf16 --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400)
