mkdir -p gpurun_out/ivf
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ivf/ivf_bench.jsonl | cut -c1-1500
import json, torch, sys
sys.path.insert(0, '.')
from bench_retrieve import measure_ivf
dev = torch.device('cuda', 0)
for corpus in ("clustered", "iid"):
    print(json.dumps(measure_ivf(dev, corpus=corpus)), flush=True)
PY
