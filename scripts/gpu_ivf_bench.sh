python -m pytest tests/test_gpu_ivf.py -x -q -s 2>&1 | tail -4
python3 - <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
from bench_retrieve import measure_ivf
r = measure_ivf(torch.device('cuda', 0), corpus="clustered")
for leg in r["legs"]:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in leg.items()})
PY
