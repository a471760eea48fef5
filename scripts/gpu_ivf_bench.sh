python -m pytest tests/test_gpu_ivf.py tests/test_gpu_retrieve.py -x -q 2>&1 | tail -3
python3 - <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
from bench_retrieve import measure_ivf
for corpus, kw in (("clustered", {}), ("hierarchical", dict(ks=(500,), nprobes=(16, 32, 64), nlist=4096))):
    r = measure_ivf(torch.device('cuda', 0), corpus=corpus, **kw)
    print(corpus, "build %.2f s, longest list %d" % (r["build_s"], r["longest_list"]))
    for leg in r["legs"]:
        print("  ", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in leg.items()})
PY
