"""Where the sparse Adagrad time goes under skewed ids: time ops.sparse_adagrad for GloVe-sized inputs with
(a) uniform ids, (b) Zipf ids, (c) Zipf with every id's multiplicity capped."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esrecsys_amd import ops
dev = torch.device("cuda", 0)
V, D, n = 465_537, 256, 131_072
rng = np.random.default_rng(0)
table = torch.randn((V, D), device=dev) * 0.06
accum = torch.full((V, D), 0.1, device=dev)
rows = torch.randn((n, D), device=dev) * 0.01
w = 1.0 / np.arange(1, V + 1); cdf = np.cumsum(w / w.sum())
zipf = rng.permutation(V)[np.searchsorted(cdf, rng.random(n)).clip(max=V - 1)].astype(np.int32)
def capped(ids, cap):
    out = ids.copy(); seen = {}
    fresh = iter(np.setdiff1d(np.arange(V, dtype=np.int32), ids)[:n])
    for i, v in enumerate(ids):
        c = seen.get(v, 0)
        if c >= cap: out[i] = next(fresh)
        else: seen[v] = c + 1
    return out
cases = {"uniform": rng.integers(0, V, n).astype(np.int32), "zipf": zipf, "zipf cap 31": capped(zipf, 31),
         "zipf cap 63": capped(zipf, 63), "zipf cap 127": capped(zipf, 127), "zipf cap 500": capped(zipf, 500),
         "zipf cap 2000": capped(zipf, 2000)}
for name, ids in cases.items():
    t = torch.from_numpy(ids).to(dev)
    sid, perm = ops.segment_sort(t, V)
    u, c = np.unique(ids, return_counts=True)
    for _ in range(3): ops.sparse_adagrad(table, accum, sid, perm, rows.clone(), 0.01)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rr = [rows.clone() for _ in range(10)]
    e0.record()
    for r in rr: ops.sparse_adagrad(table, accum, sid, perm, r, 0.01)
    e1.record(); torch.cuda.synchronize()
    print("%-12s unique %6d  max run %5d  runs>63 %4d  : %.1f us" % (name, len(u), c.max(), (c > 63).sum(), e0.elapsed_time(e1) / 10 * 1e3))
