"""Randomised check of the IVF search: whatever it probes, the answer must be the exact top-k of the probed lists
(numpy), with -inf / -1 where the lists hold fewer than k candidates.  SEED, CASES."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esrecsys_amd import ops
from esrecsys_amd.ivf import IVFIndex
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N_ = int(os.environ.get("CASES", "30"))
bad = 0
for case in range(N_):
    N = int(rng.choice([200, 3000, 20000, 70000]))
    D = int(rng.choice([4, 32, 64, 100, 128, 512]))
    nlist = int(rng.choice([1, 4, 16, 64, 128]))
    nlist = min(nlist, N // 8)
    nprobe = int(rng.integers(1, nlist + 1))
    k = int(rng.choice([1, 10, 100, 500, 1024]))
    nq = int(rng.choice([1, 37, 300]))
    centers = rng.standard_normal((max(4, nlist * 2), D)).astype(np.float32)
    cands = (centers[rng.integers(0, len(centers), N)] + 0.7 * rng.standard_normal((N, D)).astype(np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, len(centers), nq)] + 0.7 * rng.standard_normal((nq, D)).astype(np.float32)).astype(np.float32)
    cd, qd = torch.from_numpy(cands).to(dev), torch.from_numpy(q).to(dev)
    index = IVFIndex(cd, nlist, iters=2)
    s, i = index.search(qd, k, nprobe)
    gs, gi = s.cpu().numpy(), i.cpu().numpy().astype(np.int64)
    _, lists = ops.retrieve_topk(qd, index.centroids, nprobe, mode="exact")
    lists = lists.cpu().numpy()
    off, orig = index.list_off.cpu().numpy(), index.orig.cpu().numpy()
    full = q.astype(np.float64) @ cands.astype(np.float64).T
    ok = True
    for r in range(nq):
        rows = np.concatenate([orig[off[l]:off[l + 1]] for l in lists[r]]) if len(lists[r]) else np.zeros(0, np.int64)
        best = rows[np.argsort(-full[r, rows], kind="stable")][:k]
        n = len(best)
        if not (np.all(gi[r, n:] == -1) and np.all(np.isneginf(gs[r, n:]))):
            ok = False; break
        if n == 0:
            continue
        es = full[r, best]
        tol = 1e-5 * max(1.0, np.abs(full[r]).max())
        if np.abs(gs[r, :n] - es).max() > tol or len(set(gi[r, :n])) != n or not set(gi[r, :n]) <= set(rows.tolist()):
            ok = False; break
        if np.abs(full[r, gi[r, :n]] - gs[r, :n]).max() > tol:
            ok = False; break
    if os.environ.get("VERBOSE") == "1" or not ok:
        print("ok  " if ok else "MISMATCH", dict(N=N, D=D, nlist=nlist, nprobe=nprobe, k=k, nq=nq, max_list=index.max_list), flush=True)
    bad += 0 if ok else 1
print("cases", N_, "mismatches", bad)
