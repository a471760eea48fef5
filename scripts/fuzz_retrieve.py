"""Randomised check of the MFMA retrieval (esr_retrieve_topk, every mode) with real-valued operands against an fp64
brute force: the scores returned must be the fp64 scores of the returned rows (mode-dependent tolerance) and the k-th
returned score must not be below the true k-th by more than that tolerance; index_base / index_step; shapes around the
chunk and tile boundaries.  SEED, CASES."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esrecsys_amd import ops
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N_ = int(os.environ.get("CASES", "40"))
bad = 0
TOL = {"exact": 2e-6, "f32": 2e-6, "f16x2": 2e-6, "f16r": 2e-6, "bf16x3": 2e-6, "bf16": 2e-2}
for case in range(N_):
    nq = int(rng.choice([1, 37, 255, 256, 257, 1000]))
    N = int(rng.choice([1, 127, 128, 129, 8000, 8001, 40000, 70001, 200000]))
    D = int(rng.choice([4, 17, 32, 96, 128, 130, 512]))
    k = int(min(N, rng.choice([1, 10, 500, 1024])))
    mq, mc = 10 ** rng.uniform(-3, 2), 10 ** rng.uniform(-3, 2)
    q = (rng.standard_normal((nq, D)) * mq).astype(np.float32)
    c = (rng.standard_normal((N, D)) * mc).astype(np.float32)
    if rng.random() < 0.3:
        c[rng.integers(0, N, max(1, N // 50))] *= 30.0     # outlier rows
    full = q.astype(np.float64) @ c.astype(np.float64).T
    kth = -np.sort(-full, axis=1)[:, k - 1]
    # the error of a dot product scales with |q| |c|, not with the score: with one query and one candidate the only score
    # can cancel to far below that (seed 505: nq = N = k = 1, D = 96 in the one-plane bf16 mode) -- never let the yardstick
    # fall below the typical score of such a pair
    scale = max(np.abs(full).max(), np.linalg.norm(q.astype(np.float64), axis=1).max() *
                np.linalg.norm(c.astype(np.float64), axis=1).max() / np.sqrt(D))
    base, step = int(rng.integers(0, 100)), int(rng.choice([1, 2, 8]))
    for mode in ("exact", "f16x2", "f16r", "bf16x3", "bf16"):
        s, i = ops.retrieve_topk(torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev), k, mode=mode, index_base=base, index_step=step)
        if mode != "exact":  # the prepared-corpus form gives the same answer bit for bit
            qd, cd = torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev)
            s, i = ops.retrieve_topk(qd, cd, k, mode=mode, index_base=base, index_step=step)
            s2, i2 = ops.retrieve_topk(qd, cd, k, mode=mode, index_base=base, index_step=step,
                                       prepared=ops.retrieve_prepare(cd, mode=mode))
            if not (torch.equal(s, s2) and torch.equal(i, i2)):
                print("MISMATCH prepared vs plain", dict(nq=nq, N=N, D=D, k=k), flush=True)
                bad += 1
        gs, gi = s.cpu().numpy().astype(np.float64), i.cpu().numpy().astype(np.int64)
        rows = (gi - base) // step
        ok = np.all((gi - base) % step == 0) and rows.min() >= 0 and rows.max() < N and \
            all(len(set(r)) == k for r in rows) and np.all(np.diff(gs, axis=1) <= 0)
        if ok:
            tol = TOL[mode] * scale
            ok = np.abs(np.take_along_axis(full, rows, 1) - gs).max() <= tol and np.all(gs[:, -1] >= kth - 2 * tol)
        if os.environ.get("VERBOSE") == "1" or not ok:
            print("ok  " if ok else "MISMATCH", dict(mode=mode, nq=nq, N=N, D=D, k=k, base=base, step=step), flush=True)
        bad += 0 if ok else 1
print("cases", N_, "mismatches", bad)
