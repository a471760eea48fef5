"""Randomised cross-checks of the id sorts, the column argsort, score_topk / retrieve_topk and the IVF search against NumPy
(SEED, CASES): sizes and id ranges drawn at random around every dispatch boundary of the library."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esrecsys_amd import ops
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("CASES", "60"))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
VERBOSE = os.environ.get("VERBOSE") == "1"
def note(what, **kw):
    if VERBOSE:
        torch.cuda.synchronize()
        print("..", what, kw, flush=True)
def check(ok, what, **kw):
    global bad
    note(what, **kw)
    if not ok:
        bad += 1
        print("MISMATCH", what, kw, flush=True)
edges = [1, 2, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 32767, 32768, 32769, 262143, 262144, 262145]
for case in range(N):
    # --- sorts
    n = int(rng.choice(edges + [int(rng.integers(1, 300000))]))
    V = int(rng.choice([2, 17, 2048, 2049, 1 << 21, (1 << 21) + 1, 1 << 30]))
    hot = rng.random() < 0.4
    def ids_(m):
        x = rng.integers(0, V, m)
        if hot:
            x[rng.random(m) < 0.3] = V // 2
        return x.astype(np.int32)
    a = ids_(n)
    note("before segment_sort", case=case, n=n, V=V)
    s, p = ops.segment_sort(T(a), V)
    o = np.argsort(a, kind="stable")
    check(np.array_equal(p.cpu().numpy(), o) and np.array_equal(s.cpu().numpy(), a[o]), "segment_sort", n=n, V=V)
    nb = int(rng.integers(1, 9))
    nseg = int(rng.integers(1, 4))
    m = max(1, n // (nseg * 2))
    if V * nseg < (1 << 31):
        lists = [[T(ids_(m)) for _ in range(nseg)] for _ in range(nb)]
        offs = [i * V for i in range(nseg)]
        note("before batched", m=m, nseg=nseg, nb=nb, V=V)
        sb, pb = ops.segment_sort_batched(lists, offs, V * nseg)
        for b in range(nb):
            virt = np.concatenate([t.cpu().numpy().astype(np.int64) + o_ for t, o_ in zip(lists[b], offs)])
            o = np.argsort(virt, kind="stable")
            check(np.array_equal(pb[b].cpu().numpy(), o) and np.array_equal(sb[b].cpu().numpy(), virt[o]),
                  "segment_sort_batched", m=m, nseg=nseg, nb=nb, V=V, b=b)
    # --- argsort_columns / topk_columns
    Vr, Tc = int(rng.choice([1, 5, 100, 2048, 2049, 5000, 70000])), int(rng.integers(1, 12))
    x = (rng.integers(-50, 50, (Vr, Tc)) / 4.0).astype(np.float32) if rng.random() < 0.5 else \
        rng.standard_normal((Vr, Tc)).astype(np.float32)
    note("before argsort", V=Vr, T=Tc)
    got = ops.argsort_columns(T(x)).cpu().numpy()
    check(np.array_equal(got, np.argsort(x, axis=0, kind="stable")), "argsort_columns", V=Vr, T=Tc)
    # --- score_topk
    nq, Nc, D = int(rng.integers(1, 12)), int(rng.choice([7, 100, 999, 2048, 5000, 40000])), int(rng.choice([4, 32, 96, 128]))
    k = int(min(Nc, rng.choice([1, 10, 500, 1024, 1025, 3000])))
    q = rng.integers(-3, 4, (nq, D)).astype(np.float32)
    c = rng.integers(-3, 4, (Nc, D)).astype(np.float32)
    note("before score_topk", nq=nq, N=Nc, D=D, k=k)
    s_, i_ = ops.score_topk(T(q), T(c), k)
    full = q @ c.T
    order = np.argsort(-full, axis=1, kind="stable")[:, :k]
    check(np.array_equal(i_.cpu().numpy(), order) and np.array_equal(s_.cpu().numpy(), np.take_along_axis(full, order, 1)),
          "score_topk", nq=nq, N=Nc, D=D, k=k)
    # --- retrieve_topk (integer-valued operands: every mode is exact, ties -> lower index)
    if k <= 1024:
        for mode in ("exact", "f16x2", "bf16x3"):
            note("before retrieve", mode=mode)
            s2, i2 = ops.retrieve_topk(T(q), T(c), k, mode=mode)
            check(np.array_equal(i2.cpu().numpy(), order) and np.array_equal(s2.cpu().numpy(), np.take_along_axis(full, order, 1)),
                  "retrieve_topk", mode=mode, nq=nq, N=Nc, D=D, k=k)
print("cases", N, "mismatches", bad)
