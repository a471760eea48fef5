"""Randomised cross-checks of the routing kernels of the row-sharded path against their NumPy statements
(tests/_cpu_kernels.py): bucket by owner (one list, segments, batched), unique rows by owner, segment sums -- worlds 1..8,
list lengths around every dispatch boundary, narrow and wide virtual row ranges, hot ids.  SEED, CASES."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from esrecsys_amd import ops
import _cpu_kernels as ref
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("CASES", "60"))
bad = 0
def check(ok, what, **kw):
    global bad
    if os.environ.get("VERBOSE") == "1":
        torch.cuda.synchronize(); print("..", what, kw, flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", what, kw, flush=True)
eq = lambda a, b: np.array_equal(a.cpu().numpy(), b.numpy())
edges = [1, 2, 255, 256, 257, 2047, 2048, 2049, 32767, 32768, 32769, 100000, 1 << 20, (1 << 20) + 1]
for case in range(N):
    world = int(rng.integers(1, 9))
    n = int(rng.choice(edges + [int(rng.integers(1, 200000))]))
    V = int(rng.choice([world, 97, 5000, 1 << 21, 3_000_000, 100_000_000]))
    hot = rng.random() < 0.4
    def ids_(m):
        x = rng.integers(0, V, m)
        if hot:
            x[rng.random(m) < 0.3] = V // 3
        return torch.from_numpy(x.astype(np.int32))
    a = ids_(n)
    got = ops.bucket_ids_by_owner(a.to(dev), world, want_inverse=True)
    exp = ref.bucket_ids_by_owner(a, world, want_inverse=True)
    check(all(eq(g, e) for g, e in zip(got, exp)), "bucket", n=n, world=world, V=V)
    # segments
    nseg = int(rng.integers(1, 4))
    m = max(1, min(n, 150000) // nseg)
    if V * nseg < (1 << 31):
        segs = [ids_(m) for _ in range(nseg)]
        offs = [i * V for i in range(nseg)]
        got = ops.bucket_ids_by_owner([t.to(dev) for t in segs], world, want_inverse=True, offsets=offs)
        exp = ref.bucket_ids_by_owner(segs, world, want_inverse=True, offsets=offs)
        check(all(eq(g, e) for g, e in zip(got, exp)), "bucket_multi", m=m, nseg=nseg, world=world, V=V)
        # unique rows by owner + segment sums
        Lv = -(-(V * nseg) // world)
        if world * Lv < (1 << 31):
            got = ops.unique_by_owner([t.to(dev) for t in segs], world, Lv, offsets=offs)
            exp = ref.unique_by_owner(segs, world, Lv, offsets=offs)
            nu = int(exp[1].sum())
            ok = eq(got[1], exp[1]) and np.array_equal(got[0].cpu().numpy()[:nu], exp[0].numpy()[:nu]) and \
                all(eq(g, e) for g, e in zip(got[2:], exp[2:]))
            check(ok, "unique_by_owner", m=m, nseg=nseg, world=world, V=V, Lv=Lv)
            D = int(rng.choice([4, 32, 100, 128]))
            if nu * D < 40_000_000 and m * nseg * D < 40_000_000:
                g = torch.from_numpy(rng.integers(-4, 5, (m * nseg, D)).astype(np.float32))  # integers: sums are exact
                out = ops.segment_sum_rows(nu, got[3], got[4], g.to(dev).clone())
                check(eq(out, ref.segment_sum_rows(nu, exp[3], exp[4], g)), "segment_sum_rows", nu=nu, D=D, m=m * nseg)
        # batched (plans of several coming batches)
        nb = int(rng.integers(1, 9))
        mb = max(1, m // 4)
        lists = [[ids_(mb) for _ in range(nseg)] for _ in range(nb)]
        gotb = ops.bucket_ids_by_owner_batched([[t.to(dev) for t in s_] for s_ in lists], world, offs)
        for b in range(nb):
            exp = ref.bucket_ids_by_owner(lists[b], world, want_inverse=True, offsets=offs)
            ok = eq(gotb[0][b], exp[0]) and eq(gotb[1][b], exp[1]) and eq(gotb[2][b], exp[2]) and eq(gotb[3][b], exp[3])
            check(ok, "bucket_batched", mb=mb, nseg=nseg, nb=nb, world=world, V=V, b=b)
print("cases", N, "mismatches", bad)
