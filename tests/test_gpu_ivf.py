"""GPU: the IVF index (esrecsys_amd/ivf.py, esr_ivf.hip) against the exact brute force (the reference's jax.lax.top_k,
pinterest/make_recommendations.py:49-65): probing ALL lists must reproduce it; probing a few must return exact scores
of real candidates and, on a clustered corpus, most of the true top-k."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clustered(rng, n, D, centers, sigma):
    c = rng.standard_normal((centers, D)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    pick = rng.integers(0, centers, n)
    return (c[pick] + sigma * rng.standard_normal((n, D)).astype(np.float32) / np.sqrt(D)).astype(np.float32), c


@pytest.mark.parametrize("N,D,nlist,k", [(20_000, 128, 64, 10), (50_000, 512, 128, 500), (3_000, 64, 16, 100)])
def test_ivf_probing_every_list_is_the_brute_force(dev, N, D, nlist, k):
    from esrecsys_amd import ops
    from esrecsys_amd.ivf import IVFIndex
    rng = np.random.default_rng(N)
    cands, _ = _clustered(rng, N, D, 200, 0.7)
    q, _ = _clustered(rng, 300, D, 200, 0.7)
    cd, qd = torch.from_numpy(cands).to(dev), torch.from_numpy(q).to(dev)
    index = IVFIndex(cd, nlist, iters=3)
    assert int(index.list_off[-1]) == N and sorted(index.orig.cpu().tolist()) == list(range(N))
    s, i = index.search(qd, k, nlist)
    full = q.astype(np.float64) @ cands.astype(np.float64).T
    order = np.lexsort((np.arange(N)[None, :].repeat(len(q), 0), -full), axis=1)[:, :k]
    es = np.take_along_axis(full, order, 1)
    gs, gi = s.cpu().numpy(), i.cpu().numpy().astype(np.int64)
    assert np.abs(gs - es).max() <= 1e-5 * np.abs(es).max()
    assert np.abs(np.take_along_axis(full, gi, 1) - gs).max() <= 1e-5 * np.abs(es).max()   # every score belongs to its index
    assert np.mean(gi == order) > 0.99 and all(len(set(r)) == k for r in gi)
    bs, bi = ops.retrieve_topk(qd, cd, k, mode="exact")
    assert np.mean(bi.cpu().numpy() == gi) > 0.99


def test_ivf_few_probes_on_a_clustered_corpus(dev):
    from esrecsys_amd.ivf import IVFIndex
    from esrecsys_amd.pinterest.make_recommendations import recall_at_k
    from esrecsys_amd import ops
    rng = np.random.default_rng(7)
    N, D, k = 100_000, 128, 10
    cands, centers = _clustered(rng, N, D, 400, 0.5)
    q = (centers[rng.integers(0, 400, 500)] + 0.5 * rng.standard_normal((500, D)).astype(np.float32) / np.sqrt(D)).astype(np.float32)
    cd, qd = torch.from_numpy(cands).to(dev), torch.from_numpy(q).to(dev)
    index = IVFIndex(cd, 256, iters=5)
    _, exact = ops.retrieve_topk(qd, cd, k, mode="exact")
    recalls = []
    for nprobe in (1, 8, 32):
        s, i = index.search(qd, k, nprobe)
        gi = i.cpu().numpy().astype(np.int64)
        valid = gi >= 0
        got = (q.astype(np.float64)[:, None, :] * cands.astype(np.float64)[np.maximum(gi, 0)]).sum(-1)
        assert np.abs(np.where(valid, got - s.cpu().numpy(), 0.0)).max() <= 1e-5 * np.abs(got).max()   # exact scores of real rows
        assert np.all(np.diff(s.cpu().numpy(), axis=1) <= 0)
        recalls.append(recall_at_k(i, exact))
    print("IVF recall@10 at nprobe 1 / 8 / 32 of 256 lists:", ["%.3f" % r for r in recalls])
    assert recalls[0] <= recalls[1] + 1e-9 <= recalls[2] + 2e-9 and recalls[2] > 0.9


@pytest.mark.parametrize("N,D,nlist,k,nprobe", [(30_000, 64, 64, 10, 5), (30_000, 64, 64, 500, 7), (60_000, 128, 32, 1000, 20),
                                                (4_000, 32, 16, 300, 3)])
def test_ivf_returns_the_exact_top_k_of_the_probed_lists(dev, N, D, nlist, k, nprobe):
    """The search is exact INSIDE what it probes: head lists scored densely, the others through the tau filter in rounds
    of eight probe slots with a compacting select in between -- the answer must be the k best candidates of the union
    of the query's nprobe lists (k beyond one list's length, lists shorter than k, fewer candidates than k: -inf / -1)."""
    from esrecsys_amd import ops
    from esrecsys_amd.ivf import IVFIndex
    rng = np.random.default_rng(N + k)
    cands, _ = _clustered(rng, N, D, 150, 0.8)
    q, _ = _clustered(rng, 200, D, 150, 0.8)
    cd, qd = torch.from_numpy(cands).to(dev), torch.from_numpy(q).to(dev)
    index = IVFIndex(cd, nlist, iters=3)
    s, i = index.search(qd, k, nprobe)
    gs, gi = s.cpu().numpy(), i.cpu().numpy().astype(np.int64)
    _, lists = ops.retrieve_topk(qd, index.centroids, nprobe, mode="exact")
    lists = lists.cpu().numpy()
    off, orig = index.list_off.cpu().numpy(), index.orig.cpu().numpy()
    full = q.astype(np.float64) @ cands.astype(np.float64).T
    for r in range(len(q)):
        rows = np.concatenate([orig[off[l]:off[l + 1]] for l in lists[r]])
        best = rows[np.argsort(-full[r, rows], kind="stable")][:k]
        n = len(best)
        assert np.all(gi[r, n:] == -1) and np.all(np.isneginf(gs[r, n:]))
        es = full[r, best]
        assert np.abs(gs[r, :n] - es).max() <= 1e-5 * max(1.0, np.abs(es).max())
        assert len(set(gi[r, :n])) == n and set(gi[r, :n]) <= set(rows.tolist())
        assert np.mean(np.isin(gi[r, :n], best)) > 0.99   # (near-ties in f32 may swap the last entries)
