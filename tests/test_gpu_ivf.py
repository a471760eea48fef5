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
