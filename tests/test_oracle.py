"""CPU: the oracle against the committed golden vectors and against the independent autograd
transliteration (the only pins available -- the reference has no test vectors)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import autograd_ref, glove, optim, shard, stl_head, topk

F64 = np.float64
GLOVE_CASES = ["glove_uniform_d16_b64", "glove_uniform_d64_b128", "glove_same_d16_b64", "glove_zipf_d64_b128"]
STL_CASES = ["stl_b32_d8_lam0", "stl_b32_d8_lam01", "stl_b128_d32_lam01"]
INBATCH_CASES = ["inbatch_b64_d32", "inbatch_b96_d64_scale4", "inbatch_b320_d128"]


@pytest.mark.parametrize("case", GLOVE_CASES)
@pytest.mark.parametrize("mode", ["reference", "diagonal"])
def test_glove_oracle_matches_golden_and_autograd(case, mode):
    g = load_golden(case)
    grads, loss = glove.dense_grads(g["emb"].astype(F64), g["bias"].astype(F64), g["inputs"], g["target"], mode, F64)
    assert abs(loss - g["loss_" + mode]) <= 1e-13
    assert np.abs(grads["_token_embedding"]["embedding"] - g["gemb_" + mode]).max() <= 1e-13
    assert np.abs(grads["_bias"]["embedding"] - g["gbias_" + mode]).max() <= 1e-13
    l2, ge, gb = autograd_ref.glove_value_and_grad(g["emb"], g["bias"], g["inputs"], g["target"], mode)
    assert abs(loss - l2) <= 1e-12
    assert np.abs(grads["_token_embedding"]["embedding"] - ge).max() <= 1e-12
    assert np.abs(grads["_bias"]["embedding"] - gb).max() <= 1e-12


@pytest.mark.parametrize("case", GLOVE_CASES)
def test_glove_bb_broadcast_quirk(case):
    """The (B,B) output of models.py:37 and the O(B) centred loss agree with the literal O(B^2) mean."""
    g = load_golden(case)
    emb, bias = g["emb"].astype(F64), g["bias"].astype(F64)
    pred = glove.forward(emb, bias, g["inputs"], F64)
    B = g["inputs"].shape[1]
    assert pred.shape == (B, B)
    np.testing.assert_allclose(pred, g["pred_reference"], rtol=0, atol=1e-14)
    dot, s = glove.pair_terms(emb, bias, g["inputs"], F64)
    np.testing.assert_allclose(pred[3, 5], dot[5] + s[3], rtol=0, atol=1e-15)
    lit = glove.loss_literal(emb, bias, g["inputs"], g["target"], F64)
    assert abs(lit - g["loss_reference"]) <= 1e-13


def test_glove_fp32_oracle_close_to_fp64():
    g = load_golden("glove_uniform_d64_b128")
    g32, l32 = glove.dense_grads(g["emb"], g["bias"], g["inputs"], g["target"], "reference", np.float32)
    assert abs(l32 - g["loss_reference"]) / abs(g["loss_reference"]) < 1e-5
    assert rel_err(g32["_token_embedding"]["embedding"], g["gemb_reference"]) < 1e-5


@pytest.mark.parametrize("case", STL_CASES)
def test_stl_oracle_matches_golden_and_autograd(case):
    g = load_golden(case)
    lam, bs = float(g["lam"]), float(g["batch_size"])
    loss, gs, gp, gn = stl_head.triplet_loss_and_grads(g["scene"], g["pos"], g["neg"], lam, bs, F64)
    assert abs(loss - g["loss"]) <= 1e-13
    for a, b in ((gs, g["g_scene"]), (gp, g["g_pos"]), (gn, g["g_neg"])):
        assert np.abs(a - b).max() <= 1e-13
    l2, a, b, c = autograd_ref.stl_value_and_grad(g["scene"], g["pos"], g["neg"], lam, bs)
    assert abs(loss - l2) <= 1e-12 and np.abs(gs - a).max() <= 1e-12
    assert np.abs(gp - b).max() <= 1e-12 and np.abs(gn - c).max() <= 1e-12
    assert abs(stl_head.eval_loss(g["scene"], g["pos"], g["neg"], F64) - g["eval_loss"]) <= 1e-13


@pytest.mark.parametrize("case", INBATCH_CASES)
def test_inbatch_oracle_matches_golden_and_autograd(case):
    g = load_golden(case)
    lam, bs, scale = float(g["lam"]), float(g["batch_size"]), float(g["scale"])
    loss, lse, gq, gc = stl_head.inbatch_softmax_loss_and_grads(g["q"], g["c"], lam, bs, scale, F64)
    assert abs(loss - g["loss"]) <= 1e-13 and np.abs(gq - g["g_q"]).max() <= 1e-13
    l2, a, b = autograd_ref.inbatch_value_and_grad(g["q"], g["c"], lam, bs, scale)
    assert abs(loss - l2) <= 1e-12 and np.abs(gq - a).max() <= 1e-12 and np.abs(gc - b).max() <= 1e-12


def test_optimizers_match_golden_and_torch():
    g = load_golden("optim_v64_d8")
    p0 = g["p0"].astype(F64)
    st, p = optim.adam_init(p0), p0
    for k in range(3):
        p, st = optim.adam_update(p, g["adam_grads"][k].astype(F64), st, 1e-3, dtype=F64)
        assert np.abs(p - g["adam_params"][k]).max() <= 1e-14
    assert np.abs(p - autograd_ref.adam_steps(g["p0"], list(g["adam_grads"]), 1e-3)).max() <= 1e-12
    p, a = p0, optim.adagrad_init(p0)
    for k in range(3):
        p, a = optim.sparse_adagrad_update(p, a, g["ada_ids"][k], g["ada_rows"][k].astype(F64), 0.05, dtype=F64)
    assert np.abs(p - g["ada_param"]).max() <= 1e-14 and np.abs(a - g["ada_accum"]).max() <= 1e-14


def test_sparse_adagrad_equals_dense_adagrad():
    """A zero gradient leaves parameter and accumulator untouched, so the row-sparse update IS optax.adagrad."""
    rng = np.random.default_rng(5)
    V, D, n = 40, 4, 25
    p0 = rng.standard_normal((V, D))
    ids = rng.integers(0, V, n)
    rows = rng.standard_normal((n, D))
    p, a = optim.sparse_adagrad_update(p0, optim.adagrad_init(p0), ids, rows, 0.1, dtype=F64)
    dense = np.zeros((V, D))
    np.add.at(dense, ids, rows)
    pd, ad = autograd_ref.adagrad_steps(p0, [dense], 0.1)
    assert np.abs(p - pd).max() <= 1e-14 and np.abs(a - ad).max() <= 1e-14
    untouched = np.setdiff1d(np.arange(V), ids)
    assert np.array_equal(p[untouched], p0[untouched])


def test_inplace_sparse_adagrad_equals_the_plain_one():
    """The config-size form the GPU trajectory tests use (in place, reduceat sums) against sparse_adagrad_update:
    three steps with heavy duplication, zero rows and a zero accumulator."""
    rng = np.random.default_rng(6)
    V, D, n = 60, 5, 400
    p0 = rng.standard_normal((V, D))
    a0 = optim.adagrad_init(p0)
    a0[3] = 0.0
    p, a = p0.copy(), a0.copy()
    q, b = p0.copy(), a0.copy()
    for k in range(3):
        ids = rng.integers(0, V // 2, n)
        rows = rng.standard_normal((n, D))
        rows[ids == 3] = 0.0
        p, a = optim.sparse_adagrad_update(p, a, ids, rows, 0.05, dtype=F64)
        uniq = optim.sparse_adagrad_update_inplace(q, b, ids, rows, 0.05)
        assert np.array_equal(uniq, np.unique(ids))
    assert np.abs(p - q).max() <= 1e-13 and np.abs(a - b).max() <= 1e-12
    assert np.array_equal(q[V // 2:], p0[V // 2:]) and np.all(np.isfinite(q))


def test_topk_and_knn_golden_with_ties():
    g = load_golden("topk_n500_d8_k10")
    vals, idx = topk.find_top_k(g["query"], g["cand"], int(g["k"]), F64)
    assert np.array_equal(idx, g["topk_indices"]) and np.array_equal(vals, g["topk_scores"])
    assert np.all(np.diff(vals) <= 0)
    full = (g["cand"].astype(F64) * g["query"].astype(F64)).sum(-1)
    # ties broken towards the lower index (jax.lax.top_k)
    for a, b in zip(idx[:-1], idx[1:]):
        assert full[a] > full[b] or (full[a] == full[b] and a < b)
    scores, indices = glove.find_knn(g["cand"], g["token"], F64)
    assert np.array_equal(indices, g["knn_indices"])
    assert indices.shape == (500, 5) and indices.dtype == np.int32
    col = scores[indices[:, 2], 2]
    assert np.all(np.diff(col) >= 0)


def test_bucket_by_owner_roundtrip():
    rng = np.random.default_rng(3)
    ids = rng.integers(0, 1000, 257).astype(np.int32)
    for world in (1, 2, 3, 8):
        local, counts, perm = shard.bucket_by_owner(ids, world)
        assert counts.sum() == ids.size
        owners = np.repeat(np.arange(world), counts)
        assert np.array_equal(local.astype(np.int64) * world + owners, ids[perm])
        # stability: original order preserved within an owner
        for g in range(world):
            pos = perm[owners == g]
            assert np.all(np.diff(pos) > 0)


def test_cpu_port_matches_oracle():
    """oracle/cpu_port.py (the in-place torch-CPU port bench.py times as cpu_baseline) == the NumPy oracle."""
    import torch
    from oracle import cpu_port
    rng = np.random.default_rng(0)
    V, D, B, lam, lr = 200, 16, 64, 0.1, 0.05
    st0, pt0 = (rng.standard_normal((V, D)) * 0.4 for _ in range(2))
    sid, pid, nid = (rng.integers(0, V, B) for _ in range(3))
    t = lambda a: torch.from_numpy(a.copy())  # noqa: E731
    # in-batch
    st, pt, a_s, a_p = t(st0), t(pt0), torch.full((V, D), 0.1, dtype=torch.float64), torch.full((V, D), 0.1, dtype=torch.float64)
    loss = cpu_port.inbatch_step_(st, pt, a_s, a_p, t(sid), t(pid), lam, float(B), 2.0, lr)
    el, _, gq, gc = stl_head.inbatch_softmax_loss_and_grads(st0[sid], pt0[pid], lam, B, 2.0, F64)
    es, _ = optim.sparse_adagrad_update(st0, np.full((V, D), 0.1), sid, gq, lr, dtype=F64)
    ep, _ = optim.sparse_adagrad_update(pt0, np.full((V, D), 0.1), pid, gc, lr, dtype=F64)
    assert abs(float(loss) - el) <= 1e-12 and np.abs(st.numpy() - es).max() <= 1e-12
    assert np.abs(pt.numpy() - ep).max() <= 1e-12
    # triplet
    st, pt, a_s, a_p = t(st0), t(pt0), torch.full((V, D), 0.1, dtype=torch.float64), torch.full((V, D), 0.1, dtype=torch.float64)
    loss = cpu_port.triplet_step_(st, pt, a_s, a_p, t(sid), t(pid), t(nid), lam, float(B), lr)
    el, gs, gp, gn = stl_head.triplet_loss_and_grads(st0[sid], pt0[pid], pt0[nid], lam, B, F64)
    es, _ = optim.sparse_adagrad_update(st0, np.full((V, D), 0.1), sid, gs, lr, dtype=F64)
    ep, _ = optim.sparse_adagrad_update(pt0, np.full((V, D), 0.1), np.concatenate([pid, nid]),
                                        np.concatenate([gp, gn]), lr, dtype=F64)
    assert abs(float(loss) - el) <= 1e-12 and np.abs(st.numpy() - es).max() <= 1e-12
    assert np.abs(pt.numpy() - ep).max() <= 1e-12
    # glove
    emb0, bias0 = rng.standard_normal((V, D)) * 0.3, rng.standard_normal((V, 1)) * 0.05
    inputs = rng.integers(0, V, (2, B))
    target = rng.uniform(0.1, 300, B)
    emb, bias = t(emb0), t(bias0)
    ae, ab = torch.full((V, D), 0.1, dtype=torch.float64), torch.full((V, 1), 0.1, dtype=torch.float64)
    loss = cpu_port.glove_step_(emb, bias, ae, ab, t(inputs), t(target), lr)
    el, gdot, gs_ = glove.loss_and_grads(emb0, bias0, inputs, target, "reference", F64)
    ids, rows, gb = glove.row_grads(emb0, inputs, gdot, gs_, F64)
    ee, _ = optim.sparse_adagrad_update(emb0, np.full((V, D), 0.1), ids, rows, lr, dtype=F64)
    eb, _ = optim.sparse_adagrad_update(bias0, np.full((V, 1), 0.1), ids, gb[:, None], lr, dtype=F64)
    assert abs(float(loss) - el) <= 1e-12 and np.abs(emb.numpy() - ee).max() <= 1e-12
    assert np.abs(bias.numpy() - eb).max() <= 1e-12


def test_cpu_port_dense_adam_variants_match_oracle():
    """The reference-faithful CPU baseline variants (dense V x D gradient + optax.adam on every element) == the NumPy
    oracle's dense gradient + adam_update, two steps (so the moment estimates and the bias correction are exercised)."""
    import torch
    from oracle import cpu_port
    rng = np.random.default_rng(3)
    V, D, B, lam, lr = 150, 12, 48, 0.1, 1e-3
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float64))  # noqa: E731

    def dense(ids, rows, shape):
        g = np.zeros(shape)
        np.add.at(g, ids, rows)
        return g

    # triplet head, the reference's own step (pinterest/train_shop_the_look.py:93-109)
    st0, pt0 = (rng.standard_normal((V, D)) * 0.4 for _ in range(2))
    ds, dp = cpu_port.DenseAdam(t(st0), lr), cpu_port.DenseAdam(t(pt0), lr)
    es, ep, ss, sp = st0.copy(), pt0.copy(), optim.adam_init(st0), optim.adam_init(pt0)
    for _ in range(2):
        sid, pid, nid = (rng.integers(0, V, B) for _ in range(3))
        loss = cpu_port.triplet_step_dense_adam_(ds, dp, torch.from_numpy(sid), torch.from_numpy(pid),
                                                 torch.from_numpy(nid), lam, float(B))
        el, gs, gp, gn = stl_head.triplet_loss_and_grads(es[sid], ep[pid], ep[nid], lam, B, F64)
        assert abs(float(loss) - el) <= 1e-12
        es, ss = optim.adam_update(es, dense(sid, gs, es.shape), ss, lr, dtype=F64)
        ep, sp = optim.adam_update(ep, dense(np.concatenate([pid, nid]), np.concatenate([gp, gn]), ep.shape), sp, lr,
                                   dtype=F64)
    assert np.abs(ds.p.numpy() - es).max() <= 1e-12 and np.abs(dp.p.numpy() - ep).max() <= 1e-12
    # GloVe (wikipedia/train_cooccurence.py:71-101)
    emb0, bias0 = rng.standard_normal((V, D)) * 0.3, rng.standard_normal((V, 1)) * 0.05
    ae, ab = cpu_port.DenseAdam(t(emb0), lr), cpu_port.DenseAdam(t(bias0), lr)
    ee, eb, se, sb = emb0.copy(), bias0.copy(), optim.adam_init(emb0), optim.adam_init(bias0)
    for _ in range(2):
        inputs, target = rng.integers(0, V, (2, B)), rng.uniform(0.1, 300, B)
        loss = cpu_port.glove_step_dense_adam_(ae, ab, torch.from_numpy(inputs), torch.from_numpy(target))
        el, gdot, gs_ = glove.loss_and_grads(ee, eb, inputs, target, "reference", F64)
        ids, rows, gb = glove.row_grads(ee, inputs, gdot, gs_, F64)
        assert abs(float(loss) - el) <= 1e-12
        ee, se = optim.adam_update(ee, dense(ids, rows, ee.shape), se, lr, dtype=F64)
        eb, sb = optim.adam_update(eb, dense(ids, gb[:, None], eb.shape), sb, lr, dtype=F64)
    assert np.abs(ae.p.numpy() - ee).max() <= 1e-12 and np.abs(ab.p.numpy() - eb).max() <= 1e-12
    # in-batch loss with the dense update
    ds, dp = cpu_port.DenseAdam(t(st0), lr), cpu_port.DenseAdam(t(pt0), lr)
    sid, pid = rng.integers(0, V, B), rng.integers(0, V, B)
    loss = cpu_port.inbatch_step_dense_adam_(ds, dp, torch.from_numpy(sid), torch.from_numpy(pid), lam, float(B), 2.0)
    el, _, gq, gc = stl_head.inbatch_softmax_loss_and_grads(st0[sid], pt0[pid], lam, B, 2.0, F64)
    es, _ = optim.adam_update(st0, dense(sid, gq, st0.shape), optim.adam_init(st0), lr, dtype=F64)
    assert abs(float(loss) - el) <= 1e-12 and np.abs(ds.p.numpy() - es).max() <= 1e-12


def test_round_bf16_matches_torch_rne():
    """oracle.optim.round_bf16 (RNE to 8 significant bits, straight from fp64) against torch's float32 -> bfloat16
    conversion on f32-representable inputs, ties included."""
    import torch
    from oracle import optim as o_optim
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.uniform(-6, 6, 20000), [0.0, -0.0, 1.0, -1.5],
                        # exact ties: 1 + k 2^-8 for odd k (halfway between two bf16 neighbours)
                        1.0 + (2 * np.arange(64) + 1) * 2.0 ** -8, -(2.0 + (2 * np.arange(64) + 1) * 2.0 ** -7)])
    x = x.astype(np.float32)
    want = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    got = o_optim.round_bf16(x.astype(np.float64))
    assert np.array_equal(got, want.astype(np.float64))
