"""GPU: edge cases and error behaviour through the C ABI -- empty and one-element inputs, ragged sizes, maximum k,
argument errors that must raise instead of launching ('returns 0 / negative ESR_E* code, never throws';
the Python layer turns the code into EsrLibraryError with esr_last_error()'s text)."""
import numpy as np
import pytest
import torch

from oracle import optim as o_optim
from oracle import stl_head as o_stl
from oracle import topk as o_topk

pytestmark = pytest.mark.gpu
F64 = np.float64


def T(x, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return t.to(dtype) if dtype is not None else t


def N(t):
    return t.detach().cpu().numpy()


def test_empty_occurrence_lists_are_no_ops(dev):
    from esrecsys_amd import ops
    table = torch.randn((100, 16), device=dev)
    before = table.clone()
    accum = torch.full((100, 16), 0.1, device=dev)
    ids = torch.empty(0, dtype=torch.int32, device=dev)
    assert ops.gather_rows(table, ids).shape == (0, 16)
    sid, perm = ops.segment_sort(ids, 100)
    assert sid.numel() == 0 and perm.numel() == 0
    ops.sparse_adagrad(table, accum, sid, perm, torch.empty((0, 16), device=dev), 0.1)
    local, p2, counts = ops.bucket_ids_by_owner(ids, 8)
    assert local.numel() == 0 and int(counts.sum()) == 0
    srt, prm = ops.segment_sort_batched([[ids, ids], [ids, ids]], (0, 100), 200)   # two empty two-segment lists
    assert srt.shape == (2, 0) and prm.shape == (2, 0)
    torch.cuda.synchronize()
    assert torch.equal(table, before) and bool((accum == 0.1).all())


def test_single_pair_and_single_triplet(dev):
    """B = 1: the in-batch softmax over one candidate has loss 0 + reg and zero score gradient"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(0)
    q, c, n = (rng.standard_normal((1, 32)).astype(np.float32) for _ in range(3))
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 2.0, 0.0, 1.0)
    assert abs(float(loss)) <= 1e-6 and float(gq.abs().max()) <= 1e-6 and float(gc.abs().max()) <= 1e-6
    st, pt = T(q, dev), T(np.concatenate([c, n]), dev)
    z = torch.zeros(1, dtype=torch.int32, device=dev)
    o = torch.ones(1, dtype=torch.int32, device=dev)
    out = ops.triplet_fwd_bwd(st, pt, pt, z, z, o, 1, 0.1, 1.0)
    el, gs, gp, gn = o_stl.triplet_loss_and_grads(q.astype(F64), c.astype(F64), n.astype(F64), 0.1, 1, F64)
    assert abs(float(out[0]) - el) <= 1e-5 * max(1.0, abs(el))
    assert np.abs(N(out[3]) - gs).max() <= 1e-5 * max(1e-6, np.abs(gs).max())


def test_all_occurrences_hit_one_row(dev):
    """every id equal: one segment of length n; the update equals the oracle's left-to-right sum"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(1)
    V, D, n = 50, 64, 5000
    table = rng.standard_normal((V, D)).astype(np.float32)
    rows = rng.standard_normal((n, D)).astype(np.float32)
    ids = np.full(n, 17, np.int32)
    t, a = T(table, dev), torch.full((V, D), 0.1, device=dev)
    sid, perm = ops.segment_sort(T(ids, dev), V)
    ops.sparse_adagrad(t, a, sid, perm, T(rows, dev), 0.05)
    ep, ea = o_optim.sparse_adagrad_update(table.astype(F64), np.full((V, D), 0.1), ids, rows.astype(F64), 0.05, dtype=F64)
    assert np.abs(N(t) - ep).max() <= 1e-4 * np.abs(ep).max()     # 5000-term f32 sum against f64
    assert np.array_equal(N(t)[np.arange(V) != 17], table[np.arange(V) != 17])  # other rows untouched, bit for bit


def test_retrieve_one_query_k_one_and_k_max(dev):
    from esrecsys_amd import ops
    rng = np.random.default_rng(2)
    q = (rng.integers(-8, 9, (1, 48)) / 4.0).astype(np.float32)
    c = (rng.integers(-8, 9, (3000, 48)) / 4.0).astype(np.float32)
    for k in (1, 1024):
        s, i = ops.retrieve_topk(T(q, dev), T(c, dev), k)
        es, ei = o_topk.batched_top_k(q, c, k, F64)
        assert np.array_equal(N(i), ei) and np.array_equal(N(s), es.astype(np.float32))


def test_argument_errors_raise_with_a_message(dev):
    from esrecsys_amd import _lib, ops
    q = torch.randn((4, 32), device=dev)
    c = torch.randn((100, 32), device=dev)
    with pytest.raises(_lib.EsrLibraryError, match="k"):
        ops.retrieve_topk(q, c, 101)             # k > N
    with pytest.raises(_lib.EsrLibraryError, match="k"):
        ops.retrieve_topk(q, torch.randn((5000, 32), device=dev), 2000)   # k > 1024
    with pytest.raises(ValueError, match="128"):
        ops.inbatch_softmax_fwd_bwd(torch.randn((100, 128), device=dev), torch.randn((100, 128), device=dev), 1.0, 0.0,
                                    100.0, precision="bf16x3")            # B not a multiple of 128
    with pytest.raises(_lib.EsrLibraryError, match="not supported"):
        ops.inbatch_softmax_fwd_bwd(torch.randn((64, 98), device=dev), torch.randn((64, 98), device=dev), 1.0, 0.0,
                                    64.0)                                 # D must be a multiple of 4 (96 is fine now)
    with pytest.raises(TypeError):
        ops.gather_rows(c, torch.zeros(3, dtype=torch.int64, device=dev))  # ids must be int32
    three = torch.zeros(3, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.EsrLibraryError, match="nbatch"):
        ops.segment_sort_batched([[three]] * 9, (0,), 10)                  # at most eight lists per call
    with pytest.raises(ValueError, match="same segment lengths"):
        ops.segment_sort_batched([[three], [torch.zeros(4, dtype=torch.int32, device=dev)]], (0,), 10)
    with pytest.raises(TypeError, match="no CPU fallback"):
        ops.gather_rows(c.cpu(), torch.zeros(3, dtype=torch.int32))
    with pytest.raises(_lib.EsrLibraryError):
        ops.spotify_fwd_bwd(torch.randn((100, 200), device=dev), torch.randn((10, 200), device=dev),
                            torch.zeros(3, dtype=torch.int32, device=dev), torch.zeros(3, dtype=torch.int32, device=dev),
                            1, 1, 1, 1.0)                                  # 2F > 256
    # the failed calls left the library usable
    assert ops.gather_rows(c, torch.zeros(3, dtype=torch.int32, device=dev)).shape == (3, 32)


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_inbatch_bf16x3_nonfinite_input_gives_nan_loss(dev, bad, precision):
    """the bf16x3 path reduces its loss in fixed point through integer atomics: a non-finite partial must still come
    out as NaN (poison word), and the next call on the same workspace must be clean again"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(3)
    B, D = 256, 128
    q = (rng.standard_normal((B, D)) * D ** -0.5).astype(np.float32)
    c = (rng.standard_normal((B, D)) * D ** -0.5).astype(np.float32)
    qb = q.copy()
    qb[77, 5] = bad
    loss, _, _, _ = ops.inbatch_softmax_fwd_bwd(T(qb, dev), T(c, dev), 4.0, 0.1, float(B), precision=precision)
    assert np.isnan(float(loss))
    loss2, _, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 4.0, 0.1, float(B), precision=precision)
    el, _, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(F64), c.astype(F64), 0.1, float(B), 4.0, F64)
    assert abs(float(loss2) - el) <= 1e-5 * abs(el)
    assert np.max(np.abs(N(gq) - egq)) <= 1e-5 * np.max(np.abs(egq))


@pytest.mark.parametrize("n,D", [(100_000, 128), (70_001, 256), (33, 16)])
def test_sparse_adagrad_one_row_takes_every_gradient(dev, n, D):
    """the worst hot row: every occurrence is the same id (one run of n, thousands of chunk partials)"""
    import time
    from esrecsys_amd import ops
    rng = np.random.default_rng(n)
    V = 1000
    ids = np.full(n, 123, np.int32)
    p0 = rng.standard_normal((V, D)).astype(np.float32)
    a0 = np.full((V, D), 0.1, np.float32)
    rows = (rng.standard_normal((n, D)) * 0.01).astype(np.float32)
    table, accum, grads = T(p0, dev), T(a0, dev), T(rows, dev)
    sid, perm = ops.segment_sort(T(ids, dev), V)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.sparse_adagrad(table, accum, sid, perm, grads, 0.05, 1e-7)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.5   # seconds: chunked, not one thread walking the run
    g = rows.astype(F64).sum(0)
    ea = a0[123].astype(F64) + g * g
    ep = p0[123].astype(F64) - 0.05 * g / np.sqrt(ea + 1e-7)
    assert np.max(np.abs(N(accum)[123] - ea)) <= 1e-5 * np.max(np.abs(ea))
    assert np.max(np.abs(N(table)[123] - ep)) <= 1e-5 * np.max(np.abs(ep))
    keep = np.arange(V) != 123
    assert np.array_equal(N(table)[keep], p0[keep]) and np.array_equal(N(accum)[keep], a0[keep])


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("scale", [-6.0, 0.0, 1e-3])
def test_inbatch_odd_temperatures(dev, precision, scale):
    """negative temperature (the softmax prefers the LEAST similar candidate), zero (uniform: loss = log B + reg) and a
    tiny one; batch_size != B as the reference's caller may pass any divisor (train_shop_the_look.py:104)"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(21)
    B, D, bs = 256, 128, 77.0
    q = (rng.standard_normal((B, D)) * 0.2).astype(np.float32)
    c = (rng.standard_normal((B, D)) * 0.2).astype(np.float32)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), scale, 0.1, bs, precision=precision)
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(F64), c.astype(F64), 0.1, bs, scale, F64)
    assert abs(float(loss) - el) <= 1e-5 * abs(el)
    assert np.max(np.abs(N(lse) - else_)) <= 1e-5 * np.max(np.abs(else_))
    assert np.max(np.abs(N(gq) - egq)) <= 1e-5 * np.max(np.abs(egq))
    assert np.max(np.abs(N(gc) - egc)) <= 1e-5 * np.max(np.abs(egc))
    if scale == 0.0:
        reg = 0.1 * (np.maximum(np.linalg.norm(q.astype(F64), axis=1) - 1, 0).sum()
                     + np.maximum(np.linalg.norm(c.astype(F64), axis=1) - 1, 0).sum())
        assert abs(float(loss) - (B * np.log(B) + reg) / bs) <= 1e-5 * abs(el)


def test_triplet_head_on_the_kinks(dev):
    """relu at exactly 0 (jax.nn.relu has derivative 0 there): margin 1 + neg - pos == 0 contributes no
    gradient; a row of norm exactly 1 has no regulariser gradient, a row just above has; an all-zero row does not
    produce NaN (the reference's sqrt'(0) * 0 does: DESIGN 5)."""
    from esrecsys_amd import ops
    D, B = 4, 4
    scene = np.array([[1, 0, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 0]], np.float32)
    pos = np.array([[1, 0, 0, 0], [0.5, 0, 0, 0], [0, 0, 2, 0], [1, 1, 1, 1]], np.float32)
    neg = np.array([[0, 0, 0, 0], [0.5, 0, 0, 0], [0, 0, 0, 3], [1, 0, 0, 0]], np.float32)
    # row 0: 1 + 0 - 1 == 0 exactly -> no triplet gradient; row 1: 1 + .5 - .5 = 1 > 0; row 2: scene norm exactly 1
    loss, ps, ns, gs, gp, gn = ops.triplet_fwd_bwd(T(scene, dev), T(pos, dev), T(neg, dev), None, None, None, B, 0.5,
                                                   float(B))
    gs, gp, gn = N(gs), N(gp), N(gn)
    assert np.all(np.isfinite(gs)) and np.all(np.isfinite(gp)) and np.all(np.isfinite(gn)) and np.isfinite(float(loss))
    assert np.array_equal(gs[0], np.zeros(D)) and np.array_equal(gp[0], np.zeros(D))      # on the kink: nothing
    assert np.allclose(gs[1], (neg[1] - pos[1]) / B) and np.allclose(gp[1], -scene[1] / B) and np.allclose(gn[1], scene[1] / B)
    # row 2: margin 1 + 0 - 2 < 0: only regularisers; scene norm == 1 -> none; pos norm 2 -> lam * p / |p| / B
    assert np.array_equal(gs[2], np.zeros(D))
    assert np.allclose(gp[2], 0.5 * pos[2] / 2.0 / B) and np.allclose(gn[2], 0.5 * neg[2] / 3.0 / B)
    assert np.array_equal(gs[3], (neg[3] - pos[3]) / B)                                   # zero row: margin 1 > 0
    exp_loss = (0.0 + 1.0 + 0.0 + 1.0 + 0.5 * ((2 - 1) + (3 - 1) + (2 - 1))) / B
    assert abs(float(loss) - exp_loss) <= 1e-6
    el, egs, egp, egn = o_stl.triplet_loss_and_grads(scene, pos, neg, 0.5, float(B))
    assert abs(el - exp_loss) <= 1e-12 and np.allclose(gs, egs) and np.allclose(gp, egp) and np.allclose(gn, egn)


@pytest.mark.parametrize("mode", ["reference", "diagonal"])
def test_glove_weight_function_corners(dev, mode):
    """counts on the corners of w = min(1, c / 100) ** 0.75 and log10(1 + c) (train_cooccurence.py:79-82): zero (weight 0,
    target 0), the knee at exactly 100 and one float on either side, very large and denormal-small counts; t1 == t2 pairs
    (a token co-occurring with itself: both occurrences hit the same row)"""
    from esrecsys_amd import ops
    from oracle import glove as o_glove
    rng = np.random.default_rng(31)
    V, D, B = 50, 64, 16
    emb = (rng.standard_normal((V, D)) * 0.3).astype(np.float32)
    bias = (rng.standard_normal((V, 1)) * 0.1).astype(np.float32)
    inputs = rng.integers(0, V, (2, B)).astype(np.int32)
    inputs[1, :4] = inputs[0, :4]
    hundred = np.float32(100.0)
    target = np.array([0.0, 100.0, np.nextafter(hundred, np.float32(0)), np.nextafter(hundred, np.float32(1e9)), 1e6,
                       3.0e38, 1e-38, 1e-45, 1.0, 99.0, 101.0, 0.5, 7.25, 250.0, 1e-3, 50.0], np.float32)
    loss, grows, gbias = ops.glove_fwd_bwd(T(emb, dev), T(bias, dev), T(inputs, dev), T(target, dev),
                                           ops.GLOVE_REFERENCE if mode == "reference" else ops.GLOVE_DIAGONAL)
    el, gdot, gs = o_glove.loss_and_grads(emb, bias, inputs, target, mode, F64)
    _, erows, ebias = o_glove.row_grads(emb, inputs, gdot, gs, F64)
    assert np.isfinite(float(loss)) and abs(float(loss) - el) <= 1e-5 * abs(el)
    assert np.max(np.abs(N(grows) - erows)) <= 1e-5 * np.max(np.abs(erows))
    assert np.max(np.abs(N(gbias).reshape(-1) - ebias)) <= 1e-5 * np.max(np.abs(ebias))
    if mode == "diagonal":   # a zero count has zero weight: its pair contributes no gradient at all
        assert np.array_equal(N(grows)[0], np.zeros(D)) and np.array_equal(N(grows)[B], np.zeros(D))


def test_out_of_range_device_ids_raise_under_check_ids(dev, monkeypatch):
    """ESR_CHECK_IDS=1: device-resident ids are screened by esr_check_ids before any gather / scatter sees them
    (host ids are always range-checked; device ids are trusted by default because the check costs a sync)."""
    from esrecsys_amd import TrainState, ops, optim
    from esrecsys_amd.pinterest.models import STLModel
    from esrecsys_amd.pinterest.train_shop_the_look import eval_step, train_step
    from esrecsys_amd.wikipedia.models import Glove
    from esrecsys_amd.wikipedia.train_cooccurence import apply_model
    V, D, B = 500, 32, 64
    stl = STLModel(output_size=D, num_scenes=V, num_products=V, device=dev)
    st = TrainState.create(apply_fn=stl.apply, params=stl.init(0), tx=optim.sparse_adagrad(0.05))
    good = torch.arange(B, dtype=torch.int32, device=dev)
    for bad_value, pos in ((V, 17), (-1, 0), (2 ** 31 - 1, B - 1)):
        bad = good.clone()
        bad[pos] = bad_value
        monkeypatch.setenv("ESR_CHECK_IDS", "1")
        with pytest.raises(IndexError, match=r"out of range \[0, %d\); first at flat position %d" % (V, pos)):
            train_step(st, good, bad, good, 0.1, B)
        with pytest.raises(IndexError):
            train_step(st, bad, good, None, 0.1, B, scale=2.0)
        with pytest.raises(IndexError):
            eval_step(st, good, good, bad)
    # the screen itself: counts every offender, reports the first
    ids = torch.tensor([3, 700, 5, -4, 499, 500], dtype=torch.int32, device=dev)
    with pytest.raises(IndexError, match=r"^3 device-resident id\(s\) out of range \[0, 500\); first at flat position 1 "):
        ops.check_device_ids(ids, V)
    ops.check_device_ids(good, V)                       # in-range ids pass
    ops.check_device_ids(good[:0], V)                   # empty list passes
    model = Glove(num_embeddings=V, features=D, device=dev)
    gs = TrainState.create(apply_fn=model.apply, params=model.init(1, None)["params"], tx=optim.sparse_adagrad(0.05))
    inputs = torch.stack([good, good]).contiguous()
    inputs[1, 9] = V + 3
    with pytest.raises(IndexError):
        apply_model(gs, inputs, torch.ones(B, device=dev))
    # a state that saw only rejected batches is untouched
    assert int(st.step) == 0
    monkeypatch.delenv("ESR_CHECK_IDS")
    st2, _ = train_step(st, good, good, good, 0.1, B)   # default: no screen, no sync, in-range ids just run
    assert int(st2.step) == 1
