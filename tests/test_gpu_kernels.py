"""GPU parity tests: every HIP kernel, through the C ABI, against the CPU oracle / golden fixtures.

Bar (north_star): bit-exact for integer / index / gather work; <= 1e-5 relative (max-abs error over the
max-abs of the expected tensor) for fp32 loss and gradients.
"""
import numpy as np
import pytest
import torch

from conftest import elem_rel_err, load_golden, rel_err
from oracle import glove as o_glove
from oracle import optim as o_optim
from oracle import shard as o_shard
from oracle import stl_head as o_stl
from oracle import topk as o_topk

pytestmark = pytest.mark.gpu
TOL = 1e-5  # fp32 loss / grad tolerance stated by north_star
F64 = np.float64
F32 = np.float32


def T(x, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return t.to(dtype) if dtype is not None else t


def N(t):
    return t.detach().float().cpu().numpy() if t.dtype == torch.bfloat16 else t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
# gather (bit-exact)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [1, 3, 32, 64, 96, 128, 256, 512])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_rows_bit_exact(dev, D, dtype):
    from esrecsys_amd import ops
    g = torch.Generator().manual_seed(D)
    V, n = 5000, 777  # n not a multiple of 64
    table = torch.randn((V, D), generator=g).to(dtype).to(dev)
    ids = torch.randint(0, V, (n,), generator=g, dtype=torch.int32)
    ids[0], ids[1], ids[2], ids[3] = 0, V - 1, 5, 5  # edge ids + a duplicate
    out = ops.gather_rows(table, ids.to(dev))
    exp = table.cpu()[ids.long()]
    assert torch.equal(out.cpu().view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                       exp.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))


def test_gather_rows_empty_and_single(dev):
    from esrecsys_amd import ops
    table = torch.randn((10, 8), device=dev)
    assert ops.gather_rows(table, torch.zeros(0, dtype=torch.int32, device=dev)).shape == (0, 8)
    out = ops.gather_rows(table, torch.tensor([9], dtype=torch.int32, device=dev))
    assert torch.equal(out[0], table[9])


def test_gather_full_size_checksum(dev):
    """BASELINE config C2 shape: 1M x 128 fp32 table, 2^20 ids; checksum of rows == checksum via torch indexing."""
    from esrecsys_amd import ops
    g = torch.Generator(device=dev).manual_seed(1701)
    V, D, n = 1_000_000, 128, 1 << 20
    table = torch.randn((V, D), generator=g, device=dev)
    ids = torch.randint(0, V, (n,), generator=g, device=dev, dtype=torch.int32)
    out = ops.gather_rows(table, ids)
    exp = table[ids.long()]
    assert torch.equal(out.view(torch.int32), exp.view(torch.int32))


def test_unpermute_rows(dev):
    from esrecsys_amd import ops
    g = torch.Generator().manual_seed(3)
    n, D = 1000, 128
    rows = torch.randn((n, D), generator=g).to(dev)
    perm = torch.randperm(n, generator=g).to(torch.int32).to(dev)
    out = ops.unpermute_rows(rows, perm)
    exp = torch.empty_like(rows)
    exp[perm.long()] = rows
    assert torch.equal(out, exp)


# ------------------------------------------------------------------------------------------------
# GloVe
# ------------------------------------------------------------------------------------------------
GLOVE_CASES = ["glove_uniform_d16_b64", "glove_uniform_d64_b128", "glove_same_d16_b64", "glove_zipf_d64_b128"]


def _dense_from_rows(V, D, ids, rows):
    out = np.zeros((V, D), F64)
    np.add.at(out, ids.astype(np.int64), rows.astype(F64).reshape(len(ids), D))
    return out


@pytest.mark.parametrize("case", GLOVE_CASES)
@pytest.mark.parametrize("mode", ["reference", "diagonal"])
def test_glove_fwd_bwd_vs_golden(dev, case, mode):
    from esrecsys_amd import ops
    g = load_golden(case)
    emb, bias = T(g["emb"], dev), T(g["bias"], dev)
    inputs, target = T(g["inputs"], dev), T(g["target"], dev)
    m = ops.GLOVE_REFERENCE if mode == "reference" else ops.GLOVE_DIAGONAL
    loss, grows, gbias = ops.glove_fwd_bwd(emb, bias, inputs, target, m)
    assert abs(float(loss) - g["loss_" + mode]) / abs(g["loss_" + mode]) <= TOL
    V, D = g["emb"].shape
    ids = g["inputs"].reshape(-1)
    assert rel_err(_dense_from_rows(V, D, ids, N(grows)), g["gemb_" + mode]) <= TOL
    assert rel_err(_dense_from_rows(V, 1, ids, N(gbias)), g["gbias_" + mode]) <= TOL
    # the dense gradient the reference materialises, through sort + segment-sum on the GPU
    sid, perm = ops.segment_sort(inputs.reshape(-1), V)
    dense = ops.rows_to_dense(V, D, sid, perm, grows)
    assert rel_err(N(dense), g["gemb_" + mode]) <= TOL


@pytest.mark.parametrize("case", GLOVE_CASES[:2])
def test_glove_forward_bb_output(dev, case):
    from esrecsys_amd import ops
    g = load_golden(case)
    dot, s = ops.glove_forward(T(g["emb"], dev), T(g["bias"], dev), T(g["inputs"], dev))
    pred = N(dot)[None, :] + N(s)[:, None]
    assert pred.shape == g["pred_reference"].shape
    assert rel_err(pred, g["pred_reference"]) <= TOL


@pytest.mark.parametrize("D,B", [(4, 7), (100, 65), (256, 2048), (512, 300), (3, 50), (1024, 64)])
def test_glove_shapes_vs_oracle(dev, D, B):
    """D not a power of two / not a multiple of 4, B not a multiple of the block, duplicates."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(D * 1000 + B)
    V = 500
    emb = (rng.standard_normal((V, D)) / np.sqrt(D)).astype(np.float32)
    bias = (rng.standard_normal((V, 1)) * 0.1).astype(np.float32)
    inputs = rng.integers(0, V, (2, B)).astype(np.int32)
    inputs[:, : B // 3] = inputs[:, :1]
    target = np.exp(rng.uniform(np.log(0.1), np.log(1000), B)).astype(np.float32)
    for mode, m in (("reference", ops.GLOVE_REFERENCE), ("diagonal", ops.GLOVE_DIAGONAL)):
        loss, grows, gbias = ops.glove_fwd_bwd(T(emb, dev), T(bias, dev), T(inputs, dev), T(target, dev), m)
        el, gdot, gs = o_glove.loss_and_grads(emb.astype(F64), bias.astype(F64), inputs, target, mode, F64)
        _, erows, ebias = o_glove.row_grads(emb.astype(F64), inputs, gdot, gs, F64)
        assert abs(float(loss) - el) / abs(el) <= TOL
        assert rel_err(N(grows), erows) <= TOL
        assert rel_err(N(gbias), ebias) <= TOL


def test_glove_config_c3_size_properties(dev):
    """BASELINE config C3 shape (V=465537, D=256, B=65536): oracle fp64 on the same inputs (loss + a
    sample of gradient rows), plus linearity: grad rows are gdot * partner row exactly."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(1701)
    V, D, B = 400_000 + 65_537, 256, 65_536
    emb = (rng.standard_normal((V, D), dtype=np.float32) / np.sqrt(D)).astype(np.float32)
    bias = (rng.standard_normal((V, 1)) * 0.05).astype(np.float32)
    inputs = rng.integers(0, V, (2, B)).astype(np.int32)
    target = np.exp(rng.uniform(np.log(0.1), np.log(1000), B)).astype(np.float32)
    loss, grows, gbias = ops.glove_fwd_bwd(T(emb, dev), T(bias, dev), T(inputs, dev), T(target, dev),
                                           ops.GLOVE_REFERENCE)
    el, gdot, gs = o_glove.loss_and_grads(emb, bias, inputs, target, "reference", F64)
    assert abs(float(loss) - el) / abs(el) <= TOL
    sel = rng.integers(0, B, 512)
    exp1 = gdot[sel, None] * emb[inputs[1, sel]].astype(F64)
    assert rel_err(N(grows[torch.from_numpy(sel).to(dev)]), exp1) <= TOL
    assert rel_err(N(gbias)[:B], gs) <= TOL


# ------------------------------------------------------------------------------------------------
# STL triplet head
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["stl_b32_d8_lam0", "stl_b32_d8_lam01", "stl_b128_d32_lam01"])
def test_triplet_head_vs_golden(dev, case):
    from esrecsys_amd import ops
    g = load_golden(case)
    s, p, n = T(g["scene"], dev), T(g["pos"], dev), T(g["neg"], dev)
    B = s.shape[0]
    loss, ps, ns, gs, gp, gn = ops.triplet_fwd_bwd(s, p, n, None, None, None, B, float(g["lam"]),
                                                   float(g["batch_size"]))
    assert abs(float(loss) - g["loss"]) / abs(g["loss"]) <= TOL
    assert rel_err(N(ps), g["pos_score"]) <= TOL and rel_err(N(ns), g["neg_score"]) <= TOL
    assert rel_err(N(gs), g["g_scene"]) <= TOL
    assert rel_err(N(gp), g["g_pos"]) <= TOL
    assert rel_err(N(gn), g["g_neg"]) <= TOL
    # eval_step semantics: plain hinge sum, no reg, not divided (train_shop_the_look.py:118)
    ev, _, _, _, _, _ = ops.triplet_fwd_bwd(s, p, n, None, None, None, B, 0.0, 1.0, with_reg=False,
                                            want_grads=False)
    assert abs(float(ev) - g["eval_loss"]) / abs(g["eval_loss"]) <= TOL


@pytest.mark.parametrize("D,B", [(32, 128), (128, 8192), (96, 1000), (64, 33)])
def test_triplet_with_id_towers_vs_oracle(dev, D, B):
    from esrecsys_amd import ops
    rng = np.random.default_rng(D + B)
    Vs, Vp = 3000, 10_000
    st = (rng.standard_normal((Vs, D)) * rng.uniform(0.02, 0.25, (Vs, 1))).astype(np.float32)
    pt = (rng.standard_normal((Vp, D)) * rng.uniform(0.02, 0.25, (Vp, 1))).astype(np.float32)
    sid = rng.integers(0, Vs, B).astype(np.int32)
    pid = rng.integers(0, Vp, B).astype(np.int32)
    nid = rng.integers(0, Vp, B).astype(np.int32)
    sid[:5], pid[:5], nid[:5] = 0, Vp - 1, Vp - 1
    loss, ps, ns, gs, gp, gn = ops.triplet_fwd_bwd(T(st, dev), T(pt, dev), T(pt, dev), T(sid, dev), T(pid, dev),
                                                   T(nid, dev), B, 0.1, B)
    el, egs, egp, egn = o_stl.triplet_loss_and_grads(st[sid], pt[pid], pt[nid], 0.1, B, F64)
    assert abs(float(loss) - el) / abs(el) <= TOL
    assert rel_err(N(gs), egs) <= TOL and rel_err(N(gp), egp) <= TOL and rel_err(N(gn), egn) <= TOL
    assert gs._base is gn._base and gs._base.shape == (3 * B, D)  # [scene ; pos ; neg] in one buffer


# ------------------------------------------------------------------------------------------------
# in-batch softmax (FP32 MFMA)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["inbatch_b64_d32", "inbatch_b96_d64_scale4", "inbatch_b320_d128"])
def test_inbatch_softmax_vs_golden(dev, case):
    from esrecsys_amd import ops
    g = load_golden(case)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(g["q"], dev), T(g["c"], dev), float(g["scale"]),
                                                    float(g["lam"]), float(g["batch_size"]))
    assert abs(float(loss) - g["loss"]) / abs(g["loss"]) <= TOL
    assert rel_err(N(lse), g["lse"]) <= TOL
    assert rel_err(N(gq), g["g_q"]) <= TOL
    assert rel_err(N(gc), g["g_c"]) <= TOL


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_inbatch_transpose_detecting(dev, precision):
    """Asymmetric operands: a swapped Q/C role or a transposed MFMA fragment cannot pass."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(77)
    B, D = 256, 128
    q = (rng.standard_normal((B, D)) * 0.3).astype(np.float32)
    c = (rng.standard_normal((B, D)) * 0.05 + np.linspace(-0.2, 0.2, D)[None, :]).astype(np.float32)
    q[:, :7] *= 4.0
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 3.0, 0.2, B, precision=precision)
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.2, B, 3.0, F64)
    assert abs(float(loss) - el) / abs(el) <= TOL
    assert rel_err(N(lse), else_) <= TOL and rel_err(N(gq), egq) <= TOL and rel_err(N(gc), egc) <= TOL


@pytest.mark.parametrize("B,D", [(1, 32), (2, 128), (31, 64), (33, 128), (100, 32), (1000, 128), (4099, 64)])
def test_inbatch_ragged_batch_vs_oracle(dev, B, D):
    """batch sizes that are not a multiple of the 32-row tile: the ragged last tile is masked, not padded"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(B)
    q = (rng.standard_normal((B, D)) * 0.2).astype(np.float32)
    c = (rng.standard_normal((B, D)) * 0.2).astype(np.float32)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 3.0, 0.2, float(B))
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.2, B, 3.0, F64)
    assert abs(float(loss) - el) <= TOL * max(abs(el), 1e-3)
    assert rel_err(N(lse), else_) <= TOL
    assert np.abs(N(gq) - egq).max() <= TOL * max(np.abs(egq).max(), 1e-6)
    assert np.abs(N(gc) - egc).max() <= TOL * max(np.abs(egc).max(), 1e-6)


@pytest.mark.parametrize("D", [4, 20, 32, 48, 64, 96, 100, 128, 192, 256, 384, 512])
@pytest.mark.parametrize("B", [96, 1000])
def test_inbatch_embedding_widths_vs_oracle(dev, B, D):
    """every embedding width the reference trains (output_size 32 / 64 / 96: pinterest/sweep.yaml:13-14, README 64) and
    the wide ones of SURVEY 4.3 (256, 512): D <= 128 runs in the next tile width with zero columns that never touch
    memory, 128 < D <= 512 in the column-panel kernel.  Norm-wise AND element-wise bounds against the fp64 oracle."""
    from conftest import elem_rel_err
    from esrecsys_amd import ops
    rng = np.random.default_rng(1000 * B + D)
    q = (rng.standard_normal((B, D)) * (1.1 / np.sqrt(D))).astype(np.float32)   # |row| ~ 1.1: the regulariser is live
    c = (rng.standard_normal((B, D)) * (1.1 / np.sqrt(D))).astype(np.float32)
    c[3] = c[4]
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 5.0, 0.2, float(B))
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.2, B, 5.0, F64)
    assert abs(float(loss) - el) <= TOL * abs(el)
    assert rel_err(N(lse), else_) <= TOL and rel_err(N(gq), egq) <= TOL and rel_err(N(gc), egc) <= TOL
    assert elem_rel_err(N(gq), egq) <= 20 * TOL and elem_rel_err(N(gc), egc) <= 20 * TOL
    # columns beyond D do not exist: the output buffers end where the rows end (a guard row behind them stays intact)
    guard = torch.full((2 * B + 1, D), 7.0, device=dev)
    qt, ct = T(q, dev), T(c, dev)
    loss2, _, gq2, gc2 = ops.inbatch_softmax_fwd_bwd(qt, ct, 5.0, 0.2, float(B))
    assert torch.equal(gq2, gq) and torch.equal(gc2, gc) and bool((guard == 7.0).all())


@pytest.mark.parametrize("precision", ["f16x2", "bf16x3", "auto"])
@pytest.mark.parametrize("D", [32, 64, 96, 100, 124])
def test_inbatch_split_paths_on_narrow_rows(dev, D, precision):
    """the reference's own embedding widths (output_size 32 / 64 / 96: pinterest/sweep.yaml:13-14, README 64) on the
    split-precision MFMA paths: rows narrower than the 128-column tile are zero-padded inside the split kernels, the
    gradient rows come back D wide.  Dense entry and tower entry (gather folded in), against the fp64 oracle."""
    from conftest import elem_rel_err
    from esrecsys_amd import ops
    B, V = 1024, 5000
    rng = np.random.default_rng(D)
    qt = (rng.standard_normal((V, D)) * (1.1 / np.sqrt(D))).astype(np.float32)
    ct = (rng.standard_normal((V, D)) * (1.1 / np.sqrt(D))).astype(np.float32)
    qi, ci = rng.integers(0, V, B).astype(np.int32), rng.integers(0, V, B).astype(np.int32)
    q, c = qt[qi], ct[ci]
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(F64), c.astype(F64), 0.2, float(B), 6.0, F64)
    guard = torch.full((2 * B + 1, D), 7.0, device=dev)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 6.0, 0.2, float(B), precision=precision)
    assert gq.shape == (B, D) and gc.shape == (B, D) and bool((guard == 7.0).all())
    assert abs(float(loss) - el) <= TOL * abs(el) and rel_err(N(lse), else_) <= TOL
    assert rel_err(N(gq), egq) <= TOL and rel_err(N(gc), egc) <= TOL
    assert elem_rel_err(N(gq), egq) <= 20 * TOL and elem_rel_err(N(gc), egc) <= 20 * TOL
    if precision != "auto":
        l2, lse2, gq2, gc2 = ops.inbatch_towers_fwd_bwd(T(qt, dev), T(ct, dev), T(qi, dev), T(ci, dev), 6.0, 0.2, float(B),
                                                        precision=precision)
        assert torch.equal(l2, loss) and torch.equal(gq2, gq) and torch.equal(gc2, gc)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_inbatch_gradients_elementwise_at_c2_size(dev, precision):
    """C2 size, both MFMA paths: every gradient entry within 1e-4 of the fp64 oracle relative to max(|entry|, 1e-3 of
    the largest entry) -- the norm-wise 1e-5 bound alone would let small entries of the bf16x3 path be arbitrarily bad."""
    from conftest import elem_rel_err
    from esrecsys_amd import ops
    rng = np.random.default_rng(77)
    B, D = 8192, 128
    q = (rng.standard_normal((B, D)) * (1.2 / np.sqrt(D))).astype(np.float32)
    c = (rng.standard_normal((B, D)) * (1.2 / np.sqrt(D))).astype(np.float32)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 8.0, 0.1, float(B), precision=precision)
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.1, B, 8.0, F64)
    eq, ec = elem_rel_err(N(gq), egq), elem_rel_err(N(gc), egc)
    print("inbatch %s element-wise rel.err (floor 1e-3 max): gQ %.2e gC %.2e; norm-wise %.2e %.2e"
          % (precision, eq, ec, rel_err(N(gq), egq), rel_err(N(gc), egc)))
    assert eq <= 1e-4 and ec <= 1e-4   # measured: f32 2.5e-5 / 3.3e-5, bf16x3 5.2e-5 / 4.7e-5
    assert elem_rel_err(N(lse), else_, floor_frac=1e-6) <= TOL


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_inbatch_rows_of_mixed_norms_inside_one_matrix(dev, precision):
    """Trained-table-like operands: row norms spread over four decades INSIDE each matrix (log-uniform 1e-4 .. 1) with a
    few hot rows 100 x the typical one.  The fp16 x 2 path keeps ONE power-of-two exponent per matrix, so elements below
    2^-16 of the matrix maximum carry an absolute error; what matters for training is the gradient error per ROW BLOCK
    relative to that block's own largest entry.  Asserted per 128-row block, against the fp64 oracle, for all three
    paths (measured values are printed: the bound the f16x2 path is held to is the one the exact-f32 MFMA path meets)."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(5)
    B, D = 2048, 128

    def rows():
        x = rng.standard_normal((B, D))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        norms = 10.0 ** rng.uniform(-4, 0, B)
        norms[rng.choice(B, 16, replace=False)] = 3.0      # hot rows: 100 x the median
        return (x * norms[:, None]).astype(np.float32)
    q, c = rows(), rows()
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 8.0, 0.1, float(B), precision=precision)
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.1, B, 8.0, F64)
    assert abs(float(loss) - el) <= TOL * abs(el)
    worst = 0.0
    for got, exp in ((N(gq), egq), (N(gc), egc)):
        for b0 in range(0, B, 128):
            g, e = got[b0:b0 + 128].astype(F64), exp[b0:b0 + 128]
            worst = max(worst, float(np.abs(g - e).max() / np.abs(e).max()))
    row_worst = max(float((np.abs(got.astype(F64) - exp).max(1) / np.maximum(np.abs(exp).max(1), 1e-30)).max())
                    for got, exp in ((N(gq), egq), (N(gc), egc)))
    print("inbatch %s, mixed norms: worst block error %.2e (relative to the block's largest entry), worst row %.2e"
          % (precision, worst, row_worst))
    assert worst <= 2e-5 and row_worst <= 1e-4
    assert elem_rel_err(N(lse), else_, floor_frac=1e-6) <= TOL


def test_inbatch_trajectory_at_c2_size_vs_fp64_oracle(dev):
    """20 in-batch steps at C2 size (two 1 M x 128 fp32 towers, B = 8192, scale 8, sparse Adagrad) on the default
    (fp16 x 2) score path against the fp64 oracle stepping the same rows: every loss within 1e-5, and the rows the steps
    touched -- tables and accumulators -- within 1e-5 norm-wise at the end.  Ids are drawn from a 20 000-row window so
    that rows are revisited (errors would compound), as the hot part of a real id stream is."""
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.pinterest.models import STLModel
    from esrecsys_amd.pinterest.train_shop_the_look import train_steps
    from oracle import optim as o_optim
    V, D, B, steps, lam, lr, scale, W = 1_000_000, 128, 8192, 20, 0.1, 0.05, 8.0, 20_000
    g = torch.Generator(device=dev).manual_seed(1701)
    st = torch.randn((V, D), generator=g, device=dev) * D ** -0.5
    pt = torch.randn((V, D), generator=g, device=dev) * D ** -0.5
    rng = np.random.default_rng(3)
    base_s, base_p = 123_456, 654_321
    es, ep = st[base_s:base_s + W].double().cpu().numpy(), pt[base_p:base_p + W].double().cpu().numpy()
    a_s, a_p = np.full_like(es, 0.1), np.full_like(ep, 0.1)
    model = STLModel(output_size=D, num_scenes=V, num_products=V, device=dev)
    state = TrainState.create(apply_fn=model.apply, tx=optim.sparse_adagrad(lr),
                              params={"params": {"scene_tower": {"embedding": st}, "product_tower": {"embedding": pt}}})
    batches = [(rng.integers(0, W, B).astype(np.int32), rng.integers(0, W, B).astype(np.int32)) for _ in range(steps)]
    dev_batches = [(T(a + base_s, dev), T(b + base_p, dev), None) for a, b in batches]
    state, losses = train_steps(state, iter(dev_batches), steps, lam, float(B), scale=scale)
    losses = losses.cpu().numpy()
    for k, (sid, pid) in enumerate(batches):
        el, _, gq, gc = o_stl.inbatch_softmax_loss_and_grads(es[sid], ep[pid], lam, B, scale, F64)
        assert abs(float(losses[k]) - el) <= TOL * abs(el), (k, float(losses[k]), el)
        es, a_s = o_optim.sparse_adagrad_update(es, a_s, sid, gq, lr, dtype=F64)
        ep, a_p = o_optim.sparse_adagrad_update(ep, a_p, pid, gc, lr, dtype=F64)
    p = state.params["params"]
    acc = state.opt_state["sum_of_squares"]["params"]
    got = [p["scene_tower"]["embedding"][base_s:base_s + W], p["product_tower"]["embedding"][base_p:base_p + W],
           acc["scene_tower"]["embedding"][base_s:base_s + W], acc["product_tower"]["embedding"][base_p:base_p + W]]
    errs = [rel_err(N(a), b) for a, b in zip(got, (es, ep, a_s, a_p))]
    print("in-batch 20-step trajectory at C2 size: tables / accumulators norm-wise error", ["%.2e" % e for e in errs])
    assert max(errs) <= TOL
    # rows outside the window were never touched: bit-identical to the initial draw
    g2 = torch.Generator(device=dev).manual_seed(1701)
    st0 = torch.randn((V, D), generator=g2, device=dev) * D ** -0.5
    assert torch.equal(p["scene_tower"]["embedding"][:base_s], st0[:base_s])


def test_fused_heads_write_grads_at_ids(dev):
    """ESR_GRADS_AT_IDS: with a private [n, D] copy of the looked-up rows in any order and ids = positions in it,
    the triplet / GloVe heads emit their gradient rows at those positions (what the row-sharded step feeds the
    gradient exchange) -- equal to the plain call after the same permutation, bit for bit."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(33)
    B, D = 1000, 128
    rows = T(rng.standard_normal((3 * B, D)).astype(np.float32) * 0.3, dev)
    inv = T(rng.permutation(3 * B).astype(np.int32), dev)               # occurrence o reads row inv[o]
    plain = ops.triplet_fwd_bwd(rows, rows, rows, inv[:B], inv[B:2 * B], inv[2 * B:], B, 0.1, float(B),
                                want_scores=False)
    at = ops.triplet_fwd_bwd(rows, rows, rows, inv[:B], inv[B:2 * B], inv[2 * B:], B, 0.1, float(B),
                             want_scores=False, grads_at_ids=True)
    assert torch.equal(plain[0], at[0]) and at[4] is None
    assert torch.equal(at[3][inv.long()], plain[3]._base)
    emb = T(rng.standard_normal((2 * B, D)).astype(np.float32) * 0.3, dev)
    bias = T(rng.standard_normal((2 * B, 1)).astype(np.float32) * 0.05, dev)
    inv2 = T(rng.permutation(2 * B).astype(np.int32).reshape(2, B), dev)
    tgt = T(rng.uniform(0.1, 300, B).astype(np.float32), dev)
    for mode in (ops.GLOVE_REFERENCE, ops.GLOVE_DIAGONAL):
        l0, g0, b0 = ops.glove_fwd_bwd(emb, bias, inv2, tgt, mode)
        l1, g1, b1 = ops.glove_fwd_bwd(emb, bias, inv2, tgt, mode, grads_at_ids=True)
        idx = inv2.reshape(-1).long()
        assert torch.equal(l0, l1) and torch.equal(g1[idx], g0) and torch.equal(b1[idx], b0)


@pytest.mark.parametrize("dtype,precision", [(torch.float32, "bf16x3"), (torch.float32, "f16x2"),
                                             (torch.bfloat16, "bf16x3"), (torch.bfloat16, "auto")])
def test_inbatch_towers_gather_folded_in(dev, dtype, precision):
    """the step head that reads the tower rows itself == gather + dense head, bit for bit; and vs the oracle"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(21)
    Vq, Vc, D, B = 3000, 7000, 128, 640
    qt = T((rng.standard_normal((Vq, D)) * 0.1).astype(np.float32), dev, dtype)
    ct = T((rng.standard_normal((Vc, D)) * 0.1).astype(np.float32), dev, dtype)
    qi = rng.integers(0, Vq, B).astype(np.int32)
    ci = rng.integers(0, Vc, B).astype(np.int32)
    ci[3] = ci[2]                                   # the same candidate row twice in the batch
    loss, lse, gq, gc = ops.inbatch_towers_fwd_bwd(qt, ct, T(qi, dev), T(ci, dev), 6.0, 0.1, float(B),
                                                   precision=precision)
    q = ops.unpermute_rows_to_f32(ops.gather_rows(qt, T(qi, dev)), None)
    c = ops.unpermute_rows_to_f32(ops.gather_rows(ct, T(ci, dev)), None)
    l2, lse2, gq2, gc2 = ops.inbatch_softmax_fwd_bwd(q, c, 6.0, 0.1, float(B),
                                                     precision="f16x2" if precision == "auto" else precision)
    # ("auto" on bf16 towers: the fp16 ONE-plane kernels -- the dense head on the same bf16-valued rows runs the two-plane
    # ones, whose second operand planes are all zero: the same products in the same order)
    assert torch.equal(loss, l2) and torch.equal(lse, lse2) and torch.equal(gq, gq2)
    # pass C: bf16 towers recompute S^T (one-plane kernels), f32 rows take the stored-P kernel, which normalises p / l
    # instead of forming exp2(s - lse): the same probabilities to an f32 rounding
    assert torch.equal(gc, gc2) if dtype == torch.float32 else rel_err(N(gc), N(gc2)) <= 1e-6
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(N(q).astype(F64), N(c).astype(F64), 0.1, B, 6.0, F64)
    assert abs(float(loss) - el) / abs(el) <= TOL and rel_err(N(gq), egq) <= TOL and rel_err(N(gc), egc) <= TOL


@pytest.mark.parametrize("B", [128, 512, 1024, 2176, 8192])
def test_inbatch_one_plane_kernels_equal_the_full_ones(dev, B):
    """bf16 towers take the one-plane kernels (8 live cross terms of 24): same bits as the full kernels on the same
    (bf16-valued) rows, at chunk counts per split of 1, 2, 4, 17 and 64 -- every arm of the pipelined loop"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(B)
    V, D = 5000, 128
    qt = T((rng.standard_normal((V, D)) * 0.12).astype(np.float32), dev, torch.bfloat16)
    ct = T((rng.standard_normal((V, D)) * 0.12).astype(np.float32), dev, torch.bfloat16)
    qi = T(rng.integers(0, V, B).astype(np.int32), dev)
    ci = T(rng.integers(0, V, B).astype(np.int32), dev)
    loss, lse, gq, gc = ops.inbatch_towers_fwd_bwd(qt, ct, qi, ci, 7.0, 0.1, float(B), precision="bf16x3")
    l2, lse2, gq2, gc2 = ops.inbatch_towers_fwd_bwd(qt.float(), ct.float(), qi, ci, 7.0, 0.1, float(B),
                                                    precision="bf16x3")
    assert torch.equal(loss, l2) and torch.equal(lse, lse2) and torch.equal(gq, gq2)
    assert rel_err(N(gc), N(gc2)) <= 1e-6   # (pass C of the full path reads stored probabilities; see above)
    if B <= 2176:
        q, c = N(qt.float())[N(qi)].astype(F64), N(ct.float())[N(ci)].astype(F64)
        el, _, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.1, B, 7.0, F64)
        assert abs(float(loss) - el) / abs(el) <= TOL and rel_err(N(gq), egq) <= TOL and rel_err(N(gc), egc) <= TOL


@pytest.mark.parametrize("B,hot", [(128, False), (384, False), (512, True), (1024, False), (2176, True), (8192, False),
                                   (8192, True), (16384, False)])
def test_inbatch_bf16_tables_on_one_fp16_plane(dev, B, hot, monkeypatch):
    """bf16 towers (BASELINE config 4's dtype) on the fp16 entry points (round 5, the default for them): ONE fp16 plane
    per operand -- a bf16 element is exact in it -- two for the probabilities, S^T recomputed by pass C: six GEMMs, no
    stored probabilities (inbatch1h_kernel).  Against the fp64 oracle on the same (bf16-valued) rows at the 1e-5 bound --
    with row norms spread over four decades and a few hot rows ("hot"), every 128-row block within 2e-5 of its own
    largest entry -- against the two-plane kernels on the same rows (whose second operand planes are all zero: the same
    products, pass C from stored probabilities instead), and with every workgroup forced through its redo."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(B + hot)
    V, D = 6000, 128

    def table():
        x = rng.standard_normal((V, D))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        # (not exactly 1: bf16-rounded unit rows would sit within an f32 rounding of the regulariser's kink at |x| = 1)
        norms = 10.0 ** rng.uniform(-4, 0, V) if hot else np.where(rng.random(V) < 0.5, 0.8, 1.3)
        if hot:
            norms[rng.choice(V, 16, replace=False)] = 3.0
        return T((x * norms[:, None]).astype(np.float32), dev, torch.bfloat16)
    qt, ct = table(), table()
    qi = T(rng.integers(0, V, B).astype(np.int32), dev)
    ci = T(rng.integers(0, V, B).astype(np.int32), dev)
    assert ops.inbatch_split_path("auto", B, D, bf16_tables=True) == "f16x2"
    out = [t.clone() for t in ops.inbatch_towers_fwd_bwd(qt, ct, qi, ci, 8.0, 0.1, float(B))]
    assert all(bool(torch.isfinite(t).all()) for t in out)
    q, c = N(qt.float())[N(qi)].astype(F64), N(ct.float())[N(ci)].astype(F64)
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.1, B, 8.0, F64)
    assert abs(float(out[0]) - el) <= TOL * abs(el) and rel_err(N(out[1]), else_) <= TOL
    assert rel_err(N(out[2]), egq) <= TOL and rel_err(N(out[3]), egc) <= TOL
    worst = 0.0
    for got, exp in ((N(out[2]), egq), (N(out[3]), egc)):
        for b0 in range(0, B, 128):
            g, e = got[b0:b0 + 128].astype(F64), exp[b0:b0 + 128]
            worst = max(worst, float(np.abs(g - e).max() / np.abs(e).max()))
    print("bf16 towers, one fp16 plane, B = %d%s: worst 128-row block error %.2e" % (B, " (mixed norms)" if hot else "", worst))
    assert worst <= 2e-5
    monkeypatch.setenv("ESR_IB2H_BF16", "two")   # two planes per operand (the second all zero), stored probabilities
    two = ops.inbatch_towers_fwd_bwd(qt, ct, qi, ci, 8.0, 0.1, float(B), precision="f16x2")
    monkeypatch.delenv("ESR_IB2H_BF16")
    if hot:
        # a workgroup whose optimistic exponent reference overflows redoes its rows against their exact maxima: the
        # one-plane kernel's workgroups own 256 rows, the two-plane kernel's 128 -- other rows take the redo, whose
        # probabilities are rounded on another grid
        assert abs(float(out[0]) - float(two[0])) <= 1e-6 * abs(float(two[0]))
        assert rel_err(N(out[1]), N(two[1])) <= 1e-6 and rel_err(N(out[2]), N(two[2])) <= 1e-6
    else:
        assert torch.equal(out[0], two[0]) and torch.equal(out[1], two[1]) and torch.equal(out[2], two[2])
    assert rel_err(N(out[3]), N(two[3])) <= 1e-6
    monkeypatch.setenv("ESR_IB2H_REF", "redo")   # every pass-Q workgroup redoes itself against its exact maximum
    redo = ops.inbatch_towers_fwd_bwd(qt, ct, qi, ci, 8.0, 0.1, float(B))
    monkeypatch.delenv("ESR_IB2H_REF")
    assert abs(float(redo[0]) - el) <= TOL * abs(el) and rel_err(N(redo[2]), egq) <= TOL and rel_err(N(redo[3]), egc) <= TOL
    again = ops.inbatch_towers_fwd_bwd(qt, ct, qi, ci, 8.0, 0.1, float(B))
    assert all(torch.equal(a, b) for a, b in zip(out, again))   # repeatable bit for bit
    monkeypatch.setenv("ESR_IB1H_DBG", "1")      # test hook: the same phases one chunk at a time, nothing pipelined
    plain = ops.inbatch_towers_fwd_bwd(qt, ct, qi, ci, 8.0, 0.1, float(B))
    monkeypatch.delenv("ESR_IB1H_DBG")
    assert all(torch.equal(a, b) for a, b in zip(out, plain))


@pytest.mark.parametrize("B", [128, 256, 1024, 2176, 8192, 16384])
def test_inbatch_pstore_matches_recompute(dev, B, monkeypatch):
    """fp32 towers: pass C reading the probabilities pass Q stored (default) against pass C recomputing S^T
    (ESR_IB3_PSTORE=0).  Loss, lse and gQ come from the same pass-Q arithmetic (bit-equal); gC differs by f32 roundings
    of p (exp2(s - ref) / l against exp2(s - lse)).  B = 16384 is the largest stored-P batch."""
    from conftest import elem_rel_err
    from esrecsys_amd import ops
    rng = np.random.default_rng(B + 7)
    V, D = 20000, 128
    qt = T((rng.standard_normal((V, D)) * 0.12).astype(np.float32), dev)
    ct = T((rng.standard_normal((V, D)) * 0.12).astype(np.float32), dev)
    qi = T(rng.integers(0, V, B).astype(np.int32), dev)
    ci = T(rng.integers(0, V, B).astype(np.int32), dev)
    monkeypatch.setenv("ESR_IB3_PSTORE", "1")
    loss, lse, gq, gc = [x.clone() for x in ops.inbatch_towers_fwd_bwd(qt, ct, qi, ci, 7.0, 0.1, float(B),
                                                                       precision="bf16x3")]
    monkeypatch.setenv("ESR_IB3_PSTORE", "0")
    l2, lse2, gq2, gc2 = ops.inbatch_towers_fwd_bwd(qt, ct, qi, ci, 7.0, 0.1, float(B), precision="bf16x3")
    assert torch.equal(loss, l2) and torch.equal(lse, lse2) and torch.equal(gq, gq2)
    # element-wise with the floor at 1e-3 of the largest entry: 1e-6 norm-wise allows up to 1e-3 there
    assert rel_err(N(gc), N(gc2)) <= 1e-6 and elem_rel_err(N(gc), N(gc2)) <= 5e-4
    if B <= 2176:
        q, c = N(qt)[N(qi)].astype(F64), N(ct)[N(ci)].astype(F64)
        el, _, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.1, B, 7.0, F64)
        for g in (gc, gc2):
            assert rel_err(N(g), egc) <= TOL


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_inbatch_config_c2_full_size(dev, precision):
    """BASELINE config C2: B = 8192, D = 128, fp64 oracle on the same inputs + checksum properties
    (softmax rows sum to one => column sum of gC vanishes when reg = 0)."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(1701)
    B, D = 8192, 128
    q = (rng.standard_normal((B, D)) / np.sqrt(D)).astype(np.float32)
    c = (rng.standard_normal((B, D)) / np.sqrt(D)).astype(np.float32)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 8.0, 0.0, B, precision=precision)
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.0, B, 8.0, F64)
    assert abs(float(loss) - el) / abs(el) <= TOL
    assert rel_err(N(lse), else_) <= TOL
    assert rel_err(N(gq), egq) <= TOL and rel_err(N(gc), egc) <= TOL
    colsum = N(gc).astype(F64).sum(0)
    assert np.abs(colsum).max() <= 1e-5 * np.abs(N(gc)).sum(0).max()


@pytest.mark.parametrize("B", [128, 384, 1024, 2176])
def test_inbatch_bf16x3_shapes_and_hard_inputs(dev, B):
    """bf16x3 path on split counts 1..8, wide score range (scale 12: |S| up to ~25), big-norm rows, duplicate
    rows; also reports how close each precision is to the fp64 oracle (both must hold the 1e-5 bound; at
    much larger |S| fp32 itself leaves it: exp(x) amplifies the ulp(|x|) rounding of the score)."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(B)
    D = 128
    q = (rng.standard_normal((B, D)) * rng.uniform(0.02, 0.3, (B, 1))).astype(np.float32)
    c = (rng.standard_normal((B, D)) * rng.uniform(0.02, 0.3, (B, 1))).astype(np.float32)
    c[5] = c[6]          # duplicate candidate
    q[7] *= 3.0          # a dominant row: sharp softmax, forces online-max rescales late in the stream
    c[B - 1] = q[7]
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q, c, 0.1, B, 12.0, F64)
    errs = {}
    for precision in ("f32", "bf16x3", "f16x2"):
        loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 12.0, 0.1, B, precision=precision)
        errs[precision] = (abs(float(loss) - el) / abs(el), rel_err(N(lse), else_), rel_err(N(gq), egq),
                           rel_err(N(gc), egc))
    print("inbatch B=%d rel.err vs fp64 (loss, lse, gQ, gC): f32 %s | bf16x3 %s | f16x2 %s" % (
        B, " ".join("%.1e" % e for e in errs["f32"]), " ".join("%.1e" % e for e in errs["bf16x3"]),
        " ".join("%.1e" % e for e in errs["f16x2"])))
    for precision in errs:
        assert max(errs[precision]) <= TOL, (precision, errs)


@pytest.mark.parametrize("B,D", [(640, 100), (1024, 128), (256, 64)])
@pytest.mark.parametrize("scale", [-12.0, 12.0])
def test_inbatch_negative_temperature_with_a_far_out_candidate(dev, B, D, scale):
    """A negative temperature makes the row maximum of the scores the MINIMUM of the dot products; with one candidate
    far out on that side (score ~ +150 log2 units) an exponent reference taken from the wrong end overflows exp2 (found
    by scripts/inbatch_stress.py: the bf16 x 3 row-max pre-pass took max(dot) * scale -- inf / nan in that row).  Every
    precision, and bf16 tables (the fp16 one-plane kernels by default), must hold the bound."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(B + D)
    q = (rng.standard_normal((B, D)) * 2.97 / np.sqrt(D)).astype(np.float32)
    c = (rng.standard_normal((B, D)) * 0.99 / np.sqrt(D)).astype(np.float32)
    c[B - 5] = (3.0 * 0.99) * q[7] / np.linalg.norm(q[7]) * (1 if scale > 0 else -1)
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(F64), c.astype(F64), 0.1, 77.0, scale, F64)
    for precision in ("f32", "bf16x3", "f16x2"):
        if precision != "f32" and ops.inbatch_split_path(precision, B, D, bf16_tables=False) is None:
            continue
        loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), scale, 0.1, 77.0, precision=precision)
        errs = (abs(float(loss) - el) / abs(el), rel_err(N(lse), else_), rel_err(N(gq), egq), rel_err(N(gc), egc))
        assert np.all(np.isfinite(errs)) and max(errs) <= TOL, (precision, errs)


@pytest.mark.parametrize("mag_q,mag_c,scale", [(1e-4, 1e-4, 5e6), (3e-3, 40.0, 4.0), (200.0, 150.0, 2e-4),
                                               (0.09, 0.09, 8.0), (0.09, 0.09, 40.0), (0.3, 0.3, 30.0)])
def test_inbatch_f16x2_operand_and_score_ranges(dev, mag_q, mag_c, scale):
    """fp16's range is what the two-plane path has to manage: element magnitudes from 1e-4 to 200 (per-matrix
    power-of-two scale), score ranges on both sides of the bound that decides between "Cauchy-Schwarz bound as exponent
    reference" (<= 14 log2 units) and the true row maximum, rows whose scores are ALL far below zero (anti-aligned
    with every candidate: the probabilities must not underflow fp16), a dominant pair and exact zeros."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(int(scale * 7) % 1000 + 1)
    B, D = 1024, 128
    q = (rng.standard_normal((B, D)) * mag_q).astype(np.float32)
    c = (rng.standard_normal((B, D)) * mag_c).astype(np.float32)
    common = rng.standard_normal(D).astype(np.float32) * mag_c
    c[: B // 2] += 2.0 * common                  # half of the candidates share a direction ...
    q[11] = -3.0 * (mag_q / mag_c) * common      # ... and this query opposes it (and is small against the rest)
    q[12] = 0.0                                  # exact zeros
    c[13, ::2] = 0.0
    c[B - 2] = q[17] * (mag_c / mag_q) * 2.0     # a dominant pair
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(F64), c.astype(F64), 0.1, float(B), scale, F64)
    errs = {}
    for precision in ("f32", "f16x2"):
        loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), scale, 0.1, float(B), precision=precision)
        errs[precision] = (abs(float(loss) - el) / abs(el), rel_err(N(lse), else_), rel_err(N(gq), egq),
                           rel_err(N(gc), egc))
    print("inbatch ranges |q|~%g |c|~%g scale %g: f32 %s | f16x2 %s" % (
        mag_q, mag_c, scale, " ".join("%.1e" % e for e in errs["f32"]), " ".join("%.1e" % e for e in errs["f16x2"])))
    # wherever exact f32 arithmetic holds the bound, so must the two-plane path (at extreme |S| the f32 rounding of the
    # score itself, amplified by exp, leaves 1e-5: both paths then have to stay within 4x of each other)
    if max(errs["f32"]) <= TOL:
        assert max(errs["f16x2"]) <= TOL, errs
    else:
        assert max(errs["f16x2"]) <= 4 * max(errs["f32"]), errs


@pytest.mark.parametrize("B", [128, 256, 384, 768, 1280, 1664, 2048, 2304, 4096])
def test_inbatch_f16x2_pass_c_forms_agree(dev, B, monkeypatch):
    """Pass C (round 6) takes the stored fp16 planes of the probabilities as its MFMA operand and streams the scaled
    copy of Q (f_i q_i), or -- the general form: ESR_IB2H_PC=general, a flagged split -- applies the factors to the
    probabilities in registers.  Different roundings (f (hi + lo) vs P' (f q)), same bound: both forms against the
    exact-f32 kernel to the path's tolerance and against each other -- at chunk counts per split of 1, 2, 3, 5, 8, 9, 13 and
    16 (the three-slot rings' prologue, every remainder of the three-step loop, half-empty last workgroups) -- and each
    form bit-stable from run to run (a request left in flight past the end of a column once corrupted the epilogue of
    the 1- and 2-chunk launches)."""
    from esrecsys_amd import ops
    g = torch.Generator(device=dev).manual_seed(7 * B)
    D = 128
    q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    outs = {}
    forms = torch.full((8,), -1, dtype=torch.int32, device=dev)
    for form in ("copy", "general"):
        if form == "general":
            monkeypatch.setenv("ESR_IB2H_PC", "general")
        else:
            monkeypatch.delenv("ESR_IB2H_PC", raising=False)
        for rep in range(3):  # repeated: a dangling request shows up as run-to-run differences
            cur = [t.clone() for t in ops.inbatch_softmax_fwd_bwd(q, c, -6.0, 0.1, 77.0, precision="f16x2",
                                                                  pass_c_forms=forms)]
            assert all(bool(torch.isfinite(t).all()) for t in cur)
            assert int(forms.abs().sum()) == 0   # benign rows: one reference per row, no split is flagged
            if form in outs:
                assert all(torch.equal(a, b) for a, b in zip(outs[form], cur)), (form, rep)
            outs[form] = cur
    monkeypatch.delenv("ESR_IB2H_PC", raising=False)
    ref = ops.inbatch_softmax_fwd_bwd(q, c, -6.0, 0.1, 77.0, precision="f32")
    for name in outs:
        for a, b in zip(outs[name], ref):
            assert rel_err(N(a), N(b)) <= 1e-5, name
    for a, b in zip(outs["copy"], outs["general"]):
        assert rel_err(N(a), N(b)) <= 2e-6


@pytest.mark.parametrize("B", [128, 384, 8192, 16384])
def test_inbatch_exponent_poll_two_levels_equals_flat(dev, B, monkeypatch):
    """prepsplit2h_kernel learns the two matrix maxima from tagged per-chunk words: workgroup 0 gathers them and the others
    poll ONE word (round 6), or every workgroup polls every word (ESR_IB2H_POLL=flat, rounds 4 - 5).  Same maxima, so
    every output is bit-identical -- at 4, 12, 256 and 512 chunks (two words per gathering thread), repeated calls on one
    workspace (the words are cleared for the next call)."""
    from esrecsys_amd import ops
    g = torch.Generator(device=dev).manual_seed(B + 1)
    D = 128
    q = torch.randn((B, D), generator=g, device=dev) * 0.3
    c = torch.randn((B, D), generator=g, device=dev) * 0.05
    c[B // 2] *= 40.0   # the candidates' maximum sits in one chunk in the middle of the grid
    outs = {}
    for form in ("two", "flat", "two"):
        monkeypatch.setenv("ESR_IB2H_POLL", form)
        for rep in range(2):
            cur = [t.clone() for t in ops.inbatch_softmax_fwd_bwd(q, c, 4.0, 0.1, float(B), precision="f16x2")]
            assert all(bool(torch.isfinite(t).all()) for t in cur)
            if outs:
                assert all(torch.equal(a, b) for a, b in zip(outs["first"], cur)), (form, rep)
            outs.setdefault("first", cur)
    monkeypatch.delenv("ESR_IB2H_POLL", raising=False)


@pytest.mark.parametrize("B", [2048, 8192])
def test_inbatch_f16x2_pass_c_redone_split_vs_oracle(dev, B):
    """A pass-Q workgroup that overflows its optimistic reference redoes itself against the exact maximum of ITS range:
    its rows' probabilities then carry a reference of their own for that split, the row factor no longer fits, and
    scaleq2h_kernel must flag the split (esr_inbatch2h_pass_c_forms) so that pass C applies the per-(row, split) factors
    in registers there.  Row 5 is dominated by a candidate of the last split (score 36 against ~0): against the fp64 oracle,
    element-wise on gC."""
    from esrecsys_amd import ops
    from oracle import stl_head as o_stl
    rng = np.random.default_rng(B)
    D = 128
    q = (rng.standard_normal((B, D)) * D ** -0.5).astype(F32)
    c = (rng.standard_normal((B, D)) * D ** -0.5).astype(F32)
    u = rng.standard_normal(D).astype(F32)
    u /= np.linalg.norm(u)
    q[5] = 6.0 * u
    c[B - 100] = 6.0 * u
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(F64), c.astype(F64), 0.1, float(B), 1.0, F64)
    forms = torch.full((8,), -1, dtype=torch.int32, device=dev)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 1.0, 0.1, float(B), precision="f16x2",
                                                    pass_c_forms=forms)
    f = forms.cpu().numpy()
    assert f.min() >= 0 and f.any(), f
    assert abs(float(loss) - el) <= 1e-5 * abs(el)
    assert rel_err(N(lse), else_) <= 1e-5 and rel_err(N(gq), egq) <= 1e-5 and rel_err(N(gc), egc) <= 1e-5
    tol = 1e-4 * np.maximum(np.abs(egc), 1e-3 * np.abs(egc).max())
    assert (np.abs(N(gc).astype(F64) - egc) <= tol).all()


@pytest.mark.parametrize("B", [1024, 8192])
def test_inbatch_f16x2_pass_c_range_guard_vs_oracle(dev, B):
    """The range guard of pass C's scaled copy of Q.  Every row but one is sharply peaked on its positive (normaliser ~16,
    factor ~2^10); row `hot` sees ~B candidates 2^15 above its reference (normaliser ~2^28, factor ~2^-14): f q of that
    row lies ~20 binades under the copy's largest element, its second fp16 plane would be subnormal.  scaleq2h_kernel
    must flag every split, and the result must hold the oracle's bound INCLUDING column 0 of gC, which only the hot row
    feeds (q[:, 0] is zero elsewhere): its entries are p_hot,j q_hot / B -- exactly what the copy form would have lost."""
    from esrecsys_amd import ops
    from oracle import stl_head as o_stl
    rng = np.random.default_rng(3 * B)
    D = 128
    q = (rng.standard_normal((B, D)) * D ** -0.5).astype(F32)
    q[:, 0] = 0.0
    c = (30.0 * q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F32)   # peaked rows: diagonal score ~30
    hot = 77
    alpha = 2.76                                   # alpha^2 log2(e) = 11 bits above the reference's 2^4
    q[hot] = 0.0
    q[hot, 0] = alpha
    c[:, 0] = alpha
    c[:32, 0] = 0.0                                # chunk 0 (the reference's sample) and the positive: score 0
    c[hot] = 0.0
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(F64), c.astype(F64), 0.1, float(B), 1.0, F64)
    forms = torch.full((8,), -1, dtype=torch.int32, device=dev)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), 1.0, 0.1, float(B), precision="f16x2",
                                                    pass_c_forms=forms)
    f = forms.cpu().numpy()
    nsplit = 8 if B >= 1024 else 4
    assert f.min() >= 0 and (f[:nsplit] != 0).all(), f
    assert abs(float(loss) - el) <= 1e-5 * abs(el)
    assert rel_err(N(lse), else_) <= 1e-5 and rel_err(N(gq), egq) <= 1e-5 and rel_err(N(gc), egc) <= 1e-5
    col = N(gc)[:, 0].astype(F64)
    assert np.abs(egc[:, 0]).max() > 0
    assert (np.abs(col - egc[:, 0]) <= 1e-4 * np.abs(egc[:, 0]) + 1e-12).all()


def test_inbatch_f16x2_largest_batch_against_exact_f32(dev):
    """B = 16384 is the largest batch of the two-plane path (1 GiB of stored probabilities): against the exact-f32 MFMA
    kernel on the same inputs (the fp64 oracle needs 2 GiB per B x B matrix at this size), plus the column-sum property
    of gC when the regulariser is off (softmax rows sum to one)."""
    from esrecsys_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    B, D = 16384, 128
    q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    loss, lse, gq, gc = [t.clone() for t in ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.0, float(B), precision="f16x2")]
    l2, lse2, gq2, gc2 = ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.0, float(B), precision="f32")
    assert abs(float(loss) - float(l2)) <= 2e-6 * abs(float(l2))
    assert rel_err(N(lse), N(lse2)) <= 2e-6 and rel_err(N(gq), N(gq2)) <= 5e-6 and rel_err(N(gc), N(gc2)) <= 5e-6
    colsum = N(gc).astype(F64).sum(0)
    assert np.abs(colsum).max() <= 1e-5 * np.abs(N(gc)).sum(0).max()
    with pytest.raises(ValueError):
        ops.inbatch_softmax_fwd_bwd(torch.zeros((16512, D), device=dev), torch.zeros((16512, D), device=dev), 1.0, 0.0,
                                    1.0, precision="f16x2")  # beyond the stored-P limit: bf16x3 / auto take it


@pytest.mark.parametrize("ref_mode", ["opt", "redo", "rowmax"])
@pytest.mark.parametrize("B", [128, 1024, 2176])
def test_inbatch_f16x2_exponent_reference_modes(dev, B, ref_mode, monkeypatch):
    """The two-plane path's exponent reference: optimistic (diagonal + first chunk; default), every block through the
    redo launch (exact maximum of the block's range), and the row-max pass.  All three against the fp64 oracle, on
    inputs where a late candidate beats the positive pair and the whole first chunk by ~30 binades (the optimistic
    reference overflows fp16 there and the redo launch has to repair exactly those blocks)."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(B + 3)
    D, scale = 128, 8.0
    q = (rng.standard_normal((B, D)) * D ** -0.5).astype(np.float32)
    c = (rng.standard_normal((B, D)) * D ** -0.5).astype(np.float32)
    late = B - 24                                  # not in the first chunk of any split
    c[late] = 3.0 * q[5] / np.linalg.norm(q[5])    # score 3 |q_5| ~ 3 against ~0 for the rest: +35 binades at scale 8
    c[B // 2 + 40] = 2.5 * q[B // 2] / np.linalg.norm(q[B // 2])
    monkeypatch.setenv("ESR_IB2H_REF", ref_mode)
    loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(T(q, dev), T(c, dev), scale, 0.1, float(B), precision="f16x2")
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(F64), c.astype(F64), 0.1, float(B), scale, F64)
    assert np.isfinite(N(gq)).all() and np.isfinite(N(gc)).all()
    assert abs(float(loss) - el) / abs(el) <= TOL and rel_err(N(lse), else_) <= TOL
    assert rel_err(N(gq), egq) <= TOL and rel_err(N(gc), egc) <= TOL


@pytest.mark.parametrize("ref_mode", ["opt", "redo"])
@pytest.mark.parametrize("B", [128, 384, 1024, 2176, 8192, 16384])
def test_inbatch_f16x2_fused_launches_equal_round3_launches(dev, B, ref_mode, monkeypatch):
    """Round 4's launches (prep + split in ONE kernel behind a tagged all-gather of the chunk maxima; pass Q redoing its
    overflowed workgroups itself instead of a flag + redo launch) against round 3's (ESR_IB2H_FUSED=0): the same planes
    and the same sweeps, so lse and both gradients agree to a rounding per element, on inputs where some workgroups
    overflow and redo, and the fused path repeats itself bit for bit (the pre-pass polls words other workgroups
    publish: a stale read would show up as run-to-run differences)."""
    from esrecsys_amd import ops
    g = torch.Generator(device=dev).manual_seed(31 * B)
    D = 128
    q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    c[B - 24] = 3.0 * q[5] / q[5].norm()          # a late dominant candidate: the optimistic reference overflows
    c[B // 2 + 40] = 2.5 * q[B // 2] / q[B // 2].norm()
    monkeypatch.setenv("ESR_IB2H_REF", ref_mode)
    monkeypatch.setenv("ESR_IB2H_FUSED", "0")
    ref = [t.clone() for t in ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision="f16x2")]
    assert all(bool(torch.isfinite(t).all()) for t in ref)
    monkeypatch.setenv("ESR_IB2H_FUSED", "1")
    first = None
    for rep in range(20 if B >= 8192 else 6):
        out = [t.clone() for t in ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision="f16x2")]
        assert abs(float(out[0]) - float(ref[0])) <= 1e-6 * abs(float(ref[0])), rep
        for name, a, b in zip(("lse", "gQ", "gC"), ref[1:], out[1:]):
            # (the merge arithmetic is the same source compiled into two kernels: fused-multiply-add contraction may
            # differ, so a rounding per element, not bit equality, between the two builds ...)
            assert rel_err(N(b), N(a)) <= 5e-7, (name, rep)
            assert float((a - b).abs().max()) <= 4e-6 * float(a.abs().max()), (name, rep)
        if first is not None:  # ... but the fused path must repeat itself bit for bit
            assert all(torch.equal(a, b) for a, b in zip(first, out)), rep
        first = out


def test_inbatch_f16x2_fused_towers_with_bf16_and_gathered_rows(dev, monkeypatch):
    """The fused pre-pass reads its rows through the row source (tower table + ids, f32 or bf16): towers entry point,
    duplicate ids, against round 3's launches."""
    from esrecsys_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    V, D, B = 5000, 128, 1024
    st = torch.randn((V, D), generator=g, device=dev) * D ** -0.5
    pt = torch.randn((V, D), generator=g, device=dev) * D ** -0.5
    sid = torch.randint(0, V, (B,), generator=g, device=dev, dtype=torch.int32)
    pid = torch.randint(0, 50, (B,), generator=g, device=dev, dtype=torch.int32)  # many duplicates
    monkeypatch.setenv("ESR_IB2H_FUSED", "0")
    ref = [t.clone() for t in ops.inbatch_towers_fwd_bwd(st, pt, sid, pid, 4.0, 0.1, float(B), precision="f16x2")]
    monkeypatch.setenv("ESR_IB2H_FUSED", "1")
    out = ops.inbatch_towers_fwd_bwd(st, pt, sid, pid, 4.0, 0.1, float(B), precision="f16x2")
    assert abs(float(out[0]) - float(ref[0])) <= 1e-6 * abs(float(ref[0]))
    for a, b in zip(ref[1:], out[1:]):
        assert rel_err(N(b), N(a)) <= 5e-7


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_inbatch_repeatable_under_load(dev, precision):
    """Race screen for the LDS-DMA ring: 60 back-to-back launches at the headline size must be bit-identical
    (a tile read before its DMA landed shows up as run-to-run differences; an earlier build that relied on
    the compiler's implicit vmcnt(0) at the loop-top barrier failed exactly this way after ~200 steps)."""
    from esrecsys_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    B, D = 8192, 128
    q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    first = [t.clone() for t in ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision=precision)]
    assert all(bool(torch.isfinite(t).all()) for t in first)
    for _ in range(60):
        out = ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision=precision)
        for a, b in zip(first, out):
            assert torch.equal(a, b)


def test_inbatch_golden_b320_falls_back_to_f32_when_not_splittable(dev):
    from esrecsys_amd import ops
    g = load_golden("inbatch_b320_d128")  # 320 % 128 != 0 -> "auto" must use the f32 kernel
    loss, _, _, _ = ops.inbatch_softmax_fwd_bwd(T(g["q"], dev), T(g["c"], dev), float(g["scale"]), float(g["lam"]),
                                                float(g["batch_size"]), precision="auto")
    assert abs(float(loss) - g["loss"]) / abs(g["loss"]) <= TOL
    with pytest.raises(ValueError):
        ops.inbatch_softmax_fwd_bwd(T(g["q"], dev), T(g["c"], dev), 1.0, 0.0, 320.0, precision="bf16x3")


# ------------------------------------------------------------------------------------------------
# sort + sparse optimizers
# ------------------------------------------------------------------------------------------------
def _ids(kind, rng, V, n):
    if kind == "uniform":
        return rng.integers(0, V, n).astype(np.int32)
    if kind == "same":
        return np.full(n, V - 1, np.int32)
    p = 1.0 / np.arange(1, V + 1)
    return rng.permutation(V)[rng.choice(V, size=n, p=p / p.sum())].astype(np.int32)


@pytest.mark.parametrize("n", [40_000, 32_768, 24_576, 16_384 + 13, 8193, 8192, 4097, 4096, 4095, 2049, 2048, 2047, 1025,
                               1024, 1000, 513, 512, 511, 74, 2, 1])  # single tile (3 sizes) / tiles + rank / radix
@pytest.mark.parametrize("kind", ["uniform", "same", "zipf"])
def test_segment_sort_is_stable_sort(dev, kind, n):
    from esrecsys_amd import ops
    rng = np.random.default_rng(9)
    V = 100_000
    ids = _ids(kind, rng, V, n)
    sid, perm = ops.segment_sort(T(ids, dev), V)
    order = np.argsort(ids, kind="stable")
    assert np.array_equal(N(perm), order.astype(np.int32))
    assert np.array_equal(N(sid), ids[order])


@pytest.mark.parametrize("n,V", [(131_072, 465_537), (196_608 + 5, 2_000_000), (262_144, 1_000), (262_145, 465_537),
                                 (40_001, 2_047), (50_000, 2_048), (100_000, 30_000_000), (2_049 + 32_768, 2 ** 22 + 1),
                                 (786_432, 2_000_000), (300_001, 1_500), (2 ** 21, 465_537), (2 ** 21 + 1, 465_537),
                                 (1_000_003, 2 ** 31 - 1), (5_000_003, 1_000_000), (3_000_000, 2 ** 31 - 1)])
@pytest.mark.parametrize("kind", ["uniform", "same", "zipf"])
def test_segment_sort_hand_written_radix_path(dev, kind, n, V):
    """32 768 < n <= 262 144: the two-launches-per-pass LSD radix sort (11-bit digits; 1, 2 or 3 passes by the id range);
    up to 2 097 152 ids (the 786 432 of a triplet step at B = 262 144) the same with the histogram matrix summed over
    32-tile segments first; longer lists (round 5: no device-library sort is left) the three-launch passes with the
    histogram matrix scanned down its columns in between.  Stable: perm == numpy's stable argsort."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(n % 1000 + 1)
    if kind == "zipf" and V > 5_000_000:
        ids = np.where(rng.random(n) < 0.3, V - 7, rng.integers(0, V, n)).astype(np.int32)  # one hot id, wide range
    else:
        ids = _ids(kind, rng, V, n)
    order = np.argsort(ids, kind="stable")
    sid, perm = ops.segment_sort(T(ids, dev), V)
    assert np.array_equal(N(perm), order.astype(np.int32))
    assert np.array_equal(N(sid), ids[order])


def test_segment_sort_multi_segments_on_the_radix_path(dev):
    """the towers of one step as segments with offsets (virtual rows), read in place by the radix kernels"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(77)
    sizes, tables = (65_536, 65_536, 65_536), (1_000_000, 1_000_000)
    segs = [rng.integers(0, tables[min(i, 1)], s).astype(np.int32) for i, s in enumerate(sizes)]
    offsets = [0, tables[0], tables[0]]
    virt = np.concatenate([s.astype(np.int64) + o for s, o in zip(segs, offsets)])
    sid, perm = ops.segment_sort_multi([T(s, dev) for s in segs], offsets, sum(tables))
    order = np.argsort(virt, kind="stable")
    assert np.array_equal(N(perm), order.astype(np.int32)) and np.array_equal(N(sid).astype(np.int64), virt[order])


@pytest.mark.parametrize("n", [1, 48, 600, 4096])
def test_segment_sort_short_lists_of_wide_ids(dev, n):
    """ids that do not fit the 32-bit tile composites, short list: the one-workgroup 64-bit bitonic kernel"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(n)
    V = 50_000_000
    ids = rng.integers(0, V, n).astype(np.int32)
    ids[::5] = V - 1
    sid, perm = ops.segment_sort(T(ids, dev), V)
    order = np.argsort(ids, kind="stable")
    assert np.array_equal(N(perm), order.astype(np.int32)) and np.array_equal(N(sid), ids[order])


def test_segment_sort_wide_ids_take_the_radix_path(dev):
    """ids >= 2^21 do not fit the 32-bit tile composites: mid-size lists take the radix path (three passes here)"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(10)
    V, n = 50_000_000, 20_000
    ids = rng.integers(0, V, n).astype(np.int32)
    ids[::7] = V - 1
    sid, perm = ops.segment_sort(T(ids, dev), V)
    order = np.argsort(ids, kind="stable")
    assert np.array_equal(N(perm), order.astype(np.int32)) and np.array_equal(N(sid), ids[order])


@pytest.mark.parametrize("kind", ["uniform", "same", "zipf"])
@pytest.mark.parametrize("D", [1, 16, 128, 256])
def test_sparse_adagrad_vs_oracle(dev, kind, D):
    from esrecsys_amd import ops
    rng = np.random.default_rng(D + len(kind))
    V, n = 2000, 3000
    ids = _ids(kind, rng, V, n)
    p0 = rng.standard_normal((V, D)).astype(np.float32)
    a0 = np.full((V, D), 0.1, np.float32)
    rows = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
    table, accum = T(p0, dev), T(a0, dev)
    sid, perm = ops.segment_sort(T(ids, dev), V)
    ops.sparse_adagrad(table, accum, sid, perm, T(rows, dev), 0.05, 1e-7)
    ep, ea = o_optim.sparse_adagrad_update(p0, a0, ids, rows, 0.05, 1e-7, np.float32)
    assert rel_err(N(table), ep) <= TOL and rel_err(N(accum), ea) <= TOL
    # fp64 oracle as well (the 1e-5 bar is against the exact answer)
    ep64, ea64 = o_optim.sparse_adagrad_update(p0.astype(F64), a0.astype(F64), ids, rows.astype(F64), 0.05, 1e-7, F64)
    assert rel_err(N(table), ep64) <= TOL and rel_err(N(accum), ea64) <= TOL
    untouched = np.setdiff1d(np.arange(V), ids)
    assert np.array_equal(N(table)[untouched], p0[untouched])  # bit-exact: rows without gradient never move


def test_segment_sum_order_is_fixed(dev):
    """Duplicate ids are summed in a FIXED order: a run of up to 32 occurrences (up to 63, depending on where it
    starts) left to right == a sequential fp32 scatter-add, bit for bit; a longer run as chunk partials (head chunk
    up to the first multiple of 32 at least 32 positions on, then 32 at a time), the partials round-robin over the
    row groups and then in group order.  Restated here in numpy, compared bit for bit; same bits on a second launch."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(4)
    V, n, D = 50, 4000, 32        # ~80 occurrences per row: every run crosses chunk boundaries
    ids = rng.integers(0, V, n).astype(np.int32)
    ids[:7] = V - 1               # (every row has ~80 occurrences here; short runs are covered by the Adagrad tests)
    rows = rng.standard_normal((n, D)).astype(np.float32)
    sid, perm = ops.segment_sort(T(ids, dev), V)
    dense = N(ops.rows_to_dense(V, D, sid, perm, T(rows, dev)))
    again = N(ops.rows_to_dense(V, D, sid, perm, T(rows, dev)))
    assert np.array_equal(dense, again)
    order = np.argsort(ids, kind="stable")
    sids = ids[order]
    exp = np.zeros((V, D), np.float32)
    NG = 256 // 8                 # D = 32 -> 8 lanes per row group -> 32 groups per workgroup
    p = 0
    while p < n:
        q = p
        while q < n and sids[q] == sids[p]:
            q += 1
        parts, c = [], p          # chunk partials: head chunk, then aligned chunks of 32
        while c < q:
            e = min(((c + 63) // 32) * 32 if c == p else c + 32, q)
            acc = rows[order[c]].copy()
            for k in range(c + 1, e):
                acc = acc + rows[order[k]]
            parts.append(acc)
            c = e
        if len(parts) == 1:
            total = parts[0]
        else:
            sums = []
            for g in range(min(NG, len(parts))):
                acc = np.zeros(D, np.float32)
                for i in range(g, len(parts), NG):
                    acc = acc + parts[i]
                sums.append(acc)
            total = sums[0]
            for g in range(1, len(sums)):
                total = total + sums[g]
        exp[sids[p]] = total
        p = q
    assert np.array_equal(dense, exp)
    seq = np.zeros((V, D), np.float64)
    np.add.at(seq, ids, rows.astype(np.float64))
    assert rel_err(dense, seq) <= 1e-6


def test_sparse_adagrad_bf16_table(dev):
    from esrecsys_amd import ops
    rng = np.random.default_rng(8)
    V, n, D = 1000, 1500, 128
    ids = rng.integers(0, V, n).astype(np.int32)
    p0 = torch.from_numpy(rng.standard_normal((V, D)).astype(np.float32)).to(torch.bfloat16)
    rows = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
    table, accum = p0.to(dev), torch.full((V, D), 0.1, device=dev)
    sid, perm = ops.segment_sort(T(ids, dev), V)
    ops.sparse_adagrad(table, accum, sid, perm, T(rows, dev), 0.05, 1e-7)
    ep, ea = o_optim.sparse_adagrad_update(p0.float().numpy().astype(F64), np.full((V, D), 0.1), ids,
                                           rows.astype(F64), 0.05, 1e-7, F64)
    exp_bf16 = torch.from_numpy(ep).to(torch.bfloat16).float().numpy()
    got = N(table)
    # result is the fp64 answer rounded to bf16 (RNE), up to one bf16 ulp where fp32 rounding straddles a tie
    assert np.mean(got == exp_bf16) > 0.999
    assert rel_err(got, ep) <= 2.0 ** -8
    assert rel_err(N(accum), ea) <= TOL


def test_gather_rows_multi(dev):
    from esrecsys_amd import ops
    g = torch.Generator().manual_seed(5)
    t0, t1, t2 = (torch.randn((n, 128), generator=g).to(dev) for n in (300, 500, 64))
    offs = [0, 304, 808, 872]  # padded boundaries, as the sharded groups use
    vids = torch.cat([torch.randint(0, 300, (100,), generator=g), 304 + torch.randint(0, 500, (150,), generator=g),
                      808 + torch.randint(0, 64, (33,), generator=g)]).to(torch.int32)
    vids = vids[torch.randperm(vids.numel(), generator=g)]
    out = ops.gather_rows_multi([t0, t1, t2], offs, vids.to(dev))
    full = torch.zeros((872, 128))
    full[0:300], full[304:804], full[808:872] = t0.cpu(), t1.cpu(), t2.cpu()
    assert torch.equal(out.cpu(), full[vids.long()])


def test_fused_multi_table_adagrad_equals_per_table(dev):
    """concat_offset_ids + one sort + esr_sparse_adagrad_scatter_multi == the per-table path (bit for bit on rows whose
    occurrences sit in one 32-position chunk in both layouts; to fp32 rounding on the 50-fold duplicated rows, whose
    chunk boundaries fall differently in the concatenated list)."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(21)
    V0, V1, D, B = 700, 1100, 128, 600
    t0, t1 = rng.standard_normal((V0, D)).astype(np.float32), rng.standard_normal((V1, D)).astype(np.float32)
    i0 = rng.integers(0, V0, B).astype(np.int32)
    i1 = rng.integers(0, V1, B).astype(np.int32)
    i2 = rng.integers(0, V1, B).astype(np.int32)
    i0[:50], i1[:50], i2[:50] = 0, V1 - 1, V1 - 1  # duplicates, shared rows between segments 1 and 2, edge rows
    rows = (rng.standard_normal((3 * B, D)) * 0.1).astype(np.float32)
    # fused: segments [i0 -> table0, i1 -> table1, i2 -> table1]
    f0, f1 = T(t0, dev), T(t1, dev)
    a0, a1 = torch.full((V0, D), 0.1, device=dev), torch.full((V1, D), 0.1, device=dev)
    vids = ops.concat_offset_ids([T(i0, dev), T(i1, dev), T(i2, dev)], [0, V0, V0])
    assert np.array_equal(N(vids), np.concatenate([i0, i1 + V0, i2 + V0]))
    sv, perm = ops.segment_sort(vids, V0 + V1)
    sv2, perm2 = ops.segment_sort_multi([T(i0, dev), T(i1, dev), T(i2, dev)], [0, V0, V0], V0 + V1)   # no concat copy
    assert torch.equal(sv, sv2) and torch.equal(perm, perm2)
    big = [T(rng.integers(0, 5_000_000, 30_000).astype(np.int32), dev) for _ in range(2)]              # radix path
    sb, pb = ops.segment_sort_multi(big, [0, 5_000_000], 10_000_000)
    sb1, pb1 = ops.segment_sort(ops.concat_offset_ids(big, [0, 5_000_000]), 10_000_000)
    assert torch.equal(sb, sb1) and torch.equal(pb, pb1)
    ops.sparse_adagrad_multi([f0, f1], [a0, a1], [0, V0, V0 + V1], sv, perm, T(rows, dev), 0.05, 1e-7)
    # per-table reference path
    p0, p1 = T(t0, dev), T(t1, dev)
    b0, b1 = torch.full((V0, D), 0.1, device=dev), torch.full((V1, D), 0.1, device=dev)
    s0, q0 = ops.segment_sort(T(i0, dev), V0)
    ops.sparse_adagrad(p0, b0, s0, q0, T(rows[:B], dev), 0.05, 1e-7)
    s1, q1 = ops.segment_sort(T(np.concatenate([i1, i2]), dev), V1)
    ops.sparse_adagrad(p1, b1, s1, q1, T(rows[B:], dev), 0.05, 1e-7)
    hot0, hot1 = np.arange(V0) == 0, np.arange(V1) == V1 - 1
    assert np.array_equal(N(f0)[~hot0], N(p0)[~hot0]) and np.array_equal(N(f1)[~hot1], N(p1)[~hot1])
    assert np.array_equal(N(a0)[~hot0], N(b0)[~hot0]) and np.array_equal(N(a1)[~hot1], N(b1)[~hot1])
    assert rel_err(N(f0), N(p0)) <= 1e-6 and rel_err(N(f1), N(p1)) <= 1e-6
    e1, _ = o_optim.sparse_adagrad_update(t1.astype(F64), np.full((V1, D), 0.1), np.concatenate([i1, i2]),
                                          rows[B:].astype(F64), 0.05, 1e-7, F64)
    assert rel_err(N(f1), e1) <= TOL


def test_sparse_sgd_vs_oracle(dev):
    from esrecsys_amd import ops
    rng = np.random.default_rng(12)
    V, n, D = 300, 1000, 64
    ids = rng.integers(0, V, n).astype(np.int32)
    p0 = rng.standard_normal((V, D)).astype(np.float32)
    rows = rng.standard_normal((n, D)).astype(np.float32)
    table = T(p0, dev)
    sid, perm = ops.segment_sort(T(ids, dev), V)
    ops.sparse_sgd(table, sid, perm, T(rows, dev), 0.01)
    g = np.zeros((V, D), F64)
    np.add.at(g, ids, rows.astype(F64))
    assert rel_err(N(table), p0 - 0.01 * g) <= TOL


def test_dense_adam_vs_golden(dev):
    from esrecsys_amd import ops
    g = load_golden("optim_v64_d8")
    p = T(g["p0"], dev)
    mu, nu = torch.zeros_like(p), torch.zeros_like(p)
    for k in range(3):
        ops.dense_adam(p, mu, nu, T(g["adam_grads"][k], dev), 1e-3, k + 1)
        # Adam divides small numbers: compare the UPDATE, not just the parameter
        assert rel_err(N(p), g["adam_params"][k]) <= 1e-6
        step_exp = g["adam_params"][k] - (g["adam_params"][k - 1] if k else g["p0"].astype(F64))
        step_got = N(p).astype(F64) - (g["adam_params"][k - 1] if k else g["p0"].astype(F64))
        assert np.abs(step_got - step_exp).max() <= 5e-7  # one fp32 ulp of the O(1) parameters


def test_dense_adam_odd_length_tail(dev):
    from esrecsys_amd import ops
    rng = np.random.default_rng(2)
    n = 4 * 1000 + 3
    p0, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    p = T(p0, dev)
    mu, nu = torch.zeros_like(p), torch.zeros_like(p)
    ops.dense_adam(p, mu, nu, T(g, dev), 1e-2, 1)
    ep, _ = o_optim.adam_update(p0.astype(F64), g.astype(F64), o_optim.adam_init(p0.astype(F64)), 1e-2, dtype=F64)
    assert np.abs(N(p) - ep).max() <= 1e-6


# ------------------------------------------------------------------------------------------------
# retrieval
# ------------------------------------------------------------------------------------------------
def test_score_all_and_argsort_vs_golden_with_ties(dev):
    from esrecsys_amd import ops
    g = load_golden("topk_n500_d8_k10")
    emb = T(g["cand"], dev)
    scores = ops.score_all(emb, T(g["token"], dev))
    assert scores.shape == (500, 5)
    assert np.array_equal(N(scores), g["knn_scores"].astype(np.float32))  # small integers: exact
    idx = ops.argsort_columns(scores)
    assert idx.dtype == torch.int32 and np.array_equal(N(idx), g["knn_indices"])


@pytest.mark.parametrize("V,T_,k", [(500, 5, 10), (465_537, 8, 10), (70_000, 3, 1024), (40, 2, 40), (3000, 1, 1)])
def test_topk_columns_equals_the_tail_of_the_stable_argsort(dev, V, T_, k):
    """dump_knn reads indices[-10:] (wikipedia/train_cooccurence.py:114-126): the radix select per column must give exactly
    the rows the full stable ascending argsort puts last -- ties included (scores on a coarse grid: thousands of them;
    among equal scores the HIGHER row sorts later) -- through ops.topk_columns and through find_knn's lazy indices."""
    from esrecsys_amd import ops
    from esrecsys_amd.wikipedia.train_cooccurence import ColumnArgsort
    rng = np.random.default_rng(V + k)
    scores = T((rng.integers(-40, 40, (V, T_)) / 4.0).astype(np.float32), dev)
    full = N(ops.argsort_columns(scores)).astype(np.int64)
    s, i = ops.topk_columns(scores, k)
    assert np.array_equal(N(i).T, full[::-1][:k])
    assert np.array_equal(N(s).T, np.take_along_axis(N(scores), full[::-1][:k], axis=0))
    lazy = ColumnArgsort(scores)
    assert np.array_equal(N(lazy[-k:]), full[-k:]) and lazy._full is None      # no full sort was run for the tail
    assert np.array_equal(N(lazy[: 7]), full[:7]) and lazy._full is not None   # anything else materialises it
    assert lazy.shape == (V, T_) and np.array_equal(N(lazy), full)


@pytest.mark.parametrize("V,T_", [(1, 1), (37, 3), (2048, 2), (4097, 1), (5000, 9), (40_000, 8), (300_000, 3),
                                  (465_537, 10), (2_097_152, 1), (2_097_153, 2), (3_000_001, 1)])
def test_argsort_columns_equals_numpy_stable_argsort(dev, V, T_):
    """esr_argsort_columns on this library's own radix sort (V <= 2^21; three 11-bit passes over the order-preserving
    images of the floats, eight columns per launch sequence): exactly numpy's stable ascending argsort per column --
    negative values, ties (a coarse grid), infinities; one column, a full group of eight, eight plus a remainder."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(V + T_)
    x = rng.standard_normal((V, T_)).astype(np.float32) * 50.0
    grid = rng.random((V, T_)) < 0.4
    x[grid] = np.round(x[grid] / 8.0) * 8.0          # many exact ties
    if V > 10:
        x[3, 0], x[5, 0] = np.inf, -np.inf
    xz = np.where(x == 0.0, np.float32(0.0), x)                            # (no -0.0 in the data: round() may give it)
    want = np.argsort(xz, axis=0, kind="stable")
    got = N(ops.argsort_columns(T(xz, dev))).astype(np.int64)
    assert np.array_equal(got, want)


def test_find_top_k_vs_golden_with_ties(dev):
    from esrecsys_amd import ops
    g = load_golden("topk_n500_d8_k10")
    s, i = ops.score_topk(T(g["query"], dev), T(g["cand"], dev), int(g["k"]))
    assert np.array_equal(N(i)[0], g["topk_indices"])
    assert np.array_equal(N(s)[0], g["topk_scores"].astype(np.float32))


@pytest.mark.parametrize("nq,N_,D,k", [(1, 100_000, 128, 10), (3, 20_000, 512, 500), (5, 999, 96, 999),
                                       (2, 5_000, 64, 1024), (2, 5_000, 64, 1025), (1, 1_000_000, 32, 500),
                                       (11, 60_000, 32, 3000), (1, 300_000, 16, 2048), (9, 3000, 8, 3000)])
def test_score_topk_random_vs_oracle(dev, nq, N_, D, k):
    from esrecsys_amd import ops
    rng = np.random.default_rng(nq + k)
    q = rng.standard_normal((nq, D)).astype(np.float32)
    c = rng.standard_normal((N_, D)).astype(np.float32)
    s, i = ops.score_topk(T(q, dev), T(c, dev), k)
    es, ei = o_topk.batched_top_k(q, c, k, F64)
    got_s = N(s)
    assert rel_err(got_s, es) <= TOL
    assert np.all(np.diff(got_s, axis=1) <= 0)
    # indices may legitimately differ only where fp32 scores tie / nearly tie: check by score instead
    full = q.astype(F64) @ c.astype(F64).T
    picked = np.take_along_axis(full, N(i).astype(np.int64), axis=1)
    assert rel_err(picked, es) <= TOL
    assert np.mean(N(i) == ei) > 0.99


# ------------------------------------------------------------------------------------------------
# shard routing
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world,sizes", [(8, (8192, 8192)), (8, (8192, 8192, 8192)), (3, (5000, 1, 70000)),
                                         (8, (700, 900)), (8, (600_000, 600_000)), (2, (0, 4096, 0, 100))])
def test_bucket_ids_by_owner_segments_vs_oracle(dev, world, sizes):
    """the segmented entry point == the plain one on the concatenated virtual ids, on every path (tiled, one
    workgroup, radix), with the counts written into a caller buffer"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(len(sizes) * 100 + world)
    offsets = [int(x) for x in np.cumsum([0] + [1_000_000 + 8 * i for i in range(len(sizes) - 1)])]
    segs = [rng.integers(0, 1_000_000, n).astype(np.int32) for n in sizes]
    vids = np.concatenate([s + o for s, o in zip(segs, offsets)]).astype(np.int32)
    co = torch.full((world,), -1, dtype=torch.int64, device=dev)
    local, perm, counts, inv = ops.bucket_ids_by_owner([T(s, dev) for s in segs], world, want_inverse=True,
                                                       offsets=offsets, counts_out=co)
    assert counts.data_ptr() == co.data_ptr()
    el, ec, ep = o_shard.bucket_by_owner(vids, world)
    assert np.array_equal(N(co), ec)
    assert np.array_equal(N(perm), ep)
    assert np.array_equal(N(local), el)
    assert np.array_equal(N(inv)[N(perm)], np.arange(len(vids), dtype=np.int32))



@pytest.mark.parametrize("world,n", [(1, 16_389), (2, 16_389), (3, 16_389), (8, 16_389), (8, 7), (8, 2048), (8, 2049),
                                     (5, 3000), (8, 65_536), (8, 100_003), (7, 1_048_576), (8, 1_048_577),
                                     (16, 16_389), (16, 3_000_001), (8, 2_500_000)])
def test_bucket_ids_by_owner_vs_oracle(dev, world, n):
    """one-workgroup kernel (n <= 2048), tiled two-launch path (<= 1 Mi ids, world <= 8), the library's radix sort of the
    owner keys beyond (more ranks than the tiled path counts, or longer lists)"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(world)
    ids = rng.integers(0, 1_000_000, n).astype(np.int32)
    local, perm, counts, inv = ops.bucket_ids_by_owner(T(ids, dev), world, want_inverse=True)
    assert np.array_equal(N(inv)[N(perm)], np.arange(n, dtype=np.int32))
    el, ec, ep = o_shard.bucket_by_owner(ids, world)
    assert np.array_equal(N(counts), ec)
    assert np.array_equal(N(perm), ep)
    assert np.array_equal(N(local), el)


@pytest.mark.parametrize("n_seg,nb", [((100, 100, 100), 8), ((683, 683, 682), 5), ((2048,), 3), ((2049,), 2),
                                      ((2048, 2048, 2048), 8), ((8192, 8192, 8192), 8), ((8192, 8192, 8192), 1),
                                      ((10923, 10923, 10922), 4), ((20000, 20000), 3), ((65536, 65536), 8),
                                      ((131072, 131072), 3), ((50001, 50000), 5), ((131073, 131072), 2),
                                      ((262144, 262144, 262144), 3), ((400000, 300001), 2)])
def test_segment_sort_batched_equals_list_by_list(dev, n_seg, nb):
    """esr_segment_sort_ids_batched: one-tile lists (one launch for all), mid lists (two launches for all: the triplet
    step's 24 576 ids; the largest, 32 768), lists up to 262 144 ids (the radix passes over all lists at once: GloVe's
    131 072 at C3), longer ones (list after list) -- every list exactly what the one-list sort gives, which is the stable
    sort of [ids_k + offset_k]."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(sum(n_seg) + nb)
    V = 1_000_000
    offsets = [0, V, V][:len(n_seg)]
    lists = [[torch.from_numpy(np.where(rng.random(n) < 0.2, rng.integers(0, 5, n), rng.integers(0, V, n)).astype(np.int32)).to(dev)
              for n in n_seg] for _ in range(nb)]
    srt, prm = ops.segment_sort_batched(lists, offsets, 2 * V)
    assert srt.shape == (nb, sum(n_seg)) and prm.shape == srt.shape
    for b, segs in enumerate(lists):
        virt = np.concatenate([N(t).astype(np.int64) + o for t, o in zip(segs, offsets)])
        order = np.argsort(virt, kind="stable")
        assert np.array_equal(N(prm[b]), order.astype(np.int32))
        assert np.array_equal(N(srt[b]), virt[order].astype(np.int32))


@pytest.mark.parametrize("n_seg,nb,world", [((8192, 8192), 8, 8), ((8192, 8192, 8192), 8, 2), ((3000, 3001), 3, 5),
                                            ((500, 500), 4, 8), ((1100,), 2, 7), ((40000,), 2, 8)])
def test_bucket_ids_by_owner_batched_equals_list_by_list(dev, n_seg, nb, world):
    """esr_bucket_ids_by_owner_batched (the routing plans of several coming batches in one launch pair): every list
    exactly what bucket_ids_by_owner gives for it alone -- stable order by owner = id mod world, local rows, inverse and
    counts -- on the tiled path (n > 2048) and on the list-after-list fallback (short lists)."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(sum(n_seg) + nb + world)
    V = 1_000_000
    offsets = [0, 8 * ((V + 7) // 8), 8 * ((V + 7) // 8)][:len(n_seg)]
    offsets = [o - o % world + (world if o % world else 0) if o else 0 for o in offsets]  # multiples of the world size
    lists = [[torch.from_numpy(rng.integers(0, V, n).astype(np.int32)).to(dev) for n in n_seg] for _ in range(nb)]
    local, perm, counts, inv = ops.bucket_ids_by_owner_batched(lists, world, offsets)
    n = sum(n_seg)
    assert local.shape == perm.shape == inv.shape == (nb, n) and counts.shape == (nb, world)
    for b, segs in enumerate(lists):
        l1, p1, c1, i1 = ops.bucket_ids_by_owner(list(segs), world, want_inverse=True, offsets=offsets)
        assert torch.equal(local[b], l1) and torch.equal(perm[b], p1) and torch.equal(counts[b], c1)
        assert torch.equal(inv[b], i1)
        virt = np.concatenate([N(t).astype(np.int64) + o for t, o in zip(segs, offsets)])
        order = np.argsort(virt % world, kind="stable")
        assert np.array_equal(N(perm[b]), order.astype(np.int32))
        assert np.array_equal(N(local[b]), (virt[order] // world).astype(np.int32))


def test_sparse_adagrad_multi_long_runs_hint(dev):
    """esr_sparse_adagrad_scatter_multi(long_runs=0) skips the launch that combines chunk partials: on a list the hint
    kernel clears (no id with a run beyond 32 positions) tables and accumulators are bit-identical to the full call;
    esr_long_run_hint says "long" exactly when some id has more than 32 consecutive sorted positions."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(5)
    V, D, n = 4000, 64, 6000
    for hot, expect_long in ((0, False), (33, True), (32, False)):
        ids = np.concatenate([rng.integers(10, V, n - hot).astype(np.int32) % (V - 10) + 10, np.full(hot, 3, np.int32)])
        # (ids >= 10 are drawn at random: a few repeats, none near 32; id 3 appears `hot` times)
        rng.shuffle(ids)
        a_ids, b_ids = T(ids[:n // 2].copy(), dev), T(ids[n // 2:].copy(), dev)
        srt, prm = ops.segment_sort_multi([a_ids, b_ids], [0, V], 2 * V)
        hint = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.long_run_hint(srt, 32, hint, 9)
        counts = np.bincount(np.concatenate([ids[:n // 2], ids[n // 2:] + V]))
        assert (int(hint) == 9) == bool(counts.max() > 32)
        if counts.max() > 32:
            continue
        grads = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)).to(dev)
        res = []
        for long_runs in (-1, 0):
            t0 = torch.from_numpy(np.random.default_rng(1).standard_normal((V, D)).astype(np.float32)).to(dev)
            t1 = t0.clone() * 0.5
            a0, a1 = torch.zeros_like(t0), torch.zeros_like(t1)
            ops.sparse_adagrad_multi([t0, t1], [a0, a1], [0, V, 2 * V], srt, prm, grads.clone(), 0.05, 1e-7,
                                     long_runs=long_runs)
            res.append((t0, t1, a0, a1))
        for x, y in zip(*res):
            assert torch.equal(x, y)


@pytest.mark.parametrize("nq,N_", [(10, 20_000), (9, 2 ** 21 + 5), (1, 2 ** 21 + 5)])
def test_score_topk_beyond_1024_is_stable_on_ties(dev, nq, N_):
    """k > 1024: full descending sort per query row on the library's radix sort, eight rows per launch sequence -- with
    scores on a coarse integer grid (exact in f32: thousands of ties) the result must be exactly lax.top_k's order,
    lower index first among equal scores."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(12)
    D, k = 4, 1500   # (beyond 2 097 152 candidates: the lists that took the device-library sort before round 5)
    q = rng.integers(-3, 4, (nq, D)).astype(np.float32)
    c = rng.integers(-3, 4, (N_, D)).astype(np.float32)
    s, i = ops.score_topk(T(q, dev), T(c, dev), k)
    full = q @ c.T
    order = np.argsort(-full, axis=1, kind="stable")[:, :k]
    assert np.array_equal(N(i), order.astype(np.int32))
    assert np.array_equal(N(s), np.take_along_axis(full, order, axis=1))


@pytest.mark.parametrize("n_seg,nb", [((16384,), 6), ((3000, 3000, 3000), 8), ((100,), 3), ((40000, 40000), 4), ((2048,), 2)])
def test_segment_sort_batched_wide_ids(dev, n_seg, nb):
    """Virtual rows beyond 2^21 (two towers of 3 M rows each; config 4's 100 M rows): the 32-bit (id << 11 | position)
    composites of the tile sorts do not fit, every batched list goes through the radix passes whatever its length --
    found by scripts/fuzz_ops.py: the workspace query sized those passes only for lists beyond 32 768 ids and the sort
    wrote past its workspace."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(sum(n_seg) + nb)
    V = 3_000_000
    offsets = [0, V, V][:len(n_seg)]
    lists = [[torch.from_numpy(np.where(rng.random(n) < 0.2, rng.integers(0, 5, n), rng.integers(0, V, n)).astype(np.int32)).to(dev)
              for n in n_seg] for _ in range(nb)]
    srt, prm = ops.segment_sort_batched(lists, offsets, 2 * V)
    for b, segs in enumerate(lists):
        virt = np.concatenate([N(t).astype(np.int64) + o for t, o in zip(segs, offsets)])
        order = np.argsort(virt, kind="stable")
        assert np.array_equal(N(prm[b]), order.astype(np.int32))
        assert np.array_equal(N(srt[b]), virt[order].astype(np.int32))
