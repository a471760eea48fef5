"""GPU: the one-pass steps bench.py TIMES, at BASELINE config size, straight against the fp64 oracle.

The direct triplet step (esr_triplet_plan + triplet_direct [+ triplet_direct_long], driven by ``train_steps`` in groups
of eight planned batches) and the one-pass GloVe step (``train_epoch``: glove_step on plans for short lists,
glove_resolve -> glove_step_resolved for long ones) are compared elsewhere with the build's own multi-launch paths at
these sizes and with the oracle at toy sizes; here the oracle (oracle/stl_head.py, oracle/glove.py, oracle/optim.py --
pinterest/train_shop_the_look.py:93-109, wikipedia/train_cooccurence.py:71-101 + row-sparse Adagrad) steps full-size
fp64 copies of the same tables on the same batches: every loss within 1e-5, every touched row of tables AND
accumulators within 1e-5 of its largest entry at the end, every other row bit-identical to the initial draw.

Id streams: half of every batch is uniform over the whole table (the plan the bench sees: mostly rows that occur
once), half comes from a 20 000-row window, so rows are revisited within a batch (runs of 2 - 8: parked gradients and
the completing arrival; > 8 at the saturating batch: the long-run launch) and across steps (errors would compound);
"zipf": Zipf(1) over a permutation of the whole table (runs of hundreds to thousands: triplet_direct_long /
the GloVe long-run chunks)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
F64 = np.float64
TOL = 1e-5
WINDOW = 20_000


def _zipf_sampler(V, rng):
    w = 1.0 / np.arange(1, V + 1)
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(V)
    return lambda shape: perm[np.minimum(np.searchsorted(cdf, rng.random(shape)), V - 1)].astype(np.int32)


def _mixed_sampler(V, base, rng):
    def draw(shape):
        ids = rng.integers(0, V, shape)
        win = base + rng.integers(0, WINDOW, shape)
        return np.where(rng.random(shape) < 0.5, ids, win).astype(np.int32)
    return draw


def _row_err(got, want, rows):
    """Largest |got - want| over the given rows relative to that row's largest entry, and the norm-wise error."""
    g, w = got[rows].double().cpu().numpy(), want[rows]
    per_row = float((np.abs(g - w).max(1) / np.maximum(np.abs(w).max(1), 1e-30)).max())
    return per_row, rel_err(g, w)


def _untouched_equal(table, initial, touched_rows):
    mask = torch.ones(table.shape[0], dtype=torch.bool, device=table.device)
    mask[torch.as_tensor(touched_rows, device=table.device)] = False
    return bool(torch.equal(table[mask], initial[mask]))


# ---------------------------------------------------------------------------------------------------------------------
# Shop-The-Look triplet step, BASELINE configs[1] tables (two 1 M x 128 fp32 towers)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,steps,kind", [(8192, 12, "mixed"), (8192, 10, "zipf"), (262_144, 4, "mixed")])
def test_direct_triplet_trajectory_at_c2_size_vs_fp64_oracle(dev, B, steps, kind):
    from esrecsys_amd import TrainState, ops, optim
    from esrecsys_amd.pinterest.models import STLModel
    from esrecsys_amd.pinterest.train_shop_the_look import fused_triplet_step_available, train_steps
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    V, D, lam, lr = 1_000_000, 128, 0.1, 0.05
    g = torch.Generator(device=dev).manual_seed(1701)

    def tower():
        # row norms 0.6 or 3.0 (never within an f32 rounding of the regulariser's kink at 1); margins 1 + neg - pos on
        # both sides of 0
        t = torch.randn((V, D), generator=g, device=dev) * D ** -0.5
        t *= torch.where(torch.rand((V, 1), generator=g, device=dev) < 0.3, 0.6, 3.0)
        return t
    st, pt = tower(), tower()
    st0, pt0 = st.clone(), pt.clone()
    model = STLModel(output_size=D, num_scenes=V, num_products=V, device=dev)
    state = TrainState.create(apply_fn=model.apply, tx=optim.sparse_adagrad(lr),
                              params={"params": {"scene_tower": {"embedding": st}, "product_tower": {"embedding": pt}}})
    assert fused_triplet_step_available(state) and ops.triplet_direct_mode()
    rng = np.random.default_rng(B + steps)
    if kind == "zipf":
        draw_s, draw_p = _zipf_sampler(V, rng), _zipf_sampler(V, rng)
    else:
        draw_s, draw_p = _mixed_sampler(V, 123_456, rng), _mixed_sampler(V, 654_321, rng)
    batches = [(draw_s(B), draw_p(B), draw_p(B)) for _ in range(steps)]
    es, ep = st.double().cpu().numpy(), pt.double().cpu().numpy()
    a_s, a_p = np.full_like(es, 0.1), np.full_like(ep, 0.1)
    dev_batches = [tuple(torch.as_tensor(x, device=dev) for x in b) for b in batches]
    state, losses = train_steps(state, iter(dev_batches), steps, lam, float(B))
    losses = losses.cpu().numpy()
    worst_loss, hinge_on, longest = 0.0, [], 0
    for k, (sid, pid, nid) in enumerate(batches):
        s, p, n = es[sid], ep[pid], ep[nid]
        el, gs, gp, gn = o_stl.triplet_loss_and_grads(s, p, n, lam, B, F64)
        hinge_on.append(float(np.mean(1.0 + (s * n).sum(1) - (s * p).sum(1) > 0)))
        longest = max(longest, int(np.bincount(np.concatenate([pid, nid])).max()))
        worst_loss = max(worst_loss, abs(float(losses[k]) - el) / abs(el))
        assert abs(float(losses[k]) - el) <= TOL * abs(el), (k, float(losses[k]), el)
        o_optim.sparse_adagrad_update_inplace(es, a_s, sid, gs, lr)
        o_optim.sparse_adagrad_update_inplace(ep, a_p, np.concatenate([pid, nid]), np.concatenate([gp, gn]), lr)
    p_ = state.params["params"]
    acc = state.opt_state["sum_of_squares"]["params"]
    ts = np.unique(np.concatenate([b[0] for b in batches]))
    tp = np.unique(np.concatenate([np.concatenate(b[1:]) for b in batches]))
    errs = [_row_err(p_["scene_tower"]["embedding"], es, ts), _row_err(p_["product_tower"]["embedding"], ep, tp),
            _row_err(acc["scene_tower"]["embedding"], a_s, ts), _row_err(acc["product_tower"]["embedding"], a_p, tp)]
    print("direct triplet step, C2 towers, B = %d %s, %d steps: worst loss error %.2e; hinge active on %.0f-%.0f %% of the "
          "triplets; longest run of one product row %d; touched rows %d / %d; per-row / norm-wise error towers %.2e / %.2e, "
          "%.2e / %.2e, accumulators %.2e / %.2e, %.2e / %.2e"
          % (B, kind, steps, worst_loss, 100 * min(hinge_on), 100 * max(hinge_on), longest, ts.size, tp.size,
             *errs[0], *errs[1], *errs[2], *errs[3]))
    assert 0.02 < min(hinge_on) and max(hinge_on) < 0.98, "the batches must have margins on both sides of the hinge"
    assert max(e[0] for e in errs) <= TOL
    # rows no triplet touched: bit-identical to the initial draw, accumulators still exactly 0.1
    assert _untouched_equal(p_["scene_tower"]["embedding"], st0, ts)
    assert _untouched_equal(p_["product_tower"]["embedding"], pt0, tp)
    for t, rows in (("scene_tower", ts), ("product_tower", tp)):
        a = acc[t]["embedding"]
        assert _untouched_equal(a, torch.full_like(a, 0.1), rows)


# ---------------------------------------------------------------------------------------------------------------------
# GloVe one-pass step, BASELINE configs[2] table (V = 400 000 + 65 537 = 465 537 rows x 256)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,steps,kind,mode", [(65_536, 6, "mixed", "reference"), (2048, 20, "mixed", "reference"),
                                               (2048, 12, "zipf", "reference"), (65_536, 3, "zipf", "reference"),
                                               (2048, 12, "mixed", "diagonal")])
def test_glove_one_pass_trajectory_at_c3_size_vs_fp64_oracle(dev, B, steps, kind, mode):
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.wikipedia.models import Glove
    from esrecsys_amd.wikipedia.train_cooccurence import fused_step_available, train_epoch
    from oracle import glove as o_glove
    from oracle import optim as o_optim
    V, D, lr = 465_537, 256, 0.05
    model = Glove(num_embeddings=V, features=D, loss_mode=mode, device=dev)
    params = model.init(1701, None)["params"]
    g = torch.Generator(device=dev).manual_seed(7)
    params["_bias"]["embedding"].copy_(torch.randn((V, 1), generator=g, device=dev) * 0.05)
    state = TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(lr))
    assert fused_step_available(state)
    emb0, bias0 = params["_token_embedding"]["embedding"].clone(), params["_bias"]["embedding"].clone()
    emb, bias = emb0.double().cpu().numpy(), bias0.double().cpu().numpy()
    a_e, a_b = np.full_like(emb, 0.1), np.full_like(bias, 0.1)
    rng = np.random.default_rng(B + steps)
    draw = _zipf_sampler(V, rng) if kind == "zipf" else _mixed_sampler(V, 222_222, rng)
    # counts log-uniform on (0.1, 1000): about a third saturate the weight
    batches = [(draw((2, B)), np.exp(rng.uniform(np.log(0.1), np.log(1000.0), B)).astype(np.float32))
               for _ in range(steps)]
    dev_batches = [(torch.as_tensor(i, device=dev), torch.as_tensor(t, device=dev)) for i, t in batches]
    got = []
    state, mean_loss = train_epoch(state, steps, iter(dev_batches), losses_out=got)
    losses = got[0].cpu().numpy()
    assert losses.shape == (steps,) and abs(float(losses.mean()) - mean_loss) <= 1e-6 * abs(mean_loss)
    worst_loss, longest = 0.0, 0
    for k, (inputs, target) in enumerate(batches):
        el, gdot, gs = o_glove.loss_and_grads(emb, bias, inputs, target.astype(F64), mode, F64)
        ids, rows, gb = o_glove.row_grads(emb, inputs, gdot, gs, F64)
        longest = max(longest, int(np.bincount(ids).max()))
        worst_loss = max(worst_loss, abs(float(losses[k]) - el) / abs(el))
        assert abs(float(losses[k]) - el) <= TOL * abs(el), (k, float(losses[k]), el)
        o_optim.sparse_adagrad_update_inplace(emb, a_e, ids, rows, lr)
        o_optim.sparse_adagrad_update_inplace(bias, a_b, ids, gb[:, None], lr)
    p = state.params
    acc = state.opt_state["sum_of_squares"]
    touched = np.unique(np.concatenate([b[0].reshape(-1) for b in batches]))
    errs = [_row_err(p["_token_embedding"]["embedding"], emb, touched),
            _row_err(acc["_token_embedding"]["embedding"], a_e, touched),
            _row_err(acc["_bias"]["embedding"], a_b, touched)]
    # a bias is ONE number per row (|b| ~ 0.05, some ~ 1e-5): bound its error by the table's scale, not its own size
    bias_err = rel_err(p["_bias"]["embedding"][touched].double().cpu().numpy(), bias[touched])
    print("one-pass GloVe step (%s), C3 table, B = %d %s, %d steps: worst loss error %.2e; longest run of one token %d; "
          "touched rows %d; per-row / norm-wise error table %.2e / %.2e, its accumulator %.2e / %.2e, bias accumulator "
          "%.2e / %.2e, bias table (norm-wise) %.2e"
          % (mode, B, kind, steps, worst_loss, longest, touched.size, *errs[0], *errs[1], *errs[2], bias_err))
    assert max(e[0] for e in errs) <= TOL and bias_err <= TOL
    assert _untouched_equal(p["_token_embedding"]["embedding"], emb0, touched)
    assert _untouched_equal(p["_bias"]["embedding"], bias0, touched)
    for name in ("_token_embedding", "_bias"):
        a = acc[name]["embedding"]
        assert _untouched_equal(a, torch.full_like(a, 0.1), touched)


# ---------------------------------------------------------------------------------------------------------------------
# bf16 tables (BASELINE configs[3]'s dtype: bf16 rows, fp32 accumulators) in the one-pass steps (round 6)
# ---------------------------------------------------------------------------------------------------------------------
# The oracle steps fp64 copies of the bf16-valued rows and rounds every touched row to the nearest bf16 after each step
# (oracle.optim.round_bf16): what a table that is rounded once per update holds.  The build steps the row in f32 and
# rounds on its one store; an f32 result within an f32 rounding of a bf16 midpoint lands on the neighbouring bf16 value
# (measured: 1 element in 10^5), and from then on the two trajectories of THAT row differ by a bf16 step: later updates
# of the row may round differently again, and a triplet whose margin lies within that difference of the hinge takes the
# other branch (its rows then differ by one occurrence's update -- a few steps).  Asserted: >= 99.99 % of all elements of
# all touched rows IDENTICAL to the oracle's, at most 2 rows in 10^4 with an element more than one bf16 step off (one
# step = 2^-7 of the larger of the element's old and new magnitude), none beyond 32 steps (a wrong row would be hundreds
# off).  The step normalisers / learning rates are chosen so that an update moves an element by several bf16 steps (at
# the configs' own 1 / B an update is far below one step and the comparison would have no teeth).
def _bf16_rows_check(table, want, start, rows, what):
    got = table[torch.as_tensor(rows, device=table.device)].float().cpu().numpy().astype(F64)
    w, s0 = want[rows], start[rows]
    step = np.maximum(np.abs(w), np.abs(s0)) * 2.0 ** -7 + 1e-30
    steps_off = np.abs(got - w) / step
    same = float(np.mean(got == w))
    moved = float(np.mean(w != s0))
    worst = float(steps_off.max())
    rows_off = float(np.mean(steps_off.max(1) > 1.0))
    print("%s: %d touched rows; elements moved by the steps %.1f %%; identical to the oracle's bf16 rows %.4f %%; rows "
          "with an element more than one bf16 step off %.4f %%; worst difference %.2f bf16 steps"
          % (what, len(rows), 100 * moved, 100 * same, 100 * rows_off, worst))
    assert moved > 0.5, "the updates must move most elements by at least one bf16 step"
    assert same >= 0.9999 and rows_off <= 2e-4 and worst <= 32.0
    return same


def _acc_err(acc, want, rows):
    g = acc[torch.as_tensor(rows, device=acc.device)].double().cpu().numpy()
    return float((np.abs(g - want[rows]).max(1) / np.maximum(np.abs(want[rows]).max(1), 1e-30)).max())


@pytest.mark.parametrize("B,steps,kind", [(8192, 12, "mixed"), (8192, 8, "zipf"), (262_144, 3, "mixed")])
def test_direct_triplet_bf16_tables_trajectory_at_c2_size_vs_fp64_oracle(dev, B, steps, kind):
    from esrecsys_amd import TrainState, ops, optim
    from esrecsys_amd.pinterest.models import STLModel
    from esrecsys_amd.pinterest.train_shop_the_look import fused_triplet_step_available, train_steps
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    V, D, lam, lr, norm = 1_000_000, 128, 0.1, 0.2, 64.0
    g = torch.Generator(device=dev).manual_seed(1701)

    def tower():
        t = torch.randn((V, D), generator=g, device=dev) * D ** -0.5
        t *= torch.where(torch.rand((V, 1), generator=g, device=dev) < 0.3, 0.6, 3.0)
        return t.to(torch.bfloat16)
    st, pt = tower(), tower()
    st0, pt0 = st.clone(), pt.clone()
    model = STLModel(output_size=D, num_scenes=V, num_products=V, device=dev)
    state = TrainState.create(apply_fn=model.apply, tx=optim.sparse_adagrad(lr),
                              params={"params": {"scene_tower": {"embedding": st}, "product_tower": {"embedding": pt}}})
    assert fused_triplet_step_available(state) and ops.triplet_direct_mode()
    rng = np.random.default_rng(B + steps + 1)
    if kind == "zipf":
        draw_s, draw_p = _zipf_sampler(V, rng), _zipf_sampler(V, rng)
    else:
        draw_s, draw_p = _mixed_sampler(V, 123_456, rng), _mixed_sampler(V, 654_321, rng)
    batches = [(draw_s(B), draw_p(B), draw_p(B)) for _ in range(steps)]
    es, ep = st.float().double().cpu().numpy(), pt.float().double().cpu().numpy()
    es0, ep0 = es.copy(), ep.copy()
    a_s, a_p = np.full_like(es, 0.1), np.full_like(ep, 0.1)
    dev_batches = [tuple(torch.as_tensor(x, device=dev) for x in b) for b in batches]
    state, losses = train_steps(state, iter(dev_batches), steps, lam, norm)
    losses = losses.cpu().numpy()
    for k, (sid, pid, nid) in enumerate(batches):
        el, gs, gp, gn = o_stl.triplet_loss_and_grads(es[sid], ep[pid], ep[nid], lam, norm, F64)
        assert abs(float(losses[k]) - el) <= 2e-5 * abs(el), (k, float(losses[k]), el)
        o_optim.sparse_adagrad_update_inplace(es, a_s, sid, gs, lr)
        pn = np.concatenate([pid, nid])
        o_optim.sparse_adagrad_update_inplace(ep, a_p, pn, np.concatenate([gp, gn]), lr)
        us, up = np.unique(sid), np.unique(pn)
        es[us] = o_optim.round_bf16(es[us])
        ep[up] = o_optim.round_bf16(ep[up])
    p_ = state.params["params"]
    acc = state.opt_state["sum_of_squares"]["params"]
    assert p_["scene_tower"]["embedding"].dtype == torch.bfloat16
    ts = np.unique(np.concatenate([b[0] for b in batches]))
    tp = np.unique(np.concatenate([np.concatenate(b[1:]) for b in batches]))
    _bf16_rows_check(p_["scene_tower"]["embedding"], es, es0, ts, "direct triplet step, bf16 scene tower, B = %d %s" % (B, kind))
    _bf16_rows_check(p_["product_tower"]["embedding"], ep, ep0, tp, "direct triplet step, bf16 product tower, B = %d %s" % (B, kind))
    aerr = max(_acc_err(acc["scene_tower"]["embedding"], a_s, ts), _acc_err(acc["product_tower"]["embedding"], a_p, tp))
    print("accumulators: worst per-row error %.2e" % aerr)
    assert aerr <= 1e-4   # (a row one bf16 step off feeds gradients 2^-8 off into later steps' accumulators)
    assert _untouched_equal(p_["scene_tower"]["embedding"], st0, ts)
    assert _untouched_equal(p_["product_tower"]["embedding"], pt0, tp)
    for t, rows in (("scene_tower", ts), ("product_tower", tp)):
        a = acc[t]["embedding"]
        assert a.dtype == torch.float32 and _untouched_equal(a, torch.full_like(a, 0.1), rows)


@pytest.mark.parametrize("B,steps,kind,lr", [(65_536, 4, "mixed", 200.0), (2048, 12, "mixed", 8.0), (2048, 8, "zipf", 8.0)])
def test_glove_one_pass_bf16_table_trajectory_at_c3_size_vs_fp64_oracle(dev, B, steps, kind, lr):
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.wikipedia.models import Glove
    from esrecsys_amd.wikipedia.train_cooccurence import fused_step_available, train_epoch
    from oracle import glove as o_glove
    from oracle import optim as o_optim
    V, D, mode = 465_537, 256, "reference"
    model = Glove(num_embeddings=V, features=D, loss_mode=mode, device=dev)
    params = model.init(1701, None)["params"]
    g = torch.Generator(device=dev).manual_seed(7)
    params["_bias"]["embedding"].copy_(torch.randn((V, 1), generator=g, device=dev) * 0.05)
    params["_token_embedding"]["embedding"] = params["_token_embedding"]["embedding"].to(torch.bfloat16)
    state = TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(lr))
    assert fused_step_available(state)
    emb0t, bias0 = params["_token_embedding"]["embedding"].clone(), params["_bias"]["embedding"].clone()
    emb, bias = emb0t.float().double().cpu().numpy(), bias0.double().cpu().numpy()
    emb0 = emb.copy()
    a_e, a_b = np.full_like(emb, 0.1), np.full_like(bias, 0.1)
    rng = np.random.default_rng(B + steps + 2)
    draw = _zipf_sampler(V, rng) if kind == "zipf" else _mixed_sampler(V, 222_222, rng)
    batches = [(draw((2, B)), np.exp(rng.uniform(np.log(0.1), np.log(1000.0), B)).astype(np.float32))
               for _ in range(steps)]
    dev_batches = [(torch.as_tensor(i, device=dev), torch.as_tensor(t, device=dev)) for i, t in batches]
    got = []
    state, mean_loss = train_epoch(state, steps, iter(dev_batches), losses_out=got)
    losses = got[0].cpu().numpy()
    for k, (inputs, target) in enumerate(batches):
        el, gdot, gs = o_glove.loss_and_grads(emb, bias, inputs, target.astype(F64), mode, F64)
        ids, rows, gb = o_glove.row_grads(emb, inputs, gdot, gs, F64)
        assert abs(float(losses[k]) - el) <= 2e-5 * abs(el), (k, float(losses[k]), el)
        o_optim.sparse_adagrad_update_inplace(emb, a_e, ids, rows, lr)
        o_optim.sparse_adagrad_update_inplace(bias, a_b, ids, gb[:, None], lr)
        u = np.unique(ids)
        emb[u] = o_optim.round_bf16(emb[u])
    p = state.params
    acc = state.opt_state["sum_of_squares"]
    assert p["_token_embedding"]["embedding"].dtype == torch.bfloat16
    touched = np.unique(np.concatenate([b[0].reshape(-1) for b in batches]))
    _bf16_rows_check(p["_token_embedding"]["embedding"], emb, emb0, touched,
                     "one-pass GloVe step, bf16 table, B = %d %s" % (B, kind))
    aerr = _acc_err(acc["_token_embedding"]["embedding"], a_e, touched)
    bias_err = rel_err(p["_bias"]["embedding"][touched].double().cpu().numpy(), bias[touched])
    print("accumulator: worst per-row error %.2e; bias table (fp32, norm-wise) %.2e" % (aerr, bias_err))
    assert aerr <= 1e-4 and bias_err <= 1e-4
    assert _untouched_equal(p["_token_embedding"]["embedding"], emb0t, touched)
    assert _untouched_equal(p["_bias"]["embedding"], bias0, touched)
    for name in ("_token_embedding", "_bias"):
        a = acc[name]["embedding"]
        assert _untouched_equal(a, torch.full_like(a, 0.1), touched)


def test_inbatch_one_call_step_bf16_tables_trajectory_at_c2_size_vs_fp64_oracle(dev):
    """The one-call in-batch train step on bf16 towers (the `inbatch_c2_bf16_tables` bench leg: one fp16 plane per operand,
    S recomputed by pass C, merges inside the sparse-Adagrad update, RNE-to-bf16 row writes): 14 steps at V = 1 M,
    B = 8192 against the fp64 oracle with bf16-rounded table writes."""
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.pinterest.models import STLModel
    from esrecsys_amd.pinterest.train_shop_the_look import train_steps
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    V, D, B, steps, lam, lr, scale, norm, W = 1_000_000, 128, 8192, 14, 0.1, 0.2, 8.0, 64.0, 20_000
    g = torch.Generator(device=dev).manual_seed(1701)
    st = (torch.randn((V, D), generator=g, device=dev) * D ** -0.5).to(torch.bfloat16)
    pt = (torch.randn((V, D), generator=g, device=dev) * D ** -0.5).to(torch.bfloat16)
    st0, pt0 = st.clone(), pt.clone()
    rng = np.random.default_rng(4)
    draw_s, draw_p = _mixed_sampler(V, 123_456, rng), _mixed_sampler(V, 654_321, rng)
    batches = [(draw_s(B), draw_p(B)) for _ in range(steps)]
    es, ep = st.float().double().cpu().numpy(), pt.float().double().cpu().numpy()
    es0, ep0 = es.copy(), ep.copy()
    a_s, a_p = np.full_like(es, 0.1), np.full_like(ep, 0.1)
    model = STLModel(output_size=D, num_scenes=V, num_products=V, device=dev)
    state = TrainState.create(apply_fn=model.apply, tx=optim.sparse_adagrad(lr),
                              params={"params": {"scene_tower": {"embedding": st}, "product_tower": {"embedding": pt}}})
    dev_batches = [(torch.as_tensor(a, device=dev), torch.as_tensor(b, device=dev), None) for a, b in batches]
    state, losses = train_steps(state, iter(dev_batches), steps, lam, norm, scale=scale)
    losses = losses.cpu().numpy()
    for k, (sid, pid) in enumerate(batches):
        el, _, gq, gc = o_stl.inbatch_softmax_loss_and_grads(es[sid], ep[pid], lam, norm, scale, F64)
        assert abs(float(losses[k]) - el) <= 2e-5 * abs(el), (k, float(losses[k]), el)
        o_optim.sparse_adagrad_update_inplace(es, a_s, sid, gq, lr)
        o_optim.sparse_adagrad_update_inplace(ep, a_p, pid, gc, lr)
        us, up = np.unique(sid), np.unique(pid)
        es[us] = o_optim.round_bf16(es[us])
        ep[up] = o_optim.round_bf16(ep[up])
    p = state.params["params"]
    acc = state.opt_state["sum_of_squares"]["params"]
    ts, tp = np.unique(np.concatenate([b[0] for b in batches])), np.unique(np.concatenate([b[1] for b in batches]))
    _bf16_rows_check(p["scene_tower"]["embedding"], es, es0, ts, "in-batch one-call step, bf16 scene tower")
    _bf16_rows_check(p["product_tower"]["embedding"], ep, ep0, tp, "in-batch one-call step, bf16 product tower")
    aerr = max(_acc_err(acc["scene_tower"]["embedding"], a_s, ts), _acc_err(acc["product_tower"]["embedding"], a_p, tp))
    print("accumulators: worst per-row error %.2e" % aerr)
    assert aerr <= 1e-4
    assert _untouched_equal(p["scene_tower"]["embedding"], st0, ts)
    assert _untouched_equal(p["product_tower"]["embedding"], pt0, tp)
