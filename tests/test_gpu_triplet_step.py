"""GPU: the one-pass Shop-The-Look train step (esr_triplet_train_step: plan + update + long-run combine on
double-buffered towers, gradients formed on chip) against (a) the six-launch path esr_triplet_fwd_bwd + sort +
esr_sparse_adagrad_scatter_multi and (b) the fp64 oracle of pinterest/train_shop_the_look.py:93-109 + sparse Adagrad."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ids(kind, V, n, rng):
    if kind == "uniform":
        return rng.integers(0, V, n).astype(np.int32)
    if kind == "same":
        return np.full(n, 3 % V, np.int32)
    w = 1.0 / np.arange(1, V + 1)
    return rng.permutation(V)[rng.choice(V, size=n, p=w / w.sum())].astype(np.int32)


def _state(Vs, Vp, D, dev, seed=2, lr=0.05, scale=1.0):
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.pinterest.models import STLModel
    stl = STLModel(output_size=D, num_scenes=Vs, num_products=Vp, device=dev)
    params = stl.init(seed)
    if scale != 1.0:  # rows with |e| > 1 so that the regulariser and its gradient are live
        for t in ("scene_tower", "product_tower"):
            params["params"][t]["embedding"].mul_(scale)
    return TrainState.create(apply_fn=stl.apply, params=params, tx=optim.sparse_adagrad(lr))


def _tables(state):
    p = state.params["params"]
    return p["scene_tower"]["embedding"], p["product_tower"]["embedding"]


@pytest.mark.parametrize("kind", ["uniform", "zipf", "same"])
@pytest.mark.parametrize("Vs,Vp,D,B", [(5000, 7000, 128, 8192), (300, 200, 32, 128), (2000, 1000, 96, 1000),
                                       (50, 60, 6, 33), (40000, 30000, 64, 30000)])
def test_fused_triplet_step_equals_six_launch_path(dev, monkeypatch, kind, Vs, Vp, D, B):
    from esrecsys_amd.pinterest.train_shop_the_look import fused_triplet_step_available, train_step
    rng = np.random.default_rng(Vs + B)
    a, b = _state(Vs, Vp, D, dev, scale=3.0), _state(Vs, Vp, D, dev, scale=3.0)
    assert fused_triplet_step_available(a)
    lam = 0.1
    for step in range(3):
        sid, pid, nid = _ids(kind, Vs, B, rng), _ids(kind, Vp, B, rng), _ids("uniform", Vp, B, rng)
        monkeypatch.setenv("ESR_STL_FUSED", "1")
        a, la = train_step(a, sid, pid, nid, lam, B)
        monkeypatch.setenv("ESR_STL_FUSED", "0")
        b, lb = train_step(b, sid, pid, nid, lam, B)
        assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb)), (step, float(la), float(lb))
    from esrecsys_amd import ops
    versions = a.versions
    if ops.triplet_direct_mode():  # rows stepped in place: nothing is double-buffered
        assert not versions
    else:
        assert len(versions) == 2 and all(v.dirty for v in versions.values())
    assert not b.versions
    (sa, pa), (sb, pb) = _tables(a), _tables(b)
    assert not any(v.dirty for v in versions.values()) and all(int(v.loc.sum()) == 0 for v in versions.values())
    assert int(a.step) == int(b.step) == 3
    assert rel_err(sa.cpu().numpy(), sb.cpu().numpy()) <= 1e-6 and rel_err(pa.cpu().numpy(), pb.cpu().numpy()) <= 1e-6
    # (runs of thousands of occurrences of ONE row -- "same" -- are summed in a different fixed association by the two
    # paths: both are f32 roundings of the same sum of ~10^4 terms, and the accumulator squares it)
    acc_tol = 4e-6 if kind == "same" else 1e-6
    for t in ("scene_tower", "product_tower"):
        assert rel_err(a.opt_state["sum_of_squares"]["params"][t]["embedding"].cpu().numpy(),
                       b.opt_state["sum_of_squares"]["params"][t]["embedding"].cpu().numpy()) <= acc_tol


def test_fused_triplet_trajectory_vs_fp64_oracle(dev):
    from esrecsys_amd.pinterest.train_shop_the_look import eval_step, train_step
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    Vs, Vp, D, B, lam, lr = 400, 600, 32, 256, 0.1, 0.05
    state = _state(Vs, Vp, D, dev, lr=lr, scale=2.5)
    st, pt = (t.cpu().numpy().astype(np.float64) for t in _tables(state))
    a_s, a_p = np.full_like(st, 0.1), np.full_like(pt, 0.1)
    rng = np.random.default_rng(11)
    for step in range(4):
        kind = "zipf" if step % 2 else "uniform"
        sid, pid, nid = _ids(kind, Vs, B, rng), _ids(kind, Vp, B, rng), _ids("uniform", Vp, B, rng)
        state, loss = train_step(state, sid, pid, nid, lam, B)
        el, gs, gp, gn = o_stl.triplet_loss_and_grads(st[sid], pt[pid], pt[nid], lam, B, np.float64)
        assert abs(float(loss) - el) <= 1e-5 * abs(el)
        st, a_s = o_optim.sparse_adagrad_update(st, a_s, sid, gs, lr, dtype=np.float64)
        pt, a_p = o_optim.sparse_adagrad_update(pt, a_p, np.concatenate([pid, nid]), np.concatenate([gp, gn]), lr,
                                                dtype=np.float64)
    gs_, gp_ = _tables(state)
    assert rel_err(gs_.cpu().numpy(), st) <= 1e-5 and rel_err(gp_.cpu().numpy(), pt) <= 1e-5
    # eval_step after fused steps reads the consolidated towers (train_shop_the_look.py:111-122)
    sid, pid, nid = _ids("uniform", Vs, B, rng), _ids("uniform", Vp, B, rng), _ids("uniform", Vp, B, rng)
    want = np.maximum(1.0 + (st[sid] * pt[nid]).sum(1) - (st[sid] * pt[pid]).sum(1), 0.0).sum()
    assert abs(float(eval_step(state, sid, pid, nid)) - want) <= 1e-5 * abs(want)


def test_presorted_triplets_equal_inline_sort(dev):
    """the sort of batch k + 1 on the side stream gives the same step as sorting in line"""
    from esrecsys_amd.pinterest.train_shop_the_look import presort_triplets, train_step
    Vs, Vp, D, B = 3000, 2000, 128, 4096
    rng = np.random.default_rng(4)
    a, b = _state(Vs, Vp, D, dev, scale=2.0), _state(Vs, Vp, D, dev, scale=2.0)
    batches = [(_ids("uniform", Vs, B, rng), _ids("zipf", Vp, B, rng), _ids("uniform", Vp, B, rng)) for _ in range(4)]
    ahead = presort_triplets(a, *batches[0])
    for k, bt in enumerate(batches):
        cur = ahead
        ahead = presort_triplets(a, *batches[k + 1]) if k + 1 < len(batches) else None
        a, la = train_step(a, cur, None, None, 0.1, B)
        b, lb = train_step(b, *bt, 0.1, B)
        assert float(la) == float(lb)
    for ta, tb in zip(_tables(a), _tables(b)):
        assert torch.equal(ta, tb)


def test_config_c2_triplet_full_size_fused_step(dev):
    """BASELINE configs[1] tables (1 M x 128 per tower), B = 8192, the reference's own loss: fused == six-launch path to
    an f32 rounding; rows no triplet touches keep their bits."""
    import os
    from esrecsys_amd.pinterest.train_shop_the_look import train_step
    V, D, B = 1_000_000, 128, 8192
    a, b = _state(V, V, D, dev, scale=1.2), _state(V, V, D, dev, scale=1.2)
    before = _tables(a)[1].clone()
    g = torch.Generator(device=dev).manual_seed(5)
    touched = torch.zeros(V, dtype=torch.bool, device=dev)
    for _ in range(2):
        ids = torch.randint(0, V, (3, B), generator=g, device=dev, dtype=torch.int32)
        os.environ["ESR_STL_FUSED"] = "1"
        a, la = train_step(a, ids[0].contiguous(), ids[1].contiguous(), ids[2].contiguous(), 0.1, B)
        os.environ["ESR_STL_FUSED"] = "0"
        try:
            b, lb = train_step(b, ids[0].contiguous(), ids[1].contiguous(), ids[2].contiguous(), 0.1, B)
        finally:
            os.environ["ESR_STL_FUSED"] = "1"
        touched[ids[1].long()] = True
        touched[ids[2].long()] = True
        assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb))
    (sa, pa), (sb, pb) = _tables(a), _tables(b)
    assert rel_err(sa.cpu().numpy(), sb.cpu().numpy()) <= 1e-6 and rel_err(pa.cpu().numpy(), pb.cpu().numpy()) <= 1e-6
    assert torch.equal(pa[~touched], before[~touched]) and not torch.equal(pa[touched], before[touched])


@pytest.mark.parametrize("Vs,Vp,D,B,steps,kind", [(200_000, 300_000, 128, 65_536, 4, "uniform"),
                                                  (1_000_000, 1_000_000, 128, 262_144, 2, "uniform"),
                                                  (4000, 6000, 64, 8192, 6, "uniform"), (3000, 3000, 128, 2048, 6, "zipf")])
def test_direct_step_equals_stamped_step_and_six_launch_path(dev, monkeypatch, Vs, Vp, D, B, steps, kind):
    """The direct step (one row group per TRIPLET, rows stepped in place; duplicated rows by the arrival that completes
    their run, through gradient rows parked at the memory side) against the stamped walk over the sorted occurrences and
    against the six-launch path: the three share trip_grad / adagrad_elem and the association of every row sum, and differ
    only in how many lanes reduce a dot product -- towers and accumulators agree to an f32 rounding (1e-6 of the largest
    entry; one Adagrad step moves a row by ~lr = 5e-2 of it: a gradient row read stale, or missed, by the arrival that
    completes a run would be off by four orders of magnitude more).  Batches where 2 - 60 % of the occurrences share
    their row with another triplet of the batch, in another workgroup."""
    from esrecsys_amd.pinterest.train_shop_the_look import train_step
    rng = np.random.default_rng(B + steps)
    batches = [(_ids(kind, Vs, B, rng), _ids(kind, Vp, B, rng), _ids("uniform", Vp, B, rng)) for _ in range(steps)]
    outs = []
    for mode, fused in (("direct", "1"), ("stamped", "1"), ("direct", "0")):
        monkeypatch.setenv("ESR_TRIPLET_STEP", mode)
        monkeypatch.setenv("ESR_STL_FUSED", fused)
        st = _state(Vs, Vp, D, dev, scale=3.0)
        losses = []
        for sid, pid, nid in batches:
            st, l = train_step(st, sid, pid, nid, 0.1, B)
            losses.append(float(l))
        sa, pa = _tables(st)
        acc = st.opt_state["sum_of_squares"]["params"]
        outs.append((losses, sa.clone(), pa.clone(), acc["scene_tower"]["embedding"].clone(),
                     acc["product_tower"]["embedding"].clone()))
        del st
    for other in outs[1:]:
        for x, y in zip(outs[0][1:], other[1:]):
            assert rel_err(x.cpu().numpy(), y.cpu().numpy()) <= 1e-6
        for la, lb in zip(outs[0][0], other[0]):
            assert abs(la - lb) <= 2e-6 * abs(lb)


def _plan_into(buf, ids3, Vs, gen, dev, Vp=2500):
    """esr_triplet_plan of ONE batch into an existing (zeroed-once) plan buffer; returns (sorted, perm) for the step."""
    import ctypes
    from esrecsys_amd import _lib, ops
    sid, pid, nid = ids3
    B = sid.numel()
    srt, prm = ops.segment_sort_multi([sid, pid, nid], [0, Vs, Vs], Vs + Vp)
    ptrs = (ctypes.c_void_p * 3)(sid.data_ptr(), pid.data_ptr(), nid.data_ptr())
    _lib.check(_lib.load().esr_triplet_plan(ptrs, 1, B, int(Vs), srt.data_ptr(), prm.data_ptr(), buf.data_ptr(), None,
                                            int(gen), ops._stream()), "esr_triplet_plan")
    return srt, prm


def test_direct_plan_generation_is_32_bits_and_a_plan_feeds_repeated_steps(dev, monkeypatch):
    """(1) A plan buffer planned at generation g and again at g + 4096 (a run of more than 4096 groups of train_steps:
    the buffers are reused for the whole run) -- the long-run count of the first plan must not be continued by the
    second (rounds 3-4 kept a 12-bit tag: the stale heads then got an extra Adagrad step from whatever the side buffer
    held).  (2) The arrival that completes a run clears its counter, so a direct plan can feed the same batch's step
    twice.  Both against steps on fresh plans, bit for bit; Zipf ids (runs of 2-8 AND long runs)."""
    from esrecsys_amd import ops
    monkeypatch.setenv("ESR_TRIPLET_STEP", "direct")
    Vs, Vp, D, B = 3000, 2500, 128, 2048
    rng = np.random.default_rng(77)

    def batch():
        return tuple(torch.as_tensor(x, device=dev) for x in
                     (_ids("zipf", Vs, B, rng), _ids("zipf", Vp, B, rng), _ids("uniform", Vp, B, rng)))

    def towers():
        g = torch.Generator(device=dev).manual_seed(5)
        t = [torch.randn((V, D), generator=g, device=dev) * 0.3 for V in (Vs, Vp)]
        return t[0], torch.full_like(t[0], 0.1), t[1], torch.full_like(t[1], 0.1)

    def step(tw, ids3, srt, prm, plan):
        return ops.triplet_train_step(tw[0], None, None, tw[1], tw[2], None, None, tw[3], *ids3, 0.1, float(B), 0.05,
                                      presorted=(srt, prm), plan=plan, long_runs=-1)
    b1, b2 = batch(), batch()
    pb = ops._ws_bytes("esr_triplet_plan_bytes", B)
    for g1, g2 in ((7, 7 + 4096), (1, 1 + (1 << 20)), (4095, 4096)):
        X, Y = towers(), towers()
        buf = ops._aligned_bytes(pb, dev)
        lx = [step(X, b1, *_plan_into(buf, b1, Vs, g1, dev), buf)]
        lx.append(step(X, b2, *_plan_into(buf, b2, Vs, g2, dev), buf))
        ly = []
        for b, gen in ((b1, 3), (b2, 4)):
            fresh = ops._aligned_bytes(pb, dev)
            ly.append(step(Y, b, *_plan_into(fresh, b, Vs, gen, dev), fresh))
        assert all(torch.equal(a, b) for a, b in zip(lx, ly)), (g1, g2)
        assert all(torch.equal(a, b) for a, b in zip(X, Y)), (g1, g2)
    # (2) one plan, two steps of the same batch
    X, Y = towers(), towers()
    buf = ops._aligned_bytes(pb, dev)
    srt, prm = _plan_into(buf, b1, Vs, 9, dev)
    lx = [step(X, b1, srt, prm, buf), step(X, b1, srt, prm, buf)]
    ly = []
    for gen in (1, 2):
        fresh = ops._aligned_bytes(pb, dev)
        ly.append(step(Y, b1, *_plan_into(fresh, b1, Vs, gen, dev), fresh))
    assert all(torch.equal(a, b) for a, b in zip(lx, ly)) and float(lx[0]) != float(lx[1])
    assert all(torch.equal(a, b) for a, b in zip(X, Y))
