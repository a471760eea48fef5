"""GPU: the row-sharded step through RCCL with a world of one rank (the driver owns the 8-GPU runs):
exercises bucket / all_to_all_single / un-permute / grad routing with the real kernels and checks it is
identical to the single-device path."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg(dev):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # these tests are about the exchange MACHINERY at world 1 (bucket, self-exchange, gather, gradient rows, owner-side
    # update); the default at world 1 is the single-GPU step on the shard (test_world1_direct_path... below switches back)
    os.environ["ESR_SHARDED_WORLD1_DIRECT"] = "0"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dist
    os.environ.pop("ESR_SHARDED_WORLD1_DIRECT", None)
    from esrecsys_amd import rccl
    rccl.reset()
    dist.destroy_process_group()


def test_sharded_world1_equals_single_device(dev, pg):
    from esrecsys_amd import TrainState, ops, optim, sharded
    from esrecsys_amd.pinterest.models import STLModel
    from esrecsys_amd.pinterest.train_shop_the_look import train_step
    Vs, Vp, D, B, lam, lr = 5000, 7000, 128, 1024, 0.1, 0.05
    stl = STLModel(output_size=D, num_scenes=Vs, num_products=Vp, device=dev)
    params = stl.init(0)
    state = TrainState.create(apply_fn=stl.apply, params=params, tx=optim.sparse_adagrad(lr))
    st = params["params"]["scene_tower"]["embedding"].clone()
    pt = params["params"]["product_tower"]["embedding"].clone()
    scene = sharded.RowShardedTable(st, torch.full_like(st, 0.1), Vs)
    prod = sharded.RowShardedTable(pt, torch.full_like(pt, 0.1), Vp)
    towers = sharded.ShardedTableGroup([scene, prod], kernels=ops)
    rng = np.random.default_rng(0)
    for step in range(3):
        sid, pid, nid = (torch.from_numpy(rng.integers(0, n, B).astype(np.int32)).to(dev) for n in (Vs, Vp, Vp))
        if step % 2 == 0:
            l_sh = sharded.sharded_triplet_step(towers, sid, pid, nid, lam, float(B), lr)
            state, l_1 = train_step(state, sid, pid, nid, lam, B)
        else:
            l_sh = sharded.sharded_inbatch_step(towers, sid, pid, lam, float(B), 4.0, lr)
            state, l_1 = train_step(state, sid, pid, None, lam, B, scale=4.0)
        assert abs(float(l_sh) - float(l_1)) <= 1e-6 * abs(float(l_1))
    # same occurrence order, same element arithmetic; the single-device triplet step is the one-pass kernel (its dots are
    # reduced over 8 lanes, the sharded path's over 32): tables agree to an f32 rounding
    from conftest import rel_err
    assert rel_err(scene.local.cpu().numpy(), state.params["params"]["scene_tower"]["embedding"].cpu().numpy()) <= 1e-6
    assert rel_err(prod.local.cpu().numpy(), state.params["params"]["product_tower"]["embedding"].cpu().numpy()) <= 1e-6


def test_sharded_glove_world1(dev, pg):
    from esrecsys_amd import ops, sharded
    from oracle import glove as o_glove
    V, D, B = 3000, 64, 512
    g = torch.Generator().manual_seed(1)
    emb0 = (torch.randn((V, D), generator=g) * D ** -0.5)
    bias0 = torch.randn((V, 1), generator=g) * 0.05
    emb_t = sharded.RowShardedTable(emb0.to(dev), torch.full((V, D), 0.1, device=dev), V)
    emb = sharded.ShardedTableGroup([emb_t], kernels=ops)
    bias = sharded.ShardedTableGroup([sharded.RowShardedTable(bias0.to(dev), torch.full((V, 1), 0.1, device=dev), V)],
                                     kernels=ops)
    rng = np.random.default_rng(2)
    inputs = rng.integers(0, V, (2, B)).astype(np.int32)
    target = rng.uniform(0.1, 300, B).astype(np.float32)
    loss = sharded.sharded_glove_step(emb, bias, torch.from_numpy(inputs).to(dev), torch.from_numpy(target).to(dev),
                                      ops.GLOVE_REFERENCE, 0.05)
    el, _, _ = o_glove.loss_and_grads(emb0.numpy().astype(np.float64), bias0.numpy().astype(np.float64), inputs,
                                      target, "reference", np.float64)
    assert abs(float(loss) - el) / abs(el) <= 1e-5
    assert not torch.equal(emb_t.local.cpu(), emb0)  # rows moved


def test_sharded_bf16_towers_world1(dev, pg):
    """BASELINE config 4 shape in miniature: bf16 tables + fp32 accumulators, row-sharded in-batch step.
    Rows cross the exchange as bf16 and are widened exactly; the update rounds to bf16 (RNE)."""
    from esrecsys_amd import ops, sharded
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    Vs, Vp, D, B, lam, lr, scale = 4000, 6000, 128, 256, 0.1, 0.05, 4.0
    g = torch.Generator().manual_seed(9)
    st0 = (torch.randn((Vs, D), generator=g) * 0.12).to(torch.bfloat16)
    pt0 = (torch.randn((Vp, D), generator=g) * 0.12).to(torch.bfloat16)
    scene = sharded.RowShardedTable(st0.to(dev), torch.full((Vs, D), 0.1, device=dev), Vs)
    prod = sharded.RowShardedTable(pt0.to(dev), torch.full((Vp, D), 0.1, device=dev), Vp)
    towers = sharded.ShardedTableGroup([scene, prod], kernels=ops)
    rng = np.random.default_rng(4)
    sid = rng.integers(0, Vs, B).astype(np.int32)
    pid = rng.integers(0, Vp, B).astype(np.int32)
    loss = sharded.sharded_inbatch_step(towers, torch.from_numpy(sid).to(dev), torch.from_numpy(pid).to(dev), lam,
                                        float(B), scale, lr)
    q, c = st0.float().numpy()[sid].astype(np.float64), pt0.float().numpy()[pid].astype(np.float64)
    el, _, gq, gc = o_stl.inbatch_softmax_loss_and_grads(q, c, lam, B, scale, np.float64)
    assert abs(float(loss) - el) / abs(el) <= 1e-5
    es, ea = o_optim.sparse_adagrad_update(st0.float().numpy().astype(np.float64), np.full((Vs, D), 0.1), sid, gq, lr,
                                           dtype=np.float64)
    got = scene.local.float().cpu().numpy()
    exp_bf16 = torch.from_numpy(es).to(torch.bfloat16).float().numpy()
    assert scene.local.dtype == torch.bfloat16 and np.mean(got == exp_bf16) > 0.999
    assert np.abs(scene.accum.cpu().numpy() - ea).max() <= 1e-5 * np.abs(ea).max()


@pytest.mark.parametrize("workload", ["inbatch", "triplet"])
def test_sharded_routing_plans_made_in_groups(dev, pg, workload):
    """begin_plans with several lookups (bench_sharded.py's ESR_SHARDED_PLAN_GROUP): the plans of six batches made
    together -- one counts exchange, one host wait -- give the same tables and losses, bit for bit, as six plans made
    one at a time."""
    from esrecsys_amd import ops, sharded
    V, D, B, lam, lr, steps = 20000, 128, 1024, 0.1, 0.05, 6
    g = torch.Generator().manual_seed(9)
    t0 = torch.randn((V, D), generator=g) * D ** -0.5
    t1 = torch.randn((V, D), generator=g) * D ** -0.5
    rng = np.random.default_rng(10)
    batches = [[torch.from_numpy(rng.integers(0, V, B).astype(np.int32)).to(dev) for _ in range(3)] for _ in range(steps)]

    def run(grouped):
        tabs = [sharded.RowShardedTable(t.clone().to(dev), torch.full(t.shape, 0.1, device=dev), V) for t in (t0, t1)]
        towers = sharded.ShardedTableGroup(tabs, kernels=ops)
        segs = (lambda b: (b, [0, 1, 1])) if workload == "triplet" else (lambda b: (b[:2], [0, 1]))
        plans = None
        if grouped:
            plans = sharded.begin_plans([(towers, towers.virtual_id_segments(*segs(b))) for b in batches]).finish()
        losses = []
        for i, b in enumerate(batches):
            plan = plans[i] if plans is not None else None
            if workload == "triplet":
                losses.append(sharded.sharded_triplet_step(towers, *b, lam, float(B), lr, plan=plan))
            else:
                losses.append(sharded.sharded_inbatch_step(towers, b[0], b[1], lam, float(B), 8.0, lr, plan=plan))
        torch.cuda.synchronize()
        return [t.local.clone() for t in tabs] + [t.accum.clone() for t in tabs], torch.stack([l.reshape(()) for l in losses])

    tabs_a, loss_a = run(True)
    tabs_b, loss_b = run(False)
    assert torch.equal(loss_a, loss_b)
    for a, b in zip(tabs_a, tabs_b):
        assert torch.equal(a, b)


@pytest.mark.parametrize("unique", ["0", "1"])
def test_world1_direct_path_and_machinery_agree(dev, pg, unique, monkeypatch):
    """A world of one rank: the default (single-GPU steps on the shard, no exchange) and the full machinery -- with
    per-occurrence or per-distinct-row exchange (ESR_SHARDED_UNIQUE) -- leave the same towers for the triplet, in-batch and
    GloVe steps, hot ids included."""
    from conftest import rel_err
    from esrecsys_amd import ops, sharded
    V, D, B, lam, lr = 6000, 128, 1024, 0.1, 0.05
    g = torch.Generator().manual_seed(3)
    t0 = torch.randn((V, D), generator=g) * D ** -0.5
    t1 = torch.randn((V, D), generator=g) * D ** -0.5
    e0, b0 = torch.randn((V, 64), generator=g) * 0.12, torch.randn((V, 1), generator=g) * 0.05
    rng = np.random.default_rng(5)

    def ids(n):  # 6 % of the occurrences on three hot rows: runs of ~20, longer than the triplet step's 8-position chunks
        # (with a third of a batch on three rows the f32 association noise of their 340-term gradient sums -- 8-position
        # chunks on one path, 32 on the other -- reaches 1e-4 in the next loss: every occurrence of the batch reads them)
        return torch.from_numpy(np.where(rng.random(n) < 0.06, rng.integers(0, 3, n), rng.integers(0, V, n)).astype(np.int32)).to(dev)
    batches = [(ids(B), ids(B), ids(B)) for _ in range(4)]
    gb = [(torch.stack([ids(B), ids(B)]).contiguous(), torch.from_numpy(rng.uniform(0.1, 300, B).astype(np.float32)).to(dev))
          for _ in range(3)]

    def run(direct):
        monkeypatch.setenv("ESR_SHARDED_WORLD1_DIRECT", "1" if direct else "0")
        monkeypatch.setenv("ESR_SHARDED_UNIQUE", unique)
        mk = lambda t: sharded.RowShardedTable(t.clone().to(dev), torch.full(t.shape, 0.1, device=dev), V)  # noqa: E731
        towers = sharded.ShardedTableGroup([mk(t0), mk(t1)], kernels=ops)
        emb, bias = sharded.ShardedTableGroup([mk(e0)], kernels=ops), sharded.ShardedTableGroup([mk(b0)], kernels=ops)
        assert towers.world1_direct == direct and (direct or towers.unique == (unique == "1"))
        losses = []
        for i, (s_, p_, n_) in enumerate(batches):
            if i % 2 == 0:
                losses.append(float(sharded.sharded_triplet_step(towers, s_, p_, n_, lam, float(B), lr)))
            else:
                losses.append(float(sharded.sharded_inbatch_step(towers, s_, p_, lam, float(B), 4.0, lr)))
        for inp, tgt in gb:
            losses.append(float(sharded.sharded_glove_step(emb, bias, inp, tgt, ops.GLOVE_REFERENCE, lr)))
        towers.consolidate(), emb.consolidate()
        return losses, [t.local.clone() for t in towers.tables + emb.tables + bias.tables]
    la, ta = run(True)
    lb, tb = run(False)
    # (the in-batch head sees its rows in a different order on the two paths -- id order against exchange order -- and
    # its split-precision exponent references follow the order: the same 1e-5 class as against the fp64 oracle)
    assert np.allclose(la, lb, rtol=2e-5), (la, lb)
    errs = [rel_err(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(ta, tb)]  # scene, product, GloVe emb, GloVe bias
    assert max(errs) <= 2e-5, errs


def test_world1_direct_mixed_steps_with_odd_batch(dev, pg, monkeypatch):
    """with B % 128 != 0 the in-batch step of a world-1 group takes the plan-based path, which reads and
    updates the PLAIN shards -- after a one-pass triplet step left rows in the second buffers.  Both paths (direct and
    machinery) must leave the same towers: the plan-based branches consolidate first."""
    from conftest import rel_err
    from esrecsys_amd import ops, sharded
    V, D, B, lam, lr = 5000, 128, 1000, 0.1, 0.05  # 1000 % 128 != 0
    g = torch.Generator().manual_seed(13)
    t0 = torch.randn((V, D), generator=g) * D ** -0.5
    t1 = torch.randn((V, D), generator=g) * D ** -0.5
    rng = np.random.default_rng(14)
    batches = [tuple(torch.from_numpy(rng.integers(0, V, B).astype(np.int32)).to(dev) for _ in range(3)) for _ in range(5)]

    def run(direct):
        monkeypatch.setenv("ESR_SHARDED_WORLD1_DIRECT", "1" if direct else "0")
        monkeypatch.setenv("ESR_SHARDED_UNIQUE", "0")
        mk = lambda t: sharded.RowShardedTable(t.clone().to(dev), torch.full(t.shape, 0.1, device=dev), V)  # noqa: E731
        towers = sharded.ShardedTableGroup([mk(t0), mk(t1)], kernels=ops)
        assert towers.world1_direct == direct
        losses = []
        for i, (s_, p_, n_) in enumerate(batches):
            if i % 2 == 0:
                losses.append(float(sharded.sharded_triplet_step(towers, s_, p_, n_, lam, float(B), lr)))
            else:
                losses.append(float(sharded.sharded_inbatch_step(towers, s_, p_, lam, float(B), 4.0, lr)))
        towers.consolidate()
        return losses, [t.local.clone() for t in towers.tables] + [t.accum.clone() for t in towers.tables]
    la, ta = run(True)
    lb, tb = run(False)
    assert np.allclose(la, lb, rtol=2e-5), (la, lb)
    errs = [rel_err(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(ta, tb)]
    assert max(errs) <= 2e-5, errs


def test_bf16_gradient_rows_narrow_and_widen_like_torch(dev):
    """The two conversions of the bf16 gradient exchange (esr_rows_f32_to_bf16 on the asker, esr_unpermute_rows_bf16_to_f32
    on the owner): round-to-nearest-even like torch's, NaN / inf / signed zero / subnormals kept, widening exact."""
    from esrecsys_amd import ops
    g = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn((1001, 128), generator=g, device=dev) * torch.exp(torch.randn((1001, 1), generator=g, device=dev) * 8)
    x[0, :8] = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1e-40, -1e-40, 3.3895314e38], device=dev)
    x[1, :4] = torch.tensor([1.00390625, 1.01171875, 1.0039062, 1.0039064], device=dev)  # ties: to even; just below / above
    got = ops.rows_f32_to_bf16(x)
    want = x.to(torch.bfloat16)
    assert torch.equal(got.view(torch.int16)[~torch.isnan(x)], want.view(torch.int16)[~torch.isnan(x)])
    assert bool(torch.isnan(got.float()[torch.isnan(x)]).all())
    back = ops.unpermute_rows_to_f32(got, None)
    assert torch.equal(back[~torch.isnan(x)], want.float()[~torch.isnan(x)])


@pytest.mark.parametrize("workload", ["triplet", "glove", "inbatch"])
def test_sharded_train_steps_equals_per_step_calls_on_the_gpu(dev, pg, workload):
    """sharded_train_steps (plans of a group of batches made together, the next group's ahead; every step through the
    esr_sharded_* library calls) against the per-step calls with in-line plans, world-1 machinery: the same tables, bit for
    bit."""
    from esrecsys_amd import ops, sharded
    V, D, B, K = 5000, 128, 1024, 11
    g = torch.Generator(device=dev).manual_seed(9)

    def groups():
        gg = torch.Generator(device=dev).manual_seed(1)
        def tab(d):
            t = torch.randn((V, d), generator=gg, device=dev) * d ** -0.5
            return sharded.RowShardedTable(t, torch.full((V, d), 0.1, device=dev), V)
        if workload == "glove":
            return (sharded.ShardedTableGroup([tab(D)], kernels=ops), sharded.ShardedTableGroup([tab(1)], kernels=ops))
        return (sharded.ShardedTableGroup([tab(D), tab(D)], kernels=ops),)
    if workload == "glove":
        batches = [(torch.randint(0, V, (2, B), generator=g, device=dev, dtype=torch.int32),
                    torch.exp(torch.rand(B, generator=g, device=dev) * 6 - 2)) for _ in range(K)]
    else:
        batches = [tuple(torch.randint(0, V, (B,), generator=g, device=dev, dtype=torch.int32) for _ in range(3))
                   for _ in range(K)]
    kw = dict(regularization=0.1, global_batch_size=float(B), scale=4.0, lr=0.05, mode=ops.GLOVE_REFERENCE)
    a = groups()
    la = sharded.sharded_train_steps(workload, a, batches, plan_group=4, **kw)
    b = groups()
    lb = []
    for bt in batches:
        if workload == "glove":
            lb.append(sharded.sharded_glove_step(b[0], b[1], bt[0], bt[1], ops.GLOVE_REFERENCE, 0.05))
        elif workload == "inbatch":
            lb.append(sharded.sharded_inbatch_step(b[0], bt[0], bt[1], 0.1, float(B), 4.0, 0.05))
        else:
            lb.append(sharded.sharded_triplet_step(b[0], bt[0], bt[1], bt[2], 0.1, float(B), 0.05))
    assert len(la) == K and torch.equal(torch.stack([x.reshape(()) for x in la]), torch.stack([x.reshape(()) for x in lb]))
    for ga, gb in zip(a, b):
        for ta, tb in zip(ga.tables, gb.tables):
            assert torch.equal(ta.local, tb.local) and torch.equal(ta.accum, tb.accum)


@pytest.mark.parametrize("calls", ["one", "ops"])
@pytest.mark.parametrize("unique", [False, True])
@pytest.mark.parametrize("workload", ["triplet", "glove", "inbatch"])
def test_overlapped_lookups_equal_the_sequential_loop_on_the_gpu(dev, pg, workload, unique, calls, monkeypatch):
    """sharded_train_steps(overlap=True) -- the next batch's gather + rows exchange on a side stream, issued before the
    current batch's loss kernel, its stale rows served again after the update -- against the same steps with
    every lookup made in line: the same tables and losses, bit for bit.  World-1 machinery: the side stream, the event
    hand-overs, the membership search of begin_stale_sets and the patch scatter are all the real ones.  Zipf-like ids: a
    third of every lookup names rows the step before it writes; the control (no patch) must differ.
    calls "one": every step ONE library call (esr_sharded_*_step_overlapped) against the one-call steps without overlap;
    "ops": lookup / patch / loss kernel / update issued from Python (the in-batch head's only form)."""
    from esrecsys_amd import ops, sharded
    if calls == "one" and workload == "inbatch":
        pytest.skip("the in-batch step has no one-call form")
    monkeypatch.setenv("ESR_SHARDED_OVERLAP_CALLS", calls)
    V, D, B, K = 5000, 128, 1024, 11
    g = torch.Generator(device=dev).manual_seed(19)

    def ids(*shape):  # (a few hot rows + a uniform tail)
        u = torch.rand(shape, generator=g, device=dev)
        return (u * u * u * V).to(torch.int32).clamp_(max=V - 1)

    def groups():
        gg = torch.Generator(device=dev).manual_seed(1)
        def tab(d):
            t = torch.randn((V, d), generator=gg, device=dev) * d ** -0.5
            return sharded.RowShardedTable(t, torch.full((V, d), 0.1, device=dev), V)
        if workload == "glove":
            return (sharded.ShardedTableGroup([tab(D)], kernels=ops, unique=unique),
                    sharded.ShardedTableGroup([tab(1)], kernels=ops, unique=unique))
        return (sharded.ShardedTableGroup([tab(D), tab(D)], kernels=ops, unique=unique),)
    if workload == "glove":
        batches = [(ids(2, B), torch.exp(torch.rand(B, generator=g, device=dev) * 6 - 2)) for _ in range(K)]
    else:
        batches = [tuple(ids(B) for _ in range(3)) for _ in range(K)]
    kw = dict(regularization=0.1, global_batch_size=float(B), scale=4.0, lr=0.05, mode=ops.GLOVE_REFERENCE)
    stack = lambda ls: torch.stack([x.reshape(()) for x in ls])  # noqa: E731

    a = groups()
    la = sharded.sharded_train_steps(workload, a, batches, plan_group=4, overlap=True, **kw)
    b = groups()
    if calls == "one":  # the one-call steps, every lookup in line
        lb = sharded.sharded_train_steps(workload, b, batches, plan_group=4, overlap=False, **kw)
    else:  # the sequential loop through the same kernels: plan, lookup, step on the looked-up rows
        lb = []
        for bt in batches:
            if workload == "glove":
                plan = sharded.plan_glove(b[0], bt[0])
                rows = (b[0].lookup_bucketed(plan), b[1].lookup_bucketed(plan))
                lb.append(sharded.sharded_glove_step(b[0], b[1], bt[0], bt[1], ops.GLOVE_REFERENCE, 0.05, plan=plan, rows=rows))
            elif workload == "inbatch":
                plan = sharded.plan_inbatch(b[0], bt[0], bt[1])
                lb.append(sharded.sharded_inbatch_step(b[0], bt[0], bt[1], 0.1, float(B), 4.0, 0.05, plan=plan,
                                                       rows=b[0].lookup_bucketed(plan)))
            else:
                plan = sharded.plan_triplet(b[0], *bt)
                lb.append(sharded.sharded_triplet_step(b[0], *bt, 0.1, float(B), 0.05, plan=plan,
                                                       rows=b[0].lookup_bucketed(plan)))
    torch.cuda.synchronize()
    assert len(la) == K and torch.equal(stack(la), stack(lb))
    for ga, gb in zip(a, b):
        for ta, tb in zip(ga.tables, gb.tables):
            assert torch.equal(ta.local, tb.local) and torch.equal(ta.accum, tb.accum)
    # control: without the stale rows served again, the early lookups change the result
    real = sharded.ShardedTableGroup.patch_rows
    real_struct = ops.step_overlap_struct

    def no_stale(backs, ready, stale, *rest):
        if stale is not None:
            zero = ops.i64_array([0])
            stale = (stale[0], zero, stale[2], zero)
        return real_struct(backs, ready, stale, *rest)
    monkeypatch.setattr(sharded.ShardedTableGroup, "patch_rows", lambda self, plan, back: back)
    monkeypatch.setattr(ops, "step_overlap_struct", no_stale)
    d = groups()
    sharded.sharded_train_steps(workload, d, batches, plan_group=4, overlap=True, **kw)
    monkeypatch.setattr(sharded.ShardedTableGroup, "patch_rows", real)
    monkeypatch.setattr(ops, "step_overlap_struct", real_struct)
    assert not all(torch.equal(ta.local, td.local) for ga, gd in zip(a, d) for ta, td in zip(ga.tables, gd.tables))


@pytest.mark.parametrize("L,n,m,G", [(1, 1, 1, 1), (8, 5000, 7000, 8), (3, 1024, 1, 2), (8, 100_003, 90_001, 8),
                                     (2, 2049, 4096, 64)])
def test_plan_phase_kernels_membership_and_stable_partition(dev, L, n, m, G):
    """esr_sorted_membership + esr_flagged_first (the overlapped loop's plan phase: which asked rows does the previous
    step update, served again asker by asker) against NumPy: flags, the flagged entries first in order, per-slice counts
    through a strided [G, L] view; values = None gives positions."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(L * n + m)
    sentinel = 1_000_000
    cur = rng.integers(0, 3 * m + 5, (L, n)).astype(np.int32)
    cur[:, -max(1, n // 7):] = sentinel                                  # padding
    seq = np.sort(rng.integers(0, 3 * m + 5, (L, m)).astype(np.int32), axis=1)
    seq[:, -max(1, m // 5):] = sentinel + 1                              # padding of the sorted lists
    flags = ops.sorted_membership(torch.from_numpy(cur).to(dev), torch.from_numpy(seq).to(dev), sentinel)
    want = np.stack([np.isin(cur[l], seq[l]) & (cur[l] != sentinel) for l in range(L)])
    assert np.array_equal(flags.cpu().numpy().astype(bool), want)
    cuts = np.sort(rng.integers(0, n + 1, (L, G - 1)), axis=1)
    lens = np.diff(np.concatenate([np.zeros((L, 1), np.int64), cuts, np.full((L, 1), n)], axis=1), axis=1).astype(np.int64)
    both = torch.full((2, G, L), -7, dtype=torch.int64, device=dev)
    for values, slot in ((cur, 0), (None, 1)):
        out = ops.flagged_first(flags, None if values is None else torch.from_numpy(values).to(dev),
                                torch.from_numpy(lens).to(dev), both[slot])
        got = out.cpu().numpy()
        for l in range(L):
            sel = (cur[l] if values is not None else np.arange(n, dtype=np.int32))[want[l]]
            assert np.array_equal(got[l, :sel.size], sel), (l, slot)
            e = np.concatenate([[0], np.cumsum(lens[l])])
            assert [int(x) for x in both[slot, :, l].cpu()] == [int(want[l, e[g]:e[g + 1]].sum()) for g in range(G)]


def test_ivf_build_kernels_run_offsets_and_centroids(dev):
    """esr_run_offsets (list boundaries + longest run from the sorted assignments, absent values included) and
    esr_ivf_centroids (unit vectors; empty lists take their fallback training row) against NumPy."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(5)
    for n, nv in ((0, 3), (1, 1), (10_000, 257), (100_001, 4096), (50, 1000)):
        a = np.sort(rng.integers(0, nv, n).astype(np.int32))
        if n > 10:
            a = a[a != a[n // 2]]                                    # a value of the middle goes missing entirely
        off, mx = ops.run_offsets(torch.from_numpy(a).to(dev), nv, want_max=True)
        want = np.searchsorted(a, np.arange(nv + 1), side="left").astype(np.int32)
        assert np.array_equal(off.cpu().numpy(), want)
        assert int(mx) == int(np.diff(want).max())
    # values outside [0, nvalues) are counted with the nearest list -- nothing is written outside off[0 .. nvalues]
    a = np.array([-7, -1, 0, 2, 2, 5, 9, 12], np.int32)
    guard = torch.full((5 + 1 + 64,), -99, dtype=torch.int32, device=dev)
    off, _ = ops.run_offsets(torch.from_numpy(a).to(dev), 5, want_max=True)
    assert off.cpu().tolist() == np.searchsorted(np.clip(a, 0, 4), np.arange(6), side="left").tolist()
    assert off.cpu().tolist()[-1] == len(a) and bool((guard == -99).all())
    nlist, D, nt = 300, 64, 1000
    sums = rng.standard_normal((nlist, D)).astype(np.float32) * 5
    train = rng.standard_normal((nt, D)).astype(np.float32)
    off = np.arange(nlist + 1, dtype=np.int32)
    off[101:] -= 1                                                   # list 100 is empty
    fb = rng.integers(0, nt, nlist).astype(np.int32)
    T = lambda x: torch.from_numpy(x).to(dev)  # noqa: E731
    cent = ops.ivf_centroids(T(sums), T(off), T(train), T(fb)).cpu().numpy()
    src = sums.copy()
    src[100] = train[fb[100]]
    want = src / np.linalg.norm(src, axis=1, keepdims=True)
    assert np.abs(cent - want).max() <= 1e-6
    assert np.abs(ops.ivf_centroids(T(sums)).cpu().numpy() - sums / np.linalg.norm(sums, axis=1, keepdims=True)).max() <= 1e-6
