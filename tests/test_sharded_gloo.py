"""CPU, world_size 2, gloo: the row-sharded exchange (ids -> owners, rows <- owners, grads -> owners) of
esrecsys_amd/sharded.py, with the HIP kernels replaced by the oracle-backed test double
(tests/_cpu_kernels.py).  Checks sharded == single-device on identical batches."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import free_port  # noqa: E402
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORLD = 2
V_S, V_P, D, B, LAM, LR, STEPS = 101, 203, 8, 24, 0.1, 0.05, 3  # odd V: uneven shards


def _full_tables():
    rng = np.random.default_rng(7)
    st = rng.standard_normal((V_S, D)) * 0.4
    pt = rng.standard_normal((V_P, D)) * 0.4
    return st, pt


ZIPF = False  # set (in every process) by the tests that draw Zipf(1) ids: a few rows take most occurrences


def _draw(rng, V, n):
    if not ZIPF:
        return rng.integers(0, V, n).astype(np.int32)
    w = 1.0 / np.arange(1, V + 1)
    return rng.choice(V, size=n, p=w / w.sum()).astype(np.int32)


def _batch(step, rank):
    rng = np.random.default_rng(1000 * step + rank)
    sid, pid, nid = _draw(rng, V_S, B), _draw(rng, V_P, B), _draw(rng, V_P, B)
    sid[:3] = 5  # duplicates that live on one owner
    return sid, pid, nid


def _worker(rank, port, outdir, workload, grouped=False, unique=None, zipf=False, sort_limit=None):
    global ZIPF
    ZIPF = zipf
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if unique is not None:
        os.environ["ESR_SHARDED_UNIQUE"] = "1" if unique else "0"
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import _cpu_kernels as K
    from esrecsys_amd import sharded
    if sort_limit is not None:  # only rank 0 may batch its owner-side sorts: the ranks sort differently, exchange alike
        sharded._OWNER_SORT_BATCH_MAX = sort_limit if rank == 0 else 0
    st, pt = _full_tables()
    mk = lambda full: torch.from_numpy(np.ascontiguousarray(full[rank::WORLD]))  # noqa: E731
    scene = sharded.RowShardedTable(mk(st), torch.full_like(mk(st), 0.1), V_S)
    prod = sharded.RowShardedTable(mk(pt), torch.full_like(mk(pt), 0.1), V_P)
    towers = sharded.ShardedTableGroup([scene, prod], kernels=K)
    assert towers.unique == (True if unique is None else unique)   # world 2: distinct rows once, unless switched off
    assert scene.local.shape[0] == sharded.RowShardedTable.local_rows_for(V_S, WORLD, rank)
    assert towers.voff[1] % WORLD == 0 and towers.voff[1] >= V_S
    losses, rows_sent, occurrences = [], 0, 0
    plans = None
    if grouped:
        # the routing plans of ALL the batches made together (bench_sharded.py, ESR_SHARDED_PLAN_GROUP): one counts
        # all-to-all and one host wait for the group, the ids exchanges and owner-side sorts ahead of the steps
        lookups = []
        for step in range(STEPS):
            sid, pid, nid = (torch.from_numpy(x) for x in _batch(step, rank))
            segs = ([sid, pid, nid], [0, 1, 1]) if workload == "triplet" else ([sid, pid], [0, 1])
            lookups.append((towers, towers.virtual_id_segments(*segs)))
        plans = sharded.begin_plans(lookups).finish()
        assert len(plans) == STEPS
    for step in range(STEPS):
        sid, pid, nid = (torch.from_numpy(x) for x in _batch(step, rank))
        plan = plans[step] if plans is not None else None
        if plan is None:  # (made here so that the rows that cross the exchange can be counted)
            plan = sharded.plan_triplet(towers, sid, pid, nid) if workload == "triplet" else \
                sharded.plan_inbatch(towers, sid, pid)
        rows_sent += plan.n_rows
        occurrences += plan.n
        if workload == "triplet":
            loss = sharded.sharded_triplet_step(towers, sid, pid, nid, LAM, float(WORLD * B), LR, plan=plan)
        else:
            loss = sharded.sharded_inbatch_step(towers, sid, pid, LAM, float(WORLD * B), 2.0, LR, plan=plan)
        total = loss.clone()
        dist.all_reduce(total)
        losses.append(float(total))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), scene=scene.local.numpy(), prod=prod.local.numpy(),
             scene_acc=scene.accum.numpy(), losses=np.array(losses), rows_sent=rows_sent, occurrences=occurrences)
    dist.barrier()
    dist.destroy_process_group()


def _run(workload, grouped=False, unique=None, zipf=False, sort_limit=None):
    import socket
    global ZIPF
    ZIPF = zipf  # (the checking side draws the same batches)
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(port, d, workload, grouped, unique, zipf, sort_limit), nprocs=WORLD, join=True)
        outs = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(WORLD)]
    return outs


def _reassemble(outs, key, V):
    full = np.zeros((V, D))
    for r in range(WORLD):
        full[r::WORLD] = outs[r][key]
    return full


@pytest.mark.timeout(300)
@pytest.mark.parametrize("grouped,unique,zipf", [(False, None, False), (True, None, False), (False, False, False),
                                                 (True, False, True), (False, True, True), (True, True, True)])
def test_sharded_triplet_equals_single_device(grouped, unique, zipf):
    """unique (the default at world 2): every distinct row crosses the exchange once and ONE summed gradient row goes
    back -- with Zipf ids far fewer rows than occurrences, and still the single-device tables to 1e-12."""
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    outs = _run("triplet", grouped, unique, zipf)
    for o in outs:
        if unique is False:
            assert int(o["rows_sent"]) == int(o["occurrences"]) == 3 * B * STEPS
        else:
            assert int(o["rows_sent"]) < int(o["occurrences"])   # (sid[:3] = 5 alone makes two duplicates per batch)
            if zipf:
                assert int(o["rows_sent"]) < 0.8 * int(o["occurrences"])
    st, pt = _full_tables()
    a_s, a_p = np.full_like(st, 0.1), np.full_like(pt, 0.1)
    for step in range(STEPS):
        parts = [_batch(step, r) for r in range(WORLD)]
        sid, pid, nid = (np.concatenate([p[i] for p in parts]) for i in range(3))
        loss, gs, gp, gn = o_stl.triplet_loss_and_grads(st[sid], pt[pid], pt[nid], LAM, WORLD * B, np.float64)
        assert abs(outs[0]["losses"][step] - loss) <= 1e-12 * max(1.0, abs(loss))
        st, a_s = o_optim.sparse_adagrad_update(st, a_s, sid, gs, LR, dtype=np.float64)
        pt, a_p = o_optim.sparse_adagrad_update(pt, a_p, np.concatenate([pid, nid]), np.concatenate([gp, gn]), LR,
                                                dtype=np.float64)
    assert np.abs(_reassemble(outs, "scene", V_S) - st).max() <= 1e-12
    assert np.abs(_reassemble(outs, "prod", V_P) - pt).max() <= 1e-12
    assert np.abs(_reassemble(outs, "scene_acc", V_S) - a_s).max() <= 1e-12
    assert outs[0]["losses"].tolist() == outs[1]["losses"].tolist()


@pytest.mark.timeout(300)
def test_grouped_plans_when_ranks_sort_differently():
    """One rank batches its owner-side sorts, the other (past the batched sort's limit) sorts list by list: both must
    issue the SAME ids exchanges (the decision how to exchange never looks at rank-local sizes) and the step must still
    equal the single device."""
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    outs = _run("triplet", grouped=True, unique=False, sort_limit=1 << 20)
    st, pt = _full_tables()
    a_s, a_p = np.full_like(st, 0.1), np.full_like(pt, 0.1)
    for step in range(STEPS):
        parts = [_batch(step, r) for r in range(WORLD)]
        sid, pid, nid = (np.concatenate([p[i] for p in parts]) for i in range(3))
        _, gs, gp, gn = o_stl.triplet_loss_and_grads(st[sid], pt[pid], pt[nid], LAM, WORLD * B, np.float64)
        st, a_s = o_optim.sparse_adagrad_update(st, a_s, sid, gs, LR, dtype=np.float64)
        pt, a_p = o_optim.sparse_adagrad_update(pt, a_p, np.concatenate([pid, nid]), np.concatenate([gp, gn]), LR,
                                                dtype=np.float64)
    assert np.abs(_reassemble(outs, "scene", V_S) - st).max() <= 1e-12
    assert np.abs(_reassemble(outs, "prod", V_P) - pt).max() <= 1e-12


@pytest.mark.timeout(300)
@pytest.mark.parametrize("grouped,unique,zipf", [(False, None, False), (True, False, False), (True, True, True)])
def test_sharded_inbatch_matches_per_rank_oracle(grouped, unique, zipf):
    """In-batch negatives are per rank, so the single-device equivalent is: each rank's local-batch
    gradients (normalised by the global batch), scattered into one shared table."""
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    outs = _run("inbatch", grouped, unique, zipf)
    st, pt = _full_tables()
    a_s, a_p = np.full_like(st, 0.1), np.full_like(pt, 0.1)
    for step in range(STEPS):
        ids_s, ids_p, g_s, g_p, total = [], [], [], [], 0.0
        for r in range(WORLD):
            sid, pid, _ = _batch(step, r)
            loss, _, gq, gc = o_stl.inbatch_softmax_loss_and_grads(st[sid], pt[pid], LAM, WORLD * B, 2.0, np.float64)
            total += loss
            ids_s.append(sid), ids_p.append(pid), g_s.append(gq), g_p.append(gc)
        assert abs(outs[0]["losses"][step] - total) <= 1e-12 * max(1.0, abs(total))
        st, a_s = o_optim.sparse_adagrad_update(st, a_s, np.concatenate(ids_s), np.concatenate(g_s), LR,
                                                dtype=np.float64)
        pt, a_p = o_optim.sparse_adagrad_update(pt, a_p, np.concatenate(ids_p), np.concatenate(g_p), LR,
                                                dtype=np.float64)
    assert np.abs(_reassemble(outs, "scene", V_S) - st).max() <= 1e-12
    assert np.abs(_reassemble(outs, "prod", V_P) - pt).max() <= 1e-12


def _glove_worker(rank, port, outdir, unique=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if unique is not None:
        os.environ["ESR_SHARDED_UNIQUE"] = "1" if unique else "0"
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import _cpu_kernels as K
    from esrecsys_amd import sharded
    rng = np.random.default_rng(11)
    V, Dg, Bg = 97, 8, 20
    emb0, bias0 = rng.standard_normal((V, Dg)) * 0.3, rng.standard_normal((V, 1)) * 0.05
    mk = lambda full: torch.from_numpy(np.ascontiguousarray(full[rank::WORLD]))  # noqa: E731
    emb_t = sharded.RowShardedTable(mk(emb0), torch.full_like(mk(emb0), 0.1), V)
    bias_t = sharded.RowShardedTable(mk(bias0), torch.full_like(mk(bias0), 0.1), V)
    emb = sharded.ShardedTableGroup([emb_t], kernels=K)
    bias = sharded.ShardedTableGroup([bias_t], kernels=K)
    brng = np.random.default_rng(100 + rank)
    batches = [(brng.integers(0, V, (2, Bg)).astype(np.int32), brng.uniform(0.1, 300, Bg)) for _ in range(3)]
    # routing plans pipelined two batches deep, as the bench loop does: begin(k+2) after step k, finish(k+1) after it
    cur = sharded.begin_plan_glove(emb, torch.from_numpy(batches[0][0])).finish()
    pend = sharded.begin_plan_glove(emb, torch.from_numpy(batches[1][0]))
    for i, (inp, tgt) in enumerate(batches):
        sharded.sharded_glove_step(emb, bias, torch.from_numpy(inp), torch.from_numpy(tgt), K.GLOVE_DIAGONAL, 0.05,
                                   plan=cur)
        nxt_pend = sharded.begin_plan_glove(emb, torch.from_numpy(batches[i + 2][0])) if i + 2 < len(batches) else None
        cur = pend.finish() if pend is not None else None
        pend = nxt_pend
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), emb=emb_t.local.numpy(), bias=bias_t.local.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("unique", [None, False])
def test_sharded_glove_with_prefetched_plans_equals_single_device(unique):
    """Diagonal-mode GloVe is a sum over pairs with a 1/B factor: G ranks x B pairs with the per-rank 1/B
    equals a single device applying each rank's batch gradients to one table."""
    import socket
    from oracle import glove as o_glove
    from oracle import optim as o_optim
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_glove_worker, args=(port, d, unique), nprocs=WORLD, join=True)
        outs = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(WORLD)]
    rng = np.random.default_rng(11)
    V, Dg, Bg = 97, 8, 20
    emb, bias = rng.standard_normal((V, Dg)) * 0.3, rng.standard_normal((V, 1)) * 0.05
    a_e, a_b = np.full_like(emb, 0.1), np.full_like(bias, 0.1)
    brngs = [np.random.default_rng(100 + r) for r in range(WORLD)]
    per_rank = [[(g.integers(0, V, (2, Bg)).astype(np.int32), g.uniform(0.1, 300, Bg)) for _ in range(3)] for g in brngs]
    for step in range(3):
        ids_all, rows_all, gb_all = [], [], []
        for r in range(WORLD):
            inp, tgt = per_rank[r][step]
            _, gdot, gs = o_glove.loss_and_grads(emb, bias, inp, tgt, "diagonal", np.float64)
            ids, rows, gb = o_glove.row_grads(emb, inp, gdot, gs, np.float64)
            ids_all.append(ids), rows_all.append(rows), gb_all.append(gb)
        ids_c = np.concatenate(ids_all)
        emb, a_e = o_optim.sparse_adagrad_update(emb, a_e, ids_c, np.concatenate(rows_all), 0.05, dtype=np.float64)
        bias, a_b = o_optim.sparse_adagrad_update(bias, a_b, ids_c, np.concatenate(gb_all)[:, None], 0.05,
                                                  dtype=np.float64)
    full_e, full_b = np.zeros((V, Dg)), np.zeros((V, 1))
    for r in range(WORLD):
        full_e[r::WORLD], full_b[r::WORLD] = outs[r]["emb"], outs[r]["bias"]
    assert np.abs(full_e - emb).max() <= 1e-12 and np.abs(full_b - bias).max() <= 1e-12


def _topk_worker(rank, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import _cpu_kernels as K
    from esrecsys_amd import sharded
    rng = np.random.default_rng(21)
    N, Dq, nq, k = 301, 16, 7, 9
    cands = np.round(rng.standard_normal((N, Dq)) * 4) / 4         # coarse grid: exact ties across shards
    queries = np.round(np.random.default_rng(50 + rank).standard_normal((nq, Dq)) * 4) / 4
    local = torch.from_numpy(np.ascontiguousarray(cands[rank::WORLD]))
    s, i = sharded.sharded_find_top_k(torch.from_numpy(queries), local, k, kernels=K)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), s=s.numpy(), i=i.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_top_k_equals_single_device():
    """Candidates row-sharded id mod G, every rank asks its own queries: the merged answer equals brute force
    over the full candidate set, including the tie rule (lower GLOBAL index first) across shards."""
    import socket
    from oracle import topk as o_topk
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_topk_worker, args=(port, d), nprocs=WORLD, join=True)
        outs = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(WORLD)]
    rng = np.random.default_rng(21)
    cands = np.round(rng.standard_normal((301, 16)) * 4) / 4
    for r in range(WORLD):
        queries = np.round(np.random.default_rng(50 + r).standard_normal((7, 16)) * 4) / 4
        es, ei = o_topk.batched_top_k(queries, cands, 9)
        assert np.array_equal(outs[r]["i"], ei) and np.array_equal(outs[r]["s"], es)


def _helper_worker(rank, port, outdir):
    """(a) sharded_train_steps (plans of a group of batches made together, next group's enqueued ahead) against the
    per-step calls; (b) the same steps with the gradient rows crossing the exchange as bf16."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import _cpu_kernels as K
    from esrecsys_amd import sharded
    st, pt = _full_tables()
    mk = lambda full: torch.from_numpy(np.ascontiguousarray(full[rank::WORLD]))  # noqa: E731

    def towers(grad_dtype=None):
        scene = sharded.RowShardedTable(mk(st), torch.full_like(mk(st), 0.1), V_S)
        prod = sharded.RowShardedTable(mk(pt), torch.full_like(mk(pt), 0.1), V_P)
        return sharded.ShardedTableGroup([scene, prod], kernels=K, grad_dtype=grad_dtype)
    n_steps = 7  # groups of 3 + 3 + 1
    batches = [tuple(torch.from_numpy(x) for x in _batch(step, rank)) for step in range(n_steps)]
    out = {}
    for name, grad_dtype, helper in (("steps", None, False), ("helper", None, True), ("bf16", "bf16", True)):
        tw = towers(grad_dtype)
        if helper:
            losses = sharded.sharded_train_steps("triplet", (tw,), batches, regularization=LAM,
                                                 global_batch_size=float(WORLD * B), lr=LR, plan_group=3)
        else:
            losses = [sharded.sharded_triplet_step(tw, *b, LAM, float(WORLD * B), LR) for b in batches]
        out[name + "_scene"] = tw.tables[0].local.numpy().copy()
        out[name + "_prod"] = tw.tables[1].local.numpy().copy()
        out[name + "_loss"] = np.array([float(l) for l in losses])
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_loop_helper_equals_per_step_calls_and_bf16_gradient_exchange_error():
    """sharded_train_steps is the per-step loop, bit for bit.  With the gradient rows crossing the exchange as bf16
    (config 4's budget) every gradient element is rounded to 8 significant bits after the per-distinct-row sum:
    the tables stay within 2^-7 of the f32 exchange's update over these steps -- and do differ (the option is live)."""
    import socket
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_helper_worker, args=(port, d), nprocs=WORLD, join=True)
        outs = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(WORLD)]
    st, pt = _full_tables()
    for r, o in enumerate(outs):
        assert np.array_equal(o["steps_scene"], o["helper_scene"]) and np.array_equal(o["steps_prod"], o["helper_prod"])
        assert np.array_equal(o["steps_loss"], o["helper_loss"])
        for key, full in (("scene", st), ("prod", pt)):
            exact, half, start = o["helper_" + key], o["bf16_" + key], full[r::WORLD]
            moved = np.abs(exact - start).max()
            err = np.abs(half - exact).max()
            assert 0.0 < err <= 2.0 ** -7 * moved, (key, err, moved)


def _overlap_worker(rank, port, outdir):
    """sharded_train_steps with the next batch's lookup issued BEFORE the current batch's update against the
    sequential loop: all three workloads, Zipf ids (most lookups name rows the step before them writes), distinct-row
    and per-occurrence plans, groups of 3 + 3 + 1 plans.  Under gloo there is no side stream -- the early lookup simply
    executes first, so every stale row really is read too early and only the patch makes the results equal."""
    global ZIPF
    ZIPF = True
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import _cpu_kernels as K
    from esrecsys_amd import sharded
    st, pt = _full_tables()
    mk = lambda full: torch.from_numpy(np.ascontiguousarray(full[rank::WORLD]))  # noqa: E731
    n_steps = 7
    trip = [tuple(torch.from_numpy(x) for x in _batch(step, rank)) for step in range(n_steps)]
    grng = np.random.default_rng(300 + rank)
    Vg, Bg = 97, 20
    glove = [(torch.from_numpy(np.stack([_draw(grng, Vg, Bg), _draw(grng, Vg, Bg)])),
              torch.from_numpy(grng.uniform(0.1, 300, Bg))) for _ in range(n_steps)]
    erng = np.random.default_rng(11)
    emb0, bias0 = erng.standard_normal((Vg, 8)) * 0.3, erng.standard_normal((Vg, 1)) * 0.05

    def groups(workload, unique):
        if workload == "glove":
            e = sharded.RowShardedTable(mk(emb0), torch.full_like(mk(emb0), 0.1), Vg)
            b = sharded.RowShardedTable(mk(bias0), torch.full_like(mk(bias0), 0.1), Vg)
            return (sharded.ShardedTableGroup([e], kernels=K, unique=unique),
                    sharded.ShardedTableGroup([b], kernels=K, unique=unique))
        scene = sharded.RowShardedTable(mk(st), torch.full_like(mk(st), 0.1), V_S)
        prod = sharded.RowShardedTable(mk(pt), torch.full_like(mk(pt), 0.1), V_P)
        return (sharded.ShardedTableGroup([scene, prod], kernels=K, unique=unique),)

    def run(workload, unique, overlap):
        gs = groups(workload, unique)
        kw = dict(mode=K.GLOVE_DIAGONAL) if workload == "glove" else dict(regularization=LAM, global_batch_size=float(WORLD * B))
        batches = glove if workload == "glove" else ([b[:2] for b in trip] if workload == "inbatch" else trip)
        losses = sharded.sharded_train_steps(workload, gs, batches, lr=LR, plan_group=3, overlap=overlap, **kw)
        tabs = [t.local.numpy().copy() for g in gs for t in g.tables] + [t.accum.numpy().copy() for g in gs for t in g.tables]
        return tabs, np.array([float(l) for l in losses])

    out = {}
    patched = [0, 0]
    real_patch = sharded.ShardedTableGroup.patch_rows

    def counting_patch(self, plan, back):
        patched[0] += int(plan.stale.ids.numel())
        patched[1] += int(plan.stale.pos.numel())
        return real_patch(self, plan, back)

    for workload in ("triplet", "inbatch", "glove"):
        for unique in (True, False):
            key = "%s_%d" % (workload, unique)
            want_t, want_l = run(workload, unique, False)
            sharded.ShardedTableGroup.patch_rows = counting_patch
            patched[:] = [0, 0]
            got_t, got_l = run(workload, unique, True)
            out[key + "_equal"] = np.array(all(np.array_equal(a, b) for a, b in zip(want_t, got_t)) and
                                           np.array_equal(want_l, got_l))
            out[key + "_patched"] = np.array(patched)
            # the control: the same loop WITHOUT the patch must differ (the early lookups do read rows too early)
            sharded.ShardedTableGroup.patch_rows = lambda self, plan, back: back
            bad_t, _ = run(workload, unique, True)
            out[key + "_control_differs"] = np.array(not all(np.array_equal(a, b) for a, b in zip(want_t, bad_t)))
            sharded.ShardedTableGroup.patch_rows = real_patch
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_overlapped_lookups_equal_the_sequential_loop_bit_for_bit():
    import socket
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_overlap_worker, args=(port, d), nprocs=WORLD, join=True)
        outs = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(WORLD)]
    for workload in ("triplet", "inbatch", "glove"):
        for unique in (1, 0):
            key = "%s_%d" % (workload, unique)
            assert all(bool(o[key + "_equal"]) for o in outs), key
            # rows were re-served (sent == received over the ranks) and leaving them out changes the tables
            sent = sum(int(o[key + "_patched"][0]) for o in outs)
            assert sent > 0 and sent == sum(int(o[key + "_patched"][1]) for o in outs), key
            assert any(bool(o[key + "_control_differs"]) for o in outs), key
