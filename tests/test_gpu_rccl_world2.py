"""GPU, world_size 2, one process per GPU, the DIRECT RCCL exchange (esr_alltoall_* on the compute stream,
esrecsys_amd/rccl.py): sharded_{triplet,inbatch,glove}_step with the real HIP kernels against the fp64 oracle applied
to the unsharded tables.  Skips cleanly on a box with fewer than two GPUs (RCCL refuses two ranks on one device).

The same three comparisons also run as TWO PROCESSES ON ONE GPU, on a box where RCCL cannot form a two-rank
communicator -- every HIP kernel of the sharded steps at world 2 (uneven shards, remote rows, owner-side sorts, segment
sums over received gradient rows); only the wire is substituted:
* "loop1": the library's OWN exchange code (esr_comm.hip's grouped send / recv, the one-call sharded steps of
  esr_shard_step.hip, the group-of-plans exchanges) bound to tests/wire's loopback wire through ESR_RCCL_LIB -- the
  code path of an N-GPU run with the bytes carried by sockets instead of xGMI;
* "gloo1": no direct exchange at all: the op-by-op path over torch.distributed's gloo backend (the fallback a rank takes
  when the RCCL bootstrap fails).

Also here: the BASELINE config-4-shaped step (bf16 towers, one rank's 12.5 M-row share per tower, B = 8192) through
the same RCCL path at world 1, which every 1-GPU box can run."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import free_port  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

WORLD = 2
V_S, V_P, D, B, LAM, LR, SCALE, STEPS = 4001, 6003, 128, 256, 0.1, 0.05, 4.0, 3  # odd V: uneven shards
V_G, D_G, B_G = 3001, 64, 512


def _towers_full():
    rng = np.random.default_rng(7)
    return (rng.standard_normal((V_S, D)) * 0.12).astype(np.float32), \
           (rng.standard_normal((V_P, D)) * 0.12).astype(np.float32)


def _glove_full():
    rng = np.random.default_rng(11)
    return (rng.standard_normal((V_G, D_G)) * D_G ** -0.5).astype(np.float32), \
           (rng.standard_normal((V_G, 1)) * 0.05).astype(np.float32)


def _batch(step, rank):
    rng = np.random.default_rng(1000 * step + rank)
    sid = rng.integers(0, V_S, B).astype(np.int32)
    pid = rng.integers(0, V_P, B).astype(np.int32)
    nid = rng.integers(0, V_P, B).astype(np.int32)
    sid[:3] = 5  # duplicates that live on one owner
    return sid, pid, nid


def _glove_batch(step, rank):
    rng = np.random.default_rng(5000 + 10 * step + rank)
    return rng.integers(0, V_G, (2, B_G)).astype(np.int32), rng.uniform(0.1, 300.0, B_G).astype(np.float32)


def _worker(rank, port, outdir, transport="rccl", wire_lib=None):
    direct = transport in ("rccl", "loop1")  # the library's own exchange (esr_comm.hip) is the path under test
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      ESR_RCCL_DIRECT="1" if direct else "0")
    if transport == "loop1":
        os.environ["ESR_RCCL_LIB"] = wire_lib
    import torch.distributed as dist
    index = rank if transport == "rccl" else 0  # "loop1" / "gloo1": both ranks share cuda:0
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    if transport == "rccl":
        dist.init_process_group("nccl", rank=rank, world_size=WORLD, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from esrecsys_amd import ops, sharded

    def shard(full):
        t = torch.from_numpy(np.ascontiguousarray(full[rank::WORLD])).to(dev)
        return sharded.RowShardedTable(t, torch.full(t.shape, 0.1, device=dev), full.shape[0])

    out = {}
    for workload in ("triplet", "inbatch"):
        st, pt = _towers_full()
        scene, prod = shard(st), shard(pt)
        towers = sharded.ShardedTableGroup([scene, prod], kernels=ops)
        x = towers.exchange()
        if direct:
            assert x is not None, "the direct exchange (esr_comm.hip) must be the path under test"
            assert x.ranks_seen() == (WORLD, rank)
            assert towers._fused() is not None, "the one-call sharded steps must be the path under test"
        else:
            assert x is None and towers.world == WORLD
        losses = []
        plans = None
        if workload == "inbatch":
            # the routing plans of all the batches made together, as bench_sharded.py's loop does: one batched bucket
            # launch pair, one counts exchange, the ids exchanges as one RCCL group, one batched owner-side sort
            ids = [[torch.from_numpy(a).to(dev) for a in _batch(step, rank)[:2]] for step in range(STEPS)]
            plans = sharded.begin_plans([(towers, towers.virtual_id_segments(b, [0, 1])) for b in ids]).finish()
        for step in range(STEPS):
            sid, pid, nid = (torch.from_numpy(a).to(dev) for a in _batch(step, rank))
            if workload == "triplet":
                loss = sharded.sharded_triplet_step(towers, sid, pid, nid, LAM, float(WORLD * B), LR)
            else:
                loss = sharded.sharded_inbatch_step(towers, sid, pid, LAM, float(WORLD * B), SCALE, LR, plan=plans[step])
            total = loss.detach().clone() if transport == "rccl" else loss.detach().cpu().clone()
            dist.all_reduce(total)
            losses.append(float(total))
        out[workload + "_scene"] = scene.local.cpu().numpy()
        out[workload + "_prod"] = prod.local.cpu().numpy()
        out[workload + "_scene_acc"] = scene.accum.cpu().numpy()
        out[workload + "_losses"] = np.array(losses)
    # GloVe with routing plans pipelined two batches deep, as the bench loop runs them
    emb0, bias0 = _glove_full()
    emb_t, bias_t = shard(emb0), shard(bias0)
    emb = sharded.ShardedTableGroup([emb_t], kernels=ops)
    bias = sharded.ShardedTableGroup([bias_t], kernels=ops)
    assert (emb.exchange() is not None) == direct
    batches = [_glove_batch(s, rank) for s in range(STEPS)]
    dv = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    cur = sharded.begin_plan_glove(emb, dv(batches[0][0])).finish()
    pend = sharded.begin_plan_glove(emb, dv(batches[1][0]))
    for i, (inp, tgt) in enumerate(batches):
        sharded.sharded_glove_step(emb, bias, dv(inp), dv(tgt), ops.GLOVE_DIAGONAL, LR, plan=cur)
        nxt = sharded.begin_plan_glove(emb, dv(batches[i + 2][0])) if i + 2 < len(batches) else None
        cur = pend.finish() if pend is not None else None
        pend = nxt
    out["glove_emb"], out["glove_bias"] = emb_t.local.cpu().numpy(), bias_t.local.cpu().numpy()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    from esrecsys_amd import rccl
    rccl.reset()
    dist.destroy_process_group()


def _reassemble(outs, key, V, width):
    full = np.zeros((V, width))
    for r in range(WORLD):
        full[r::WORLD] = outs[r][key]
    return full


_OUTPUTS = {}


@pytest.fixture(scope="module", params=["rccl", "loop1", "gloo1"])
def world2_outputs(request):
    transport = request.param
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if transport == "rccl" and torch.cuda.device_count() < WORLD:
        pytest.skip("needs %d GPUs (RCCL refuses two ranks on one device); this box has %d"
                    % (WORLD, torch.cuda.device_count()))
    if transport not in _OUTPUTS:
        import torch.multiprocessing as mp
        wire_lib = None
        if transport == "loop1":
            import importlib.util
            spec = importlib.util.spec_from_file_location("build_wire", os.path.join(ROOT, "tests", "wire", "build_wire.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            wire_lib = mod.build()
        port = free_port()
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker, args=(port, d, transport, wire_lib), nprocs=WORLD, join=True)
            _OUTPUTS[transport] = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(WORLD)]
    return _OUTPUTS[transport]


TOL = 1e-5


def _close(got, exp):
    return np.abs(got - exp).max() <= TOL * max(np.abs(exp).max(), 1e-30)


@pytest.mark.timeout(600)
def test_world2_rccl_triplet_equals_single_device(world2_outputs):
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    outs = world2_outputs
    st, pt = (t.astype(np.float64) for t in _towers_full())
    a_s, a_p = np.full_like(st, 0.1), np.full_like(pt, 0.1)
    for step in range(STEPS):
        parts = [_batch(step, r) for r in range(WORLD)]
        sid, pid, nid = (np.concatenate([p[i] for p in parts]) for i in range(3))
        loss, gs, gp, gn = o_stl.triplet_loss_and_grads(st[sid], pt[pid], pt[nid], LAM, WORLD * B, np.float64)
        assert abs(outs[0]["triplet_losses"][step] - loss) <= TOL * abs(loss)
        st, a_s = o_optim.sparse_adagrad_update(st, a_s, sid, gs, LR, dtype=np.float64)
        pt, a_p = o_optim.sparse_adagrad_update(pt, a_p, np.concatenate([pid, nid]), np.concatenate([gp, gn]), LR,
                                                dtype=np.float64)
    assert _close(_reassemble(outs, "triplet_scene", V_S, D), st)
    assert _close(_reassemble(outs, "triplet_prod", V_P, D), pt)
    assert _close(_reassemble(outs, "triplet_scene_acc", V_S, D), a_s)
    assert outs[0]["triplet_losses"].tolist() == outs[1]["triplet_losses"].tolist()


@pytest.mark.timeout(600)
def test_world2_rccl_inbatch_matches_per_rank_oracle(world2_outputs):
    """In-batch negatives are per rank: the single-device equivalent applies each rank's local-batch gradients
    (normalised by the global batch) to one shared table."""
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    outs = world2_outputs
    st, pt = (t.astype(np.float64) for t in _towers_full())
    a_s, a_p = np.full_like(st, 0.1), np.full_like(pt, 0.1)
    for step in range(STEPS):
        ids_s, ids_p, g_s, g_p, total = [], [], [], [], 0.0
        for r in range(WORLD):
            sid, pid, _ = _batch(step, r)
            loss, _, gq, gc = o_stl.inbatch_softmax_loss_and_grads(st[sid], pt[pid], LAM, WORLD * B, SCALE, np.float64)
            total += loss
            ids_s.append(sid), ids_p.append(pid), g_s.append(gq), g_p.append(gc)
        assert abs(outs[0]["inbatch_losses"][step] - total) <= TOL * abs(total)
        st, a_s = o_optim.sparse_adagrad_update(st, a_s, np.concatenate(ids_s), np.concatenate(g_s), LR,
                                                dtype=np.float64)
        pt, a_p = o_optim.sparse_adagrad_update(pt, a_p, np.concatenate(ids_p), np.concatenate(g_p), LR,
                                                dtype=np.float64)
    assert _close(_reassemble(outs, "inbatch_scene", V_S, D), st)
    assert _close(_reassemble(outs, "inbatch_prod", V_P, D), pt)


@pytest.mark.timeout(600)
def test_world2_rccl_glove_with_pipelined_plans(world2_outputs):
    from oracle import glove as o_glove
    from oracle import optim as o_optim
    outs = world2_outputs
    emb, bias = (t.astype(np.float64) for t in _glove_full())
    a_e, a_b = np.full_like(emb, 0.1), np.full_like(bias, 0.1)
    for step in range(STEPS):
        ids_all, rows_all, gb_all = [], [], []
        for r in range(WORLD):
            inp, tgt = _glove_batch(step, r)
            _, gdot, gs = o_glove.loss_and_grads(emb, bias, inp, tgt.astype(np.float64), "diagonal", np.float64)
            ids, rows, gb = o_glove.row_grads(emb, inp, gdot, gs, np.float64)
            ids_all.append(ids), rows_all.append(rows), gb_all.append(gb)
        ids_c = np.concatenate(ids_all)
        emb, a_e = o_optim.sparse_adagrad_update(emb, a_e, ids_c, np.concatenate(rows_all), LR, dtype=np.float64)
        bias, a_b = o_optim.sparse_adagrad_update(bias, a_b, ids_c, np.concatenate(gb_all)[:, None], LR,
                                                  dtype=np.float64)
    assert _close(_reassemble(outs, "glove_emb", V_G, D_G), emb)
    assert _close(_reassemble(outs, "glove_bias", V_G, 1), bias)


# ---- BASELINE config 4, one rank's share, through the RCCL path at world 1 ------------------------------------------
@pytest.fixture(scope="module")
def pg1(dev):
    import torch.distributed as dist
    created = not dist.is_initialized()
    if created:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29581")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dist
    if created:
        from esrecsys_amd import rccl
        rccl.reset()
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_config4_share_bf16_towers_b8192(dev, pg1):
    """BASELINE configs[3] as ONE of its eight ranks sees it: two bf16 towers of 12.5 M x 128 rows (100 M / 8) with fp32
    Adagrad accumulators (19 GB), B = 8192 in-batch pairs, ids -> bucket -> exchange -> one-plane score kernels ->
    gradient exchange -> fused Adagrad with RNE rounding to bf16.  Checked against the fp64 oracle on the rows the
    batch touches, and every other row must be bit-untouched (checksum)."""
    from esrecsys_amd import ops, sharded
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    V, Dm, Bm, lam, lr, scale = 12_500_000, 128, 8192, 0.1, 0.05, 8.0
    g = torch.Generator(device=dev).manual_seed(1701)
    tabs = []
    for _ in range(2):
        t = torch.empty((V, Dm), device=dev, dtype=torch.bfloat16)
        for lo in range(0, V, 2_500_000):  # fill in slices: no 6.4 GB fp32 temporary
            n = min(2_500_000, V - lo)
            t[lo:lo + n] = (torch.randn((n, Dm), generator=g, device=dev) * Dm ** -0.5).to(torch.bfloat16)
        tabs.append(sharded.RowShardedTable(t, torch.full((V, Dm), 0.1, device=dev), V))
    towers = sharded.ShardedTableGroup(tabs, kernels=ops)
    assert towers.exchange() is not None, "config 4 runs over the direct RCCL exchange"
    ids = torch.randint(0, V, (2, Bm), generator=g, device=dev, dtype=torch.int32)
    ids[0, :4] = ids[0, 4]  # a few duplicates
    sid, pid = ids[0].contiguous(), ids[1].contiguous()
    before = [t.local[i.long()].float().cpu().numpy().astype(np.float64) for t, i in zip(tabs, (sid, pid))]
    sums0 = [t.local.view(torch.int16).sum(dtype=torch.int64) for t in tabs]
    loss = sharded.sharded_inbatch_step(towers, sid, pid, lam, float(Bm), scale, lr)
    el, _, gq, gc = o_stl.inbatch_softmax_loss_and_grads(before[0], before[1], lam, Bm, scale, np.float64)
    assert abs(float(loss) - el) / abs(el) <= 1e-5
    for t, i, rows0, grads in zip(tabs, (sid, pid), before, (gq, gc)):
        ih = i.cpu().numpy()
        uniq, first = np.unique(ih, return_index=True)
        # oracle update on a compact copy of the touched rows
        remap = np.searchsorted(uniq, ih)
        new_rows, new_acc = o_optim.sparse_adagrad_update(rows0[first], np.full((len(uniq), Dm), 0.1), remap, grads, lr,
                                                          dtype=np.float64)
        got = t.local[torch.from_numpy(uniq).to(dev).long()]
        exp_bf16 = torch.from_numpy(new_rows).to(torch.bfloat16).float().numpy()
        assert np.mean(got.float().cpu().numpy() == exp_bf16) > 0.999  # RNE of an fp32 vs fp64 value may differ by 1 ulp
        acc = t.accum[torch.from_numpy(uniq).to(dev).long()].cpu().numpy()
        assert np.abs(acc - new_acc).max() <= 1e-5 * np.abs(new_acc).max()
        # untouched rows: sum of all bf16 bit patterns changes exactly by the touched rows' change
        delta = (got.view(torch.int16).sum(dtype=torch.int64) -
                 torch.from_numpy(rows0[first]).to(torch.bfloat16).to(dev).view(torch.int16).sum(dtype=torch.int64))
        s0 = sums0[0] if t is tabs[0] else sums0[1]
        assert int(t.local.view(torch.int16).sum(dtype=torch.int64) - s0) == int(delta)
