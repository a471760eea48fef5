"""CPU, gloo: randomised check of the row-sharded steps at world sizes 2..4 (kernels replaced by the NumPy doubles of
tests/_cpu_kernels.py) against the single-device oracle: random table sizes (uneven shards), widths, batch sizes, Zipf
ids, per-occurrence and per-distinct-row exchange, routing plans made per step or for all steps together, the loop helper
with overlapped lookups (next batch's rows fetched before the current update, stale rows re-served).  SEED, CASES.

WIRE=1 (GPU box): the same cases with the REAL HIP kernels (esrecsys_amd.ops, fp32 tables on cuda:0, every rank on the
one GPU) and the library's own exchange code over tests/wire's loopback wire; tolerance 1e-5 instead of 1e-11; larger
shapes (widths up to 128, batches up to 3000, tables up to 100 000 rows: the sort / plan dispatch boundaries)."""
import os, sys, socket, tempfile
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
LAM, LR = 0.1, 0.05
WIRE = os.environ.get("WIRE") == "1"
TOL = 1e-5 if WIRE else 1e-11


def _free_port():
    """A TCP port for a rendezvous that starts a few seconds from now.  Drawn OUTSIDE the kernel's ephemeral range: a
    port handed out by bind(("", 0)) comes from that range and can be given to somebody's outgoing connection before rank
    0 listens on it (EADDRINUSE once in ~500 spawns of the fuzzers -- seen in scripts/fuzz_sharded_gloo.py)."""
    import random
    import socket
    lo, hi = 20000, 32000
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            hi = max(lo + 1000, min(hi, int(f.read().split()[0]) - 1))
    except (OSError, ValueError, IndexError):
        pass
    rnd = random.SystemRandom()  # (never the seeded global generator of a test)
    for _ in range(64):
        p = rnd.randrange(lo, hi)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", p))
                return p
            except OSError:
                continue
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def setup(rank, W, port, cfg):
    """(kernels module, numpy -> tensor on the device under test, tensor -> numpy)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ESR_SHARDED_UNIQUE="1" if cfg["unique"] else "0")
    if WIRE:
        os.environ.update(ESR_RCCL_LIB=os.path.join(ROOT, "tests", "wire", "libesr_loopback_wire.so"), ESR_RCCL_DIRECT="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    if WIRE:
        from esrecsys_amd import ops as K
        dev = torch.device("cuda", 0)

        def T(x):
            t = torch.from_numpy(np.ascontiguousarray(x))
            return (t.float() if t.is_floating_point() else t).to(dev)
    else:
        import _cpu_kernels as K
        T = lambda x: torch.from_numpy(np.ascontiguousarray(x))  # noqa: E731
    return K, T, (lambda t: t.detach().cpu().numpy())


def summed(loss):
    total = loss.detach().cpu().clone()
    dist.all_reduce(total)
    return float(total)


def tables(cfg):
    rng = np.random.default_rng(cfg["seed"])
    return rng.standard_normal((cfg["Vs"], cfg["D"])) * 0.4, rng.standard_normal((cfg["Vp"], cfg["D"])) * 0.4


def batch(cfg, step, rank):
    rng = np.random.default_rng(cfg["seed"] * 7919 + 1000 * step + rank)
    def draw(V, n):
        if not cfg["zipf"]:
            return rng.integers(0, V, n).astype(np.int32)
        w = 1.0 / np.arange(1, V + 1)
        return rng.choice(V, size=n, p=w / w.sum()).astype(np.int32)
    B = cfg["B"]
    return draw(cfg["Vs"], B), draw(cfg["Vp"], B), draw(cfg["Vp"], B)


def worker(rank, port, outdir, cfg):
    W = cfg["world"]
    K, T, N = setup(rank, W, port, cfg)
    from esrecsys_amd import sharded
    st, pt = tables(cfg)
    mk = lambda full: T(full[rank::W])  # noqa: E731
    scene = sharded.RowShardedTable(mk(st), torch.full_like(mk(st), 0.1), cfg["Vs"])
    prod = sharded.RowShardedTable(mk(pt), torch.full_like(mk(pt), 0.1), cfg["Vp"])
    towers = sharded.ShardedTableGroup([scene, prod], kernels=K)
    plans = None
    if cfg["grouped"]:
        lookups = []
        for step in range(cfg["steps"]):
            sid, pid, nid = (T(x) for x in batch(cfg, step, rank))
            segs = ([sid, pid, nid], [0, 1, 1]) if cfg["workload"] == "triplet" else ([sid, pid], [0, 1])
            lookups.append((towers, towers.virtual_id_segments(*segs)))
        plans = sharded.begin_plans(lookups).finish()
    losses = []
    if cfg.get("overlap"):  # the loop helper with the next lookup issued before the current update (+ stale-row patch)
        bs = [tuple(T(x) for x in batch(cfg, step, rank)) for step in range(cfg["steps"])]
        kw = dict(regularization=LAM, global_batch_size=float(W * cfg["B"]), lr=LR, plan_group=cfg["plan_group"], overlap=True)
        if cfg["workload"] == "triplet":
            ls = sharded.sharded_train_steps("triplet", (towers,), bs, **kw)
        else:
            ls = sharded.sharded_train_steps("inbatch", (towers,), [b[:2] for b in bs], scale=2.0, **kw)
        for loss in ls:
            losses.append(summed(loss))
    for step in range(0 if cfg.get("overlap") else cfg["steps"]):
        sid, pid, nid = (T(x) for x in batch(cfg, step, rank))
        plan = plans[step] if plans is not None else None
        if cfg["workload"] == "triplet":
            loss = sharded.sharded_triplet_step(towers, sid, pid, nid, LAM, float(W * cfg["B"]), LR, plan=plan)
        else:
            loss = sharded.sharded_inbatch_step(towers, sid, pid, LAM, float(W * cfg["B"]), 2.0, LR, plan=plan)
        losses.append(summed(loss))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), scene=N(scene.local), prod=N(prod.local), losses=np.array(losses))
    leave()


def leave():
    dist.barrier()
    if WIRE:
        from esrecsys_amd import rccl
        rccl.reset()
    dist.destroy_process_group()


def glove_worker(rank, port, outdir, cfg):
    W, V, D, B = cfg["world"], cfg["Vs"], cfg["D"], cfg["B"]
    K, T, N = setup(rank, W, port, cfg)
    from esrecsys_amd import sharded
    rng = np.random.default_rng(cfg["seed"])
    emb0, bias0 = rng.standard_normal((V, D)) * 0.3, rng.standard_normal((V, 1)) * 0.05
    mk = lambda full: T(full[rank::W])  # noqa: E731
    emb_t = sharded.RowShardedTable(mk(emb0), torch.full_like(mk(emb0), 0.1), V)
    bias_t = sharded.RowShardedTable(mk(bias0), torch.full_like(mk(bias0), 0.1), V)
    emb = sharded.ShardedTableGroup([emb_t], kernels=K)
    bias = sharded.ShardedTableGroup([bias_t], kernels=K)
    batches = glove_batches(cfg, rank)
    if cfg.get("overlap"):
        sharded.sharded_train_steps("glove", (emb, bias), [(T(a), T(b)) for a, b in batches],
                                    mode=K.GLOVE_DIAGONAL, lr=LR, plan_group=cfg["plan_group"], overlap=True)
        batches = []
    cur = sharded.begin_plan_glove(emb, T(batches[0][0])).finish() if batches else None
    pend = sharded.begin_plan_glove(emb, T(batches[1][0])) if len(batches) > 1 else None
    for i, (inp, tgt) in enumerate(batches):
        sharded.sharded_glove_step(emb, bias, T(inp), T(tgt), K.GLOVE_DIAGONAL, LR,
                                   plan=cur if cfg["grouped"] else None)
        nxt = sharded.begin_plan_glove(emb, T(batches[i + 2][0])) if i + 2 < len(batches) else None
        cur = pend.finish() if pend is not None else None
        pend = nxt
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), emb=N(emb_t.local), bias=N(bias_t.local))
    leave()


def glove_batches(cfg, rank):
    g = np.random.default_rng(cfg["seed"] * 31 + rank)
    V, B = cfg["Vs"], cfg["B"]
    def draw():
        if not cfg["zipf"]:
            return g.integers(0, V, (2, B)).astype(np.int32)
        w = 1.0 / np.arange(1, V + 1)
        return g.choice(V, size=(2, B), p=w / w.sum()).astype(np.int32)
    return [(draw(), g.uniform(0.1, 300, B)) for _ in range(cfg["steps"])]


def run_glove(cfg):
    from oracle import glove as o_glove
    from oracle import optim as o_optim
    port = _free_port()
    W, V, D = cfg["world"], cfg["Vs"], cfg["D"]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(glove_worker, args=(port, d, cfg), nprocs=W, join=True)
        outs = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(W)]
    rng = np.random.default_rng(cfg["seed"])
    emb, bias = rng.standard_normal((V, D)) * 0.3, rng.standard_normal((V, 1)) * 0.05
    a_e, a_b = np.full_like(emb, 0.1), np.full_like(bias, 0.1)
    per_rank = [glove_batches(cfg, r) for r in range(W)]
    for step in range(cfg["steps"]):
        ids_all, rows_all, gb_all = [], [], []
        for r in range(W):
            inp, tgt = per_rank[r][step]
            _, gdot, gs = o_glove.loss_and_grads(emb, bias, inp, tgt, "diagonal", np.float64)
            ids, rows, gb = o_glove.row_grads(emb, inp, gdot, gs, np.float64)
            ids_all.append(ids), rows_all.append(rows), gb_all.append(gb)
        ids_c = np.concatenate(ids_all)
        emb, a_e = o_optim.sparse_adagrad_update(emb, a_e, ids_c, np.concatenate(rows_all), LR, dtype=np.float64)
        bias, a_b = o_optim.sparse_adagrad_update(bias, a_b, ids_c, np.concatenate(gb_all)[:, None], LR, dtype=np.float64)
    full_e, full_b = np.zeros((V, D)), np.zeros((V, 1))
    for r in range(W):
        full_e[r::W], full_b[r::W] = outs[r]["emb"], outs[r]["bias"]
    return np.abs(full_e - emb).max() <= TOL * max(1.0, np.abs(emb).max()) and \
        np.abs(full_b - bias).max() <= TOL * max(1.0, np.abs(bias).max())


def run_case(cfg):
    if cfg["workload"] == "glove":
        return run_glove(cfg)
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    port = _free_port()
    W = cfg["world"]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(port, d, cfg), nprocs=W, join=True)
        outs = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(W)]
    st, pt = tables(cfg)
    a_s, a_p = np.full_like(st, 0.1), np.full_like(pt, 0.1)
    ok = True
    for step in range(cfg["steps"]):
        parts = [batch(cfg, step, r) for r in range(W)]
        if cfg["workload"] == "triplet":
            sid, pid, nid = (np.concatenate([p[i] for p in parts]) for i in range(3))
            loss, gs, gp, gn = o_stl.triplet_loss_and_grads(st[sid], pt[pid], pt[nid], LAM, W * cfg["B"], np.float64)
            ids_p, g_p = np.concatenate([pid, nid]), np.concatenate([gp, gn])
            ids_s, g_s = sid, gs
        else:
            loss, ids_s, ids_p, g_s, g_p = 0.0, [], [], [], []
            for sid, pid, _ in parts:
                l, _, gq, gc = o_stl.inbatch_softmax_loss_and_grads(st[sid], pt[pid], LAM, W * cfg["B"], 2.0, np.float64)
                loss += l
                ids_s.append(sid), ids_p.append(pid), g_s.append(gq), g_p.append(gc)
            ids_s, ids_p, g_s, g_p = (np.concatenate(x) for x in (ids_s, ids_p, g_s, g_p))
        ok = ok and abs(outs[0]["losses"][step] - loss) <= TOL * max(1.0, abs(loss))
        st, a_s = o_optim.sparse_adagrad_update(st, a_s, ids_s, g_s, LR, dtype=np.float64)
        pt, a_p = o_optim.sparse_adagrad_update(pt, a_p, ids_p, g_p, LR, dtype=np.float64)
    def whole(key, V):
        full = np.zeros((V, cfg["D"]))
        for r in range(W):
            full[r::W] = outs[r][key]
        return full
    ok = ok and np.abs(whole("scene", cfg["Vs"]) - st).max() <= TOL * max(1.0, np.abs(st).max()) and \
        np.abs(whole("prod", cfg["Vp"]) - pt).max() <= TOL * max(1.0, np.abs(pt).max())
    return ok


if __name__ == "__main__":
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    N = int(os.environ.get("CASES", "12"))
    bad = 0
    for case in range(N):
        big = WIRE and rng.random() < 0.5
        cfg = dict(world=int(rng.integers(2, 5)), Vs=int(rng.choice([5, 37, 101, 1000] + ([20_011, 100_003] if big else []))),
                   Vp=int(rng.choice([9, 64, 203, 3001] + ([50_021] if big else []))),
                   D=int(rng.choice([4, 8, 16] + ([64, 128] if WIRE else []))),
                   B=int(rng.choice([1, 7, 24, 130] + ([512, 1100, 3000] if big else []))), steps=int(rng.integers(1, 6)),
                   overlap=bool(rng.random() < 0.5), plan_group=int(rng.integers(1, 4)),
                   zipf=bool(rng.random() < 0.5), unique=bool(rng.random() < 0.6), grouped=bool(rng.random() < 0.5),
                   workload=str(rng.choice(["triplet", "inbatch", "glove"])), seed=int(rng.integers(1, 10000)))
        ok = run_case(cfg)
        if os.environ.get("VERBOSE") == "1" or not ok:
            print("ok  " if ok else "MISMATCH", cfg, flush=True)
        bad += 0 if ok else 1
    print("cases", N, "mismatches", bad)
