"""CPU, gloo: randomised check of the row-sharded steps at world sizes 2..4 (kernels replaced by the NumPy doubles of
tests/_cpu_kernels.py) against the single-device oracle: random table sizes (uneven shards), widths, batch sizes, Zipf
ids, per-occurrence and per-distinct-row exchange, routing plans made per step or for all steps together.  SEED, CASES."""
import os, sys, socket, tempfile
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
LAM, LR = 0.1, 0.05


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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ESR_SHARDED_UNIQUE="1" if cfg["unique"] else "0")
    W = cfg["world"]
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import _cpu_kernels as K
    from esrecsys_amd import sharded
    st, pt = tables(cfg)
    mk = lambda full: torch.from_numpy(np.ascontiguousarray(full[rank::W]))  # noqa: E731
    scene = sharded.RowShardedTable(mk(st), torch.full_like(mk(st), 0.1), cfg["Vs"])
    prod = sharded.RowShardedTable(mk(pt), torch.full_like(mk(pt), 0.1), cfg["Vp"])
    towers = sharded.ShardedTableGroup([scene, prod], kernels=K)
    plans = None
    if cfg["grouped"]:
        lookups = []
        for step in range(cfg["steps"]):
            sid, pid, nid = (torch.from_numpy(x) for x in batch(cfg, step, rank))
            segs = ([sid, pid, nid], [0, 1, 1]) if cfg["workload"] == "triplet" else ([sid, pid], [0, 1])
            lookups.append((towers, towers.virtual_id_segments(*segs)))
        plans = sharded.begin_plans(lookups).finish()
    losses = []
    for step in range(cfg["steps"]):
        sid, pid, nid = (torch.from_numpy(x) for x in batch(cfg, step, rank))
        plan = plans[step] if plans is not None else None
        if cfg["workload"] == "triplet":
            loss = sharded.sharded_triplet_step(towers, sid, pid, nid, LAM, float(W * cfg["B"]), LR, plan=plan)
        else:
            loss = sharded.sharded_inbatch_step(towers, sid, pid, LAM, float(W * cfg["B"]), 2.0, LR, plan=plan)
        total = loss.clone()
        dist.all_reduce(total)
        losses.append(float(total))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), scene=scene.local.numpy(), prod=prod.local.numpy(), losses=np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


def run_case(cfg):
    from oracle import optim as o_optim
    from oracle import stl_head as o_stl
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
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
        ok = ok and abs(outs[0]["losses"][step] - loss) <= 1e-11 * max(1.0, abs(loss))
        st, a_s = o_optim.sparse_adagrad_update(st, a_s, ids_s, g_s, LR, dtype=np.float64)
        pt, a_p = o_optim.sparse_adagrad_update(pt, a_p, ids_p, g_p, LR, dtype=np.float64)
    def whole(key, V):
        full = np.zeros((V, cfg["D"]))
        for r in range(W):
            full[r::W] = outs[r][key]
        return full
    ok = ok and np.abs(whole("scene", cfg["Vs"]) - st).max() <= 1e-11 and np.abs(whole("prod", cfg["Vp"]) - pt).max() <= 1e-11
    return ok


if __name__ == "__main__":
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    N = int(os.environ.get("CASES", "12"))
    bad = 0
    for case in range(N):
        cfg = dict(world=int(rng.integers(2, 5)), Vs=int(rng.choice([5, 37, 101, 1000])), Vp=int(rng.choice([9, 64, 203, 3001])),
                   D=int(rng.choice([4, 8, 16])), B=int(rng.choice([1, 7, 24, 130])), steps=int(rng.integers(1, 4)),
                   zipf=bool(rng.random() < 0.5), unique=bool(rng.random() < 0.6), grouped=bool(rng.random() < 0.5),
                   workload=str(rng.choice(["triplet", "inbatch"])), seed=int(rng.integers(1, 10000)))
        ok = run_case(cfg)
        if os.environ.get("VERBOSE") == "1" or not ok:
            print("ok  " if ok else "MISMATCH", cfg, flush=True)
        bad += 0 if ok else 1
    print("cases", N, "mismatches", bad)
