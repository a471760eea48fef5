"""Randomised cross-check of the row-sharded steps at world 1 -- the whole exchange machinery (bucket, self-exchange,
gather, gradient rows, owner-side update), per occurrence and per distinct row -- against the single-device train_step:
random table sizes (both sides of 2^21 virtual rows), widths, batch sizes, hot ids; then the loop helper with
overlapped lookups against the loop with every lookup in line, bit for bit.  SEED, CASES."""
import os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["ESR_SHARDED_WORLD1_DIRECT"] = "0"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from esrecsys_amd import TrainState, ops, optim, sharded
from esrecsys_amd.pinterest.models import STLModel
from esrecsys_amd.pinterest.train_shop_the_look import train_step
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("CASES", "24"))
def rel(a, b):
    a = a.double(); b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
bad = 0
for case in range(N):
    big = rng.random() < 0.3
    Vs, Vp = (1_300_000, 1_200_000) if big else (int(rng.choice([300, 5000, 60000])), int(rng.choice([700, 7000, 90000])))
    D = int(rng.choice([32, 64, 128]))
    B = int(rng.choice([128, 256, 1024, 2048, 4096]))
    hot = rng.random() < 0.4
    unique = rng.random() < 0.5
    lam, lr = 0.1, 0.05
    stl = STLModel(output_size=D, num_scenes=Vs, num_products=Vp, device=dev)
    params = stl.init(int(rng.integers(0, 1000)))
    state = TrainState.create(apply_fn=stl.apply, params=params, tx=optim.sparse_adagrad(lr))
    st = params["params"]["scene_tower"]["embedding"].clone()
    pt = params["params"]["product_tower"]["embedding"].clone()
    scene = sharded.RowShardedTable(st, torch.full_like(st, 0.1), Vs)
    prod = sharded.RowShardedTable(pt, torch.full_like(pt, 0.1), Vp)
    towers = sharded.ShardedTableGroup([scene, prod], kernels=ops, unique=unique)
    ok = True
    def draw(V):
        x = rng.integers(0, V, B)
        if hot:
            x[rng.random(B) < 0.3] = rng.integers(0, 4)
        return torch.from_numpy(x.astype(np.int32)).to(dev)
    steps = int(rng.integers(2, 7))
    for step in range(steps):
        sid, pid, nid = draw(Vs), draw(Vp), draw(Vp)
        if rng.random() < 0.5:
            l_sh = sharded.sharded_triplet_step(towers, sid, pid, nid, lam, float(B), lr)
            state, l_1 = train_step(state, sid, pid, nid, lam, B)
        else:
            l_sh = sharded.sharded_inbatch_step(towers, sid, pid, lam, float(B), 4.0, lr)
            state, l_1 = train_step(state, sid, pid, None, lam, B, scale=4.0)
        ok = ok and abs(float(l_sh) - float(l_1)) <= 2e-6 * abs(float(l_1))
    p = state.params["params"]
    e1, e2 = rel(scene.local, p["scene_tower"]["embedding"]), rel(prod.local, p["product_tower"]["embedding"])
    ok = ok and e1 <= 2e-5 and e2 <= 2e-5
    if os.environ.get("VERBOSE") == "1" or not ok:
        print("ok  " if ok else "MISMATCH", dict(Vs=Vs, Vp=Vp, D=D, B=B, hot=hot, unique=unique, steps=steps, e=(e1, e2)), flush=True)
    bad += 0 if ok else 1
    del towers, scene, prod, state, params, st, pt
    torch.cuda.empty_cache()
print("cases", N, "mismatches", bad)

# ---- the loop helper with overlapped lookups (next batch's rows fetched before this batch's update, stale rows served
# again) against the same helper with every lookup in line: bit for bit, all three workloads, random shapes and plan groups
bad2 = 0
for case in range(N):
    workload = str(rng.choice(["triplet", "inbatch", "glove"]))
    V = int(rng.choice([300, 5000, 60000, 400000]))
    D = int(rng.choice([32, 64, 128]))
    B = int(rng.choice([128, 256, 1024, 4096]))
    hot, unique = rng.random() < 0.5, rng.random() < 0.5
    steps, pg_ = int(rng.integers(1, 12)), int(rng.integers(1, 6))
    seed = int(rng.integers(0, 1 << 30))
    def draw(*shape):
        x = rng.integers(0, V, shape)
        if hot:
            x[rng.random(shape) < 0.3] = rng.integers(0, 4)
        return torch.from_numpy(x.astype(np.int32)).to(dev)
    if workload == "glove":
        batches = [(draw(2, B), torch.from_numpy(rng.uniform(0.1, 300, B).astype(np.float32)).to(dev)) for _ in range(steps)]
    else:
        batches = [tuple(draw(B) for _ in range(3 if workload == "triplet" else 2)) for _ in range(steps)]
    def groups():
        g = torch.Generator(device=dev).manual_seed(seed)
        def tab(d):
            t = torch.randn((V, d), generator=g, device=dev) * d ** -0.5
            return sharded.RowShardedTable(t, torch.full((V, d), 0.1, device=dev), V)
        if workload == "glove":
            return (sharded.ShardedTableGroup([tab(D)], kernels=ops, unique=unique),
                    sharded.ShardedTableGroup([tab(1)], kernels=ops, unique=unique))
        return (sharded.ShardedTableGroup([tab(D), tab(D)], kernels=ops, unique=unique),)
    kw = dict(regularization=0.1, global_batch_size=float(B), scale=4.0, lr=0.05, mode=ops.GLOVE_REFERENCE, plan_group=pg_)
    os.environ["ESR_SHARDED_OVERLAP_CALLS"] = "one" if rng.random() < 0.6 else "ops"
    a, b = groups(), groups()
    la = sharded.sharded_train_steps(workload, a, batches, overlap=True, **kw)
    if os.environ["ESR_SHARDED_OVERLAP_CALLS"] == "ops" and workload != "inbatch":
        os.environ["ESR_SHARDED_FUSED"] = "1"
        lb = []  # (the op-by-op overlapped step is compared with the op-by-op sequential step: same kernels)
        for bt in batches:
            plan = (sharded.plan_glove(b[0], bt[0]) if workload == "glove" else sharded.plan_triplet(b[0], *bt))
            rows = [g.lookup_bucketed(plan) for g in b]
            if workload == "glove":
                lb.append(sharded.sharded_glove_step(b[0], b[1], bt[0], bt[1], ops.GLOVE_REFERENCE, 0.05, plan=plan, rows=tuple(rows)))
            else:
                lb.append(sharded.sharded_triplet_step(b[0], *bt, 0.1, float(B), 0.05, plan=plan, rows=rows[0]))
    else:
        lb = sharded.sharded_train_steps(workload, b, batches, overlap=False, **kw)
    torch.cuda.synchronize()
    ok = all(torch.equal(x.reshape(()), y.reshape(())) for x, y in zip(la, lb))
    ok = ok and all(torch.equal(ta.local, tb.local) and torch.equal(ta.accum, tb.accum)
                    for ga, gb in zip(a, b) for ta, tb in zip(ga.tables, gb.tables))
    if os.environ.get("VERBOSE") == "1" or not ok:
        print("ok  " if ok else "MISMATCH", dict(overlap=workload, V=V, D=D, B=B, hot=hot, unique=unique, steps=steps,
                                                 plan_group=pg_, calls=os.environ["ESR_SHARDED_OVERLAP_CALLS"]), flush=True)
    bad2 += 0 if ok else 1
    del a, b
    torch.cuda.empty_cache()
print("overlap cases", N, "mismatches", bad2)
dist.destroy_process_group()
