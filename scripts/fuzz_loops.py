"""Randomised cross-checks of the loop helpers against the per-step calls (bit-identical tables, accumulators, losses):
train_steps (triplet and in-batch batches) and train_epoch, with table sizes on both sides of the 2^21-row boundary of
the tile sorts, batch sizes around the sort dispatch boundaries, hot ids, ragged batches.  SEED, CASES."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esrecsys_amd import TrainState, optim
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("CASES", "30"))
bad = 0
def stl_state(Vs, Vp, D, seed):
    from esrecsys_amd.pinterest.models import STLModel
    g = torch.Generator(device=dev).manual_seed(seed)
    params = {"params": {"scene_tower": {"embedding": torch.randn((Vs, D), generator=g, device=dev) * D ** -0.5},
                         "product_tower": {"embedding": torch.randn((Vp, D), generator=g, device=dev) * D ** -0.5}}}
    model = STLModel(output_size=D, num_scenes=Vs, num_products=Vp, device=dev)
    return TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(0.05))
def glove_state(V, D, seed):
    from esrecsys_amd.wikipedia.models import Glove
    model = Glove(num_embeddings=V, features=D, device=dev)
    params = model.init(seed, torch.zeros((2, 4), dtype=torch.int32, device=dev))
    return TrainState.create(apply_fn=model.apply, params=params["params"], tx=optim.sparse_adagrad(0.05))
def draw(V, B, hot):
    x = rng.integers(0, V, B)
    if hot:
        x[rng.random(B) < 0.35] = rng.integers(0, 3)
    return torch.from_numpy(x.astype(np.int32)).to(dev)
for case in range(N):
    kind = ["triplet", "inbatch", "glove"][case % 3]
    big = rng.random() < 0.4
    D = int(rng.choice([32, 64, 128]))
    K = int(rng.integers(1, 26))
    hot = rng.random() < 0.4
    if kind == "glove":
        import esrecsys_amd.wikipedia.train_cooccurence as tc
        V = int(rng.choice([5000, 70000, 2_200_000 if big else 30000]))
        B = int(rng.choice([16, 300, 2048, 2049, 5000, 16384, 16385, 20000]))
        sizes = [B] * K
        if K > 3 and rng.random() < 0.3:
            sizes[K // 2] = max(1, B // 2 + 1)
        batches = [(torch.stack([draw(V, b, hot), draw(V, b, hot)]), torch.from_numpy(rng.uniform(0.1, 300.0, b).astype(np.float32)).to(dev)) for b in sizes]
        a, la = tc.train_epoch(glove_state(V, D, 3), K, iter(batches))
        b_ = glove_state(V, D, 3)
        ls = []
        for inp, tgt in batches:
            b_, l = tc.train_step(b_, inp, tgt)
            ls.append(float(l))
        ok = torch.equal(a.params["_token_embedding"]["embedding"], b_.params["_token_embedding"]["embedding"]) and \
            torch.equal(a.params["_bias"]["embedding"], b_.params["_bias"]["embedding"]) and \
            abs(la - float(np.mean(np.asarray(ls, np.float32)))) <= 1e-6 * abs(la)
        desc = dict(V=V, D=D, B=B, K=K, hot=hot, sizes=sorted(set(sizes)))
    else:
        from esrecsys_amd.pinterest.train_shop_the_look import train_step, train_steps
        Vs, Vp = (1_200_000, 1_100_000) if big else (int(rng.choice([3000, 50000])), int(rng.choice([5000, 80000])))
        B = int(rng.choice([16, 128, 256, 683, 2048, 8192, 11000]))
        if kind == "inbatch":
            B = int(rng.choice([128, 256, 384, 1024, 100, 2048]))
        sizes = [B] * K
        if K > 3 and rng.random() < 0.3:
            sizes[K // 2] = 128 if kind == "inbatch" else max(1, B // 2 + 1)
        batches = [(draw(Vs, b, hot), draw(Vp, b, hot), None if kind == "inbatch" else draw(Vp, b, hot)) for b in sizes]
        kw = dict(scale=6.0) if kind == "inbatch" else {}
        a, losses = train_steps(stl_state(Vs, Vp, D, 4), iter(batches), K, 0.1, float(B), **kw)
        b_ = stl_state(Vs, Vp, D, 4)
        ref = []
        for s_, p_, n_ in batches:
            b_, l = train_step(b_, s_, p_, n_, 0.1, float(B), **kw)
            ref.append(l)
        ok = torch.equal(losses, torch.stack(ref)) and all(
            torch.equal(a.params["params"][t]["embedding"], b_.params["params"][t]["embedding"]) and
            torch.equal(a.opt_state["sum_of_squares"]["params"][t]["embedding"], b_.opt_state["sum_of_squares"]["params"][t]["embedding"])
            for t in ("scene_tower", "product_tower"))
        desc = dict(Vs=Vs, Vp=Vp, D=D, B=B, K=K, hot=hot, sizes=sorted(set(sizes)))
    torch.cuda.synchronize()
    if os.environ.get("VERBOSE") == "1" or not ok:
        print("ok  " if ok else "MISMATCH", kind, desc, flush=True)
    bad += 0 if ok else 1
    del a, b_, batches
    torch.cuda.empty_cache()
print("cases", N, "mismatches", bad)
