"""Randomised cross-check of the one-pass steps against the two-call paths (same sort, same association of every sum):
GloVe (esr_glove_train_step vs apply_model + update_model) and triplets (esr_triplet_train_step vs triplet_fwd_bwd +
sort + fused scatter), random V / D / B / mode, uniform, Zipf and all-equal ids, tables beyond 2^21 rows.  SEED, CASES."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esrecsys_amd import TrainState, optim
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("CASES", "40"))
bad = 0
def rel(a, b):
    a = a.double(); b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
def ids(kind, V, shape):
    if kind == "uniform":
        return rng.integers(0, V, shape).astype(np.int32)
    if kind == "same":
        return np.full(shape, 7 % V, np.int32)
    x = rng.integers(0, V, shape)
    hot = rng.random(shape) < 0.5
    x[hot] = rng.integers(0, min(V, 4), int(hot.sum()))
    return x.astype(np.int32)
def glove_state(V, D, mode, seed):
    from esrecsys_amd.wikipedia.models import Glove
    model = Glove(num_embeddings=V, features=D, loss_mode=mode, device=dev)
    params = model.init(seed, None)["params"]
    g = torch.Generator().manual_seed(seed + 1)
    params["_bias"]["embedding"].copy_((torch.randn((V, 1), generator=g) * 0.05).to(dev))
    return TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(0.05))
def stl_state(Vs, Vp, D, seed):
    from esrecsys_amd.pinterest.models import STLModel
    g = torch.Generator(device=dev).manual_seed(seed)
    params = {"params": {"scene_tower": {"embedding": torch.randn((Vs, D), generator=g, device=dev) * D ** -0.5},
                         "product_tower": {"embedding": torch.randn((Vp, D), generator=g, device=dev) * D ** -0.5}}}
    model = STLModel(output_size=D, num_scenes=Vs, num_products=Vp, device=dev)
    return TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(0.05))
for case in range(N):
    kind = str(rng.choice(["uniform", "hot", "same"]))
    steps = int(rng.integers(1, 5))
    if case % 2 == 0:
        from esrecsys_amd.wikipedia.train_cooccurence import apply_model, train_step, update_model
        V = int(rng.choice([7, 300, 5000, 60000, 2_300_000]))
        D = int(rng.choice([4, 6, 32, 64, 100, 128, 256, 512]))
        if V > 1_000_000:
            D = min(D, 64)
        B = int(rng.choice([1, 31, 32, 33, 64, 777, 1000, 2048, 2049, 4096, 16384, 16385, 40000]))
        mode = str(rng.choice(["reference", "diagonal"]))
        a, b = glove_state(V, D, mode, 5), glove_state(V, D, mode, 5)
        ok = True
        for _ in range(steps):
            inp = ids(kind, V, (2, B)); tgt = np.exp(rng.uniform(np.log(0.1), np.log(1000.0), B)).astype(np.float32)
            a, la = train_step(a, inp, tgt)
            grads, lb = apply_model(b, inp, tgt)
            b = update_model(b, grads)
            ok = ok and abs(float(la) - float(lb)) <= 4e-6 * max(abs(float(lb)), 1e-30)
        pa, pb = a.params, b.params
        e = (rel(pa["_token_embedding"]["embedding"], pb["_token_embedding"]["embedding"]), rel(pa["_bias"]["embedding"], pb["_bias"]["embedding"]))
        ok = ok and e[0] <= 2e-6 and e[1] <= 2e-5
        desc = dict(op="glove", V=V, D=D, B=B, mode=mode, kind=kind, steps=steps, e=e)
    else:
        import esrecsys_amd.pinterest.train_shop_the_look as stl
        big = rng.random() < 0.25
        Vs, Vp = (1_200_000, 1_100_000) if big else (int(rng.choice([5, 300, 5000, 60000])), int(rng.choice([9, 700, 7000, 90000])))
        D = int(rng.choice([4, 32, 64, 100, 128, 256])) if not big else 32
        B = int(rng.choice([1, 16, 31, 33, 128, 683, 2048, 8192, 11000, 30000]))
        a, b = stl_state(Vs, Vp, D, 3), stl_state(Vs, Vp, D, 3)
        ok = True
        for _ in range(steps):
            s_, p_, n_ = (torch.from_numpy(ids(kind, V_, B)).to(dev) for V_ in (Vs, Vp, Vp))
            a, la = stl.train_step(a, s_, p_, n_, 0.1, float(B))
            os.environ["ESR_STL_FUSED"] = "0"
            try:
                b, lb = stl.train_step(b, s_, p_, n_, 0.1, float(B))
            finally:
                os.environ["ESR_STL_FUSED"] = "1"
            ok = ok and abs(float(la) - float(lb)) <= 4e-6 * max(abs(float(lb)), 1e-30)
        e = tuple(rel(a.params["params"][t]["embedding"], b.params["params"][t]["embedding"]) for t in ("scene_tower", "product_tower"))
        ok = ok and max(e) <= 5e-6
        desc = dict(op="triplet", Vs=Vs, Vp=Vp, D=D, B=B, kind=kind, steps=steps, e=e)
    torch.cuda.synchronize()
    if os.environ.get("VERBOSE") == "1" or not ok:
        print("ok  " if ok else "MISMATCH", desc, flush=True)
    bad += 0 if ok else 1
    del a, b
    torch.cuda.empty_cache()
print("cases", N, "mismatches", bad)
