"""Randomised check of the one-pass steps on bf16 tables (round 6: esr_triplet_train_step direct mode, esr_glove_train_step;
bf16 rows, fp32 accumulators) against the fp64 oracle that rounds every touched row to bf16 after each step
(tests/test_gpu_config_size_oracle.py's criterion at random small shapes): random V / D / B, uniform / hot / all-equal ids
(runs far beyond a chunk: the long-run launches), 1-4 steps.  SEED, CASES."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from esrecsys_amd import TrainState, ops, optim
from oracle import glove as o_glove, optim as o_optim, stl_head as o_stl
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("CASES", "40"))
F64 = np.float64
bad = 0
def ids(kind, V, shape):
    if kind == "uniform":
        return rng.integers(0, V, shape).astype(np.int32)
    if kind == "same":
        return np.full(shape, 7 % V, np.int32)
    x = rng.integers(0, V, shape)
    hot = rng.random(shape) < 0.5
    x[hot] = rng.integers(0, min(V, 4), int(hot.sum()))
    return x.astype(np.int32)
def rows_ok(table, want, rows, tag, nocc):
    got = table[torch.as_tensor(rows, device=dev)].float().cpu().numpy().astype(F64)
    w = want[rows]
    # one bf16 step of the element -- but never finer than that of an element 1/16 of the row's largest: a near-zero
    # element differs by many of ITS steps as soon as a partner row differs by one of its own
    step = np.maximum(np.abs(w), np.abs(w).max(axis=1, keepdims=True) / 16.0) * 2.0 ** -7 + 1e-30
    off = np.abs(got - w) / step
    same = float(np.mean(got == w))
    # all-equal ids: ONE row takes a gradient of thousands of terms, its Adagrad step saturates at exactly +-lr, and
    # old -+ lr lands on bf16 ties (the low bits of a bf16 number shifted under the result's last place) which the f32
    # step (tie -> even) and the fp64 oracle (lr (1 - 1e-10): no tie) round apart -- a tenth of the elements may differ
    # (rows that take hundreds of occurrences a step -- a tiny table, hot ids -- are a milder case of the same: 0.97)
    heavy = tag == "hot" or len(rows) * 8 <= nocc
    return (same >= (0.85 if tag == "same" else (0.97 if heavy else 0.995)) and float(off.max()) <= 64.0), same, float(off.max())
for case in range(N):
    kind = str(rng.choice(["uniform", "hot", "same"]))
    steps = int(rng.integers(1, 5)) if kind != "same" else int(rng.integers(1, 3))  # (same: a self-amplifying trajectory)
    ok = True
    if case % 2 == 0:
        from esrecsys_amd.pinterest.models import STLModel
        from esrecsys_amd.pinterest.train_shop_the_look import train_steps
        Vs, Vp = int(rng.choice([5, 300, 5000, 60000])), int(rng.choice([9, 700, 7000, 90000]))
        D = int(rng.choice([4, 8, 20, 32, 64, 100, 128, 256]))
        B = int(rng.choice([1, 16, 31, 33, 128, 683, 2048, 8192, 11000]))
        if 8 * min(Vs, Vp) <= B:  # rows that take many occurrences a step: a self-amplifying trajectory -- two steps at most
            steps = min(steps, 2)
        lam, lr, norm = 0.1, 0.2, float(max(1, B // 128))
        g = torch.Generator(device=dev).manual_seed(case)
        st = (torch.randn((Vs, D), generator=g, device=dev) * (2.0 / np.sqrt(D))).to(torch.bfloat16)
        pt = (torch.randn((Vp, D), generator=g, device=dev) * (2.0 / np.sqrt(D))).to(torch.bfloat16)
        es, ep = st.float().double().cpu().numpy(), pt.float().double().cpu().numpy()
        es0, ep0 = es.copy(), ep.copy()
        model = STLModel(output_size=D, num_scenes=Vs, num_products=Vp, device=dev)
        state = TrainState.create(apply_fn=model.apply, tx=optim.sparse_adagrad(lr),
                                  params={"params": {"scene_tower": {"embedding": st}, "product_tower": {"embedding": pt}}})
        batches = [(ids(kind, Vs, B), ids(kind, Vp, B), ids("uniform", Vp, B)) for _ in range(steps)]
        a_s, a_p = np.full_like(es, 0.1), np.full_like(ep, 0.1)
        state, losses = train_steps(state, iter([tuple(torch.as_tensor(x, device=dev) for x in b) for b in batches]), steps, lam, norm)
        losses = losses.cpu().numpy()
        for k, (sid, pid, nid) in enumerate(batches):
            el, gs, gp, gn = o_stl.triplet_loss_and_grads(es[sid], ep[pid], ep[nid], lam, norm, F64)
            ok = ok and abs(float(losses[k]) - el) <= 1e-4 * max(abs(el), 1e-6)
            o_optim.sparse_adagrad_update_inplace(es, a_s, sid, gs, lr)
            pn = np.concatenate([pid, nid])
            o_optim.sparse_adagrad_update_inplace(ep, a_p, pn, np.concatenate([gp, gn]), lr)
            us, up = np.unique(sid), np.unique(pn)
            es[us] = o_optim.round_bf16(es[us]); ep[up] = o_optim.round_bf16(ep[up])
        p_ = state.params["params"]
        ts = np.unique(np.concatenate([b[0] for b in batches])); tp = np.unique(np.concatenate([np.concatenate(b[1:]) for b in batches]))
        r1 = rows_ok(p_["scene_tower"]["embedding"], es, ts, kind, B); r2 = rows_ok(p_["product_tower"]["embedding"], ep, tp, kind, 2 * B)
        mask = np.ones(Vs, bool); mask[ts] = False
        untouched = bool(np.array_equal(p_["scene_tower"]["embedding"].float().cpu().numpy()[mask], es0[mask].astype(np.float32)))
        ok = ok and r1[0] and r2[0] and untouched and p_["scene_tower"]["embedding"].dtype == torch.bfloat16
        desc = dict(op="triplet-bf16", Vs=Vs, Vp=Vp, D=D, B=B, kind=kind, steps=steps, same=(round(r1[1], 5), round(r2[1], 5)), worst=(round(r1[2], 2), round(r2[2], 2)), untouched=untouched)
    else:
        from esrecsys_amd.wikipedia.models import Glove
        from esrecsys_amd.wikipedia.train_cooccurence import train_epoch
        V = int(rng.choice([7, 300, 5000, 60000]))
        D = int(rng.choice([4, 8, 32, 64, 100, 128, 256, 512]))
        B = int(rng.choice([1, 31, 32, 33, 64, 777, 1000, 2048, 2049, 4096, 16384, 16385, 40000]))
        mode = str(rng.choice(["reference", "diagonal"]))
        lr = float(rng.choice([0.5, 4.0])) if kind != "same" else 0.5
        if 8 * V <= 2 * B:  # (as above)
            steps, lr = min(steps, 2), 0.5
        model = Glove(num_embeddings=V, features=D, loss_mode=mode, device=dev)
        params = model.init(case + 11, None)["params"]
        g = torch.Generator(device=dev).manual_seed(case)
        params["_bias"]["embedding"].copy_(torch.randn((V, 1), generator=g, device=dev) * 0.05)
        params["_token_embedding"]["embedding"] = (params["_token_embedding"]["embedding"] * 3.0).to(torch.bfloat16)
        state = TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(lr))
        emb, bias = params["_token_embedding"]["embedding"].float().double().cpu().numpy(), params["_bias"]["embedding"].double().cpu().numpy()
        emb0 = emb.copy()
        a_e, a_b = np.full_like(emb, 0.1), np.full_like(bias, 0.1)
        batches = [(ids(kind, V, (2, B)), np.exp(rng.uniform(np.log(0.1), np.log(1000.0), B)).astype(np.float32)) for _ in range(steps)]
        got = []
        state, _ = train_epoch(state, steps, iter([(torch.as_tensor(i, device=dev), torch.as_tensor(t, device=dev)) for i, t in batches]), losses_out=got)
        losses = got[0].cpu().numpy()
        for k, (inputs, target) in enumerate(batches):
            el, gdot, gs = o_glove.loss_and_grads(emb, bias, inputs, target.astype(F64), mode, F64)
            idl, rows, gb = o_glove.row_grads(emb, inputs, gdot, gs, F64)
            ok = ok and abs(float(losses[k]) - el) <= 1e-4 * max(abs(el), 1e-6)
            o_optim.sparse_adagrad_update_inplace(emb, a_e, idl, rows, lr)
            o_optim.sparse_adagrad_update_inplace(bias, a_b, idl, gb[:, None], lr)
            u = np.unique(idl)
            emb[u] = o_optim.round_bf16(emb[u])
        p = state.params
        touched = np.unique(np.concatenate([b[0].reshape(-1) for b in batches]))
        r1 = rows_ok(p["_token_embedding"]["embedding"], emb, touched, kind, 2 * B)
        mask = np.ones(V, bool); mask[touched] = False
        untouched = bool(np.array_equal(p["_token_embedding"]["embedding"].float().cpu().numpy()[mask], emb0[mask].astype(np.float32)))
        berr = float(np.abs(p["_bias"]["embedding"].double().cpu().numpy() - bias).max() / max(np.abs(bias).max(), 1e-30))
        ok = ok and r1[0] and untouched and berr <= 1e-3 and p["_token_embedding"]["embedding"].dtype == torch.bfloat16
        desc = dict(op="glove-bf16", V=V, D=D, B=B, mode=mode, kind=kind, steps=steps, lr=lr, same=round(r1[1], 5), worst=round(r1[2], 2), bias=berr, untouched=untouched)
    torch.cuda.synchronize()
    if os.environ.get("VERBOSE") == "1" or not ok:
        print("ok  " if ok else "MISMATCH", desc, flush=True)
    bad += 0 if ok else 1
    del state
    torch.cuda.empty_cache()
print("cases", N, "mismatches", bad)
