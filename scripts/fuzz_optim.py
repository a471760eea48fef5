"""Randomised check of the row-sparse optimizers behind TrainState.apply_gradients (sparse Adagrad, SGD, SGD + momentum
lazy and eager, dense Adam) on one [V, D] table against their dense fp64 statements (optax rules as restated in
oracle/optim.py): random V / D / occurrences per step, hot ids, fp32 and bf16 tables, up to a few hundred steps for the
lazy momentum (rows untouched for long stretches).  SEED, CASES."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esrecsys_amd import TrainState, optim
from esrecsys_amd.train_state import RowGrads
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N_ = int(os.environ.get("CASES", "40"))
bad = 0
def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))
for case in range(N_):
    V = int(rng.choice([1, 7, 300, 5000, 100000]))
    D = int(rng.choice([1, 4, 6, 32, 100, 128, 256]))
    n = int(rng.choice([1, 31, 33, 500, 4096, 20000]))
    kind = str(rng.choice(["adagrad", "sgd", "momentum", "momentum_eager", "adam"]))
    bf16 = kind == "adagrad" and D % 8 == 0 and rng.random() < 0.3
    steps = int(rng.integers(1, 6)) if kind != "momentum" else int(rng.choice([1, 5, 70, 200]))
    lr, mom = 0.05, 0.9
    p0 = (rng.standard_normal((V, D)) * 0.3).astype(np.float32)
    pt = torch.from_numpy(p0).to(dev)
    if bf16:
        pt = pt.to(torch.bfloat16); p0 = pt.float().cpu().numpy()
    tx = {"adagrad": optim.sparse_adagrad(lr), "sgd": optim.sgd(lr), "momentum": optim.sgd(lr, mom),
          "momentum_eager": optim.sgd(lr, mom, lazy=False), "adam": optim.adam(lr)}[kind]
    state = TrainState.create(apply_fn=None, params={"t": {"embedding": pt}}, tx=tx)
    p = p0.astype(np.float64); acc = np.full_like(p, 0.1); tr = np.zeros_like(p); mu = np.zeros_like(p); nu = np.zeros_like(p)
    adam_keep = np.ones((V, D), bool)
    for step in range(1, steps + 1):
        ids = rng.integers(0, V, n)
        if rng.random() < 0.4:
            ids[rng.random(n) < 0.5] = rng.integers(0, V)
        g = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
        grads = {"t": {"embedding": RowGrads([torch.from_numpy(ids.astype(np.int32)).to(dev)], torch.from_numpy(g).to(dev), (V, D))}}
        state = state.apply_gradients(grads=grads)
        G = np.zeros_like(p); np.add.at(G, ids, g.astype(np.float64))
        touched = np.zeros(V, bool); touched[ids] = True
        adam_keep &= ~(touched[:, None] & (np.abs(G) < 1e-6))
        if kind == "adagrad":
            t = G != 0; rows = np.zeros(V, bool); rows[ids] = True
            acc[rows] += G[rows] ** 2
            upd = p[rows] - lr * G[rows] / np.sqrt(acc[rows] + 1e-7)
            p[rows] = upd.astype(np.float32).astype(np.float64) if not bf16 else \
                torch.from_numpy(upd).to(torch.bfloat16).double().numpy()
        elif kind == "sgd":
            p -= lr * G
        elif kind in ("momentum", "momentum_eager"):
            tr = mom * tr + G; p -= lr * tr
        else:
            mu = 0.9 * mu + 0.1 * G; nu = 0.999 * nu + 0.001 * G * G
            p -= lr * (mu / (1 - 0.9 ** step)) / (np.sqrt(nu / (1 - 0.999 ** step)) + 1e-8)
    got = state.params["t"]["embedding"].float().cpu().numpy().astype(np.float64)
    tol = 2e-2 if bf16 else (2e-5 if kind != "adam" else 1e-4)
    if kind == "adam":
        # Adam's step is lr * m / (sqrt(v) + 1e-8): where a row's gradients cancel to |G| < 1e-6 the f32 sum's last bits
        # decide the step (seeds 919, 4001: 1.2e-4 / 1.7e-4 of the table's scale on one such element) -- elements that
        # met such a sum in any step are not compared
        got, p = np.where(adam_keep, got, 0.0), np.where(adam_keep, p, 0.0)
    e = rel(got, p)
    ok = np.isfinite(e) and e <= tol
    if os.environ.get("VERBOSE") == "1" or not ok:
        print("ok  " if ok else "MISMATCH", dict(kind=kind, V=V, D=D, n=n, steps=steps, bf16=bf16, e=e), flush=True)
    bad += 0 if ok else 1
print("cases", N_, "mismatches", bad)
