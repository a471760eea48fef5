"""Generates the golden fixtures in this directory (run from the repo root:
``python tests/golden/make_golden.py``).

The reference cannot be run here (JAX/Flax/Optax are not installed) and holds no test
vectors of its own, so every fixture is produced by the fp64 oracle (oracle/*.py) and, before it
is written, cross-checked against the independent torch-autograd transliteration
(oracle/autograd_ref.py) to <= 1e-12.  Fixtures are data only: seeded inputs + expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import autograd_ref, glove, optim, stl_head, topk  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
F64 = np.float64


def _close(a, b, tol=1e-12):
    err = float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0
    assert err <= tol, err


def zipf_ids(rng, V, n):
    p = 1.0 / np.arange(1, V + 1)
    p /= p.sum()
    perm = rng.permutation(V)
    return perm[rng.choice(V, size=n, p=p)]


def glove_case(name, V, D, B, id_kind, seed):
    rng = np.random.default_rng(seed)
    emb = (rng.standard_normal((V, D)) / np.sqrt(D)).astype(np.float32)
    bias = (rng.standard_normal((V, 1)) * 0.05).astype(np.float32)
    if id_kind == "uniform":
        inputs = rng.integers(0, V, (2, B))
        inputs[0, 0], inputs[1, 1] = 0, V - 1  # edge ids
    elif id_kind == "same":
        inputs = np.full((2, B), 7)
    else:
        inputs = zipf_ids(rng, V, 2 * B).reshape(2, B)
    inputs = inputs.astype(np.int32)
    target = rng.uniform(0.01, 300.0, B).astype(np.float32)  # both weight branches (c < 100, c >= 100)
    out = {"emb": emb, "bias": bias, "inputs": inputs, "target": target}
    for mode in ("reference", "diagonal"):
        grads, loss = glove.dense_grads(emb.astype(F64), bias.astype(F64), inputs, target, mode, F64)
        l2, ge, gb = autograd_ref.glove_value_and_grad(emb, bias, inputs, target, mode)
        _close(loss, l2), _close(grads["_token_embedding"]["embedding"], ge), _close(grads["_bias"]["embedding"], gb)
        out["loss_" + mode] = np.float64(loss)
        out["gemb_" + mode] = grads["_token_embedding"]["embedding"]
        out["gbias_" + mode] = grads["_bias"]["embedding"]
    out["pred_reference"] = glove.forward(emb.astype(F64), bias.astype(F64), inputs, F64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def stl_case(name, B, D, lam, seed):
    rng = np.random.default_rng(seed)
    # norms on both sides of 1, margins on both sides of 0
    s = rng.standard_normal((B, D)) * rng.uniform(0.05, 0.6, (B, 1))
    p = rng.standard_normal((B, D)) * rng.uniform(0.05, 0.6, (B, 1))
    n = rng.standard_normal((B, D)) * rng.uniform(0.05, 0.6, (B, 1))
    p[: B // 4] = s[: B // 4] * 3.0  # strongly positive pos score -> inactive hinge for some rows
    s, p, n = (x.astype(np.float32) for x in (s, p, n))
    loss, gs, gp, gn = stl_head.triplet_loss_and_grads(s, p, n, lam, B, F64)
    l2, a, b, c = autograd_ref.stl_value_and_grad(s, p, n, lam, B)
    _close(loss, l2), _close(gs, a), _close(gp, b), _close(gn, c)
    ps, ns = stl_head.scores(s, p, n, F64)
    margin = 1.0 + ns - ps
    assert (margin > 0).any() and (margin < 0).any()
    norms = np.sqrt((s.astype(F64) ** 2).sum(-1))
    assert (norms > 1).any() and (norms < 1).any()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), scene=s, pos=p, neg=n, lam=np.float64(lam),
                        batch_size=np.float64(B), loss=np.float64(loss), g_scene=gs, g_pos=gp, g_neg=gn,
                        pos_score=ps, neg_score=ns, eval_loss=np.float64(stl_head.eval_loss(s, p, n, F64)))


def inbatch_case(name, B, D, lam, scale, seed):
    rng = np.random.default_rng(seed)
    q = (rng.standard_normal((B, D)) * rng.uniform(0.05, 0.5, (B, 1))).astype(np.float32)
    c = (rng.standard_normal((B, D)) * rng.uniform(0.05, 0.5, (B, 1))).astype(np.float32)
    loss, lse, gq, gc = stl_head.inbatch_softmax_loss_and_grads(q, c, lam, B, scale, F64)
    l2, a, b = autograd_ref.inbatch_value_and_grad(q, c, lam, B, scale)
    _close(loss, l2), _close(gq, a), _close(gc, b)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), q=q, c=c, lam=np.float64(lam), scale=np.float64(scale),
                        batch_size=np.float64(B), loss=np.float64(loss), lse=lse, g_q=gq, g_c=gc)


def optim_case(name, V, D, seed):
    rng = np.random.default_rng(seed)
    p0 = rng.standard_normal((V, D)).astype(np.float32)
    grads = [(rng.standard_normal((V, D)) * 0.1).astype(np.float32) for _ in range(3)]
    # Adam, 3 dense steps
    st, p = optim.adam_init(p0.astype(F64)), p0.astype(F64)
    adam_traj = []
    for g in grads:
        p, st = optim.adam_update(p, g.astype(F64), st, 1e-3, dtype=F64)
        adam_traj.append(p.copy())
    _close(p, autograd_ref.adam_steps(p0, grads, 1e-3))
    # sparse Adagrad, 3 steps with duplicate ids
    n = 3 * V // 2
    ids = [rng.integers(0, V, n).astype(np.int32) for _ in range(3)]
    rows = [(rng.standard_normal((n, D)) * 0.1).astype(np.float32) for _ in range(3)]
    p, a = p0.astype(F64), optim.adagrad_init(p0.astype(F64))
    dense = []
    for i, r in zip(ids, rows):
        p, a = optim.sparse_adagrad_update(p, a, i, r.astype(F64), 0.05, dtype=F64)
        g = np.zeros((V, D))
        np.add.at(g, i, r.astype(F64))
        dense.append(g)
    pd, ad = autograd_ref.adagrad_steps(p0, dense, 0.05)
    _close(p, pd), _close(a, ad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), p0=p0, adam_grads=np.stack(grads),
                        adam_params=np.stack(adam_traj), ada_ids=np.stack(ids), ada_rows=np.stack(rows),
                        ada_param=p, ada_accum=a)


def topk_case(name, N, D, k, seed):
    rng = np.random.default_rng(seed)
    cand = rng.integers(-3, 4, (N, D)).astype(np.float32)  # small integers: exact scores, many ties
    q = rng.integers(-3, 4, (1, D)).astype(np.float32)
    vals, idx = topk.find_top_k(q, cand, k, F64)
    tok = rng.integers(0, N, 5).astype(np.int32)
    scores, indices = glove.find_knn(cand, tok, F64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), cand=cand, query=q, k=np.int64(k), topk_scores=vals,
                        topk_indices=idx, token=tok, knn_scores=scores, knn_indices=indices)


if __name__ == "__main__":
    glove_case("glove_uniform_d16_b64", 1000, 16, 64, "uniform", 11)
    glove_case("glove_uniform_d64_b128", 1000, 64, 128, "uniform", 12)
    glove_case("glove_same_d16_b64", 1000, 16, 64, "same", 13)
    glove_case("glove_zipf_d64_b128", 1000, 64, 128, "zipf", 14)
    stl_case("stl_b32_d8_lam0", 32, 8, 0.0, 21)
    stl_case("stl_b32_d8_lam01", 32, 8, 0.1, 22)
    stl_case("stl_b128_d32_lam01", 128, 32, 0.1, 23)
    inbatch_case("inbatch_b64_d32", 64, 32, 0.1, 1.0, 31)
    inbatch_case("inbatch_b96_d64_scale4", 96, 64, 0.1, 4.0, 32)
    inbatch_case("inbatch_b320_d128", 320, 128, 0.05, 2.0, 33)
    optim_case("optim_v64_d8", 64, 8, 41)
    topk_case("topk_n500_d8_k10", 500, 8, 10, 51)
    print("wrote fixtures to", OUT)
