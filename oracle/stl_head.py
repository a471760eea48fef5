"""Oracle: Shop-The-Look score head and losses (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows pinterest/models.py:63-74 (score head; the CNN towers at :23-46 are
out of scope and are replaced by id-embedding gathers) and
pinterest/train_shop_the_look.py:93-122 (train_step / eval_step losses).
The in-batch sampled-softmax variant is build-defined (north_star), it has
no counterpart in the reference.
"""
import numpy as np


def scores(scene_e, pos_e, neg_e, dtype=np.float64):
    """STLModel.__call__ score head -- pinterest/models.py:67-72."""
    s, p, n = (np.asarray(x).astype(dtype) for x in (scene_e, pos_e, neg_e))
    pos_score = np.sum(s * p, axis=-1, dtype=dtype)
    neg_score = np.sum(s * n, axis=-1, dtype=dtype)
    return pos_score, neg_score


def _reg(e, dtype):
    """reg_fn -- pinterest/train_shop_the_look.py:100-101: relu(||e||_2 - 1) per row."""
    norm = np.sqrt(np.sum(np.square(e), axis=-1, dtype=dtype))
    return np.maximum(norm - dtype(1.0), dtype(0.0)), norm


def triplet_loss_and_grads(scene_e, pos_e, neg_e, regularization, batch_size, dtype=np.float64):
    """loss_fn + value_and_grad w.r.t. the three embedding matrices.

    pinterest/train_shop_the_look.py:99-104:
      triplet = sum_b relu(1 + neg_b - pos_b)
      reg     = sum_b [reg(scene_b) + reg(pos_b) + reg(neg_b)]
      loss    = (triplet + regularization * reg) / batch_size
    Gradients: m_b = [1 + neg_b - pos_b > 0]  (relu'(0) = 0 [upstream])
      d/dscene = (m (neg_e - pos_e) + lam [||s||>1] s/||s||) / B
      d/dpos_e = (-m s + lam [||p||>1] p/||p||) / B
      d/dneg_e = ( m s + lam [||n||>1] n/||n||) / B
    """
    s, p, n = (np.asarray(x).astype(dtype) for x in (scene_e, pos_e, neg_e))
    lam = dtype(regularization)
    bs = dtype(batch_size)
    pos_score, neg_score = scores(s, p, n, dtype)
    margin = dtype(1.0) + neg_score - pos_score
    triplet = np.sum(np.maximum(margin, dtype(0.0)), dtype=dtype)
    (rs, ns), (rp, npn), (rn, nn_) = _reg(s, dtype), _reg(p, dtype), _reg(n, dtype)
    reg = np.sum(rs + rp + rn, dtype=dtype)
    loss = (triplet + lam * reg) / bs

    m = (margin > 0).astype(dtype)[:, None]

    def dreg(e, norm):
        on = (norm > dtype(1.0)).astype(dtype)[:, None]
        with np.errstate(invalid="ignore", divide="ignore"):
            unit = e / norm[:, None]
        return np.where(on > 0, unit, dtype(0.0)) * lam

    g_s = (m * (n - p) + dreg(s, ns)) / bs
    g_p = (-m * s + dreg(p, npn)) / bs
    g_n = (m * s + dreg(n, nn_)) / bs
    return loss, g_s, g_p, g_n


def eval_loss(scene_e, pos_e, neg_e, dtype=np.float64):
    """eval_step -- pinterest/train_shop_the_look.py:111-122: sum_b relu(1 + neg - pos); no reg, not / B."""
    pos_score, neg_score = scores(scene_e, pos_e, neg_e, dtype)
    return np.sum(np.maximum(dtype(1.0) + neg_score - pos_score, dtype(0.0)), dtype=dtype)


def inbatch_softmax_loss_and_grads(query_e, cand_e, regularization, batch_size, scale=1.0,
                                   dtype=np.float64):
    """In-batch-negative sampled softmax (build-defined; north_star, no reference counterpart).

    S[i, j] = scale * q_i . c_j ;   ce_i = logsumexp_j S[i, j] - S[i, i]
    loss = (sum_i ce_i + lam * sum_i [reg(q_i) + reg(c_i)]) / batch_size
    The normalisation and the norm-excess regulariser mirror the reference's
    train_step (pinterest/train_shop_the_look.py:100-104) with the triplet hinge
    replaced by the in-batch cross entropy.
      dS = (softmax(S) - I) / B ; dQ = scale * dS C + dreg(Q) ; dC = scale * dS^T Q + dreg(C)
    Returns (loss, lse[B], gQ[B, D], gC[B, D]).
    """
    q = np.asarray(query_e).astype(dtype)
    c = np.asarray(cand_e).astype(dtype)
    lam = dtype(regularization)
    bs = dtype(batch_size)
    S = dtype(scale) * (q @ c.T)
    mx = np.max(S, axis=1, keepdims=True)
    ex = np.exp(S - mx)
    den = np.sum(ex, axis=1, keepdims=True, dtype=dtype)
    lse = (mx + np.log(den))[:, 0]
    ce = lse - np.diagonal(S)
    (rq, nq), (rc, nc) = _reg(q, dtype), _reg(c, dtype)
    loss = (np.sum(ce, dtype=dtype) + lam * np.sum(rq + rc, dtype=dtype)) / bs

    P = ex / den
    dS = (P - np.eye(S.shape[0], dtype=dtype)) / bs

    def dreg(e, norm):
        on = (norm > dtype(1.0))[:, None]
        with np.errstate(invalid="ignore", divide="ignore"):
            unit = e / norm[:, None]
        return np.where(on, unit, dtype(0.0)) * lam / bs

    g_q = dtype(scale) * (dS @ c) + dreg(q, nq)
    g_c = dtype(scale) * (dS.T @ q) + dreg(c, nc)
    return loss, lse, g_q, g_c
