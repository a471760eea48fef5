"""Oracle: in-place CPU port of one training step (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Used ONLY by bench.py's ``cpu_baseline`` leg (kind "port") and by tests: the same arithmetic as
oracle/stl_head.py + oracle/glove.py + oracle/optim.py, written with torch-CPU tensor ops that update the tables in place
on the touched rows so that a 1M-row table is not copied every step.  This is a restatement, NOT
the reference's JAX/XLA-CPU executable (JAX is not installed).  fp32, all host
threads torch was given.
"""
import torch


def _reg_terms(e, lam):
    norm = e.norm(dim=1)
    on = norm > 1.0
    reg = torch.clamp(norm - 1.0, min=0.0).sum()
    dreg = torch.where(on[:, None], e / norm.clamp_min(1e-30)[:, None], torch.zeros_like(e)) * lam
    return reg, dreg


def sparse_adagrad_(table, accum, ids, rows, lr, eps=1e-7):
    """Row-sparse Adagrad in place: duplicates are summed first (index_add in occurrence order)."""
    uniq, inv = torch.unique(ids, return_inverse=True)
    g = torch.zeros((uniq.numel(), rows.shape[1]), dtype=rows.dtype)
    g.index_add_(0, inv, rows)
    acc = accum[uniq] + g * g
    accum[uniq] = acc
    table[uniq] -= lr * g * torch.rsqrt(acc + eps)


def inbatch_step_(scene_table, product_table, scene_accum, product_accum, scene_ids, pos_ids, lam, batch_size,
                  scale, lr):
    """One in-batch-softmax two-tower step (same math as esr_inbatch_softmax_fwd_bwd + sparse Adagrad)."""
    q = scene_table[scene_ids]
    c = product_table[pos_ids]
    S = scale * (q @ c.T)
    lse = torch.logsumexp(S, dim=1)
    ce = lse - torch.diagonal(S)
    rq, dq = _reg_terms(q, lam)
    rc, dc = _reg_terms(c, lam)
    loss = (ce.sum() + lam * (rq + rc)) / batch_size
    P = torch.exp(S - lse[:, None])
    P.diagonal().sub_(1.0)
    P /= batch_size
    gq = scale * (P @ c) + dq / batch_size
    gc = scale * (P.T @ q) + dc / batch_size
    sparse_adagrad_(scene_table, scene_accum, scene_ids, gq, lr)
    sparse_adagrad_(product_table, product_accum, pos_ids, gc, lr)
    return loss


def triplet_step_(scene_table, product_table, scene_accum, product_accum, scene_ids, pos_ids, neg_ids, lam,
                  batch_size, lr):
    """One reference-loss (triplet hinge + norm-excess) step -- pinterest/train_shop_the_look.py:93-109 with
    the id towers and the sparse optimizer of the build."""
    s, p, n = scene_table[scene_ids], product_table[pos_ids], product_table[neg_ids]
    margin = 1.0 + (s * n).sum(1) - (s * p).sum(1)
    m = (margin > 0).to(s.dtype)[:, None]
    rs, ds = _reg_terms(s, lam)
    rp, dp = _reg_terms(p, lam)
    rn, dn = _reg_terms(n, lam)
    loss = (torch.clamp(margin, min=0).sum() + lam * (rs + rp + rn)) / batch_size
    gs = (m * (n - p) + ds) / batch_size
    gp = (-m * s + dp) / batch_size
    gn = (m * s + dn) / batch_size
    sparse_adagrad_(scene_table, scene_accum, scene_ids, gs, lr)
    sparse_adagrad_(product_table, product_accum, torch.cat([pos_ids, neg_ids]), torch.cat([gp, gn]), lr)
    return loss


def glove_step_(emb, bias, emb_accum, bias_accum, inputs, target, lr):
    """One GloVe step with the reference's (B,B) loss (wikipedia/train_cooccurence.py:76-87) in O(B) form and
    the build's sparse Adagrad."""
    t1, t2 = inputs[0], inputs[1]
    e1, e2 = emb[t1], emb[t2]
    dot = (e1 * e2).sum(1)
    s = bias[t1, 0] + bias[t2, 0]
    B = float(dot.numel())
    w = torch.clamp(target / 100.0, max=1.0).pow(0.75)
    r = torch.log10(1.0 + target) - dot
    sbar = s.mean()
    loss = (w * (B * (r - sbar) ** 2 + ((s - sbar) ** 2).sum())).sum() / (B * B)
    gdot = -(2.0 * w / B) * (r - sbar)
    gs = -(2.0 / (B * B)) * ((w * r).sum() - s * w.sum())
    ids = torch.cat([t1, t2])
    sparse_adagrad_(emb, emb_accum, ids, torch.cat([gdot[:, None] * e2, gdot[:, None] * e1]), lr)
    sparse_adagrad_(bias, bias_accum, ids, torch.cat([gs, gs])[:, None], lr)
    return loss
