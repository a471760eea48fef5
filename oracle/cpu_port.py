"""Oracle: in-place CPU port of one training step (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Used ONLY by bench.py's ``cpu_baseline`` leg (kind "port") and by tests: the same arithmetic as
oracle/stl_head.py + oracle/glove.py + oracle/optim.py, written with torch-CPU tensor ops that update the tables in place
so that a 1M-row table is not copied every step.  This is a restatement, NOT the reference's JAX/XLA-CPU executable
(JAX is not installed).  fp32, the host threads torch was given.

Two variants of every step (SURVEY.md 8d "How the CPU path is timed beside it"):
  * sparse   -- index_add + row-sparse Adagrad on the touched rows: like for like with the HIP path;
  * dense    -- what the reference does (wikipedia/train_cooccurence.py:86-101,171): a dense V x D gradient
               (zero-fill + scatter-add, as JAX's autodiff yields for nn.Embed) and optax.adam over EVERY element.
The B x B buffers of the in-batch step are reused across steps (a fresh 256 MB tensor per step costs more in page
faults than the GEMM that fills it).
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


class DenseAdam:
    """optax.adam(lr) [upstream optax 0.1.2] over every element of one table, in place, with preallocated state and
    gradient buffers: the reference's optimizer (wikipedia/train_cooccurence.py:171, pinterest/train_shop_the_look.py:175)."""

    def __init__(self, table, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps = table, lr, b1, b2, eps
        self.mu, self.nu = torch.zeros_like(table), torch.zeros_like(table)
        self.g, self.tmp = torch.zeros_like(table), torch.empty_like(table)
        self.t = 0

    def step_(self, ids, rows):
        """dense grad = zeros ; grad[ids] += rows (the scatter-add of nn.Embed's VJP) ; Adam on every element."""
        g = self.g
        g.zero_()
        g.index_add_(0, ids, rows)
        self.t += 1
        self.mu.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
        self.nu.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
        bc1, bc2 = 1.0 - self.b1 ** self.t, 1.0 - self.b2 ** self.t
        torch.div(self.nu, bc2, out=self.tmp)
        self.tmp.sqrt_().add_(self.eps)
        self.p.addcdiv_(self.mu, self.tmp, value=-self.lr / bc1)


def _inbatch_grads(q, c, lam, batch_size, scale, bufs):
    B = q.shape[0]
    if bufs.get("S") is None or bufs["S"].shape[0] != B:
        bufs["S"] = torch.empty((B, B), dtype=q.dtype)
    S = bufs["S"]
    torch.matmul(q, c.T, out=S)
    S.mul_(scale)
    lse = torch.logsumexp(S, dim=1)
    ce = lse - torch.diagonal(S)
    rq, dq = _reg_terms(q, lam)
    rc, dc = _reg_terms(c, lam)
    loss = (ce.sum() + lam * (rq + rc)) / batch_size
    S.sub_(lse[:, None]).exp_()            # S becomes P
    S.diagonal().sub_(1.0)
    S.div_(batch_size)
    gq = scale * (S @ c) + dq / batch_size
    gc = scale * (S.T @ q) + dc / batch_size
    return loss, gq, gc


def inbatch_step_(scene_table, product_table, scene_accum, product_accum, scene_ids, pos_ids, lam, batch_size,
                  scale, lr, bufs=None):
    """One in-batch-softmax two-tower step (same math as esr_inbatch_softmax_fwd_bwd + sparse Adagrad)."""
    loss, gq, gc = _inbatch_grads(scene_table[scene_ids], product_table[pos_ids], lam, batch_size, scale,
                                  bufs if bufs is not None else {})
    sparse_adagrad_(scene_table, scene_accum, scene_ids, gq, lr)
    sparse_adagrad_(product_table, product_accum, pos_ids, gc, lr)
    return loss


def inbatch_step_dense_adam_(scene_adam, product_adam, scene_ids, pos_ids, lam, batch_size, scale, bufs=None):
    """The same loss with the reference's update: dense gradients + dense Adam on both towers."""
    loss, gq, gc = _inbatch_grads(scene_adam.p[scene_ids], product_adam.p[pos_ids], lam, batch_size, scale,
                                  bufs if bufs is not None else {})
    scene_adam.step_(scene_ids, gq)
    product_adam.step_(pos_ids, gc)
    return loss


def _triplet_grads(s, p, n, lam, batch_size):
    margin = 1.0 + (s * n).sum(1) - (s * p).sum(1)
    m = (margin > 0).to(s.dtype)[:, None]
    rs, ds = _reg_terms(s, lam)
    rp, dp = _reg_terms(p, lam)
    rn, dn = _reg_terms(n, lam)
    loss = (torch.clamp(margin, min=0).sum() + lam * (rs + rp + rn)) / batch_size
    return loss, (m * (n - p) + ds) / batch_size, (-m * s + dp) / batch_size, (m * s + dn) / batch_size


def triplet_step_(scene_table, product_table, scene_accum, product_accum, scene_ids, pos_ids, neg_ids, lam,
                  batch_size, lr):
    """One reference-loss (triplet hinge + norm-excess) step -- pinterest/train_shop_the_look.py:93-109 with
    the id towers and the sparse optimizer of the build."""
    loss, gs, gp, gn = _triplet_grads(scene_table[scene_ids], product_table[pos_ids], product_table[neg_ids], lam,
                                      batch_size)
    sparse_adagrad_(scene_table, scene_accum, scene_ids, gs, lr)
    sparse_adagrad_(product_table, product_accum, torch.cat([pos_ids, neg_ids]), torch.cat([gp, gn]), lr)
    return loss


def triplet_step_dense_adam_(scene_adam, product_adam, scene_ids, pos_ids, neg_ids, lam, batch_size):
    """pinterest/train_shop_the_look.py:93-109 as the reference runs it: value_and_grad gives dense tower-table
    gradients, state.apply_gradients runs optax.adam over every element (:108, :175)."""
    loss, gs, gp, gn = _triplet_grads(scene_adam.p[scene_ids], product_adam.p[pos_ids], product_adam.p[neg_ids], lam,
                                      batch_size)
    scene_adam.step_(scene_ids, gs)
    product_adam.step_(torch.cat([pos_ids, neg_ids]), torch.cat([gp, gn]))
    return loss


def _glove_grads(emb, bias, inputs, target):
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
    return loss, ids, torch.cat([gdot[:, None] * e2, gdot[:, None] * e1]), torch.cat([gs, gs])[:, None]


def glove_step_(emb, bias, emb_accum, bias_accum, inputs, target, lr):
    """One GloVe step with the reference's (B,B) loss (wikipedia/train_cooccurence.py:76-87) in O(B) form and
    the build's sparse Adagrad."""
    loss, ids, rows, brows = _glove_grads(emb, bias, inputs, target)
    sparse_adagrad_(emb, emb_accum, ids, rows, lr)
    sparse_adagrad_(bias, bias_accum, ids, brows, lr)
    return loss


def glove_step_dense_adam_(emb_adam, bias_adam, inputs, target):
    """wikipedia/train_cooccurence.py:71-101 as the reference runs it: apply_model's dense gradient tree, then
    update_model = optax.adam over every element of both tables (O(V D) per step whatever the batch)."""
    loss, ids, rows, brows = _glove_grads(emb_adam.p, bias_adam.p, inputs, target)
    emb_adam.step_(ids, rows)
    bias_adam.step_(ids, brows)
    return loss
