"""Oracle: GloVe-style co-occurrence model (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows wikipedia/models.py:15-55 and wikipedia/train_cooccurence.py:76-97.
Every function takes a ``dtype`` (np.float64 for the reference-grade answer,
np.float32 to mimic the reference's own precision) and does all arithmetic
in that dtype.
"""
import numpy as np


def init_params(num_embeddings, features, seed, dtype=np.float32):
    """Glove.setup -- wikipedia/models.py:15-19.

    Token table: nn.Embed default init, variance_scaling(1.0, 'fan_in',
    'normal', out_axis=0) => N(0, 1/features) [upstream flax 0.5.2].  Bias
    table: zeros (models.py:18-19).  The JAX threefry stream cannot be
    reproduced without JAX, so the draw itself is oracle-defined (NumPy
    PCG64); only the distribution is pinned.
    """
    rng = np.random.default_rng(seed)
    emb = rng.standard_normal((num_embeddings, features)) / np.sqrt(features)
    return {
        "_token_embedding": {"embedding": emb.astype(dtype)},
        "_bias": {"embedding": np.zeros((num_embeddings, 1), dtype=dtype)},
    }


def gather_rows(table, ids):
    """nn.Embed.__call__ == jnp.take(embedding, ids, axis=0); bit-exact row copy."""
    return table[np.asarray(ids, dtype=np.int64)]


def pair_terms(emb, bias, inputs, dtype=np.float64):
    """The per-pair pieces of Glove.__call__ -- wikipedia/models.py:30-36.

    Returns dot[j] = sum_d E[t1[j],d] * E[t2[j],d] and s[i] = Bias[t1[i]] + Bias[t2[i]].
    """
    t1, t2 = np.asarray(inputs[0], np.int64), np.asarray(inputs[1], np.int64)
    e1 = emb[t1].astype(dtype)
    e2 = emb[t2].astype(dtype)
    dot = np.sum(e1 * e2, axis=-1, dtype=dtype)
    s = (bias[t1, 0].astype(dtype) + bias[t2, 0].astype(dtype))
    return dot, s


def forward(emb, bias, inputs, dtype=np.float64):
    """Glove.__call__ -- wikipedia/models.py:30-38.

    ``output = dot + bias1 + bias2`` with dot (B,), bias1/bias2 (B,1) broadcasts
    to (B,B): output[i, j] = dot[j] + bias1[i] + bias2[i]  (SURVEY.md 8a-G2).
    """
    dot, s = pair_terms(emb, bias, inputs, dtype)
    return dot[None, :] + s[:, None]


def loss_weights(target, dtype=np.float64):
    """weight and log_target -- wikipedia/train_cooccurence.py:79-82."""
    c = np.asarray(target).astype(dtype)
    w = np.minimum(np.ones_like(c), c / dtype(100.0))
    w = np.power(w, dtype(0.75))
    lt = np.log10(dtype(1.0) + c)
    return w, lt


def loss_literal(emb, bias, inputs, target, dtype=np.float64):
    """glove_loss exactly as written -- wikipedia/train_cooccurence.py:78-83.

    mean over the (B,B) broadcast of square(log_target - predicted) * weight.
    O(B^2); use for small B only.
    """
    pred = forward(emb, bias, inputs, dtype)
    w, lt = loss_weights(target, dtype)
    return np.mean(np.square(lt - pred) * w, dtype=dtype)


def loss_and_grads(emb, bias, inputs, target, mode="reference", dtype=np.float64):
    """value_and_grad(glove_loss) in closed form -- wikipedia/train_cooccurence.py:86-87.

    mode "reference": L = (1/B^2) sum_i sum_j w_j (r_j - s_i)^2, r_j = lt_j - dot_j
      dL/ddot_j = -(2 w_j / B^2) (B r_j - sum_i s_i)
      dL/ds_i   = -(2 / B^2) (sum_j w_j r_j - s_i sum_j w_j)
    mode "diagonal" (the textbook GloVe loss, build-defined option):
      L = (1/B) sum_j w_j (r_j - s_j)^2 ; dL/ddot_j = dL/ds_j = -(2 w_j / B)(r_j - s_j)

    Returns (loss, gdot[B], gs[B]) -- the per-pair cotangents.  Row gradients
    follow by the chain rule in ``row_grads``.
    """
    dot, s = pair_terms(emb, bias, inputs, dtype)
    w, lt = loss_weights(target, dtype)
    B = dtype(dot.shape[0])
    r = lt - dot
    if mode == "reference":
        sbar = np.sum(s, dtype=dtype) / B
        # centred form of the double sum (SURVEY.md section 7 "hard parts")
        loss = (np.sum(w * (B * np.square(r - sbar) + np.sum(np.square(s - sbar), dtype=dtype)),
                       dtype=dtype) / (B * B))
        gdot = -(dtype(2.0) * w / (B * B)) * (B * r - np.sum(s, dtype=dtype))
        gs = -(dtype(2.0) / (B * B)) * (np.sum(w * r, dtype=dtype) - s * np.sum(w, dtype=dtype))
    elif mode == "diagonal":
        loss = np.sum(w * np.square(r - s), dtype=dtype) / B
        gdot = -(dtype(2.0) * w / B) * (r - s)
        gs = gdot.copy()
    else:
        raise ValueError(mode)
    return loss, gdot, gs


def row_grads(emb, inputs, gdot, gs, dtype=np.float64):
    """Per-occurrence row gradients (what the HIP kernel emits before the scatter).

    gE occurrence for t1[j] is gdot[j] * E[t2[j]]; for t2[j] it is gdot[j] * E[t1[j]].
    Returns (ids[2B], gE_rows[2B, D], gBias_rows[2B]) with occurrence order
    [t1[0..B-1], t2[0..B-1]] -- the row order of ``inputs`` flattened.
    """
    t1, t2 = np.asarray(inputs[0], np.int64), np.asarray(inputs[1], np.int64)
    e1 = emb[t1].astype(dtype)
    e2 = emb[t2].astype(dtype)
    g1 = gdot[:, None].astype(dtype) * e2
    g2 = gdot[:, None].astype(dtype) * e1
    ids = np.concatenate([t1, t2])
    return ids, np.concatenate([g1, g2], axis=0), np.concatenate([gs, gs]).astype(dtype)


def dense_grads(emb, bias, inputs, target, mode="reference", dtype=np.float64):
    """(grads, loss) with the reference's DENSE gradient tree -- train_cooccurence.py:86-89.

    Duplicated ids accumulate (scatter-add), in occurrence order.
    """
    loss, gdot, gs = loss_and_grads(emb, bias, inputs, target, mode, dtype)
    ids, ge, gb = row_grads(emb, inputs, gdot, gs, dtype)
    g_emb = np.zeros(emb.shape, dtype=dtype)
    g_bias = np.zeros(bias.shape, dtype=dtype)
    np.add.at(g_emb, ids, ge)
    np.add.at(g_bias[:, 0], ids, gb)
    grads = {"_token_embedding": {"embedding": g_emb}, "_bias": {"embedding": g_bias}}
    return grads, loss


def score_all(emb, token, dtype=np.float64):
    """Glove.score_all -- wikipedia/models.py:50-55: scores[v, t] = E[v] . E[token[t]]; no bias."""
    e1 = emb[np.asarray(token, np.int64)].astype(dtype)  # (T, D)
    return emb.astype(dtype) @ e1.T  # (V, T)


def find_knn(emb, token, dtype=np.float64):
    """find_knn -- wikipedia/train_cooccurence.py:91-97.

    indices = jnp.argsort(scores, axis=0): ascending, stable [upstream: jnp.argsort
    lowers to lax.sort with is_stable=True].
    """
    scores = score_all(emb, token, dtype)
    indices = np.argsort(scores, axis=0, kind="stable").astype(np.int32)
    return scores, indices
