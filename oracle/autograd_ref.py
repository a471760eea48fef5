"""Oracle cross-check: literal torch-CPU-autograd transliteration (TEST INFRASTRUCTURE).

An independent second derivation used only to pin the closed-form oracle
(oracle/glove.py, oracle/stl_head.py, oracle/optim.py): every function here
writes the reference's jnp expression one-for-one with torch ops on CPU in
fp64 and lets autograd differentiate it, including the (B,B) broadcasting
quirk of the GloVe loss.  tests/test_oracle.py requires agreement <= 1e-12.
"""
import torch


def glove_value_and_grad(emb, bias, inputs, target, mode="reference"):
    """wikipedia/models.py:30-38 + wikipedia/train_cooccurence.py:78-87, literally."""
    emb = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
    bias = torch.tensor(bias, dtype=torch.float64, requires_grad=True)
    token1 = torch.as_tensor(inputs[0], dtype=torch.long)
    token2 = torch.as_tensor(inputs[1], dtype=torch.long)
    target = torch.tensor(target, dtype=torch.float64)
    embed1 = emb[token1]            # self._token_embedding(token1)
    bias1 = bias[token1]            # (B, 1)
    embed2 = emb[token2]
    bias2 = bias[token2]
    dot = (embed1 * embed2).sum(-1)  # vmap(jnp.dot)
    if mode == "reference":
        output = dot + bias1 + bias2     # (B,) + (B,1) + (B,1) -> (B,B)
    else:
        output = dot + bias1[:, 0] + bias2[:, 0]
    ones = torch.ones_like(target)
    weight = torch.minimum(ones, target / 100.0)
    weight = torch.pow(weight, 0.75)
    log_target = torch.log10(1.0 + target)
    loss = torch.mean(torch.square(log_target - output) * weight)
    loss.backward()
    return loss.item(), emb.grad.numpy(), bias.grad.numpy()


def stl_value_and_grad(scene_e, pos_e, neg_e, regularization, batch_size):
    """pinterest/models.py:67-72 + pinterest/train_shop_the_look.py:99-104, literally."""
    s = torch.tensor(scene_e, dtype=torch.float64, requires_grad=True)
    p = torch.tensor(pos_e, dtype=torch.float64, requires_grad=True)
    n = torch.tensor(neg_e, dtype=torch.float64, requires_grad=True)
    pos_score = (s * p).sum(-1)
    neg_score = (s * n).sum(-1)
    triplet_loss = torch.relu(1.0 + neg_score - pos_score).sum()

    def reg_fn(embed):
        return torch.relu(torch.sqrt(torch.square(embed).sum(-1)) - 1.0)

    reg_loss = (reg_fn(s) + reg_fn(p) + reg_fn(n)).sum()
    loss = (triplet_loss + regularization * reg_loss) / batch_size
    loss.backward()
    return loss.item(), s.grad.numpy(), p.grad.numpy(), n.grad.numpy()


def inbatch_value_and_grad(query_e, cand_e, regularization, batch_size, scale=1.0):
    """Build-defined in-batch softmax head, written with torch primitives."""
    q = torch.tensor(query_e, dtype=torch.float64, requires_grad=True)
    c = torch.tensor(cand_e, dtype=torch.float64, requires_grad=True)
    S = scale * (q @ c.T)
    ce = torch.logsumexp(S, dim=1) - torch.diagonal(S)

    def reg_fn(embed):
        return torch.relu(torch.sqrt(torch.square(embed).sum(-1)) - 1.0)

    loss = (ce.sum() + regularization * (reg_fn(q) + reg_fn(c)).sum()) / batch_size
    loss.backward()
    return loss.item(), q.grad.numpy(), c.grad.numpy()


def adam_steps(param, grads, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam has the same update rule as optax.adam (eps outside the sqrt,
    bias-corrected moments); used as a third opinion on oracle/optim.adam_update."""
    p = torch.tensor(param, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps)
    for g in grads:
        p.grad = torch.tensor(g, dtype=torch.float64)
        opt.step()
    return p.detach().numpy()


def adagrad_steps(param, grads, lr, initial_accumulator_value=0.1, eps=1e-7):
    """Dense optax.adagrad written out with torch ops: acc += g^2 ; p -= lr g rsqrt(acc + eps)."""
    p = torch.tensor(param, dtype=torch.float64)
    acc = torch.full_like(p, initial_accumulator_value)
    for g in grads:
        g = torch.tensor(g, dtype=torch.float64)
        acc = acc + g * g
        p = p - lr * g * torch.where(acc > 0, torch.rsqrt(acc + eps), torch.zeros_like(acc))
    return p.numpy(), acc.numpy()


def spotify_value_and_grad(album_table, artist_table, x, regularization):
    """spotify/models.py:37-90 + spotify/train_spotify.py:78-107 written one-for-one with torch ops; autograd
    differentiates it.  torch's amax / amin backward splits the gradient evenly over ties like JAX's max / min;
    torch.relu'(0) = 0 like jax.nn.relu."""
    at = torch.tensor(album_table, dtype=torch.float64, requires_grad=True)
    rt = torch.tensor(artist_table, dtype=torch.float64, requires_grad=True)
    L = lambda v: torch.as_tensor(v, dtype=torch.long)  # noqa: E731

    def get_embeddings(album, artist):
        album_modded = torch.remainder(L(album), 100000)
        return torch.cat([at[album_modded], rt[L(artist)]], dim=-1)

    context_embed = get_embeddings(x["album_context"], x["artist_context"])
    next_embed = get_embeddings(x["next_album"], x["next_artist"])
    neg_embed = get_embeddings(x["neg_album"], x["neg_artist"])
    pos_affinity = torch.amax(next_embed @ context_embed.T, dim=-1)
    pos_affinity = pos_affinity + 0.1 * torch.isin(L(x["next_album"]), L(x["album_context"]))
    pos_affinity = pos_affinity + 0.1 * torch.isin(L(x["next_artist"]), L(x["artist_context"]))
    neg_affinity = torch.amax(neg_embed @ context_embed.T, dim=-1)
    neg_affinity = neg_affinity + 0.1 * torch.isin(L(x["neg_album"]), L(x["album_context"]))
    neg_affinity = neg_affinity + 0.1 * torch.isin(L(x["neg_artist"]), L(x["artist_context"]))
    all_embeddings = torch.cat([context_embed, next_embed, neg_embed], dim=-2)
    all_embeddings_l2 = torch.sqrt(torch.sum(torch.square(all_embeddings), dim=-1))
    context_self_affinity = torch.flip(context_embed, dims=[-2]) @ context_embed.T
    next_self_affinity = torch.flip(next_embed, dims=[-2]) @ next_embed.T
    neg_self_affinity = torch.flip(neg_embed, dims=[-2]) @ neg_embed.T

    mean_triplet_loss = torch.relu(1.0 + torch.mean(neg_affinity) - torch.mean(pos_affinity))
    extremal_triplet_loss = torch.relu(1.0 + torch.amax(neg_affinity) - torch.amin(pos_affinity))
    context_self_affinity_loss = torch.mean(torch.relu(0.5 - context_self_affinity))
    next_self_affinity_loss = torch.mean(torch.relu(0.5 - next_self_affinity))
    neg_self_affinity_loss = torch.mean(torch.relu(neg_self_affinity))
    reg_loss = torch.sum(torch.relu(all_embeddings_l2 - regularization))
    loss = (extremal_triplet_loss + mean_triplet_loss + reg_loss + context_self_affinity_loss +
            next_self_affinity_loss + neg_self_affinity_loss)
    loss.backward()
    return loss.item(), at.grad.numpy(), rt.grad.numpy()
