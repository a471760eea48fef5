"""Oracle: Spotify id-embedding two-tower (TEST INFRASTRUCTURE -- see oracle/__init__.py).  PARITY UNPINNED:
the reference holds no tests or golden vectors for this path; the closed forms below are pinned against an
independent torch-autograd transliteration (oracle/autograd_ref.py: spotify_value_and_grad) instead.

Follows spotify/models.py:27-90 (SpotifyModel) and spotify/train_spotify.py:77-131 (train_step, eval_step).
JAX semantics restated [upstream jax 0.4.10]: the VJP of max / min splits the cotangent EVENLY over tied
arguments (two context tracks of one album + artist have identical embeddings, so ties are real);
relu'(0) = 0; isin compares the raw (un-hashed) ids; optax.sgd(lr, momentum) is trace = g + momentum * trace,
p -= lr * trace on EVERY element [upstream optax]."""
import numpy as np

MAX_ALBUMS = 100000   # spotify/models.py:33
BOOST = 0.1           # spotify/models.py:76-81


def get_embeddings(album_table, artist_table, album, artist, dtype=np.float64):
    """spotify/models.py:37-51: concat(album_embed[album mod 100000], artist_embed[artist])."""
    a = np.asarray(album_table, dtype)[np.mod(np.asarray(album, np.int64), MAX_ALBUMS)]
    r = np.asarray(artist_table, dtype)[np.asarray(artist, np.int64)]
    return np.concatenate([a, r], axis=-1)


def _affinity(emb, ctx, album, artist, album_ctx, artist_ctx):
    raw = emb @ ctx.T
    aff = raw.max(axis=-1)
    aff = aff + BOOST * np.isin(album, album_ctx) + BOOST * np.isin(artist, artist_ctx)
    return raw, aff


def forward(album_table, artist_table, x, dtype=np.float64):
    """SpotifyModel.__call__ (spotify/models.py:53-90) -> the 6-tuple it returns."""
    C = get_embeddings(album_table, artist_table, x["album_context"], x["artist_context"], dtype)
    X = get_embeddings(album_table, artist_table, x["next_album"], x["next_artist"], dtype)
    Y = get_embeddings(album_table, artist_table, x["neg_album"], x["neg_artist"], dtype)
    _, pos = _affinity(X, C, x["next_album"], x["next_artist"], x["album_context"], x["artist_context"])
    _, neg = _affinity(Y, C, x["neg_album"], x["neg_artist"], x["album_context"], x["artist_context"])
    allemb = np.concatenate([C, X, Y], axis=-2)
    l2 = np.sqrt(np.sum(np.square(allemb), axis=-1))
    return pos, neg, C[::-1] @ C.T, X[::-1] @ X.T, Y[::-1] @ Y.T, l2


def loss_and_row_grads(album_table, artist_table, x, regularization, dtype=np.float64):
    """train_step's loss_fn (spotify/train_spotify.py:78-107) and its gradient as per-occurrence rows.

    Returns (loss, album_ids[R], album_rows[R, F], artist_ids[R], artist_rows[R, F]) with R = n + m + o
    occurrences in the order context, next, neg; album_ids are already hashed (mod 100000)."""
    F = np.asarray(album_table).shape[1]
    C = get_embeddings(album_table, artist_table, x["album_context"], x["artist_context"], dtype)
    X = get_embeddings(album_table, artist_table, x["next_album"], x["next_artist"], dtype)
    Y = get_embeddings(album_table, artist_table, x["neg_album"], x["neg_artist"], dtype)
    n, m, o = len(C), len(X), len(Y)
    P, pos = _affinity(X, C, x["next_album"], x["next_artist"], x["album_context"], x["artist_context"])
    Q, neg = _affinity(Y, C, x["neg_album"], x["neg_artist"], x["album_context"], x["artist_context"])
    mt_arg = 1.0 + neg.mean() - pos.mean()
    et_arg = 1.0 + neg.max() - pos.min()
    sa_c, sa_x, sa_y = C[::-1] @ C.T, X[::-1] @ X.T, Y[::-1] @ Y.T
    allemb = np.concatenate([C, X, Y], axis=0)
    l2 = np.sqrt(np.sum(np.square(allemb), axis=-1))
    relu = lambda v: np.maximum(v, 0)  # noqa: E731
    loss = (relu(et_arg) + relu(mt_arg) + relu(l2 - regularization).sum() + relu(0.5 - sa_c).mean() +
            relu(0.5 - sa_x).mean() + relu(sa_y).mean())

    # ---- d loss / d pos, d neg
    dpos = np.zeros(m, dtype)
    dneg = np.zeros(o, dtype)
    if mt_arg > 0:
        dpos -= 1.0 / m
        dneg += 1.0 / o
    if et_arg > 0:
        tie = pos == pos.min()
        dpos -= tie / tie.sum()
        tie = neg == neg.max()
        dneg += tie / tie.sum()

    def through_rowmax(raw, dvec):
        tie = raw == raw.max(axis=-1, keepdims=True)
        return dvec[:, None] * tie / tie.sum(axis=-1, keepdims=True)

    dP, dQ = through_rowmax(P, dpos), through_rowmax(Q, dneg)
    dC = dP.T @ X + dQ.T @ Y
    dX = dP @ C
    dY = dQ @ C

    def self_affinity_grad(Z, sa, kind):
        R = len(Z)
        dA = (-(sa < 0.5).astype(dtype) if kind == "pull" else (sa > 0).astype(dtype)) / (R * R)
        # A[a, b] = Z[R-1-a] . Z[b]
        return dA.T @ Z[::-1] + (dA @ Z)[::-1]

    dC = dC + self_affinity_grad(C, sa_c, "pull")
    dX = dX + self_affinity_grad(X, sa_x, "pull")
    dY = dY + self_affinity_grad(Y, sa_y, "push")
    dall = np.concatenate([dC, dX, dY], axis=0)
    active = l2 > regularization
    dall = dall + np.where(active[:, None], allemb / np.where(l2 > 0, l2, 1.0)[:, None], 0.0)
    album = np.concatenate([x["album_context"], x["next_album"], x["neg_album"]]).astype(np.int64)
    artist = np.concatenate([x["artist_context"], x["next_artist"], x["neg_artist"]]).astype(np.int64)
    return loss, np.mod(album, MAX_ALBUMS), dall[:, :F], artist, dall[:, F:]


def dense_grads(album_table, artist_table, x, regularization, dtype=np.float64):
    """(loss, d album_table, d artist_table) -- what jax.value_and_grad returns."""
    loss, aid, arows, rid, rrows = loss_and_row_grads(album_table, artist_table, x, regularization, dtype)
    ga = np.zeros(np.asarray(album_table).shape, dtype)
    gr = np.zeros(np.asarray(artist_table).shape, dtype)
    np.add.at(ga, aid, arows)
    np.add.at(gr, rid, rrows)
    return loss, ga, gr


def sgd_momentum_update(param, trace, grad, lr, momentum, dtype=None):
    """optax.sgd(lr, momentum) [upstream]: trace = grad + momentum * trace ; param -= lr * trace (dense)."""
    dtype = dtype or param.dtype.type
    t = grad.astype(dtype) + dtype(momentum) * trace.astype(dtype)
    return param.astype(dtype) - dtype(lr) * t, t


def all_track_affinity(album_table, artist_table, y, all_albums, all_artists, dtype=np.float64):
    """eval_step's result[1] (spotify/train_spotify.py:113-119): the affinity of EVERY track to the context."""
    C = get_embeddings(album_table, artist_table, y["album_context"], y["artist_context"], dtype)
    E = get_embeddings(album_table, artist_table, all_albums, all_artists, dtype)
    return _affinity(E, C, all_albums, all_artists, y["album_context"], y["artist_context"])[1]


def eval_step(album_table, artist_table, y, all_tracks, all_albums, all_artists, k=500, dtype=np.float64):
    """spotify/train_spotify.py:113-131 -> (metrics[2], top_k_indices)."""
    from . import topk
    aff = all_track_affinity(album_table, artist_table, y, all_albums, all_artists, dtype)
    _, idx = topk.top_k(aff, k)
    top_tracks, top_artists = np.asarray(all_tracks)[idx], np.asarray(all_artists)[idx]
    tracks_recall = np.isin(top_tracks, y["next_track"]).sum() / len(y["next_track"])
    artists_recall = np.isin(top_artists, y["next_artist"]).sum() / len(y["next_artist"])
    return np.array([tracks_recall, artists_recall], np.float32), idx
