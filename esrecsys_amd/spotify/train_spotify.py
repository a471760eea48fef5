"""Training / eval steps of the Spotify model -- drop-in for ``spotify/train_spotify.py:77-150``.

``train_step(state, x, regularization) -> (new_state, loss)`` and ``eval_step(state, y, all_tracks, all_albums,
all_artists) -> metrics[2]`` keep the reference's signatures; ``x`` / ``y`` are the reference's feature dicts.
The optimizer of the reference is ``optax.sgd(learning_rate, momentum)`` = ``esrecsys_amd.optim.sgd(lr, momentum)``.
"""
import numpy as np
import torch

from .. import ops
from ..train_state import RowGrads

FLAGS = dict(num_negatives=64, learning_rate=1e-3, momentum=0.98, regularization=10.0, feature_size=32,
             log_every_steps=1000, eval_every_steps=10000, eval_steps=1000, checkpoint_every_steps=100000,
             max_steps=2000000)  # train_spotify.py:60-70
TOP_K = 500  # train_spotify.py:120


def _model_of(state):
    fn = state.apply_fn
    return getattr(fn, "__self__", None)


def train_step(state, x, regularization):
    """train_spotify.py:77-111: value_and_grad of the six-term loss, then apply_gradients."""
    model = _model_of(state)
    raw = state.raw_params  # (the hot loop: state.params would bring EVERY row of the lazily updated tables up to date)
    p = raw["params"]
    at, rt = p["album_embed"]["embedding"], p["artist_embed"]["embedding"]
    bound = model.apply(raw, method=lambda m: m)  # bind the tables to read the occurrence ids
    al, ar, n, m, o = bound.occurrence_ids(x["album_context"], x["artist_context"], x["next_album"], x["next_artist"],
                                           x["neg_album"], x["neg_artist"])
    tx = state.tx
    if getattr(tx, "lazy", False) and hasattr(tx, "_lazy_state") and at.is_cuda and at.shape[1] == rt.shape[1]:
        # the reference's optimizer (optax.sgd(lr, momentum), train_spotify.py:238-241) in its lazy form: the whole step --
        # catch-up of the playlist's rows, loss, gradient rows, one sort, the momentum step on the touched rows of both
        # tables -- is ONE library call (eight launches; issued from Python the step was host-bound)
        lz = tx._lazy_state(raw, state.opt_state)
        lz["step"] += 1
        tr = state.opt_state["trace"]["params"]
        loss = ops.spotify_train_step(at, tr["album_embed"]["embedding"], lz["last"][("params", "album_embed", "embedding")],
                                      rt, tr["artist_embed"]["embedding"],
                                      lz["last"][("params", "artist_embed", "embedding")], al, ar, n, m, o, regularization,
                                      lz["step"], tx.lr, tx.momentum)
        lz["dirty"] = True
        return state.replace(step=state.step + 1), loss.reshape(())
    if hasattr(state.tx, "prepare"):
        # optax.sgd(lr, momentum) in its lazy form: the rows this playlist reads are brought up to date here (albums are
        # hashed into the table: row = album mod rows, spotify/models.py:37-41), nobody else's are touched
        state.tx.prepare(raw, state.opt_state, [(("params", "album_embed", "embedding"), al, at.shape[0]),
                                                (("params", "artist_embed", "embedding"), ar, 0)])
    loss, album_rows, ga, gr = ops.spotify_fwd_bwd(at, rt, al, ar, n, m, o, regularization)
    grads = {"params": {"album_embed": {"embedding": RowGrads([album_rows], ga, at.shape)},
                        "artist_embed": {"embedding": RowGrads([ar], gr, rt.shape)}}}
    return state.apply_gradients(grads=grads), loss.reshape(())


def all_track_top_k(state, y, all_albums, all_artists, k=TOP_K, segments=64):
    """jax.lax.top_k(all_affinity, 500) over the whole corpus (train_spotify.py:119-120): (scores[k], indices[k]).
    Two levels through the batched select kernel: top-k of `segments` slices, then of their union."""
    p = state.params["params"]
    at, rt = p["album_embed"]["embedding"], p["artist_embed"]["embedding"]
    dev = at.device
    ca, cr = ops.as_ids(y["album_context"], dev).reshape(-1), ops.as_ids(y["artist_context"], dev).reshape(-1)
    aa, rr = ops.as_ids(all_albums, dev).reshape(-1), ops.as_ids(all_artists, dev).reshape(-1)
    aff = ops.spotify_affinity_all(at, rt, ca, cr, aa, rr)
    T = aff.numel()
    k = min(k, T)
    seg = max(k, -(-T // segments))
    rows = -(-T // seg)
    pad = rows * seg - T
    idx = torch.arange(rows * seg, dtype=torch.int32, device=dev)
    if pad:
        aff = torch.cat([aff, torch.full((pad,), float("-inf"), device=dev)])
    s1, i1 = ops.topk_merge(aff.reshape(rows, seg), idx.reshape(rows, seg), k)
    s, i = ops.topk_merge(s1.reshape(1, -1), i1.reshape(1, -1), k)
    return s[0], i[0]


def eval_step(state, y, all_tracks, all_albums, all_artists):
    """train_spotify.py:113-131: recall of the next tracks / artists among the 500 best-scoring tracks."""
    dev = state.params["params"]["album_embed"]["embedding"].device
    _, top = all_track_top_k(state, y, all_albums, all_artists)
    tracks = ops.as_ids(all_tracks, dev).reshape(-1)[top.long()]
    artists = ops.as_ids(all_artists, dev).reshape(-1)[top.long()]
    nt, na = ops.as_ids(y["next_track"], dev).reshape(-1), ops.as_ids(y["next_artist"], dev).reshape(-1)
    tracks_recall = torch.isin(tracks, nt).sum().float() / nt.numel()
    artists_recall = torch.isin(artists, na).sum().float() / na.numel()
    return torch.stack([tracks_recall, artists_recall])


def sample_negative(x, rng, num_negatives, all_tracks, all_albums, all_artists):
    """train_spotify.py:139-150.  `rng` is a numpy Generator (JAX's threefry stream is not reproducible without
    JAX); like the reference, the upper bound is exclusive of the last track."""
    idx = rng.integers(0, len(all_tracks) - 1, num_negatives)
    x["neg_track"] = np.asarray(all_tracks)[idx]
    x["neg_album"] = np.asarray(all_albums)[idx]
    x["neg_artist"] = np.asarray(all_artists)[idx]
    return rng
