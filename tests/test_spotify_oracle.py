"""CPU: the Spotify oracle (oracle/spotify.py) against the golden fixtures, the independent torch-autograd
transliteration (oracle/autograd_ref.py) and torch.optim.SGD for the momentum rule."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import autograd_ref, spotify

CASES = ["spotify_n5_m17_o64_f32", "spotify_n5_m40_o64_f32_reg", "spotify_n3_m1_o8_f8"]
KEYS = ("album_context", "artist_context", "track_context", "next_album", "next_artist", "next_track", "neg_album",
        "neg_artist", "neg_track")


def tables_of(g, dtype=np.float64):
    F = int(g["F"])
    at = np.zeros((spotify.MAX_ALBUMS, F), dtype)
    rt = np.zeros((int(g["n_artists"]), F), dtype)
    at[g["used_album_rows"]] = g["album_rows_values"]
    rt[g["used_artist_rows"]] = g["artist_rows_values"]
    return at, rt


def batch_of(g):
    return {k: g[k] for k in KEYS}


@pytest.mark.parametrize("case", CASES)
def test_spotify_oracle_matches_golden_and_autograd(case):
    g = load_golden(case)
    at, rt = tables_of(g)
    x = batch_of(g)
    loss, aid, arows, rid, rrows = spotify.loss_and_row_grads(at, rt, x, float(g["reg"]))
    assert abs(loss - float(g["loss"])) <= 1e-12 * max(1.0, abs(loss))
    assert np.array_equal(aid, g["hashed_album"])
    assert np.abs(arows - g["g_album_rows"]).max() <= 1e-12 and np.abs(rrows - g["g_artist_rows"]).max() <= 1e-12
    l2, ga, gr = autograd_ref.spotify_value_and_grad(at, rt, x, float(g["reg"]))
    assert abs(loss - l2) <= 1e-8 * max(1.0, abs(loss))     # the transliteration's 0.1 * bool is a float32 0.1
    _, dga, dgr = spotify.dense_grads(at, rt, x, float(g["reg"]))
    assert np.abs(dga - ga).max() <= 1e-12 and np.abs(dgr - gr).max() <= 1e-12
    fwd = spotify.forward(at, rt, x)
    for got, key in zip(fwd, ("pos", "neg", "ctx_self", "next_self", "neg_self", "l2")):
        assert np.abs(got - g[key]).max() <= 1e-12


def test_spotify_tie_rule_splits_evenly():
    """two identical context rows: the row max ties and each of them receives half of the cotangent"""
    g = load_golden(CASES[0])
    at, rt = tables_of(g)
    x = batch_of(g)
    assert x["album_context"][0] == x["album_context"][1] and x["artist_context"][0] == x["artist_context"][1]
    _, _, arows, _, _ = spotify.loss_and_row_grads(at, rt, x, float(g["reg"]))
    assert np.array_equal(arows[0], arows[1])


def test_sgd_momentum_matches_torch_sgd():
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal((50, 8))
    grads = [rng.standard_normal((50, 8)) * (rng.random((50, 1)) < 0.3) for _ in range(4)]   # row-sparse
    p, t = p0.copy(), np.zeros_like(p0)
    for gr in grads:
        p, t = spotify.sgd_momentum_update(p, t, gr, 1e-2, 0.98, np.float64)
    tp = torch.tensor(p0, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.SGD([tp], lr=1e-2, momentum=0.98)
    for gr in grads:
        tp.grad = torch.tensor(gr)
        opt.step()
    assert np.abs(tp.detach().numpy() - p).max() <= 1e-14


def test_eval_step_oracle_counts_hits():
    rng = np.random.default_rng(2)
    F, T = 8, 3000
    at = rng.standard_normal((spotify.MAX_ALBUMS, F)) * 0.3
    rt = rng.standard_normal((400, F)) * 0.3
    all_tracks = np.arange(T, dtype=np.int32)
    all_albums = rng.integers(0, 250_000, T).astype(np.int32)
    all_artists = rng.integers(0, 400, T).astype(np.int32)
    y = {"album_context": all_albums[:5], "artist_context": all_artists[:5], "next_track": all_tracks[5:9],
         "next_artist": all_artists[5:9]}
    m, idx = spotify.eval_step(at, rt, y, all_tracks, all_albums, all_artists, k=500)
    aff = spotify.all_track_affinity(at, rt, y, all_albums, all_artists)
    assert np.all(np.diff(aff[idx]) <= 0) and len(idx) == 500
    assert m[0] == np.isin(idx, y["next_track"]).sum() / 4
