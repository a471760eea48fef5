"""Generates tests/golden/spotify_*.npz (run from the repo root: ``python tests/golden/make_spotify_golden.py``).

The reference's Spotify path cannot run here (JAX / Flax / Optax are not installed) and holds no test vectors, so
the fixtures come from the fp64 oracle (oracle/spotify.py) after it agreed with the independent torch-autograd
transliteration (oracle/autograd_ref.py: spotify_value_and_grad).  Fixtures are data: seeded inputs + expected
outputs.  Tables are stored compactly: only the rows a fixture touches are non-zero in the full tables, so the
file holds those rows and their row numbers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import autograd_ref, spotify  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N_ARTISTS = 5000


def make_batch(rng, n, m, o, raw_albums, ties):
    x = {"album_context": rng.integers(0, raw_albums, n), "artist_context": rng.integers(0, N_ARTISTS, n),
         "track_context": rng.integers(0, 10 ** 6, n),
         "next_album": rng.integers(0, raw_albums, m), "next_artist": rng.integers(0, N_ARTISTS, m),
         "next_track": rng.integers(0, 10 ** 6, m),
         "neg_album": rng.integers(0, raw_albums, o), "neg_artist": rng.integers(0, N_ARTISTS, o),
         "neg_track": rng.integers(0, 10 ** 6, o)}
    if ties:
        x["album_context"][1], x["artist_context"][1] = x["album_context"][0], x["artist_context"][0]  # tied row max
        x["next_album"][0] = x["album_context"][2]                 # isin boost on a next track
        x["neg_artist"][min(3, o - 1)] = x["artist_context"][n - 1]  # isin boost on a negative
        if m > 5:
            x["next_album"][5] = x["next_album"][4] + spotify.MAX_ALBUMS  # same hashed row, different raw id
        if o > 9:
            x["neg_album"][9], x["neg_artist"][9] = x["neg_album"][8], x["neg_artist"][8]  # tied extremal neg
    return {k: v.astype(np.int32) for k, v in x.items()}


def case(name, n, m, o, F, reg, seed, ties=True, scale=0.5):
    rng = np.random.default_rng(seed)
    album_table = (rng.standard_normal((spotify.MAX_ALBUMS, F)) * scale).astype(np.float32)
    artist_table = (rng.standard_normal((N_ARTISTS, F)) * scale).astype(np.float32)
    x = make_batch(rng, n, m, o, 300_000, ties)
    a64, r64 = album_table.astype(np.float64), artist_table.astype(np.float64)
    loss, ga, gr = spotify.dense_grads(a64, r64, x, reg)
    l2, ga2, gr2 = autograd_ref.spotify_value_and_grad(a64, r64, x, reg)
    assert abs(loss - l2) <= 1e-8 * max(1, abs(loss)) and np.abs(ga - ga2).max() <= 1e-12 and np.abs(gr - gr2).max() <= 1e-12
    fwd = spotify.forward(a64, r64, x)
    _, aid, arows, rid, rrows = spotify.loss_and_row_grads(a64, r64, x, reg)
    used_a, used_r = np.unique(aid), np.unique(rid)
    out = dict(x)
    out.update(F=F, reg=reg, n_artists=N_ARTISTS, used_album_rows=used_a, album_rows_values=album_table[used_a],
               used_artist_rows=used_r, artist_rows_values=artist_table[used_r], loss=np.float64(loss),
               g_album_rows=arows, g_artist_rows=rrows, hashed_album=aid.astype(np.int32),
               g_album_dense_rows=ga[used_a], g_artist_dense_rows=gr[used_r],
               pos=fwd[0], neg=fwd[1], ctx_self=fwd[2], next_self=fwd[3], neg_self=fwd[4], l2=fwd[5])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


if __name__ == "__main__":
    case("spotify_n5_m17_o64_f32", 5, 17, 64, 32, 10.0, 1)           # the reference's shapes, reg inactive
    case("spotify_n5_m40_o64_f32_reg", 5, 40, 64, 32, 1.5, 2, scale=0.4)  # norm term active on some rows
    case("spotify_n3_m1_o8_f8", 3, 1, 8, 8, 0.3, 3, ties=False)      # a single next track, tiny F
    print("ok")
