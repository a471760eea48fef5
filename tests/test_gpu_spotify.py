"""GPU parity: the Spotify two-tower kernels (esr_spotify.hip) and the sgd-momentum optimizer, through the C ABI /
the drop-in API, against the oracle and the golden fixtures.  Bar: hashed rows bit-exact; <= 1e-5 relative on
fp32 loss, gradients, affinities; top-k indices equal wherever f32 scores do not tie."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import spotify as o_sp
from test_spotify_oracle import CASES, KEYS, batch_of, tables_of

pytestmark = pytest.mark.gpu
TOL = 1e-5
F64 = np.float64


def T(x, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return t.to(dtype) if dtype is not None else t


def N(t):
    return t.detach().cpu().numpy()


def occ(x, dev):
    al = np.concatenate([x["album_context"], x["next_album"], x["neg_album"]]).astype(np.int32)
    ar = np.concatenate([x["artist_context"], x["next_artist"], x["neg_artist"]]).astype(np.int32)
    return T(al, dev), T(ar, dev), len(x["album_context"]), len(x["next_album"]), len(x["neg_album"])


@pytest.mark.parametrize("case", CASES)
def test_spotify_fwd_bwd_vs_golden(dev, case):
    from esrecsys_amd import ops
    g = load_golden(case)
    at, rt = tables_of(g, np.float32)
    x = batch_of(g)
    al, ar, n, m, o = occ(x, dev)
    loss, rows, ga, gr = ops.spotify_fwd_bwd(T(at, dev), T(rt, dev), al, ar, n, m, o, float(g["reg"]))
    assert np.array_equal(N(rows), g["hashed_album"])
    assert abs(float(loss) - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    assert rel_err(N(ga), g["g_album_rows"]) <= TOL and rel_err(N(gr), g["g_artist_rows"]) <= TOL


@pytest.mark.parametrize("case", CASES)
def test_spotify_forward_vs_golden(dev, case):
    from esrecsys_amd import ops
    g = load_golden(case)
    at, rt = tables_of(g, np.float32)
    al, ar, n, m, o = occ(batch_of(g), dev)
    out = ops.spotify_forward(T(at, dev), T(rt, dev), al, ar, n, m, o)
    for got, key in zip(out, ("pos", "neg", "ctx_self", "next_self", "neg_self", "l2")):
        assert rel_err(N(got), g[key]) <= TOL, key


@pytest.mark.parametrize("n,m,o,F,reg", [(5, 250, 64, 32, 10.0), (5, 3, 64, 32, 0.8), (1, 2, 2, 4, 0.1),
                                         (32, 70, 100, 64, 2.0), (4, 33, 17, 128, 1.0), (5, 20, 64, 24, 1.0)])
def test_spotify_fwd_bwd_shapes_vs_oracle(dev, n, m, o, F, reg):
    """playlist-length extremes, non-default widths (F = 24: 2F not a multiple of 64), duplicated tracks"""
    from esrecsys_amd import ops
    rng = np.random.default_rng(n * 1000 + m)
    A, R = 100000, 3000
    at = (rng.standard_normal((A, F)) * (2.0 / np.sqrt(F))).astype(np.float32)
    rt = (rng.standard_normal((R, F)) * (2.0 / np.sqrt(F))).astype(np.float32)
    x = {"album_context": rng.integers(0, 400_000, n), "artist_context": rng.integers(0, R, n),
         "next_album": rng.integers(0, 400_000, m), "next_artist": rng.integers(0, R, m),
         "neg_album": rng.integers(0, 400_000, o), "neg_artist": rng.integers(0, R, o)}
    if m > 1:
        x["next_album"][1], x["next_artist"][1] = x["next_album"][0], x["next_artist"][0]   # duplicate next track
    if n > 1:
        x["album_context"][n - 1], x["artist_context"][n - 1] = x["album_context"][0], x["artist_context"][0]
    x = {k: v.astype(np.int32) for k, v in x.items()}
    al, ar, _, _, _ = occ(x, dev)
    loss, rows, ga, gr = ops.spotify_fwd_bwd(T(at, dev), T(rt, dev), al, ar, n, m, o, reg)
    el, aid, arows, rid, rrows = o_sp.loss_and_row_grads(at.astype(F64), rt.astype(F64), x, reg)
    assert np.array_equal(N(rows), aid)
    assert abs(float(loss) - el) <= TOL * abs(el)
    assert rel_err(N(ga), arows) <= TOL and rel_err(N(gr), rrows) <= TOL


def _small_world(dev, F=32, n_artists=4000, T_=50_000, seed=4):
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.spotify.models import SpotifyModel
    rng = np.random.default_rng(seed)
    model = SpotifyModel(feature_size=F, device=dev, num_artists=n_artists)
    params = model.init(1701)
    all_tracks = np.arange(T_, dtype=np.int32)
    all_albums = rng.integers(0, 700_000, T_).astype(np.int32)
    all_artists = rng.integers(0, n_artists, T_).astype(np.int32)
    return model, params, rng, all_tracks, all_albums, all_artists, TrainState, optim


def test_spotify_train_step_momentum_vs_oracle(dev):
    """three steps of the drop-in train_step with optim.sgd(lr, momentum) against the dense oracle update"""
    from esrecsys_amd.spotify.train_spotify import sample_negative, train_step
    model, params, rng, all_tracks, all_albums, all_artists, TrainState, optim = _small_world(dev)
    lr, mom, reg = 0.05, 0.9, 0.9
    state = TrainState.create(apply_fn=model.apply, params=params, tx=optim.sgd(lr, mom))
    at = N(params["params"]["album_embed"]["embedding"]).astype(F64)
    rt = N(params["params"]["artist_embed"]["embedding"]).astype(F64)
    ta, tr = np.zeros_like(at), np.zeros_like(rt)
    for step in range(3):
        pick = rng.integers(0, len(all_tracks), 5 + 12)
        x = {"track_context": all_tracks[pick[:5]], "album_context": all_albums[pick[:5]],
             "artist_context": all_artists[pick[:5]], "next_track": all_tracks[pick[5:]],
             "next_album": all_albums[pick[5:]], "next_artist": all_artists[pick[5:]]}
        sample_negative(x, rng, 64, all_tracks, all_albums, all_artists)
        state, loss = train_step(state, x, reg)
        el, ga, gr = o_sp.dense_grads(at, rt, x, reg)
        at, ta = o_sp.sgd_momentum_update(at, ta, ga, lr, mom, F64)
        rt, tr = o_sp.sgd_momentum_update(rt, tr, gr, lr, mom, F64)
        assert abs(float(loss) - el) <= TOL * abs(el), step
    assert state.step == 3
    assert rel_err(N(state.params["params"]["album_embed"]["embedding"]), at) <= TOL
    assert rel_err(N(state.params["params"]["artist_embed"]["embedding"]), rt) <= TOL
    assert rel_err(N(state.opt_state["trace"]["params"]["artist_embed"]["embedding"]), tr) <= TOL


def _playlist(rng, all_tracks, all_albums, all_artists, n_next=12):
    pick = rng.integers(0, len(all_tracks), 5 + n_next)
    return {"track_context": all_tracks[pick[:5]], "album_context": all_albums[pick[:5]],
            "artist_context": all_artists[pick[:5]], "next_track": all_tracks[pick[5:]],
            "next_album": all_albums[pick[5:]], "next_artist": all_artists[pick[5:]]}


def test_lazy_momentum_equals_flushing_every_step_and_the_dense_oracle(dev):
    """optim.sgd(lr, momentum) in its lazy form (rows decay when they are next read) over 40 playlists: bit-identical to
    the same run with every row brought up to date after every step (state.params: the dense form's arithmetic, one
    step at a time), within 1e-5 of the fp64 oracle's dense update, and the lazy run launches no dense pass."""
    from esrecsys_amd.spotify.train_spotify import sample_negative, train_step
    model, params, rng, all_tracks, all_albums, all_artists, TrainState, optim = _small_world(dev)
    lr, mom, reg, steps = 0.02, 0.9, 0.9, 40
    clone = lambda tree: {"params": {k: {"embedding": v["embedding"].clone()} for k, v in tree["params"].items()}}  # noqa: E731
    a = TrainState.create(apply_fn=model.apply, params=clone(params), tx=optim.sgd(lr, mom))
    b = TrainState.create(apply_fn=model.apply, params=clone(params), tx=optim.sgd(lr, mom))
    at = N(params["params"]["album_embed"]["embedding"]).astype(F64)
    rt = N(params["params"]["artist_embed"]["embedding"]).astype(F64)
    ta, tr = np.zeros_like(at), np.zeros_like(rt)
    for step in range(steps):
        x = _playlist(rng, all_tracks, all_albums, all_artists)
        sample_negative(x, rng, 64, all_tracks, all_albums, all_artists)
        a, la = train_step(a, x, reg)
        b, lb = train_step(b, x, reg)
        _ = b.params                               # flushes: every row of b is current after every step
        assert float(la) == float(lb), step
        el, ga, gr = o_sp.dense_grads(at, rt, x, reg)
        at, ta = o_sp.sgd_momentum_update(at, ta, ga, lr, mom, F64)
        rt, tr = o_sp.sgd_momentum_update(rt, tr, gr, lr, mom, F64)
        assert abs(float(la) - el) <= TOL * abs(el), step
    lz = a.opt_state["_lazy"]
    assert lz["step"] == steps and lz["dirty"]
    behind = int((lz["last"][("params", "artist_embed", "embedding")] < steps).sum())
    assert behind > 0, "most artist rows were not read by the last playlist: they are still behind"
    for k in ("album_embed", "artist_embed"):
        assert torch.equal(a.params["params"][k]["embedding"], b.params["params"][k]["embedding"])
        assert torch.equal(a.opt_state["trace"]["params"][k]["embedding"], b.opt_state["trace"]["params"][k]["embedding"])
    assert not a.opt_state["_lazy"]["dirty"]
    assert rel_err(N(a.params["params"]["album_embed"]["embedding"]), at) <= TOL
    assert rel_err(N(a.params["params"]["artist_embed"]["embedding"]), rt) <= TOL
    assert rel_err(N(a.opt_state["trace"]["params"]["artist_embed"]["embedding"]), tr) <= TOL


def test_lazy_momentum_long_gaps_take_the_closed_form(dev):
    """A row that nobody reads for more than 2048 steps is caught up by the closed form of the geometric decay
    (esr_optim.hip decay_steps): against the fp64 statement of n dense steps."""
    from esrecsys_amd import ops
    V, D, lr, mom = 1000, 32, 0.01, 0.98
    g = torch.Generator(device=dev).manual_seed(2)
    p0 = torch.randn((V, D), generator=g, device=dev)
    t0 = torch.randn((V, D), generator=g, device=dev) * 0.3
    for n in (1, 7, 2048, 2049, 5000, 200_000):
        p, t = p0.clone(), t0.clone()
        last = torch.zeros(V, dtype=torch.int32, device=dev)
        ids = torch.arange(0, V, 3, dtype=torch.int32, device=dev).repeat(2)          # every third row, each asked twice
        ops.momentum_catchup_rows(p, t, last, ids, 0, n + 1, lr, mom)                  # step n + 1 reads them: n missed steps
        geo = mom * (1.0 - mom ** n) / (1.0 - mom)
        ep = p0.double() - lr * t0.double() * geo
        et = t0.double() * mom ** n
        touched = torch.zeros(V, dtype=torch.bool, device=dev)
        touched[::3] = True
        assert rel_err(N(p[touched]), N(ep[touched])) <= 1e-5 and np.abs(N(t[touched]) - N(et[touched])).max() <= 1e-6
        assert torch.equal(p[~touched], p0[~touched]) and torch.equal(t[~touched], t0[~touched])
        assert torch.equal(last[touched], torch.full_like(last[touched], n + 1)) and int(last[~touched].sum()) == 0
        ops.momentum_flush(p, t, last, n + 1, lr, mom)                                  # everybody else: n + 1 steps
        geo1 = mom * (1.0 - mom ** (n + 1)) / (1.0 - mom)
        assert rel_err(N(p[~touched]), N((p0.double() - lr * t0.double() * geo1)[~touched])) <= 1e-5
        assert int((last != n + 1).sum()) == 0


def test_spotify_eval_step_vs_oracle(dev):
    from esrecsys_amd.spotify.train_spotify import all_track_top_k, eval_step
    model, params, rng, all_tracks, all_albums, all_artists, TrainState, optim = _small_world(dev, T_=200_003)
    state = TrainState.create(apply_fn=model.apply, params=params, tx=optim.sgd(1e-3, 0.98))
    pick = rng.integers(0, len(all_tracks), 9)
    y = {"album_context": all_albums[pick[:5]], "artist_context": all_artists[pick[:5]],
         "next_track": all_tracks[pick[5:]], "next_artist": all_artists[pick[5:]]}
    at = N(params["params"]["album_embed"]["embedding"]).astype(F64)
    rt = N(params["params"]["artist_embed"]["embedding"]).astype(F64)
    aff = o_sp.all_track_affinity(at, rt, y, all_albums, all_artists)
    em, eidx = o_sp.eval_step(at, rt, y, all_tracks, all_albums, all_artists, k=500)
    s, i = all_track_top_k(state, y, all_albums, all_artists)
    assert rel_err(N(s), aff[eidx]) <= TOL
    assert np.mean(N(i) == eidx) > 0.99 and rel_err(aff[N(i).astype(np.int64)], aff[eidx]) <= TOL
    m = N(eval_step(state, y, all_tracks, all_albums, all_artists))
    assert np.abs(m - em).max() <= 2.0 / 4       # recall moves by 1/4 per swapped boundary track at most
    assert m.shape == (2,)


def test_spotify_model_call_matches_reference_tuple(dev):
    from esrecsys_amd.spotify.models import SpotifyModel
    model, params, rng, all_tracks, all_albums, all_artists, _, _ = _small_world(dev)
    pick = rng.integers(0, len(all_tracks), 5 + 7 + 11)
    f = lambda a, sl: a[pick[sl]]  # noqa: E731
    c, nx, ng = slice(0, 5), slice(5, 12), slice(12, 23)
    out = model.apply(params, f(all_tracks, c), f(all_albums, c), f(all_artists, c), f(all_tracks, nx),
                      f(all_albums, nx), f(all_artists, nx), f(all_tracks, ng), f(all_albums, ng), f(all_artists, ng))
    x = {"album_context": f(all_albums, c), "artist_context": f(all_artists, c), "next_album": f(all_albums, nx),
         "next_artist": f(all_artists, nx), "neg_album": f(all_albums, ng), "neg_artist": f(all_artists, ng)}
    at = N(params["params"]["album_embed"]["embedding"]).astype(F64)
    rt = N(params["params"]["artist_embed"]["embedding"]).astype(F64)
    exp = o_sp.forward(at, rt, x)
    assert len(out) == 6 and [tuple(t.shape) for t in out] == [(7,), (11,), (5, 5), (7, 7), (11, 11), (23,)]
    for got, e in zip(out, exp):
        assert rel_err(N(got), e) <= TOL
    emb = model.apply(params, f(all_albums, c), f(all_artists, c), method=SpotifyModel.get_embeddings)
    assert np.array_equal(N(emb), o_sp.get_embeddings(at, rt, x["album_context"], x["artist_context"]).astype(np.float32))


@pytest.mark.parametrize("n_next,F", [(12, 32), (40, 32), (1, 16), (100, 64), (250, 32)])
def test_affinity_from_lds_equals_affinity_from_global_memory(dev, n_next, F, monkeypatch):
    """spotify_affinity_lds_kernel (rows staged in LDS, a thread per (scored row, context row) pair) against
    spotify_affinity_kernel (ESR_SPOTIFY_AFFINITY=global): every dot product adds its terms in the same order, so the
    per-occurrence gradient rows agree to the last bit or two (only the fp64 loss sums associate differently); and the
    row-gradient kernel's ballot walk over a context row's weights visits them in the old order."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(5 + n_next)
    A, n_art, n, o = 3000, 800, 5, 64
    g = torch.Generator(device=dev).manual_seed(3)
    at = torch.randn((A, F), generator=g, device=dev) * 0.2
    rt = torch.randn((n_art, F), generator=g, device=dev) * 0.2
    R = n + n_next + o
    for trial in range(4):
        albums = rng.integers(0, 50_000, R).astype(np.int32)
        artists = rng.integers(0, n_art if trial % 2 else 40, R).astype(np.int32)
        albums[n:n + 2] = albums[0]
        artists[n:n + 2] = artists[0]                                   # "in context" boosts; two equal context rows: ties
        albums[1], artists[1] = albums[0], artists[0]
        al, ar = T(albums, dev), T(artists, dev)
        monkeypatch.setenv("ESR_SPOTIFY_AFFINITY", "global")
        lb, rb, gab, grb = ops.spotify_fwd_bwd(at, rt, al, ar, n, n_next, o, 0.9)
        monkeypatch.setenv("ESR_SPOTIFY_AFFINITY", "lds")
        la, ra, gaa, gra = ops.spotify_fwd_bwd(at, rt, al, ar, n, n_next, o, 0.9)
        assert torch.equal(ra, rb)
        assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
        assert float((gaa - gab).abs().max()) <= 1e-6 * float(gab.abs().max())
        assert float((gra - grb).abs().max()) <= 1e-6 * float(grb.abs().max())


def test_inline_catch_up_equals_the_catch_up_launch(dev, monkeypatch):
    """The train step reads rows that are behind through spotify_gather_lazy_kernel and writes them caught-up in the
    momentum step itself (kMomentumStepLazy); ESR_SPOTIFY_CATCHUP=launch keeps round 3's catch-up launch in front.  Same
    operations on the same values: losses, tables, traces and step marks are bit-identical over 30 steps of a playlist
    stream in which rows sleep for a few steps, for more than 64 steps (closed-form gaps), and repeat inside a playlist."""
    from esrecsys_amd import ops
    rng = np.random.default_rng(77)
    A, n_art, F, n, m, o = 3000, 800, 32, 5, 20, 64
    R = n + m + o

    def fresh():
        gg = torch.Generator(device=dev).manual_seed(3)
        at = torch.randn((A, F), generator=gg, device=dev) * 0.2
        rt = torch.randn((n_art, F), generator=gg, device=dev) * 0.2
        return [at, torch.randn((A, F), generator=gg, device=dev) * 0.01, torch.zeros(A, dtype=torch.int32, device=dev),
                rt, torch.randn((n_art, F), generator=gg, device=dev) * 0.01, torch.zeros(n_art, dtype=torch.int32, device=dev)]
    a, b = fresh(), fresh()
    step = 0
    for it in range(30):
        step += 1 if it % 7 else 90          # now and then a long pause: gaps beyond kLazyExact take the closed form
        albums = rng.integers(0, 50_000, R).astype(np.int32)
        artists = rng.integers(0, n_art if it % 3 else 30, R).astype(np.int32)
        al, ar = T(albums, dev), T(artists, dev)
        monkeypatch.setenv("ESR_SPOTIFY_CATCHUP", "launch")
        lb = ops.spotify_train_step(b[0], b[1], b[2], b[3], b[4], b[5], al, ar, n, m, o, 0.9, step, 0.05, 0.9)
        monkeypatch.setenv("ESR_SPOTIFY_CATCHUP", "inline")
        la = ops.spotify_train_step(a[0], a[1], a[2], a[3], a[4], a[5], al, ar, n, m, o, 0.9, step, 0.05, 0.9)
        assert float(la) == float(lb), it
    for x, y in zip(a, b):
        assert torch.equal(x, y)
