"""Spotify train-step / eval-step timing on one MI355X (reference shapes: 5 context tracks, ~20 next tracks,
64 negatives, feature_size 32, 100 000 hashed albums + 295 861 artists, 2 262 292 tracks in the corpus).
Prints one JSON line.  The CPU line is the oracle's dense fp32 restatement of the same step (a port, not JAX)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.spotify.models import SpotifyModel
    from esrecsys_amd.spotify.train_spotify import all_track_top_k, sample_negative, train_step
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    T = 2_262_292
    all_tracks = np.arange(T, dtype=np.int32)
    all_albums = rng.integers(0, 734_684, T).astype(np.int32)
    all_artists = rng.integers(0, 295_861, T).astype(np.int32)
    model = SpotifyModel(feature_size=32, device=dev)
    state = TrainState.create(apply_fn=model.apply, params=model.init(1701), tx=optim.sgd(1e-3, 0.98))
    batches = []
    for _ in range(64):
        m = int(rng.integers(5, 40))
        pick = rng.integers(0, T, 5 + m)
        x = {"track_context": all_tracks[pick[:5]], "album_context": all_albums[pick[:5]],
             "artist_context": all_artists[pick[:5]], "next_track": all_tracks[pick[5:]],
             "next_album": all_albums[pick[5:]], "next_artist": all_artists[pick[5:]]}
        sample_negative(x, rng, 64, all_tracks, all_albums, all_artists)
        batches.append(x)
    for x in batches[:8]:
        state, loss = train_step(state, x, 10.0)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for i in range(K):
        state, loss = train_step(state, batches[i % len(batches)], 10.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same steps with the playlists already on the device (the features of the next playlists uploaded ahead, as an
    # input pipeline would): what the GPU side of the step costs
    dbatches = [{k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in x.items()} for x in batches]
    for x in dbatches[:8]:
        state, loss = train_step(state, x, 10.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        state, loss = train_step(state, dbatches[i % len(dbatches)], 10.0)
    torch.cuda.synchronize()
    dt_dev = time.perf_counter() - t0
    d_alb, d_art = torch.from_numpy(all_albums).to(dev), torch.from_numpy(all_artists).to(dev)
    all_track_top_k(state, batches[0], d_alb, d_art)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(20):
        all_track_top_k(state, batches[i], d_alb, d_art)
    torch.cuda.synchronize()
    de = (time.perf_counter() - t1) / 20
    # CPU restatement of the same step: dense gradient + dense momentum over both tables, fp32
    from oracle import spotify as o_sp
    at = state.params["params"]["album_embed"]["embedding"].cpu().numpy()
    rt = state.params["params"]["artist_embed"]["embedding"].cpu().numpy()
    ta, tr = np.zeros_like(at), np.zeros_like(rt)
    t2 = time.perf_counter()
    nc = 5
    for i in range(nc):
        _, ga, gr = o_sp.dense_grads(at, rt, batches[i], 10.0, np.float32)
        at, ta = o_sp.sgd_momentum_update(at, ta, ga, 1e-3, 0.98, np.float32)
        rt, tr = o_sp.sgd_momentum_update(rt, tr, gr, 1e-3, 0.98, np.float32)
    dc = (time.perf_counter() - t2) / nc
    table_bytes = (at.size + rt.size) * 4
    print(json.dumps({"op": "spotify train_step (playlist = 5 context, 5-40 next, 64 negatives, F=32; sgd momentum)",
                      "steps_per_s": K / dt, "ms_per_step": dt / K * 1e3, "loss": float(loss),
                      "ms_per_step_device_resident_playlists": dt_dev / K * 1e3,
                      "optimizer": "optax.sgd(lr, momentum), lazy: rows decay when they are next read (no dense pass)",
                      "eval_all_tracks_top500_ms": de * 1e3, "eval_tracks_per_s": T / de,
                      "cpu_port_ms_per_step": dc * 1e3, "cpu_threads": torch.get_num_threads()}))


if __name__ == "__main__":
    main()
