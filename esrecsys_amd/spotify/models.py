"""Spotify million-playlist model -- drop-in for ``spotify/models.py:23-90``.

Two id-embedding tables (albums hashed ``mod 100000``, 295 861 artists); a track embeds as the concatenation of
its album and artist rows.  ``__call__`` returns the reference's 6-tuple.  Compute is libesr_hip.so
(esr_spotify_forward); there is no torch fallback."""
import copy

import numpy as np
import torch

from .. import ops
from ..wikipedia.models import _default_device


class SpotifyModel:
    """Spotify model that takes a context and predicts the next tracks (models.py:23)."""

    MAX_ALBUMS = 100000    # models.py:33
    NUM_ARTISTS = 295861   # models.py:35

    def __init__(self, feature_size, device=None, max_albums=None, num_artists=None):
        self.feature_size = int(feature_size)
        self.max_albums = int(max_albums or self.MAX_ALBUMS)
        self.num_artists = int(num_artists or self.NUM_ARTISTS)
        self.device = device
        self._params = None

    def init(self, key, *example_inputs):
        """``spotify.init(subkey, *x)`` (train_spotify.py:225-229).  nn.Embed default init: N(0, 1/F) [upstream
        flax variance_scaling(1.0, 'fan_in', 'normal', out_axis=0)]."""
        dev = self.device or _default_device()
        if isinstance(key, torch.Generator):
            gen = key
        else:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(int(key))
        F = self.feature_size

        def table(rows):
            return (torch.randn((rows, F), generator=gen, dtype=torch.float32) * F ** -0.5).to(dev)

        return {"params": {"album_embed": {"embedding": table(self.max_albums)},
                           "artist_embed": {"embedding": table(self.num_artists)}}}

    def apply(self, variables, *args, method=None, **kwargs):
        bound = copy.copy(self)
        bound._params = variables["params"]
        fn = method if method is not None else SpotifyModel.__call__
        return fn(bound, *args, **kwargs)

    def _tables(self):
        if self._params is None:
            raise RuntimeError("unbound module: call through model.apply({'params': ...}, ...)")
        return self._params["album_embed"]["embedding"], self._params["artist_embed"]["embedding"]

    def get_embeddings(self, album, artist):
        """models.py:37-51: [n, 2F] = concat(album_embed[album mod max_albums], artist_embed[artist])."""
        at, rt = self._tables()
        al = ops.as_ids(album, at.device).reshape(-1)
        ar = ops.as_ids(artist, rt.device, check_range=rt.shape[0]).reshape(-1)
        return ops.spotify_get_embeddings(at, rt, al, ar)

    def occurrence_ids(self, album_context, artist_context, next_album, next_artist, neg_album, neg_artist):
        """(album_ids, artist_ids, n, m, o): the int32 occurrence lists (context, next, neg) the kernels take."""
        at, rt = self._tables()
        groups_a, groups_r = (album_context, next_album, neg_album), (artist_context, next_artist, neg_artist)
        if not any(isinstance(v, torch.Tensor) for v in groups_a + groups_r):
            # host features (the reference's numpy iterator): pack both lists into ONE transfer
            a = np.concatenate([np.asarray(v).reshape(-1) for v in groups_a]).astype(np.int32)
            r = np.concatenate([np.asarray(v).reshape(-1) for v in groups_r]).astype(np.int32)
            if r.size and (r.min() < 0 or r.max() >= rt.shape[0]):
                raise IndexError("artist id out of range [0, %d)" % rt.shape[0])
            both = torch.from_numpy(np.stack([a, r])).to(at.device, non_blocking=True)
            n, m, o = (int(np.asarray(v).size) for v in groups_a)
            return both[0], both[1], n, m, o
        parts_a = [ops.as_ids(v, at.device).reshape(-1) for v in groups_a]
        parts_r = [ops.as_ids(v, rt.device, check_range=rt.shape[0]).reshape(-1) for v in groups_r]
        n, m, o = (p.numel() for p in parts_a)
        return torch.cat(parts_a), torch.cat(parts_r), n, m, o

    def __call__(self, track_context, album_context, artist_context, next_track, next_album, next_artist,
                 neg_track, neg_album, neg_artist):
        """models.py:53-90 -> (pos_affinity [m], neg_affinity [o], context_self_affinity [n, n],
        next_self_affinity [m, m], neg_self_affinity [o, o], all_embeddings_l2 [n + m + o])."""
        at, rt = self._tables()
        al, ar, n, m, o = self.occurrence_ids(album_context, artist_context, next_album, next_artist, neg_album,
                                              neg_artist)
        return ops.spotify_forward(at, rt, al, ar, n, m, o)
