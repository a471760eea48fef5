"""Shop-The-Look two-tower model -- drop-in for the score head of ``pinterest/models.py:48-74``.

The reference's towers are CNNs over 512x512 JPEGs (models.py:23-46): out of scope.  As north_star
specifies, each tower here is an id-embedding table (scene ids / product ids -> output_size floats)
and everything downstream of the towers -- pos/neg dot-product scores, the 5-tuple return value,
``get_scene_embed`` / ``get_product_embed`` -- follows the reference.
"""
import copy

import torch

from .. import ops
from ..wikipedia.models import _default_device


class STLModel:
    """Shop the look model: takes a scene and items and computes a score for them (models.py:48)."""

    def __init__(self, output_size, num_scenes=None, num_products=None, device=None):
        self.output_size = int(output_size)
        self.num_scenes = num_scenes
        self.num_products = num_products
        self.device = device
        self._params = None

    def init(self, key, scene=None, pos_product=None, neg_product=None):
        """``stl.init(subkey, x[0], x[1], x[2])`` (pinterest/train_shop_the_look.py:174).  Tables ~ N(0, 1/D)."""
        if self.num_scenes is None or self.num_products is None:
            raise ValueError("STLModel needs num_scenes and num_products (id-embedding towers)")
        dev = self.device or _default_device()
        if isinstance(key, torch.Generator):
            gen = key
        else:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(int(key))
        D = self.output_size

        def table(n):
            return (torch.randn((n, D), generator=gen, dtype=torch.float32) * D ** -0.5).to(dev)

        return {"params": {"scene_tower": {"embedding": table(self.num_scenes)},
                           "product_tower": {"embedding": table(self.num_products)}}}

    def apply(self, variables, *args, method=None, mutable=None, **kwargs):
        """``stl.apply(params, scene, pos, neg, True, mutable=['batch_stats'])`` returns ``(result, state)``
        when ``mutable`` is given, as Flax does (pinterest/train_shop_the_look.py:95-98)."""
        bound = copy.copy(self)
        bound._params = variables["params"]
        fn = method if method is not None else STLModel.__call__
        out = fn(bound, *args, **kwargs)
        return (out, {}) if mutable is not None else out

    def _tables(self):
        if self._params is None:
            raise RuntimeError("unbound module: call through model.apply({'params': ...}, ...)")
        return self._params["scene_tower"]["embedding"], self._params["product_tower"]["embedding"]

    def get_scene_embed(self, scene):
        """models.py:57-58."""
        st, _ = self._tables()
        return ops.gather_rows(st, ops.as_ids(scene, st.device, check_range=st.shape[0]).reshape(-1))

    def get_product_embed(self, product):
        """models.py:60-61."""
        _, pt = self._tables()
        return ops.gather_rows(pt, ops.as_ids(product, pt.device, check_range=pt.shape[0]).reshape(-1))

    def __call__(self, scene, pos_product, neg_product, train=True):
        """models.py:63-74: returns (pos_score, neg_score, scene_embed, pos_product_embed, neg_product_embed)."""
        st, pt = self._tables()
        sid = ops.as_ids(scene, st.device, check_range=st.shape[0]).reshape(-1)
        pid = ops.as_ids(pos_product, pt.device, check_range=pt.shape[0]).reshape(-1)
        nid = ops.as_ids(neg_product, pt.device, check_range=pt.shape[0]).reshape(-1)
        _, pos_score, neg_score, _, _, _ = ops.triplet_fwd_bwd(st, pt, pt, sid, pid, nid, sid.numel(), 0.0, 1.0,
                                                               with_reg=False, want_grads=False)
        return (pos_score, neg_score, ops.gather_rows(st, sid), ops.gather_rows(pt, pid), ops.gather_rows(pt, nid))


def score_head(scene_embed, pos_product_embed, neg_product_embed):
    """The reference-pinned head on three given (B, D) matrices (models.py:67-72): (pos_score, neg_score)."""
    B = scene_embed.shape[0]
    _, ps, ns, _, _, _ = ops.triplet_fwd_bwd(scene_embed, pos_product_embed, neg_product_embed, None, None, None, B,
                                             0.0, 1.0, with_reg=False, want_grads=False)
    return ps, ns
