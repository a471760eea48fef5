"""GloVe-style co-occurrence model -- drop-in for the reference's ``wikipedia/models.py:8-55``.

Same constructor fields (``num_embeddings=1024, features=64``), same parameter tree
(``params/_token_embedding/embedding (V, D)``, ``params/_bias/embedding (V, 1)``), same
``init / apply / score_all`` call shapes.  Tables are torch tensors resident in HBM; the arithmetic is
libesr_hip.so (no Flax, no XLA).
"""
import copy

import torch

from .. import ops


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("esrecsys_amd needs an MI355X (ROCm) device: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class Glove:
    """A simple embedding model based on GloVe (reference: wikipedia/models.py:8).

    ``loss_mode`` is a build-side knob read by ``apply_model``: "reference" reproduces the (B, B)
    broadcast of models.py:37 + train_cooccurence.py:83; "diagonal" is the textbook per-pair loss.
    """

    def __init__(self, num_embeddings=1024, features=64, loss_mode="reference", device=None):
        self.num_embeddings = int(num_embeddings)
        self.features = int(features)
        if loss_mode not in ("reference", "diagonal"):
            raise ValueError("loss_mode must be 'reference' or 'diagonal'")
        self.loss_mode = loss_mode
        self.device = device
        self._params = None

    # -- Flax-shaped plumbing -----------------------------------------------------------------
    def init(self, key, inputs=None):
        """``model.init(key, x)`` (wikipedia/train_cooccurence.py:168-170).  ``key`` is an int seed or a
        torch.Generator (JAX's threefry stream is not reproducible without JAX).  Token table ~ N(0, 1/D)
        (nn.Embed default init [upstream flax 0.5.2]); bias table zeros (models.py:18-19)."""
        dev = self.device or _default_device()
        if isinstance(key, torch.Generator):
            gen = key
        else:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(int(key))
        emb = torch.randn((self.num_embeddings, self.features), generator=gen, dtype=torch.float32)
        emb.mul_(self.features ** -0.5)
        return {"params": {
            "_token_embedding": {"embedding": emb.to(dev)},
            "_bias": {"embedding": torch.zeros((self.num_embeddings, 1), dtype=torch.float32, device=dev)},
        }}

    def apply(self, variables, *args, method=None, **kwargs):
        """``model.apply({'params': p}, inputs)`` / ``model.apply(..., token, method=Glove.score_all)``
        (wikipedia/train_cooccurence.py:78,92-95)."""
        bound = copy.copy(self)
        bound._params = variables["params"]
        fn = method if method is not None else Glove.__call__
        if isinstance(fn, str):
            fn = getattr(Glove, fn)
        return fn(bound, *args, **kwargs)

    def _tables(self):
        if self._params is None:
            raise RuntimeError("unbound module: call through model.apply({'params': ...}, ...)")
        return self._params["_token_embedding"]["embedding"], self._params["_bias"]["embedding"]

    # -- the model ----------------------------------------------------------------------------
    def __call__(self, inputs):
        """Approximate log count between tokens 1 and 2 (wikipedia/models.py:21-38).

        inputs: int [2, B].  Returns the reference's (B, B) output, ``out[i, j] = dot[j] + bias1[i] + bias2[i]``
        (models.py:37: (B,) + (B,1) + (B,1)), or the (B,) per-pair prediction in "diagonal" mode."""
        emb, bias = self._tables()
        inputs = ops.as_ids(inputs, emb.device, check_range=self.num_embeddings)
        dot, s = ops.glove_forward(emb, bias, inputs)
        if self.loss_mode == "diagonal":
            return dot + s
        return dot[None, :] + s[:, None]

    def score_all(self, token):
        """Score of token(s) vs all tokens: [V, T], no bias (wikipedia/models.py:40-55)."""
        emb, _ = self._tables()
        token = ops.as_ids(token, emb.device, check_range=self.num_embeddings).reshape(-1)
        return ops.score_all(emb, token)
