"""Brute-force retrieval -- drop-in for ``find_top_k`` of ``pinterest/make_recommendations.py:49-65``."""
import torch

from .. import ops


def find_top_k(scene_embedding, product_embeddings, k):
    """Top K nearest product embeddings to the scene embedding (make_recommendations.py:49-65).

    scores = sum(scene_embedding * product_embeddings, axis=-1); returns ``(scores[k], indices[k])``
    descending, ties to the lower index (jax.lax.top_k [upstream]).  A [Q, D] scene batch returns [Q, k]."""
    dev = product_embeddings.device if isinstance(product_embeddings, torch.Tensor) and product_embeddings.is_cuda \
        else torch.device("cuda", torch.cuda.current_device())
    q = ops.as_f32(scene_embedding, dev)
    p = ops.as_f32(product_embeddings, dev)
    single = q.dim() == 1 or q.shape[0] == 1
    q2 = q.reshape(-1, p.shape[1])
    s, i = ops.score_topk(q2, p, int(k))
    return (s[0], i[0]) if single else (s, i)
