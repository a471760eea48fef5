"""Oracle: brute-force retrieval (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows pinterest/make_recommendations.py:49-65 (find_top_k).  The stable
full-argsort kNN of the GloVe trainer is in oracle/glove.py (find_knn).
"""
import numpy as np


def find_top_k(scene_embedding, product_embeddings, k, dtype=np.float64):
    """find_top_k -- pinterest/make_recommendations.py:62-65.

    scores = sum(scene_embedding * product_embeddings, axis=-1); jax.lax.top_k(scores, k)
    returns (values, indices) sorted descending; among equal values the lower
    index comes first [upstream jax.lax.top_k].
    """
    s = np.asarray(scene_embedding).astype(dtype)
    p = np.asarray(product_embeddings).astype(dtype)
    scores = np.sum(s * p, axis=-1, dtype=dtype)
    return top_k(scores, k)


def top_k(scores, k):
    """jax.lax.top_k along the last axis (descending, ties -> lower index first)."""
    scores = np.asarray(scores)
    order = np.argsort(-scores, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(scores, order, axis=-1), order.astype(np.int32)


def batched_top_k(queries, candidates, k, dtype=np.float64):
    """Config-5 brute force: scores[b, n] = q_b . c_n ; top-k per query row."""
    s = np.asarray(queries).astype(dtype) @ np.asarray(candidates).astype(dtype).T
    return top_k(s, k)
