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


def prepare_products(product_embeddings, mode="f16r"):
    """The per-corpus half of the brute force done once for a product table that serves many scene batches (the reference
    scores every scene against the same ``product_embeddings``: make_recommendations.py:123-132): pass the result as
    ``find_top_k_batch(..., prepared=...)`` for as long as the table is unchanged (ops.retrieve_prepare).  The table
    must already be a float32 device tensor (the prepared planes belong to THAT storage)."""
    dev = product_embeddings.device if isinstance(product_embeddings, torch.Tensor) and product_embeddings.is_cuda \
        else torch.device("cuda", torch.cuda.current_device())
    return ops.retrieve_prepare(ops.as_f32(product_embeddings, dev), mode=mode)


def find_top_k_batch(scene_embeddings, product_embeddings, k, approximate=False, probe=None, mode="exact", prepared=None):
    """``find_top_k`` for a batch of scenes in one call (the loop of make_recommendations.py:123-132; the
    score-everything-then-top_k eval of spotify/train_spotify.py:113-121): [Q, D] x [N, D] -> ([Q, k], [Q, k]).

    approximate=False: brute force with f32-equivalent scores on MFMA (three exact bf16 planes per operand).
    approximate=True : candidate stage in plain bf16 (one plane, 6x fewer MFMA flops) keeps ``probe`` >= k
    candidates per scene (default k + max(64, k/2), at most 1024), which are re-scored in f32 and re-ranked -- the reference has
    no ANN index; this is the build's approximate path and ``recall_at_k`` below measures it against brute force.
    mode: the brute force's arithmetic (ops.retrieve_topk: "exact" = three bf16 planes, "f16x2", "f16r" = the exact top-k of
    the f32 dot products from a one-plane filter + f32 re-score, 1.7x faster than "f16x2"); prepared = prepare_products(
    product_embeddings): the table's half of "f16r" made once."""
    dev = product_embeddings.device if isinstance(product_embeddings, torch.Tensor) and product_embeddings.is_cuda \
        else torch.device("cuda", torch.cuda.current_device())
    q = ops.as_f32(scene_embeddings, dev)
    p = ops.as_f32(product_embeddings, dev)
    q = q.reshape(-1, p.shape[1])
    k = int(k)
    if not approximate:
        if prepared is not None:
            return ops.retrieve_topk(q, p, k, mode=next(n for n, c in ops._RETRIEVE_MODES.items() if c == prepared.mode),
                                     prepared=prepared)
        return ops.retrieve_topk(q, p, k, mode=mode)
    probe = min(p.shape[0], 1024, max(k, int(probe) if probe is not None else k + max(64, k // 2)))
    _, cand = ops.retrieve_topk(q, p, probe, mode="bf16")
    exact = ops.rescore_candidates(q, p, cand)
    return ops.topk_merge(exact, cand, k)


def recall_at_k(approx_indices, exact_indices):
    """Mean fraction of the brute-force top-k that the approximate top-k also returned (esr_recall_at_k: a hash set per
    query in LDS, O(k) per query -- the comparison cube of rounds 1-4 was 2 GB at 8192 x 500 x 500)."""
    return ops.recall_at_k(approx_indices, exact_indices)
