"""Shop-The-Look trainer hot path -- drop-in for ``pinterest/train_shop_the_look.py:72-122``.

``train_step`` / ``eval_step`` / ``generate_triplets`` keep the reference's names, argument order and
return order.  ``train_step`` with ``neg_product=None`` switches to the north_star in-batch-negative
sampled-softmax loss (build-defined) on the FP32 MFMA path.
"""
import types

import numpy as np
import torch

from .. import ops
from ..train_state import FusedScatter, RowGrads, SegmentIndex

# Flags with the reference's names and defaults (pinterest/train_shop_the_look.py:46-69).
FLAGS = types.SimpleNamespace(
    input_file="STL-Dataset/fashion.json",
    image_dir="artifacts/shop_the_look:v1",
    num_neg=5,
    learning_rate=1e-3,
    regularization=0.1,
    output_size=32,
    batch_size=16,
    log_every_steps=100,
    eval_every_steps=2000,
    checkpoint_every_steps=100000,
    max_steps=30000,
    work_dir="/tmp",
    model_name="pinterest_stl_model",
    restore_checkpoint=False,
)


def generate_triplets(scene_product, num_neg, seed=0):
    """Generate positive and negative triplets (pinterest/train_shop_the_look.py:72-91).

    Same protocol: per positive pair ``num_neg`` negatives drawn from ``randint(0, count - 1)`` (upper bound
    exclusive, so the last item is never a negative -- reference quirk kept), every 10th positive goes to
    the test split.  JAX's threefry stream is not reproducible without JAX; NumPy PCG64 is used instead."""
    count = len(scene_product)
    rng = np.random.default_rng(seed)
    train, test = [], []
    for i in range(count):
        scene, pos = scene_product[i]
        is_test = i % 10 == 0
        for neg_idx in rng.integers(0, count - 1, size=num_neg):
            _, neg = scene_product[int(neg_idx)]
            (test if is_test else train).append((scene, pos, neg))
    return train, test


def _tables(state):
    p = state.params["params"] if "params" in state.params else state.params
    return p, p["scene_tower"]["embedding"], p["product_tower"]["embedding"]


def _wrap(state, inner):
    return {"params": inner} if "params" in state.params else inner


import os as _os
# Sorting the occurrence ids on a side stream beside the gather / MFMA kernels paid off while the sort was rocPRIM's
# ~6-launch radix chain (13.4 vs 12.0 M pairs/s).  With the two-launch tile sort (21 us) the side stream only steals
# bandwidth from the split kernel it overlaps: in line it is 0.406 ms per step, on the side stream 0.418.
_PRESORT = _os.environ.get("ESR_INBATCH_PRESORT", "0") == "1"


class PresortedTriplets:
    """The ids of one triplet batch on the device with their occurrence list [scene ; Vs + pos ; Vs + neg] already sorted
    on the side stream (``presort_triplets``): pass it as ``train_step(state, that, None, None, ...)``.  The sort needs
    the ids only, so a training loop runs it for batch k + 1 while batch k's kernels are in flight."""

    def __init__(self, scene, pos, neg, sorted_ids, perm, event):
        self.scene, self.pos, self.neg = scene, pos, neg
        self.sorted_ids, self.perm, self.event = sorted_ids, perm, event

    def take(self):
        if self.event is not None:
            torch.cuda.current_stream(self.scene.device).wait_event(self.event)
            self.event = None
        return self.sorted_ids, self.perm


def fused_triplet_step_available(state):
    """True when the triplet ``train_step`` can run in one pass (esr_triplet_train_step): the build's row-sparse Adagrad
    on fp32 towers of equal width.  ``ESR_STL_FUSED=0`` forces the fwd_bwd + sort + scatter path."""
    from ..train_state import _SparseAdagrad
    if _os.environ.get("ESR_STL_FUSED", "1") != "1" or not isinstance(state.tx, _SparseAdagrad):
        return False
    try:
        p = state.raw_params["params"] if "params" in state.raw_params else state.raw_params
        st, pt = p["scene_tower"]["embedding"], p["product_tower"]["embedding"]
    except (KeyError, TypeError):
        return False
    return (st.is_cuda and st.dtype == torch.float32 and pt.dtype == torch.float32 and st.shape[1] == pt.shape[1] and
            st.shape[0] + pt.shape[0] < (1 << 30))


def presort_triplets(state, scene, pos_product, neg_product):
    """Move the ids of a batch to the device and sort their occurrence list on the side stream.  neg_product = None:
    an in-batch batch, occurrence list [scene ; Vs + pos]."""
    from ..train_state import _side_stream
    p = state.raw_params["params"] if "params" in state.raw_params else state.raw_params
    st, pt = p["scene_tower"]["embedding"], p["product_tower"]["embedding"]
    dev = st.device
    sid = ops.as_ids(scene, dev, check_range=st.shape[0]).reshape(-1)
    pid = ops.as_ids(pos_product, dev, check_range=pt.shape[0]).reshape(-1)
    nid = ops.as_ids(neg_product, dev, check_range=pt.shape[0]).reshape(-1) if neg_product is not None else None
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        Vs = st.shape[0]
        segs, offs = ([sid, pid, nid], [0, Vs, Vs]) if nid is not None else ([sid, pid], [0, Vs])
        sorted_ids, perm = ops.segment_sort_multi(segs, offs, Vs + pt.shape[0])
        event = torch.cuda.Event()
        event.record(side)
    for t in segs:
        t.record_stream(side)
    sorted_ids.record_stream(main)
    perm.record_stream(main)
    return PresortedTriplets(sid, pid, nid, sorted_ids, perm, event)


def _fused_triplet_step(state, sid, pid, nid, regularization, batch_size, presorted):
    from ..train_state import row_versions
    prefix = ("params",) if "params" in state.raw_params else ()
    p = state.raw_params["params"] if prefix else state.raw_params
    st, pt = p["scene_tower"]["embedding"], p["product_tower"]["embedding"]
    acc = state.opt_state["sum_of_squares"]
    acc = acc["params"] if prefix else acc
    rs = row_versions(state, prefix + ("scene_tower", "embedding"))
    rp = row_versions(state, prefix + ("product_tower", "embedding"))
    rs.dirty = rp.dirty = True
    loss = ops.triplet_train_step(st, rs.shadow, rs.loc, acc["scene_tower"]["embedding"], pt, rp.shadow, rp.loc,
                                  acc["product_tower"]["embedding"], sid, pid, nid, regularization, batch_size,
                                  state.tx.lr, state.tx.eps, presorted=presorted)
    return state.replace(step=state.step + 1), loss.reshape(())


def train_step(state, scene, pos_product, neg_product, regularization, batch_size, scale=1.0, precision="auto"):
    """One optimizer step (pinterest/train_shop_the_look.py:93-109).  Returns ``(new_state, loss)``.

    loss = (sum relu(1 + neg - pos) + regularization * sum norm-excess) / batch_size.  One fused HIP launch
    gathers the three rows per triplet, scores them and writes the three gradient rows; the optimizer
    update is sort + segment-reduce + RMW.  ``neg_product=None``: in-batch softmax (north_star) with
    temperature ``scale``; ``precision`` picks its MFMA path (see ops.inbatch_softmax_fwd_bwd)."""
    presorted = None
    if isinstance(scene, PresortedTriplets):
        presorted, scene, pos_product, neg_product = scene, scene.scene, scene.pos, scene.neg
    if neg_product is not None and fused_triplet_step_available(state):
        # the reference's own loss with the build's sparse Adagrad: the whole step in one pass (esr_triplet_train_step)
        p = state.raw_params["params"] if "params" in state.raw_params else state.raw_params
        st, pt = p["scene_tower"]["embedding"], p["product_tower"]["embedding"]
        dev = st.device
        sid = ops.as_ids(scene, dev, check_range=st.shape[0]).reshape(-1)
        pid = ops.as_ids(pos_product, dev, check_range=pt.shape[0]).reshape(-1)
        nid = ops.as_ids(neg_product, dev, check_range=pt.shape[0]).reshape(-1)
        return _fused_triplet_step(state, sid, pid, nid, regularization, batch_size,
                                   presorted.take() if presorted is not None else None)
    _, st, pt = _tables(state)
    dev = st.device
    sid = ops.as_ids(scene, dev, check_range=st.shape[0]).reshape(-1)
    pid = ops.as_ids(pos_product, dev, check_range=pt.shape[0]).reshape(-1)
    B = sid.numel()
    sparse = not getattr(state.tx, "wants_dense", False)
    prefix = ("params",) if "params" in state.params else ()
    paths = [prefix + ("scene_tower", "embedding"), prefix + ("product_tower", "embedding")]
    Vs, Vp = st.shape[0], pt.shape[0]
    if neg_product is None:
        fused = FusedScatter([sid, pid], [0, 1], [Vs, Vp], None, paths) if sparse else None
        if fused is not None and presorted is not None:
            fused.index._sorted = presorted.take()  # sorted one batch ahead on the side stream (presort_triplets)
        elif fused is not None and _PRESORT:
            fused.index.presort()
        if precision != "f32" and st.shape[1] == 128 and B % 128 == 0 and st.dtype == pt.dtype:
            # the split-precision paths read the tower rows themselves (gather folded into their split and merge kernels)
            loss, _, gq, gc = ops.inbatch_towers_fwd_bwd(st, pt, sid, pid, scale, regularization, batch_size,
                                                         precision=precision)
        else:
            q = ops.gather_rows(st, sid)
            c = ops.gather_rows(pt, pid)
            if q.dtype != torch.float32:  # bf16 towers (fp32 accumulators): scores and gradients are computed in f32
                q, c = ops.unpermute_rows_to_f32(q, None), ops.unpermute_rows_to_f32(c, None)
            loss, _, gq, gc = ops.inbatch_softmax_fwd_bwd(q, c, scale, regularization, batch_size, precision=precision)
        if fused is not None:
            fused.rows = gq._base  # [gQ ; gC]
        g_scene = RowGrads([sid], gq, st.shape, fused)
        g_prod = RowGrads([pid], gc, pt.shape, fused)
    else:
        nid = ops.as_ids(neg_product, dev, check_range=pt.shape[0]).reshape(-1)
        # (no side-stream pre-sort here: the fused triplet kernel is ~12 us, there is nothing to hide the sort
        # behind and the cross-stream event round trip costs more than it saves: 0.19 vs 0.14 ms/step)
        loss, _, _, gs, gp, gn = ops.triplet_fwd_bwd(st, pt, pt, sid, pid, nid, B, regularization, batch_size,
                                                     with_reg=True, want_grads=True, want_scores=False)
        gall = gs._base  # [scene ; pos ; neg] gradient rows, one buffer
        # pos and neg both index the product table (slot 1)
        fused = FusedScatter([sid, pid, nid], [0, 1, 1], [Vs, Vp], gall, paths) if sparse else None
        g_scene = RowGrads([sid], gs, st.shape, fused)
        g_prod = RowGrads([pid, nid], gall[B:], pt.shape, fused)
    grads = _wrap(state, {"scene_tower": {"embedding": g_scene}, "product_tower": {"embedding": g_prod}})
    if getattr(state.tx, "wants_dense", False):
        from ..train_state import tree_map
        grads = tree_map(lambda g: g.to_dense(), grads)
    new_state = state.apply_gradients(grads=grads)
    return new_state, loss.reshape(())


def eval_step(state, scene, pos_product, neg_product):
    """sum relu(1 + neg - pos): fixed margin, no reg, not divided by the batch size (train_shop_the_look.py:111-122)."""
    _, st, pt = _tables(state)
    dev = st.device
    sid = ops.as_ids(scene, dev, check_range=st.shape[0]).reshape(-1)
    pid = ops.as_ids(pos_product, dev, check_range=pt.shape[0]).reshape(-1)
    nid = ops.as_ids(neg_product, dev, check_range=pt.shape[0]).reshape(-1)
    loss, _, _, _, _, _ = ops.triplet_fwd_bwd(st, pt, pt, sid, pid, nid, sid.numel(), 0.0, 1.0, with_reg=False,
                                              want_grads=False, want_scores=False)
    return loss.reshape(())
