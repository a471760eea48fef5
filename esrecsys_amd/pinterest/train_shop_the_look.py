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


# ESR_INBATCH_ONECALL=0: the in-batch step as fwd_bwd + apply_gradients (rounds 1-4).  ESR_INBATCH_OVERLAP=1: the one call
# with merge<Q> and the scene tower's update on a second stream beside pass C.  Bit-identical, and measured SLOWER at C2
# (same box, alternating runs: 0.2424-0.2447 ms per step against 0.2347-0.2351 in line): pass C loses 9 us to the traffic
# beside it (74 -> 83 us), the factor launch pass C then needs is 7 us and the row copies the merges must read add 4 us
# to the gather + split -- more than the 15 + 7 us that left the critical path.  Off by default; kept because the
# one-call step is what a multi-GPU exchange would hide behind pass C.
_INBATCH_ONECALL = _os.environ.get("ESR_INBATCH_ONECALL", "1") == "1"
_INBATCH_OVERLAP = _os.environ.get("ESR_INBATCH_OVERLAP", "0") == "1"


def _inbatch_one_call(state, st, pt, B, precision):
    """True when an in-batch step can be the one library call: the build's row-sparse Adagrad, towers of one dtype and
    width on the fp16 x 2 score path."""
    from ..train_state import _SparseAdagrad
    if not _INBATCH_ONECALL or not isinstance(state.tx, _SparseAdagrad) or not st.is_cuda or st.dtype != pt.dtype or \
            st.shape[1] != pt.shape[1] or precision == "f32" or st.shape[0] + pt.shape[0] >= (1 << 31):
        return False
    try:
        return ops.inbatch_split_path(precision, B, st.shape[1], bf16_tables=st.dtype == torch.bfloat16) == "f16x2"
    except ValueError:
        return False


class PresortedTriplets:
    """The ids of one triplet batch on the device with their occurrence list [scene ; Vs + pos ; Vs + neg] already sorted
    on the side stream (``presort_triplets``): pass it as ``train_step(state, that, None, None, ...)``.  The sort needs
    the ids only, so a training loop runs it for batch k + 1 while batch k's kernels are in flight."""

    def __init__(self, scene, pos, neg, sorted_ids, perm, event, hint=None):
        self.scene, self.pos, self.neg = scene, pos, neg
        self.sorted_ids, self.perm, self.event = sorted_ids, perm, event
        self.hint = hint  # (pinned int32 [1], gen, event recorded behind the hint kernel) or None

    def long_runs(self):
        """0 when the long-run hint has reached the host and says no id has a run longer than a 32-position chunk, 1
        when it says one has, -1 while nobody knows."""
        if self.hint is None or not self.hint[2].query():
            return -1
        from ..wikipedia.train_cooccurence import hint_value
        return hint_value(self.hint)

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
    # (bf16 towers -- BASELINE config 4's dtype, fp32 accumulators -- are stepped by the direct kernels only)
    ok_dtype = (st.dtype == torch.float32 and pt.dtype == torch.float32) or (
        st.dtype == torch.bfloat16 and pt.dtype == torch.bfloat16 and ops.triplet_direct_mode())
    if not (st.is_cuda and ok_dtype and st.shape[1] == pt.shape[1] and st.shape[0] + pt.shape[0] < (1 << 30)):
        return False
    if ops.triplet_direct_mode():  # rows are stepped in place: nothing to double-buffer
        return True
    # the stamped step keeps both towers double-buffered: without room for the second buffers the six-launch path runs
    from ..train_state import can_double_buffer
    prefix = ("params",) if "params" in state.raw_params else ()
    return can_double_buffer(state, [prefix + ("scene_tower", "embedding"), prefix + ("product_tower", "embedding")])


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
    from ..train_state import next_stamp, row_versions
    prefix = ("params",) if "params" in state.raw_params else ()
    p = state.raw_params["params"] if prefix else state.raw_params
    st, pt = p["scene_tower"]["embedding"], p["product_tower"]["embedding"]
    acc = state.opt_state["sum_of_squares"]
    acc = acc["params"] if prefix else acc
    if ops.triplet_direct_mode():  # in place: no second buffers, no stamps
        if state.versions:  # (rows an earlier stamped step left in second buffers go home first)
            state.consolidate()
        loss = ops.triplet_train_step(st, None, None, acc["scene_tower"]["embedding"], pt, None, None,
                                      acc["product_tower"]["embedding"], sid, pid, nid, regularization, batch_size,
                                      state.tx.lr, state.tx.eps, presorted=presorted)
        return state.replace(step=state.step + 1), loss.reshape(())
    rs = row_versions(state, prefix + ("scene_tower", "embedding"))
    rp = row_versions(state, prefix + ("product_tower", "embedding"))
    loss = ops.triplet_train_step(st, rs.shadow, rs.loc, acc["scene_tower"]["embedding"], pt, rp.shadow, rp.loc,
                                  acc["product_tower"]["embedding"], sid, pid, nid, regularization, batch_size,
                                  state.tx.lr, state.tx.eps, presorted=presorted, stamp=next_stamp(rs, rp))
    return state.replace(step=state.step + 1), loss.reshape(())


def train_step(state, scene, pos_product, neg_product, regularization, batch_size, scale=1.0, precision="auto"):
    """One optimizer step (pinterest/train_shop_the_look.py:93-109).  Returns ``(new_state, loss)``.

    loss = (sum relu(1 + neg - pos) + regularization * sum norm-excess) / batch_size.  One fused HIP launch
    gathers the three rows per triplet, scores them and writes the three gradient rows; the optimizer
    update is sort + segment-reduce + RMW.  ``neg_product=None``: in-batch softmax (north_star) with
    temperature ``scale``; ``precision`` picks its MFMA path (see ops.inbatch_softmax_fwd_bwd)."""
    if isinstance(scene, PlannedTriplets):  # a batch of ``presorted(state, batches)``: sorted and planned with its group
        return scene.step(state, regularization, batch_size)
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
    if neg_product is None and _inbatch_one_call(state, st, pt, B, precision):
        # the whole step -- gather + split, pass Q, [merge<Q> + scene-tower Adagrad] beside [pass C, merge<C>, product-tower
        # Adagrad] -- as ONE library call (esr_inbatch_train_step_f16x2; round 5)
        from ..train_state import _side_stream
        acc = state.opt_state["sum_of_squares"]
        acc = acc["params"] if prefix else acc
        srt, long_runs = None, -1
        if presorted is not None and presorted.sorted_ids is not None:
            srt, long_runs = presorted.take(), presorted.long_runs()
        loss = ops.inbatch_train_step(st, acc["scene_tower"]["embedding"], pt, acc["product_tower"]["embedding"], sid, pid,
                                      scale, regularization, batch_size, state.tx.lr, state.tx.eps, presorted=srt,
                                      long_runs=long_runs, side_stream=_side_stream(dev) if _INBATCH_OVERLAP else None)
        return state.replace(step=state.step + 1), loss.reshape(())
    if neg_product is None:
        fused = FusedScatter([sid, pid], [0, 1], [Vs, Vp], None, paths) if sparse else None
        if fused is not None and presorted is not None:
            fused.index._sorted = presorted.take()  # sorted one batch ahead on the side stream (presort_triplets)
            fused.index.long_runs = presorted.long_runs()
        elif fused is not None and _PRESORT:
            fused.index.presort()
        if precision != "f32" and st.dtype == pt.dtype and \
                ops.inbatch_split_path(precision, B, st.shape[1], bf16_tables=st.dtype == torch.bfloat16) is not None:
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


class _Group:
    """A group of batches whose id lists were sorted and planned together (``_FusedTripletLoop.sort_batch``)."""
    __slots__ = ("nb", "B", "ptrs", "sorted_ptr", "perm_ptr", "plans_ptr", "which", "gen", "keep", "side")

    def __init__(self, nb, B, ptrs, sorted_ptr, perm_ptr, plans_ptr, which, gen, keep, side=False):
        self.nb, self.B, self.ptrs, self.sorted_ptr, self.perm_ptr = nb, B, ptrs, sorted_ptr, perm_ptr
        self.plans_ptr, self.which, self.gen, self.keep = plans_ptr, which, gen, keep  # keep: the id tensors, alive
        self.side = side  # sorted + planned on the second stream: the group's first step waits for its event


class _FusedTripletLoop:
    """What the one-pass triplet step needs that does not change from batch to batch, resolved once: tower / accumulator
    / RowVersions pointers, the library entry points, a loss slot per step, and a ring of id-sort buffers on the side
    stream.  At the reference's batch sizes the step is three short kernels behind a two-launch id sort; the sort needs
    the ids only, so it runs `depth` batches ahead on a second stream (same idea as wikipedia.train_epoch), and the
    per-step Python is two library calls and three event operations."""

    def __init__(self, state, steps, depth):
        import ctypes
        from .. import _lib
        from ..train_state import _side_stream, next_stamp, row_versions
        self.next_stamp = next_stamp
        prefix = ("params",) if "params" in state.raw_params else ()
        p = state.raw_params["params"] if prefix else state.raw_params
        acc = state.opt_state["sum_of_squares"]
        acc = acc["params"] if prefix else acc
        self.st, self.pt = p["scene_tower"]["embedding"], p["product_tower"]["embedding"]
        self.direct = ops.triplet_direct_mode()  # rows stepped in place: no second buffers, location bytes or stamps
        if self.direct and state.versions:  # (rows an earlier stamped step left in second buffers go home first)
            state.consolidate()
        self.rs = None if self.direct else row_versions(state, prefix + ("scene_tower", "embedding"))
        self.rp = None if self.direct else row_versions(state, prefix + ("product_tower", "embedding"))
        self.acc_s, self.acc_p = acc["scene_tower"]["embedding"], acc["product_tower"]["embedding"]
        self.Vs, self.D = self.st.shape
        self.Vp = self.pt.shape[0]
        self.dt = ops._table_dtype(self.st, "scene tower")
        self.dev = self.st.device
        self.lr, self.eps = float(state.tx.lr), float(state.tx.eps)
        self.lib, self.check, self.ct = _lib.load(), _lib.check, ctypes
        self.losses = torch.empty(max(steps, 1), dtype=torch.float32, device=self.dev)
        self.losses_ptr = self.losses.data_ptr()
        self.main = torch.cuda.current_stream(self.dev)
        self.main_raw = self.main.cuda_stream
        self.side = _side_stream(self.dev)
        self.depth = depth
        self.ring, self.B = [], -1
        self.ws = None
        self.batch_buf, self.sort_batch_max = [None, None], _SORT_BATCH  # two sets: group g + 1 is made before g steps
        self.batch_ptrs = [(ctypes.c_void_p * (3 * _SORT_BATCH))(), (ctypes.c_void_p * (3 * _SORT_BATCH))()]
        self.long_arr = (ctypes.c_int32 * _SORT_BATCH)()
        # esr_triplet_plan's hints ("list b has a run longer than a chunk": only then is the long-run launch needed),
        # written by the plan kernel straight into pinned host memory; a group is planned one group ahead of its steps,
        # so the words have landed when they are issued (if they have not, the launch is simply made)
        self.hints_host = torch.zeros((2, _SORT_BATCH), dtype=torch.int32).pin_memory()
        self.hints_event = [torch.cuda.Event(), torch.cuda.Event()]
        self.drawn_event = [torch.cuda.Event(), torch.cuda.Event()]
        self.hints_known = [None, None]  # per set: list of long_runs values once the event has been seen complete
        self.group_ws, self.group_ws_B = None, -1  # esr_triplet_train_steps' workspace, sized for the group it steps
        self.gen = 0
        self.allow_side = True  # (presorted()'s per-step calls keep sort + plan on the main stream: see _plan_on_side)
        if self.direct:
            self.fixed_s = (self.st.data_ptr(), None, None, self.acc_s.data_ptr(), self.Vs)
            self.fixed_p = (self.pt.data_ptr(), None, None, self.acc_p.data_ptr(), self.Vp)
        else:
            self.fixed_s = (self.st.data_ptr(), self.rs.shadow.data_ptr(), self.rs.loc.data_ptr(), self.acc_s.data_ptr(),
                            self.Vs)
            self.fixed_p = (self.pt.data_ptr(), self.rp.shadow.data_ptr(), self.rp.loc.data_ptr(), self.acc_p.data_ptr(),
                            self.Vp)

    def check_state(self, state):
        """Raise unless `state` holds the towers / accumulators / optimizer settings this context stepped so far and the
        current stream is the one its launches are ordered on."""
        prefix = ("params",) if "params" in state.raw_params else ()
        p = state.raw_params["params"] if prefix else state.raw_params
        acc = state.opt_state["sum_of_squares"]
        acc = acc["params"] if prefix else acc
        same = (p["scene_tower"]["embedding"].data_ptr() == self.st.data_ptr() and
                p["product_tower"]["embedding"].data_ptr() == self.pt.data_ptr() and
                acc["scene_tower"]["embedding"].data_ptr() == self.acc_s.data_ptr() and
                acc["product_tower"]["embedding"].data_ptr() == self.acc_p.data_ptr())
        if not same:
            raise RuntimeError("this planned batch belongs to another state: the towers / accumulators of the state passed "
                               "to train_step are not the ones presorted() was given (after a restore or a swap, make a "
                               "new presorted() iterator)")
        if float(state.tx.lr) != self.lr or float(state.tx.eps) != self.eps:
            raise RuntimeError("the optimizer's learning rate / eps changed since presorted() was called: make a new "
                               "presorted() iterator (the loop context keeps them)")
        if torch.cuda.current_stream(self.dev).cuda_stream != self.main_raw:
            raise RuntimeError("planned batches are ordered on the stream presorted() was called under: step them under "
                               "that stream")

    def stamp(self, count=1):
        """The (first) stamp of the next `count` steps on both towers; direct mode has none."""
        return 1 if self.direct else self.next_stamp(self.rs, self.rp, count=count)

    def _sized(self, B):
        if B == self.B:
            return
        n = 3 * B
        self.ring = []
        for _ in range(self.depth + 1):
            self.ring.append({
                "sorted": torch.empty(n, dtype=torch.int32, device=self.dev),
                "perm": torch.empty(n, dtype=torch.int32, device=self.dev),
                "ws": torch.empty(ops._ws_bytes("esr_segment_sort_workspace_bytes", n), dtype=torch.uint8,
                                  device=self.dev),
                "done": torch.cuda.Event(), "ids_ready": torch.cuda.Event(), "free": torch.cuda.Event()})
        self.ws = ops._ws(ops._ws_bytes("esr_triplet_step_workspace_bytes", B, self.D), self.dev)
        self.ws_ptr, self.ws_n = self.ws.data_ptr(), self.ws.numel()
        self.cnt = (self.ct.c_int64 * 3)(B, B, B)
        self.off = (self.ct.c_int64 * 3)(0, self.Vs, self.Vs)
        self.B = B

    def ids(self, scene, pos, neg):
        sid = ops.as_ids(scene, self.dev, check_range=self.Vs)
        pid = ops.as_ids(pos, self.dev, check_range=self.Vp)
        nid = ops.as_ids(neg, self.dev, check_range=self.Vp)
        if sid.dim() != 1:
            sid = sid.reshape(-1)
        if pid.dim() != 1:
            pid = pid.reshape(-1)
        if nid.dim() != 1:
            nid = nid.reshape(-1)
        return sid, pid, nid

    def presort(self, slot_index, sid, pid, nid):
        """Sort [scene ; Vs + pos ; Vs + neg] of a coming batch on the side stream into ring slot `slot_index`."""
        self._sized(sid.numel())
        slot = self.ring[slot_index]
        slot["ids_ready"].record(self.main)          # the ids may have been copied / produced on the main stream
        self.side.wait_event(slot["ids_ready"])
        self.side.wait_event(slot["free"])           # the step that last read this slot's buffers has been issued
        ptrs = (self.ct.c_void_p * 3)(sid.data_ptr(), pid.data_ptr(), nid.data_ptr())
        self.check(self.lib.esr_segment_sort_ids_multi(ptrs, self.cnt, self.off, 3, self.Vs + self.Vp,
                                                       slot["sorted"].data_ptr(), slot["perm"].data_ptr(),
                                                       slot["ws"].data_ptr(), slot["ws"].numel(),
                                                       self.side.cuda_stream), "esr_segment_sort_ids_multi")
        slot["done"].record(self.side)
        return (slot_index, sid, pid, nid)

    def sort_batch(self, group, which=0):
        """The id lists of the next len(group) batches (each (sid, pid, nid)) sorted by ONE library call and planned by
        one more (esr_segment_sort_ids_batched, esr_triplet_plan) on the main stream, into buffer set `which`: handles
        for ``step``.  At the reference's batch sizes the two-launch sort is 16 us of latency whether it sorts one list
        or eight, and nothing a plan holds depends on the tables."""
        B = group[0][0].numel()
        self._sized(B)
        nb, n = len(group), 3 * B
        pb = ops._ws_bytes("esr_triplet_plan_bytes", B)
        if self.batch_buf[which] is None or self.batch_buf[which][0].shape != (self.sort_batch_max, n):
            self.batch_buf[which] = (torch.empty((self.sort_batch_max, n), dtype=torch.int32, device=self.dev),
                                     torch.empty((self.sort_batch_max, n), dtype=torch.int32, device=self.dev),
                                     ops._ws(ops._ws_bytes("esr_segment_sort_batched_workspace_bytes", n,
                                                           self.sort_batch_max), self.dev),
                                     ops._aligned_bytes(self.sort_batch_max * pb, self.dev))
        srt, prm, ws, plans = self.batch_buf[which]
        # (the library call itself, not ops.segment_sort_batched: its per-tensor checks and ctypes arrays were 67 us per
        # group -- the ids were validated by ``ids``, the arrays are kept)
        ptrs = self.batch_ptrs[which]
        i = 0
        for g in group:
            ptrs[i], ptrs[i + 1], ptrs[i + 2] = g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr()
            i += 3
        stream, raw = self.main, self.main_raw
        on_side = _plan_on_side(B) and self.allow_side
        if on_side:
            # sort + plan of group g + 1 beside the steps of group g: the side stream waits for what the main stream held
            # when the group was drawn (the steps of group g - 1, whose buffers this set reuses; id copies), the main
            # stream waits for the plan before the group's first step (step_group)
            stream, raw = self.side, self.side.cuda_stream
            self.drawn_event[which].record(self.main)
            self.side.wait_event(self.drawn_event[which])
        self.check(self.lib.esr_segment_sort_ids_batched(ptrs, self.cnt, self.off, 3, nb, self.Vs + self.Vp,
                                                         srt.data_ptr(), prm.data_ptr(), ws.data_ptr(), ws.numel(),
                                                         raw), "esr_segment_sort_ids_batched")
        self.gen += 1
        self.check(self.lib.esr_triplet_plan(ptrs, nb, B, self.Vs, srt.data_ptr(), prm.data_ptr(), plans.data_ptr(),
                                             self.hints_host[which].data_ptr(), self.gen, raw),
                   "esr_triplet_plan")
        self.hints_event[which].record(stream)
        self.hints_known[which] = None
        # the whole group is stepped by ONE library call (esr_triplet_train_steps): at the reference's own batch sizes
        # (16 - 128 triplets: train_shop_the_look.py:60) a step is ~12 us of kernels, at 8192 it is 21 us -- less than
        # a 29-argument foreign call plus the Python around it
        return _Group(nb, B, ptrs, srt.data_ptr(), prm.data_ptr(), plans.data_ptr(), which, self.gen, group, on_side)

    def step_group(self, k, gr, regularization, batch_size):
        """Steps k .. k + gr.nb - 1: the batches of a sorted and planned group, issued by one library call."""
        known = self.hints_known[gr.which]
        done = known is not None
        if known is None:
            # the HOST waits for the group's plan launch: it was queued in front of the steps of the group before, so
            # the wait ends with those steps (a group's worth of work) still queued -- the device never runs dry, the
            # host stays at most two groups ahead, and every step knows whether it needs its long-run launch (a loop
            # that ran further ahead never saw a hint in time and made all of them: 4.8 us per step at B = 8192)
            if _HINT_WAIT:
                self.hints_event[gr.which].synchronize()
            if self.hints_event[gr.which].query():
                done = True
                known = self.hints_known[gr.which] = self.hints_host[gr.which].tolist()
        if gr.side and not done:
            # (the plan was made on the second stream: the steps wait for it ON THE DEVICE only when the host has not seen
            # its event complete -- a stream wait is a barrier packet in front of the group's first step, ~5 us of idle
            # main stream per group for an event that has long fired)
            self.main.wait_event(self.hints_event[gr.which])
        long_runs = None
        if known is not None:  # (else: the hint has not reached the host; the library makes every long-run launch)
            long_runs = self.long_arr
            for j in range(gr.nb):
                long_runs[j] = 1 if known[j] == gr.gen else 0
        # (the group's OWN batch size: the next group -- sorted before this one is stepped -- may have resized self.ws)
        if gr.B != self.group_ws_B:
            self.group_ws = ops._ws(ops._ws_bytes("esr_triplet_step_workspace_bytes", gr.B, self.D), self.dev)
            self.group_ws_B = gr.B
        self.check(self.lib.esr_triplet_train_steps(*self.fixed_s, *self.fixed_p, self.dt, self.D, gr.nb, gr.ptrs, gr.B,
                                                    regularization, batch_size, self.lr, self.eps,
                                                    self.stamp(gr.nb), gr.sorted_ptr,
                                                    gr.perm_ptr, gr.plans_ptr, long_runs, self.losses_ptr + 4 * k,
                                                    self.group_ws.data_ptr(), self.group_ws.numel(), self.main_raw),
                   "esr_triplet_train_steps")

    def step(self, k, handle, regularization, batch_size):
        slot_index, sid, pid, nid = handle
        sorted_ptr = perm_ptr = 0  # in-line sort inside the library call
        slot = None
        if slot_index is not None:
            slot = self.ring[slot_index]
            self.main.wait_event(slot["done"])
            sorted_ptr, perm_ptr = slot["sorted"].data_ptr(), slot["perm"].data_ptr()
        else:
            self._sized(sid.numel())
        self.check(self.lib.esr_triplet_train_step(*self.fixed_s, *self.fixed_p, self.dt, self.D, sid.data_ptr(), pid.data_ptr(),
                                                   nid.data_ptr(), self.B, float(regularization), float(batch_size),
                                                   self.lr, self.eps, self.stamp(), sorted_ptr,
                                                   perm_ptr, 0, -1, self.losses.data_ptr() + 4 * k,
                                                   self.ws.data_ptr(), self.ws.numel(), self.main.cuda_stream),
                   "esr_triplet_train_step")
        if slot is not None:
            slot["free"].record(self.main)


class PlannedTriplets:
    """One batch of a group that ``presorted`` drew from the iterator, sorted and planned together (esr_segment_sort_ids_batched
    + esr_triplet_plan, one call pair per group of up to eight batches).  ``train_step(state, handle, None, None,
    regularization, batch_size)`` -- or with the ids it carries -- steps it with ONE library call (esr_triplet_train_step
    on the group's sorted ids and plan record)."""
    __slots__ = ("ctx", "group", "j", "scene", "pos", "neg", "used")

    def __init__(self, ctx, group, j):
        self.ctx, self.group, self.j = ctx, group, j
        self.scene, self.pos, self.neg = group.keep[j]
        self.used = False

    def __iter__(self):  # ``for scene, pos, neg in presorted(...)`` -- the reference's loop shape: scene IS the handle
        return iter((self, self.pos, self.neg))

    def step(self, state, regularization, batch_size):
        if self.used:
            raise RuntimeError("a planned batch feeds exactly one train_step (its plan holds the step's accumulators)")
        ctx, gr, j = self.ctx, self.group, self.j
        # the loop context captured the towers, accumulators, learning rate and stream of the state ``presorted`` was given:
        # a state swapped mid-iteration (checkpoint restore, another learning rate, other tables) must not step the OLD
        # tables while the returned state claims step + 1
        ctx.check_state(state)
        self.used = True
        if ctx.group_of[gr.which] is not gr:
            raise RuntimeError("this planned batch is two groups old: its sorted ids and plan have been overwritten "
                               "(step the batches of presorted() in the order it yields them)")
        k = ctx.next_loss_slot()
        n = 3 * gr.B
        if gr.side and j == 0:  # (the group's sort + plan ran on the second stream: its first step waits for them)
            ctx.main.wait_event(ctx.hints_event[gr.which])
        known = ctx.hints_known[gr.which]
        if known is None:
            # the group's plan launch was queued a whole group ahead of this step (see presorted): the wait ends with that
            # group's steps still in the queue, and every step then knows whether it needs its long-run launch
            if _HINT_WAIT:
                ctx.hints_event[gr.which].synchronize()
            if ctx.hints_event[gr.which].query():
                known = ctx.hints_known[gr.which] = ctx.hints_host[gr.which].tolist()
        long_runs = -1 if known is None else (1 if known[j] == gr.gen else 0)
        if gr.B != ctx.group_ws_B:
            ctx.group_ws = ops._ws(ops._ws_bytes("esr_triplet_step_workspace_bytes", gr.B, ctx.D), ctx.dev)
            ctx.group_ws_B = gr.B
        pb = ops._ws_bytes("esr_triplet_plan_bytes", gr.B)
        ctx.check(ctx.lib.esr_triplet_train_step(*ctx.fixed_s, *ctx.fixed_p, ctx.dt, ctx.D, self.scene.data_ptr(),
                                                 self.pos.data_ptr(), self.neg.data_ptr(), gr.B, float(regularization),
                                                 float(batch_size), ctx.lr, ctx.eps, ctx.stamp(),
                                                 gr.sorted_ptr + 4 * n * j, gr.perm_ptr + 4 * n * j,
                                                 gr.plans_ptr + pb * j, long_runs, ctx.losses_ptr + 4 * k,
                                                 ctx.group_ws.data_ptr(), ctx.group_ws.numel(), ctx.main_raw),
                  "esr_triplet_train_step")
        return state.replace(step=state.step + 1), ctx.losses[k]


def presorted(state, batches):
    """Iterator adapter for the reference's OWN loop shape (pinterest/train_shop_the_look.py:190-221):

        for scene, pos, neg in presorted(state, train_it):
            state, loss = train_step(state, scene, pos, neg, regularization, batch_size)

    at the rate of ``train_steps``: the batches are drawn from `batches` up to eight at a time, their id lists sorted and
    planned by one call pair per group -- the NEXT group's before this group's first step is issued, as in train_steps --
    and what the loop receives as ``scene`` is a PlannedTriplets handle that train_step steps with one library call.
    Needs the one-pass triplet step (sparse Adagrad, fp32 towers with room for their second buffers); otherwise, and for
    in-batch batches (neg = None), the batches pass through unchanged.  Each loss is a view into a ring of 4096 slots:
    read (or copy) it before 4096 further steps."""
    it = iter(batches)
    if not fused_triplet_step_available(state):
        yield from it
        return
    ctx = _FusedTripletLoop(state, 4096, 0)
    ctx.allow_side = _PLAN_STREAM == "side"
    ctx.group_of = [None, None]
    ctx.loss_k = -1

    def next_loss_slot():
        ctx.loss_k = (ctx.loss_k + 1) % 4096
        return ctx.loss_k
    ctx.next_loss_slot = next_loss_slot
    which = 0

    def draw():
        """('group', _Group) of 2 .. 8 equal-sized triplet batches, or ('plain', [batches]) to pass through, or None."""
        nonlocal which
        first = next(it, None)
        if first is None:
            return None
        if first[2] is None:
            return ("plain", [first])
        group = [ctx.ids(*first)]
        plain_tail = []
        while len(group) < _SORT_BATCH:
            b = next(it, None)
            if b is None:
                break
            if b[2] is None:
                plain_tail.append(b)
                break
            group.append(ctx.ids(*b))
        B = group[0][0].numel()
        if len(group) == 1 or any(g[0].numel() != B for g in group) or 3 * B > _SORT_BATCH_MAX_IDS:
            return ("plain", [g for g in group] + plain_tail)  # (own sorts inside their steps)
        which ^= 1
        gr = ctx.sort_batch(group, which)
        ctx.group_of[which] = gr
        return ("group", gr, plain_tail)

    cur = draw()
    while cur is not None:
        nxt = draw()  # sorted and planned BEFORE this group's steps are issued: its hints reach the host a group ahead
        if cur[0] == "group":
            gr = cur[1]
            for j in range(gr.nb):
                yield PlannedTriplets(ctx, gr, j)
            yield from cur[2]
        else:
            yield from cur[1]
        cur = nxt


# Batches whose ids are sorted ahead on the side stream; 0 = the sort runs in line, inside the step's library call.
# Default 0: measured with depth 2 at B = 8192 the loop is host-bound (52 us per step against 41 us in line: three event
# operations and a second library call per step cost more than the 15 us sort they hide), and at B = 262 144 the sort
# beside the HBM-bound update kernel slows that kernel by more than its own length (0.689 against 0.640 ms) -- the same
# finding as for the GloVe step, where only filling the gaps BETWEEN steps pays (wikipedia.train_epoch).
_LOOP_DEPTH = max(0, int(_os.environ.get("ESR_STL_PRESORT_DEPTH", "0")))
# Batches whose id lists are sorted together, by one library call on the main stream, before the first of them is
# stepped (esr_segment_sort_ids_batched; ESR_STL_SORT_BATCH=1: every step sorts its own list in line).  Up to 2^20 ids per
# list (the batched radix passes: the sort of ONE long list is a chain of launches of a few hundred workgroups each --
# 92 us for the 786 432 ids of a 262 144-triplet batch, latency more than work).
_SORT_BATCH = min(8, max(1, int(_os.environ.get("ESR_STL_SORT_BATCH", "8"))))
_SORT_BATCH_MAX_IDS = int(_os.environ.get("ESR_STL_SORT_BATCH_MAX_IDS", str(1 << 20)))
# ESR_STL_PLAN_STREAM=side: the sort + plan of a group on the second stream, beside the steps of the group before it
# Default "auto" (round 5): on the second stream for groups of 4096 .. 131072 triplets stepped by train_steps' group calls --
# same box, alternating runs: B = 8192 333 -> 365 M triplets/s (the 41 us of sort + plan per group leave the steps'
# stream), B = 65536 556 -> 574 M; at B = 2048 the cross-stream hand-over costs more than the 10 us it hides (132 -> 111 M),
# at B = 262144 the sort steals bandwidth from the HBM-bound step kernel (555 -> 540 M), and the per-step calls of
# presorted() pay an event wait per group for nothing (296 -> 272 M): those stay on the main stream.
_PLAN_STREAM = _os.environ.get("ESR_STL_PLAN_STREAM", "auto")


def _plan_on_side(B):
    if _PLAN_STREAM == "auto":
        return 4096 <= B <= 131072
    return _PLAN_STREAM == "side"
# ESR_STL_HINT_WAIT=0: a group whose long-run hints have not reached the host is stepped without them (A/B knob)
_HINT_WAIT = _os.environ.get("ESR_STL_HINT_WAIT", "1") == "1"


def _inbatch_steps(state, it, first, num_steps, regularization, batch_size, scale, precision):
    """train_steps for in-batch batches ``(scene, pos, None)``: ``train_step`` per batch, the occurrence lists
    [scene ; Vs + pos] of up to eight coming batches sorted by one batched call in front of their steps."""
    _, st, pt = _tables(state)
    dev, Vs, Vp = st.device, st.shape[0], pt.shape[0]
    sparse = not getattr(state.tx, "wants_dense", False)

    def ids_of(batch):
        return (ops.as_ids(batch[0], dev, check_range=Vs).reshape(-1), ops.as_ids(batch[1], dev, check_range=Vp).reshape(-1))
    losses, pending, dry = [], [ids_of(first)], False
    drawn = 0  # steps whose batches have been drawn and grouped

    def next_group():
        """The handles of the next group of steps (None when all have been drawn): batches drawn, their lists sorted
        and screened for long runs -- launches queued NOW, i.e. in front of the steps of the group before."""
        nonlocal pending, dry, drawn
        k0 = drawn
        if k0 >= num_steps:
            return None
        if not pending:
            if dry:
                raise StopIteration("train_steps: the batch iterator ended after %d of %d steps" % (k0, num_steps))
            pending = [ids_of(next(it))]
        # the FIRST step goes out alone, sorting its own list in line: the GPU starts after one batch's worth of host work,
        # and the first group of eight is drawn and sorted while that step runs (drawing eight batches first left the
        # device idle for ~0.2 ms at the head of every call -- 4 % of a 20-step run)
        # (then groups of 2, 4, 8: drawing and sorting a group of eight is ~0.2 ms of host work, more than the one step
        # that is in flight behind it)
        want = min(_SORT_BATCH, num_steps - k0, 1 if k0 == 0 else (2 if k0 < 3 else (4 if k0 < 7 else _SORT_BATCH)))
        while len(pending) < want and not dry:
            try:
                pending.append(ids_of(next(it)))
            except StopIteration:
                dry = True
        group, pending = pending[:want], pending[want:]
        drawn += len(group)
        n = 2 * group[0][0].numel()
        if sparse and _SORT_BATCH > 1 and len(group) > 1 and n <= _SORT_BATCH_MAX_IDS and \
                all(g[0].numel() == group[0][0].numel() == g[1].numel() for g in group):
            srt, prm = ops.segment_sort_batched([list(g) for g in group], (0, Vs), Vs + Vp)
            # one screening launch for the whole group (the [nb, n] array as one list: a run across two lists can only
            # make the answer "long"): hint word in pinned memory, read once its event is complete -- a group without
            # hot ids then skips the long-run launch of every optimizer step (5 us of 237 at C2)
            from ..wikipedia.train_cooccurence import _hint_slot
            hh, gen = _hint_slot(dev)
            ops.long_run_hint(srt.reshape(-1), 32, hh, gen)
            ev = torch.cuda.Event()
            ev.record()
            hint = (hh, gen, ev)
            return [PresortedTriplets(g[0], g[1], None, srt[j], prm[j], None, hint) for j, g in enumerate(group)]
        return [PresortedTriplets(g[0], g[1], None, None, None, None) for g in group]

    cur, head = next_group(), True
    while cur is not None:
        # the NEXT group is sorted and screened in front of this group's steps: its hint is then on the host by the time
        # its first step is issued -- the wait below ends when the group before this one has run, i.e. with this whole
        # group still queued, so the device never runs dry while the host stays at most two groups ahead
        # (the very first step is issued before anything else is drawn: the GPU starts at once)
        nxt = None if head else next_group()
        if cur[0].hint is not None:
            cur[0].hint[2].synchronize()
        for h in cur:
            if h.sorted_ids is None:
                state, loss = train_step(state, h.scene, h.pos, None, regularization, batch_size, scale=scale,
                                         precision=precision)
            else:
                state, loss = train_step(state, h, None, None, regularization, batch_size, scale=scale,
                                         precision=precision)
            losses.append(loss)
        if head:
            nxt, head = next_group(), False
        cur = nxt
    return state, torch.stack(losses)


def train_steps(state, batches, num_steps, regularization, batch_size, scale=1.0, precision="auto", consolidate=True):
    """`num_steps` iterations of the reference's training loop body (pinterest/train_shop_the_look.py:195-204:
    ``state, loss = train_step(state, scene, pos, neg, regularization, batch_size)`` for each batch of the iterator
    `batches`, yielding ``(scene, pos_product, neg_product)``).  Returns ``(state, losses)`` with the per-step losses
    as one device tensor -- the loop never synchronises.  Batches with ``neg_product = None`` are in-batch-softmax steps
    (``scale`` / ``precision`` as for ``train_step``), their id lists sorted eight batches at a time as well.  Under
    ``optim.sparse_adagrad`` every triplet step is the one-pass step
    driven through a per-loop context: one library call per step, and the id lists of up to eight coming batches are
    drawn from the iterator and sorted together by one batched call in front of their steps (``ESR_STL_SORT_BATCH``;
    ``ESR_STL_PRESORT_DEPTH=n`` instead sorts the ids of the next n batches on a second stream -- measured slower, see
    _LOOP_DEPTH); otherwise it is ``train_step`` as is.  consolidate (default): the towers are plain tables again when the
    loop returns (TrainState.consolidate)."""
    from ..train_state import quiet_gc
    with quiet_gc():
        state, losses = _train_steps(state, batches, num_steps, regularization, batch_size, scale, precision)
    if consolidate:  # rows the one-pass steps left in the towers' second buffers go back: aliases of the tables are current
        state.consolidate()
    return state, losses


def _train_steps(state, batches, num_steps, regularization, batch_size, scale, precision):
    it = iter(batches)
    if num_steps > 0:
        first = next(it)
        if first[2] is None:  # in-batch batches (north_star): scale / precision as for train_step
            return _inbatch_steps(state, it, first, num_steps, regularization, batch_size, scale, precision)
        import itertools
        it = itertools.chain([first], it)
    if not fused_triplet_step_available(state) or num_steps <= 0:
        losses = []
        for _ in range(max(num_steps, 0)):
            scene, pos, neg = next(it)
            state, loss = train_step(state, scene, pos, neg, regularization, batch_size)
            losses.append(loss)
        dev = _tables(state)[1].device
        return state, (torch.stack(losses) if losses else torch.empty(0, device=dev))
    from collections import deque
    ctx = _FusedTripletLoop(state, num_steps, _LOOP_DEPTH)
    regularization, batch_size = float(regularization), float(batch_size)
    if _LOOP_DEPTH == 0:
        import time
        t_host = time.perf_counter()
        state_ = {"drawn": 0, "dry": False, "which": 0}

        def draw_group(first_alone):
            """Handles of the next group of steps (None when nothing is left to draw): the group's id lists sorted and
            planned by one call pair, or single batches that sort and plan inside their own step."""
            left = num_steps - state_["drawn"]
            if left <= 0 or state_["dry"]:
                return None
            try:
                first = ctx.ids(*next(it))
            except StopIteration:
                state_["dry"] = True
                return None
            state_["drawn"] += 1
            # (the very first step goes out alone with its sort in line, see _inbatch_steps)
            if first_alone or _SORT_BATCH <= 1 or 3 * first[0].numel() > _SORT_BATCH_MAX_IDS or left <= 1:
                return [(None,) + first]
            group = [first]
            for _ in range(min(_SORT_BATCH, left) - 1):
                try:
                    group.append(ctx.ids(*next(it)))
                except StopIteration:
                    state_["dry"] = True
                    break
                state_["drawn"] += 1
            if any(g[0].numel() != first[0].numel() for g in group) or len(group) == 1:  # ragged: own sorts
                return [(None,) + g for g in group]
            state_["which"] ^= 1
            return ctx.sort_batch(group, state_["which"])

        k = 0
        handles = draw_group(True)
        while k < num_steps:
            if handles is None:  # the iterator ended early: as the reference's loop, after the steps it did feed
                raise StopIteration("train_steps: the batch iterator ended after %d of %d steps" % (k, num_steps))
            # the NEXT group is drawn, sorted and planned before this one is stepped: its long-run hints reach the host
            # a whole group ahead of the steps that ask for them
            following = draw_group(False)
            if type(handles) is _Group:
                ctx.step_group(k, handles, regularization, batch_size)
                k += handles.nb
            else:
                for h in handles:
                    ctx.step(k, h, regularization, batch_size)
                    k += 1
            handles = following
        if _os.environ.get("ESR_TRACE_HOST") == "1":  # is the loop issuing steps faster than the GPU retires them?
            import logging
            logging.warning("train_steps: host issued %d steps in %.1f us each (no sync yet)", num_steps,
                            (time.perf_counter() - t_host) / num_steps * 1e6)
        return state.replace(step=state.step + num_steps), ctx.losses[:num_steps]
    queue, fetched = deque(), 0
    for k in range(num_steps):
        while fetched < num_steps and len(queue) < _LOOP_DEPTH + 1:
            scene, pos, neg = next(it)
            queue.append(ctx.presort(fetched % (_LOOP_DEPTH + 1), *ctx.ids(scene, pos, neg)))
            fetched += 1
        ctx.step(k, queue.popleft(), regularization, batch_size)
    return state.replace(step=state.step + num_steps), ctx.losses[:num_steps]


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
