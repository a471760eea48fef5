"""GloVe trainer hot path -- drop-in for the reference's ``wikipedia/train_cooccurence.py:71-134``.

``apply_model``, ``update_model``, ``train_epoch``, ``find_knn``, ``dump_knn`` and ``save_state`` keep
the reference's names, argument order and return order.  The step is three fused HIP launches for
loss + gradients (esr_glove_fwd_bwd) plus a sort + segment-reduce optimizer update; nothing is
traced or compiled at run time.
"""
import logging
import os
import types

import numpy as np
import torch

from .. import ops
from ..train_state import RowGrads, SegmentIndex
from .models import Glove

# Flags with the reference's names and defaults (wikipedia/train_cooccurence.py:34-65); absl is not
# a dependency, so this is a plain namespace a caller may overwrite.
FLAGS = types.SimpleNamespace(
    train_input_pattern="data/wikipedia.cooccur.pb.b64.bz2/part-?????.bz2",
    token_dictionary="data/dictionaries/token.tstat.pb.b64.bz2",
    max_terms=20,
    embedding_dim=64,
    batch_size=2048,
    seed=1701,
    shuffle_buffer_size=5000000,
    terms="news,apple,computer,physics,neural,democracy,singapore,livermore",
    checkpoint_dir="data/wikipedia_training",
    checkpoint_every_epochs=20,
    resume_checkpoint=None,
    steps_per_epoch=10000,
    num_epochs=20,
    learning_rate=0.001,
)

_MODES = {"reference": ops.GLOVE_REFERENCE, "diagonal": ops.GLOVE_DIAGONAL}


def _model_of(state):
    model = getattr(state.apply_fn, "__self__", None)
    return model if isinstance(model, Glove) else None


def apply_model(state, inputs, target):
    """Computes the gradients and loss for a single batch (wikipedia/train_cooccurence.py:71-89).

    Returns ``(grads, loss)`` -- grads first, as the reference does.  ``grads`` has the reference's tree
    ``{'_token_embedding': {'embedding': g}, '_bias': {'embedding': g}}``; the leaves are dense [V, D] /
    [V, 1] tensors when the optimizer is the reference's dense Adam, else row-sparse ``RowGrads``.
    ``loss`` is a 0-dim device tensor (float(loss) synchronises, like a JAX device scalar)."""
    emb = state.params["_token_embedding"]["embedding"]
    bias = state.params["_bias"]["embedding"]
    model = _model_of(state)
    mode = _MODES[model.loss_mode if model is not None else "reference"]
    V = emb.shape[0]
    inputs = ops.as_ids(inputs, emb.device, check_range=V)
    target = ops.as_f32(target, emb.device)
    index = SegmentIndex(inputs.reshape(-1), V)  # occurrence ids = [token1 ; token2]
    loss, grad_rows, grad_bias = ops.glove_fwd_bwd(emb, bias, inputs, target, mode)
    g_emb = RowGrads(index, grad_rows, emb.shape)
    g_bias = RowGrads(index, grad_bias.reshape(-1, 1), bias.shape)
    if getattr(state.tx, "wants_dense", False):
        g_emb, g_bias = g_emb.to_dense(), g_bias.to_dense()
    grads = {"_token_embedding": {"embedding": g_emb}, "_bias": {"embedding": g_bias}}
    return grads, loss.reshape(())


class ColumnArgsort:
    """``indices`` of find_knn: the stable ascending argsort of every column of `scores` [V, T], computed ON DEMAND.
    The reference's only consumer reads the last ten rows (dump_knn, wikipedia/train_cooccurence.py:114-126), so
    ``indices[-k:]`` (k <= 1024) is a radix select per column (esr_topk_columns) -- exactly the rows the full argsort would
    put there, ties included -- and the full [V, T] argsort (``indices.full()``, any other indexing, ``.cpu()``) is only
    run for a caller that really reads all of it."""

    def __init__(self, scores):
        self._scores = scores
        self._full = None
        self.shape = tuple(scores.shape)
        self.dtype = torch.int32
        self.device = scores.device

    def top(self, k):
        """(values [k, T], rows [k, T]) of the k largest entries of every column, best first."""
        s, i = ops.topk_columns(self._scores, k)
        return s.t().contiguous(), i.t().contiguous()

    def full(self):
        if self._full is None:
            self._full = ops.argsort_columns(self._scores)
        return self._full

    def __getitem__(self, key):
        V = self.shape[0]
        if self._full is None and isinstance(key, slice) and key.step is None and key.stop is None and \
                isinstance(key.start, int) and key.start < 0 and -key.start <= min(V, 1024):
            return self.top(-key.start)[1].flip(0)      # rows V - k .. V - 1 of the ascending argsort
        return self.full()[key]

    def __len__(self):
        return self.shape[0]

    def __getattr__(self, name):  # anything else a tensor can do: on the full argsort
        return getattr(self.full(), name)


def find_knn(model, params, token):
    """scores [V, T] and ``indices``, the stable ascending argsort over V of every column (wikipedia/train_cooccurence.py:
    91-97) -- as a ColumnArgsort: ``indices[-10:]`` (all the reference reads) costs a radix select, not a sort of V."""
    scores = model.apply({"params": params}, token, method=Glove.score_all)
    return scores, ColumnArgsort(scores)


def update_model(state, grads):
    """wikipedia/train_cooccurence.py:99-101."""
    return state.apply_gradients(grads=grads)


def fused_step_available(state):
    """True when ``train_step`` can run the whole step in one pass (esr_glove_train_step): the build's row-sparse
    Adagrad on fp32 tables.  ``ESR_GLOVE_FUSED=0`` forces the apply_model + update_model path."""
    from ..train_state import _SparseAdagrad
    if os.environ.get("ESR_GLOVE_FUSED", "1") != "1" or not isinstance(state.tx, _SparseAdagrad):
        return False
    p = state.raw_params
    try:
        emb, bias = p["_token_embedding"]["embedding"], p["_bias"]["embedding"]
    except (KeyError, TypeError):
        return False
    # (bf16 embedding rows -- BASELINE config 4's dtype -- keep fp32 accumulators and an fp32 bias table)
    if not (emb.is_cuda and emb.dtype in (torch.float32, torch.bfloat16) and bias.dtype == torch.float32 and
            emb.shape[0] < (1 << 30)):
        return False
    # the one-pass step keeps the embedding table double-buffered: without room for the second buffer (a table that
    # fills the card) the gradient-row path runs instead
    from ..train_state import can_double_buffer
    return can_double_buffer(state, [("_token_embedding", "embedding")])


class PresortedInputs:
    """The occurrence ids of a batch, already on the device and sorted on the side stream (``presort_inputs``): the sort
    needs the ids only, so ``train_epoch`` runs it for batch k + 1 while batch k's update kernel streams the rows.  With
    the batch's counts at hand the step's plan record is made there too (``plan``; ``hint`` = (pinned int32 [1], gen):
    the word equals gen when a token has a run longer than a chunk -- only then does the step need its long-run launch)."""

    def __init__(self, inputs, sorted_ids, perm, event, plan=None, hint=None, target=None):
        self.inputs, self.sorted_ids, self.perm, self.event = inputs, sorted_ids, perm, event
        self.plan, self.hint, self.target = plan, hint, target
        self._done = event

    def take(self, host_wait_s=0.0):
        """(sorted_ids, perm) for the current stream (waits for the side stream's event, once).  host_wait_s > 0 (the
        epoch loop): the HOST waits up to that long for the event first, and a finished event costs the stream nothing --
        a wait queued on an unfinished one is a cross-queue barrier in front of the step, ~12 us of idle main queue at
        C3 although the sort had ended a step earlier by the time the barrier was reached.  The loop sorts two batches
        ahead, so the host still runs a step or more in front of the GPU."""
        if self.event is not None:
            ev = self.event
            if host_wait_s > 0.0 and not ev.query():
                import time
                t_end = time.perf_counter() + host_wait_s
                while not ev.query() and time.perf_counter() < t_end:
                    pass
            if not ev.query():
                torch.cuda.current_stream(self.inputs.device).wait_event(ev)
            self.event = None
        return self.sorted_ids, self.perm

    def long_runs(self):
        """0 / 1 from the plan's hint once it has reached the host, -1 while it has not (or without a plan)."""
        if self.hint is None or self._done is None or not self._done.query():
            return -1
        return hint_value(self.hint)


# ESR_GLOVE_PRESORT=0 sorts in line instead of one batch ahead on the side stream.  ESR_GLOVE_STEP_BLOCKS_PER_CU caps the
# update kernel's residency (experiment knob: making room for the sort to run BESIDE it measured slower).
_PRESORT = os.environ.get("ESR_GLOVE_PRESORT", "1") == "1"
_PRESORT_DEPTH = max(1, int(os.environ.get("ESR_GLOVE_PRESORT_DEPTH", "2")))  # batches sorted ahead (train_epoch)
_SORT_BATCH = min(8, max(1, int(os.environ.get("ESR_GLOVE_SORT_BATCH", "8"))))  # short lists sorted together (train_epoch)
_STEP_BLOCKS_PER_CU = int(os.environ.get("ESR_GLOVE_STEP_BLOCKS_PER_CU", "0"))
# train_epoch: how long the host waits for a side-stream sort before it queues a stream wait instead (PresortedInputs.take)
_HINT_WAIT = os.environ.get("ESR_GLOVE_HINT_WAIT", "1") == "1"
_HOST_WAIT_S = float(os.environ.get("ESR_GLOVE_HOST_WAIT_US", "0")) * 1e-6


_hint_ring = {}  # device -> [pinned int32 [R], next slot, next gen]
_start_words = {}  # device -> [int32 [1] device word, sequence number of the latest step that announces itself there]


def _start_word(dev):
    """The device word in which the one-pass steps of the epoch loops announce themselves (esr_glove_train_step's
    start_flag) and its running sequence number; one per device for the life of the process, so a gate left over from
    an earlier epoch never reads freed memory."""
    key = (dev.type, dev.index)
    if key not in _start_words:
        _start_words[key] = [torch.zeros(1, dtype=torch.int32, device=dev), 0]
    return _start_words[key]
_RESOLVE_MIN_IDS = 32768  # esr_glove.hip kResolveMinIds: longer lists resolve their records per step, plans are unused


def _hint_slot(dev):
    """One word of a small ring of PINNED host memory for a long-run hint: (view [1], gen).  The kernels write the word
    themselves (pinned memory is mapped into the device's address space): no copy launch."""
    R = 16
    key = (dev.type, dev.index)
    if key not in _hint_ring:
        _hint_ring[key] = [torch.zeros(R, dtype=torch.int32).pin_memory(), 0, 1]
    ring = _hint_ring[key]
    i, gen = ring[1], ring[2]
    ring[1], ring[2] = (i + 1) % R, gen + 1 if gen < 2 ** 31 - 2 else 1
    word = ring[0][i:i + 1]
    _hint_owner[word.data_ptr()] = gen  # the slot is this generation's until the ring comes round to it again
    return word, gen


_hint_owner = {}  # address of a ring word -> the generation it was last handed to


def hint_value(hint):
    """1 / 0 = the hint kernel of `hint` = (word, gen, ...) found / did not find a long run -- valid only while the word
    is still this generation's.  A kernel writes its gen into the word only when it finds a long run, so a word that a
    LATER hint has been handed (more than a ring's worth of handles outstanding) would read as "no long run" whatever
    this generation's kernel had found: that case answers -1 (unknown: the step makes its long-run launch)."""
    word, gen = hint[0], hint[1]
    if _hint_owner.get(word.data_ptr()) != gen:
        return -1
    return 1 if int(word[0]) == gen else 0


def presort_inputs(state, inputs, target=None, after=None, out=None):
    """Move `inputs` to the device and sort its occurrence ids on the side stream; with `target` (the batch's counts) the
    step's plan record is made there as well (ops.glove_plan: it needs ids and counts only).  Returns a PresortedInputs
    to pass as ``train_step(..., inputs=that)``.  `out` = (sorted_ids, perm) buffers of the caller's to sort into: the
    caller then answers for their lifetime (train_epoch's ring) -- tensors allocated here are handed to the main stream
    with record_stream, and freeing such a tensor makes the allocator record an event on the main stream: two markers
    between a step's last kernel and the next step's first, ~11 us of idle queue per step at C3."""
    from ..train_state import _side_stream
    emb = state.raw_params["_token_embedding"]["embedding"]
    V = emb.shape[0]
    ids = ops.as_ids(inputs, emb.device, check_range=V)
    tgt = ops.as_f32(target, emb.device) if target is not None else None
    main = torch.cuda.current_stream(emb.device)
    side = _side_stream(emb.device)
    if after is None:
        side.wait_stream(main)  # the ids may have been produced (copied) on the main stream
    elif not isinstance(after, tuple):
        side.wait_event(after)  # an event of the main stream behind which the ids are ready
    plan = hint = None
    with torch.cuda.stream(side):
        if isinstance(after, tuple):
            # (flag, value): the start word of a step NOT YET ISSUED when the ids were drawn (train_epoch) -- whatever
            # the main stream still has to do to produce them is queued in front of that step's update kernel, and the
            # sort arrives after that kernel has taken its wave slots.  No marker on the main queue.
            ops.stream_gate(after[0], after[1])
        sorted_ids, perm = ops.segment_sort(ids.reshape(-1), V, out=out)
        if tgt is not None and ids.dim() == 2 and ids.shape[0] == 2 and tgt.numel() == ids.shape[1]:
            # (longer lists: the step resolves its own records in front of its update kernel and makes every launch --
            # its long-run launch also reduces the loss partials -- so neither a plan nor the hint is of use)
            if ids.numel() <= _RESOLVE_MIN_IDS:
                hh, gen = _hint_slot(emb.device)
                plan = ops.glove_plan([ids], [tgt], sorted_ids, perm, hints=hh, gen=gen)
                hint = (hh, gen)
        event = torch.cuda.Event()
        event.record(side)
    ids.record_stream(side)
    if out is None:
        sorted_ids.record_stream(main)
        perm.record_stream(main)
    if plan is not None:
        plan.record_stream(main)
    if tgt is not None:
        tgt.record_stream(side)
    return PresortedInputs(ids, sorted_ids, perm, event, plan, hint, tgt)


def train_step(state, inputs, target):
    """``apply_model`` + ``update_model`` (wikipedia/train_cooccurence.py:71-101) in ONE pass over the rows, for the
    build's sparse Adagrad: returns ``(new_state, loss)``.  `inputs` may be a PresortedInputs (see presort_inputs).
    No gradient is materialised -- the update kernel re-reads each occurrence's partner row and forms its gradient on
    chip -- which the double-buffered embedding table makes safe (train_state.RowVersions; ``state.params``
    consolidates on access).  Tables, accumulators and loss agree with the two-call path to an f32 rounding."""
    from ..train_state import next_stamp, row_versions
    presorted = None
    if isinstance(inputs, PresortedInputs):
        presorted, inputs = inputs, inputs.inputs
    if not fused_step_available(state):
        grads, loss = apply_model(state, inputs, target)
        return update_model(state, grads), loss
    p = state.raw_params
    emb, bias = p["_token_embedding"]["embedding"], p["_bias"]["embedding"]
    model = _model_of(state)
    mode = _MODES[model.loss_mode if model is not None else "reference"]
    inputs = ops.as_ids(inputs, emb.device, check_range=emb.shape[0])
    target = ops.as_f32(target, emb.device)
    rv = row_versions(state, ("_token_embedding", "embedding"))
    acc = state.opt_state["sum_of_squares"]
    if presorted is not None and presorted.plan is not None:
        # a plan's statistics / loss words are zeroed by the plan kernel only: a second step on the same plan would add
        # to the first step's sums (doubled loss and gradient, nothing to report it)
        if getattr(presorted, "_consumed", False):
            raise RuntimeError("this PresortedInputs' plan has already been stepped: presort_inputs again for a second step")
        presorted._consumed = True
    loss = ops.glove_train_step(emb, rv.shadow, rv.loc, acc["_token_embedding"]["embedding"], bias,
                                acc["_bias"]["embedding"], inputs, target, mode, state.tx.lr, state.tx.eps,
                                presorted=presorted.take() if presorted is not None else None,
                                blocks_per_cu=_STEP_BLOCKS_PER_CU if presorted is not None else 0,
                                stamp=next_stamp(rv), plan=presorted.plan if presorted is not None else None,
                                long_runs=presorted.long_runs() if presorted is not None else -1)
    return state.replace(step=state.step + 1), loss.reshape(())


# train_epoch: lists of up to _GROUP_SORT_MAX_IDS ids are sorted a group of _SORT_BATCH batches at a time by one batched
# call on the main stream (esr_segment_sort_ids_batched); longer ones one by one on the side stream, _PRESORT_DEPTH ahead
_GROUP_SORT_MAX_IDS = int(os.environ.get("ESR_GLOVE_GROUP_SORT_MAX_IDS", str(1 << 21)))
_PRESORT_MIN_IDS = _GROUP_SORT_MAX_IDS


def _ids_count(inputs):
    x = inputs.inputs if isinstance(inputs, PresortedInputs) else inputs
    if type(x) is torch.Tensor:
        return x.numel()
    return int(np.prod(x.shape)) if hasattr(x, "shape") else 2 * len(x[0])


class _Group:
    """A group of batches whose id lists were sorted and planned together (``_FusedEpoch.sort_batch``): the raw pointers
    of everything esr_glove_train_steps takes, and the tensors, kept alive."""
    __slots__ = ("nb", "B", "in_ptrs", "tgt_ptrs", "sorted_ptr", "perm_ptr", "plans_ptr", "which", "gen", "keep")

    def __init__(self, nb, B, in_ptrs, tgt_ptrs, sorted_ptr, perm_ptr, plans_ptr, which, gen, keep):
        self.nb, self.B, self.in_ptrs, self.tgt_ptrs = nb, B, in_ptrs, tgt_ptrs
        self.sorted_ptr, self.perm_ptr, self.plans_ptr = sorted_ptr, perm_ptr, plans_ptr
        self.which, self.gen, self.keep = which, gen, keep


class _FusedEpoch:
    """What the one-pass step needs that does not change from batch to batch, resolved once per epoch: table / accumulator
    / RowVersions pointers, the library entry point, one loss slot per step and a reusable workspace.  At the reference's
    default batch (2048 pairs: wikipedia/train_cooccurence.py:45) the step is five short kernels and the per-step Python
    of ``train_step`` (state lookups, argument checks, two allocations) took longer than they do."""

    def __init__(self, state, steps):
        from .. import _lib
        from ..train_state import next_stamp, row_versions
        self.next_stamp = next_stamp
        p = state.raw_params
        self.emb, self.bias = p["_token_embedding"]["embedding"], p["_bias"]["embedding"]
        acc = state.opt_state["sum_of_squares"]
        self.acc_e, self.acc_b = acc["_token_embedding"]["embedding"], acc["_bias"]["embedding"]
        self.rv = row_versions(state, ("_token_embedding", "embedding"))
        model = _model_of(state)
        self.mode = _MODES[model.loss_mode if model is not None else "reference"]
        self.lr, self.eps = float(state.tx.lr), float(state.tx.eps)
        self.V, self.D = self.emb.shape
        self.dev = self.emb.device
        self.lib = _lib.load()
        self.check = _lib.check
        self.losses = torch.empty(max(steps, 1), dtype=torch.float32, device=self.dev)
        self.losses_ptr = self.losses.data_ptr()
        self.sort_ring = None
        self.steps_issued = 0
        self.uses_gate = False  # a side-stream sort waits on the start word: every issued step must advance it
        self.ws, self.ws_B = None, -1
        self.sort_buf = [None, None]  # two sets: a group is sorted and planned while the one before it is stepped
        self.long_buf = [None, None]  # the same for lists too long for plans (_sort_batch_long)
        self.which = 0
        import ctypes
        self.sort_ptrs = [(ctypes.c_void_p * _SORT_BATCH)(), (ctypes.c_void_p * _SORT_BATCH)()]
        self.tgt_ptrs = [(ctypes.c_void_p * _SORT_BATCH)(), (ctypes.c_void_p * _SORT_BATCH)()]
        self.long_arr = (ctypes.c_int32 * _SORT_BATCH)()
        # esr_glove_plan's long-run hints, copied to pinned memory behind the plan launch (see _FusedTripletLoop)
        self.hints_host = torch.zeros((2, _SORT_BATCH), dtype=torch.int32).pin_memory()  # written by the plan kernel
        self.hints_event = [torch.cuda.Event(), torch.cuda.Event()]
        self.hints_known = [None, None]
        self.gen = 0
        self.sort_cnt, self.sort_off = (ctypes.c_int64 * 1)(0), (ctypes.c_int64 * 1)(0)
        self.start = _start_word(self.dev)  # [int32 [1] device word, sequence number of the latest step issued]
        self.check_ids = os.environ.get("ESR_CHECK_IDS") == "1"
        self.fixed = (self.emb.data_ptr(), self.rv.shadow.data_ptr(), self.rv.loc.data_ptr(), self.acc_e.data_ptr(),
                      self.bias.data_ptr(), self.acc_b.data_ptr(), self.V, ops._table_dtype(self.emb, "embedding table"),
                      self.D)

    def sort_batch(self, group):
        """[(inputs, target), ...] of the coming batches -> a _Group: their id lists sorted by one batched call and their
        plan records made by one more on the current stream, everything ``step_group`` hands the library resolved here
        (batches of unequal size or longer than the two-launch sort takes: the list as it is, every step on its own)."""
        ids = [ops.as_ids(inp, self.dev, check_range=self.V) for inp, _ in group]
        n = ids[0].numel()
        if n > _GROUP_SORT_MAX_IDS or any(t.numel() != n for t in ids) or \
                any(t.dim() != 2 or t.shape[0] != 2 for t in ids):
            return [(i, t) for i, (_, t) in zip(ids, group)]  # as they are: every step sorts and plans its own
        if self.check_ids:
            for i in ids:
                ops.check_device_ids(i, self.V)
        nb = len(ids)
        self.which ^= 1
        which = self.which
        B = n // 2
        if n > _RESOLVE_MIN_IDS:
            return self._sort_batch_long(ids, group, n, which)
        pbytes = ops._ws_bytes("esr_glove_plan_bytes", B)
        if self.sort_buf[which] is None or self.sort_buf[which][0].shape[1] != n:
            self.sort_buf[which] = (torch.empty((_SORT_BATCH, n), dtype=torch.int32, device=self.dev),
                                    torch.empty((_SORT_BATCH, n), dtype=torch.int32, device=self.dev),
                                    ops._ws(ops._ws_bytes("esr_segment_sort_batched_workspace_bytes", n, _SORT_BATCH),
                                            self.dev),
                                    ops._aligned_bytes(_SORT_BATCH * pbytes, self.dev))
        srt, prm, ws, plans = self.sort_buf[which]
        tgts = []
        for _, t in group:
            if not (type(t) is torch.Tensor and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                t = ops.as_f32(t, self.dev)
            if t.numel() * 2 != n:
                raise ValueError("inputs must be [2, B] and target [B]")
            tgts.append(t)
        # (the library calls themselves: ops.segment_sort_batched re-validates every tensor and rebuilds its ctypes arrays)
        sort_ptrs, tgt_ptrs = self.sort_ptrs[which], self.tgt_ptrs[which]
        for b, (i, t) in enumerate(zip(ids, tgts)):
            sort_ptrs[b] = i.data_ptr()
            tgt_ptrs[b] = t.data_ptr()
        self.sort_cnt[0] = n
        st = ops._stream()
        self.check(self.lib.esr_segment_sort_ids_batched(sort_ptrs, self.sort_cnt, self.sort_off, 1, nb, self.V,
                                                         srt.data_ptr(), prm.data_ptr(), ws.data_ptr(), ws.numel(),
                                                         st), "esr_segment_sort_ids_batched")
        self.gen += 1
        self.check(self.lib.esr_glove_plan(sort_ptrs, tgt_ptrs, nb, B, srt.data_ptr(), prm.data_ptr(),
                                           plans.data_ptr(), self.hints_host[which].data_ptr(), self.gen, st),
                   "esr_glove_plan")
        self.hints_event[which].record()
        self.hints_known[which] = None
        # the whole group is stepped by ONE library call (esr_glove_train_steps): at 2048 pairs a step is ~20 us of
        # kernels, less than a 25-argument foreign call plus the Python around it
        return _Group(nb, B, self.sort_ptrs[which], self.tgt_ptrs[which], srt.data_ptr(), prm.data_ptr(),
                      plans.data_ptr(), which, self.gen, (ids, tgts))

    def _sort_batch_long(self, ids, group, n, which):
        """Lists of more than _RESOLVE_MIN_IDS ids (the steps resolve their own records: no plans): the lists of the group
        sorted by ONE batched call on the current stream, each step then issued on its own with its slice.  The sort of
        one such list is four launches of 64 workgroups -- latency, not work -- so eight lists cost little more than one,
        the update kernels run with nothing beside them and no second stream is involved (a sort on the side stream
        shares the chip with the update kernel: at C3 it stretched to the length of a step and held the loop to its own
        pace, 0.143 ms)."""
        nb = len(ids)
        if self.long_buf[which] is None or self.long_buf[which][0].shape[1] != n:
            self.long_buf[which] = (torch.empty((_SORT_BATCH, n), dtype=torch.int32, device=self.dev),
                                    torch.empty((_SORT_BATCH, n), dtype=torch.int32, device=self.dev),
                                    ops._ws(ops._ws_bytes("esr_segment_sort_batched_workspace_bytes", n, _SORT_BATCH),
                                            self.dev))
        srt, prm, ws = self.long_buf[which]
        sort_ptrs = self.sort_ptrs[which]
        for b, i in enumerate(ids):
            sort_ptrs[b] = i.data_ptr()
        self.sort_cnt[0] = n
        self.check(self.lib.esr_segment_sort_ids_batched(sort_ptrs, self.sort_cnt, self.sort_off, 1, nb, self.V,
                                                         srt.data_ptr(), prm.data_ptr(), ws.data_ptr(), ws.numel(),
                                                         ops._stream()), "esr_segment_sort_ids_batched")
        # one screening launch for the group (the [nb, n] array as one list: a run across two lists can only make the
        # answer "long"): on a cleared hint the steps skip their long-run launch (5 us of 134 at C3)
        hh, gen = _hint_slot(self.dev)
        ops.long_run_hint(srt[:nb].reshape(-1), 32, hh, gen)
        ev = torch.cuda.Event()
        ev.record()
        out = []
        for b, (i, (_, t)) in enumerate(zip(ids, group)):
            pre = PresortedInputs(i, srt[b], prm[b], None, hint=(hh, gen))
            pre._done = ev
            out.append((pre, t))
        return out

    def step_group(self, k, gr):
        """Steps k .. k + gr.nb - 1: the batches of a sorted and planned group, issued by one library call."""
        known = self.hints_known[gr.which]
        if known is None:
            # the HOST waits for the group's plan launch (queued in front of the steps of the group before it: the wait
            # ends with a group's worth of work still queued), so that every step knows whether it needs its long-run
            # launch -- see pinterest.train_shop_the_look._FusedTripletLoop.step_group
            if _HINT_WAIT:
                self.hints_event[gr.which].synchronize()
            if self.hints_event[gr.which].query():
                known = self.hints_known[gr.which] = self.hints_host[gr.which].tolist()
        long_runs = None
        if known is not None:  # (else: the hint has not reached the host; the library makes every long-run launch)
            long_runs = self.long_arr
            for j in range(gr.nb):
                long_runs[j] = 1 if known[j] == gr.gen else 0
        if gr.B != self.ws_B:
            self.ws = ops._ws(ops._ws_bytes("esr_glove_step_workspace_bytes", gr.B, self.D), self.dev)
            self.ws_B = gr.B
        self.check(self.lib.esr_glove_train_steps(*self.fixed, gr.nb, gr.in_ptrs, gr.tgt_ptrs, gr.B, self.mode, self.lr,
                                                  self.eps, self.next_stamp(self.rv, count=gr.nb), gr.sorted_ptr,
                                                  gr.perm_ptr, gr.plans_ptr, long_runs, self.losses_ptr + 4 * k,
                                                  self.ws.data_ptr(), self.ws.numel(), ops._stream()),
                   "esr_glove_train_steps")
        self.steps_issued += gr.nb
        if self.uses_gate:
            # (an epoch that mixes lists beyond the grouped sort's limit with short ones: the group call announces no
            # start, so the word is advanced behind it -- a gated sort never waits for its timeout)
            start = self.start
            start[1] = (start[1] + 1) & 0xFFFFFFFF
            start[0].fill_(start[1] if start[1] < 2 ** 31 else start[1] - 2 ** 32)

    def sort_slot(self, n, j):
        """(sorted_ids, perm) buffers for the side-stream sort of the j-th batch drawn: a ring of _PRESORT_DEPTH + 3.
        The sort of batch j is released by the mark of step j - depth - 1 at the latest (refill runs behind a step and
        tops the queue up to depth + 1), i.e. after step j - depth - 2 has finished on the main stream -- the last
        reader of the slot was step j - depth - 3."""
        R = _PRESORT_DEPTH + 3
        if self.sort_ring is None or self.sort_ring[0][0].numel() != n:
            self.sort_ring = [(torch.empty(n, dtype=torch.int32, device=self.dev),
                               torch.empty(n, dtype=torch.int32, device=self.dev)) for _ in range(R)]
        return self.sort_ring[j % R]

    def step(self, k, inputs, target):
        presorted, plan_ptr, long_runs = None, 0, -1
        if isinstance(inputs, PresortedInputs):
            if _HINT_WAIT and inputs.hint is not None and inputs.event is None and inputs._done is not None and \
                    not inputs._done.query():
                # (a group-sorted list: its screening launch was queued in front of the previous group's steps -- see
                # step_group; the wait ends with that group's work still queued)
                inputs._done.synchronize()
            long_runs = inputs.long_runs()
            if inputs.plan is not None:
                plan_ptr = inputs.plan.data_ptr()
            if inputs.target is not None:
                target = inputs.target
            presorted, inputs = inputs.take(_HOST_WAIT_S), inputs.inputs
        if not (type(inputs) is torch.Tensor and inputs.is_cuda and inputs.dtype == torch.int32 and
                inputs.is_contiguous()):
            inputs = ops.as_ids(inputs, self.dev, check_range=self.V)
        elif self.check_ids:
            ops.check_device_ids(inputs, self.V)
        if not (type(target) is torch.Tensor and target.is_cuda and target.dtype == torch.float32 and
                target.is_contiguous()):
            target = ops.as_f32(target, self.dev)
        B = inputs.shape[1]
        if inputs.shape[0] != 2 or target.numel() != B:
            raise ValueError("inputs must be [2, B] and target [B]")
        if B != self.ws_B:
            self.ws = ops._ws(ops._ws_bytes("esr_glove_step_workspace_bytes", B, self.D), self.dev)
            self.ws_B = B
        sid, perm = (presorted[0].data_ptr(), presorted[1].data_ptr()) if presorted is not None else (0, 0)
        # the update kernel stores this step's sequence number in the start word: side-stream sorts are gated on it
        start = self.start
        start[1] = seq = (start[1] + 1) & 0xFFFFFFFF
        self.check(self.lib.esr_glove_train_step(*self.fixed, inputs.data_ptr(), target.data_ptr(), B, self.mode,
                                                 self.lr, self.eps, self.next_stamp(self.rv), sid, perm, plan_ptr,
                                                 long_runs, 0, start[0].data_ptr(), seq,
                                                 self.losses.data_ptr() + 4 * k, self.ws.data_ptr(), self.ws.numel(),
                                                 ops._stream()),
                   "esr_glove_train_step")
        self.steps_issued += 1

    def next_start(self):
        """What gates the sort of a batch drawn NOW: (start word, sequence number of the next step to be issued) once a
        step of this epoch has been issued, else None (the first batches: the side stream waits for the main one)."""
        if self.steps_issued == 0:
            return None
        self.uses_gate = True
        return (self.start[0], (self.start[1] + 1) & 0xFFFFFFFF)


def train_epoch(state, steps_per_epoch, train_it, consolidate=True, losses_out=None):
    """Trains for an epoch (wikipedia/train_cooccurence.py:103-112).  Losses stay on the device until the
    epoch mean is taken, so the loop never synchronises.  With the build's sparse Adagrad every step is the
    one-pass step (``train_step``'s kernels, driven through a per-epoch context that keeps the per-step Python to one
    library call per group of steps); with the reference's dense Adam it is apply_model + update_model as there.
    consolidate (default): rows the one-pass steps left in the embedding table's second buffer are copied back before
    the epoch returns, so every holder of the table tensor sees current rows (one launch over the displaced rows).
    losses_out: a list that receives the per-step losses as one device tensor [steps_per_epoch] (the reference keeps
    them in ``losses`` before taking their mean, train_cooccurence.py:105-112)."""
    from ..train_state import quiet_gc
    with quiet_gc():  # (a full cyclic collection inside the loop is a 40 ms hole in the launch stream)
        state, loss = _train_epoch(state, steps_per_epoch, train_it, losses_out)
    if consolidate:
        state.consolidate()
    return state, loss


def _train_epoch(state, steps_per_epoch, train_it, losses_out=None):
    if fused_step_available(state) and steps_per_epoch > 0:
        ctx = _FusedEpoch(state, steps_per_epoch)
        # Batches are fetched _PRESORT_DEPTH ahead so that their ids can be sorted on the side stream under the update
        # kernels of the batches before them; worth it only when the list is long enough for the sort to be a chain of
        # launches (> 4096 ids).  Two ahead, not one: the update kernel fills every wave slot, so a sort issued beside
        # step k mostly runs in the gaps after it -- one ahead, its second radix pass (23 us) still sat between step k
        # and step k + 1 (profiles/r2/glove_kernel_stats.csv: the first scatter "takes" 106 us, stretched over the step).
        # Short lists (the reference's default batch of 2048 pairs: wikipedia/train_cooccurence.py:45) are pure launch
        # latency to sort: the lists of up to _SORT_BATCH coming batches go through ONE batched sort on the main stream
        # (esr_segment_sort_ids_batched) in front of their steps.
        from collections import deque
        import time
        queue, fetched, queued = deque(), 0, 0  # queue items: (inputs, targets) of one step, or a _Group; queued = steps
        t_host = time.perf_counter()
        grouped = False  # short lists: a whole group is drawn, sorted and planned at a time
        k, dry = 0, False

        def refill():
            nonlocal fetched, queued, grouped, dry
            # (grouped: the next group is sorted and planned while the one before it is still queued, so its long-run
            # hints reach the host a whole group ahead of the steps that ask for them)
            while fetched < steps_per_epoch and (queued <= (_SORT_BATCH if k > 1 else 0) if grouped
                                                 else queued < _PRESORT_DEPTH + 1) and not dry:
                try:
                    inputs, targets = next(train_it)
                except StopIteration:  # the iterator ended early: the steps it did feed run, then the loop raises
                    dry = True
                    break
                fetched += 1
                if _PRESORT and _ids_count(inputs) > _PRESORT_MIN_IDS:
                    grouped = False
                    queue.append((presort_inputs(state, inputs, targets, after=ctx.next_start(),
                                                 out=ctx.sort_slot(_ids_count(inputs), fetched)), targets))
                    queued += 1
                elif _SORT_BATCH > 1 and grouped and k > 0:  # (the first step goes out alone: the GPU starts at once)
                    group = [(inputs, targets)]
                    while len(group) < _SORT_BATCH and fetched < steps_per_epoch:
                        try:
                            group.append(next(train_it))
                        except StopIteration:  # ended early: the steps it did feed run, then the loop's own next() raises
                            break
                        fetched += 1
                    made = ctx.sort_batch(group) if len(group) > 1 else group
                    if type(made) is _Group:
                        queue.append(made)
                    else:
                        queue.extend(made)
                    queued += len(group)
                else:
                    grouped = _SORT_BATCH > 1  # short lists: refill when the queue has run low (then a group at a time)
                    queue.append((inputs, targets))
                    queued += 1

        refill()
        while k < steps_per_epoch:
            if not queue:
                raise StopIteration("train_epoch: the batch iterator ended after %d of %d steps" % (k, steps_per_epoch))
            item = queue.popleft()
            if type(item) is _Group:
                ctx.step_group(k, item)
                k += item.nb
                queued -= item.nb
            else:
                ctx.step(k, item[0], item[1])
                k += 1
                queued -= 1
            # the NEXT batches are drawn (and their ids sorted on the side stream) AFTER this step has been issued: the
            # side stream then waits for this step's last kernel, so a sort starts as the next step's update kernel does
            # and its passes run beside that kernel and in the gap behind it.  Issued in front of the step, the sort
            # started one cross-queue wait (~11 us) after the PREVIOUS step, ran its first pass in the gap in front of this
            # step's kernels and its second beside the update kernel: 0.166 against 0.152 ms per step at C3.
            refill()
        if os.environ.get("ESR_TRACE_HOST") == "1":  # is the loop issuing steps faster than the GPU retires them?
            logging.warning("train_epoch: host issued %d steps in %.1f us each (no sync yet)", steps_per_epoch,
                            (time.perf_counter() - t_host) / steps_per_epoch * 1e6)
        state = state.replace(step=state.step + steps_per_epoch)
        if losses_out is not None:
            losses_out.append(ctx.losses[:steps_per_epoch].clone())
        return state, float(ctx.losses[:steps_per_epoch].mean())
    epoch_loss = []
    for _ in range(steps_per_epoch):
        inputs, targets = next(train_it)
        grads, loss = apply_model(state, inputs, targets)
        state = update_model(state, grads)
        epoch_loss.append(loss)
    train_loss = float(torch.stack(epoch_loss).mean()) if epoch_loss else float("nan")
    if losses_out is not None and epoch_loss:
        losses_out.append(torch.stack(epoch_loss))
    return state, train_loss


def dump_knn(model, params, tokens, token_dictionary):
    """Dumps the 10 nearest neighbours of each probe token (wikipedia/train_cooccurence.py:114-126)."""
    scores, indices = find_knn(model, params, tokens)
    tokens_h = np.asarray(tokens.cpu() if isinstance(tokens, torch.Tensor) else tokens)
    top_scores, top = indices.top(10) if isinstance(indices, ColumnArgsort) else \
        (torch.gather(scores, 0, indices[-10:].flip(0).long()), indices[-10:].flip(0))   # rows -1 .. -10
    top, top_scores = top.cpu().numpy(), top_scores.cpu().numpy()
    lines = []
    for i in range(tokens_h.shape[0]):
        query_word = token_dictionary.get_token_from_embedding_index(int(tokens_h[i]))
        knn = ["%s:%f" % (token_dictionary.get_token_from_embedding_index(int(top[j, i])), top_scores[j, i])
               for j in range(top.shape[0])]
        line = "Nearest neighbors for %s: %s" % (query_word, " ".join(knn))
        logging.info(line)
        lines.append(line)
    return lines


def save_state(state, step, checkpoint_dir=None):
    """Saves the state of the model (wikipedia/train_cooccurence.py:129-134) as ``checkpoint-%05d.flax``."""
    from ..checkpoint import to_bytes
    filename = os.path.join(checkpoint_dir or FLAGS.checkpoint_dir, "checkpoint-%05d.flax" % step)
    with open(filename, "wb") as f:
        f.write(to_bytes(state))
    return filename
