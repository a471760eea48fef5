"""TrainState / optimizer objects mirroring what the reference's hot path uses from Flax + Optax.

Reference call sites: ``optax.adam(lr)`` + ``train_state.TrainState.create(apply_fn=..., params=..., tx=tx)``
(wikipedia/train_cooccurence.py:171-172, pinterest/train_shop_the_look.py:175-177) and
``state.apply_gradients(grads=grads)`` (train_cooccurence.py:101, train_shop_the_look.py:108).

Differences from JAX, by design (MI355X-first): parameter and optimizer tensors live in HBM and are
updated IN PLACE by HIP kernels; ``apply_gradients`` returns a new TrainState object that shares those
buffers (functional call shape, resident memory).  Gradients of embedding tables are row-sparse
(``RowGrads``) unless the optimizer asks for the reference's dense V x D layout.
"""
import numpy as np
import torch

from . import ops


_side_streams = {}


import contextlib


@contextlib.contextmanager
def quiet_gc():
    """For the loop helpers (train_epoch, train_steps): everything alive at entry is parked in the collector's permanent
    generation for the duration (``gc.freeze``, O(1)), so a cyclic collection that falls into the loop only walks the
    loop's own few objects.  A full collection walks ~10^6 objects once torch is imported -- 37-40 ms measured, during
    which nothing is launched: in a 200-step in-batch run (47 ms of GPU work) ONE such pause cut the throughput from 34
    to 20-30 M pairs/s.  ``ESR_LOOP_GC_FREEZE=0`` leaves the collector alone."""
    import gc
    import os
    if os.environ.get("ESR_LOOP_GC_FREEZE", "1") != "1" or not hasattr(gc, "freeze") or gc.get_freeze_count() > 0:
        yield  # switched off, or somebody (the application, an enclosing loop helper) has frozen the heap already
        return
    gc.freeze()
    try:
        yield
    finally:
        gc.unfreeze()


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class SegmentIndex:
    """Occurrence ids of one gradient scatter, sorted once and shared by every table that is indexed by
    the same ids (GloVe's embedding + bias tables; STL's pos + neg product rows).

    The sort depends on the ids only, not on the gradients, so ``presort()`` launches it on a side stream
    at the top of the step where it overlaps the gather / loss kernels (the device-wide radix sort is a
    chain of ~6 short launches, 25 us of pure latency at these sizes -- profiles/r1); ``sorted()`` makes the
    consuming stream wait on its event."""

    def __init__(self, ids, num_rows, segments=None):
        self._ids = ids  # int32 [n] on device, or None while only `segments` is known
        self.segments = segments  # optional ([id tensors], [offsets]): the list is their virtual concatenation
        self.num_rows = int(num_rows)
        self._sorted = None
        self._event = None
        self.long_runs = -1  # 0 once somebody knows (ops.long_run_hint) that no run outgrows its 32-position head chunk

    @property
    def ids(self):
        if self._ids is None:
            self._ids = ops.concat_offset_ids(list(self.segments[0]), list(self.segments[1]))
        return self._ids

    def presort(self):
        if self._sorted is not None or not self.ids.is_cuda:
            return self
        main = torch.cuda.current_stream(self.ids.device)
        side = _side_stream(self.ids.device)
        side.wait_stream(main)  # the ids are produced on the main stream
        with torch.cuda.stream(side):
            self._sorted = ops.segment_sort(self.ids, self.num_rows)
            self._event = torch.cuda.Event()
            self._event.record(side)
        self.ids.record_stream(side)
        for t in self._sorted:
            t.record_stream(main)
        return self

    def sorted(self):
        if self._sorted is None:
            if self._ids is None and self.segments is not None:  # sort the segments in place: no concatenated copy
                self._sorted = ops.segment_sort_multi(self.segments[0], self.segments[1], self.num_rows)
            else:
                self._sorted = ops.segment_sort(self.ids, self.num_rows)
        elif self._event is not None:
            torch.cuda.current_stream(self.ids.device).wait_event(self._event)
            self._event = None
        return self._sorted


class FusedScatter:
    """The gradients of several same-width tables as ONE occurrence list: ``rows`` is the concatenation of the
    per-segment gradient rows and the ids are virtual rows ``row_offsets[slot] + id`` of the concatenated
    tables, so a step needs one sort chain and one optimizer launch for all its towers.

    id_tensors[i] indexes table ``slots[i]``; ``num_rows[t]`` / ``paths[t]`` describe table t (its row count
    and the path of its parameter leaf)."""

    def __init__(self, id_tensors, slots, num_rows, rows, paths):
        offs = [0]
        for v in num_rows:
            offs.append(offs[-1] + int(v))
        self.row_offsets = offs
        self.paths = [tuple(p) for p in paths]
        self.rows = rows  # f32 [sum n_i, D]; may be filled in later (the ids are known before the gradients)
        self.index = SegmentIndex(None, offs[-1], segments=(list(id_tensors), [offs[s] for s in slots]))
        self.consumed = False

    def consume(self):
        _consume(self, "FusedScatter")


def _consume(obj, what):
    """The segment-reduce kernels park partial sums of long runs of equal ids IN the gradient rows (include/esr_hip.h:
    "may OVERWRITE grad_rows"), so a gradient can feed exactly one to_dense() / optimizer update: a second use would
    silently read clobbered rows for hot ids."""
    if obj.consumed:
        raise RuntimeError("%s was already consumed by an optimizer update / to_dense(): the scatter kernels "
                           "overwrite the gradient rows of hot ids; recompute the gradients to apply them again" % what)
    obj.consumed = True


class RowGrads:
    """Row-sparse gradient of a [V, D] table: ``rows[k]`` is the gradient contribution of occurrence k to
    row ``index.ids[k]``; duplicates accumulate (same meaning as JAX's scatter-add for nn.Embed).
    ``index`` may be given as a list of id tensors (concatenated on first use).  ``fused`` (optional) is the
    FusedScatter this leaf is a member of."""

    def __init__(self, index, rows, shape, fused=None):
        self._index = index
        self.rows = rows      # f32 [n, D]
        self.shape = tuple(shape)
        self.fused = fused
        self.consumed = False

    @property
    def index(self):
        if not isinstance(self._index, SegmentIndex):
            parts = list(self._index)
            ids = parts[0] if len(parts) == 1 else torch.cat(parts)
            self._index = SegmentIndex(ids, self.shape[0])
        return self._index

    def to_dense(self):
        """The dense gradient the reference materialises (wikipedia/train_cooccurence.py:86-87)."""
        _consume(self, "RowGrads")
        sorted_ids, perm = self.index.sorted()
        V = self.shape[0]
        D = self.shape[1] if len(self.shape) > 1 else 1
        return ops.rows_to_dense(V, D, sorted_ids, perm, self.rows.reshape(-1, D)).reshape(self.shape)


def tree_leaves_with_path(tree, prefix=()):
    if isinstance(tree, dict):
        for k in tree:
            yield from tree_leaves_with_path(tree[k], prefix + (k,))
    else:
        yield prefix, tree


def tree_get(tree, path):
    for k in path:
        tree = tree[k]
    return tree


def tree_map(fn, tree):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v) for k, v in tree.items()}
    return fn(tree)


class GradientTransformation:
    """Minimal optax-shaped object: ``init(params) -> opt_state`` and an in-place ``apply``."""
    wants_dense = False

    def init(self, params):
        raise NotImplementedError

    def apply(self, params, grads, opt_state, step):
        raise NotImplementedError


class _Adam(GradientTransformation):
    """optax.adam(learning_rate, b1=0.9, b2=0.999, eps=1e-8) [upstream optax 0.1.2] -- dense, every element
    of every table moves every step (wikipedia/train_cooccurence.py:171)."""
    wants_dense = True

    def __init__(self, learning_rate, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, b1, b2, eps

    def init(self, params):
        return {"count": 0, "mu": tree_map(torch.zeros_like, params), "nu": tree_map(torch.zeros_like, params)}

    # optax.adam's state is (ScaleByAdamState(count, mu, nu), EmptyState()) -> {'0': {...}, '1': {}} on the wire
    def to_optax_state(self, opt_state):
        return {"0": {"count": np.asarray(opt_state["count"], np.int32), "mu": opt_state["mu"], "nu": opt_state["nu"]},
                "1": {}}

    def from_optax_state(self, tree):
        return {"count": int(np.asarray(tree["0"]["count"])), "mu": tree["0"]["mu"], "nu": tree["0"]["nu"]}

    def apply(self, params, grads, opt_state, step):
        count = opt_state["count"] + 1
        for path, p in tree_leaves_with_path(params):
            g = tree_get(grads, path)
            if isinstance(g, RowGrads):
                g = g.to_dense()
            ops.dense_adam(p, tree_get(opt_state["mu"], path), tree_get(opt_state["nu"], path), g.reshape(p.shape),
                           self.lr, count, self.b1, self.b2, self.eps)
        return {"count": count, "mu": opt_state["mu"], "nu": opt_state["nu"]}


class _SparseAdagrad(GradientTransformation):
    """Row-sparse optax.adagrad(learning_rate, initial_accumulator_value=0.1, eps=1e-7) [upstream] -- the
    build's production optimizer (north_star).  Identical to dense Adagrad: rows with zero gradient do
    not move.  Accumulators are fp32 whatever the table dtype."""

    def __init__(self, learning_rate, initial_accumulator_value=0.1, eps=1e-7):
        self.lr, self.init_acc, self.eps = learning_rate, initial_accumulator_value, eps

    def init(self, params):
        return {"sum_of_squares": tree_map(
            lambda p: torch.full(p.shape, self.init_acc, dtype=torch.float32, device=p.device), params)}

    # optax.adagrad's state is (ScaleByRssState(sum_of_squares), EmptyState())
    def to_optax_state(self, opt_state):
        return {"0": {"sum_of_squares": opt_state["sum_of_squares"]}, "1": {}}

    def from_optax_state(self, tree):
        return {"sum_of_squares": tree["0"]["sum_of_squares"]}

    def apply(self, params, grads, opt_state, step):
        done = set()
        for path, p in tree_leaves_with_path(params):
            g = tree_get(grads, path)
            if g is None:
                continue
            if not isinstance(g, RowGrads):
                raise TypeError("sparse_adagrad needs RowGrads leaves (got %s at %s)" % (type(g).__name__, path))
            f = g.fused
            if f is not None and tuple(path) in f.paths:
                if id(f) in done:
                    continue
                done.add(id(f))  # all members of the fused scatter in one sort + one launch
                f.consume()
                sorted_vids, perm = f.index.sorted()
                ops.sparse_adagrad_multi([tree_get(params, q) for q in f.paths],
                                         [tree_get(opt_state["sum_of_squares"], q) for q in f.paths],
                                         f.row_offsets, sorted_vids, perm, f.rows, self.lr, self.eps,
                                         long_runs=f.index.long_runs)
                continue
            _consume(g, "RowGrads")
            sorted_ids, perm = g.index.sorted()
            ops.sparse_adagrad(p, tree_get(opt_state["sum_of_squares"], path), sorted_ids, perm, g.rows, self.lr,
                               self.eps)
        return opt_state


class _SparseSgd(GradientTransformation):
    """Row-sparse optax.sgd(learning_rate) without momentum."""

    def __init__(self, learning_rate):
        self.lr = learning_rate

    def init(self, params):
        return {}

    # optax.sgd(lr) without momentum is chain(identity(), scale(-lr)): (EmptyState(), EmptyState())
    def to_optax_state(self, opt_state):
        return {"0": {}, "1": {}}

    def from_optax_state(self, tree):
        return {}

    def apply(self, params, grads, opt_state, step):
        for path, p in tree_leaves_with_path(params):
            g = tree_get(grads, path)
            if g is None:
                continue
            _consume(g, "RowGrads")
            sorted_ids, perm = g.index.sorted()
            ops.sparse_sgd(p, sorted_ids, perm, g.rows, self.lr)
        return opt_state


class _SgdMomentum(GradientTransformation):
    """optax.sgd(learning_rate, momentum) [upstream] (spotify/train_spotify.py:238-241): trace = g + momentum * trace,
    p -= lr * trace on EVERY element -- rows without a gradient keep coasting on their trace.

    lazy (default): a row that gets no gradient for n steps only undergoes n times ``trace *= momentum ; p -= lr * trace``,
    a function of n alone, so it is applied when the row is next READ instead of by a dense pass over the table every
    step (80 % of the Spotify step's bytes).  ``opt_state['_lazy']`` holds the step counter and, per table, ``last`` (int32
    [V]): the step each row is up to date with.  A step is ``prepare`` (bring the rows it will read up to date: the train
    step calls it before its forward pass), the forward / backward, then ``apply`` (the whole momentum step on the touched
    rows).  ``flush`` brings every row up to date -- ``TrainState.params`` does it, so evals, checkpoints and anybody
    else who reads the tables through the state see exactly what the dense optimizer would have left (bit-identical for
    rows whose gaps stay within 64 steps -- kLazyExact, replayed one by one -- and 1e-7-close beyond, where the closed form
    is used: esr_optim.hip decay_steps).  lazy=False keeps the dense
    decay pass every step."""

    needs_flush = True

    def __init__(self, learning_rate, momentum, lazy=True):
        self.lr, self.momentum, self.lazy = learning_rate, momentum, lazy

    def init(self, params):
        return {"trace": tree_map(torch.zeros_like, params)}

    # optax.sgd's state is (TraceState(trace), EmptyState())
    def to_optax_state(self, opt_state):
        return {"0": {"trace": opt_state["trace"]}, "1": {}}

    def from_optax_state(self, tree):
        return {"trace": tree["0"]["trace"]}

    def _lazy_state(self, params, opt_state):
        lz = opt_state.get("_lazy")
        if lz is None:
            lz = opt_state["_lazy"] = {"step": 0, "prepared": False, "dirty": False, "last": {
                path: torch.zeros(p.shape[0], dtype=torch.int32, device=p.device)
                for path, p in tree_leaves_with_path(params) if p.dim() == 2 and p.is_cuda}}
        return lz

    def prepare(self, params, opt_state, lookups):
        """Start of a lazy step: lookups = [(path, int32 ids, modulus)] -- the rows the step is about to read (row =
        id % modulus when modulus > 0).  They are brought up to the previous step and marked as handled by this one."""
        if not self.lazy:
            return
        lz = self._lazy_state(params, opt_state)
        if not lz["prepared"]:
            lz["step"] += 1
            lz["prepared"] = True
        for path, ids, modulus in lookups:
            path = tuple(path)
            if path in lz["last"]:
                ops.momentum_catchup_rows(tree_get(params, path), tree_get(opt_state["trace"], path), lz["last"][path], ids,
                                          modulus, lz["step"], self.lr, self.momentum)

    def flush(self, params, opt_state):
        lz = opt_state.get("_lazy") if isinstance(opt_state, dict) else None
        if lz is None or not lz["dirty"]:
            return
        for path, last in lz["last"].items():
            ops.momentum_flush(tree_get(params, path), tree_get(opt_state["trace"], path), last, lz["step"], self.lr,
                               self.momentum)
        lz["dirty"] = False

    def apply(self, params, grads, opt_state, step):
        lazy = self.lazy and all(isinstance(tree_get(grads, path), RowGrads) or tree_get(grads, path) is None
                                 for path, p in tree_leaves_with_path(params)) and \
            all(p.dim() == 2 and p.is_cuda for _, p in tree_leaves_with_path(params))
        if self.lazy and not lazy:
            self.flush(params, opt_state)  # a dense step on lazily updated tables: bring them up to date first
        if lazy:
            lz = self._lazy_state(params, opt_state)
            if not lz["prepared"]:  # nobody announced the rows: the caller read them through state.params (flushed)
                self.prepare(params, opt_state, [(path, tree_get(grads, path).index.ids, 0)
                                                 for path, _ in tree_leaves_with_path(params)
                                                 if tree_get(grads, path) is not None])
            for path, p in tree_leaves_with_path(params):
                g = tree_get(grads, path)
                if g is None:
                    continue
                _consume(g, "RowGrads")
                sorted_ids, perm = g.index.sorted()
                ops.sparse_momentum_step(p, tree_get(opt_state["trace"], path), sorted_ids, perm, g.rows, self.lr,
                                         self.momentum)
            lz["prepared"], lz["dirty"] = False, True
            return opt_state
        for path, p in tree_leaves_with_path(params):
            tr = tree_get(opt_state["trace"], path)
            ops.dense_momentum_decay(p, tr, self.lr, self.momentum)
            g = tree_get(grads, path)
            if g is None:
                continue
            if not isinstance(g, RowGrads):
                raise TypeError("sgd(momentum) needs RowGrads leaves (got %s at %s)" % (type(g).__name__, path))
            _consume(g, "RowGrads")
            sorted_ids, perm = g.index.sorted()
            ops.sparse_momentum(p, tr, sorted_ids, perm, g.rows, self.lr)
        if isinstance(opt_state.get("_lazy"), dict):  # (a dense step after lazy ones: everybody is up to date with it)
            opt_state["_lazy"]["step"] += 1
            for last in opt_state["_lazy"]["last"].values():
                last.fill_(opt_state["_lazy"]["step"])
        return opt_state


def adam(learning_rate, b1=0.9, b2=0.999, eps=1e-8):
    return _Adam(learning_rate, b1, b2, eps)


def sparse_adagrad(learning_rate, initial_accumulator_value=0.1, eps=1e-7):
    return _SparseAdagrad(learning_rate, initial_accumulator_value, eps)


def sgd(learning_rate, momentum=None, lazy=True):
    if momentum is not None:
        return _SgdMomentum(learning_rate, momentum, lazy=lazy)
    return _sgd_plain(learning_rate)


def _sgd_plain(learning_rate):
    return _SparseSgd(learning_rate)


class RowVersions:
    """The second buffer of a double-buffered table and the per-row STAMPED bytes that say where each row's current
    value lives (bit 0: 0 = the parameter tensor itself, 1 = `shadow`) and which step last moved it (bits 1..7, see
    esr_versioned.h).  Fused train steps (ops.glove_train_step, ops.triplet_train_step) read rows where they lived
    when the step began, write updated rows into the other buffer and stamp the bytes, so they never need a gradient
    or a snapshot in memory.  Lives on the TrainState (``state.versions[path]``); ``TrainState.params`` consolidates
    before handing the tables to anybody else.  `stamp` = the last stamp a step used on this table."""

    def __init__(self, table):
        self.shadow = torch.empty_like(table)
        self.loc = torch.zeros(table.shape[0], dtype=torch.uint8, device=table.device)
        self.dirty = False
        self.stamp = 0

    def consolidate(self, table):
        if self.dirty:
            ops.rows_consolidate(table, self.shadow, self.loc)  # (zeroes every byte: the stamps start over)
            self.dirty = False
            self.stamp = 0


STAMP_MAX = 127


def next_stamp(*versions, count=1):
    """The stamp of the next fused step on the table(s) of `versions` (several RowVersions = the tables one step
    updates together: they share the stamp).  Stamps run 1 .. 127; when the counter would pass 127 -- or the tables'
    counters disagree -- the bytes' stamps are cleared first (ops.rows_restamp: one pass over V bytes) and counting
    starts over.  count = n reserves n consecutive stamps (a loop that issues n steps) and returns the first."""
    first = versions[0].stamp
    if first + count > STAMP_MAX or any(v.stamp != first for v in versions):
        for v in versions:
            ops.rows_restamp(v.loc)
            v.stamp = 0
        first = 0
    if count > STAMP_MAX:
        raise ValueError("at most %d stamps can be reserved at once" % STAMP_MAX)
    for v in versions:
        v.stamp = first + count
        v.dirty = True
    return first + 1


class ShadowMemoryError(RuntimeError):
    """There is not enough free HBM for the second buffer of a double-buffered table."""


def shadow_fits(table, reserve=1 << 30):
    """True when a second [V, D] buffer for `table` (+ one byte per row) fits in the device's free memory with `reserve`
    bytes to spare.  The one-pass steps DOUBLE the memory of every table they update: on a 288 GB MI355X with fp32 rows and
    fp32 accumulators (3 x V x D x 4 bytes per table with its shadow) that is V <= ~180 M rows of D = 128 per GPU for one
    table, half of it per tower for two -- BASELINE config 4's 12.5 M-row shares fit, a table filling the card does not
    and takes the gradient-row path instead."""
    if not table.is_cuda:
        return False
    need = table.numel() * table.element_size() + table.shape[0]
    free, _ = torch.cuda.mem_get_info(table.device)
    free += torch.cuda.memory_reserved(table.device) - torch.cuda.memory_allocated(table.device)  # torch's own cache
    return need + reserve <= free


def row_versions(state, path, create=True):
    """The RowVersions of the table at `path` of state's parameter tree (created on first use: a second [V, D] buffer;
    None when it does not exist and create is False).  They live on the TrainState (``state.versions``) and follow it
    through ``replace``."""
    path = tuple(path)
    rv = state.versions.get(path)
    if rv is None and create:
        table = tree_get(state.raw_params, path)
        if not shadow_fits(table):
            raise ShadowMemoryError("no room for the second buffer of the table at %s (%d x %d): the one-pass step needs "
                                    "it; use the gradient-row path (ESR_GLOVE_FUSED=0 / ESR_STL_FUSED=0)"
                                    % ("/".join(path), table.shape[0], table.shape[1]))
        rv = state.versions[path] = RowVersions(table)
    return rv


def can_double_buffer(state, paths):
    """True when every table at `paths` already has its second buffer or there is room to give it one."""
    for path in paths:
        if tuple(path) not in state.versions and not shadow_fits(tree_get(state.raw_params, path)):
            return False
    return True


class TrainState:
    """flax.training.train_state.TrainState look-alike: step, apply_fn, params, tx, opt_state.

    Tables are updated IN PLACE by the optimizer; the states returned by ``apply_gradients`` / ``train_step`` share them.
    The one-pass train steps (GloVe ``train_step`` / ``train_epoch``, Shop-The-Look triplet ``train_step`` /
    ``train_steps`` under ``optim.sparse_adagrad``) keep every table they update DOUBLE-BUFFERED: a second [V, D] buffer
    and one byte per row (``versions[path]``, a RowVersions) say where each row's current value lives.  Contract:

    * ``state.params`` is always the plain parameter tree: reading it first copies displaced rows back (one launch over
      the touched rows); ``state.consolidate()`` does the same explicitly.  The loop helpers consolidate when they
      return, so a tensor taken from ``params`` before a loop is current again after it.
    * ``state.raw_params`` is the tree WITHOUT that copy: between one-pass steps a table tensor taken from it (or an alias
      kept from before the steps) may hold stale rows -- only the hot loop uses it.
    * ``versions`` follows the state through ``replace``; replacing ``params`` by OTHER tensors consolidates the old
      ones first and starts the new ones plain."""

    def __init__(self, step, apply_fn, params, tx, opt_state, versions=None):
        self.step = step
        self.apply_fn = apply_fn
        self._params = params
        self.tx = tx
        self.opt_state = opt_state
        self.versions = versions if versions is not None else {}

    @property
    def raw_params(self):
        return self._params

    def consolidate(self):
        """Copy every row that lives in a second buffer back into its table: afterwards the parameter tensors are plain
        [V, D] tables whoever holds them.  No-op when no one-pass step has run since the last call."""
        for path, rv in self.versions.items():
            rv.consolidate(tree_get(self._params, path))
        return self

    @property
    def params(self):
        if self.versions:
            self.consolidate()
        if getattr(self.tx, "needs_flush", False):  # lazily updated tables (sgd with momentum): every row up to date
            self.tx.flush(self._params, self.opt_state)
        return self._params

    @classmethod
    def create(cls, *, apply_fn, params, tx, **kwargs):
        return cls(step=0, apply_fn=apply_fn, params=params, tx=tx, opt_state=tx.init(params))

    def replace(self, **kw):
        versions = self.versions
        if "params" in kw and kw["params"] is not self._params:
            same = all(a is b for (_, a), (_, b) in zip(tree_leaves_with_path(self._params),
                                                        tree_leaves_with_path(kw["params"]))) \
                if _same_tree(self._params, kw["params"]) else False
            if not same:  # other tensors: the old ones become plain, the new ones start without second buffers
                self.consolidate()
                versions = {}
        d = dict(step=self.step, apply_fn=self.apply_fn, params=self._params, tx=self.tx, opt_state=self.opt_state,
                 versions=versions)
        d.update(kw)
        return TrainState(**d)

    def apply_gradients(self, *, grads, **kwargs):
        """step += 1 and one optimizer update (flax TrainState.apply_gradients).  The tables are updated in
        place in HBM; the returned state shares them."""
        if self.versions:
            self.consolidate()
        # (raw parameters: a lazily updated optimizer keeps its own books on which rows are current)
        new_opt = self.tx.apply(self._params, grads, self.opt_state, self.step + 1)
        return self.replace(step=self.step + 1, opt_state=new_opt)


def _same_tree(a, b):
    if isinstance(a, dict) != isinstance(b, dict):
        return False
    if isinstance(a, dict):
        return list(a.keys()) == list(b.keys()) and all(_same_tree(a[k], b[k]) for k in a)
    return True
