"""Row-sharded embedding tables across the GPUs of one node (build-defined -- the
reference is single-device and has no collective anywhere).

Partitioning: ``owner = id mod G``, ``local_row = id div G``; every rank holds V/G rows of each table and
of its optimizer accumulator.  The batch is data-parallel (each rank draws its own pairs).

Tables of the same width that are used by the same step (the two towers) form a ``ShardedTableGroup`` and
travel together: their ids become *virtual ids* ``voff[t] + id`` with every ``voff[t]`` a multiple of G, so
``owner = vid mod G`` is still the table-local owner and ``vid div G`` is a virtual local row that the
owner maps back to (table, local row).  One step of a group is then four collectives over RCCL all-to-all
(xGMI is a full mesh: every peer slice rides its own link) instead of three per table:

    counts -> peers      int64 [G, L]                      \\  the routing PLAN: depends on the ids only, made ahead --
    vids   -> owners     int32 virtual local rows          /   begin_plans (L lookups = L coming batches) / finish
    rows   <- owners     [n, D]  (multi-table gather kernel on the owner; the rows stay in exchange order and
                                  the loss kernels index them through the inverse routing permutation)
    grads  -> owners     [n, D]  (written by the loss kernels directly in exchange order; then ONE fused
                                  segment-reduce + Adagrad on the owner, its sort done in the plan phase)

The plan phase holds the step's single host read-back (all-to-all-v needs host-side split sizes): begin_plans
enqueues the bucket kernel, the counts exchange and an asynchronous copy to pinned memory; finish() -- called one
step later (or, with the plans of a whole group of coming batches made by one begin_plans call, a group of steps
later: bench_sharded.py), when the next kernels are already queued -- waits for that copy only, then exchanges the ids
and sorts them on the owner.  The exchange itself is RCCL send / recv on the compute stream (esrecsys_amd/rccl.py),
falling back to torch.distributed.all_to_all_single.

``torch.distributed`` is plumbing (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests); the kernels
are libesr_hip.so.  ``kernels`` is the module that provides them -- always ``esrecsys_amd.ops`` in the
product; the CPU test-suite injects an oracle-backed double to exercise the routing logic without a GPU.
"""
import os

import torch
import torch.distributed as dist


def _a2a(group, out, inp, out_splits=None, in_splits=None):
    """all_to_all_single of a ShardedTableGroup: RCCL send / recv on the current stream when available
    (esrecsys_amd/rccl.py: no stream hand-overs), else torch.distributed."""
    x = group.exchange()
    if x is not None:
        return x.all_to_all_single(out, inp, out_splits, in_splits)
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group.pg)


class RowShardedTable:
    """This rank's shard of one table: rows rank, rank + G, rank + 2G, ... and their fp32 accumulator."""

    def __init__(self, local_table, local_accum, num_rows):
        self.local = local_table      # [ceil((V - rank) / G), D]
        self.accum = local_accum      # fp32, same shape
        self.num_rows = int(num_rows)

    @staticmethod
    def local_rows_for(num_rows, world, rank):
        return (num_rows - rank + world - 1) // world


def shard_of(table, world, rank):
    """The rows of a full table that `rank` owns, in local-row order (rows rank, rank+G, ...)."""
    return table[rank::world].contiguous()


class RoutingPlan:
    """Where the ids of one group lookup go: everything that does not depend on table contents."""

    def __init__(self, group, n, local_rows, perm, send_counts, recv_counts, inv=None, unique=None):
        self.group = group
        self.n = n                          # occurrences (looked-up ids) of the batch
        self.send_counts = send_counts      # python ints per peer: rows this rank asks of that peer
        self.recv_counts = recv_counts      # python ints per peer: rows that peer asks of this rank
        self.n_rows = sum(send_counts)      # rows that cross the exchange: n, or the DISTINCT rows of a unique plan
        self.local_rows = local_rows[:self.n_rows]  # int32: vid // G of the rows asked for, owner-major
        self.perm = perm                    # int32 [n]: bucket order -> original position (occurrence plans)
        self.inv = inv                      # int32 [n]: original position -> bucket order (inv[perm[k]] = k)
        # unique plans (ShardedTableGroup.unique): every distinct row is asked for ONCE; occurrence i reads row uidx[i] of
        # what comes back, and (sorted_uidx, occ_perm) group the occurrences by distinct row for the segment sum of their
        # gradient rows -- one summed row per distinct row goes back to the owner
        self.unique = unique is not None
        self.uidx, self.sorted_uidx, self.occ_perm = unique if unique is not None else (None, None, None)
        self.recv_local_rows = None         # int32 [sum(recv_counts)]: virtual local rows requested of this rank
        self.owner_sorted = None            # (sorted, permutation) of recv_local_rows, for the update
        self._c_counts = None               # (ask, asked) host int64 arrays for the one-call exchange halves
        self._c_struct = None

    def c_counts(self, k):
        if self._c_counts is None:
            self._c_counts = (k.i64_array(self.send_counts), k.i64_array(self.recv_counts))
        return self._c_counts

    def c_struct(self, k):
        """esr_routing_plan_t of this plan (ids exchanged and sorted on the owner) for the one-call steps."""
        if self._c_struct is None:
            ask_c, asked_c = self.c_counts(k)
            recv = self.exchange_ids()
            srt, prm = self.owner_sorted if self.owner_sorted is not None else (None, None)
            self._c_struct = k.routing_plan_struct(recv, asked_c, ask_c, self.index, self.sorted_uidx if self.unique else None,
                                                   self.occ_perm if self.unique else None, srt, prm)
        return self._c_struct

    @property
    def index(self):
        """int32 [n]: the row of the looked-up block (lookup_bucketed) that occurrence i reads."""
        return self.uidx if self.unique else self.inv

    def exchange_ids(self):
        if self.recv_local_rows is None:
            g = self.group
            self.recv_local_rows = torch.empty(sum(self.recv_counts), dtype=torch.int32, device=self.local_rows.device)
            _a2a(g, self.recv_local_rows, self.local_rows, self.recv_counts, self.send_counts)
            if self.recv_local_rows.numel():  # the owner-side sort needs the ids only: do it ahead of the step
                self.owner_sorted = g.k.segment_sort(self.recv_local_rows, g.loff[-1])
        return self.recv_local_rows


_pinned_ring = {}
# ESR_TRACE_HOST=1: [seconds the host spent waiting for the counts copies, number of waits] -- is a sharded loop bound by
# the host (no waiting: it cannot keep up) or by the GPU (it waits every step)?
_trace = [0.0, 0] if os.environ.get("ESR_TRACE_HOST") == "1" else None


def _pinned_like(t, depth=8):
    """A pinned host buffer of t's shape from a small ring (pinning memory per step costs ~100 us)."""
    key = (tuple(t.shape), t.dtype)
    ring = _pinned_ring.setdefault(key, [[], 0])
    if len(ring[0]) < depth:
        ring[0].append(torch.empty(t.shape, dtype=t.dtype, pin_memory=True))
        return ring[0][-1]
    ring[1] = (ring[1] + 1) % depth
    return ring[0][ring[1]]


class PendingPlans:
    """The device half of make_plans, already enqueued: bucket kernels, the counts all-to-all and an asynchronous
    copy of the counts to pinned host memory.  ``finish()`` waits for that copy only -- a training loop calls it
    one step later, after the NEXT step's kernels are in the queue, so the host never stalls an idle GPU."""

    def __init__(self, parts, both_dev, both_host, event):
        self.parts, self.both_dev, self.both_host, self.event = parts, both_dev, both_host, event
        self.plans = None

    def _exchange_ids_grouped(self):
        return _grouped_owner_sort(self.plans)

    def finish(self):
        if self.plans is None:
            if self.event is not None:
                if _trace is not None:
                    import time
                    t0 = time.perf_counter()
                    self.event.synchronize()
                    _trace[0] += time.perf_counter() - t0
                    _trace[1] += 1
                else:
                    self.event.synchronize()
            both = self.both_host
            self.plans = [RoutingPlan(group, n, local_rows, perm, both[0, :, i].tolist(), both[1, :, i].tolist(), inv,
                                      unique=part[6] if len(part) > 6 else None)
                          for i, part in enumerate(self.parts) for (group, n, local_rows, perm, _, inv) in [part[:6]]]
            for p in self.plans:
                if p.unique:
                    p.group.observe_unique(p.n_rows, p.n)
            if not self._exchange_ids_grouped():
                for p in self.plans:
                    p.exchange_ids()
            self.parts = self.both_dev = None
        return self.plans


_UNIQUE_KEEP_BELOW = 0.8     # auto mode keeps asking for distinct rows while distinct / occurrences is below this
_UNIQUE_PROBE_EVERY = 32     # ... and otherwise measures again after this many plan groups
_OWNER_SORT_BATCH_MAX = 32768  # ids per list the batched owner-side sort takes (esr_segment_sort_ids_batched)


def _grouped_owner_sort(plans):
    """The ids exchanges of a group of plans of ONE table group, and their owner-side sorts as ONE batched sort
    (esr_segment_sort_ids_batched): the lists a rank is asked for differ in length from batch to batch, so they are
    received into rows of one [L, n_max] buffer pre-filled with a sentinel row id that sorts behind every real one --
    list b's sorted ids / permutation are the first n_b entries of row b.  Returns False when the kernels or the groups
    do not allow it (the caller then goes plan by plan).

    HOW the ids are exchanged (one RCCL group for the L lists, or L exchanges) is decided from what every rank knows
    alike -- L, the device, the exchange object -- never from this rank's receive sizes: with skewed ids one rank may be
    asked for more than the batched sort takes while its peers are not, and ranks that issued the same collectives with
    different group structures would hang.  Only the owner-side SORT falls back per rank."""
    g = plans[0].group
    k = g.k
    L = len(plans)
    dev = plans[0].local_rows.device
    if L < 2 or L > 8 or not hasattr(k, "segment_sort_batched") or any(p.group is not g for p in plans):
        return False
    ns = [sum(p.recv_counts) for p in plans]
    n_max, sentinel = max(ns), g.loff[-1]
    # rank-local: how THIS rank sorts what it received
    batched_sort = 0 < n_max <= _OWNER_SORT_BATCH_MAX and sentinel + 1 <= (1 << 21)
    if batched_sort:
        buf = torch.full((L, n_max), sentinel, dtype=torch.int32, device=dev)
        for i, p in enumerate(plans):
            p.recv_local_rows = buf[i, :ns[i]]
    else:
        for i, p in enumerate(plans):
            p.recv_local_rows = torch.empty(ns[i], dtype=torch.int32, device=dev)
    x = g.exchange()
    if x is not None and hasattr(x, "all_to_all_multi"):  # the L ids exchanges as one RCCL group (one kernel)
        x.all_to_all_multi([(p.recv_local_rows, p.local_rows, p.recv_counts, p.send_counts) for p in plans])
    else:
        for p in plans:
            _a2a(g, p.recv_local_rows, p.local_rows, p.recv_counts, p.send_counts)
    if batched_sort:
        srt, prm = k.segment_sort_batched([[buf[i]] for i in range(L)], (0,), sentinel + 1)
        for i, p in enumerate(plans):
            p.owner_sorted = (srt[i, :ns[i]], prm[i, :ns[i]]) if ns[i] else None
    else:
        for p in plans:
            p.owner_sorted = k.segment_sort(p.recv_local_rows, g.loff[-1]) if p.recv_local_rows.numel() else None
    return True


def _bucket_grouped(lookups, both):
    """The bucket kernels of a group of lookups as ONE launch pair (esr_bucket_ids_by_owner_batched) when they are the
    (id tensors, offsets) kind with identical shapes and the kernels provide it; else None.  Fills both[0] ([G, L])."""
    g0 = lookups[0][0]
    k, G, L = g0.k, g0.world, len(lookups)
    if L < 2 or L > 8 or not hasattr(k, "bucket_ids_by_owner_batched") or not both.is_cuda or \
            any(g.unique for g, _ in lookups):
        return None
    vids = [v if isinstance(v, tuple) else ([v.reshape(-1)], [0]) for _, v in lookups]  # a plain id tensor: one segment
    offs = list(vids[0][1])
    shapes = [int(t.numel()) for t in vids[0][0]]
    if any(list(v[1]) != offs or [int(t.numel()) for t in v[0]] != shapes for v in vids):
        return None
    local_rows, perm, counts, inv = k.bucket_ids_by_owner_batched([list(v[0]) for v in vids], G, offs)
    both[0].copy_(counts.t())
    n = sum(shapes)
    return [(group, n, local_rows[i], perm[i], counts[i], inv[i]) for i, (group, _) in enumerate(lookups)]


def begin_plans(lookups):
    """lookups: list of (ShardedTableGroup, virtual ids) -- an int32 [n] tensor or the (id tensors, offsets) pair of
    ShardedTableGroup.virtual_id_segments.  Enqueues everything of the routing plans that needs no host knowledge and
    returns a PendingPlans."""
    g0 = lookups[0][0]
    k, G, L = g0.k, g0.world, len(lookups)
    dev = g0.tables[0].local.device
    for g in {id(g): g for g, _ in lookups}.values():
        g.planning()
    # [send | recv][peer][lookup]: all_to_all_single hands every peer its L counts; with one lookup (every step of this
    # package) the bucket kernel writes its counts straight into the send half -- no stack / cat / copy launches
    both = torch.empty((2, G, L), dtype=torch.int64, device=dev)
    grouped = _bucket_grouped(lookups, both)  # one launch pair for all the lookups (fills both[0]), or None
    parts = grouped if grouped is not None else []
    for group, vids in (lookups if grouped is None else []):
        out = both[0, :, 0] if L == 1 else None
        if group.unique:  # every distinct row asked for once (esr_unique_by_owner)
            segs, offs = (list(vids[0]), list(vids[1])) if isinstance(vids, tuple) else ([vids.reshape(-1)], [0])
            n = sum(int(t.numel()) for t in segs)
            ulocal, ucounts, uidx, sorted_uidx, occ_perm = k.unique_by_owner(segs, G, group.loff[-1], offsets=offs)
            if out is not None:
                out.copy_(ucounts)
            parts.append((group, n, ulocal, None, ucounts, None, (uidx, sorted_uidx, occ_perm)))
            continue
        if isinstance(vids, tuple):   # (id tensors, virtual offsets): bucketed in place, never concatenated
            n = sum(int(t.numel()) for t in vids[0])
            local_rows, perm, counts, inv = k.bucket_ids_by_owner(list(vids[0]), G, want_inverse=True, offsets=vids[1],
                                                                  counts_out=out)
        else:
            n = vids.numel()
            local_rows, perm, counts, inv = k.bucket_ids_by_owner(vids, G, want_inverse=True, counts_out=out)
        parts.append((group, n, local_rows, perm, counts, inv))
    if L > 1 and grouped is None:
        both[0].copy_(torch.stack([p[4] for p in parts], dim=1))
    _a2a(g0, both[1], both[0])
    if both.is_cuda:
        host = _pinned_like(both)
        host.copy_(both, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return PendingPlans(parts, both, host, ev)
    return PendingPlans(parts, both, both, None)


def make_plans(lookups):
    """lookups: list of (ShardedTableGroup, virtual ids int32 [n]).  One counts all-to-all and one host sync for the
    whole list.  Returns one RoutingPlan per lookup (ids exchanged, owner-side sort done).  A training loop should
    call begin_plans / finish one step apart instead (see bench_sharded.py): the host wait then never stalls the GPU."""
    if not lookups:
        return []
    return begin_plans(lookups).finish()


class ShardedTableGroup:
    """Same-width, same-dtype row-sharded tables that one step looks up and updates together."""

    def __init__(self, tables, group=None, kernels=None, unique=None, grad_dtype=None):
        if kernels is None:
            from . import ops as kernels
        self.k = kernels
        self.tables = list(tables)
        self.pg = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        G = self.world
        self.voff = [0]                     # global virtual offsets, multiples of G
        for t in self.tables:
            self.voff.append(self.voff[-1] + G * ((t.num_rows + G - 1) // G))
        self.loff = [o // G for o in self.voff]  # the same boundaries in virtual LOCAL rows
        self.dim = self.tables[0].local.shape[1] if self.tables[0].local.dim() > 1 else 1
        self._xch = False  # DirectExchange, None (use torch.distributed) or False (not resolved yet)
        # unique: every distinct row of a batch crosses the exchange once (rows out, ONE summed gradient row back) instead
        # of once per occurrence.  A sender-side choice -- owners serve whatever list they get -- that pays where bytes
        # cross xGMI (world > 1) and ids repeat (GloVe's Zipfian stream: wikipedia/make_cooccurrence.py:33-55); at world
        # 1 it only adds the dedup and segment-sum launches.  ESR_SHARDED_UNIQUE=0 / 1 overrides.
        # Default at world > 1: AUTO, a sender-side choice made per rank from the duplicate rate its own plans measure --
        # the first plans are unique ones (they report distinct rows / occurrences); while that ratio stays above
        # _UNIQUE_KEEP_BELOW (uniform ids at C2: 0.99 -- the dedup launches cost more than the rows they save) the rank
        # asks per occurrence and probes again every _UNIQUE_PROBE_EVERY plan groups.  Owners never need to know.
        env = os.environ.get("ESR_SHARDED_UNIQUE", "")
        can = hasattr(kernels, "unique_by_owner")
        if env in ("0", "1"):
            self.unique_mode = "on" if env == "1" and can else "off"
        elif unique is not None:
            self.unique_mode = "on" if unique and can else "off"
        else:
            self.unique_mode = "auto" if self.world > 1 and can else "off"
        self._auto_unique, self._auto_skip = True, 0
        # A world of ONE rank has nothing to exchange: its steps are the single-GPU steps on the local shard (= the whole
        # table) -- the one-pass triplet / GloVe steps on double-buffered tables, the in-batch head straight from the
        # towers -- and no routing plan is made.  ESR_SHARDED_WORLD1_DIRECT=0 keeps the whole exchange machinery running
        # at world 1 (bucket, self-"exchange", gather, gradient rows, owner-side update): what the world-1 tests and
        # profiles of that machinery use.
        self.world1_direct = self.world == 1 and os.environ.get("ESR_SHARDED_WORLD1_DIRECT", "1") == "1" and \
            hasattr(kernels, "triplet_train_step")
        self._versions = None
        # gradient rows cross the exchange as f32, or as bf16 (rounded after the per-distinct-row sum, widened on the owner:
        # element error <= 2^-9 relative; SURVEY 8d budgets bf16-sized gradients for config 4).  ESR_SHARDED_GRAD_DTYPE=bf16
        self.grad_dtype = grad_dtype if grad_dtype is not None else os.environ.get("ESR_SHARDED_GRAD_DTYPE", "f32")
        if self.grad_dtype not in ("f32", "bf16"):
            raise ValueError("grad_dtype must be 'f32' or 'bf16', got %r" % (self.grad_dtype,))
        self._c_tables = None
        self._c_group_cache = None

    @property
    def unique(self):
        """Does the NEXT plan of this group ask for every distinct row once?"""
        return self.unique_mode == "on" or (self.unique_mode == "auto" and self._auto_unique)

    def planning(self):
        """begin_plans is about to make a group of plans for this group (auto mode: count down to the next probe)."""
        if self.unique_mode == "auto" and not self._auto_unique:
            self._auto_skip -= 1
            if self._auto_skip <= 0:
                self._auto_unique = True  # probe: the next plans are unique ones and report their duplicate rate

    def observe_unique(self, n_rows, n_occ):
        """A unique plan of this group came back with n_rows distinct rows for n_occ occurrences."""
        if self.unique_mode == "auto" and n_occ > 0 and n_rows >= _UNIQUE_KEEP_BELOW * n_occ:
            self._auto_unique, self._auto_skip = False, _UNIQUE_PROBE_EVERY

    def _fused(self):
        """(comm, tables_c, accums_c, loff_c, dtype code) when the exchange halves of a step run as ONE library call each
        (esr_sharded_lookup / esr_sharded_update): CUDA shards and either a world of one rank or the direct RCCL exchange.
        None: op by op through `kernels` and torch.distributed (the CPU doubles of the gloo tests, ESR_SHARDED_FUSED=0)."""
        k = self.k
        if not hasattr(k, "sharded_lookup") or os.environ.get("ESR_SHARDED_FUSED", "1") != "1" or \
                not all(t.local.is_cuda for t in self.tables):
            return None
        comm = None
        if self.world > 1:
            x = self.exchange()
            if x is None or not getattr(x, "comm", None):
                return None
            comm = x.comm
        key = tuple(t.local.data_ptr() for t in self.tables) + tuple(t.accum.data_ptr() for t in self.tables)
        if self._c_tables is None or self._c_tables[0] != key:
            dts = {t.local.dtype for t in self.tables}
            if len(dts) != 1 or not all(t.local.is_contiguous() and t.accum.is_contiguous() for t in self.tables):
                return None
            if len(self.tables) > 1 and (self.dim * self.tables[0].local.element_size()) % 16:
                return None  # (the fused multi-table kernels move whole 16-byte chunks)
            code = k.ESR_BF16 if dts.pop() == torch.bfloat16 else k.ESR_F32
            self._c_tables = (key, k.ptr_array([t.local for t in self.tables]), k.ptr_array([t.accum for t in self.tables]),
                              k.i64_array(self.loff), code)
        _, tc, ac, lc, code = self._c_tables
        return comm, tc, ac, lc, code

    def _c_group(self):
        """esr_shard_group_t of this group for the one-call steps, or None (see _fused)."""
        fused = self._fused()
        if fused is None or not hasattr(self.k, "shard_group_struct"):
            return None
        comm, tc, ac, lc, code = fused
        key = (self._c_tables[0], comm.value if hasattr(comm, "value") else comm, self.grad_dtype)
        if self._c_group_cache is None or self._c_group_cache[0] != key:
            k = self.k
            gd = k.ESR_BF16 if self.grad_dtype == "bf16" else k.ESR_F32
            self._c_group_cache = (key, k.shard_group_struct(comm, self.world, tc, ac, lc, len(self.tables), code,
                                                             self.dim, gd))
        return self._c_group_cache[1]

    def versions(self):
        """RowVersions (second buffer + stamped location bytes, train_state.py) of every table of the group: what the
        one-pass steps of the world-1 direct path keep.  None when a second buffer does not fit."""
        if self._versions is None:
            from .train_state import RowVersions, shadow_fits
            if not all(shadow_fits(t.local) for t in self.tables):
                return None
            self._versions = [RowVersions(t.local) for t in self.tables]
        return self._versions

    def consolidate(self):
        """After one-pass steps (world-1 direct path): copy the rows that live in second buffers back, so that
        ``table.local`` is the plain shard again.  No-op otherwise."""
        if self._versions is not None:
            for t, rv in zip(self.tables, self._versions):
                rv.consolidate(t.local)

    def exchange(self):
        if self._xch is False:
            dev = self.tables[0].local.device
            if dev.type == "cuda":
                from . import rccl
                self._xch = rccl.exchange_for(self.pg, dev)
            else:
                self._xch = None
        return self._xch

    def virtual_ids(self, id_tensors, slots):
        """[ids_i + voff[slots[i]]] concatenated: id_tensors[i] indexes table slots[i]."""
        if len(self.tables) == 1 and len(id_tensors) == 1:
            return id_tensors[0]
        return self.k.concat_offset_ids(list(id_tensors), [self.voff[s] for s in slots])

    def virtual_id_segments(self, id_tensors, slots):
        """The same virtual list as (id tensors, offsets) for begin_plans / make_plans: bucketed without the
        concatenated copy."""
        if os.environ.get("ESR_CHECK_IDS") == "1" and hasattr(self.k, "check_device_ids"):
            for t, s_ in zip(id_tensors, slots):   # debug screen: an out-of-range id would be routed to a wild row
                if t.is_cuda:
                    self.k.check_device_ids(t, self.tables[s_].num_rows)
        if len(self.tables) == 1 and len(id_tensors) == 1:
            return id_tensors[0]
        return (list(id_tensors), [self.voff[s] for s in slots])

    def plan(self, vids):
        return make_plans([(self, vids)])[0]

    def lookup_bucketed(self, plan):
        """The looked-up rows in BUCKET order (row plan.inv[i] belongs to virtual id i), in the tables' dtype --
        for consumers that can index them themselves and so skip the un-permute pass."""
        k = self.k
        self.consolidate()  # rows a world-1 one-pass step left in the second buffers (no-op when nothing is displaced)
        recv = plan.exchange_ids()
        fused = self._fused()
        if fused is not None:  # gather + rows exchange as one library call
            comm, tc, _, lc, code = fused
            ask_c, asked_c = plan.c_counts(k)
            dt, dev = self.tables[0].local.dtype, recv.device
            back = torch.empty((plan.n_rows, self.dim), dtype=dt, device=dev)
            served = torch.empty((recv.numel(), self.dim), dtype=dt, device=dev) if self.world > 1 else None
            return k.sharded_lookup(comm, self.world, tc, lc, len(self.tables), code, self.dim, recv, asked_c, ask_c,
                                    served, back)
        if len(self.tables) == 1:
            served = k.gather_rows(self.tables[0].local, recv)
        else:
            served = k.gather_rows_multi([t.local for t in self.tables], self.loff, recv)
        back = torch.empty((plan.n_rows, self.dim), dtype=served.dtype, device=served.device)
        _a2a(self, back, served, plan.send_counts, plan.recv_counts)
        return back

    def lookup(self, plan):
        """rows[i] = table_of(vid_i)[id_i] for this rank's virtual ids -> [n, D] in the order of the ids."""
        back = self.lookup_bucketed(plan)
        if plan.unique:  # one row per distinct id came back: occurrence i reads row uidx[i]
            return self.k.unpermute_rows_to_f32(self.k.gather_rows(back, plan.uidx), None)
        # bucket order -> id order; bf16 tables (config 4) cross xGMI as bf16 and become f32 here
        return self.k.unpermute_rows_to_f32(back, plan.perm)

    def route_grads(self, plan, grad_rows, bucketed=False):
        """Per-occurrence gradient rows (order of the looked-up ids, or already in bucket order) -> rows on their
        owners."""
        k = self.k
        if not bucketed and plan.unique:  # one summed row per distinct row (the order the rows were asked for in)
            grad_rows = k.segment_sum_rows(plan.n_rows, plan.sorted_uidx, plan.occ_perm, grad_rows)
        elif not bucketed:
            grad_rows = k.gather_rows(grad_rows, plan.perm)                     # id order -> bucket order
        if self.grad_dtype == "bf16" and self.world > 1:  # rounded after the per-row sum, widened on the owner
            half = grad_rows.to(torch.bfloat16).contiguous().view(torch.uint8)  # (bytes: every backend moves them)
            raw = torch.empty((sum(plan.recv_counts), 2 * grad_rows.shape[1]), dtype=torch.uint8, device=grad_rows.device)
            _a2a(self, raw, half, plan.recv_counts, plan.send_counts)
            return raw.view(torch.bfloat16).to(grad_rows.dtype)
        recv = torch.empty((sum(plan.recv_counts), grad_rows.shape[1]), dtype=grad_rows.dtype,
                           device=grad_rows.device)
        _a2a(self, recv, grad_rows, plan.recv_counts, plan.send_counts)
        return recv

    def apply_sparse_adagrad(self, plan, grad_rows, lr, eps=1e-7, bucketed=False):
        """Route the gradients to their owners and update the local shards: one fused segment-reduce + RMW."""
        k = self.k
        self.consolidate()  # this update writes the plain shards: displaced rows must be home first (see lookup_bucketed)
        fused = self._fused() if (bucketed or plan.unique) and grad_rows.dtype == torch.float32 and \
            grad_rows.is_contiguous() else None
        if fused is not None:  # [segment sum ->] gradient exchange -> owner-side update as one library call
            comm, tc, ac, lc, code = fused
            ask_c, asked_c = plan.c_counts(k)
            plan.exchange_ids()
            n_recv, dev = sum(plan.recv_counts), grad_rows.device
            uniq = plan.unique and not bucketed
            summed = torch.empty((plan.n_rows, self.dim), dtype=torch.float32, device=dev) if uniq else None
            remote = self.world > 1
            bf16 = remote and self.grad_dtype == "bf16" and self.dim % 8 == 0  # (the bias column stays f32)
            recv = torch.empty((n_recv, self.dim), dtype=torch.float32, device=dev) if remote else None
            send_h = torch.empty((plan.n_rows, self.dim), dtype=torch.bfloat16, device=dev) if bf16 else None
            recv_h = torch.empty((n_recv, self.dim), dtype=torch.bfloat16, device=dev) if bf16 else None
            srt, prm = plan.owner_sorted if plan.owner_sorted is not None else (None, None)
            k.sharded_update(comm, self.world, tc, ac, lc, len(self.tables), code, self.dim, grad_rows,
                             plan.sorted_uidx if uniq else None, plan.occ_perm if uniq else None, summed, ask_c, asked_c,
                             k.ESR_BF16 if bf16 else k.ESR_F32, send_h, recv_h, recv, srt, prm, lr, eps)
            return
        rows = self.route_grads(plan, grad_rows, bucketed=bucketed)
        if rows.shape[0] == 0:
            return
        sorted_rows, perm = plan.owner_sorted
        if len(self.tables) == 1:
            t = self.tables[0]
            k.sparse_adagrad(t.local, t.accum, sorted_rows, perm, rows, lr, eps)
        else:
            k.sparse_adagrad_multi([t.local for t in self.tables], [t.accum for t in self.tables], self.loff,
                                   sorted_rows, perm, rows, lr, eps)


def plan_inbatch(towers, scene_ids, pos_ids):
    return towers.plan(towers.virtual_id_segments([scene_ids, pos_ids], [0, 1]))


def plan_triplet(towers, scene_ids, pos_ids, neg_ids):
    return towers.plan(towers.virtual_id_segments([scene_ids, pos_ids, neg_ids], [0, 1, 1]))


def plan_glove(emb_group, inputs):
    """The embedding and bias tables are indexed by the same ids and sharded the same way: one routing."""
    return emb_group.plan(emb_group.virtual_id_segments([inputs.reshape(-1)], [0]))


class _Pending1:
    def __init__(self, pending):
        self.pending = pending

    def finish(self):
        return self.pending.finish()[0]


def begin_plan_inbatch(towers, scene_ids, pos_ids):
    """Non-blocking half of plan_inbatch; ``.finish()`` returns the RoutingPlan."""
    return _Pending1(begin_plans([(towers, towers.virtual_id_segments([scene_ids, pos_ids], [0, 1]))]))


def begin_plan_triplet(towers, scene_ids, pos_ids, neg_ids):
    return _Pending1(begin_plans([(towers, towers.virtual_id_segments([scene_ids, pos_ids, neg_ids], [0, 1, 1]))]))


def begin_plan_glove(emb_group, inputs):
    return _Pending1(begin_plans([(emb_group, emb_group.virtual_id_segments([inputs.reshape(-1)], [0]))]))


def _is_f32(group):
    return all(t.local.dtype == torch.float32 for t in group.tables)


def _joined(first, *rest):
    """The single buffer the gradient slices are views of, else their concatenation."""
    base = getattr(first, "_base", None)
    n = first.shape[0] + sum(r.shape[0] for r in rest)
    if base is not None and base.shape[0] == n:
        return base
    return torch.cat((first,) + rest)


def _world1_tables_ok(group):
    return group.world1_direct and all(t.local.is_cuda for t in group.tables)


def sharded_inbatch_step(towers, scene_ids, pos_ids, regularization, global_batch_size, scale, lr, plan=None):
    """Data-parallel in-batch-softmax step on row-sharded towers (group = [scene table, product table]).
    Negatives are the local batch; gradients are normalised by the GLOBAL batch size, so the sum of the
    per-rank losses is the global mean loss."""
    k = towers.k
    B = scene_ids.numel()
    if _world1_tables_ok(towers) and towers.tables[0].local.shape[1] <= 128 and B % 128 == 0 and \
            towers.tables[0].local.dtype == towers.tables[1].local.dtype:
        # world 1: the single-GPU step (score head straight from the towers, one sort, one fused update)
        towers.consolidate()  # (rows a one-pass triplet step left in the second buffers: this step reads the plain tables)
        st, pt = towers.tables
        loss, _, gq, gc = k.inbatch_towers_fwd_bwd(st.local, pt.local, scene_ids, pos_ids, scale, regularization,
                                                   global_batch_size)
        Vs = st.local.shape[0]
        sorted_vids, perm = k.segment_sort_multi([scene_ids, pos_ids], [0, Vs], Vs + pt.local.shape[0])
        k.sparse_adagrad_multi([st.local, pt.local], [st.accum, pt.accum], [0, Vs, Vs + pt.local.shape[0]], sorted_vids,
                               perm, _joined(gq, gc), lr, 1e-7)
        return loss
    plan = plan if plan is not None else plan_inbatch(towers, scene_ids, pos_ids)
    folded = getattr(k, "inbatch_towers_fwd_bwd", None)
    if folded is not None and plan.index is not None and (getattr(k, "TOWERS_ANY_SHAPE", False) or
                                                          (towers.dim == 128 and B % 128 == 0)):
        # The score head reads the exchanged rows where they landed (bucket order, table dtype) through the inverse
        # permutation and writes its gradient rows straight into bucket order: no un-permute, no widening pass, no
        # permute before the gradient all-to-all.
        back = towers.lookup_bucketed(plan)
        iq, ic = plan.index[:B], plan.index[B:]
        if plan.unique:  # several occurrences may read one row: per-occurrence gradient rows, summed per distinct row
            loss, _, gq, gc = folded(back, back, iq, ic, scale, regularization, global_batch_size)
            towers.apply_sparse_adagrad(plan, _joined(gq, gc), lr)
            return loss
        loss, _, gbuf, _ = folded(back, back, iq, ic, scale, regularization, global_batch_size,
                                  grad_positions=(iq, ic))
        towers.apply_sparse_adagrad(plan, gbuf, lr, bucketed=True)
        return loss
    rows = towers.lookup(plan)                     # [q ; c]
    loss, _, gq, gc = k.inbatch_softmax_fwd_bwd(rows[:B], rows[B:], scale, regularization, global_batch_size)
    towers.apply_sparse_adagrad(plan, _joined(gq, gc), lr)
    return loss


def sharded_triplet_step(towers, scene_ids, pos_ids, neg_ids, regularization, global_batch_size, lr, plan=None):
    """Reference triplet loss (pinterest/train_shop_the_look.py:93-109) on row-sharded towers.  The loss is a
    sum over triplets, so G ranks x B triplets == one device with G*B triplets and batch_size = G*B."""
    k = towers.k
    B = scene_ids.numel()
    if _world1_tables_ok(towers) and _is_f32(towers) and getattr(k, "triplet_direct_mode", lambda: False)():
        # world 1: the one-pass step on the local shards, rows stepped in place (esr_triplet_train_step, direct mode)
        towers.consolidate()  # (rows an earlier stamped step left in second buffers)
        st, pt = towers.tables
        return k.triplet_train_step(st.local, None, None, st.accum, pt.local, None, None, pt.accum, scene_ids, pos_ids,
                                    neg_ids, regularization, global_batch_size, lr)
    if _world1_tables_ok(towers) and _is_f32(towers) and towers.versions() is not None:
        # world 1: the one-pass step on the local shards (esr_triplet_train_step), nothing to exchange
        from .train_state import next_stamp
        st, pt = towers.tables
        rs, rp = towers.versions()
        return k.triplet_train_step(st.local, rs.shadow, rs.loc, st.accum, pt.local, rp.shadow, rp.loc, pt.accum,
                                    scene_ids, pos_ids, neg_ids, regularization, global_batch_size, lr,
                                    stamp=next_stamp(rs, rp))
    plan = plan if plan is not None else plan_triplet(towers, scene_ids, pos_ids, neg_ids)
    if plan.index is not None and _is_f32(towers) and hasattr(k, "sharded_triplet_step"):
        gs_ = towers._c_group()
        if gs_ is not None:  # lookup -> loss -> update as ONE library call (esr_sharded_triplet_step)
            towers.consolidate()
            return k.sharded_triplet_step(gs_, plan.c_struct(k), B, regularization, global_batch_size, lr, 1e-7,
                                          scene_ids.device)
    if plan.index is not None and getattr(k, "GRADS_AT_IDS", None) is not None and _is_f32(towers):
        # the loss kernel indexes the exchanged rows where they landed (bucket order) through the inverse
        # permutation and writes every gradient row back at that position: no un-permute / permute passes
        back = towers.lookup_bucketed(plan)
        inv = plan.index
        if plan.unique:  # per-occurrence gradient rows in occurrence order, summed per distinct row before they travel
            loss, _, _, gs, gp, gn = k.triplet_fwd_bwd(back, back, back, inv[:B], inv[B:2 * B], inv[2 * B:], B,
                                                       regularization, global_batch_size, with_reg=True,
                                                       want_grads=True, want_scores=False)
            towers.apply_sparse_adagrad(plan, _joined(gs, gp, gn), lr)
            return loss
        loss, _, _, gbuf, _, _ = k.triplet_fwd_bwd(back, back, back, inv[:B], inv[B:2 * B], inv[2 * B:], B,
                                                   regularization, global_batch_size, with_reg=True, want_grads=True,
                                                   want_scores=False, grads_at_ids=True)
        towers.apply_sparse_adagrad(plan, gbuf, lr, bucketed=True)
        return loss
    rows = towers.lookup(plan)                     # [scene ; pos ; neg]
    loss, _, _, gs, gp, gn = k.triplet_fwd_bwd(rows[:B], rows[B:2 * B], rows[2 * B:], None, None, None, B,
                                               regularization, global_batch_size, with_reg=True, want_grads=True,
                                               want_scores=False)
    towers.apply_sparse_adagrad(plan, _joined(gs, gp, gn), lr)
    return loss


def sharded_glove_step(emb_group, bias_group, inputs, target, mode, lr, plan=None):
    """GloVe step on row-sharded embedding + bias tables (two single-table groups sharing one routing plan:
    same ids, same sharding, different widths); the loss is over the local batch."""
    k = emb_group.k
    B = inputs.shape[1]
    if _world1_tables_ok(emb_group) and _is_f32(emb_group) and _is_f32(bias_group) and emb_group.versions() is not None:
        # world 1: the one-pass step on the local shard (esr_glove_train_step), nothing to exchange
        from .train_state import next_stamp
        et, bt = emb_group.tables[0], bias_group.tables[0]
        (rv,) = emb_group.versions()
        return k.glove_train_step(et.local, rv.shadow, rv.loc, et.accum, bt.local, bt.accum, inputs, target, mode, lr,
                                  stamp=next_stamp(rv))
    plan = plan if plan is not None else plan_glove(emb_group, inputs)
    if plan.index is not None and _is_f32(emb_group) and _is_f32(bias_group) and hasattr(k, "sharded_glove_step"):
        ge, gb_ = emb_group._c_group(), bias_group._c_group()
        if ge is not None and gb_ is not None:  # both lookups -> loss -> both updates as ONE library call
            emb_group.consolidate()
            return k.sharded_glove_step(ge, gb_, plan.c_struct(k), target, B, mode, lr, 1e-7)
    if plan.index is not None and getattr(k, "GRADS_AT_IDS", None) is not None and _is_f32(emb_group) and \
            _is_f32(bias_group):
        rows = emb_group.lookup_bucketed(plan)      # [2B (or the distinct rows), D] in exchange order
        brow = bias_group.lookup_bucketed(plan)     # [.., 1]
        loss, grad_rows, grad_bias = k.glove_fwd_bwd(rows, brow, plan.index.reshape(2, B), target, mode,
                                                     grads_at_ids=not plan.unique)
        emb_group.apply_sparse_adagrad(plan, grad_rows, lr, bucketed=not plan.unique)
        bias_group.apply_sparse_adagrad(plan, grad_bias.reshape(-1, 1), lr, bucketed=not plan.unique)
        return loss
    rows = emb_group.lookup(plan)           # [2B, D]: E[t1] ; E[t2]
    brow = bias_group.lookup(plan)          # [2B, 1]
    local_inputs = torch.arange(2 * B, dtype=torch.int32, device=rows.device).reshape(2, B)
    loss, grad_rows, grad_bias = k.glove_fwd_bwd(rows, brow, local_inputs, target, mode)
    emb_group.apply_sparse_adagrad(plan, grad_rows, lr)
    bias_group.apply_sparse_adagrad(plan, grad_bias.reshape(-1, 1), lr)
    return loss


def sharded_train_steps(workload, groups, batches, *, regularization=0.0, global_batch_size=None, scale=1.0, lr=0.05,
                        mode=None, plan_group=None):
    """The loop helper of the row-sharded API (what train_steps / train_epoch are to the single-GPU steps): every batch of
    `batches` stepped in order, the routing plans of `plan_group` coming batches made together -- one bucket / unique
    launch set, ONE counts all-to-all, ONE copy to pinned memory and ONE host wait per group, the group's ids exchanges
    as one RCCL group and its owner-side sorts as one batched sort -- and the NEXT group's plans enqueued in front of this
    group's steps, so the host's wait never finds an idle GPU.  Every step is then three library calls: esr_sharded_lookup,
    the loss kernel, esr_sharded_update.

    workload "inbatch": groups = (towers,), batches of (scene_ids, pos_ids[, ...]);  "triplet": groups = (towers,),
    batches of (scene_ids, pos_ids, neg_ids);  "glove": groups = (emb_group, bias_group), batches of (inputs [2, B],
    target [B]) and `mode` = the loss mode.  Returns the list of per-step loss tensors (this rank's share: all-reduce for
    the global loss).  The reference's loops being sharded: pinterest/train_shop_the_look.py:190-221,
    wikipedia/train_cooccurence.py:103-112."""
    from .train_state import quiet_gc
    if workload not in ("inbatch", "triplet", "glove"):
        raise ValueError("workload must be 'inbatch', 'triplet' or 'glove', got %r" % (workload,))
    g0 = groups[0]
    if plan_group is None:
        plan_group = max(1, int(os.environ.get("ESR_SHARDED_PLAN_GROUP", "8")))
    batches = list(batches)

    def lookup(b):
        if workload == "glove":
            return (g0, g0.virtual_id_segments([b[0].reshape(-1)], [0]))
        if workload == "inbatch":
            return (g0, g0.virtual_id_segments([b[0], b[1]], [0, 1]))
        return (g0, g0.virtual_id_segments([b[0], b[1], b[2]], [0, 1, 1]))

    def step(b, plan):
        if workload == "glove":
            return sharded_glove_step(g0, groups[1], b[0], b[1], mode, lr, plan=plan)
        gbs = global_batch_size if global_batch_size is not None else float(g0.world * b[0].numel())
        if workload == "inbatch":
            return sharded_inbatch_step(g0, b[0], b[1], regularization, gbs, scale, lr, plan=plan)
        return sharded_triplet_step(g0, b[0], b[1], b[2], regularization, gbs, lr, plan=plan)

    losses = []
    if not batches:
        return losses
    with quiet_gc():  # a full cyclic collection inside the loop is a 40 ms hole in the launch stream
        if g0.world1_direct:  # a world of one rank takes the single-GPU steps: nothing is routed
            for b in batches:
                losses.append(step(b, None))
            return losses
        spans = [(a, min(a + plan_group, len(batches))) for a in range(0, len(batches), plan_group)]
        pend = begin_plans([lookup(batches[i]) for i in range(*spans[0])])
        for gi, (a, e) in enumerate(spans):
            plans = pend.finish()  # ids exchanges + owner-side sorts of the whole group, ahead of its steps
            pend = begin_plans([lookup(batches[i]) for i in range(*spans[gi + 1])]) if gi + 1 < len(spans) else None
            for i in range(a, e):
                losses.append(step(batches[i], plans[i - a]))
    return losses


class _Collectives:
    """all_to_all_single / all_gather_into_tensor of a process group on the CURRENT stream through the direct RCCL
    exchange when there is one (esrecsys_amd/rccl.py), else through torch.distributed."""

    def __init__(self, group, device):
        self.pg = group
        self.x = None
        if device.type == "cuda":
            from . import rccl
            self.x = rccl.exchange_for(group, device)

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if self.x is not None:
            return self.x.all_to_all_single(out, inp, out_splits, in_splits)
        return dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.pg)

    def all_gather(self, out, inp):
        if self.x is not None:
            return self.x.all_gather_into_tensor(out, inp)
        return dist.all_gather_into_tensor(out, inp, group=self.pg)


def sharded_find_top_k(queries, local_candidates, k, group=None, kernels=None, mode="exact"):
    """Brute-force top-k over candidates that are row-sharded like the tables (this rank holds rows rank,
    rank + G, ...; BASELINE config 5).  Every rank brings its own [nq, D] queries (same nq on every rank):
    all-gather the queries, score ALL of them against the local shard, all-to-all the per-shard answers back to
    the rank that asked, merge the G lists.  Returns ([nq, k] scores, [nq, k] global row indices).  The three
    collectives run on the compute stream through the direct RCCL exchange (no stream hand-overs), as the training
    steps' do."""
    if kernels is None:
        from . import ops as kernels
    G, rank = dist.get_world_size(group), dist.get_rank(group)
    nq, D = queries.shape
    if G == 1:
        return kernels.retrieve_topk(queries, local_candidates, k, mode=mode)
    coll = _Collectives(group, queries.device)
    everyone = torch.empty((G * nq, D), dtype=queries.dtype, device=queries.device)
    coll.all_gather(everyone, queries.contiguous())
    s, i = kernels.retrieve_topk(everyone, local_candidates, k, mode=mode, index_base=rank, index_step=G)
    rs, ri = torch.empty_like(s), torch.empty_like(i)        # [G (shard), nq, k] after the exchange
    coll.all_to_all(rs, s)
    coll.all_to_all(ri, i)
    rs = rs.reshape(G, nq, k).permute(1, 0, 2).reshape(nq, G * k).contiguous()
    ri = ri.reshape(G, nq, k).permute(1, 0, 2).reshape(nq, G * k).contiguous()
    return kernels.topk_merge(rs, ri, k)
