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


def _a2a(group, out, inp, out_splits=None, in_splits=None, lane=0):
    """all_to_all_single of a ShardedTableGroup: RCCL send / recv on the current stream when available
    (esrecsys_amd/rccl.py: no stream hand-overs), else torch.distributed.  lane 1: the second communicator, for the rows
    exchange that runs on a side stream under the previous step's kernels (sharded_train_steps, overlap)."""
    x = group.exchange(lane)
    if x is not None:
        return x.all_to_all_single(out, inp, out_splits, in_splits)
    pg = group.prefetch_pg if lane and group.prefetch_pg is not None else group.pg
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=pg)


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
        self.dev_counts = None              # (send, recv) int64 [G] on the device: the same counts, for begin_stale_sets
        self.stale = None                   # StaleRows: what the PREVIOUS step of a loop updates of this lookup (overlap)

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
            for i, p in enumerate(self.plans):
                p.dev_counts = (self.both_dev[0, :, i], self.both_dev[1, :, i])
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
            p._batch = (buf, srt, i)  # (begin_stale_sets searches these matrices as they are)
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


class StaleRows:
    """Of one lookup of a training loop: the rows that the step BEFORE it updates -- on whichever rank that step's batch
    named them.  A lookup issued ahead of that update (the overlap of sharded_train_steps) carries old values in exactly
    these rows; ShardedTableGroup.patch_rows serves them again once the update is done."""

    def __init__(self, ids, send_counts, recv_counts, pos):
        self.ids = ids                    # int32: virtual local rows this rank serves AGAIN, requester-major
        self.send_counts = send_counts    # python ints per requester (rows of `ids`)
        self.recv_counts = recv_counts    # python ints per owner: rows that come back to this rank
        self.pos = pos                    # int32 [sum(recv_counts)]: the rows of the looked-up block they replace
        self._c = None

    def c_parts(self, k):
        """(rows, asked host array, pos, ask host array) for esr_step_overlap_t."""
        if self._c is None:
            self._c = (self.ids, k.i64_array(self.send_counts), self.pos, k.i64_array(self.recv_counts))
        return self._c


class PendingStale:
    """begin_stale_sets, enqueued: ``finish()`` waits for its counts copy (a loop calls it a group of steps later) and
    cuts the per-plan StaleRows out of the group's matrices -- no further device work."""

    def __init__(self, plans, ids_sorted, pos_sorted, both_host, event):
        self.plans, self.ids_sorted, self.pos_sorted, self.both_host, self.event = plans, ids_sorted, pos_sorted, both_host, event

    def finish(self):
        if self.plans is None:
            return
        if self.event is not None:
            self.event.synchronize()
        host = self.both_host
        for i, p in enumerate(self.plans):
            out_c, in_c = host[0, :, i].tolist(), host[1, :, i].tolist()
            p.stale = StaleRows(self.ids_sorted[i, :sum(out_c)], out_c, in_c, self.pos_sorted[i, :sum(in_c)])
        self.plans = None


def _padded_lists(plans, sentinel):
    """([L, n_max] lists asked of this rank, the same lists sorted), rows padded with `sentinel`: the matrices the batched
    owner-side sort left behind when there was one (_grouped_owner_sort), else assembled here."""
    first = getattr(plans[0], "_batch", None)
    if first is not None and first[0].shape[0] == len(plans) and \
            all(getattr(p, "_batch", (None, None, -1))[0] is first[0] and p._batch[2] == i for i, p in enumerate(plans)):
        return first[0], first[1]
    dev = plans[0].recv_local_rows.device
    n_max = max([int(p.recv_local_rows.numel()) for p in plans] + [1])
    cur = torch.full((len(plans), n_max), sentinel, dtype=torch.int32, device=dev)
    srt = torch.full((len(plans), n_max), sentinel, dtype=torch.int32, device=dev)
    for i, p in enumerate(plans):
        n = int(p.recv_local_rows.numel())
        if n:
            cur[i, :n] = p.recv_local_rows
            srt[i, :n] = p.owner_sorted[0]
    return cur, srt


def begin_stale_sets(plans, prev):
    """For consecutive lookups of ONE table group in a training loop -- plans[i] follows plans[i - 1], plans[0] follows
    `prev` (the last plan of the group of batches before; None: nothing precedes) -- find, on the owner, the rows of every
    lookup that its predecessor's step updates: positions of the list asked of this rank whose row is in the predecessor's
    list.  Ids only, like the routing plans, so it runs ahead with them and batched over the group: a membership search of
    the asked lists in the predecessors' sorted lists; the 0 / 1 answers go back to the askers (one byte per asked row: the
    sizes are the plans' own, ONE RCCL group for the group's plans), so that owner and asker each derive their half --
    the rows to serve again, per asker; the places they go to, per owner -- and ONE copy of the two count matrices to
    pinned memory.  SURVEY 8e: the exchange of batch k + 1 under batch k's kernels.  The search and the two stable
    partitions are the library's own launches (esr_sorted_membership, esr_flagged_first; round 5), batched over the
    group's plans."""
    g = plans[0].group
    G, L = g.world, len(plans)
    for p in plans:
        p.exchange_ids()
    sentinel = g.loff[-1]  # beyond every virtual local row
    cur, srt = _padded_lists(plans, sentinel)
    dev = cur.device
    n_max = cur.shape[1]
    first = prev.owner_sorted[0] if prev is not None and prev.owner_sorted is not None else None
    m = max(n_max, int(first.numel()) if first is not None else 0)
    seq = torch.full((L, m), sentinel + 1, dtype=torch.int32, device=dev)  # row i: what plan i's predecessor updates
    if L > 1:
        seq[1:, :n_max] = srt[:-1]
    if first is not None and first.numel():
        seq[0, :first.numel()] = first
    k = g.k
    both = torch.zeros((2, G, L), dtype=torch.int64, device=dev)  # [0]: rows to serve again per asker, [1]: rows coming back
    hit8 = k.sorted_membership(cur, seq, sentinel)                # uint8 [L, n_max]
    # owner: the rows to serve again, asker by asker (the order of the asked list is kept)
    ids_sorted = k.flagged_first(hit8, cur, torch.stack([p.dev_counts[1] for p in plans]), both[0])
    # asker: which rows of its looked-up block those are
    r_max = max([p.n_rows for p in plans] + [1])
    if G == 1:
        mask = hit8  # (a world of one rank asks itself: block order == asked order)
    else:
        mask = torch.zeros((L, r_max), dtype=torch.uint8, device=dev)
        moves = [(mask[i, :p.n_rows], hit8[i, :p.recv_local_rows.numel()], p.send_counts, p.recv_counts)
                 for i, p in enumerate(plans)]
        x = g.exchange()
        if x is not None and hasattr(x, "all_to_all_multi"):
            x.all_to_all_multi(moves)
        else:
            for mv in moves:
                _a2a(g, *mv)
    pos_sorted = k.flagged_first(mask, None, torch.stack([p.dev_counts[0] for p in plans]), both[1])
    if both.is_cuda:
        host = _pinned_like(both)
        host.copy_(both, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return PendingStale(list(plans), ids_sorted, pos_sorted, host, ev)
    return PendingStale(list(plans), ids_sorted, pos_sorted, both, None)


class ShardedTableGroup:
    """Same-width, same-dtype row-sharded tables that one step looks up and updates together."""

    def __init__(self, tables, group=None, kernels=None, unique=None, grad_dtype=None, prefetch_group=None):
        if kernels is None:
            from . import ops as kernels
        self.k = kernels
        self.tables = list(tables)
        self.pg = group
        # a second process group over the same ranks for lane 1 when the exchange runs on torch.distributed (whose
        # collectives of one group are serialised); the direct RCCL exchange makes its own second communicator
        self.prefetch_pg = prefetch_group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        G = self.world
        self.voff = [0]                     # global virtual offsets, multiples of G
        for t in self.tables:
            self.voff.append(self.voff[-1] + G * ((t.num_rows + G - 1) // G))
        self.loff = [o // G for o in self.voff]  # the same boundaries in virtual LOCAL rows
        self.dim = self.tables[0].local.shape[1] if self.tables[0].local.dim() > 1 else 1
        self._xch = [False, False]  # per lane: DirectExchange, None (use torch.distributed) or False (not resolved yet)
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

    def _fused(self, lane=0):
        """(comm, tables_c, accums_c, loff_c, dtype code) when the exchange halves of a step run as ONE library call each
        (esr_sharded_lookup / esr_sharded_update): CUDA shards and either a world of one rank or the direct RCCL exchange.
        None: op by op through `kernels` and torch.distributed (the CPU doubles of the gloo tests, ESR_SHARDED_FUSED=0)."""
        k = self.k
        if not hasattr(k, "sharded_lookup") or os.environ.get("ESR_SHARDED_FUSED", "1") != "1" or \
                not all(t.local.is_cuda for t in self.tables):
            return None
        comm = None
        if self.world > 1:
            x = self.exchange(lane)
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

    def exchange(self, lane=0):
        if self._xch[lane] is False:
            dev = self.tables[0].local.device
            if dev.type == "cuda":
                from . import rccl
                self._xch[lane] = rccl.exchange_for(self.pg, dev, lane)
            else:
                self._xch[lane] = None
        return self._xch[lane]

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

    def lookup_buffers(self, plan):
        """(back, served) of lookup_bucketed, allocated by the caller: a lookup issued on a side stream takes buffers made
        on the main stream (and kept until the main stream has waited for it)."""
        recv = plan.exchange_ids()
        dt, dev = self.tables[0].local.dtype, recv.device
        back = torch.empty((plan.n_rows, self.dim), dtype=dt, device=dev)
        served = torch.empty((recv.numel(), self.dim), dtype=dt, device=dev) if self.world > 1 else None
        return back, served

    def _gather(self, rows, out=None):
        k = self.k
        kw = {} if out is None else {"out": out}
        if len(self.tables) == 1:
            return k.gather_rows(self.tables[0].local, rows, **kw)
        return k.gather_rows_multi([t.local for t in self.tables], self.loff, rows, **kw)

    def lookup_bucketed(self, plan, lane=0, buffers=None):
        """The looked-up rows in BUCKET order (row plan.inv[i] belongs to virtual id i), in the tables' dtype --
        for consumers that can index them themselves and so skip the un-permute pass.  lane 1 / buffers: the lookup of
        the NEXT batch on a side stream and the second communicator (sharded_train_steps, overlap)."""
        k = self.k
        self.consolidate()  # rows a world-1 one-pass step left in the second buffers (no-op when nothing is displaced)
        recv = plan.exchange_ids()
        back, served = buffers if buffers is not None else self.lookup_buffers(plan)
        fused = self._fused(lane)
        if fused is not None:  # gather + rows exchange as one library call
            comm, tc, _, lc, code = fused
            ask_c, asked_c = plan.c_counts(k)
            return k.sharded_lookup(comm, self.world, tc, lc, len(self.tables), code, self.dim, recv, asked_c, ask_c,
                                    served, back)
        served = self._gather(recv, served if recv.numel() else None)
        _a2a(self, back, served, plan.send_counts, plan.recv_counts, lane=lane)
        return back

    def patch_rows(self, plan, back):
        """`back` = lookup_bucketed(plan) issued BEFORE the previous step's update: serve the rows that update wrote
        (plan.stale, begin_stale_sets) again and put them in their places.  Main stream, first communicator."""
        st = plan.stale
        k = self.k
        served = self._gather(st.ids) if st.ids.numel() else torch.empty((0, self.dim), dtype=back.dtype, device=back.device)
        if self.world == 1:
            got = served
        else:
            got = torch.empty((st.pos.numel(), self.dim), dtype=back.dtype, device=back.device)
            _a2a(self, got, served, st.recv_counts, st.send_counts)
        if got.shape[0]:
            k.unpermute_rows(got, st.pos, out=back)
        return back

    def lookup(self, plan, back=None):
        """rows[i] = table_of(vid_i)[id_i] for this rank's virtual ids -> [n, D] in the order of the ids."""
        back = back if back is not None else self.lookup_bucketed(plan)
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


def _is_bf16(group):
    return all(t.local.dtype == torch.bfloat16 for t in group.tables)


def _joined(first, *rest):
    """The single buffer the gradient slices are views of, else their concatenation."""
    base = getattr(first, "_base", None)
    n = first.shape[0] + sum(r.shape[0] for r in rest)
    if base is not None and base.shape[0] == n:
        return base
    return torch.cat((first,) + rest)


def _world1_tables_ok(group):
    return group.world1_direct and all(t.local.is_cuda for t in group.tables)


def sharded_inbatch_step(towers, scene_ids, pos_ids, regularization, global_batch_size, scale, lr, plan=None, rows=None):
    """Data-parallel in-batch-softmax step on row-sharded towers (group = [scene table, product table]).
    Negatives are the local batch; gradients are normalised by the GLOBAL batch size, so the sum of the
    per-rank losses is the global mean loss.  rows: lookup_bucketed(plan) when the caller has it already (overlap)."""
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
        back = rows if rows is not None else towers.lookup_bucketed(plan)
        iq, ic = plan.index[:B], plan.index[B:]
        if plan.unique:  # several occurrences may read one row: per-occurrence gradient rows, summed per distinct row
            loss, _, gq, gc = folded(back, back, iq, ic, scale, regularization, global_batch_size)
            towers.apply_sparse_adagrad(plan, _joined(gq, gc), lr)
            return loss
        loss, _, gbuf, _ = folded(back, back, iq, ic, scale, regularization, global_batch_size,
                                  grad_positions=(iq, ic))
        towers.apply_sparse_adagrad(plan, gbuf, lr, bucketed=True)
        return loss
    rows = towers.lookup(plan, back=rows)          # [q ; c]
    loss, _, gq, gc = k.inbatch_softmax_fwd_bwd(rows[:B], rows[B:], scale, regularization, global_batch_size)
    towers.apply_sparse_adagrad(plan, _joined(gq, gc), lr)
    return loss


def sharded_triplet_step(towers, scene_ids, pos_ids, neg_ids, regularization, global_batch_size, lr, plan=None, rows=None):
    """Reference triplet loss (pinterest/train_shop_the_look.py:93-109) on row-sharded towers.  The loss is a
    sum over triplets, so G ranks x B triplets == one device with G*B triplets and batch_size = G*B.
    rows: lookup_bucketed(plan) when the caller has it already (overlap)."""
    k = towers.k
    B = scene_ids.numel()
    if (_world1_tables_ok(towers) and (_is_f32(towers) or _is_bf16(towers)) and
            getattr(k, "triplet_direct_mode", lambda: False)()):
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
    if rows is None and plan.index is not None and _is_f32(towers) and hasattr(k, "sharded_triplet_step"):
        gs_ = towers._c_group()
        if gs_ is not None:  # lookup -> loss -> update as ONE library call (esr_sharded_triplet_step)
            towers.consolidate()
            return k.sharded_triplet_step(gs_, plan.c_struct(k), B, regularization, global_batch_size, lr, 1e-7,
                                          scene_ids.device)
    if plan.index is not None and getattr(k, "GRADS_AT_IDS", None) is not None and _is_f32(towers):
        # the loss kernel indexes the exchanged rows where they landed (bucket order) through the inverse
        # permutation and writes every gradient row back at that position: no un-permute / permute passes
        back = rows if rows is not None else towers.lookup_bucketed(plan)
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
    rows = towers.lookup(plan, back=rows)          # [scene ; pos ; neg]
    loss, _, _, gs, gp, gn = k.triplet_fwd_bwd(rows[:B], rows[B:2 * B], rows[2 * B:], None, None, None, B,
                                               regularization, global_batch_size, with_reg=True, want_grads=True,
                                               want_scores=False)
    towers.apply_sparse_adagrad(plan, _joined(gs, gp, gn), lr)
    return loss


def sharded_glove_step(emb_group, bias_group, inputs, target, mode, lr, plan=None, rows=None):
    """GloVe step on row-sharded embedding + bias tables (two single-table groups sharing one routing plan:
    same ids, same sharding, different widths); the loss is over the local batch.  rows: (lookup_bucketed of the
    embedding group, of the bias group) when the caller has them already (overlap)."""
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
    if rows is None and plan.index is not None and _is_f32(emb_group) and _is_f32(bias_group) and \
            hasattr(k, "sharded_glove_step"):
        ge, gb_ = emb_group._c_group(), bias_group._c_group()
        if ge is not None and gb_ is not None:  # both lookups -> loss -> both updates as ONE library call
            emb_group.consolidate()
            return k.sharded_glove_step(ge, gb_, plan.c_struct(k), target, B, mode, lr, 1e-7)
    if plan.index is not None and getattr(k, "GRADS_AT_IDS", None) is not None and _is_f32(emb_group) and \
            _is_f32(bias_group):
        erow, brow = rows if rows is not None else (None, None)
        rows = erow if erow is not None else emb_group.lookup_bucketed(plan)   # [2B (or the distinct rows), D], exchange order
        brow = brow if brow is not None else bias_group.lookup_bucketed(plan)  # [.., 1]
        loss, grad_rows, grad_bias = k.glove_fwd_bwd(rows, brow, plan.index.reshape(2, B), target, mode,
                                                     grads_at_ids=not plan.unique)
        emb_group.apply_sparse_adagrad(plan, grad_rows, lr, bucketed=not plan.unique)
        bias_group.apply_sparse_adagrad(plan, grad_bias.reshape(-1, 1), lr, bucketed=not plan.unique)
        return loss
    erow, brow = rows if rows is not None else (None, None)
    rows = emb_group.lookup(plan, back=erow)    # [2B, D]: E[t1] ; E[t2]
    brow = bias_group.lookup(plan, back=brow)   # [2B, 1]
    local_inputs = torch.arange(2 * B, dtype=torch.int32, device=rows.device).reshape(2, B)
    loss, grad_rows, grad_bias = k.glove_fwd_bwd(rows, brow, local_inputs, target, mode)
    emb_group.apply_sparse_adagrad(plan, grad_rows, lr)
    bias_group.apply_sparse_adagrad(plan, grad_bias.reshape(-1, 1), lr)
    return loss


def sharded_train_steps(workload, groups, batches, *, regularization=0.0, global_batch_size=None, scale=1.0, lr=0.05,
                        mode=None, plan_group=None, overlap=None):
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
    wikipedia/train_cooccurence.py:103-112.

    overlap (ESR_SHARDED_OVERLAP=1): batch k + 1's gather + rows exchange is issued on a side stream and a
    second communicator BEFORE batch k's loss kernel, so it runs under that kernel, the gradient exchange and the update.
    Rows that batch k's update writes reach it too early; they are known from the ids alone (begin_stale_sets: the
    intersection of two owner-side lists, made with the plans a group ahead), and after the update they are served again
    in a second, small exchange (patch_rows) -- results equal the sequential loop's bit for bit.  Each step is then
    lookup-patch, loss kernel, esr_sharded_update on the main stream + the next esr_sharded_lookup beside them."""
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

    def step(b, plan, rows=None):
        if workload == "glove":
            return sharded_glove_step(g0, groups[1], b[0], b[1], mode, lr, plan=plan, rows=rows)
        gbs = global_batch_size if global_batch_size is not None else float(g0.world * b[0].numel())
        rows = rows[0] if rows is not None else None
        if workload == "inbatch":
            return sharded_inbatch_step(g0, b[0], b[1], regularization, gbs, scale, lr, plan=plan, rows=rows)
        return sharded_triplet_step(g0, b[0], b[1], b[2], regularization, gbs, lr, plan=plan, rows=rows)

    losses = []
    if not batches:
        return losses
    if overlap is None:
        overlap = os.environ.get("ESR_SHARDED_OVERLAP", "0") == "1"
    with quiet_gc():  # a full cyclic collection inside the loop is a 40 ms hole in the launch stream
        if g0.world1_direct:  # a world of one rank takes the single-GPU steps: nothing is routed
            for b in batches:
                losses.append(step(b, None))
            return losses
        spans = [(a, min(a + plan_group, len(batches))) for a in range(0, len(batches), plan_group)]
        if overlap:
            looked = tuple(groups[:2]) if workload == "glove" else (g0,)
            dev = g0.tables[0].local.device
            main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
            side = _side_stream(dev) if main is not None else None

            def early(plan):
                """The next batch's lookups behind everything the main stream holds NOW (the previous update included):
                buffers from the main stream's pool, work on the side stream and lane 1."""
                bufs = [g.lookup_buffers(plan) for g in looked]
                if side is None:
                    return [g.lookup_bucketed(plan, lane=1, buffers=b) for g, b in zip(looked, bufs)], bufs, None
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    backs = [g.lookup_bucketed(plan, lane=1, buffers=b) for g, b in zip(looked, bufs)]
                    done = torch.cuda.Event()
                    done.record(side)
                return backs, bufs, done

            # one library call per step (esr_sharded_*_step_overlapped: patch, next lookup on the side stream, loss kernel,
            # update) where there is a one-call step at all; else the same sequence op by op from here
            k = g0.k
            c_groups = None
            if workload != "inbatch" and side is not None and hasattr(k, "step_overlap_struct") and \
                    all(_is_f32(g) for g in looked) and os.environ.get("ESR_SHARDED_OVERLAP_CALLS", "one") == "one":
                c_groups = [g._c_group() for g in looked]
                x2 = g0.exchange(1) if g0.world > 1 else None
                if any(c is None for c in c_groups) or (g0.world > 1 and (x2 is None or not getattr(x2, "comm", None))):
                    c_groups = None
                comm2 = x2.comm if c_groups is not None and x2 is not None else None

            def fused_step(b, plan, nxt, prev):
                """prev / returns: (backs, serveds, ready) of the lookup made ahead for this / the next batch."""
                nb = [g.lookup_buffers(nxt) for g in looked] if nxt is not None else []
                ov = k.step_overlap_struct([x for x in prev[0]] if prev else None, prev[2] if prev else None,
                                           plan.stale.c_parts(k) if prev else None,
                                           nxt.c_struct(k) if nxt is not None else None, [x[0] for x in nb],
                                           [x[1] for x in nb], comm2, side)
                if prev:
                    prev[2] = None  # (consumed -- or released -- by the call)
                try:
                    if workload == "glove":
                        loss = k.sharded_glove_step(c_groups[0], c_groups[1], plan.c_struct(k), b[1], b[0].shape[1], mode,
                                                    lr, 1e-7, overlap=ov)
                    else:
                        gbs = global_batch_size if global_batch_size is not None else float(g0.world * b[0].numel())
                        loss = k.sharded_triplet_step(c_groups[0], plan.c_struct(k), b[0].numel(), regularization, gbs,
                                                      lr, 1e-7, b[0].device, overlap=ov)
                except BaseException:
                    # a call that failed AFTER its side-stream lookup recorded next_ready: nobody will consume the event
                    # (the loop's ``finally`` only knows the event of the step before)
                    if ov.next_ready:
                        k.overlap_release(ov.next_ready)
                        ov.next_ready = None
                    raise
                return loss, ([x[0] for x in nb], [x[1] for x in nb], ov.next_ready) if nxt is not None else None

            plans = {0: begin_plans([lookup(batches[i]) for i in range(*spans[0])]).finish()}
            stale = {0: begin_stale_sets(plans[0], None)}
            pend = {1: begin_plans([lookup(batches[i]) for i in range(*spans[1])])} if len(spans) > 1 else {}
            ahead = None
            try:
                for gi, (a, e) in enumerate(spans):
                    if gi + 1 < len(spans):  # the next group: its plans finished and its stale sets begun a group ahead
                        plans[gi + 1] = pend.pop(gi + 1).finish()
                        stale[gi + 1] = begin_stale_sets(plans[gi + 1], plans[gi][-1])
                        if gi + 2 < len(spans):
                            pend[gi + 2] = begin_plans([lookup(batches[i]) for i in range(*spans[gi + 2])])
                    stale.pop(gi).finish()
                    cur = plans.pop(gi)
                    for i in range(a, e):
                        plan = cur[i - a]
                        nxt = cur[i - a + 1] if i + 1 < e else (plans[gi + 1][0] if gi + 1 < len(spans) else None)
                        if c_groups is not None:
                            for g in looked:
                                g.consolidate()
                            loss, ahead = fused_step(batches[i], plan, nxt, list(ahead) if ahead else None)
                            losses.append(loss)
                            continue
                        if ahead is None:
                            rows = [g.lookup_bucketed(plan) for g in looked]
                        else:
                            backs, _, done = ahead  # (the exchange buffers stay referenced until the main stream has waited)
                            if done is not None:
                                main.wait_event(done)
                            rows = [g.patch_rows(plan, b) for g, b in zip(looked, backs)]
                        ahead = early(nxt) if nxt is not None else None
                        losses.append(step(batches[i], plan, rows=rows))
            finally:
                if c_groups is not None and ahead and ahead[2]:
                    k.overlap_release(ahead[2])
            return losses
        pend = begin_plans([lookup(batches[i]) for i in range(*spans[0])])
        for gi, (a, e) in enumerate(spans):
            plans = pend.finish()  # ids exchanges + owner-side sorts of the whole group, ahead of its steps
            pend = begin_plans([lookup(batches[i]) for i in range(*spans[gi + 1])]) if gi + 1 < len(spans) else None
            for i in range(a, e):
                losses.append(step(batches[i], plans[i - a]))
    return losses


_side_streams = {}


def _side_stream(dev):
    """One side stream per device for the overlapped lookups (creating a stream per call costs a driver round trip)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


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


def sharded_find_top_k(queries, local_candidates, k, group=None, kernels=None, mode="exact", prepared=None):
    """Brute-force top-k over candidates that are row-sharded like the tables (this rank holds rows rank,
    rank + G, ...; BASELINE config 5).  Every rank brings its own [nq, D] queries (same nq on every rank):
    all-gather the queries, score ALL of them against the local shard, all-to-all the per-shard answers back to
    the rank that asked, merge the G lists.  Returns ([nq, k] scores, [nq, k] global row indices).  The three
    collectives run on the compute stream through the direct RCCL exchange (no stream hand-overs), as the training
    steps' do.  prepared: ops.retrieve_prepare(local_candidates, mode) -- this rank's shard prepared once."""
    if kernels is None:
        from . import ops as kernels
    kw = {} if prepared is None else {"prepared": prepared}
    if prepared is not None:
        mode = next(n for n, c in kernels._RETRIEVE_MODES.items() if c == prepared.mode)
    G, rank = dist.get_world_size(group), dist.get_rank(group)
    nq, D = queries.shape
    if G == 1:
        return kernels.retrieve_topk(queries, local_candidates, k, mode=mode, **kw)
    coll = _Collectives(group, queries.device)
    everyone = torch.empty((G * nq, D), dtype=queries.dtype, device=queries.device)
    coll.all_gather(everyone, queries.contiguous())
    s, i = kernels.retrieve_topk(everyone, local_candidates, k, mode=mode, index_base=rank, index_step=G, **kw)
    rs, ri = torch.empty_like(s), torch.empty_like(i)        # [G (shard), nq, k] after the exchange
    coll.all_to_all(rs, s)
    coll.all_to_all(ri, i)
    rs = rs.reshape(G, nq, k).permute(1, 0, 2).reshape(nq, G * k).contiguous()
    ri = ri.reshape(G, nq, k).permute(1, 0, 2).reshape(nq, G * k).contiguous()
    return kernels.topk_merge(rs, ri, k)
