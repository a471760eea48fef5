"""Row-sharded embedding tables across the GPUs of one node (build-defined -- the
reference is single-device and has no collective anywhere).

Partitioning: ``owner = id mod G``, ``local_row = id div G``; every rank holds V/G rows of the table and
of the optimizer accumulator.  The batch is data-parallel (each rank draws its own pairs).  Per step and
per table there are three exchanges over RCCL all-to-all (xGMI is a full mesh, so every peer slice rides
its own link):

    ids   -> owners      (bucket_ids_by_owner kernel, all_to_all_single of int32 local rows)
    rows  <- owners      (gather kernel on the owner, all_to_all_single of [n, D] rows, un-permute kernel)
    grads -> owners      (permute kernel, all_to_all_single, then the usual sort + segment-reduce + Adagrad)

Routing depends on the ids only, so it is split off as a *plan* (``make_plans``): bucket every lookup of
the step, exchange all per-peer counts in ONE all-to-all and read them back with ONE host
synchronisation (all_to_all_single needs host-side split sizes).  A training loop builds the plan for
batch k+1 right after it has enqueued step k, so the read-back waits behind useful GPU work instead of
draining the queue in the middle of a step.

``torch.distributed`` is plumbing (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests); the kernels
are libesr_hip.so.  ``kernels`` is the module that provides them -- always ``esrecsys_amd.ops`` in the
product; the CPU test-suite injects an oracle-backed double to exercise the routing logic without a GPU.
"""
import torch
import torch.distributed as dist


class RoutingPlan:
    """Where the ids of one lookup go: everything that does not depend on table contents."""

    def __init__(self, table, n, local_rows, perm, send_counts, recv_counts):
        self.table = table
        self.n = n
        self.local_rows = local_rows        # int32 [n]: ids // G in bucket (owner-major, stable) order
        self.perm = perm                    # int32 [n]: bucket order -> original position
        self.send_counts = send_counts      # python ints per peer: ids this rank asks of that peer
        self.recv_counts = recv_counts      # python ints per peer: ids that peer asks of this rank
        self.recv_local_rows = None         # int32 [sum(recv_counts)]: filled by exchange_ids()
        self.owner_sorted = None            # (sorted local rows, permutation) of recv_local_rows, for the update

    def exchange_ids(self):
        """ids -> owners.  Separate from make_plans so it can be issued ahead of the step as well."""
        if self.recv_local_rows is None:
            t = self.table
            self.recv_local_rows = torch.empty(sum(self.recv_counts), dtype=torch.int32, device=self.local_rows.device)
            dist.all_to_all_single(self.recv_local_rows, self.local_rows, self.recv_counts, self.send_counts,
                                   group=t.group)
            if self.recv_local_rows.numel():  # the owner-side sort needs the ids only: do it ahead of the step
                self.owner_sorted = t.k.segment_sort(self.recv_local_rows, t.local.shape[0])
        return self.recv_local_rows


def make_plans(lookups):
    """lookups: list of (RowShardedTable, global ids int32 [n]).  One counts all-to-all and one host sync
    for the whole list.  Returns one RoutingPlan per lookup (ids already exchanged)."""
    if not lookups:
        return []
    t0 = lookups[0][0]
    k, G, group = t0.k, t0.world, t0.group
    parts = []
    for table, ids in lookups:
        local_rows, perm, counts = k.bucket_ids_by_owner(ids, G)
        parts.append((table, ids.numel(), local_rows, perm, counts))
    # counts laid out [peer][lookup] so that all_to_all_single hands every peer its L counts
    send = torch.stack([p[4] for p in parts], dim=1).contiguous()          # [G, L] int64
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    both = torch.stack([send, recv]).cpu()                                  # the step's one host sync
    plans = []
    for i, (table, n, local_rows, perm, _) in enumerate(parts):
        plans.append(RoutingPlan(table, n, local_rows, perm, both[0, :, i].tolist(), both[1, :, i].tolist()))
    for p in plans:
        p.exchange_ids()
    return plans


class RowShardedTable:
    def __init__(self, local_table, local_accum, num_rows, group=None, kernels=None):
        if kernels is None:
            from . import ops as kernels
        self.k = kernels
        self.local = local_table      # [ceil((V - rank) / G), D]
        self.accum = local_accum      # fp32, same shape
        self.num_rows = int(num_rows)
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    @staticmethod
    def local_rows_for(num_rows, world, rank):
        return (num_rows - rank + world - 1) // world

    def lookup(self, plan_or_ids):
        """rows[i] = table[ids[i]] for global ids on this rank -> ([n, D] rows in the order of ids, plan)."""
        plan = plan_or_ids if isinstance(plan_or_ids, RoutingPlan) else make_plans([(self, plan_or_ids)])[0]
        k = self.k
        recv_ids = plan.exchange_ids()
        served = k.gather_rows(self.local, recv_ids)                            # [sum(recv), D]
        back = torch.empty((plan.n, self.local.shape[1]), dtype=self.local.dtype, device=served.device)
        dist.all_to_all_single(back, served, plan.send_counts, plan.recv_counts, group=self.group)
        rows = k.unpermute_rows(back, plan.perm)                                # bucket order -> id order
        return rows, plan

    def route_grads(self, plan, grad_rows):
        """Per-occurrence gradient rows (order of the looked-up ids) -> (local_row_ids, rows) on the owners."""
        k = self.k
        D = grad_rows.shape[1]
        bucketed = k.gather_rows(grad_rows, plan.perm)                          # id order -> bucket order
        recv = torch.empty((sum(plan.recv_counts), D), dtype=grad_rows.dtype, device=grad_rows.device)
        dist.all_to_all_single(recv, bucketed, plan.recv_counts, plan.send_counts, group=self.group)
        return plan.recv_local_rows, recv

    def apply_sparse_adagrad(self, plan, grad_rows, lr, eps=1e-7):
        """Route the gradients to their owners and update the local shard (sort + segment-reduce + RMW)."""
        k = self.k
        local_ids, rows = self.route_grads(plan, grad_rows)
        if local_ids.numel() == 0:
            return
        sorted_ids, perm = plan.owner_sorted
        k.sparse_adagrad(self.local, self.accum, sorted_ids, perm, rows, lr, eps)


def shard_of(table, world, rank):
    """The rows of a full table that `rank` owns, in local-row order (rows rank, rank+G, ...)."""
    return table[rank::world].contiguous()


def plan_inbatch(scene, product, scene_ids, pos_ids):
    return make_plans([(scene, scene_ids), (product, pos_ids)])


def plan_triplet(scene, product, scene_ids, pos_ids, neg_ids):
    return make_plans([(scene, scene_ids), (product, torch.cat([pos_ids, neg_ids]))])


def plan_glove(emb, bias, inputs):
    """The embedding and bias tables are indexed by the same ids and sharded the same way: one routing."""
    p = make_plans([(emb, inputs.reshape(-1))])[0]
    return [p, p]


def sharded_inbatch_step(scene, product, scene_ids, pos_ids, regularization, global_batch_size, scale, lr,
                         plans=None):
    """Data-parallel in-batch-softmax step on row-sharded towers.  Negatives are the local batch; gradients
    are normalised by the GLOBAL batch size, so the sum of the per-rank losses is the global mean loss."""
    k = scene.k
    p_q, p_c = plans if plans is not None else plan_inbatch(scene, product, scene_ids, pos_ids)
    q, _ = scene.lookup(p_q)
    c, _ = product.lookup(p_c)
    loss, _, gq, gc = k.inbatch_softmax_fwd_bwd(q, c, scale, regularization, global_batch_size)
    scene.apply_sparse_adagrad(p_q, gq, lr)
    product.apply_sparse_adagrad(p_c, gc, lr)
    return loss


def sharded_triplet_step(scene, product, scene_ids, pos_ids, neg_ids, regularization, global_batch_size, lr,
                         plans=None):
    """Reference triplet loss (pinterest/train_shop_the_look.py:93-109) on row-sharded towers.  The loss is a
    sum over triplets, so G ranks x B triplets == one device with G*B triplets and batch_size = G*B."""
    k = scene.k
    B = scene_ids.numel()
    p_s, p_pn = plans if plans is not None else plan_triplet(scene, product, scene_ids, pos_ids, neg_ids)
    s, _ = scene.lookup(p_s)
    pn, _ = product.lookup(p_pn)
    loss, _, _, gs, gp, gn = k.triplet_fwd_bwd(s, pn[:B], pn[B:], None, None, None, B, regularization,
                                               global_batch_size, with_reg=True, want_grads=True, want_scores=False)
    scene.apply_sparse_adagrad(p_s, gs, lr)
    gpn = gs._base[B:] if getattr(gs, "_base", None) is not None else torch.cat([gp, gn])  # [pos ; neg] rows
    product.apply_sparse_adagrad(p_pn, gpn, lr)
    return loss


def sharded_glove_step(emb, bias, inputs, target, mode, lr, plans=None):
    """GloVe step on row-sharded embedding + bias tables; the loss is over the local batch."""
    k = emb.k
    B = inputs.shape[1]
    p_e, p_b = plans if plans is not None else plan_glove(emb, bias, inputs)
    rows, _ = emb.lookup(p_e)          # [2B, D]: E[t1] ; E[t2]
    brow, _ = bias.lookup(p_b)         # [2B, 1]
    local_inputs = torch.arange(2 * B, dtype=torch.int32, device=rows.device).reshape(2, B)
    loss, grad_rows, grad_bias = k.glove_fwd_bwd(rows, brow, local_inputs, target, mode)
    emb.apply_sparse_adagrad(p_e, grad_rows, lr)
    bias.apply_sparse_adagrad(p_b, grad_bias.reshape(-1, 1), lr)
    return loss
