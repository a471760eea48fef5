"""Row-sharded embedding tables across the GPUs of one node (build-defined -- the
reference is single-device and has no collective anywhere).

Partitioning: ``owner = id mod G``, ``local_row = id div G``; every rank holds V/G rows of the table and
of the optimizer accumulator.  The batch is data-parallel (each rank draws its own pairs).  Per step and
per table there are three exchanges over RCCL all-to-all (xGMI is a full mesh, so every peer slice rides
its own link):

    ids   -> owners      (bucket_ids_by_owner kernel, all_to_all_single of int32 local rows)
    rows  <- owners      (gather kernel on the owner, all_to_all_single of [n, D] rows, un-permute kernel)
    grads -> owners      (permute kernel, all_to_all_single, then the usual sort + segment-reduce + Adagrad)

``torch.distributed`` is plumbing (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests); the kernels
are libesr_hip.so.  ``kernels`` is the module that provides them -- always ``esrecsys_amd.ops`` in the
product; the CPU test-suite injects an oracle-backed double to exercise the routing logic without a GPU.
"""
import torch
import torch.distributed as dist


class LookupContext:
    """What a lookup must remember to route the gradients of its rows back to their owners."""

    def __init__(self, perm, send_counts, recv_counts, recv_local_rows, n):
        self.perm = perm                          # int32 [n]: bucket order -> original position
        self.send_counts = send_counts            # python ints, per peer: ids this rank asked of that peer
        self.recv_counts = recv_counts            # python ints, per peer: ids that peer asked of this rank
        self.recv_local_rows = recv_local_rows    # int32 [sum(recv_counts)]: local rows requested of this rank
        self.n = n


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

    def lookup(self, ids):
        """rows[i] = table[ids[i]] for global ids on this rank -> ([n, D] rows in the order of ids, ctx)."""
        k, G = self.k, self.world
        n = ids.numel()
        local_rows, perm, counts = k.bucket_ids_by_owner(ids, G)
        recv_counts_t = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts_t, counts, group=self.group)
        # the only host synchronisation of the step: all_to_all_single needs host-side split sizes
        send_counts = counts.cpu().tolist()
        recv_counts = recv_counts_t.cpu().tolist()
        recv_rows_ids = torch.empty(sum(recv_counts), dtype=torch.int32, device=ids.device)
        dist.all_to_all_single(recv_rows_ids, local_rows, recv_counts, send_counts, group=self.group)
        served = k.gather_rows(self.local, recv_rows_ids)                      # [sum(recv), D]
        back = torch.empty((n, self.local.shape[1]), dtype=self.local.dtype, device=ids.device)
        dist.all_to_all_single(back, served, send_counts, recv_counts, group=self.group)
        rows = k.unpermute_rows(back, perm)                                     # bucket order -> id order
        return rows, LookupContext(perm, send_counts, recv_counts, recv_rows_ids, n)

    def route_grads(self, ctx, grad_rows):
        """Per-occurrence gradient rows (order of the looked-up ids) -> (local_row_ids, rows) on the owners."""
        k = self.k
        D = grad_rows.shape[1]
        bucketed = k.gather_rows(grad_rows, ctx.perm)                           # id order -> bucket order
        recv = torch.empty((sum(ctx.recv_counts), D), dtype=grad_rows.dtype, device=grad_rows.device)
        dist.all_to_all_single(recv, bucketed, ctx.recv_counts, ctx.send_counts, group=self.group)
        return ctx.recv_local_rows, recv

    def apply_sparse_adagrad(self, ctx, grad_rows, lr, eps=1e-7):
        """Route the gradients to their owners and update the local shard (sort + segment-reduce + RMW)."""
        k = self.k
        local_ids, rows = self.route_grads(ctx, grad_rows)
        if local_ids.numel() == 0:
            return
        sorted_ids, perm = k.segment_sort(local_ids, self.local.shape[0])
        k.sparse_adagrad(self.local, self.accum, sorted_ids, perm, rows, lr, eps)


def shard_of(table, world, rank):
    """The rows of a full table that `rank` owns, in local-row order (rows rank, rank+G, ...)."""
    return table[rank::world].contiguous()


def sharded_inbatch_step(scene, product, scene_ids, pos_ids, regularization, global_batch_size, scale, lr):
    """Data-parallel in-batch-softmax step on row-sharded towers.  Negatives are the local batch; gradients
    are normalised by the GLOBAL batch size, so the sum of the per-rank losses is the global mean loss."""
    k = scene.k
    q, ctx_q = scene.lookup(scene_ids)
    c, ctx_c = product.lookup(pos_ids)
    loss, _, gq, gc = k.inbatch_softmax_fwd_bwd(q, c, scale, regularization, global_batch_size)
    scene.apply_sparse_adagrad(ctx_q, gq, lr)
    product.apply_sparse_adagrad(ctx_c, gc, lr)
    return loss


def sharded_triplet_step(scene, product, scene_ids, pos_ids, neg_ids, regularization, global_batch_size, lr):
    """Reference triplet loss (pinterest/train_shop_the_look.py:93-109) on row-sharded towers.  The loss is a
    sum over triplets, so G ranks x B triplets == one device with G*B triplets and batch_size = G*B."""
    k = scene.k
    B = scene_ids.numel()
    s, ctx_s = scene.lookup(scene_ids)
    pn, ctx_pn = product.lookup(torch.cat([pos_ids, neg_ids]))
    loss, _, _, gs, gp, gn = k.triplet_fwd_bwd(s, pn[:B], pn[B:], None, None, None, B, regularization,
                                               global_batch_size, with_reg=True, want_grads=True, want_scores=False)
    scene.apply_sparse_adagrad(ctx_s, gs, lr)
    product.apply_sparse_adagrad(ctx_pn, gp._base if getattr(gp, "_base", None) is not None else
                                 torch.cat([gp, gn]), lr)
    return loss


def sharded_glove_step(emb, bias, inputs, target, mode, lr):
    """GloVe step on row-sharded embedding + bias tables; the loss is over the local batch."""
    k = emb.k
    B = inputs.shape[1]
    flat = inputs.reshape(-1)
    rows, ctx_e = emb.lookup(flat)          # [2B, D]: E[t1] ; E[t2]
    brow, ctx_b = bias.lookup(flat)         # [2B, 1]
    local_inputs = torch.arange(2 * B, dtype=torch.int32, device=flat.device).reshape(2, B)
    loss, grad_rows, grad_bias = k.glove_fwd_bwd(rows, brow, local_inputs, target, mode)
    emb.apply_sparse_adagrad(ctx_e, grad_rows, lr)
    bias.apply_sparse_adagrad(ctx_b, grad_bias.reshape(-1, 1), lr)
    return loss
