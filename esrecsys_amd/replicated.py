"""Replicated-table data parallelism ("pairs/sec at G in {1,2,4,8} on C2, replicated AND sharded").

Every rank holds the FULL tables and accumulators (C2: 2 x 512 MB + 2 x 512 MB per GPU).  A step is

    local     loss + per-occurrence gradient rows of this rank's B pairs from its own copy of the tables (the
              single-GPU kernels; no row lookup crosses the node)
    gather    all-gather of every rank's occurrence ids and gradient rows (equal blocks: ncclSend / ncclRecv pairs on
              the compute stream, esr_allgather_bytes)
    update    ONE sparse Adagrad update over the GLOBAL occurrence list, identical on every rank: same ids, same rows,
              same deterministic sort and segment sums -> the replicas stay bit-identical without ever being compared

against the row-sharded step's four all-to-alls.  The trade: no lookup exchange and no routing plan (no host
read-back at all), but every rank applies all G x B updates (the optimizer work does not shrink with G) and receives
G x the gradient bytes: at C2 (B = 8192, D = 128, G = 8) 67 MB per step per rank against 2 x 8 MB of rows + gradients
for the sharded step.  Build-defined: the reference is single-device (wikipedia/train_cooccurence.py:147-150 only logs
the device count).

``kernels`` is ``esrecsys_amd.ops`` in the product; the CPU tests inject the oracle-backed double.
"""
import torch
import torch.distributed as dist

from .sharded import _Collectives, _joined


class ReplicatedTables:
    """Full copies of same-width tables (and their fp32 accumulators) on every rank of `group`."""

    def __init__(self, tables, accums, group=None, kernels=None):
        if kernels is None:
            from . import ops as kernels
        self.k = kernels
        self.tables, self.accums = list(tables), list(accums)
        self.pg = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.coll = _Collectives(group, self.tables[0].device)
        self.row_offsets = [0]
        for t in self.tables:
            self.row_offsets.append(self.row_offsets[-1] + int(t.shape[0]))

    def apply_global(self, id_tensors, slots, grad_rows, lr, eps=1e-7):
        """id_tensors[i] indexes table slots[i]; grad_rows = their per-occurrence gradient rows, concatenated.  Gathers
        every rank's (virtual ids, rows) and applies the one global update."""
        k, G = self.k, self.world
        vids = k.concat_offset_ids(list(id_tensors), [self.row_offsets[s] for s in slots])
        n, D = vids.numel(), grad_rows.shape[1]
        if G > 1:
            all_ids = torch.empty(G * n, dtype=vids.dtype, device=vids.device)
            all_rows = torch.empty((G * n, D), dtype=grad_rows.dtype, device=grad_rows.device)
            self.coll.all_gather(all_ids, vids)
            self.coll.all_gather(all_rows, grad_rows.contiguous())
        else:
            all_ids, all_rows = vids, grad_rows
        sorted_vids, perm = k.segment_sort(all_ids, self.row_offsets[-1])
        k.sparse_adagrad_multi(self.tables, self.accums, self.row_offsets, sorted_vids, perm, all_rows, lr, eps)


def replicated_inbatch_step(rep, scene_ids, pos_ids, regularization, global_batch_size, scale, lr):
    """In-batch softmax on replicated towers (rep.tables = [scene, product]): negatives are the local batch, gradients
    are normalised by the GLOBAL batch size (the sum of the per-rank losses is the global mean loss) -- the same
    semantics as sharded.sharded_inbatch_step."""
    k = rep.k
    st, pt = rep.tables
    folded = getattr(k, "inbatch_towers_fwd_bwd", None)
    B = scene_ids.numel()
    if folded is not None and (getattr(k, "TOWERS_ANY_SHAPE", False) or (st.shape[1] <= 128 and B % 128 == 0)):
        loss, _, gq, gc = folded(st, pt, scene_ids, pos_ids, scale, regularization, global_batch_size)
    else:
        q, c = k.gather_rows(st, scene_ids), k.gather_rows(pt, pos_ids)
        loss, _, gq, gc = k.inbatch_softmax_fwd_bwd(q, c, scale, regularization, global_batch_size)
    rep.apply_global([scene_ids, pos_ids], [0, 1], _joined(gq, gc), lr)
    return loss


def replicated_triplet_step(rep, scene_ids, pos_ids, neg_ids, regularization, global_batch_size, lr):
    """The reference triplet loss (pinterest/train_shop_the_look.py:93-109) on replicated towers: G ranks x B triplets
    == one device with G * B triplets and batch_size = G * B."""
    k = rep.k
    st, pt = rep.tables
    B = scene_ids.numel()
    loss, _, _, gs, gp, gn = k.triplet_fwd_bwd(st, pt, pt, scene_ids, pos_ids, neg_ids, B, regularization,
                                               global_batch_size, with_reg=True, want_grads=True, want_scores=False)
    rep.apply_global([scene_ids, pos_ids, neg_ids], [0, 1, 1], _joined(gs, gp, gn), lr)
    return loss
