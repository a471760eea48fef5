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

    def gather_ids(self, id_tensors, slots):
        """Every rank's occurrence ids of one batch as virtual rows, sorted: (sorted ids, permutation, ids per rank).
        Depends on the ids only -- tables indexed by the same ids (GloVe's embedding and bias) share one."""
        k, G = self.k, self.world
        vids = k.concat_offset_ids(list(id_tensors), [self.row_offsets[s] for s in slots])
        n = vids.numel()
        all_ids = vids
        if G > 1:
            all_ids = torch.empty(G * n, dtype=vids.dtype, device=vids.device)
            self.coll.all_gather(all_ids, vids)
        sorted_vids, perm = k.segment_sort(all_ids, self.row_offsets[-1])
        return sorted_vids, perm, n

    def apply_rows(self, gathered, grad_rows, lr, eps=1e-7):
        """All-gather this rank's per-occurrence gradient rows and apply the ONE global update (`gathered` from
        gather_ids of this or of an identically indexed ReplicatedTables)."""
        k, G = self.k, self.world
        sorted_vids, perm, n = gathered
        all_rows = grad_rows
        if G > 1:
            all_rows = torch.empty((G * n, grad_rows.shape[1]), dtype=grad_rows.dtype, device=grad_rows.device)
            self.coll.all_gather(all_rows, grad_rows.contiguous())
        if len(self.tables) == 1:  # (any row width: the GloVe bias column is one float)
            k.sparse_adagrad(self.tables[0], self.accums[0], sorted_vids, perm, all_rows, lr, eps)
        else:
            k.sparse_adagrad_multi(self.tables, self.accums, self.row_offsets, sorted_vids, perm, all_rows, lr, eps)

    def apply_global(self, id_tensors, slots, grad_rows, lr, eps=1e-7):
        """id_tensors[i] indexes table slots[i]; grad_rows = their per-occurrence gradient rows, concatenated.  Gathers
        every rank's (virtual ids, rows) and applies the one global update."""
        self.apply_rows(self.gather_ids(id_tensors, slots), grad_rows, lr, eps)


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


def replicated_glove_step(rep_emb, rep_bias, inputs, target, mode, lr):
    """GloVe step (wikipedia/train_cooccurence.py:71-101 with the build's sparse Adagrad) on a replicated embedding table
    and its [V, 1] bias table (two single-table ReplicatedTables over the same group): the loss is over this rank's
    batch, as in sharded.sharded_glove_step; every rank's occurrence ids are gathered and sorted ONCE and both tables take
    their one global update from it."""
    k = rep_emb.k
    ids = inputs.reshape(-1)
    gathered = rep_emb.gather_ids([ids], [0])   # (before the loss kernel: the sort needs the ids only)
    loss, grad_rows, grad_bias = k.glove_fwd_bwd(rep_emb.tables[0], rep_bias.tables[0], inputs, target, mode)
    rep_emb.apply_rows(gathered, grad_rows, lr)
    rep_bias.apply_rows(gathered, grad_bias.reshape(-1, 1), lr)
    return loss
