"""IVF (inverted-file) approximate retrieval -- BASELINE config 5's "top-k ANN scoring vs brute-force".

Build-defined: the reference has no ANN index (its retrieval is the exact ``jax.lax.top_k`` over all candidates,
pinterest/make_recommendations.py:49-65, which ``find_top_k`` / ``ops.retrieve_topk`` reproduce and which stays the
yardstick: ``recall_at_k`` of make_recommendations.py measures this index against it).

Index: spherical k-means over (a sample of) the candidates -- assignment by the MFMA score GEMM + select
(``ops.retrieve_topk`` with k = 1), centroid sums by the sort + segment-sum kernels, list boundaries and unit-length
centroids by ``esr_run_offsets`` / ``esr_ivf_centroids`` (torch only draws the random samples) -- then the candidates
grouped by list.  Search: the nprobe best centroids per query (``ops.retrieve_topk``), then exact f32 scores inside those lists only
(``esr_ivf_search``: grouped FP32 GEMM, radix select per (query, list), merge): 2 nq nprobe (N / nlist) D flop instead
of 2 nq N D.
"""
import torch

from . import _lib, ops


class IVFIndex:
    """candidates [N, D] f32 on the device -> an index answering ``search(queries, k, nprobe)``."""

    def __init__(self, candidates, nlist, iters=6, seed=0, train_rows=None):
        c = ops._req(candidates, torch.float32, "candidates")
        N, D = c.shape
        nlist = int(min(nlist, N))
        if not 1 <= nlist <= 32768:
            raise ValueError("IVFIndex: nlist must be in [1, 32768], got %d" % nlist)
        if D % 4:
            raise ValueError("IVFIndex needs D % 4 == 0")
        g = torch.Generator(device=c.device).manual_seed(seed)
        n_train = int(min(N, train_rows if train_rows is not None else 64 * nlist))
        train = c if n_train == N else \
            ops.gather_rows(c, torch.randperm(N, generator=g, device=c.device)[:n_train].to(torch.int32))
        pick = torch.randperm(n_train, generator=g, device=c.device)[:nlist].to(torch.int32)
        cent = ops.ivf_centroids(ops.gather_rows(train, pick))
        for _ in range(iters):
            _, a = ops.retrieve_topk(train, cent, 1, mode="exact")
            sorted_a, perm = ops.segment_sort(a.reshape(-1).contiguous(), nlist)
            sums = ops.rows_to_dense(nlist, D, sorted_a, perm, train.clone())   # (the segment sum parks partials in its input)
            # list sizes from the sorted assignments; an empty list takes a random training row (drawn for every list:
            # no read-back to learn which ones are empty)
            off = ops.run_offsets(sorted_a, nlist)
            fallback = torch.randint(0, n_train, (nlist,), generator=g, device=c.device, dtype=torch.int32)
            cent = ops.ivf_centroids(sums, off, train, fallback)
        self.centroids = cent.contiguous()
        _, a = ops.retrieve_topk(c, self.centroids, 1, mode="exact")
        sorted_a, perm = ops.segment_sort(a.reshape(-1).contiguous(), nlist)
        self.list_off, longest = ops.run_offsets(sorted_a, nlist, want_max=True)
        self.orig = perm.contiguous()
        self.cands_sorted = ops.gather_rows(c, perm)
        self.nlist, self.N, self.D = nlist, N, D
        self.max_list = int(longest.item())   # (one read-back, at build time)

    def search(self, queries, k, nprobe):
        """([nq, k] scores, [nq, k] candidate rows), best first; exact f32 scores of the candidates in each query's
        nprobe best lists (entries the lists could not fill: -inf / -1)."""
        q = ops._req(queries, torch.float32, "queries")
        nq, D = q.shape
        if D != self.D:
            raise ValueError("query width %d != index width %d" % (D, self.D))
        nprobe = int(min(nprobe, self.nlist))
        _, lists = ops.retrieve_topk(q, self.centroids, nprobe, mode="exact")
        lib = _lib.load()
        out_s = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        out_i = torch.empty((nq, k), dtype=torch.int32, device=q.device)
        ws = ops._ws(ops._ws_bytes("esr_ivf_search_workspace_bytes", nq, self.nlist, self.max_list, nprobe, int(k)), q.device)
        _lib.check(lib.esr_ivf_search(q.data_ptr(), nq, D, self.cands_sorted.data_ptr(), self.list_off.data_ptr(),
                                      self.orig.data_ptr(), self.nlist, self.max_list, lists.data_ptr(), nprobe, int(k),
                                      out_s.data_ptr(), out_i.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
                   "esr_ivf_search")
        return out_s, out_i
