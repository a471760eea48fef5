"""All-to-all(v) straight on the compute stream: RCCL's grouped send / recv issued by libesr_hip.so
(``esr_alltoall_*``, include/esr_hip.h) on torch's CURRENT HIP stream through its own communicator.

torch.distributed's ProcessGroupNCCL runs every collective on an internal stream: each call costs two stream
hand-overs (event record / wait in each direction, ~10-15 us of idle GPU apiece).  A row-sharded step has four
collectives between short kernels, so those hand-overs were ~100 us of a 0.53 ms step.  Issued in stream order the
exchange needs none.  The communicator is bootstrapped through the existing process group (rank 0's ncclUniqueId is
broadcast as bytes), and the call pattern -- ncclGroupStart; per peer ncclSend + ncclRecv; ncclGroupEnd -- is the
one ProcessGroupNCCL itself uses for all_to_all_single.  ``ESR_RCCL_DIRECT=0`` keeps everything on torch.distributed.

Bootstrap is three phases, each closed by a MIN all-reduce over the torch process group so that every rank takes
the same path: (1) bind librccl (local, can fail on one rank only), (2) ncclCommInitRank (collective), (3) a
self-test all-to-all with known contents, WAITED FOR WITH A TIMEOUT: a rank whose peers never post their half
aborts its communicator (ncclCommAbort terminates the enqueued operations) instead of hanging in a device sync,
so the agreement all-reduce that follows is always reached.

``ESR_RCCL_LIB=/path/to/lib.so`` binds that library instead of torch's librccl (any library exporting the twelve
nccl* symbols esr_comm.hip binds).  With it set the direct exchange is also built over a process group that is not
"nccl" (the bootstrap bytes then travel as CPU tensors).  Used by tests/test_gpu_rccl_world2.py to run the world-2
branches of the library as two processes on ONE GPU over tests/wire's loopback wire; nothing in the package sets it.
"""
import ctypes
import os
import time

import torch
import torch.distributed as dist

from . import _lib


class DirectExchange:
    """An RCCL communicator over the ranks of `group`, used for all_to_all_single on the current stream."""

    def __init__(self, group=None, device=None, selftest_timeout_s=None):
        self.lib = _lib.load()
        self.pg = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.comm = ctypes.c_void_p()
        self.timeout = float(os.environ.get("ESR_RCCL_SELFTEST_TIMEOUT", "60")) if selftest_timeout_s is None \
            else float(selftest_timeout_s)
        self._cnt = (ctypes.c_int64 * self.world)
        # the bootstrap's own collectives run on the process group: device tensors under nccl, host tensors otherwise
        self._boot_dev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        # phase 1: bind librccl -- torch's bundled copy, the one ProcessGroupNCCL already runs on
        path = os.environ.get("ESR_RCCL_LIB") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._agree(lambda: _lib.check(self.lib.esr_comm_load(path.encode() if os.path.exists(path) else None),
                                       "esr_comm_load"), "bind librccl")
        # phase 2: the communicator (collective)
        uid = (ctypes.c_byte * 128)()
        if self.rank == 0:
            _lib.check(self.lib.esr_comm_unique_id(uid), "esr_comm_unique_id")
        raw = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(self._boot_dev)
        dist.broadcast(raw, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(uid, raw.cpu().numpy().tobytes(), 128)

        def init():
            with torch.cuda.device(self.device):  # the communicator binds to the calling thread's current HIP device
                _lib.check(self.lib.esr_comm_init(uid, self.world, self.rank, ctypes.byref(self.comm)),
                           "esr_comm_init")
        self._agree(init, "ncclCommInitRank")
        # phase 3: one uneven all-to-all with known contents, bounded in time
        self._agree(self._self_test, "self-test")

    def _agree(self, fn, what):
        """Run fn on this rank; every rank learns whether ALL ranks succeeded.  Raises on every rank if any failed."""
        err = None
        try:
            fn()
        except Exception as e:  # noqa: BLE001 -- whatever went wrong locally must reach the agreement below
            err = e
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self._boot_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.pg)
        if int(ok) != 1:
            self._abort()
            raise RuntimeError("direct RCCL exchange: %s failed on %s" %
                               (what, "this rank (%s)" % err if err is not None else "another rank"))

    def _abort(self):
        if self.comm:
            self.lib.esr_comm_abort(self.comm)
            self.comm = ctypes.c_void_p()

    def _self_test(self):
        """Rank r sends (peer + 1) rows of value 1000 r + peer to `peer`; completion is polled, never waited on."""
        G, r = self.world, self.rank
        send_rows = [p + 1 for p in range(G)]
        recv_rows = [r + 1] * G
        send = torch.cat([torch.full((p + 1, 3), 1000 * r + p, dtype=torch.int32) for p in range(G)]).to(self.device)
        recv = torch.full((sum(recv_rows), 3), -1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self.all_to_all_single(recv, send, recv_rows, send_rows)
            done = torch.cuda.Event()
            done.record()
            t0 = time.monotonic()
            while not done.query():
                if time.monotonic() - t0 > self.timeout:
                    self._abort()  # terminates the enqueued send / recv: the stream drains, nobody hangs
                    raise RuntimeError("self-test all-to-all did not complete within %.0f s" % self.timeout)
                time.sleep(0.001)
            _lib.check(self.lib.esr_comm_async_error(self.comm), "esr_comm_async_error")
        want = torch.cat([torch.full((r + 1, 3), 1000 * p + r, dtype=torch.int32) for p in range(G)])
        if not torch.equal(recv.cpu(), want):
            raise RuntimeError("direct all-to-all self-test returned wrong data")
        # the same exchange twice inside ONE group (all_to_all_multi: the ids exchanges of a group of routing plans)
        send2 = send + 7
        r1, r2 = torch.full_like(recv, -1), torch.full_like(recv, -1)
        with torch.cuda.device(self.device):
            self.all_to_all_multi([(r1, send, recv_rows, send_rows), (r2, send2, recv_rows, send_rows)])
            done = torch.cuda.Event()
            done.record()
            t0 = time.monotonic()
            while not done.query():
                if time.monotonic() - t0 > self.timeout:
                    self._abort()
                    raise RuntimeError("self-test multi all-to-all did not complete within %.0f s" % self.timeout)
                time.sleep(0.001)
            _lib.check(self.lib.esr_comm_async_error(self.comm), "esr_comm_async_error")
        if not (torch.equal(r1.cpu(), want) and torch.equal(r2.cpu(), want + 7)):
            raise RuntimeError("direct multi all-to-all self-test returned wrong data")

    def ranks_seen(self):
        """(world, rank) as RCCL itself reports them for this communicator (ncclCommCount / ncclCommUserRank)."""
        w, r = ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.esr_comm_count(self.comm, ctypes.byref(w), ctypes.byref(r)), "esr_comm_count")
        return w.value, r.value

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        """Same contract as torch.distributed.all_to_all_single (splits count rows of dim 0), on the current stream."""
        G = self.world
        row = inp.element_size()
        for d in inp.shape[1:]:
            row *= int(d)
        if in_splits is None:
            in_splits = [inp.shape[0] // G] * G
        if out_splits is None:
            out_splits = [out.shape[0] // G] * G
        sb = self._cnt(*[int(s) * row for s in in_splits])
        rb = self._cnt(*[int(s) * row for s in out_splits])
        stream = torch._C._cuda_getCurrentRawStream(self.device.index)
        _lib.check(self.lib.esr_alltoall_bytes(self.comm, inp.data_ptr(), sb, out.data_ptr(), rb, stream),
                   "esr_alltoall_bytes")
        return out

    def all_to_all_multi(self, ops_):
        """Several all_to_all_single calls -- [(out, inp, out_splits, in_splits), ...] -- as ONE RCCL group on the current
        stream (one kernel instead of len(ops_)): the ids exchanges of a group of routing plans."""
        G, n = self.world, len(ops_)
        if n == 0:
            return
        sp, rp = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        sb, rb = (ctypes.c_int64 * (n * G))(), (ctypes.c_int64 * (n * G))()
        for o, (out, inp, out_splits, in_splits) in enumerate(ops_):
            row = inp.element_size()
            for d in inp.shape[1:]:
                row *= int(d)
            if in_splits is None:
                in_splits = [inp.shape[0] // G] * G
            if out_splits is None:
                out_splits = [out.shape[0] // G] * G
            sp[o], rp[o] = inp.data_ptr(), out.data_ptr()
            for p in range(G):
                sb[o * G + p] = int(in_splits[p]) * row
                rb[o * G + p] = int(out_splits[p]) * row
        stream = torch._C._cuda_getCurrentRawStream(self.device.index)
        _lib.check(self.lib.esr_alltoall_bytes_multi(self.comm, n, sp, sb, rp, rb, stream), "esr_alltoall_bytes_multi")

    def all_gather_into_tensor(self, out, inp):
        """Same contract as torch.distributed.all_gather_into_tensor (equal blocks), on the current stream."""
        stream = torch._C._cuda_getCurrentRawStream(self.device.index)
        _lib.check(self.lib.esr_allgather_bytes(self.comm, inp.data_ptr(), inp.numel() * inp.element_size(),
                                                out.data_ptr(), stream), "esr_allgather_bytes")
        return out

    def close(self):
        if self.comm:
            self.lib.esr_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()


_cache = {}


def reset():
    """Destroy every cached communicator (call before dist.destroy_process_group(): a later process group may reuse
    the cache key)."""
    for x in _cache.values():
        if x is not None:
            x.close()
    _cache.clear()


def exchange_for(group, device, lane=0):
    """The DirectExchange of (group, device), or None when disabled / unavailable (then use torch.distributed).
    Every rank returns the same answer: the bootstrap phases agree through the process group.  lane 1 is a SECOND
    communicator over the same ranks: collectives of one communicator are serialised, so the rows exchange that a
    training loop runs on a side stream under the previous step's kernels (sharded.sharded_train_steps, overlap) needs
    its own."""
    if os.environ.get("ESR_RCCL_DIRECT", "1") != "1" or device.type != "cuda" or \
            (dist.get_backend(group) != "nccl" and not os.environ.get("ESR_RCCL_LIB")):
        return None
    key = (id(group), device.index, lane)
    if key not in _cache:
        try:
            _cache[key] = DirectExchange(group, device)
        except Exception as e:  # noqa: BLE001 -- a failed bootstrap must not take the step down
            print("esrecsys_amd.rccl: direct exchange unavailable on rank %d (%s); using torch.distributed"
                  % (dist.get_rank(group), e))
            _cache[key] = None
    return _cache[key]
