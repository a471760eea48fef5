"""All-to-all(v) straight on the compute stream: RCCL's grouped send / recv issued by this process on torch's
CURRENT HIP stream through its own communicator.

torch.distributed's ProcessGroupNCCL runs every collective on an internal stream: each call costs two stream
hand-overs (event record / wait in each direction, ~10-15 us of idle GPU apiece).  A row-sharded step has four
collectives between short kernels, so those hand-overs were ~100 us of a 0.53 ms step.  Issued in stream order the
exchange needs none.  The communicator is bootstrapped through the existing process group (rank 0's ncclUniqueId is
broadcast as bytes), and the call pattern -- ncclGroupStart; per peer ncclSend + ncclRecv; ncclGroupEnd -- is the
one ProcessGroupNCCL itself uses for all_to_all_single.  ``ESR_RCCL_DIRECT=0`` keeps everything on torch.distributed.
"""
import ctypes
import os

import torch
import torch.distributed as dist

_NCCL_INT8 = 0  # ncclInt8 / ncclChar: the exchange is counted in bytes, whatever the element type


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


def _load():
    lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclSend.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                             ctypes.c_void_p]
    lib.ncclRecv.argtypes = lib.ncclSend.argtypes
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd",
              "ncclCommDestroy"):
        getattr(lib, f).restype = ctypes.c_int
    return lib


class DirectExchange:
    """An RCCL communicator over the ranks of `group`, used for all_to_all_single on the current stream."""

    def __init__(self, group=None, device=None):
        self.lib = _load()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        raw = torch.frombuffer(bytearray(bytes(uid.internal)), dtype=torch.uint8).to(self.device)
        dist.broadcast(raw, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(ctypes.byref(uid), bytes(raw.cpu().numpy().tobytes()), 128)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):  # the communicator binds to the calling thread's current HIP device
            self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank),
                        "ncclCommInitRank")
            self._self_test()

    def _self_test(self):
        """One uneven all-to-all with known contents: rank r sends (peer + 1) rows of value 1000 r + peer to `peer`."""
        G, r = self.world, self.rank
        send_rows = [p + 1 for p in range(G)]
        recv_rows = [r + 1] * G
        send = torch.cat([torch.full((p + 1, 3), 1000 * r + p, dtype=torch.int32) for p in range(G)]).to(self.device)
        recv = torch.full((sum(recv_rows), 3), -1, dtype=torch.int32, device=self.device)
        self.all_to_all_single(recv, send, recv_rows, send_rows)
        want = torch.cat([torch.full((r + 1, 3), 1000 * p + r, dtype=torch.int32) for p in range(G)])
        if not torch.equal(recv.cpu(), want):
            raise RuntimeError("direct all-to-all self-test returned wrong data")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

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
        stream = torch._C._cuda_getCurrentRawStream(self.device.index)
        sp, rp = inp.data_ptr(), out.data_ptr()
        lib, comm = self.lib, self.comm
        self._check(lib.ncclGroupStart(), "ncclGroupStart")
        so = ro = 0
        for peer in range(G):
            sb, rb = in_splits[peer] * row, out_splits[peer] * row
            if sb:
                self._check(lib.ncclSend(sp + so, sb, _NCCL_INT8, peer, comm, stream), "ncclSend")
            if rb:
                self._check(lib.ncclRecv(rp + ro, rb, _NCCL_INT8, peer, comm, stream), "ncclRecv")
            so += sb
            ro += rb
        self._check(lib.ncclGroupEnd(), "ncclGroupEnd")
        return out

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


_cache = {}


def exchange_for(group, device):
    """The DirectExchange of (group, device), or None when disabled / unavailable (then use torch.distributed)."""
    if os.environ.get("ESR_RCCL_DIRECT", "1") != "1" or device.type != "cuda" or dist.get_backend(group) != "nccl":
        return None
    key = (id(group), device.index)
    if key not in _cache:
        x = None
        try:
            x = DirectExchange(group, device)
        except Exception as e:  # a missing symbol / failed bootstrap / wrong self-test data must not take the step down
            print("esrecsys_amd.rccl: direct exchange unavailable on rank %d (%s)" % (dist.get_rank(group), e))
        # every rank must take the same path: one failing rank sends all of them back to torch.distributed
        ok = torch.tensor([1 if x is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        _cache[key] = x if int(ok) == 1 else None
    return _cache[key]
