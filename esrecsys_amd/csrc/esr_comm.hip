// 8e: the row-shard exchange itself -- RCCL grouped send / recv over xGMI, issued on the caller's stream.
//
// The reference is single-device: nothing here replaces reference code.  RCCL is bound at run time
// with dlopen (the process usually has torch's bundled librccl.so.1 loaded already; binding to THAT copy keeps one
// RCCL per process), so libesr_hip.so carries no link-time dependency on it and loads on boxes without RCCL.
//
// On the 8-GPU xGMI full mesh every peer slice of an all-to-all rides its own direct link; the call pattern is
// ncclGroupStart ; per peer ncclSend + ncclRecv ; ncclGroupEnd, in stream order with the kernels around it.
#include <dlfcn.h>
#include <string.h>
#include <mutex>

#include "esr_common.h"

namespace {

struct NcclUid {
  char internal[128];
};
typedef void* nccl_comm_t;
typedef int (*fn_get_uid)(NcclUid*);
typedef int (*fn_init_rank)(nccl_comm_t*, int, NcclUid, int);
typedef int (*fn_sendrecv)(const void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_void)(void);
typedef int (*fn_comm)(nccl_comm_t);
typedef int (*fn_count)(nccl_comm_t, int*);
typedef int (*fn_async)(nccl_comm_t, int*);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  void* handle = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_sendrecv send = nullptr;
  fn_recv recv = nullptr;
  fn_void group_start = nullptr, group_end = nullptr;
  fn_comm destroy = nullptr, abort = nullptr;
  fn_count count = nullptr, user_rank = nullptr;
  fn_async async_error = nullptr;
  fn_errstr errstr = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

struct EsrComm {
  nccl_comm_t comm;
  int world, rank;
};

constexpr int kNcclInt8 = 0;  // ncclInt8 / ncclChar: every exchange is counted in bytes

int bind(const char* path) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return ESR_OK;
  void* h = nullptr;
  if (path && path[0]) {
    h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      esr::set_error("esr_comm_load: dlopen(%s) failed: %s", path, dlerror());
      return ESR_ENODEVICE;
    }
  } else {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // the copy already in the process
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      esr::set_error("esr_comm_load: no librccl.so.1 in the process or on the loader path: %s", dlerror());
      return ESR_ENODEVICE;
    }
  }
  Rccl r;
  r.handle = h;
#define ESR_SYM(field, type, name)                                       \
  r.field = reinterpret_cast<type>(dlsym(h, name));                      \
  if (!r.field) {                                                        \
    esr::set_error("esr_comm_load: %s not exported by librccl", name);   \
    return ESR_ENODEVICE;                                                \
  }
  ESR_SYM(get_uid, fn_get_uid, "ncclGetUniqueId");
  ESR_SYM(init_rank, fn_init_rank, "ncclCommInitRank");
  ESR_SYM(send, fn_sendrecv, "ncclSend");
  ESR_SYM(recv, fn_recv, "ncclRecv");
  ESR_SYM(group_start, fn_void, "ncclGroupStart");
  ESR_SYM(group_end, fn_void, "ncclGroupEnd");
  ESR_SYM(destroy, fn_comm, "ncclCommDestroy");
  ESR_SYM(abort, fn_comm, "ncclCommAbort");
  ESR_SYM(count, fn_count, "ncclCommCount");
  ESR_SYM(user_rank, fn_count, "ncclCommUserRank");
  ESR_SYM(async_error, fn_async, "ncclCommGetAsyncError");
  ESR_SYM(errstr, fn_errstr, "ncclGetErrorString");
#undef ESR_SYM
  g_rccl = r;
  return ESR_OK;
}

int nccl_fail(const char* what, int rc) {
  esr::set_error("%s: %s (ncclResult %d)", what, g_rccl.errstr ? g_rccl.errstr(rc) : "?", rc);
  return ESR_ELAUNCH;
}

// The slice a rank addresses to itself: one copy launch on the stream.  (hipMemcpyAsync moved the same bytes with a blit
// kernel of its own, but the call left 7 - 19 us of idle queue on either side of it -- at world 1 two of them per
// triplet step were a third of the step.)  Unaligned slices (never produced by this package: rows are multiples of 16
// bytes, ids of 4) take the runtime's copy.
__global__ __launch_bounds__(256) void self_copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void self_copy4_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
bool self_copy(void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (bytes == 0 || dst == src) return true;
  const uintptr_t a = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
  if ((a & 15) == 0) {
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(self_copy16_kernel, dim3((unsigned)std::min<size_t>(4096, (n + 255) / 256)), dim3(256), 0, s,
                       (const uint4*)src, (uint4*)dst, n);
    return hipGetLastError() == hipSuccess;
  }
  if ((a & 3) == 0) {
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(self_copy4_kernel, dim3((unsigned)std::min<size_t>(4096, (n + 255) / 256)), dim3(256), 0, s,
                       (const uint32_t*)src, (uint32_t*)dst, n);
    return hipGetLastError() == hipSuccess;
  }
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess;
}

#define ESR_NCCL(call, what)                   \
  do {                                         \
    int rc_ = (call);                          \
    if (rc_ != 0) return nccl_fail(what, rc_); \
  } while (0)

// One all-to-all(v): slice p of `send` (send_bytes[p] bytes) goes to peer p, slice p of `recv` comes from it.
// Every argument is validated BEFORE ncclGroupStart, so nothing can leave a group open.
int alltoall(const char* who, esr_comm_t comm_, const void* send, const int64_t* send_counts, void* recv,
             const int64_t* recv_counts, int64_t unit, esr_stream_t stream) {
  EsrComm* c = reinterpret_cast<EsrComm*>(comm_);
  ESR_REQUIRE(c && c->comm, "%s: null communicator", who);
  ESR_REQUIRE(send_counts && recv_counts, "%s: null count arrays (host int64 [world])", who);
  ESR_REQUIRE(unit > 0, "%s: element size must be positive", who);
  int64_t st = 0, rt = 0;
  for (int p = 0; p < c->world; ++p) {
    ESR_REQUIRE(send_counts[p] >= 0 && recv_counts[p] >= 0, "%s: negative count for peer %d", who, p);
    st += send_counts[p];
    rt += recv_counts[p];
  }
  ESR_REQUIRE(st == 0 || send, "%s: null send buffer", who);
  ESR_REQUIRE(rt == 0 || recv, "%s: null recv buffer", who);
  if (st == 0 && rt == 0) return ESR_OK;
  ESR_REQUIRE(send_counts[c->rank] == recv_counts[c->rank], "%s: the rank sends itself %lld and expects %lld", who,
              (long long)send_counts[c->rank], (long long)recv_counts[c->rank]);
  hipStream_t s = esr::as_stream(stream);
  const char* sp = static_cast<const char*>(send);
  char* rp = static_cast<char*>(recv);
  // the slice addressed to this rank itself never enters RCCL (see alltoall_multi): one copy launch on the stream
  {
    size_t so = 0, ro = 0;
    for (int p = 0; p < c->rank; ++p) {
      so += (size_t)(send_counts[p] * unit);
      ro += (size_t)(recv_counts[p] * unit);
    }
    const size_t self = (size_t)(send_counts[c->rank] * unit);
    if (self && !self_copy(rp + ro, sp + so, self, s)) {
      esr::set_error("%s: self copy failed", who);
      return ESR_ELAUNCH;
    }
    if (st == send_counts[c->rank] && rt == recv_counts[c->rank]) return ESR_OK;  // nothing for anybody else
  }
  ESR_NCCL(g_rccl.group_start(), who);
  int first = 0;
  for (int p = 0; p < c->world; ++p) {
    const size_t sb = (size_t)(send_counts[p] * unit), rb = (size_t)(recv_counts[p] * unit);
    if (p != c->rank) {
      if (sb && !first) first = g_rccl.send(sp, sb, kNcclInt8, p, c->comm, s);
      if (rb && !first) first = g_rccl.recv(rp, rb, kNcclInt8, p, c->comm, s);
    }
    sp += sb;
    rp += rb;
  }
  const int end = g_rccl.group_end();  // always closed, also after a failed send / recv
  if (first) return nccl_fail(who, first);
  if (end) return nccl_fail(who, end);
  return ESR_OK;
}

// Several all-to-all(v)s as ONE RCCL group (one kernel): operation o moves slice p of send[o] (send_bytes[o * world + p]
// bytes) to peer p and receives slice p of recv[o] from it.  Sends and receives between a pair of ranks match in issue
// order, which is the same on every rank: operation by operation, peer by peer.
int alltoall_multi(const char* who, esr_comm_t comm_, int n_ops, const void* const* send, const int64_t* send_bytes,
                   void* const* recv, const int64_t* recv_bytes, esr_stream_t stream) {
  EsrComm* c = reinterpret_cast<EsrComm*>(comm_);
  ESR_REQUIRE(c && c->comm, "%s: null communicator", who);
  ESR_REQUIRE(n_ops >= 0 && (n_ops == 0 || (send && recv && send_bytes && recv_bytes)), "%s: null arrays", who);
  bool any = false, remote = false;
  for (int o = 0; o < n_ops; ++o) {
    int64_t st = 0, rt = 0;
    for (int p = 0; p < c->world; ++p) {
      const int64_t sb = send_bytes[(size_t)o * c->world + p], rb = recv_bytes[(size_t)o * c->world + p];
      ESR_REQUIRE(sb >= 0 && rb >= 0, "%s: negative count for peer %d of operation %d", who, p, o);
      st += sb;
      rt += rb;
    }
    ESR_REQUIRE(st == 0 || send[o], "%s: null send buffer of operation %d", who, o);
    ESR_REQUIRE(rt == 0 || recv[o], "%s: null recv buffer of operation %d", who, o);
    ESR_REQUIRE(send_bytes[(size_t)o * c->world + c->rank] == recv_bytes[(size_t)o * c->world + c->rank],
                "%s: operation %d sends itself %lld bytes and expects %lld", who, o,
                (long long)send_bytes[(size_t)o * c->world + c->rank], (long long)recv_bytes[(size_t)o * c->world + c->rank]);
    any = any || st > 0 || rt > 0;
    remote = remote || st > send_bytes[(size_t)o * c->world + c->rank] || rt > recv_bytes[(size_t)o * c->world + c->rank];
  }
  if (!any) return ESR_OK;
  hipStream_t s = esr::as_stream(stream);
  // The slice a rank addresses to itself never enters RCCL: a send / recv pair to self is a copy kernel that measured
  // 0.9 TB/s (2 x 134 MB of GloVe rows and gradients in 0.29 ms at world 1); a plain copy launch on the same stream moves
  // it at the device's copy rate and, at world 1, no RCCL kernel is launched at all.
  for (int o = 0; o < n_ops; ++o) {
    size_t so = 0, ro = 0;
    for (int p = 0; p < c->rank; ++p) {
      so += (size_t)send_bytes[(size_t)o * c->world + p];
      ro += (size_t)recv_bytes[(size_t)o * c->world + p];
    }
    const size_t self = (size_t)send_bytes[(size_t)o * c->world + c->rank];
    if (self) {
      if (!self_copy(static_cast<char*>(recv[o]) + ro, static_cast<const char*>(send[o]) + so, self, s)) {
        esr::set_error("%s: self copy failed", who);
        return ESR_ELAUNCH;
      }
    }
  }
  if (!remote) return ESR_OK;
  ESR_NCCL(g_rccl.group_start(), who);
  int first = 0;
  for (int o = 0; o < n_ops; ++o) {
    const char* sp = static_cast<const char*>(send[o]);
    char* rp = static_cast<char*>(recv[o]);
    for (int p = 0; p < c->world; ++p) {
      const size_t sb = (size_t)send_bytes[(size_t)o * c->world + p], rb = (size_t)recv_bytes[(size_t)o * c->world + p];
      if (p != c->rank) {
        if (sb && !first) first = g_rccl.send(sp, sb, kNcclInt8, p, c->comm, s);
        if (rb && !first) first = g_rccl.recv(rp, rb, kNcclInt8, p, c->comm, s);
      }
      sp += sb;
      rp += rb;
    }
  }
  const int end = g_rccl.group_end();  // always closed, also after a failed send / recv
  if (first) return nccl_fail(who, first);
  if (end) return nccl_fail(who, end);
  return ESR_OK;
}

}  // namespace

extern "C" {

int esr_comm_load(const char* librccl_path) { return bind(librccl_path); }

int esr_comm_unique_id(void* uid128) {
  ESR_REQUIRE(uid128, "esr_comm_unique_id: null output (host, 128 bytes)");
  if (int rc = bind(nullptr)) return rc;
  NcclUid uid;
  ESR_NCCL(g_rccl.get_uid(&uid), "esr_comm_unique_id");
  memcpy(uid128, uid.internal, sizeof(uid.internal));
  return ESR_OK;
}

int esr_comm_init(const void* uid128, int world, int rank, esr_comm_t* comm) {
  ESR_REQUIRE(uid128 && comm, "esr_comm_init: null argument");
  ESR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "esr_comm_init: need 0 <= rank < world (got %d, %d)", rank,
              world);
  if (int rc = bind(nullptr)) return rc;
  NcclUid uid;
  memcpy(uid.internal, uid128, sizeof(uid.internal));
  nccl_comm_t nc = nullptr;
  ESR_NCCL(g_rccl.init_rank(&nc, world, uid, rank), "esr_comm_init");
  int seen = 0;
  ESR_NCCL(g_rccl.count(nc, &seen), "esr_comm_init (ncclCommCount)");
  if (seen != world) {
    g_rccl.abort(nc);
    esr::set_error("esr_comm_init: communicator has %d ranks, expected %d", seen, world);
    return ESR_ELAUNCH;
  }
  EsrComm* c = new EsrComm{nc, world, rank};
  *comm = c;
  return ESR_OK;
}

int esr_comm_count(esr_comm_t comm, int* world, int* rank) {
  EsrComm* c = reinterpret_cast<EsrComm*>(comm);
  ESR_REQUIRE(c && c->comm, "esr_comm_count: null communicator");
  int w = 0, r = 0;
  ESR_NCCL(g_rccl.count(c->comm, &w), "esr_comm_count");
  ESR_NCCL(g_rccl.user_rank(c->comm, &r), "esr_comm_count (ncclCommUserRank)");
  if (world) *world = w;
  if (rank) *rank = r;
  return ESR_OK;
}

int esr_comm_async_error(esr_comm_t comm) {
  EsrComm* c = reinterpret_cast<EsrComm*>(comm);
  ESR_REQUIRE(c && c->comm, "esr_comm_async_error: null communicator");
  int err = 0;
  ESR_NCCL(g_rccl.async_error(c->comm, &err), "esr_comm_async_error");
  if (err) return nccl_fail("esr_comm_async_error: asynchronous failure", err);
  return ESR_OK;
}

int esr_comm_abort(esr_comm_t comm) {
  EsrComm* c = reinterpret_cast<EsrComm*>(comm);
  if (!c) return ESR_OK;
  int rc = c->comm ? g_rccl.abort(c->comm) : 0;
  delete c;
  if (rc) return nccl_fail("esr_comm_abort", rc);
  return ESR_OK;
}

int esr_comm_destroy(esr_comm_t comm) {
  EsrComm* c = reinterpret_cast<EsrComm*>(comm);
  if (!c) return ESR_OK;
  int rc = c->comm ? g_rccl.destroy(c->comm) : 0;
  delete c;
  if (rc) return nccl_fail("esr_comm_destroy", rc);
  return ESR_OK;
}

int esr_alltoall_bytes(esr_comm_t comm, const void* send, const int64_t* send_bytes, void* recv,
                       const int64_t* recv_bytes, esr_stream_t stream) {
  return alltoall("esr_alltoall_bytes", comm, send, send_bytes, recv, recv_bytes, 1, stream);
}

int esr_alltoall_bytes_multi(esr_comm_t comm, int n_ops, const void* const* send, const int64_t* send_bytes,
                             void* const* recv, const int64_t* recv_bytes, esr_stream_t stream) {
  return alltoall_multi("esr_alltoall_bytes_multi", comm, n_ops, send, send_bytes, recv, recv_bytes, stream);
}

int esr_allgather_bytes(esr_comm_t comm_, const void* send, int64_t bytes, void* recv, esr_stream_t stream) {
  EsrComm* c = reinterpret_cast<EsrComm*>(comm_);
  ESR_REQUIRE(c && c->comm, "esr_allgather_bytes: null communicator");
  ESR_REQUIRE(bytes >= 0, "esr_allgather_bytes: negative size");
  if (bytes == 0) return ESR_OK;
  ESR_REQUIRE(send && recv, "esr_allgather_bytes: null buffer");
  hipStream_t s = esr::as_stream(stream);
  char* rp = static_cast<char*>(recv);
  // this rank's own block is a plain copy; every other block is one send / recv pair of the same grouped call the
  // all-to-alls use (on the xGMI full mesh each pair rides its own link, exactly like a slice of an all-to-all)
  if (rp + (size_t)c->rank * bytes != send && !self_copy(rp + (size_t)c->rank * bytes, send, (size_t)bytes, s)) {
    esr::set_error("esr_allgather_bytes: self copy failed");
    return ESR_ELAUNCH;
  }
  if (c->world == 1) return ESR_OK;
  ESR_NCCL(g_rccl.group_start(), "esr_allgather_bytes");
  int first = 0;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    if (!first) first = g_rccl.send(send, (size_t)bytes, kNcclInt8, p, c->comm, s);
    if (!first) first = g_rccl.recv(rp + (size_t)p * bytes, (size_t)bytes, kNcclInt8, p, c->comm, s);
  }
  const int end = g_rccl.group_end();
  if (first) return nccl_fail("esr_allgather_bytes", first);
  if (end) return nccl_fail("esr_allgather_bytes", end);
  return ESR_OK;
}

int esr_alltoall_ids(esr_comm_t comm, const int32_t* send_ids, const int64_t* send_counts, int32_t* recv_ids,
                     const int64_t* recv_counts, esr_stream_t stream) {
  return alltoall("esr_alltoall_ids", comm, send_ids, send_counts, recv_ids, recv_counts, 4, stream);
}

int esr_alltoall_rows(esr_comm_t comm, const void* send_rows, int dtype, int D, const int64_t* send_counts,
                      void* recv_rows, const int64_t* recv_counts, esr_stream_t stream) {
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_alltoall_rows: dtype must be ESR_F32 or ESR_BF16");
  ESR_REQUIRE(D > 0, "esr_alltoall_rows: D must be positive");
  return alltoall("esr_alltoall_rows", comm, send_rows, send_counts, recv_rows, recv_counts,
                  (int64_t)D * (dtype == ESR_BF16 ? 2 : 4), stream);
}

int esr_alltoall_grads(esr_comm_t comm, const float* send_grads, int D, const int64_t* send_counts,
                       float* recv_grads, const int64_t* recv_counts, esr_stream_t stream) {
  ESR_REQUIRE(D > 0, "esr_alltoall_grads: D must be positive");
  return alltoall("esr_alltoall_grads", comm, send_grads, send_counts, recv_grads, recv_counts, (int64_t)D * 4,
                  stream);
}

}  // extern "C"
