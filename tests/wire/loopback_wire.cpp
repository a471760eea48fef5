// TEST INFRASTRUCTURE -- not part of the product, never loaded unless ESR_RCCL_LIB points at it.
//
// A loopback wire with RCCL's send / recv surface, so that the world > 1 branches of libesr_hip.so (esr_comm.hip's
// grouped exchanges, esr_shard_step.hip's whole sharded steps, the overlapped loop's second communicator) can run with
// the REAL HIP kernels as two or more processes sharing ONE GPU.  RCCL itself refuses two ranks on one device, and the
// boxes the tests run on have one.  Only the wire is substituted: every byte the library hands to ncclSend reaches the
// peer's ncclRecv buffer, in issue order per pair of ranks, exactly as RCCL's grouped point-to-point calls deliver them.
//
// How: ranks connect pairwise over abstract Unix-domain sockets named after the unique id.  ncclGroupEnd waits for the
// stream (everything queued before the exchange has produced its send buffers), a helper thread copies each send slice
// to the host and writes it to the peer's socket while the calling thread reads its receive slices and copies them to
// the device; the call returns when both are done (blocking: later launches on the stream see the received bytes).
// Every message carries its length; a mismatch with the posted receive is reported as ncclInvalidUsage -- the check
// RCCL cannot make.  Reads time out (ESR_WIRE_TIMEOUT_S, default 120 s) instead of hanging a GPU box.
//
// Exports exactly the twelve symbols esr_comm.hip binds (esr_comm.hip: ESR_SYM list).
#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <thread>
#include <vector>

namespace {

enum { kOk = 0, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

struct Uid {
  char b[128];
};
struct Comm {
  int world = 0, rank = 0, listen_fd = -1;
  std::vector<int> fd;
};
struct Op {
  Comm* c;
  bool send;
  void* p;
  size_t n;
  int peer;
  hipStream_t s;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

int timeout_ms() {
  const char* e = getenv("ESR_WIRE_TIMEOUT_S");
  return (e && atoi(e) > 0 ? atoi(e) : 120) * 1000;
}

socklen_t make_addr(sockaddr_un* a, const char* uid, int rank) {
  memset(a, 0, sizeof(*a));
  a->sun_family = AF_UNIX;
  const int n = snprintf(a->sun_path + 1, sizeof(a->sun_path) - 1, "%s-%d", uid, rank);  // abstract: sun_path[0] = 0
  return (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + n);
}

bool write_all(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n) {
    const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    c += w;
    n -= (size_t)w;
  }
  return true;
}

bool read_all(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  const int tmo = timeout_ms();
  while (n) {
    pollfd pf{fd, POLLIN, 0};
    const int pr = poll(&pf, 1, tmo);
    if (pr == 0) return false;  // the peer never posted its half
    if (pr < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    const ssize_t r = ::recv(fd, c, n, 0);
    if (r == 0) return false;
    if (r < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    c += r;
    n -= (size_t)r;
  }
  return true;
}

int run(std::vector<Op>& ops) {
  if (ops.empty()) return kOk;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return kSystemError;
  std::vector<hipStream_t> seen;
  for (const Op& o : ops) {
    bool dup = false;
    for (hipStream_t s : seen) dup = dup || s == o.s;
    if (!dup) {
      if (hipStreamSynchronize(o.s) != hipSuccess) return kSystemError;
      seen.push_back(o.s);
    }
  }
  int send_rc = kOk;
  std::thread sender([&] {
    if (hipSetDevice(dev) != hipSuccess) {
      send_rc = kSystemError;
      return;
    }
    std::vector<char> host;
    for (const Op& o : ops) {
      if (!o.send) continue;
      host.resize(o.n);
      if (hipMemcpy(host.data(), o.p, o.n, hipMemcpyDeviceToHost) != hipSuccess) {
        send_rc = kSystemError;
        return;
      }
      const uint64_t len = o.n;
      if (!write_all(o.c->fd[o.peer], &len, sizeof(len)) || !write_all(o.c->fd[o.peer], host.data(), o.n)) {
        send_rc = kSystemError;
        return;
      }
    }
  });
  int rc = kOk;
  std::vector<char> host;
  for (const Op& o : ops) {
    if (o.send || rc != kOk) continue;
    uint64_t len = 0;
    if (!read_all(o.c->fd[o.peer], &len, sizeof(len))) {
      rc = kSystemError;
      break;
    }
    if (len != o.n) {  // the peer's send and this receive disagree about the slice: a routing-plan bug, not a wire fault
      fprintf(stderr, "loopback_wire: rank %d expected %zu bytes from rank %d, it sent %llu\n", o.c->rank, o.n, o.peer,
              (unsigned long long)len);
      rc = kInvalidUsage;
      break;
    }
    host.resize(o.n);
    if (!read_all(o.c->fd[o.peer], host.data(), o.n)) {
      rc = kSystemError;
      break;
    }
    if (hipMemcpy(o.p, host.data(), o.n, hipMemcpyHostToDevice) != hipSuccess) {
      rc = kSystemError;
      break;
    }
  }
  sender.join();
  return rc != kOk ? rc : send_rc;
}

int post(bool send, void* p, size_t n, int peer, Comm* c, hipStream_t s) {
  if (!c || peer < 0 || peer >= c->world || peer == c->rank || (n && !p)) return kInvalidArgument;
  g_ops.push_back(Op{c, send, p, n, peer, s});
  if (g_depth == 0) {
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
  }
  return kOk;
}

// a peer that stops reading (or writing) must not hang a GPU box: both directions of every connection time out
void set_timeouts(int fd) {
  timeval tv;
  tv.tv_sec = timeout_ms() / 1000;
  tv.tv_usec = 0;
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
}

void close_all(Comm* c) {
  for (int f : c->fd)
    if (f >= 0) close(f);
  if (c->listen_fd >= 0) close(c->listen_fd);
}

}  // namespace

extern "C" {

int ncclGetUniqueId(Uid* u) {
  if (!u) return kInvalidArgument;
  memset(u->b, 0, sizeof(u->b));
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(u->b, sizeof(u->b), "esr-loopback-wire-%d-%llx", (int)getpid(),
           (unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec);
  return kOk;
}

int ncclCommInitRank(Comm** out, int world, Uid uid, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world) return kInvalidArgument;
  uid.b[sizeof(uid.b) - 1] = 0;
  Comm* c = new Comm;
  c->world = world;
  c->rank = rank;
  c->fd.assign(world, -1);
  sockaddr_un a;
  socklen_t al = make_addr(&a, uid.b, rank);
  c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
  if (c->listen_fd < 0 || bind(c->listen_fd, reinterpret_cast<sockaddr*>(&a), al) != 0 || listen(c->listen_fd, world) != 0) {
    close_all(c);
    delete c;
    return kSystemError;
  }
  const int tmo = timeout_ms();
  for (int p = 0; p < rank; ++p) {  // connect to every lower rank (its listening socket may not exist yet: retry)
    al = make_addr(&a, uid.b, p);
    int f = -1;
    for (int waited = 0; waited < tmo; waited += 5) {
      f = socket(AF_UNIX, SOCK_STREAM, 0);
      if (f >= 0 && connect(f, reinterpret_cast<sockaddr*>(&a), al) == 0) break;
      if (f >= 0) close(f);
      f = -1;
      usleep(5000);
    }
    const int32_t me = rank;
    if (f < 0 || !write_all(f, &me, sizeof(me))) {
      if (f >= 0) close(f);
      close_all(c);
      delete c;
      return kSystemError;
    }
    set_timeouts(f);
    c->fd[p] = f;
  }
  for (int k = rank + 1; k < world; ++k) {  // accept every higher rank (in whatever order they arrive)
    pollfd pf{c->listen_fd, POLLIN, 0};
    int32_t who = -1;
    int f = -1;
    if (poll(&pf, 1, tmo) > 0) f = accept(c->listen_fd, nullptr, nullptr);
    if (f < 0 || !read_all(f, &who, sizeof(who)) || who <= rank || who >= world || c->fd[who] >= 0) {
      if (f >= 0) close(f);
      close_all(c);
      delete c;
      return kSystemError;
    }
    set_timeouts(f);
    c->fd[who] = f;
  }
  *out = c;
  return kOk;
}

int ncclSend(const void* p, size_t n, int /*dtype: bytes*/, int peer, Comm* c, hipStream_t s) {
  return post(true, const_cast<void*>(p), n, peer, c, s);
}
int ncclRecv(void* p, size_t n, int /*dtype: bytes*/, int peer, Comm* c, hipStream_t s) {
  return post(false, p, n, peer, c, s);
}
int ncclGroupStart(void) {
  ++g_depth;
  return kOk;
}
int ncclGroupEnd(void) {
  if (g_depth <= 0) return kInvalidUsage;
  if (--g_depth > 0) return kOk;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run(ops);
}
int ncclCommDestroy(Comm* c) {
  if (c) {
    close_all(c);
    delete c;
  }
  return kOk;
}
int ncclCommAbort(Comm* c) { return ncclCommDestroy(c); }
int ncclCommCount(Comm* c, int* n) {
  if (!c || !n) return kInvalidArgument;
  *n = c->world;
  return kOk;
}
int ncclCommUserRank(Comm* c, int* r) {
  if (!c || !r) return kInvalidArgument;
  *r = c->rank;
  return kOk;
}
int ncclCommGetAsyncError(Comm* c, int* e) {
  if (!c || !e) return kInvalidArgument;
  *e = kOk;
  return kOk;
}
const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case kOk: return "no error";
    case kSystemError: return "loopback wire: socket / copy failure or timeout";
    case kInvalidArgument: return "loopback wire: invalid argument";
    case kInvalidUsage: return "loopback wire: a send and its receive disagree about the byte count";
    default: return "loopback wire: internal error";
  }
}

}  // extern "C"
