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
//
// ESR_WIRE_ASYNC=1 (round 5) teaches the wire RCCL's hazards instead of hiding them:
//   * ncclGroupEnd no longer blocks: the group is ENQUEUED on its stream and the call returns at once, as RCCL's does --
//     device-to-pinned copies of the send slices; a host function that only hands the group to the communicator's worker
//     thread (it never blocks: the runtime may run the host functions of all streams on one thread); a one-wave kernel
//     that WAITS on the stream for the worker's completion ticket in pinned host memory -- what RCCL's point-to-point
//     kernel does while its peer has not arrived; pinned-to-device copies of the received slices.  The worker moves the
//     bytes over the sockets.  Two communicators driven from two streams (the overlapped loop: esr_shard_step.hip,
//     comm2 + side stream) then really are in flight together, their waits running beside the other stream's kernels.
//   * enqueue ORDER is checked across communicators: every group a rank enqueues is announced to its peers on a control
//     channel (one per pair of processes, shared by all communicators), and each rank compares the sequence of
//     communicators its peer enqueued groups on with its own.  Two ranks that enqueue the groups of two communicators in
//     different orders are legal here (each communicator has its own sockets) and a potential DEADLOCK on RCCL (two
//     point-to-point kernels that each wait for a peer whose matching kernel sits behind the other one): the wire
//     reports it -- stderr, ncclInvalidUsage from the next call, ncclCommGetAsyncError -- instead of working.
// Exports exactly the twelve symbols esr_comm.hip binds (esr_comm.hip: ESR_SYM list).
#include <errno.h>
#include <hip/hip_runtime.h>
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

#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

enum { kOk = 0, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

struct Uid {
  char b[128];
};
struct Slot {  // pinned staging of one enqueued group (ESR_WIRE_ASYNC)
  char *send = nullptr, *recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;
  hipEvent_t done = nullptr;
  bool used = false;
};
struct Comm {
  int world = 0, rank = 0, listen_fd = -1;
  std::vector<int> fd;
  std::vector<int> ctl;          // control sockets (the first communicator's become the process-wide channel)
  uint64_t hash = 0;             // the same on every rank: from the unique id
  uint64_t groups = 0;           // groups enqueued on this communicator
  std::atomic<int> async_error{0};
  Slot slot[4];
  int next_slot = 0;
  // ESR_WIRE_ASYNC: the worker that moves the bytes of enqueued groups, in enqueue order
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<void*> queue;       // Group*: handed over by the stream's host function
  bool stop = false;
  uint32_t* ticket_done = nullptr;  // pinned, device-visible: the ticket of the last group whose bytes have moved
  uint32_t tickets = 0;
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

// ---- ESR_WIRE_ASYNC: enqueued groups + the cross-communicator order check ------------------------------------------------
bool async_mode() {
  static const bool on = [] {
    const char* e = getenv("ESR_WIRE_ASYNC");
    return e && e[0] == '1';
  }();
  return on;
}

struct Notice {
  uint64_t magic, hash, seq;
};
constexpr uint64_t kNoticeMagic = 0x455352574952454eull;

std::mutex g_order_mu;
std::vector<int> g_ctrl;                          // control socket per peer rank (the first communicator's), -1 = none
int g_ctrl_world = 0, g_ctrl_rank = -1;
std::vector<std::vector<uint64_t>> g_mine, g_theirs;  // per peer: communicators of the groups enqueued, in order
std::vector<std::string> g_partial;               // per peer: bytes of a notice that has not arrived completely
std::atomic<int> g_order_error{0};

// read whatever notices have arrived (never blocks) and compare the two sequences as far as both are known
void order_check_locked() {
  for (int p = 0; p < g_ctrl_world; ++p) {
    if (p >= (int)g_ctrl.size() || g_ctrl[p] < 0) continue;
    char buf[sizeof(Notice) * 16];
    for (;;) {
      const ssize_t r = ::recv(g_ctrl[p], buf, sizeof(buf), MSG_DONTWAIT);
      if (r <= 0) break;
      g_partial[p].append(buf, (size_t)r);
    }
    while (g_partial[p].size() >= sizeof(Notice)) {
      Notice nt;
      memcpy(&nt, g_partial[p].data(), sizeof(nt));
      g_partial[p].erase(0, sizeof(nt));
      if (nt.magic == kNoticeMagic) g_theirs[p].push_back(nt.hash);
    }
    const size_t n = std::min(g_mine[p].size(), g_theirs[p].size());
    for (size_t k = 0; k < n; ++k) {
      if (g_mine[p][k] != g_theirs[p][k] && !g_order_error.load()) {
        fprintf(stderr,
                "loopback_wire: ORDER VIOLATION between ranks %d and %d: exchange group #%zu with that peer was enqueued "
                "on communicator %016llx here and on %016llx there.  Two ranks that enqueue the groups of two "
                "communicators in different orders work on this wire and can DEADLOCK on RCCL.\n",
                g_ctrl_rank, p, k, (unsigned long long)g_mine[p][k], (unsigned long long)g_theirs[p][k]);
        g_order_error.store(kInvalidUsage);
      }
    }
  }
}
void order_check() {
  std::lock_guard<std::mutex> lk(g_order_mu);
  order_check_locked();
}
// a group of communicator c that exchanges with `peer` is being enqueued NOW (host order = the order RCCL would see)
void order_note(Comm* c, int peer) {
  std::lock_guard<std::mutex> lk(g_order_mu);
  if (c->world != g_ctrl_world || peer >= (int)g_ctrl.size() || g_ctrl[peer] < 0) return;
  const Notice nt{kNoticeMagic, c->hash, c->groups};
  write_all(g_ctrl[peer], &nt, sizeof(nt));
  g_mine[peer].push_back(c->hash);
  order_check_locked();
}

struct Group {
  std::vector<Op> ops;
  std::vector<size_t> off;  // staging offset of every op (send ops into the send buffer, receive ops into the other)
  Slot* slot = nullptr;
  Comm* c = nullptr;
  uint64_t seq = 0;
  uint32_t ticket = 0;
};
struct MsgHead {
  uint64_t len, hash, seq;
};

// reads that give up as soon as an order violation (or another failure) has been flagged anywhere in the process: a
// deadlock the violation WOULD cause on RCCL must not become a 120 s timeout here
bool read_watch(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  int waited = 0;
  const int tmo = timeout_ms();
  while (n) {
    pollfd pf{fd, POLLIN, 0};
    const int pr = poll(&pf, 1, 50);
    if (pr == 0) {
      waited += 50;
      order_check();
      if (g_order_error.load() || waited >= tmo) return false;
      continue;
    }
    if (pr < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    const ssize_t r = ::recv(fd, c, n, 0);
    if (r == 0) return false;
    if (r < 0) {
      if (errno == EINTR || errno == EAGAIN) continue;
      return false;
    }
    c += r;
    n -= (size_t)r;
  }
  return true;
}

// the stream holding the group waits for the communicator's ticket (RCCL's kernel waits for its peer the same way);
// bounded: after ~`limit_ticks` of the 100 MHz clock the wait gives up (the copies that follow then move poisoned bytes)
__global__ void wire_wait_kernel(const uint32_t* ticket_done, uint32_t want, unsigned long long limit_ticks) {
  const unsigned long long t0 = wall_clock64();
  while ((int32_t)(__hip_atomic_load(ticket_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > limit_ticks) break;
  }
}

// the stream has reached the group (its send slices are in pinned memory): hand it to the worker and return at once
void host_handover(void* arg) {
  Group* g = static_cast<Group*>(arg);
  {
    std::lock_guard<std::mutex> lk(g->c->mu);
    g->c->queue.push_back(g);
  }
  g->c->cv.notify_one();
}

// worker: move the bytes of one group (host memory and sockets only), then publish its ticket
void host_exchange(void* arg) {
  Group* g = static_cast<Group*>(arg);
  int send_rc = kOk;
  std::thread sender([&] {
    for (size_t i = 0; i < g->ops.size(); ++i) {
      const Op& o = g->ops[i];
      if (!o.send) continue;
      const MsgHead h{o.n, g->c->hash, g->seq};
      if (!write_all(o.c->fd[o.peer], &h, sizeof(h)) || !write_all(o.c->fd[o.peer], g->slot->send + g->off[i], o.n)) {
        send_rc = kSystemError;
        return;
      }
    }
  });
  int rc = kOk;
  for (size_t i = 0; i < g->ops.size() && rc == kOk; ++i) {
    const Op& o = g->ops[i];
    if (o.send) continue;
    MsgHead h{0, 0, 0};
    if (!read_watch(o.c->fd[o.peer], &h, sizeof(h))) {
      rc = g_order_error.load() ? kInvalidUsage : kSystemError;
      break;
    }
    if (h.len != o.n || h.hash != g->c->hash || h.seq != g->seq) {
      fprintf(stderr, "loopback_wire: rank %d, communicator %016llx group %llu: expected %zu bytes from rank %d, it sent %llu "
                      "(its group %llu)\n", o.c->rank, (unsigned long long)g->c->hash, (unsigned long long)g->seq, o.n,
              o.peer, (unsigned long long)h.len, (unsigned long long)h.seq);
      rc = kInvalidUsage;
      break;
    }
    if (!read_watch(o.c->fd[o.peer], g->slot->recv + g->off[i], o.n)) rc = g_order_error.load() ? kInvalidUsage : kSystemError;
  }
  sender.join();
  if (rc == kOk) rc = send_rc;
  if (rc != kOk) {
    // the received slices are copied to the device whatever happened: poison them so that nobody computes on stale bytes
    for (size_t i = 0; i < g->ops.size(); ++i)
      if (!g->ops[i].send) memset(g->slot->recv + g->off[i], 0xFF, g->ops[i].n);
    g->c->async_error.store(rc);
    fprintf(stderr, "loopback_wire: rank %d: an enqueued exchange FAILED (%s)\n", g->c->rank,
            rc == kInvalidUsage ? "order violation or mismatched slice" : "socket failure or timeout");
  }
  __atomic_store_n(g->c->ticket_done, g->ticket, __ATOMIC_RELEASE);  // the waiting kernel goes on
  delete g;
}

void worker_loop(Comm* c) {
  for (;;) {
    void* g = nullptr;
    {
      std::unique_lock<std::mutex> lk(c->mu);
      c->cv.wait(lk, [&] { return c->stop || !c->queue.empty(); });
      if (c->queue.empty()) return;  // stop
      g = c->queue.front();
      c->queue.pop_front();
    }
    host_exchange(g);
  }
}

bool grow(char** p, size_t* cap, size_t need) {
  if (need <= *cap) return true;
  if (*p) hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = std::max<size_t>(need, 1 << 16) * 2;
  if (hipHostMalloc(reinterpret_cast<void**>(p), want, hipHostMallocDefault) != hipSuccess) return false;
  *cap = want;
  return true;
}

int enqueue(std::vector<Op>& ops) {
  if (ops.empty()) return kOk;
  if (g_order_error.load()) return kInvalidUsage;
  Comm* c = ops[0].c;
  hipStream_t s = ops[0].s;
  for (const Op& o : ops)
    if (o.c != c || o.s != s) return run(ops);  // mixed streams / communicators in one group: the blocking form
  if (c->async_error.load()) return c->async_error.load();
  std::vector<bool> noted(c->world, false);
  for (const Op& o : ops)
    if (!noted[o.peer]) {
      noted[o.peer] = true;
      order_note(c, o.peer);
    }
  if (g_order_error.load()) return kInvalidUsage;
  if (!c->ticket_done) {
    if (hipHostMalloc(reinterpret_cast<void**>(&c->ticket_done), 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
      return kSystemError;
    *c->ticket_done = 0;
    c->worker = std::thread(worker_loop, c);
  }
  Group* g = new Group;
  g->c = c;
  g->seq = c->groups++;
  g->ticket = ++c->tickets;
  g->ops = ops;
  g->off.resize(ops.size());
  size_t ns = 0, nr = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    size_t& acc = ops[i].send ? ns : nr;
    g->off[i] = acc;
    acc += (ops[i].n + 63) & ~(size_t)63;
  }
  Slot* sl = &c->slot[c->next_slot];
  c->next_slot = (c->next_slot + 1) % 4;
  if (sl->used && hipEventSynchronize(sl->done) != hipSuccess) return kSystemError;  // four groups ago: long done
  if (!sl->done && hipEventCreateWithFlags(&sl->done, hipEventDisableTiming) != hipSuccess) return kSystemError;
  if (!grow(&sl->send, &sl->send_cap, ns) || !grow(&sl->recv, &sl->recv_cap, nr)) return kSystemError;
  g->slot = sl;
  for (size_t i = 0; i < ops.size(); ++i)
    if (ops[i].send && ops[i].n &&
        hipMemcpyAsync(sl->send + g->off[i], ops[i].p, ops[i].n, hipMemcpyDeviceToHost, s) != hipSuccess)
      return kSystemError;
  // g belongs to the worker from the hand-over on -- it deletes the group as soon as the bytes have moved, which on an
  // idle stream can be before this thread has queued the copies below: what they need is copied out first.  (Reading
  // g->off after the hand-over was a use-after-free that took a rank down with SIGSEGV once in a few full test runs.)
  const uint32_t ticket = g->ticket;
  const std::vector<size_t> off = g->off;
  if (hipLaunchHostFunc(s, host_handover, g) != hipSuccess) return kSystemError;
  hipLaunchKernelGGL(wire_wait_kernel, dim3(1), dim3(1), 0, s, (const uint32_t*)c->ticket_done, ticket,
                     (unsigned long long)(timeout_ms() / 1000 + 30) * 100000000ull);
  if (hipGetLastError() != hipSuccess) return kSystemError;
  for (size_t i = 0; i < ops.size(); ++i)
    if (!ops[i].send && ops[i].n &&
        hipMemcpyAsync(ops[i].p, sl->recv + off[i], ops[i].n, hipMemcpyHostToDevice, s) != hipSuccess)
      return kSystemError;
  sl->used = true;
  return hipEventRecord(sl->done, s) == hipSuccess ? kOk : kSystemError;
}

int dispatch(std::vector<Op>& ops) { return async_mode() ? enqueue(ops) : run(ops); }

int post(bool send, void* p, size_t n, int peer, Comm* c, hipStream_t s) {
  if (!c || peer < 0 || peer >= c->world || peer == c->rank || (n && !p)) return kInvalidArgument;
  g_ops.push_back(Op{c, send, p, n, peer, s});
  if (g_depth == 0) {
    std::vector<Op> ops;
    ops.swap(g_ops);
    return dispatch(ops);
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
  {
    std::lock_guard<std::mutex> lk(g_order_mu);
    if (!c->ctl.empty() && c->ctl == g_ctrl) {  // the process-wide control channel goes with its communicator
      g_ctrl.clear();
      g_ctrl_world = 0;
    }
  }
  // every enqueued group has run to its end (host function, wait kernel, copies) before the worker is told to stop: a
  // host function that ran later would hand its group to a communicator that no longer exists
  for (Slot& sl : c->slot)
    if (sl.used && sl.done) hipEventSynchronize(sl.done);
  if (c->worker.joinable()) {
    {
      std::lock_guard<std::mutex> lk(c->mu);
      c->stop = true;
    }
    c->cv.notify_all();
    c->worker.join();
  }
  if (c->ticket_done) {
    hipHostFree(c->ticket_done);
    c->ticket_done = nullptr;
  }
  for (int f : c->fd)
    if (f >= 0) close(f);
  for (int f : c->ctl)
    if (f >= 0) close(f);
  if (c->listen_fd >= 0) close(c->listen_fd);
  for (Slot& sl : c->slot) {
    if (sl.used && sl.done) hipEventSynchronize(sl.done);
    if (sl.done) hipEventDestroy(sl.done);
    if (sl.send) hipHostFree(sl.send);
    if (sl.recv) hipHostFree(sl.recv);
    sl = Slot{};
  }
}

uint64_t fnv1a(const char* s) {
  uint64_t h = 1469598103934665603ull;
  for (; *s; ++s) h = (h ^ (unsigned char)*s) * 1099511628211ull;
  return h;
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
  c->ctl.assign(world, -1);
  c->hash = fnv1a(uid.b);
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
    // the control connection to the same peer (announced as rank + world)
    int f2 = socket(AF_UNIX, SOCK_STREAM, 0);
    const int32_t me2 = rank + world;
    if (f2 < 0 || connect(f2, reinterpret_cast<sockaddr*>(&a), al) != 0 || !write_all(f2, &me2, sizeof(me2))) {
      if (f2 >= 0) close(f2);
      close_all(c);
      delete c;
      return kSystemError;
    }
    set_timeouts(f2);
    c->ctl[p] = f2;
  }
  for (int k = 0; k < 2 * (world - 1 - rank); ++k) {  // accept every higher rank, data + control (in whatever order)
    pollfd pf{c->listen_fd, POLLIN, 0};
    int32_t who = -1;
    int f = -1;
    if (poll(&pf, 1, tmo) > 0) f = accept(c->listen_fd, nullptr, nullptr);
    const bool is_ctl = f >= 0 && read_all(f, &who, sizeof(who)) && who >= world;
    if (is_ctl) who -= world;
    std::vector<int>& tab = is_ctl ? c->ctl : c->fd;
    if (f < 0 || who <= rank || who >= world || tab[who] >= 0) {
      if (f >= 0) close(f);
      close_all(c);
      delete c;
      return kSystemError;
    }
    set_timeouts(f);
    tab[who] = f;
  }
  {
    std::lock_guard<std::mutex> lk(g_order_mu);
    if (g_ctrl.empty()) {  // the first communicator of the process lends its control sockets to the order check
      g_ctrl = c->ctl;
      g_ctrl_world = world;
      g_ctrl_rank = rank;
      g_mine.assign(world, {});
      g_theirs.assign(world, {});
      g_partial.assign(world, std::string());
      g_order_error.store(0);
    }
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
  return dispatch(ops);
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
  if (async_mode()) order_check();
  *e = c->async_error.load() ? c->async_error.load() : g_order_error.load();
  return kOk;
}
const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case kOk: return "no error";
    case kSystemError: return "loopback wire: socket / copy failure or timeout";
    case kInvalidArgument: return "loopback wire: invalid argument";
    case kInvalidUsage:
      return "loopback wire: a send and its receive disagree about the byte count, or two ranks enqueued the groups of two "
             "communicators in different orders (a deadlock on RCCL)";
    default: return "loopback wire: internal error";
  }
}

}  // extern "C"
