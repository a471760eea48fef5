// Error plumbing, device query and the row gather / un-permute kernels.
#include <dlfcn.h>
#include <stdlib.h>

#include "esr_common.h"
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

namespace esr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return ESR_ELAUNCH;
  }
  return ESR_OK;
}

// ---------------------------------------------------------------------------------------------
// Per-kernel launch timing: a measurement facility for bench.py's `roofline.achieved` (HIP events on the stream the
// kernel is launched on).  Records are (static name, start event, stop event); esr_kernel_timing_read waits for them.
// ---------------------------------------------------------------------------------------------
int g_ktimer_on = 0;

// ---- roctx markers ----------------------------------------------------------------------------------------------------
namespace {
typedef int (*fn_roctx_push)(const char*);
typedef int (*fn_roctx_pop)(void);
fn_roctx_push g_roctx_push = nullptr;
fn_roctx_pop g_roctx_pop = nullptr;
bool roctx_bind() {
  static const bool ok = [] {
    const char* libs[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    for (const char* l : libs) {
      void* h = dlopen(l, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      g_roctx_push = reinterpret_cast<fn_roctx_push>(dlsym(h, "roctxRangePushA"));
      g_roctx_pop = reinterpret_cast<fn_roctx_pop>(dlsym(h, "roctxRangePop"));
      if (g_roctx_push && g_roctx_pop) return true;
    }
    return false;
  }();
  return ok;
}
int trace_env() {
  const char* e = getenv("ESR_ROCTX");
  return (e && e[0] == '1' && roctx_bind()) ? 1 : 0;
}
}  // namespace
int g_trace_on = trace_env();
void trace_push(const char* name) {
  if (g_roctx_push) g_roctx_push(name);
}
void trace_pop() {
  if (g_roctx_pop) g_roctx_pop();
}

namespace {
struct KtRecord {
  const char* name;
  hipEvent_t e0, e1;
};
std::mutex g_kt_mu;
std::vector<KtRecord> g_kt_records;
std::vector<hipEvent_t> g_kt_pool;  // events are reused across enable / read cycles
thread_local hipEvent_t g_kt_open_e0 = nullptr;
thread_local const char* g_kt_open_name = nullptr;
hipEvent_t kt_event() {
  if (!g_kt_pool.empty()) {
    hipEvent_t e = g_kt_pool.back();
    g_kt_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

void ktimer_begin(const char* name, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_kt_mu);
  if (g_kt_records.size() >= (size_t)1 << 20) return;  // bounded: a forgotten switch must not eat the host
  hipEvent_t e0 = kt_event();
  if (!e0) return;
  if (hipEventRecord(e0, st) != hipSuccess) {
    g_kt_pool.push_back(e0);
    return;
  }
  g_kt_open_e0 = e0;
  g_kt_open_name = name;
}

void ktimer_end(hipStream_t st) {
  if (!g_kt_open_e0) return;
  std::lock_guard<std::mutex> lock(g_kt_mu);
  hipEvent_t e1 = kt_event();
  if (e1 && hipEventRecord(e1, st) == hipSuccess) {
    g_kt_records.push_back(KtRecord{g_kt_open_name, g_kt_open_e0, e1});
  } else {
    g_kt_pool.push_back(g_kt_open_e0);
    if (e1) g_kt_pool.push_back(e1);
  }
  g_kt_open_e0 = nullptr;
  g_kt_open_name = nullptr;
}

// ---------------------------------------------------------------------------------------------
// Row gather.  One row = `nchunk` chunks of type T (uint4 = 16 B when the row is a multiple of
// 16 B).  A group of G lanes owns one row at a time; UNROLL rows are kept in flight per group so
// a wave has UNROLL independent 1 KiB (G=64) loads outstanding -- the gather is pure HBM latency.
//   SRC_IDX: src row = ids[r], dst row = r      (gather)
//   else   : src row = r,      dst row = ids[r] (un-permute / scatter of a permutation)
// ---------------------------------------------------------------------------------------------
template <typename T, int UNROLL, bool SRC_IDX>
__global__ __launch_bounds__(kBlock) void move_rows_kernel(const T* __restrict__ src,
                                                           const int32_t* __restrict__ ids,
                                                           int64_t n, int nchunk, int G,
                                                           T* __restrict__ dst) {
  const int lane_in_group = threadIdx.x & (G - 1);
  const int64_t groups_per_block = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
  for (int64_t r0 = group; r0 < n; r0 += ngroups * UNROLL) {
    // Tail rows re-read row n-1 (loads stay unconditional so hipcc keeps all UNROLL loads in
    // flight under one counted wait); only the store is predicated.
    int64_t idx[UNROLL], row[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t r = r0 + (int64_t)u * ngroups;
      ok[u] = r < n;
      row[u] = ok[u] ? r : n - 1;
      idx[u] = (int64_t)ids[row[u]];
    }
    for (int c = lane_in_group; c < nchunk; c += G) {
      T v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = src[(SRC_IDX ? idx[u] : row[u]) * nchunk + c];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (ok[u]) dst[(SRC_IDX ? row[u] : idx[u]) * nchunk + c] = v[u];
    }
  }
}

// out[perm[k], :] = (float) rows_bf16[k, :]: the requester side of a sharded lookup on a bf16 table
// (rows cross xGMI as bf16, the loss kernels consume f32).  One 8-byte chunk (4 bf16) -> one float4 per lane.
__global__ __launch_bounds__(kBlock) void unpermute_bf16_to_f32_kernel(const uint2* __restrict__ src,
                                                                      const int32_t* __restrict__ perm, int64_t n,
                                                                      int nchunk, int G, float4* __restrict__ dst) {
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  for (int64_t r = group; r < n; r += ngroups) {
    const int64_t d = perm ? (int64_t)perm[r] : r;
    for (int c = lig; c < nchunk; c += G) {
      const uint2 u = src[r * nchunk + c];
      dst[d * nchunk + c] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                        __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
  }
}

template <bool SRC_IDX>
static int launch_move_rows(const void* src, int dtype, int D, const int32_t* ids, int64_t n,
                            void* dst, hipStream_t st) {
  const int64_t row_bytes = (int64_t)D * (dtype == ESR_BF16 ? 2 : 4);
  if (n == 0) return ESR_OK;
  auto geom = [&](int chunk_bytes, int& nchunk, int& G) {
    nchunk = (int)(row_bytes / chunk_bytes);
    G = 1;
    while (G < nchunk && G < kWave) G <<= 1;
  };
  int nchunk, G;
  if (row_bytes % 16 == 0) {
    geom(16, nchunk, G);
    const int grid = grid_for_groups(cdiv(n, 4), G);
    hipLaunchKernelGGL((move_rows_kernel<uint4, 4, SRC_IDX>), dim3(grid), dim3(kBlock), 0, st,
                       (const uint4*)src, ids, n, nchunk, G, (uint4*)dst);
  } else if (row_bytes % 4 == 0) {
    geom(4, nchunk, G);
    const int grid = grid_for_groups(cdiv(n, 4), G);
    hipLaunchKernelGGL((move_rows_kernel<uint32_t, 4, SRC_IDX>), dim3(grid), dim3(kBlock), 0, st,
                       (const uint32_t*)src, ids, n, nchunk, G, (uint32_t*)dst);
  } else {
    geom(2, nchunk, G);
    const int grid = grid_for_groups(cdiv(n, 4), G);
    hipLaunchKernelGGL((move_rows_kernel<uint16_t, 4, SRC_IDX>), dim3(grid), dim3(kBlock), 0, st,
                       (const uint16_t*)src, ids, n, nchunk, G, (uint16_t*)dst);
  }
  return check_launch(SRC_IDX ? "esr_gather_rows" : "esr_unpermute_rows");
}

// Single block: fixed-order reduction of block partials; out = total * scale.
__global__ __launch_bounds__(kBlock) void scalar_finalize_kernel(const double* __restrict__ part, int nparts,
                                                                double scale, float* __restrict__ out) {
  __shared__ double sm[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) a += part[i];
  const double t = block_sum_d(a, sm);
  if (threadIdx.x == 0) out[0] = (float)(t * scale);
}

void finalize_scalar(const double* part, int n, double scale, float* out, hipStream_t st) {
  hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(kBlock), 0, st, part, n, scale, out);
}

// Debug screen for device-resident ids: report[0] += #ids outside [0, V), report[1] = min position of one.
__global__ __launch_bounds__(kBlock) void check_ids_kernel(const int32_t* __restrict__ ids, int64_t n, int64_t V,
                                                           int64_t* __restrict__ report) {
  int64_t bad = 0, first = INT64_MAX;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int64_t v = ids[i];
    if (v < 0 || v >= V) {
      ++bad;
      first = first < i ? first : i;
    }
  }
  if (bad) {
    atomicAdd(reinterpret_cast<unsigned long long*>(report), (unsigned long long)bad);
    atomicMin(reinterpret_cast<long long*>(report + 1), (long long)first);
  }
}

}  // namespace esr

using namespace esr;

namespace esr {
// One wave that waits until *flag has reached `value` (wrap-safe: the words are sequence numbers), asleep between
// polls; gives up after timeout_ticks of the 100 MHz constant clock (s_memrealtime), so a producer that never runs
// cannot hang the queue.  The load is agent-scope: served by the memory side, not by this XCD's L2.
__global__ __launch_bounds__(64) void stream_gate_kernel(const uint32_t* __restrict__ flag, uint32_t value,
                                                        unsigned long long timeout_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > timeout_ticks) break;
  }
}
}  // namespace esr

extern "C" {

int esr_stream_gate(const uint32_t* flag, uint32_t value, uint32_t timeout_us, esr_stream_t stream) {
  ESR_REQUIRE(flag && !((uintptr_t)flag & 3), "esr_stream_gate: null or misaligned flag");
  hipLaunchKernelGGL(stream_gate_kernel, dim3(1), dim3(64), 0, as_stream(stream), flag, value,
                     (unsigned long long)timeout_us * 100ull);
  return check_launch("esr_stream_gate");
}

int esr_check_ids(const int32_t* ids, int64_t n, int64_t V, int64_t* report, esr_stream_t stream) {
  ESR_REQUIRE(n >= 0 && V >= 0, "esr_check_ids: n and V must be non-negative");
  ESR_REQUIRE(report, "esr_check_ids: null report (device int64 [2])");
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(ids, "esr_check_ids: null ids");
  const int grid = (int)std::min<int64_t>(kMaxGrid, cdiv(n, kBlock));
  hipLaunchKernelGGL(check_ids_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), ids, n, V, report);
  return check_launch("esr_check_ids");
}

const char* esr_last_error(void) { return g_err; }

int esr_version(void) { return 101; }

int esr_trace_markers(int enable) {
  if (enable && !roctx_bind()) {
    set_error("esr_trace_markers: no roctx library (librocprofiler-sdk-roctx.so.1 / libroctx64.so.4) on the loader path");
    g_trace_on = 0;
    return ESR_ENODEVICE;
  }
  g_trace_on = enable ? 1 : 0;
  return ESR_OK;
}

int esr_kernel_timing(int enable) {
  std::lock_guard<std::mutex> lock(g_kt_mu);
  for (const KtRecord& r : g_kt_records) {  // drop what nobody read
    g_kt_pool.push_back(r.e0);
    g_kt_pool.push_back(r.e1);
  }
  g_kt_records.clear();
  g_ktimer_on = enable ? 1 : 0;
  return ESR_OK;
}

long esr_kernel_timing_read(char* buf, size_t cap) {
  std::vector<KtRecord> recs;
  {
    std::lock_guard<std::mutex> lock(g_kt_mu);
    recs.swap(g_kt_records);
  }
  struct Agg {
    const char* name;
    long calls;
    double total, mn, mx;
  };
  std::vector<Agg> agg;
  for (const KtRecord& r : recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = -1.f;
    if (ms >= 0.f) {
      size_t i = 0;
      for (; i < agg.size(); ++i)
        if (agg[i].name == r.name || strcmp(agg[i].name, r.name) == 0) break;
      if (i == agg.size()) agg.push_back(Agg{r.name, 0, 0.0, 1e30, 0.0});
      agg[i].calls += 1;
      agg[i].total += ms;
      agg[i].mn = std::min(agg[i].mn, (double)ms);
      agg[i].mx = std::max(agg[i].mx, (double)ms);
    }
  }
  {
    std::lock_guard<std::mutex> lock(g_kt_mu);
    for (const KtRecord& r : recs) {
      g_kt_pool.push_back(r.e0);
      g_kt_pool.push_back(r.e1);
    }
  }
  std::string out;
  char line[256];
  for (const Agg& a : agg) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.6f\t%.6f\n", a.name, a.calls, a.total, a.mn, a.mx);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t n = std::min(cap - 1, out.size());
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (long)out.size() + 1;
}

int esr_device_info(int* cu_count, int* wave_size, size_t* hbm_bytes, char* arch, int arch_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("esr_device_info: no HIP device");
    return ESR_ENODEVICE;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    set_error("esr_device_info: hipGetDeviceProperties failed");
    return ESR_ENODEVICE;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return ESR_OK;
}

int esr_gather_rows(const void* table, int dtype, int64_t V, int D, const int32_t* ids, int64_t n,
                    void* out, esr_stream_t stream) {
  ESR_REQUIRE(n >= 0 && V >= 0 && D > 0, "esr_gather_rows: bad sizes V=%lld D=%d n=%lld",
              (long long)V, D, (long long)n);
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_gather_rows: bad dtype %d", dtype);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(table && ids && out, "esr_gather_rows: null pointer");
  return launch_move_rows<true>(table, dtype, D, ids, n, out, as_stream(stream));
}

int esr_unpermute_rows_bf16_to_f32(const void* rows_bf16, int D, const int32_t* perm, int64_t n, float* out,
                                   esr_stream_t stream) {
  ESR_REQUIRE(n >= 0 && D > 0 && D % 4 == 0, "esr_unpermute_rows_bf16_to_f32: bad sizes D=%d (multiple of 4) n=%lld", D,
              (long long)n);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(rows_bf16 && out, "esr_unpermute_rows_bf16_to_f32: null pointer");
  const int nchunk = D / 4;
  int G = 1;
  while (G < nchunk && G < kWave) G <<= 1;
  const int grid = grid_for_groups(n, G);
  hipLaunchKernelGGL(unpermute_bf16_to_f32_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                     (const uint2*)rows_bf16, perm, n, nchunk, G, (float4*)out);
  return check_launch("esr_unpermute_rows_bf16_to_f32");
}

int esr_unpermute_rows(const void* rows, int dtype, int D, const int32_t* perm, int64_t n,
                       void* out, esr_stream_t stream) {
  ESR_REQUIRE(n >= 0 && D > 0, "esr_unpermute_rows: bad sizes D=%d n=%lld", D, (long long)n);
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_unpermute_rows: bad dtype %d", dtype);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(rows && perm && out, "esr_unpermute_rows: null pointer");
  return launch_move_rows<false>(rows, dtype, D, perm, n, out, as_stream(stream));
}

}  // extern "C"
