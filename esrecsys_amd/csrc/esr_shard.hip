// Requester side of the row-sharded exchange with every DISTINCT row asked for once (build-defined -- the
// reference is single-device).  A batch's lookups are virtual rows vid of the concatenated tables, owner = vid mod G,
// local row = vid div G.  esr_unique_by_owner turns the occurrence list into
//
//   ulocal [n_u]        the distinct rows, owner-major and ascending inside an owner: slice o (ucounts[o] entries) IS the
//                       list of local rows this rank asks of owner o, and the order the rows come back in
//   ucounts [G]         distinct rows per owner
//   uidx [n]            occurrence i reads row uidx[i] of what came back ...
//   sorted_uidx, perm   ... and the occurrences grouped by distinct row (sorted_uidx ascending, perm = the occurrence),
//                       which is what the segment sum of the per-occurrence gradient rows needs (esr_segment_sum_rows)
//
// so that rows and gradients cross xGMI once per distinct (rank, row) instead of once per occurrence -- GloVe's id stream
// is Zipfian (wikipedia/make_cooccurrence.py:33-55: a few tokens take most pairs).  Owners need nothing new: they serve
// whatever list they are sent and segment-reduce what comes back by row.
//
//   keys      key = owner * Lv + local row per occurrence                                  (one launch)
//   sort      esr_segment_sort_ids on the keys -> (sorted keys, perm)                      (the id sort of esr_sort.hip)
//   heads     per 1024-position tile: how many positions start a run of equal keys         (one launch; zeroes ucounts)
//   scatter   every workgroup adds up the tiles before its own, scans its tile, writes sorted_uidx / uidx / ulocal and
//             counts its heads per owner (integer atomics: order-free)                      (one launch)
#include "esr_common.h"

namespace esr {

constexpr int kUqTile = 1024;  // positions per workgroup of the heads / scatter kernels (4 per thread)
constexpr int kUqMaxSegs = 4;
constexpr int kBucketMaxWorldShard = 64;  // most ranks of one exchange group the plan-phase kernels count for

struct UqSegs {
  const int32_t* ids[kUqMaxSegs];
  int64_t start[kUqMaxSegs + 1];
  int64_t offset[kUqMaxSegs];
  int n;
};

__global__ __launch_bounds__(kBlock) void uq_keys_kernel(UqSegs sg, int64_t n, int world, int64_t Lv,
                                                        int32_t* __restrict__ keys) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    int k = 0;
#pragma unroll
    for (int s = 1; s < kUqMaxSegs; ++s)
      if (s < sg.n && i >= sg.start[s]) k = s;
    const int64_t vid = (int64_t)sg.ids[k][i - sg.start[k]] + sg.offset[k];
    keys[i] = (int32_t)((vid % world) * Lv + vid / world);
  }
}

__device__ __forceinline__ bool uq_head(const int32_t* __restrict__ sorted_keys, int64_t p) {
  return p == 0 || sorted_keys[p] != sorted_keys[p - 1];
}

__global__ __launch_bounds__(kBlock) void uq_heads_kernel(const int32_t* __restrict__ sorted_keys, int64_t n,
                                                         int32_t* __restrict__ tile_heads, int64_t* __restrict__ ucounts,
                                                         int world) {
  __shared__ int sm[4];
  if (blockIdx.x == 0 && (int)threadIdx.x < world) ucounts[threadIdx.x] = 0;
  const int64_t base = (int64_t)blockIdx.x * kUqTile + 4 * threadIdx.x;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n && uq_head(sorted_keys, base + j)) ++c;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, kWave);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_heads[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(kBlock) void uq_scatter_kernel(const int32_t* __restrict__ sorted_keys,
                                                           const int32_t* __restrict__ perm, int64_t n, int64_t Lv,
                                                           const int32_t* __restrict__ tile_heads,
                                                           int32_t* __restrict__ ulocal, int32_t* __restrict__ uidx,
                                                           int32_t* __restrict__ sorted_uidx,
                                                           int64_t* __restrict__ ucounts, int world) {
  __shared__ int sm[8];
  __shared__ int s_owner[8];
  // heads in the tiles before this one: every workgroup re-reduces the (at most n / 1024) tile counts out of L2
  int before = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += kBlock) before += tile_heads[t];
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, kWave);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = before;
  if ((int)threadIdx.x < 8) s_owner[threadIdx.x] = 0;
  __syncthreads();
  before = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kUqTile + 4 * threadIdx.x;
  bool h[4];
  int32_t key[4];
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = base + j < n && uq_head(sorted_keys, base + j);
    key[j] = base + j < n ? sorted_keys[base + j] : 0;
    c += h[j] ? 1 : 0;
  }
  // exclusive scan of the per-thread head counts across the workgroup
  int incl = c;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 1; off < kWave; off <<= 1) {
    const int v = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += v;
  }
  if (lane == 63) sm[4 + wid] = incl;
  __syncthreads();
  int wave_before = 0;
  for (int w = 0; w < wid; ++w) wave_before += sm[4 + w];
  int u = before + wave_before + incl - c;  // distinct rows in front of this thread's first position
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j >= n) break;
    if (h[j]) {
      ulocal[u] = (int32_t)(key[j] % Lv);
      atomicAdd(&s_owner[key[j] / Lv], 1);
      ++u;
    }
    sorted_uidx[base + j] = u - 1;
    uidx[perm[base + j]] = u - 1;
  }
  __syncthreads();
  if ((int)threadIdx.x < world && s_owner[threadIdx.x])
    atomicAdd(reinterpret_cast<unsigned long long*>(ucounts + threadIdx.x), (unsigned long long)s_owner[threadIdx.x]);
}

struct UqWs {
  int32_t* keys;         // [n]
  int32_t* sorted_keys;  // [n]
  int32_t* tile_heads;   // [ceil(n / 1024)]
  void* sort_ws;
  size_t sort_ws_bytes;
};
static size_t uq_ws_layout(int64_t n, char* base, UqWs* out) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  UqWs w;
  w.keys = (int32_t*)take(sizeof(int32_t) * (size_t)n);
  w.sorted_keys = (int32_t*)take(sizeof(int32_t) * (size_t)n);
  w.tile_heads = (int32_t*)take(sizeof(int32_t) * (size_t)cdiv(n, kUqTile));
  w.sort_ws_bytes = esr_segment_sort_workspace_bytes(n);
  w.sort_ws = take(w.sort_ws_bytes);
  if (out) *out = w;
  return off;
}


// ---- the overlapped loop's plan phase (esrecsys_amd/sharded.py begin_stale_sets; round 5: torch.searchsorted / cumsum /
// scatter_ / gather there are gone) -------------------------------------------------------------------------------------------
// sorted_membership: flags[l][j] = 1 iff cur[l][j] != sentinel and cur[l][j] occurs in the ascending list seq[l][0 .. m).
__global__ __launch_bounds__(kBlock) void sorted_membership_kernel(const int32_t* __restrict__ cur, int64_t n,
                                                                  const int32_t* __restrict__ seq, int64_t m,
                                                                  int32_t sentinel, uint8_t* __restrict__ flags) {
  const int l = blockIdx.y;
  const int32_t* s = seq + (int64_t)l * m;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
    const int32_t v = cur[(int64_t)l * n + j];
    int64_t lo = 0, hi = m;  // first position with s[pos] >= v
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (s[mid] < v) lo = mid + 1; else hi = mid;
    }
    flags[(int64_t)l * n + j] = (v != sentinel && lo < m && s[lo] == v) ? 1 : 0;
  }
}
// flagged_first: every row of `values` [L][n] (values == NULL: the positions 0 .. n - 1) with its flagged entries FIRST,
// order kept (a stable partition; only the flagged prefix is written), and, per row, the number of flagged entries inside
// each of G consecutive slices of lengths slice_len[l][0 .. G).  Two launches over 1024-entry tiles: flagged entries per
// tile; then every workgroup adds up the tiles before its own, ranks its flagged entries by ballots and counts them per
// slice (integer atomics on zeroed words: order-free).
constexpr int kFfTile = 1024;
__global__ __launch_bounds__(kBlock) void flagged_count_kernel(const uint8_t* __restrict__ flags, int64_t n, int ntiles,
                                                              int32_t* __restrict__ tile_cnt) {
  __shared__ int wsum[kBlock / 64];
  const int l = blockIdx.y, tile = blockIdx.x, t = threadIdx.x;
  int c = 0;
#pragma unroll
  for (int q = 0; q < kFfTile / kBlock; ++q) {
    const int64_t j = (int64_t)tile * kFfTile + q * kBlock + t;
    c += (j < n && flags[(int64_t)l * n + j]) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((t & 63) == 0) wsum[t >> 6] = c;
  __syncthreads();
  if (t == 0) tile_cnt[(int64_t)l * ntiles + tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(kBlock) void flagged_scatter_kernel(const uint8_t* __restrict__ flags,
                                                                const int32_t* __restrict__ values, int64_t n, int ntiles,
                                                                const int32_t* __restrict__ tile_cnt,
                                                                const int64_t* __restrict__ slice_len, int G,
                                                                int32_t* __restrict__ out,
                                                                unsigned long long* __restrict__ counts,
                                                                int64_t cstride_l, int64_t cstride_g) {
  __shared__ int red[kBlock / 64];
  __shared__ int wave_base[kBlock / 64];
  __shared__ int64_t ends[kBucketMaxWorldShard];
  __shared__ int scount[kBucketMaxWorldShard];
  const int l = blockIdx.y, tile = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  int before = 0;
  for (int u = t; u < tile; u += kBlock) before += tile_cnt[(int64_t)l * ntiles + u];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) red[w] = before;
  if (t < G) scount[t] = 0;
  if (t == 0) {
    int64_t e = 0;
    for (int g = 0; g < G; ++g) {
      e += slice_len[(int64_t)l * G + g];
      ends[g] = e;
    }
  }
  __syncthreads();
  int base = red[0] + red[1] + red[2] + red[3];
  for (int q = 0; q < kFfTile / kBlock; ++q) {  // 256 consecutive entries per round: ranks by ballot, in order
    const int64_t j = (int64_t)tile * kFfTile + q * kBlock + t;
    const bool f = j < n && flags[(int64_t)l * n + j] != 0;
    const unsigned long long b = __ballot(f);
    const int rank_in_wave = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();  // (wave_base of the previous round has been read)
    if (lane == 0) wave_base[w] = __popcll(b);
    __syncthreads();
    int wb = 0;
    for (int k = 0; k < w; ++k) wb += wave_base[k];
    if (f) {
      out[(int64_t)l * n + base + wb + rank_in_wave] = values ? values[(int64_t)l * n + j] : (int32_t)j;
      int g = 0;
      while (g < G - 1 && j >= ends[g]) ++g;
      atomicAdd(&scount[g], 1);
    }
    base += wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
  }
  __syncthreads();
  if (t < G && scount[t]) atomicAdd(counts + (int64_t)l * cstride_l + (int64_t)t * cstride_g, (unsigned long long)scount[t]);
}

}  // namespace esr

using namespace esr;

extern "C" {

size_t esr_unique_by_owner_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return uq_ws_layout(n, nullptr, nullptr);
}

int esr_unique_by_owner(const int32_t* const* ids, const int64_t* seg_counts, const int64_t* offsets, int nseg, int world,
                        int64_t local_rows, int32_t* ulocal, int32_t* uidx, int32_t* sorted_uidx, int32_t* perm,
                        int64_t* ucounts, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  ESR_REQUIRE(nseg >= 1 && nseg <= kUqMaxSegs && world >= 1 && world <= 8 && local_rows > 0 && ids && seg_counts &&
                  offsets,
              "esr_unique_by_owner: nseg=%d not in [1, %d], world=%d not in [1, 8], or a null / bad argument", nseg,
              kUqMaxSegs, world);
  ESR_REQUIRE((int64_t)world * local_rows < ((int64_t)1 << 31), "esr_unique_by_owner: %lld x %d virtual rows exceed 2^31",
              (long long)local_rows, world);
  UqSegs sg;
  sg.n = nseg;
  sg.start[0] = 0;
  for (int i = 0; i < kUqMaxSegs; ++i) {
    ESR_REQUIRE(i >= nseg || (seg_counts[i] >= 0 && (seg_counts[i] == 0 || ids[i])), "esr_unique_by_owner: bad segment %d", i);
    sg.ids[i] = i < nseg ? ids[i] : nullptr;
    sg.offset[i] = i < nseg ? offsets[i] : 0;
    sg.start[i + 1] = sg.start[i] + (i < nseg ? seg_counts[i] : 0);
  }
  const int64_t n = sg.start[nseg];
  ESR_REQUIRE(n < ((int64_t)1 << 31), "esr_unique_by_owner: n=%lld", (long long)n);
  ESR_REQUIRE(ucounts, "esr_unique_by_owner: null counts");
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    if (hipMemsetAsync(ucounts, 0, sizeof(int64_t) * world, st) != hipSuccess) return check_launch("esr_unique_by_owner");
    return ESR_OK;
  }
  ESR_REQUIRE(ulocal && uidx && sorted_uidx && perm, "esr_unique_by_owner: null output");
  if (!workspace || workspace_bytes < esr_unique_by_owner_workspace_bytes(n) || ((uintptr_t)workspace & 15)) {
    set_error("esr_unique_by_owner: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_unique_by_owner_workspace_bytes(n));
    return ESR_EWORKSPACE;
  }
  UqWs ws;
  uq_ws_layout(n, (char*)workspace, &ws);
  const int grid = (int)std::min<int64_t>(kMaxGrid, cdiv(n, kBlock));
  hipLaunchKernelGGL(uq_keys_kernel, dim3(grid), dim3(kBlock), 0, st, sg, n, world, local_rows, ws.keys);
  if (int rc = esr_segment_sort_ids(ws.keys, n, (int64_t)world * local_rows, ws.sorted_keys, perm, ws.sort_ws,
                                    ws.sort_ws_bytes, stream))
    return rc;
  const int tiles = (int)cdiv(n, kUqTile);
  hipLaunchKernelGGL(uq_heads_kernel, dim3(tiles), dim3(kBlock), 0, st, (const int32_t*)ws.sorted_keys, n, ws.tile_heads,
                     ucounts, world);
  hipLaunchKernelGGL(uq_scatter_kernel, dim3(tiles), dim3(kBlock), 0, st, (const int32_t*)ws.sorted_keys,
                     (const int32_t*)perm, n, local_rows, (const int32_t*)ws.tile_heads, ulocal, uidx, sorted_uidx,
                     ucounts, world);
  return check_launch("esr_unique_by_owner");
}


int esr_sorted_membership(const int32_t* cur, int64_t n, const int32_t* seq, int64_t m, int nlists, int32_t sentinel,
                          uint8_t* flags, esr_stream_t stream) {
  TraceScope trace_scope_("esr_sorted_membership");
  ESR_REQUIRE(n >= 0 && m >= 0 && nlists >= 1 && nlists <= 65535, "esr_sorted_membership: bad sizes n=%lld m=%lld nlists=%d",
              (long long)n, (long long)m, nlists);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(cur && flags && (m == 0 || seq), "esr_sorted_membership: null pointer");
  const dim3 grid((unsigned)std::min<int64_t>(1024, cdiv(n, kBlock)), nlists);
  hipLaunchKernelGGL(sorted_membership_kernel, grid, dim3(kBlock), 0, as_stream(stream), cur, n, seq, m, sentinel, flags);
  return check_launch("esr_sorted_membership");
}

size_t esr_flagged_first_workspace_bytes(int64_t n, int nlists) {
  if (n <= 0 || nlists <= 0) return 256;
  return align_up((size_t)nlists * (size_t)cdiv(n, kFfTile) * sizeof(int32_t), 256);
}

int esr_flagged_first(const uint8_t* flags, const int32_t* values, int64_t n, int nlists, const int64_t* slice_len, int G,
                      int32_t* out, int64_t* counts, int64_t counts_stride_list, int64_t counts_stride_slice,
                      void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_flagged_first");
  ESR_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && nlists >= 1 && nlists <= 65535 && G >= 1 && G <= kBucketMaxWorldShard,
              "esr_flagged_first: bad sizes n=%lld nlists=%d G=%d (G <= %d)", (long long)n, nlists, G, kBucketMaxWorldShard);
  ESR_REQUIRE(counts && slice_len, "esr_flagged_first: null counts / slice lengths");
  hipStream_t st = as_stream(stream);
  // the caller's count words (any strides) are zeroed here: nlists x G of them
  for (int l = 0; l < nlists; ++l)
    if (hipMemset2DAsync(counts + (int64_t)l * counts_stride_list, (size_t)counts_stride_slice * sizeof(int64_t), 0,
                         sizeof(int64_t), (size_t)G, st) != hipSuccess)
      return check_launch("esr_flagged_first");
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(flags && out && workspace && !((uintptr_t)workspace & 3) &&
                  workspace_bytes >= esr_flagged_first_workspace_bytes(n, nlists),
              "esr_flagged_first: null pointer or workspace too small");
  const int ntiles = (int)cdiv(n, kFfTile);
  const dim3 grid(ntiles, nlists);
  hipLaunchKernelGGL(flagged_count_kernel, grid, dim3(kBlock), 0, st, flags, n, ntiles, (int32_t*)workspace);
  hipLaunchKernelGGL(flagged_scatter_kernel, grid, dim3(kBlock), 0, st, flags, values, n, ntiles,
                     (const int32_t*)workspace, slice_len, G, out, reinterpret_cast<unsigned long long*>(counts),
                     counts_stride_list, counts_stride_slice);
  return check_launch("esr_flagged_first");
}

}  // extern "C"
