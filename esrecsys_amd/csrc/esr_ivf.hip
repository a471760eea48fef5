// IVF (inverted-file) approximate retrieval for BASELINE config 5 ("top-k ANN scoring vs brute-force"): a build-defined
// extra -- the reference has no ANN index; its retrieval is the exact jax.lax.top_k of pinterest/make_recommendations.py:
// 49-65, which stays the parity reference (esr_retrieve_topk) and the yardstick for recall@k.
//
// Index (built once, host side drives it: esrecsys_amd/ivf.py): candidates grouped by their nearest coarse centroid --
// cands_sorted [N, D] list after list, list_off [nlist + 1], orig [N] = the candidate's row in the caller's matrix.
// A search scores every query against the centroids (esr_retrieve_topk: the MFMA GEMM + radix select), keeps its nprobe
// best lists, and then only looks inside those: nq x nprobe (query, list) PAIRS.
//
//   sort      the pairs by list (esr_segment_sort_ids)            -> pairs of one list are contiguous
//   prep      per list: where its pairs start, how many 64-row tiles they make; prefix over the lists
//   score     grouped GEMM on the FP32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products -- what is approximate
//             about the answer is only WHICH lists are looked at): a workgroup owns a 64 pairs x 64 candidates tile of
//             ONE list; a pair's score row is `pitch` long, -inf behind the end of its list
//   head      the pairs of every query's BEST list(s) first (as many probe slots as hold more than k candidates: one,
//             usually): dense scores, radix select -> the running top-k of the query and tau = its k-th best score
//   filter    the other pairs, kIvfRound probe slots at a time: the same GEMM, but a score is kept only if it reaches its
//             query's tau -- appended to the query's list behind the running top-k (the brute force's filtered
//             epilogue: tau is a lower bound of the final k-th score, so nothing that belongs to the answer is dropped);
//             between rounds a select compacts the list to its k best and raises tau
//   tail      radix select over each query's short list -> the k best, best first
//   map       position -> (probe slot, position in list) -> candidate row
// (Measured on the way: per-pair selects + a merge of nprobe x k survivors were 2/3 of a k = 500 search; one select per
// query over all nprobe x pitch scores streamed 270 KB rows from memory at 0.4 TB/s.)
//
// Work: 2 nq nprobe (N / nlist) D flop instead of 2 nq N D: nlist / nprobe times less (64x at nlist 1024, nprobe 16).
#include "esr_common.h"

namespace esr {

constexpr int kIvfTile = 64;  // pairs x candidates per workgroup
constexpr int kIvfK = 16;     // embedding columns per LDS stage
constexpr int kIvfRound = 8;  // probe slots per filtered round (8192 queries x 8 slots / 1024 lists = full 64-row tiles)

// one workgroup (<= 1024 threads): pair_off[l] = first sorted pair of list l (lower bound), tile_start = exclusive prefix
// of ceil(pairs of l / 64)
__global__ __launch_bounds__(1024) void ivf_prep_kernel(const int32_t* __restrict__ sorted_lists, int64_t P, int nlist,
                                                       int32_t* __restrict__ pair_off, int32_t* __restrict__ tile_start) {
  __shared__ int s_carry;
  __shared__ int s_part[16];
  for (int l = threadIdx.x; l <= nlist; l += blockDim.x) {
    int64_t lo = 0, hi = P;  // first position whose list id >= l
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted_lists[mid] < l) lo = mid + 1; else hi = mid;
    }
    pair_off[l] = (int32_t)lo;
  }
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nlist; base += blockDim.x) {
    const int l = base + threadIdx.x;
    const int tiles = l < nlist ? (pair_off[l + 1] - pair_off[l] + kIvfTile - 1) / kIvfTile : 0;
    int incl = tiles;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) s_part[wid] = incl;
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < wid; ++w) before += s_part[w];
    if (l < nlist) tile_start[l] = before + incl - tiles;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_start[nlist] = s_carry;
}

typedef float ivf_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ constexpr int ivf_mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// grouped GEMM tile: blockIdx.y = row tile over all lists (tile_start locates its list), blockIdx.x = candidate tile.
// Four waves, each a 32 x 32 quarter of the tile: per 16-column stage eight v_mfma_f32_32x32x2_f32, operands one
// ds_read_b32 each from the k-major LDS images (lane = (row % 32, k % 2)).
// `perm` indexes a SET of pairs laid out [queries][ppq] (ppq probe slots per query, the first of them slot0 of the
// query's probe list).  FILTER = false: dense scores S[pair][pitch], -inf behind the end of the list.  FILTER = true:
// scores >= tau[query] are appended to the query's record list as (score, (slot * pitch + position in list)).
struct IvfFilter {
  const float* tau;  // [queries of the chunk]
  int32_t* cnt;      // [queries]: records in the list so far
  int2* pairs;       // [queries][ppitch]
  int64_t ppitch;
};
template <bool FILTER>
__global__ __launch_bounds__(kBlock) void ivf_score_kernel(const float* __restrict__ queries, int D,
                                                          const float* __restrict__ cands,
                                                          const int32_t* __restrict__ list_off,
                                                          const int32_t* __restrict__ perm, int ppq, int slot0,
                                                          int64_t q_base, const int32_t* __restrict__ pair_off,
                                                          const int32_t* __restrict__ tile_start, int nlist,
                                                          float* __restrict__ S, int64_t pitch, IvfFilter flt) {
  __shared__ float As[kIvfK][kIvfTile + 4];
  __shared__ float Bs[kIvfK][kIvfTile + 4];
  __shared__ int s_pair[kIvfTile];
  // blockIdx.x = candidate tile (gridDim.x is a multiple of 8: workgroup b runs on XCD b % 8 = blockIdx.x % 8, so the
  // same candidate tile of a list -- read by every row tile of the list -- always meets the same L2), blockIdx.y = row
  // tile: the column tiles of one row tile are dispatched together and share its query rows
  const int rt = blockIdx.y;
  if (rt >= tile_start[nlist] || (int64_t)blockIdx.x * kIvfTile >= pitch) return;
  int lo = 0, hi = nlist - 1;  // the last list whose first tile is <= rt and that HAS tiles
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= rt) lo = mid; else hi = mid - 1;
  }
  const int l = lo;
  const int L = list_off[l + 1] - list_off[l];
  const int c0 = blockIdx.x * kIvfTile;
  const int row0 = pair_off[l] + (rt - tile_start[l]) * kIvfTile;
  const int nrows = min(kIvfTile, pair_off[l + 1] - row0);
  const int t = threadIdx.x;
  if (t < kIvfTile) s_pair[t] = t < nrows ? perm[row0 + t] : -1;
  __syncthreads();
  const int lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1, l32 = lane & 31, h = lane >> 5;
  if (c0 >= L) {  // behind the end of the list: the head select reads whole rows
    if (FILTER) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = s_pair[wr * 32 + ivf_mfma_row(r, h)];
      if (p >= 0) S[(int64_t)p * pitch + c0 + wc * 32 + l32] = -INFINITY;
    }
    return;
  }
  const int lr = t >> 2, seg = t & 3;  // loader: row lr, 4 floats at column 4 * seg of the stage
  const int my_pair = s_pair[lr];
  const float* qrow = my_pair >= 0 ? queries + (q_base + my_pair / ppq) * (int64_t)D : nullptr;
  const float* crow = c0 + lr < L ? cands + ((int64_t)list_off[l] + c0 + lr) * D : nullptr;
  ivf_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (qrow && 4 * seg < D) a = *reinterpret_cast<const float4*>(qrow + 4 * seg);
  if (crow && 4 * seg < D) b = *reinterpret_cast<const float4*>(crow + 4 * seg);
  for (int k0 = 0; k0 < D; k0 += kIvfK) {
    __syncthreads();  // (the previous stage has been consumed)
    As[4 * seg + 0][lr] = a.x; As[4 * seg + 1][lr] = a.y; As[4 * seg + 2][lr] = a.z; As[4 * seg + 3][lr] = a.w;
    Bs[4 * seg + 0][lr] = b.x; Bs[4 * seg + 1][lr] = b.y; Bs[4 * seg + 2][lr] = b.z; Bs[4 * seg + 3][lr] = b.w;
    __syncthreads();
    {  // the next stage's rows travel while this one is multiplied
      const int kc = k0 + kIvfK + 4 * seg;
      a = make_float4(0.f, 0.f, 0.f, 0.f);
      b = a;
      if (qrow && kc < D) a = *reinterpret_cast<const float4*>(qrow + kc);
      if (crow && kc < D) b = *reinterpret_cast<const float4*>(crow + kc);
    }
#pragma unroll
    for (int kk = 0; kk < kIvfK; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + h][wr * 32 + l32], Bs[kk + h][wc * 32 + l32], acc, 0, 0, 0);
  }
  const int c = c0 + wc * 32 + l32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int p = s_pair[wr * 32 + ivf_mfma_row(r, h)];
    if (p < 0) continue;
    if (!FILTER) {
      S[(int64_t)p * pitch + c] = c < L ? acc[r] : -INFINITY;
    } else if (c < L) {
      const int q = p / ppq;
      if (acc[r] >= flt.tau[q]) {
        const int at = atomicAdd(flt.cnt + q, 1);
        flt.pairs[(int64_t)q * flt.ppitch + at] =
            make_int2(__float_as_int(acc[r]), (int)((int64_t)(slot0 + (p - q * ppq)) * pitch + c));
      }
    }
  }
}

// out [queries][g] = probe slots s0 .. s0 + g - 1 of lists [queries][nprobe]
__global__ __launch_bounds__(kBlock) void ivf_take_slots_kernel(const int32_t* __restrict__ lists, int64_t nq, int nprobe,
                                                               int s0, int g, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nq * g; i += (int64_t)gridDim.x * kBlock) {
    const int64_t q = i / g;
    out[i] = lists[q * nprobe + s0 + (int)(i - q * g)];
  }
}

// position in a query's nprobe x pitch score row -> candidate row of the caller's matrix (-1 where the lists ran out)
__global__ __launch_bounds__(kBlock) void ivf_map_kernel(const int32_t* __restrict__ lists, int64_t nq, int nprobe, int k,
                                                        int64_t pitch, const int32_t* __restrict__ list_off,
                                                        const int32_t* __restrict__ orig, const float* __restrict__ scores,
                                                        int32_t* __restrict__ idx) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nq * k; i += (int64_t)gridDim.x * kBlock) {
    const int64_t q = i / k;
    const int32_t pos = idx[i];
    if (scores[i] == -INFINITY || pos < 0) {
      idx[i] = -1;
      continue;
    }
    const int slot = (int)(pos / pitch), c = (int)(pos - slot * pitch);
    idx[i] = orig[list_off[lists[q * nprobe + slot]] + c];
  }
}

struct IvfWs {
  int32_t* set_lists;     // [cq * max(f, kIvfRound)]  the probe slots of the set being scored
  int32_t* sorted_lists;  // [the same]
  int32_t* perm;          // [the same]
  int32_t* pair_off;      // [nlist + 1]
  int32_t* tile_start;    // [nlist + 1]
  float* S;               // [cq * f, pitch]  head scores
  int2* pairs;            // [cq, ppitch]     running top-k + what the filter lets through
  int32_t* cnt;           // [cq]
  float* tau;             // [cq]
  void* sort_ws;
  size_t sort_ws_bytes;
};
struct IvfGeom {
  int64_t pitch;   // a pair's score row: the longest list, and enough that nprobe rows hold more than k entries
  int f;           // probe slots scored densely first: the fewest whose rows hold more than k entries
  int64_t ppitch;  // a query's record list: k + everything one round could let through
  int64_t chunk;   // queries per pass (head scores + record lists stay under ~2 GiB)
  int mark;        // lazy compaction: lists up to this long are not selected down between rounds
};
static IvfGeom ivf_geom(int64_t nq, int max_list, int nprobe, int k, int nlist) {
  IvfGeom g;
  g.pitch = (int64_t)align_up((size_t)std::max<int64_t>(max_list, cdiv((int64_t)k + 1, nprobe)), 64);
  // (as many as the head select still caches in LDS -- 8192 scores: a list with few pairs fills little of a 64-row
  // tile, so three slots cost the scoring pass what one does, and tau starts tighter)
  g.f = (int)std::min<int64_t>(nprobe, std::max<int64_t>(k / g.pitch + 1, 8192 / g.pitch));
  // (lazy compaction, as in esr_retrieve_topk: a list is selected down to its k best between rounds only once it holds
  // more than `mark` records -- the select was 27 % of a k = 500 search in round 3)
  g.mark = (int)std::max<int64_t>(3 * (int64_t)k, 1536);
  g.ppitch = std::max<int64_t>(k, g.mark) + (int64_t)std::min(kIvfRound, nprobe - g.f) * g.pitch;
  const int64_t per_query = g.f * g.pitch * 4 + g.ppitch * 8;
  g.chunk = std::min(nq, std::max<int64_t>(64, ((int64_t)1 << 31) / per_query));
  // the score kernel's row tiles are gridDim.y (<= 65 535): pairs of a set / 64 + one partly filled tile per list
  const int64_t max_pairs = ((int64_t)65535 - nlist) * kIvfTile;
  g.chunk = std::max<int64_t>(1, std::min(g.chunk, max_pairs / std::max(g.f, kIvfRound)));
  return g;
}
static size_t ivf_layout(const IvfGeom& g, int64_t cq, int nprobe, int nlist, char* base, IvfWs* out) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  const int64_t P1 = cq * g.f, Pm = cq * std::max(g.f, std::min(kIvfRound, std::max(1, nprobe - g.f)));
  IvfWs w;
  w.set_lists = (int32_t*)take(4 * (size_t)Pm);
  w.sorted_lists = (int32_t*)take(4 * (size_t)Pm);
  w.perm = (int32_t*)take(4 * (size_t)Pm);
  w.pair_off = (int32_t*)take(4 * (size_t)(nlist + 1));
  w.tile_start = (int32_t*)take(4 * (size_t)(nlist + 1));
  w.S = (float*)take(4 * (size_t)P1 * (size_t)g.pitch);
  w.pairs = (int2*)take(8 * (size_t)cq * (size_t)g.ppitch);
  w.cnt = (int32_t*)take(4 * (size_t)cq);
  w.tau = (float*)take(4 * (size_t)cq);
  w.sort_ws_bytes = esr_segment_sort_workspace_bytes(Pm);
  w.sort_ws = take(w.sort_ws_bytes);
  if (out) *out = w;
  return off;
}


// ---- index build (esrecsys_amd/ivf.py; round 5: torch.bincount / searchsorted / norm there are gone) --------------------------
// list_off[v] = first position of the ascending `sorted` [n] whose value is >= v, v = 0 .. nvalues (list_off[nvalues] = n);
// max_len[0] = the longest run.  One thread per position fills the offsets of the values between its predecessor's and
// its own.  Values are expected in [0, nvalues); one outside is counted with the nearest list (both ends clamped, so no
// offset outside off[0 .. nvalues] is written and off[nvalues] has ONE writer, the p == n thread).
__global__ __launch_bounds__(kBlock) void run_offsets_kernel(const int32_t* __restrict__ sorted, int64_t n, int nvalues,
                                                            int32_t* __restrict__ off, int32_t* __restrict__ max_len) {
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p <= n; p += (int64_t)gridDim.x * kBlock) {
    const int32_t prev = p == 0 ? -1 : min(max(sorted[p - 1], 0), nvalues - 1);
    const int32_t cur = p == n ? nvalues : min(max(sorted[p], 0), nvalues - 1);
    for (int32_t v = prev + 1; v <= cur; ++v) off[v] = (int32_t)p;
  }
}
__global__ __launch_bounds__(kBlock) void run_max_kernel(const int32_t* __restrict__ off, int nvalues,
                                                        int32_t* __restrict__ max_len) {
  int m = 0;
  for (int v = blockIdx.x * kBlock + threadIdx.x; v < nvalues; v += gridDim.x * kBlock) m = max(m, off[v + 1] - off[v]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(max_len, m);
}
// cent[v] = sums[v] / |sums[v]| for a list that has members (off[v + 1] > off[v]), else the unit vector of training row
// fallback[v] (spherical k-means: an empty list takes a random training row).  One 64-lane wave per list.
__global__ __launch_bounds__(kBlock) void ivf_centroids_kernel(const float* __restrict__ sums, const int32_t* __restrict__ off,
                                                              const float* __restrict__ train,
                                                              const int32_t* __restrict__ fallback, int nlist, int D,
                                                              float* __restrict__ cent) {
  const int v = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (v >= nlist) return;
  const float* src = (off == nullptr || off[v + 1] > off[v]) ? sums + (int64_t)v * D : train + (int64_t)fallback[v] * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss = fmaf(src[d], src[d], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-30f);
  for (int d = lane; d < D; d += 64) cent[(int64_t)v * D + d] = src[d] * inv;
}

}  // namespace esr

using namespace esr;

// the pairs of one set (lists [P] = cq x ppq probe slots, the first being slot0): sort by list, tile, score
template <bool FILTER>
static int ivf_score_set(const float* queries, int D, const float* cands_sorted, const int32_t* list_off, int nlist,
                         const int32_t* lists, int64_t P, int ppq, int slot0, int64_t q0, const IvfGeom& g,
                         const IvfWs& ws, esr_stream_t stream) {
  hipStream_t st = as_stream(stream);
  if (int rc = esr_segment_sort_ids(lists, P, nlist, ws.sorted_lists, ws.perm, ws.sort_ws, ws.sort_ws_bytes, stream))
    return rc;
  hipLaunchKernelGGL(ivf_prep_kernel, dim3(1), dim3(1024), 0, st, (const int32_t*)ws.sorted_lists, P, nlist, ws.pair_off,
                     ws.tile_start);
  const int row_tiles = (int)(P / kIvfTile + nlist);  // >= sum over lists of ceil(pairs / 64)
  IvfFilter flt;
  flt.tau = ws.tau; flt.cnt = ws.cnt; flt.pairs = ws.pairs; flt.ppitch = g.ppitch;
  ESR_KT("ivf_score_kernel", st, hipLaunchKernelGGL((ivf_score_kernel<FILTER>), dim3((int)align_up((size_t)(g.pitch / kIvfTile), 8), row_tiles),
                     dim3(kBlock), 0, st, queries, D, cands_sorted, list_off, (const int32_t*)ws.perm, ppq, slot0, q0,
                     (const int32_t*)ws.pair_off, (const int32_t*)ws.tile_start, nlist, ws.S, g.pitch, flt));
  return ESR_OK;
}

extern "C" {

size_t esr_ivf_search_workspace_bytes(int64_t nq, int nlist, int max_list, int nprobe, int k) {
  if (nq <= 0 || nlist <= 0 || max_list <= 0 || nprobe <= 0 || k <= 0) return 256;
  const IvfGeom g = ivf_geom(nq, max_list, nprobe, k, nlist);
  return ivf_layout(g, g.chunk, nprobe, nlist, nullptr, nullptr);
}

int esr_ivf_search(const float* queries, int64_t nq, int D, const float* cands_sorted, const int32_t* list_off,
                   const int32_t* orig, int nlist, int max_list, const int32_t* probe_lists, int nprobe, int k,
                   float* out_scores, int32_t* out_indices, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_ivf_search");
  ESR_REQUIRE(nlist <= 32768, "esr_ivf_search: nlist=%d exceeds 32768 (the score kernel's row tiles are a grid dimension)", nlist);
  ESR_REQUIRE(nq > 0 && D > 0 && D % 4 == 0 && nlist > 0 && max_list > 0 && nprobe > 0 && nprobe <= nlist && k > 0 &&
                  k <= kSelectMaxK && (int64_t)nprobe * k < ((int64_t)1 << 24),
              "esr_ivf_search: bad sizes nq=%lld D=%d nlist=%d max_list=%d nprobe=%d k=%d", (long long)nq, D, nlist,
              max_list, nprobe, k);
  ESR_REQUIRE(queries && cands_sorted && list_off && orig && probe_lists && out_scores && out_indices && workspace,
              "esr_ivf_search: null pointer");
  if (workspace_bytes < esr_ivf_search_workspace_bytes(nq, nlist, max_list, nprobe, k) || ((uintptr_t)workspace & 15)) {
    set_error("esr_ivf_search: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_ivf_search_workspace_bytes(nq, nlist, max_list, nprobe, k));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const IvfGeom g = ivf_geom(nq, max_list, nprobe, k, nlist);
  ESR_REQUIRE((int64_t)nprobe * g.pitch < ((int64_t)1 << 31), "esr_ivf_search: nprobe x longest list exceeds 2^31");
  IvfWs ws;
  ivf_layout(g, g.chunk, nprobe, nlist, (char*)workspace, &ws);
  const int f = g.f;
  auto take_slots = [&](int64_t q0, int64_t cq, int s0, int n) {
    hipLaunchKernelGGL(ivf_take_slots_kernel, dim3((int)std::min<int64_t>(kMaxGrid, cdiv(cq * n, kBlock))), dim3(kBlock),
                       0, st, probe_lists + q0 * nprobe, cq, nprobe, s0, n, ws.set_lists);
  };
  for (int64_t q0 = 0; q0 < nq; q0 += g.chunk) {
    const int64_t cq = std::min(g.chunk, nq - q0);
    // head: the best list(s) of every query, dense; their k best open the query's record list and give tau
    take_slots(q0, cq, 0, f);
    if (int rc = ivf_score_set<false>(queries, D, cands_sorted, list_off, nlist, ws.set_lists, cq * f, f, 0, q0, g, ws,
                                      stream))
      return rc;
    if (int rc = select_topk_head(ws.S, (int64_t)f * g.pitch, cq, (int)(f * g.pitch), k, ws.pairs, g.ppitch, ws.cnt,
                                  ws.tau, st))
      return rc;
    // filter: the other lists, a round of slots at a time, add what reaches tau; a select between rounds compacts
    for (int s0 = f; s0 < nprobe; s0 += kIvfRound) {
      const int n = std::min(kIvfRound, nprobe - s0);
      take_slots(q0, cq, s0, n);
      if (int rc = ivf_score_set<true>(queries, D, cands_sorted, list_off, nlist, ws.set_lists, cq * n, n, s0, q0, g, ws,
                                       stream))
        return rc;
      if (s0 + n < nprobe)
        if (int rc = select_topk_compact(ws.pairs, g.ppitch, ws.cnt, cq, k, ws.tau, st, g.mark)) return rc;
    }
    if (int rc = select_topk_tail(ws.pairs, g.ppitch, ws.cnt, cq, k, out_scores + q0 * k, out_indices + q0 * k, st))
      return rc;
    hipLaunchKernelGGL(ivf_map_kernel, dim3((int)std::min<int64_t>(kMaxGrid, cdiv(cq * k, kBlock))), dim3(kBlock), 0, st,
                       probe_lists + q0 * nprobe, cq, nprobe, k, g.pitch, list_off, orig,
                       (const float*)(out_scores + q0 * k), out_indices + q0 * k);
  }
  return check_launch("esr_ivf_search");
}


int esr_run_offsets(const int32_t* sorted, int64_t n, int nvalues, int32_t* off, int32_t* max_len, esr_stream_t stream) {
  TraceScope trace_scope_("esr_run_offsets");
  ESR_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && nvalues >= 1, "esr_run_offsets: bad sizes n=%lld nvalues=%d", (long long)n,
              nvalues);
  ESR_REQUIRE(off && (n == 0 || sorted), "esr_run_offsets: null pointer");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(run_offsets_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, cdiv(n + 1, kBlock))), dim3(kBlock), 0, st,
                     sorted, n, nvalues, off, max_len);
  if (max_len) {
    if (hipMemsetAsync(max_len, 0, sizeof(int32_t), st) != hipSuccess) return check_launch("esr_run_offsets");
    hipLaunchKernelGGL(run_max_kernel, dim3((unsigned)std::min<int64_t>(256, cdiv(nvalues, kBlock))), dim3(kBlock), 0, st,
                       (const int32_t*)off, nvalues, max_len);
  }
  return check_launch("esr_run_offsets");
}

int esr_ivf_centroids(const float* sums, const int32_t* list_off, const float* train, const int32_t* fallback_rows,
                      int nlist, int D, float* centroids, esr_stream_t stream) {
  TraceScope trace_scope_("esr_ivf_centroids");
  ESR_REQUIRE(nlist >= 1 && D >= 1 && sums && centroids, "esr_ivf_centroids: bad arguments");
  ESR_REQUIRE(list_off == nullptr || (train && fallback_rows), "esr_ivf_centroids: empty lists need training rows to fall back on");
  hipLaunchKernelGGL(ivf_centroids_kernel, dim3((unsigned)cdiv(nlist, kBlock / 64)), dim3(kBlock), 0, as_stream(stream), sums,
                     list_off, train, fallback_rows, nlist, D, centroids);
  return check_launch("esr_ivf_centroids");
}

}  // extern "C"
