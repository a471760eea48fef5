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
//   score     grouped FP32 GEMM: a workgroup owns a 64 pairs x 64 candidates tile of ONE list (exact f32 products, fmaf in
//             k order: what is approximate about the answer is only WHICH lists are looked at)
//   select    radix select of the k best per pair (ragged rows: a pair's row is as long as its list)
//   map       position in list -> candidate row; then the nprobe lists of a query are merged (esr_topk_merge)
//
// Work: 2 nq nprobe (N / nlist) D flop instead of 2 nq N D: nlist / nprobe times less (64x at nlist 1024, nprobe 16).
#include "esr_common.h"

namespace esr {

constexpr int kIvfTile = 64;  // pairs x candidates per workgroup
constexpr int kIvfK = 16;     // embedding columns per LDS stage

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

// per pair: the length of its list (the ragged select's row length); outputs pre-filled for lists shorter than k
__global__ __launch_bounds__(kBlock) void ivf_pairs_kernel(const int32_t* __restrict__ lists, int64_t P,
                                                          const int32_t* __restrict__ list_off, int32_t* __restrict__ npr,
                                                          float* __restrict__ pair_scores, int32_t* __restrict__ pair_idx,
                                                          int k) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < P * k; i += (int64_t)gridDim.x * kBlock) {
    pair_scores[i] = -INFINITY;
    pair_idx[i] = -1;
    if (i < P) {
      const int l = lists[i];
      npr[i] = list_off[l + 1] - list_off[l];
    }
  }
}

// grouped GEMM tile: blockIdx.x = row tile over all lists (tile_start locates its list), blockIdx.y = candidate tile
__global__ __launch_bounds__(kBlock) void ivf_score_kernel(const float* __restrict__ queries, int D,
                                                          const float* __restrict__ cands,
                                                          const int32_t* __restrict__ list_off,
                                                          const int32_t* __restrict__ perm, int nprobe, int64_t q_base,
                                                          const int32_t* __restrict__ pair_off,
                                                          const int32_t* __restrict__ tile_start, int nlist,
                                                          float* __restrict__ S, int64_t pitch) {
  __shared__ float As[kIvfK][kIvfTile + 4];
  __shared__ float Bs[kIvfK][kIvfTile + 4];
  __shared__ int s_pair[kIvfTile];
  const int rt = blockIdx.x;
  if (rt >= tile_start[nlist]) return;
  int lo = 0, hi = nlist - 1;  // the last list whose first tile is <= rt and that HAS tiles
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= rt) lo = mid; else hi = mid - 1;
  }
  const int l = lo;
  const int L = list_off[l + 1] - list_off[l];
  const int c0 = blockIdx.y * kIvfTile;
  if (c0 >= L) return;
  const int row0 = pair_off[l] + (rt - tile_start[l]) * kIvfTile;
  const int nrows = min(kIvfTile, pair_off[l + 1] - row0);
  const int t = threadIdx.x;
  if (t < kIvfTile) s_pair[t] = t < nrows ? perm[row0 + t] : -1;
  __syncthreads();
  const int lr = t >> 2, seg = t & 3;  // loader: row lr, 4 floats at column 4 * seg of the stage
  const int my_pair = s_pair[lr];
  const float* qrow = my_pair >= 0 ? queries + (q_base + my_pair / nprobe) * (int64_t)D : nullptr;
  const float* crow = c0 + lr < L ? cands + ((int64_t)list_off[l] + c0 + lr) * D : nullptr;
  const int ty = t >> 4, tx = t & 15;  // computer: pairs 4 ty .. + 3, candidates 4 tx .. + 3
  float acc[4][4] = {};
  for (int k0 = 0; k0 < D; k0 += kIvfK) {
    const int kc = k0 + 4 * seg;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (qrow && kc < D) a = *reinterpret_cast<const float4*>(qrow + kc);
    if (crow && kc < D) b = *reinterpret_cast<const float4*>(crow + kc);
    __syncthreads();  // (the previous stage has been consumed)
    As[4 * seg + 0][lr] = a.x; As[4 * seg + 1][lr] = a.y; As[4 * seg + 2][lr] = a.z; As[4 * seg + 3][lr] = a.w;
    Bs[4 * seg + 0][lr] = b.x; Bs[4 * seg + 1][lr] = b.y; Bs[4 * seg + 2][lr] = b.z; Bs[4 * seg + 3][lr] = b.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kIvfK; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(&As[kk][4 * ty]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[kk][4 * tx]);
      const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ar[i], br[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = s_pair[4 * ty + i];
    if (p < 0) continue;
    float* out = S + (int64_t)p * pitch + c0 + 4 * tx;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + 4 * tx + j < L) out[j] = acc[i][j];
  }
}

// position in list -> candidate row of the caller's matrix, for every (pair, rank)
__global__ __launch_bounds__(kBlock) void ivf_map_kernel(const int32_t* __restrict__ lists, int64_t P, int k,
                                                        const int32_t* __restrict__ list_off,
                                                        const int32_t* __restrict__ orig, int32_t* __restrict__ pair_idx) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < P * k; i += (int64_t)gridDim.x * kBlock) {
    const int32_t c = pair_idx[i];
    if (c >= 0) pair_idx[i] = orig[list_off[lists[i / k]] + c];
  }
}

struct IvfWs {
  int32_t* sorted_lists;  // [P]
  int32_t* perm;          // [P]
  int32_t* pair_off;      // [nlist + 1]
  int32_t* tile_start;    // [nlist + 1]
  int32_t* npr;           // [P]
  float* pair_scores;     // [P, k]
  int32_t* pair_idx;      // [P, k]
  float* S;               // [P, pitch]
  void* sort_ws;
  size_t sort_ws_bytes;
};
static size_t ivf_layout(int64_t P, int nlist, int64_t pitch, int k, char* base, IvfWs* out) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  IvfWs w;
  w.sorted_lists = (int32_t*)take(4 * (size_t)P);
  w.perm = (int32_t*)take(4 * (size_t)P);
  w.pair_off = (int32_t*)take(4 * (size_t)(nlist + 1));
  w.tile_start = (int32_t*)take(4 * (size_t)(nlist + 1));
  w.npr = (int32_t*)take(4 * (size_t)P);
  w.pair_scores = (float*)take(4 * (size_t)P * k);
  w.pair_idx = (int32_t*)take(4 * (size_t)P * k);
  w.S = (float*)take(4 * (size_t)P * (size_t)pitch);
  w.sort_ws_bytes = esr_segment_sort_workspace_bytes(P);
  w.sort_ws = take(w.sort_ws_bytes);
  if (out) *out = w;
  return off;
}

// queries per pass: the score block [chunk * nprobe, pitch] floats stays under ~1 GiB
static int64_t ivf_chunk(int64_t nq, int nprobe, int64_t pitch) {
  const int64_t rows = std::max<int64_t>(1, ((int64_t)1 << 28) / std::max<int64_t>(1, pitch * nprobe));
  return std::min(nq, std::max<int64_t>(64, rows));
}

}  // namespace esr

using namespace esr;

extern "C" {

size_t esr_ivf_search_workspace_bytes(int64_t nq, int nlist, int max_list, int nprobe, int k) {
  if (nq <= 0 || nlist <= 0 || max_list <= 0 || nprobe <= 0 || k <= 0) return 256;
  const int64_t pitch = align_up((size_t)max_list, 64);
  const int64_t chunk = ivf_chunk(nq, nprobe, pitch);
  return ivf_layout(chunk * nprobe, nlist, pitch, k, nullptr, nullptr);
}

int esr_ivf_search(const float* queries, int64_t nq, int D, const float* cands_sorted, const int32_t* list_off,
                   const int32_t* orig, int nlist, int max_list, const int32_t* probe_lists, int nprobe, int k,
                   float* out_scores, int32_t* out_indices, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
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
  const int64_t pitch = align_up((size_t)max_list, 64);
  const int64_t chunk = ivf_chunk(nq, nprobe, pitch);
  IvfWs ws;
  ivf_layout(chunk * nprobe, nlist, pitch, k, (char*)workspace, &ws);
  for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
    const int64_t cq = std::min(chunk, nq - q0), P = cq * nprobe;
    const int32_t* lists = probe_lists + q0 * nprobe;
    if (int rc = esr_segment_sort_ids(lists, P, nlist, ws.sorted_lists, ws.perm, ws.sort_ws, ws.sort_ws_bytes, stream))
      return rc;
    hipLaunchKernelGGL(ivf_prep_kernel, dim3(1), dim3(1024), 0, st, (const int32_t*)ws.sorted_lists, P, nlist, ws.pair_off,
                       ws.tile_start);
    hipLaunchKernelGGL(ivf_pairs_kernel, dim3((int)std::min<int64_t>(kMaxGrid, cdiv(P * k, kBlock))), dim3(kBlock), 0, st,
                       lists, P, list_off, ws.npr, ws.pair_scores, ws.pair_idx, k);
    const int row_tiles = (int)(P / kIvfTile + nlist);  // >= sum over lists of ceil(pairs / 64)
    hipLaunchKernelGGL(ivf_score_kernel, dim3(row_tiles, (int)(pitch / kIvfTile)), dim3(kBlock), 0, st, queries, D,
                       cands_sorted, list_off, (const int32_t*)ws.perm, nprobe, q0, (const int32_t*)ws.pair_off,
                       (const int32_t*)ws.tile_start, nlist, ws.S, pitch);
    if (int rc = select_topk_ragged(ws.S, pitch, P, ws.npr, max_list, k, ws.pair_scores, ws.pair_idx, st)) return rc;
    hipLaunchKernelGGL(ivf_map_kernel, dim3((int)std::min<int64_t>(kMaxGrid, cdiv(P * k, kBlock))), dim3(kBlock), 0, st,
                       lists, P, k, list_off, orig, ws.pair_idx);
    if (int rc = esr_topk_merge(ws.pair_scores, ws.pair_idx, cq, nprobe * k, k, out_scores + q0 * k, out_indices + q0 * k,
                                stream))
      return rc;
  }
  return check_launch("esr_ivf_search");
}

}  // extern "C"
