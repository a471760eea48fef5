// Everything that needs a device-wide stable sort -- all of it this file's own kernels (round 5: the four call sites of the
// ROCm device-library radix sort that served lists beyond 2 M ids are gone; no library sort is left):
//   esr_segment_sort_ids      occurrence ids -> (sorted ids, permutation) for the sparse optimizers
//   esr_argsort_columns       find_knn's jnp.argsort(scores, axis=0)  (train_cooccurence.py:96)
//   esr_score_topk            find_top_k's jax.lax.top_k              (make_recommendations.py:64)
//   esr_bucket_ids_by_owner   row-shard routing (owner = id mod world)
// One workgroup (bitonic), tile sort + rank merge, or a least-significant-digit radix sort with 11-bit digits -- by size.
#include "esr_common.h"

#include <algorithm>
#include <cstring>

namespace esr {

static inline int bits_for(int64_t n_values) {  // bits needed to represent values in [0, n_values)
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) < n_values) ++b;
  return b;
}

// The occurrence ids may come as up to four segments [ids_k + offset_k] (the towers of one step as virtual rows of
// their concatenation): the sort kernels read them in place, so no concatenated copy is written first.
constexpr int kMaxSortSegs = 4;
struct SortSegs {
  const int32_t* ids[kMaxSortSegs];
  int64_t start[kMaxSortSegs + 1];  // position of each segment in the virtual list
  int64_t offset[kMaxSortSegs];     // added to every id of the segment
  int n;
};
__device__ __forceinline__ int32_t seg_id(const SortSegs& sg, int64_t i) {
  const int32_t* src = sg.ids[0];
  int64_t start = 0, off = sg.offset[0];
#pragma unroll
  for (int k = 1; k < kMaxSortSegs; ++k)
    if (k < sg.n && i >= sg.start[k]) {
      src = sg.ids[k];
      start = sg.start[k];
      off = sg.offset[k];
    }
  return (int32_t)((int64_t)src[i - start] + off);
}
__global__ __launch_bounds__(kBlock) void concat_segs_kernel(SortSegs sg, int64_t n, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    out[i] = seg_id(sg, i);
}

// Short occurrence lists (one playlist of the Spotify step, the reference's own batch sizes of 16-128): one
// workgroup, bitonic sort of (id << 32 | position) in LDS -- stable by construction, one launch instead of the
// device radix sort's chain of ~6.
constexpr int kSmallSortMax = 4096;
constexpr int kSmallSortThreads = 1024;
__global__ __launch_bounds__(kSmallSortThreads) void segment_sort_small_kernel(SortSegs ids, int n,
                                                                              int32_t* __restrict__ sorted_ids,
                                                                              int32_t* __restrict__ perm) {
  __shared__ unsigned long long key[kSmallSortMax];
  const int t = threadIdx.x;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = t; i < np2; i += kSmallSortThreads)
    key[i] = i < n ? (((unsigned long long)(uint32_t)seg_id(ids, i)) << 32) | (uint32_t)i : ~0ull;
  for (int size = 2; size <= np2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = t; i < (np2 >> 1); i += kSmallSortThreads) {
        const int pos = 2 * i - (i & (stride - 1)), j = pos + stride;
        const bool asc = (pos & size) == 0;
        const unsigned long long x = key[pos], y = key[j];
        if ((x > y) == asc) {
          key[pos] = y;
          key[j] = x;
        }
      }
    }
  __syncthreads();
  for (int i = t; i < n; i += kSmallSortThreads) {
    sorted_ids[i] = (int32_t)(key[i] >> 32);
    perm[i] = (int32_t)(uint32_t)key[i];
  }
}

// Mid-size lists (4096 < n <= 32768, ids < 2^21: the occurrence ids of one C2 step): two launches instead of the
// radix sort's ~6.  (1) every workgroup bitonic-sorts one tile (512 / 1024 / 2048 keys) of 32-bit composites
// (id << log2(tile) | position in tile) in LDS; (2) every element finds its final rank = its index in its own tile + for
// each EARLIER tile the number of ids <= its id + for each LATER tile the number of ids < its id (binary searches
// that advance in lockstep so their dependent loads overlap).  Ranks are exact and the sort is stable.
// The tile is the smallest power of two >= 512 that covers the list with kMidTiles tiles (16 384 ids: 16 tiles of 1024
// -- sixteen workgroups sort instead of eight and every one of the rank kernel's 16 lanes per element has a tile to
// search; 2048-key tiles at every size left half of them idle there: 10.9 + 6.3 us -> see profiles/).
constexpr int kMidTiles = 16, kMaxTileBits = 11;
// element slots per 16-lane group of the rank kernels (see tile_rank_body): as many (<= 8) as leave >= 2048 workgroups
static inline int rank_reps(int64_t blocks) {
  int r = 16;
  while (r > 1 && blocks / r < 1024) r >>= 1;
  return r;
}
// One splitter (its last key) per 8-key block of a sorted tile.  (32-key blocks until round 5: the rank kernel then
// finished every search with five DEPENDENT probes of global memory inside the block; with 8-key blocks the level
// staged in LDS is four times as long -- 16 KB for sixteen 2048-key tiles -- and what is left is one 32-byte block,
// fetched by two loads that are in flight together and counted without a search.)
constexpr int kSplitEvery = 8;
constexpr int kMidSortMax = (1 << kMaxTileBits) * kMidTiles;
// Bitonic network over kTile = 2^TB keys held two per thread: thread t holds positions t (k0) and t + kTile / 2 (k1).
// A compare-exchange with stride < 64 has its partner in the same wave (lane ^ stride) and is one cross-lane move per key
// -- 51 of the 66 steps of a 2048-key tile; the stride kTile / 2 pairs the thread's own two registers; only the strides
// in between go through LDS and a barrier (14 steps).  With every step in LDS behind a barrier the 2048-key tile took
// 11 us.  `key` = 2 * kTile words of LDS.
template <int TB>
__device__ __forceinline__ void tile_bitonic(uint32_t& k0, uint32_t& k1, uint32_t* key) {
  constexpr int kTile = 1 << TB, kHalf = kTile / 2;
  const int t = threadIdx.x;
  int flip = 0;
  // new value of the key at position p after meeting its partner's key o at distance `stride` inside a run of `size`
  auto cx = [](uint32_t mine, uint32_t other, int p, int stride, int size) {
    const bool lower = (p & stride) == 0, asc = (p & size) == 0;
    const uint32_t mn = mine < other ? mine : other, mx = mine < other ? other : mine;
    return lower == asc ? mn : mx;
  };
#pragma unroll
  for (int size = 2; size <= kTile; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride == kHalf) {  // only for size == kTile: ascending, partner = the other register
        const uint32_t mn = k0 < k1 ? k0 : k1, mx = k0 < k1 ? k1 : k0;
        k0 = mn;
        k1 = mx;
      } else if (stride >= 64) {
        // two LDS images used alternately: the next exchange writes the other one, so a single barrier per step
        // orders everything (a thread can be at most one step ahead of the slowest reader)
        uint32_t* img = key + (flip ? kTile : 0);
        flip ^= 1;
        img[t] = k0;
        img[t + kHalf] = k1;
        __syncthreads();
        const uint32_t o0 = img[t ^ stride], o1 = img[(t ^ stride) + kHalf];
        k0 = cx(k0, o0, t, stride, size);
        k1 = cx(k1, o1, t + kHalf, stride, size);
      } else {
        const uint32_t o0 = xor_lane(k0, stride), o1 = xor_lane(k1, stride);
        k0 = cx(k0, o0, t, stride, size);
        k1 = cx(k1, o1, t + kHalf, stride, size);
      }
    }
  }
}

// (bodies shared by the one-list kernels and the batched ones, where blockIdx.y picks the list: `bx` = the workgroup's
// index inside its own list)
constexpr int kMaxSortBatch = 8;
struct SortSegsBatch {
  SortSegs b[kMaxSortBatch];
};
template <int TB>
__device__ __forceinline__ void tile_sort_body(const SortSegs& ids, int n, uint32_t* __restrict__ tiles,
                                               uint32_t* __restrict__ splitters, uint32_t* key, int bx) {
  constexpr int kTile = 1 << TB, kHalf = kTile / 2;
  const int t = threadIdx.x, base = bx * kTile;
  uint32_t k0, k1;
  {
    const int g0 = base + t, g1 = base + t + kHalf;
    k0 = g0 < n ? ((uint32_t)seg_id(ids, g0) << TB) | (uint32_t)t : 0xFFFFFFFFu;
    k1 = g1 < n ? ((uint32_t)seg_id(ids, g1) << TB) | (uint32_t)(t + kHalf) : 0xFFFFFFFFu;
  }
  tile_bitonic<TB>(k0, k1, key);
  tiles[base + t] = k0;
  tiles[base + t + kHalf] = k1;
  // the last key of every 32-key block, packed: the rank kernel stages these (it used to gather them from the tiles,
  // one cache line per splitter and workgroup)
  if ((t & (kSplitEvery - 1)) == kSplitEvery - 1) {
    constexpr int kSplitPerTile = kTile / kSplitEvery;
    splitters[bx * kSplitPerTile + t / kSplitEvery] = k0;
    splitters[bx * kSplitPerTile + (t + kHalf) / kSplitEvery] = k1;
  }
}
template <int TB>
__global__ __launch_bounds__((1 << TB) / 2) void tile_sort_kernel(SortSegs ids, int n, uint32_t* __restrict__ tiles,
                                                                 uint32_t* __restrict__ splitters) {
  __shared__ uint32_t key[2 << TB];
  tile_sort_body<TB>(ids, n, tiles, splitters, key, blockIdx.x);
}
// ws_stride: words between the (tiles | splitters) areas of consecutive lists
template <int TB>
__global__ __launch_bounds__((1 << TB) / 2) void tile_sort_batched_kernel(SortSegsBatch sb, int n,
                                                                         uint32_t* __restrict__ tiles,
                                                                         uint32_t* __restrict__ splitters,
                                                                         int64_t ws_stride) {
  __shared__ uint32_t key[2 << TB];
  const int y = blockIdx.y;
  tile_sort_body<TB>(sb.b[y], n, tiles + y * ws_stride, splitters + y * ws_stride, key, blockIdx.x);
}

// A list that fits ONE tile: the same network, the sorted composites unpacked straight into (sorted ids, perm) -- one
// launch, ~5 us.  The one-workgroup kernel above (64-bit keys in LDS, a barrier per step) took 39 us for the 4096 ids of a
// GloVe step at the reference's default batch (wikipedia/train_cooccurence.py:45): more than the rest of that step.
template <int TB>
__device__ __forceinline__ void tile_sort_single_body(const SortSegs& ids, int n, int32_t* __restrict__ sorted_ids,
                                                      int32_t* __restrict__ perm, uint32_t* key) {
  constexpr int kTile = 1 << TB, kHalf = kTile / 2;
  const int t = threadIdx.x;
  uint32_t k0 = t < n ? ((uint32_t)seg_id(ids, t) << TB) | (uint32_t)t : 0xFFFFFFFFu;
  uint32_t k1 = t + kHalf < n ? ((uint32_t)seg_id(ids, t + kHalf) << TB) | (uint32_t)(t + kHalf) : 0xFFFFFFFFu;
  tile_bitonic<TB>(k0, k1, key);
  if (t < n) {
    sorted_ids[t] = (int32_t)(k0 >> TB);
    perm[t] = (int32_t)(k0 & (kTile - 1));
  }
  if (t + kHalf < n) {
    sorted_ids[t + kHalf] = (int32_t)(k1 >> TB);
    perm[t + kHalf] = (int32_t)(k1 & (kTile - 1));
  }
}
template <int TB>
__global__ __launch_bounds__((1 << TB) / 2) void tile_sort_single_kernel(SortSegs ids, int n,
                                                                        int32_t* __restrict__ sorted_ids,
                                                                        int32_t* __restrict__ perm) {
  __shared__ uint32_t key[2 << TB];
  tile_sort_single_body<TB>(ids, n, sorted_ids, perm, key);
}
template <int TB>
__global__ __launch_bounds__((1 << TB) / 2) void tile_sort_single_batched_kernel(SortSegsBatch sb, int n,
                                                                                int32_t* __restrict__ sorted_ids,
                                                                                int32_t* __restrict__ perm) {
  __shared__ uint32_t key[2 << TB];
  const int y = blockIdx.x;
  tile_sort_single_body<TB>(sb.b[y], n, sorted_ids + (int64_t)y * n, perm + (int64_t)y * n, key);
}

// Two-level search: the last id of every 32-key block of every tile ("splitters", <= 4 KB) is staged in LDS and
// searched there; only the final 32-key window -- one 128-byte line -- is searched in global memory.  A plain
// binary search over the tiles touched ~12 scattered lines per (element, tile) and was bound by L1 line rate.
// 16 lanes per element, one per tile: the searches of one element run side by side and their counts are summed
// with shuffles (one thread walking all tiles was latency-bound at one wave per SIMD: 26 us for 24 576 keys).
template <int TB>
__device__ __forceinline__ void tile_rank_body(const uint32_t* __restrict__ tiles,
                                               const uint32_t* __restrict__ splitters, int n, int ntiles,
                                               int32_t* __restrict__ sorted_ids, int32_t* __restrict__ perm,
                                               uint32_t* spl, int bx, int kRankReps) {
  constexpr int kTile = 1 << TB, kSplitPerTile = kTile / kSplitEvery;
  // (rows padded by one word: the 16 lanes of a group search 16 tiles at the same depth -- with a stride of 32 or 64 words
  // every probe of every level was a 16-way bank conflict, 77 % of the kernel's LDS cycles)
  constexpr int kSplStride = kSplitPerTile + 1;
  for (int i = threadIdx.x; i < ntiles * kSplitPerTile; i += kBlock)
    spl[(i / kSplitPerTile) * kSplStride + (i % kSplitPerTile)] = splitters[i] >> TB;
  __syncthreads();
  const int u = threadIdx.x & (kMidTiles - 1);                                 // the tile this lane searches
  // kRankReps element slots per 16-lane group: the splitters are staged once per workgroup for up to 256 elements instead
  // of 16 (round 4: the eight 24 576-id lists of a group of triplet batches were 12 288 workgroups staging 3 KB each)
#pragma unroll 1
  for (int rep = 0; rep < kRankReps; ++rep) {
  const int g = ((bx * kRankReps + rep) * kBlock + threadIdx.x) / kMidTiles;   // slot g of the tiled array
  const int mine = g / kTile;
  const bool live = g < ntiles * kTile && g - mine * kTile < n - mine * kTile;  // not padding
  const uint32_t c = live ? tiles[g] : 0u;
  const uint32_t id = c >> TB;
  int cnt = 0;
  if (live && u < ntiles && u != mine) {
    int lo = 0, hi = kSplitPerTile;  // level 1 (LDS): whole 8-key blocks that precede this element
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const uint32_t x = spl[u * kSplStride + mid];
      if (u < mine ? x <= id : x < id) lo = mid + 1; else hi = mid;  // earlier tiles win ties
    }
    const int base = lo * kSplitEvery;
    int lo2 = 0;  // level 2 (global): the keys of that block that precede the element -- the block is sorted: a count
    if (base < kTile) {
      const uint4 a = *reinterpret_cast<const uint4*>(tiles + u * kTile + base);
      const uint4 b = *reinterpret_cast<const uint4*>(tiles + u * kTile + base + 4);
      const uint32_t lim = u < mine ? id : id - 1u;  // x <= id, or x < id  <=>  x <= id - 1 (id == 0: nothing precedes)
      if (u < mine || id != 0u)
        lo2 = (int)((a.x >> TB) <= lim) + (int)((a.y >> TB) <= lim) + (int)((a.z >> TB) <= lim) + (int)((a.w >> TB) <= lim) +
              (int)((b.x >> TB) <= lim) + (int)((b.y >> TB) <= lim) + (int)((b.z >> TB) <= lim) + (int)((b.w >> TB) <= lim);
    }
    cnt = min(base + lo2, n - u * kTile);  // never count the padding
  }
#pragma unroll
  for (int o = kMidTiles / 2; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, kMidTiles);
  if (live && u == 0) {
    const int rank = g - mine * kTile + cnt;
    sorted_ids[rank] = (int32_t)id;
    perm[rank] = mine * kTile + (int)(c & (kTile - 1));
  }
  }  // rep
}
template <int TB>
__global__ __launch_bounds__(kBlock) void tile_rank_kernel(const uint32_t* __restrict__ tiles,
                                                          const uint32_t* __restrict__ splitters, int n, int ntiles,
                                                          int32_t* __restrict__ sorted_ids,
                                                          int32_t* __restrict__ perm, int reps) {
  __shared__ uint32_t spl[kMidTiles * ((1 << TB) / kSplitEvery + 1)];
  tile_rank_body<TB>(tiles, splitters, n, ntiles, sorted_ids, perm, spl, blockIdx.x, reps);
}
template <int TB>
__global__ __launch_bounds__(kBlock) void tile_rank_batched_kernel(const uint32_t* __restrict__ tiles,
                                                                  const uint32_t* __restrict__ splitters, int n,
                                                                  int ntiles, int64_t ws_stride,
                                                                  int32_t* __restrict__ sorted_ids,
                                                                  int32_t* __restrict__ perm, int reps) {
  __shared__ uint32_t spl[kMidTiles * ((1 << TB) / kSplitEvery + 1)];
  const int y = blockIdx.y;
  tile_rank_body<TB>(tiles + y * ws_stride, splitters + y * ws_stride, n, ntiles, sorted_ids + (int64_t)y * n,
                     perm + (int64_t)y * n, spl, blockIdx.x, reps);
}
template <int TB>
static void launch_tile_sort(const SortSegs& sg, int n, uint32_t* tiles, int32_t* sorted_ids, int32_t* perm,
                             hipStream_t st) {
  constexpr int kTile = 1 << TB;
  const int ntiles = (int)cdiv(n, kTile);
  uint32_t* splitters = tiles + kMidSortMax;  // [ntiles][kTile / 32], behind the tiles
  hipLaunchKernelGGL((tile_sort_kernel<TB>), dim3(ntiles), dim3(kTile / 2), 0, st, sg, n, tiles, splitters);
  const int64_t blocks = cdiv((int64_t)ntiles * kTile * kMidTiles, kBlock);
  const int reps = rank_reps(blocks);
  hipLaunchKernelGGL((tile_rank_kernel<TB>), dim3((int)cdiv(blocks, reps)), dim3(kBlock), 0, st, (const uint32_t*)tiles,
                     (const uint32_t*)splitters, n, ntiles, sorted_ids, perm, reps);
}

constexpr int64_t kMidWsWords = kMidSortMax + kMidSortMax / kSplitEvery;  // one list's (tiles | splitters) area
template <int TB>
static void launch_tile_sort_batched(const SortSegsBatch& sb, int nbatch, int n, uint32_t* tiles, int32_t* sorted_ids,
                                     int32_t* perm, hipStream_t st) {
  constexpr int kTile = 1 << TB;
  const int ntiles = (int)cdiv(n, kTile);
  uint32_t* splitters = tiles + kMidSortMax;
  hipLaunchKernelGGL((tile_sort_batched_kernel<TB>), dim3(ntiles, nbatch), dim3(kTile / 2), 0, st, sb, n, tiles,
                     splitters, kMidWsWords);
  const int64_t blocks = cdiv((int64_t)ntiles * kTile * kMidTiles, kBlock);
  const int reps = rank_reps(blocks * nbatch);
  hipLaunchKernelGGL((tile_rank_batched_kernel<TB>), dim3((int)cdiv(blocks, reps), nbatch), dim3(kBlock), 0, st,
                     (const uint32_t*)tiles, (const uint32_t*)splitters, n, ntiles, kMidWsWords, sorted_ids, perm, reps);
}

// Long lists (32 768 < n <= 262 144: the 131 072 occurrence ids of a GloVe step at B = 65 536, the 196 608 of a triplet
// step at that batch): least-significant-digit radix sort with 11-bit digits, two launches per pass, two passes for ids
// below 2^22.  The ROCm library's device sort is a chain of ~10 short launches here (55 us for 131 072 keys, all launch latency).
//   pass launch 1: every workgroup bitonic-sorts its 2048-key tile by (digit << 11 | position in tile) -- stable -- writes
//                  the sorted composites and the tile's digit histogram (from the run boundaries: no atomics);
//   pass launch 2: every workgroup turns the histograms into its own digit offsets (digits below + same digit in earlier
//                  tiles: it re-reduces the whole [tiles x 2048] matrix, <= 1 MB out of L2) and scatters its tile;
//                  rank inside a digit = index in the sorted tile - first index of the digit.
// Stable by construction; pure integer work.
// The digit width is also the tile size (TB bits; 11: 64 workgroups of 1024 threads for 131 072 ids).  Measured and
// dropped: TB = 9 (256-thread workgroups, three passes) on a second stream BESIDE the GloVe update kernel, sized to fit
// the wave slots and registers that kernel leaves free -- the step went from 0.181 to 0.235 ms; capping the update
// kernel's residency to make room for the 1024-thread version cost more than the hidden sort returned (0.202 ms at
// three of four workgroups per CU).  What does pay is the plain sort of batch k + 1 on the second stream: it fills the
// gaps around batch k's short kernels (0.194 -> 0.181 ms).
constexpr int kRadixMaxN = 1 << 18;   // up to here a scatter workgroup re-reduces the histogram matrix itself
constexpr int kRadixLongN = 1 << 21;  // beyond kRadixMaxN: segment sums first (three launches per pass)
constexpr int kRadixMaxPasses = 3;

// FIRST: keys come from the id segments and the value is the position itself; else from (keys_in, vals_in)
template <int TB, bool FIRST>
__device__ __forceinline__ void radix_tile_body(const SortSegs& ids, const uint32_t* __restrict__ keys_in, int n,
                                                int shift, uint32_t* __restrict__ tiles, int32_t* __restrict__ hist) {
  constexpr int kTile = 1 << TB, kThreads = kTile / 2;
  __shared__ uint32_t key[2 * kTile];
  __shared__ int bstart[kTile], bend[kTile];
  const int t = threadIdx.x, base = blockIdx.x * kTile;
  constexpr uint32_t kMask = kTile - 1;
  uint32_t k0, k1;
  {
    const int g0 = base + t, g1 = base + t + kThreads;
    const uint32_t a = g0 < n ? (FIRST ? (uint32_t)seg_id(ids, g0) : keys_in[g0]) : 0u;
    const uint32_t b = g1 < n ? (FIRST ? (uint32_t)seg_id(ids, g1) : keys_in[g1]) : 0u;
    k0 = g0 < n ? (((a >> shift) & kMask) << TB) | (uint32_t)t : 0xFFFFFFFFu;
    k1 = g1 < n ? (((b >> shift) & kMask) << TB) | (uint32_t)(t + kThreads) : 0xFFFFFFFFu;
  }
  bstart[t] = 0;
  bstart[t + kThreads] = 0;
  bend[t] = 0;
  bend[t + kThreads] = 0;
  tile_bitonic<TB>(k0, k1, key);
  __syncthreads();  // the network's last LDS reads are done; `key` is free (and the zeroed histograms are visible)
  key[t] = k0;
  key[t + kThreads] = k1;
  tiles[base + t] = k0;
  tiles[base + t + kThreads] = k1;
  __syncthreads();
  const int live = min(kTile, n - base);  // padding (all ones) sorts to the end
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = t + h * kThreads;
    if (i < live) {
      const uint32_t d = key[i] >> TB;
      if (i == 0 || (key[i - 1] >> TB) != d) bstart[d] = i;
      if (i == live - 1 || (key[i + 1] >> TB) != d) bend[d] = i + 1;
    }
  }
  __syncthreads();
  hist[(int64_t)blockIdx.x * kTile + t] = bend[t] - bstart[t];
  hist[(int64_t)blockIdx.x * kTile + t + kThreads] = bend[t + kThreads] - bstart[t + kThreads];
}
template <int TB, bool FIRST>
__global__ __launch_bounds__((1 << TB) / 2) void radix_tile_kernel(SortSegs ids, const uint32_t* __restrict__ keys_in,
                                                                  int n, int shift, uint32_t* __restrict__ tiles,
                                                                  int32_t* __restrict__ hist) {
  radix_tile_body<TB, FIRST>(ids, keys_in, n, shift, tiles, hist);
}
// the same pass over the lists of a batch (blockIdx.y = list): every array of list b lies b * stride words further on
// (keys_in: b * io_stride)
template <int TB, bool FIRST>
__global__ __launch_bounds__((1 << TB) / 2) void radix_tile_batched_kernel(SortSegsBatch sb,
                                                                          const uint32_t* __restrict__ keys_in,
                                                                          int64_t io_stride, int n, int shift,
                                                                          uint32_t* __restrict__ tiles,
                                                                          int32_t* __restrict__ hist, int64_t stride) {
  const int b = blockIdx.y;
  radix_tile_body<TB, FIRST>(sb.b[b], FIRST ? nullptr : keys_in + b * io_stride, n, shift, tiles + b * stride,
                             hist + b * stride);
}

// long lists: per-segment column sums of the [tiles][digits] histogram matrix, one workgroup per (segment, 512 digits)
constexpr int kRadixSeg = 32;
template <int TB>
__global__ __launch_bounds__(kBlock) void radix_segsum_kernel(const int32_t* __restrict__ hist, int ntiles,
                                                             int32_t* __restrict__ segsum) {
  constexpr int kTile = 1 << TB;
  const int seg = blockIdx.y;
  const int d2 = blockIdx.x * kBlock + threadIdx.x;  // digit pair
  if (d2 >= kTile / 2) return;
  const int2* h2 = reinterpret_cast<const int2*>(hist);
  int a = 0, b = 0;
  const int u1 = min(ntiles, (seg + 1) * kRadixSeg);
#pragma unroll 8
  for (int u = seg * kRadixSeg; u < u1; ++u) {
    const int2 c = h2[(int64_t)u * (kTile / 2) + d2];
    a += c.x;
    b += c.y;
  }
  reinterpret_cast<int2*>(segsum)[(int64_t)seg * (kTile / 2) + d2] = make_int2(a, b);
}

// Batched sorts (throughput, not one list's latency): between the two launches of a pass, the histogram matrix of every
// list is scanned down its columns IN PLACE -- hist[tile][digit] becomes the digit's count in the tiles before `tile`,
// totals[digit] its count over all tiles -- so that a scatter workgroup reads two words per digit pair instead of
// re-reducing the whole matrix (eight lists of 131 072 ids: 512 workgroups x 512 KB out of L2, 34 us of the pass).
// One workgroup per (SD digits, list): 256 / SD tile groups x SD digits, the groups' sums combined through LDS
// (SD = 64: 4 groups, for matrices of up to 128 tiles; SD = 16: 16 groups for the taller ones).
template <int TB, int kScanDigits>
__global__ __launch_bounds__(kBlock) void radix_colscan_kernel(int32_t* __restrict__ hist, int ntiles,
                                                              int32_t* __restrict__ totals, int64_t stride) {
  constexpr int kTile = 1 << TB, kScanGroups = kBlock / kScanDigits;
  __shared__ int gsum[kScanGroups][kScanDigits];
  hist += (int64_t)blockIdx.y * stride;
  totals += (int64_t)blockIdx.y * stride;
  const int dg = threadIdx.x & (kScanDigits - 1), tg = threadIdx.x / kScanDigits;
  const int d = blockIdx.x * kScanDigits + dg;
  const int per = (ntiles + kScanGroups - 1) / kScanGroups;
  const int t0 = tg * per, t1 = min(ntiles, t0 + per);
  int sum = 0;
#pragma unroll 8
  for (int t = t0; t < t1; ++t) sum += hist[(int64_t)t * kTile + d];
  gsum[tg][dg] = sum;
  __syncthreads();
  int run = 0;
  for (int g = 0; g < tg; ++g) run += gsum[g][dg];
  if (tg == kScanGroups - 1) totals[d] = run + sum;
#pragma unroll 8
  for (int t = t0; t < t1; ++t) {
    const int c = hist[(int64_t)t * kTile + d];
    hist[(int64_t)t * kTile + d] = run;
    run += c;
  }
}

template <int TB, bool FIRST>
__device__ __forceinline__ void radix_scatter_body(const SortSegs& ids, const uint32_t* __restrict__ keys_in,
                                                   const uint32_t* __restrict__ vals_in, int n, int ntiles,
                                                   const uint32_t* __restrict__ tiles, const int32_t* __restrict__ hist,
                                                   uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                   const int32_t* __restrict__ segsum,
                                                   const int32_t* __restrict__ totals = nullptr) {
  constexpr int kTile = 1 << TB, kThreads = kTile / 2;
  __shared__ uint32_t comp[kTile];
  __shared__ int offs[kTile], bstart[kTile];
  __shared__ int wave_tot[kThreads / 64];
  const int t = threadIdx.x, tile = blockIdx.x, base = tile * kTile;
  comp[t] = tiles[base + t];
  comp[t + kThreads] = tiles[base + t + kThreads];
  // digits 2t and 2t + 1: totals over all tiles and over the tiles before this one
  // (unrolled: the loads of a pass over the matrix are independent; one at a time this loop WAS the kernel, 16 us)
  int tot0 = 0, tot1 = 0, pre0 = 0, pre1 = 0;
  const int2* h2 = reinterpret_cast<const int2*>(hist);
  if (totals) {
    // radix_colscan_kernel ran in between: `hist` holds, per digit, the count in the tiles BEFORE each tile, `totals`
    // the digit's count over all tiles -- two loads instead of a walk over the matrix
    const int2 tt = reinterpret_cast<const int2*>(totals)[t];
    const int2 pp = h2[(int64_t)tile * (kTile / 2) + t];
    tot0 = tt.x;
    tot1 = tt.y;
    pre0 = pp.x;
    pre1 = pp.y;
  } else if (segsum) {
    // long lists (> kRadixMaxN ids): the histogram matrix is too tall to re-reduce in every workgroup (384 tiles x 8 KB
    // per workgroup at 786 432 ids); radix_segsum_kernel summed it over segments of kRadixSeg tiles first
    const int2* s2 = reinterpret_cast<const int2*>(segsum);
    const int nseg = (ntiles + kRadixSeg - 1) / kRadixSeg, myseg = tile / kRadixSeg;
#pragma unroll 8
    for (int sgi = 0; sgi < nseg; ++sgi) {
      const int2 c = s2[(int64_t)sgi * (kTile / 2) + t];
      tot0 += c.x;
      tot1 += c.y;
      pre0 += sgi < myseg ? c.x : 0;
      pre1 += sgi < myseg ? c.y : 0;
    }
#pragma unroll 8
    for (int u = myseg * kRadixSeg; u < tile; ++u) {
      const int2 c = h2[(int64_t)u * (kTile / 2) + t];
      pre0 += c.x;
      pre1 += c.y;
    }
  } else {
#pragma unroll 32
    for (int u = 0; u < ntiles; ++u) {
      const int2 c = h2[(int64_t)u * (kTile / 2) + t];
      tot0 += c.x;
      tot1 += c.y;
      pre0 += u < tile ? c.x : 0;
      pre1 += u < tile ? c.y : 0;
    }
  }
  // exclusive scan of the totals: pair sums -> wave scan -> wave totals
  const int pair = tot0 + tot1;
  int incl = pair;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if ((t & 63) >= o) incl += v;
  }
  if ((t & 63) == 63) wave_tot[t >> 6] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < (t >> 6); ++w) wbase += wave_tot[w];
  const int excl = wbase + incl - pair;
  offs[2 * t] = excl + pre0;
  offs[2 * t + 1] = excl + tot0 + pre1;
  const int live = min(kTile, n - base);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = t + h * kThreads;
    if (i < live) {
      const uint32_t d = comp[i] >> TB;
      if (i == 0 || (comp[i - 1] >> TB) != d) bstart[d] = i;
    }
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = t + h * kThreads;
    if (i < live) {
      const uint32_t c = comp[i], d = c >> TB;
      const int src = base + (int)(c & (kTile - 1));
      const int dst = offs[d] + (i - bstart[d]);
      keys_out[dst] = FIRST ? (uint32_t)seg_id(ids, src) : keys_in[src];
      vals_out[dst] = FIRST ? (uint32_t)src : vals_in[src];
    }
  }
}
template <int TB, bool FIRST>
__global__ __launch_bounds__((1 << TB) / 2) void radix_scatter_kernel(SortSegs ids, const uint32_t* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in, int n,
                                                                     int ntiles, const uint32_t* __restrict__ tiles,
                                                                     const int32_t* __restrict__ hist,
                                                                     uint32_t* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ vals_out,
                                                                     const int32_t* __restrict__ segsum = nullptr) {
  radix_scatter_body<TB, FIRST>(ids, keys_in, vals_in, n, ntiles, tiles, hist, keys_out, vals_out, segsum);
}
// batched (blockIdx.y = list): tiles / hist of list b at b * stride words, keys / values in and out at b * in_stride /
// b * out_stride (the last pass writes the caller's [nbatch, n] arrays, the others the per-list workspace)
template <int TB, bool FIRST>
__global__ __launch_bounds__((1 << TB) / 2) void radix_scatter_batched_kernel(
    SortSegsBatch sb, const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, int64_t in_stride,
    int n, int ntiles, const uint32_t* __restrict__ tiles, const int32_t* __restrict__ hist,
    const int32_t* __restrict__ totals, int64_t stride, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int64_t out_stride) {
  const int b = blockIdx.y;
  radix_scatter_body<TB, FIRST>(sb.b[b], FIRST ? nullptr : keys_in + b * in_stride,
                                FIRST ? nullptr : vals_in + b * in_stride, n, ntiles, tiles + b * stride,
                                hist + b * stride, keys_out + b * out_stride, vals_out + b * out_stride, nullptr,
                                totals + b * stride);
}

struct RadixWs {
  uint32_t* tiles;  // [ntiles << TB]
  int32_t* hist;    // [ntiles][1 << TB]
  int32_t* segsum;  // [ntiles / kRadixSeg][1 << TB] (long lists)
  uint32_t* keys[2];
  uint32_t* vals[2];
};
// one layout for every tile size: ntiles << TB <= n rounded up to the largest tile
static size_t radix_ws_layout(int64_t n, char* base, RadixWs* ws) {
  const size_t padded = (size_t)cdiv(n, 2048) * 2048;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  RadixWs w;
  w.tiles = (uint32_t*)take(sizeof(uint32_t) * padded);
  w.hist = (int32_t*)take(sizeof(int32_t) * padded);
  w.segsum = (int32_t*)take(sizeof(int32_t) * (size_t)cdiv((int64_t)(padded / 2048), kRadixSeg) * 2048);
  for (int i = 0; i < 2; ++i) {
    w.keys[i] = (uint32_t*)take(sizeof(uint32_t) * (size_t)n);
    w.vals[i] = (uint32_t*)take(sizeof(uint32_t) * (size_t)n);
  }
  if (ws) *ws = w;
  return off;
}
template <int TB>
static void launch_radix_sort(const SortSegs& sg, int n, int key_bits, const RadixWs& ws, int32_t* sorted_ids,
                              int32_t* perm, hipStream_t st) {
  constexpr int kTile = 1 << TB, kThreads = kTile / 2;
  const int ntiles = (int)cdiv(n, kTile);
  const int passes = std::max(1, (int)cdiv(key_bits, TB));
  const uint32_t* kin = nullptr;
  const uint32_t* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const bool last = p == passes - 1;
    uint32_t* kout = last ? reinterpret_cast<uint32_t*>(sorted_ids) : ws.keys[p & 1];
    uint32_t* vout = last ? reinterpret_cast<uint32_t*>(perm) : ws.vals[p & 1];
    const int shift = p * TB;
    const bool tall = ntiles > (kRadixMaxN >> TB);  // more tiles than a scatter workgroup should re-reduce itself
    const int32_t* segsum = tall ? ws.segsum : nullptr;
    const dim3 seg_grid((kTile / 2 + kBlock - 1) / kBlock, (ntiles + kRadixSeg - 1) / kRadixSeg);
    if (p == 0) {
      hipLaunchKernelGGL((radix_tile_kernel<TB, true>), dim3(ntiles), dim3(kThreads), 0, st, sg, kin, n, shift, ws.tiles,
                         ws.hist);
      if (tall)
        hipLaunchKernelGGL((radix_segsum_kernel<TB>), seg_grid, dim3(kBlock), 0, st, (const int32_t*)ws.hist, ntiles,
                           ws.segsum);
      hipLaunchKernelGGL((radix_scatter_kernel<TB, true>), dim3(ntiles), dim3(kThreads), 0, st, sg, kin, vin, n, ntiles,
                         (const uint32_t*)ws.tiles, (const int32_t*)ws.hist, kout, vout, segsum);
    } else {
      hipLaunchKernelGGL((radix_tile_kernel<TB, false>), dim3(ntiles), dim3(kThreads), 0, st, sg, kin, n, shift, ws.tiles,
                         ws.hist);
      if (tall)
        hipLaunchKernelGGL((radix_segsum_kernel<TB>), seg_grid, dim3(kBlock), 0, st, (const int32_t*)ws.hist, ntiles,
                           ws.segsum);
      hipLaunchKernelGGL((radix_scatter_kernel<TB, false>), dim3(ntiles), dim3(kThreads), 0, st, sg, kin, vin, n, ntiles,
                         (const uint32_t*)ws.tiles, (const int32_t*)ws.hist, kout, vout, segsum);
    }
    kin = kout;
    vin = vout;
  }
}
// the lists of a batch (equal length n <= kRadixLongN) pass by pass: three launches per pass for all of them.  `base` = nbatch
// per-list workspaces of radix_ws_layout(n) bytes each.
template <int TB>
static void launch_radix_sort_batched(const SortSegsBatch& sb, int nbatch, int n, int key_bits, char* base,
                                      int32_t* sorted_ids, int32_t* perm, hipStream_t st) {
  constexpr int kTile = 1 << TB, kThreads = kTile / 2;
  const int ntiles = (int)cdiv(n, kTile);
  const int passes = std::max(1, (int)cdiv(key_bits, TB));
  const int64_t stride = (int64_t)(radix_ws_layout(n, nullptr, nullptr) / 4);  // words between two lists' workspaces
  RadixWs ws;
  radix_ws_layout(n, base, &ws);
  const uint32_t* kin = nullptr;
  const uint32_t* vin = nullptr;
  int64_t in_stride = 0;
  const dim3 grid(ntiles, nbatch);
  for (int p = 0; p < passes; ++p) {
    const bool last = p == passes - 1;
    uint32_t* kout = last ? reinterpret_cast<uint32_t*>(sorted_ids) : ws.keys[p & 1];
    uint32_t* vout = last ? reinterpret_cast<uint32_t*>(perm) : ws.vals[p & 1];
    const int64_t out_stride = last ? (int64_t)n : stride;
    const int shift = p * TB;
    const bool tall = ntiles > 128;
    const dim3 scan_grid(kTile / (tall ? 16 : 64), nbatch);
    auto colscan = [&]() {
      if (tall)
        hipLaunchKernelGGL((radix_colscan_kernel<TB, 16>), scan_grid, dim3(kBlock), 0, st, ws.hist, ntiles, ws.segsum, stride);
      else
        hipLaunchKernelGGL((radix_colscan_kernel<TB, 64>), scan_grid, dim3(kBlock), 0, st, ws.hist, ntiles, ws.segsum, stride);
    };
    if (p == 0) {
      hipLaunchKernelGGL((radix_tile_batched_kernel<TB, true>), grid, dim3(kThreads), 0, st, sb, kin, in_stride, n, shift,
                         ws.tiles, ws.hist, stride);
      colscan();
      hipLaunchKernelGGL((radix_scatter_batched_kernel<TB, true>), grid, dim3(kThreads), 0, st, sb, kin, vin, in_stride, n,
                         ntiles, (const uint32_t*)ws.tiles, (const int32_t*)ws.hist, (const int32_t*)ws.segsum, stride,
                         kout, vout, out_stride);
    } else {
      hipLaunchKernelGGL((radix_tile_batched_kernel<TB, false>), grid, dim3(kThreads), 0, st, sb, kin, in_stride, n, shift,
                         ws.tiles, ws.hist, stride);
      colscan();
      hipLaunchKernelGGL((radix_scatter_batched_kernel<TB, false>), grid, dim3(kThreads), 0, st, sb, kin, vin, in_stride,
                         n, ntiles, (const uint32_t*)ws.tiles, (const int32_t*)ws.hist, (const int32_t*)ws.segsum, stride,
                         kout, vout, out_stride);
    }
    kin = kout;
    vin = vout;
    in_stride = out_stride;
  }
}

// owner key of every id + the per-owner totals.  The totals are privatised per workgroup in LDS (one global
// atomic per owner per workgroup): with every thread hitting the same few global counters the kernel took
// 1.5 ms for 131072 ids (profiles/r1 sharded GloVe).  Integer atomics: order-independent result.
constexpr int kOwnerHistMax = 1024;
__global__ __launch_bounds__(kBlock) void owner_keys_kernel(const int32_t* __restrict__ ids, int64_t n, int world,
                                                           uint32_t* __restrict__ keys,
                                                           unsigned long long* __restrict__ counts) {
  __shared__ unsigned int hist[kOwnerHistMax];
  const bool priv = world <= kOwnerHistMax;
  if (priv) {
    for (int g = threadIdx.x; g < world; g += kBlock) hist[g] = 0;
    __syncthreads();
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint32_t o = (uint32_t)ids[i] % (uint32_t)world;
    keys[i] = o;
    if (priv) atomicAdd(&hist[o], 1u);
    else atomicAdd(&counts[o], 1ull);
  }
  if (priv) {
    __syncthreads();
    for (int g = threadIdx.x; g < world; g += kBlock)
      if (hist[g]) atomicAdd(&counts[g], (unsigned long long)hist[g]);
  }
}

// Single-launch stable bucket for the sizes of a training step (n <= 32768 ids, world <= 8).  Stable by
// construction.  Replaces a memset + 2 kernels + a 5-launch device radix sort (~60 us of dependent launch latency
// per lookup).
constexpr int kBucketMaxWorld = 8;  // one MI355X node
constexpr int kBucketMaxN = 32768;
constexpr int kBucketThreads = 1024;
constexpr int kBucketRounds = kBucketMaxN / kBucketThreads;  // 32
// Tiled bucket, two launches, for lists of up to kBucketTileMax tiles of 1024 ids (the one-workgroup kernel below took
// 16.5 us for the 16 384 ids of a sharded in-batch step -- the longest kernel of its routing plan).  Thread t of
// workgroup `tile` owns id 1024 tile + t; the stable position of an id is
//   owner base + same-owner ids in earlier tiles + same-owner ids in earlier waves of its tile + same-owner lanes below
// Launch 1 writes the per-(tile, wave, owner) counts and per-(tile, owner) totals; launch 2 turns them into the three
// offsets (every workgroup reduces the tile totals itself: <= 1024 x 8 integers) and scatters.  Both are pure
// integer work with coalesced loads; ranks inside a wave come from ballots, so there is no atomic and the result
// is the stable order by construction.
constexpr int kBucketTileMax = 1024;  // n <= 1 Mi ids
__device__ __forceinline__ int owner_of(uint32_t id, int world, bool pow2) {
  return (int)(pow2 ? id & (uint32_t)(world - 1) : id % (uint32_t)world);
}
__device__ __forceinline__ void bucket_count_body(const SortSegs& ids, int n, int world,
                                                  int* __restrict__ wave_cells,   // [T][16][8]
                                                  int* __restrict__ tile_tot,     // [T][8]
                                                  int tile) {
  constexpr int kWaves = kBucketThreads / 64;
  __shared__ int cnt[kWaves][kBucketMaxWorld];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const bool pow2 = (world & (world - 1)) == 0;
  const int j = tile * kBucketThreads + t;
  const int own = j < n ? owner_of((uint32_t)seg_id(ids, j), world, pow2) : -1;
#pragma unroll
  for (int g = 0; g < kBucketMaxWorld; ++g) {
    const int c = __popcll(__ballot(own == g));
    if (lane == 0) {
      cnt[w][g] = c;
      wave_cells[(tile * kWaves + w) * kBucketMaxWorld + g] = c;
    }
  }
  __syncthreads();
  if (t < kBucketMaxWorld) {
    int tot = 0;
#pragma unroll
    for (int x = 0; x < kWaves; ++x) tot += cnt[x][t];
    tile_tot[tile * kBucketMaxWorld + t] = tot;
  }
}
__global__ __launch_bounds__(kBucketThreads) void bucket_count_kernel(SortSegs ids, int n, int world,
                                                                     int* __restrict__ wave_cells,
                                                                     int* __restrict__ tile_tot) {
  bucket_count_body(ids, n, world, wave_cells, tile_tot, blockIdx.x);
}
// the lists of several coming batches at once (blockIdx.y = the list; cells: ints between the lists' workspaces)
__global__ __launch_bounds__(kBucketThreads) void bucket_count_batched_kernel(SortSegsBatch sb, int n, int world,
                                                                             int* __restrict__ wave_cells,
                                                                             int* __restrict__ tile_tot,
                                                                             int64_t cells) {
  const int y = blockIdx.y;
  bucket_count_body(sb.b[y], n, world, wave_cells + y * cells, tile_tot + y * cells, blockIdx.x);
}
__device__ __forceinline__ void bucket_scatter_body(const SortSegs& ids, int n, int world,
                                                    const int* __restrict__ wave_cells,
                                                    const int* __restrict__ tile_tot,
                                                    int32_t* __restrict__ inverse, int32_t* __restrict__ local_rows,
                                                    int32_t* __restrict__ perm, int64_t* __restrict__ counts,
                                                    int tile, int ntiles) {
  constexpr int kWaves = kBucketThreads / 64;
  __shared__ int red[2][kWaves][kBucketMaxWorld];  // [all tiles | earlier tiles] per wave and owner
  __shared__ int tot_s[kBucketMaxWorld], bef_s[kBucketMaxWorld];
  __shared__ int wcell[kWaves][kBucketMaxWorld], wpre[kWaves][kBucketMaxWorld];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const bool pow2 = (world & (world - 1)) == 0;
  const int j = tile * kBucketThreads + t;
  const uint32_t id = j < n ? (uint32_t)seg_id(ids, j) : 0u;  // in flight while the offsets are worked out
  // thread t holds the totals of tile t (ntiles <= 1024)
  int all[kBucketMaxWorld], before[kBucketMaxWorld];
#pragma unroll
  for (int g = 0; g < kBucketMaxWorld; ++g) {
    const int v = t < ntiles ? tile_tot[t * kBucketMaxWorld + g] : 0;
    all[g] = v;
    before[g] = t < tile ? v : 0;
  }
  if (t < kWaves * kBucketMaxWorld) wcell[t / kBucketMaxWorld][t % kBucketMaxWorld] = wave_cells[tile * kWaves * kBucketMaxWorld + t];
#pragma unroll
  for (int g = 0; g < kBucketMaxWorld; ++g) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      all[g] += __shfl_xor(all[g], o, 64);
      before[g] += __shfl_xor(before[g], o, 64);
    }
    if (lane == 0) { red[0][w][g] = all[g]; red[1][w][g] = before[g]; }
  }
  __syncthreads();
  if (t < kBucketMaxWorld) {  // one thread per owner: totals over all tiles / over the earlier tiles
    int tot = 0, bef = 0;
#pragma unroll
    for (int x = 0; x < kWaves; ++x) { tot += red[0][x][t]; bef += red[1][x][t]; }
    tot_s[t] = tot;
    bef_s[t] = bef;
    if (tile == 0 && t < world) counts[t] = tot;
  } else if (t >= 64 && t < 64 + kWaves * kBucketMaxWorld) {  // same-owner ids in the earlier waves of this tile
    const int ww = (t - 64) / kBucketMaxWorld, g = (t - 64) % kBucketMaxWorld;
    int a = 0;
    for (int x = 0; x < ww; ++x) a += wcell[x][g];
    wpre[ww][g] = a;
  }
  __syncthreads();
  const int own = j < n ? owner_of(id, world, pow2) : -1;
  const unsigned long long below = (1ull << lane) - 1ull;
  int pos = -1;
#pragma unroll
  for (int g = 0; g < kBucketMaxWorld; ++g) {
    const unsigned long long m = __ballot(own == g);
    if (own == g) pos = __popcll(m & below);
  }
  if (pos >= 0) {
    pos += bef_s[own] + wpre[w][own];
#pragma unroll
    for (int g = 0; g < kBucketMaxWorld; ++g)
      if (g < own) pos += tot_s[g];
  }
  if (pos >= 0) {
    perm[pos] = j;
    local_rows[pos] = (int32_t)(pow2 ? id >> __builtin_ctz(world) : id / (uint32_t)world);
    if (inverse) inverse[j] = pos;
  }
}
__global__ __launch_bounds__(kBucketThreads) void bucket_scatter_kernel(SortSegs ids, int n, int world,
                                                                       const int* __restrict__ wave_cells,
                                                                       const int* __restrict__ tile_tot,
                                                                       int32_t* __restrict__ inverse,
                                                                       int32_t* __restrict__ local_rows,
                                                                       int32_t* __restrict__ perm,
                                                                       int64_t* __restrict__ counts) {
  bucket_scatter_body(ids, n, world, wave_cells, tile_tot, inverse, local_rows, perm, counts, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(kBucketThreads) void bucket_scatter_batched_kernel(SortSegsBatch sb, int n, int world,
                                                                               const int* __restrict__ wave_cells,
                                                                               const int* __restrict__ tile_tot,
                                                                               int64_t cells,
                                                                               int32_t* __restrict__ inverse,
                                                                               int32_t* __restrict__ local_rows,
                                                                               int32_t* __restrict__ perm,
                                                                               int64_t* __restrict__ counts) {
  const int y = blockIdx.y;
  bucket_scatter_body(sb.b[y], n, world, wave_cells + y * cells, tile_tot + y * cells,
                      inverse ? inverse + (int64_t)y * n : nullptr, local_rows + (int64_t)y * n, perm + (int64_t)y * n,
                      counts + (int64_t)y * world, blockIdx.x, gridDim.x);
}

// One workgroup, ids taken round by round (round r = ids [1024 r, 1024 r + 1024), thread t = one id): every access
// is coalesced.  Stable position of an id = owner base + ids of the same owner in earlier (round, wave) cells +
// same-owner lanes below it in its own wave (ballot + popcount).  The 32 x 16 cell totals per owner are prefix-
// summed by one thread per owner.  (Earlier versions gave each thread a contiguous slice: either the loads or the
// stores were then 64 distinct lines per wave instruction and the kernel took ~30 us for 16 384 ids.)
__global__ __launch_bounds__(kBucketThreads) void bucket_small_kernel(const int32_t* __restrict__ ids, int n, int world,
                                                                     int32_t* __restrict__ inverse,
                                                                     int32_t* __restrict__ local_rows,
                                                                     int32_t* __restrict__ perm,
                                                                     int64_t* __restrict__ counts) {
  constexpr int kWaves = kBucketThreads / 64;
  __shared__ int cell[kBucketMaxWorld][kBucketRounds * kWaves + 1];  // per owner: totals, then exclusive prefixes
  __shared__ int owner_base[kBucketMaxWorld];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const bool pow2 = (world & (world - 1)) == 0;
  const int rounds = (n + kBucketThreads - 1) / kBucketThreads;
  const unsigned long long below = (1ull << lane) - 1ull;
  // pass 1: per (round, wave) cell and owner, how many ids.  Both passes batch four rounds so that four loads are in
  // flight (a rolled loop serialised 16 global-load latencies: 19 us).  Pass 2 re-reads the ids (L2) and re-derives
  // the in-wave ranks instead of keeping 64 values per thread in registers.
#pragma unroll 1
  for (int r0 = 0; r0 < rounds; r0 += 4) {
    uint32_t idv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = (r0 + u) * kBucketThreads + t;
      idv[u] = (r0 + u < rounds && j < n) ? (uint32_t)ids[j] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + u >= rounds) break;  // uniform
      const int j = (r0 + u) * kBucketThreads + t;
      const int own = j < n ? (int)(pow2 ? idv[u] & (uint32_t)(world - 1) : idv[u] % (uint32_t)world) : -1;
#pragma unroll
      for (int g = 0; g < kBucketMaxWorld; ++g) {
        const unsigned long long m = __ballot(own == g);
        if (lane == 0) cell[g][(r0 + u) * kWaves + w] = __popcll(m);
      }
    }
  }
  __syncthreads();
  if (w < kBucketMaxWorld) {  // one WAVE per owner: exclusive prefix over its (round, wave) cells
    constexpr int kPer = kBucketRounds * kWaves / 64;  // 8 consecutive cells per lane
    int v[kPer], sum = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      v[i] = cell[w][lane * kPer + i];  // cells beyond rounds * kWaves were never written: mask them
      if (lane * kPer + i >= rounds * kWaves) v[i] = 0;
      sum += v[i];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    int run = incl - sum;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      cell[w][lane * kPer + i] = run;
      run += v[i];
    }
    if (lane == 63) cell[w][kBucketRounds * kWaves] = incl;  // owner total
  }
  __syncthreads();
  if (t == 0) {
    int base = 0;
    for (int g = 0; g < kBucketMaxWorld; ++g) {
      owner_base[g] = base;
      const int tot = cell[g][kBucketRounds * kWaves];
      if (g < world) counts[g] = tot;
      base += tot;
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int r0 = 0; r0 < rounds; r0 += 4) {
    uint32_t idv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = (r0 + u) * kBucketThreads + t;
      idv[u] = (r0 + u < rounds && j < n) ? (uint32_t)ids[j] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + u >= rounds) break;  // uniform
      const int r = r0 + u, j = r * kBucketThreads + t;
      const int own = j < n ? (int)(pow2 ? idv[u] & (uint32_t)(world - 1) : idv[u] % (uint32_t)world) : -1;
      int pos = -1;
#pragma unroll
      for (int g = 0; g < kBucketMaxWorld; ++g) {
        const unsigned long long m = __ballot(own == g);
        if (own == g) pos = owner_base[g] + cell[g][r * kWaves + w] + __popcll(m & below);
      }
      if (pos >= 0) {
        perm[pos] = j;
        local_rows[pos] = (int32_t)(pow2 ? idv[u] >> __builtin_ctz(world) : idv[u] / (uint32_t)world);
        if (inverse) inverse[j] = pos;
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void local_rows_kernel(const int32_t* __restrict__ ids,
                                                           const int32_t* __restrict__ perm, int64_t n, int world,
                                                           int32_t* __restrict__ local_rows,
                                                           int32_t* __restrict__ inverse) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
    const int32_t i = perm[k];
    local_rows[k] = (int32_t)((uint32_t)ids[i] / (uint32_t)world);
    if (inverse) inverse[i] = (int32_t)k;
  }
}

// keysT[t][v] = order-preserving unsigned image of scores[v][t] (negative floats: all bits flipped; others: sign bit
// set), the key device-library float sorts use too: ascending keys = ascending floats, -0 before +0, NaNs by their bits
__global__ __launch_bounds__(kBlock) void transpose_cols_keys_kernel(const float* __restrict__ scores, int64_t V, int T,
                                                                    int32_t* __restrict__ keysT) {
  const int64_t total = V * T;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t v = i / T;
    const int t = (int)(i - v * T);
    const uint32_t f = __float_as_uint(scores[i]);
    const uint32_t mask = (uint32_t)((int32_t)f >> 31) | 0x80000000u;  // negative: all ones; else: the sign bit
    keysT[(int64_t)t * V + v] = (int32_t)(f ^ mask);
  }
}
// indices[v][t] = idxT[t][v]
__global__ __launch_bounds__(kBlock) void untranspose_idx_kernel(const int32_t* __restrict__ idxT, int64_t V, int T,
                                                                int32_t* __restrict__ indices) {
  const int64_t total = V * T;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t v = i / T;
    const int t = (int)(i - v * T);
    indices[i] = idxT[(int64_t)t * V + v];
  }
}

// scores[q][n] = queries[q] . candidates[n]; one row group per candidate row, queries staged in LDS.
// q_ids != NULL: query q is row q_ids[q] of `queries` (Glove.score_all gathers its probes from the table).
template <int VEC, int NCH>
__global__ __launch_bounds__(kBlock) void score_rows_kernel(const float* __restrict__ queries,
                                                           const int32_t* __restrict__ q_ids, int nq,
                                                           const float* __restrict__ cand, int64_t N, int D, int G,
                                                           float* __restrict__ out, int64_t out_q_stride,
                                                           int64_t out_n_stride) {
  extern __shared__ __attribute__((aligned(16))) float qs[];  // [nq][D]
  for (int i = threadIdx.x; i < nq * D; i += kBlock) {
    const int q = i / D;
    const int64_t row = q_ids ? (int64_t)q_ids[q] : (int64_t)q;
    qs[i] = queries[row * D + (i - q * D)];
  }
  __syncthreads();
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int nvec = D / VEC;
  for (int64_t r = group; r < N; r += ngroups) {
    RowRegs<VEC, NCH> c;
    row_load(c, cand + r * D, lig, G, nvec);
    for (int q = 0; q < nq; ++q) {
      RowRegs<VEC, NCH> qq;
      row_load(qq, qs + (int64_t)q * D, lig, G, nvec);
      const float d = group_sum(row_dot_partial(c, qq), G);
      if (lig == 0) out[(int64_t)q * out_q_stride + r * out_n_stride] = d;
    }
  }
}

// rows [r0, r0 + nb) of `scores` ([.][pitch] floats, N valid) -> keys[b][N]: images whose ASCENDING unsigned order is the
// floats' DESCENDING order (the complement of transpose_cols_keys_kernel's image); and the k best of sorted rows back
__global__ __launch_bounds__(kBlock) void desc_keys_kernel(const float* __restrict__ scores, int64_t pitch, int64_t N,
                                                          int32_t* __restrict__ keys) {
  const float* row = scores + (int64_t)blockIdx.y * pitch;
  int32_t* out = keys + (int64_t)blockIdx.y * N;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    const uint32_t f = __float_as_uint(row[i]);
    const uint32_t mask = (uint32_t)((int32_t)f >> 31) | 0x80000000u;  // negative: all ones; else: the sign bit
    out[i] = (int32_t)(f ^ ~mask);                                     // == ~(f ^ mask): descending image
  }
}
__global__ __launch_bounds__(kBlock) void take_k_rows_kernel(const float* __restrict__ scores, int64_t pitch,
                                                            const int32_t* __restrict__ idx, int64_t N, int k,
                                                            float* __restrict__ out_s, int32_t* __restrict__ out_i) {
  const float* row = scores + (int64_t)blockIdx.y * pitch;
  const int32_t* ix = idx + (int64_t)blockIdx.y * N;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < k; i += gridDim.x * kBlock) {
    const int32_t j = ix[i];
    out_i[(int64_t)blockIdx.y * k + i] = j;
    out_s[(int64_t)blockIdx.y * k + i] = row[j];
  }
}

// Queries are staged in <= 48 KB of LDS per launch; more queries = more passes over the candidates.
static int launch_score_rows(const float* queries, const int32_t* q_ids, int nq, const float* cand, int64_t N,
                             int D, float* out, int64_t out_q_stride, int64_t out_n_stride, hipStream_t st) {
  const RowGeom g = row_geom(D);
  const int grid = grid_for_groups(N, g.G);
  const int qmax = std::max(1, 12288 / D);
  for (int q0 = 0; q0 < nq; q0 += qmax) {
    const int nqc = std::min(qmax, nq - q0);
    const size_t lds = (size_t)nqc * D * sizeof(float);
    const float* qbase = q_ids ? queries : queries + (int64_t)q0 * D;
    const int32_t* qi = q_ids ? q_ids + q0 : nullptr;
    ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((score_rows_kernel<VEC, NCH>), dim3(grid), dim3(kBlock), lds, st, qbase,
                                           qi, nqc, cand, N, D, g.G, out + (int64_t)q0 * out_q_stride,
                                           out_q_stride, out_n_stride));
  }
  return check_launch("score_rows");
}

// A list of ANY length (below 2^30) through this file's radix sort: up to kRadixLongN ids the single-list form (a scatter
// workgroup re-reduces the histogram matrix, or its segment sums, itself: fewest launches), beyond it the batched form with
// one list -- the histogram matrix is scanned down its columns by its own launch between the two launches of a pass, so
// no workgroup walks a matrix of thousands of tiles.  Workspace: radix_ws_layout(n) bytes.
constexpr int64_t kSortMaxN = (int64_t)1 << 30;
static void radix_any(const SortSegs& sg, int64_t n, int key_bits, char* workspace, int32_t* sorted_ids, int32_t* perm,
                      hipStream_t st) {
  if (n <= kRadixLongN) {
    RadixWs ws;
    radix_ws_layout(n, workspace, &ws);
    launch_radix_sort<11>(sg, (int)n, key_bits, ws, sorted_ids, perm, st);
    return;
  }
  SortSegsBatch sb;
  for (int b = 0; b < kMaxSortBatch; ++b) sb.b[b] = sg;
  launch_radix_sort_batched<11>(sb, 1, (int)n, key_bits, workspace, sorted_ids, perm, st);
}

}  // namespace esr

using namespace esr;

extern "C" {

// ------------------------------------------------------------------------------------------------
size_t esr_segment_sort_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  const size_t tiles = n <= kMidSortMax ? align_up((size_t)(kMidSortMax + kMidSortMax / kSplitEvery) * 4, 256) : 0;
  return std::max(tiles, radix_ws_layout(n, nullptr, nullptr));
}

static int segment_sort_segs(const char* who, const SortSegs& sg, int64_t n, int64_t V, int32_t* sorted_ids,
                             int32_t* perm, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const bool fits32 = V <= ((int64_t)1 << (32 - kMaxTileBits));  // (id << 11 | position) is a 32-bit composite
  if (fits32 && n <= (1 << kMaxTileBits)) {  // one tile: one launch
    if (n <= 512) hipLaunchKernelGGL(tile_sort_single_kernel<9>, dim3(1), dim3(256), 0, st, sg, (int)n, sorted_ids, perm);
    else if (n <= 1024) hipLaunchKernelGGL(tile_sort_single_kernel<10>, dim3(1), dim3(512), 0, st, sg, (int)n, sorted_ids, perm);
    else hipLaunchKernelGGL(tile_sort_single_kernel<11>, dim3(1), dim3(1024), 0, st, sg, (int)n, sorted_ids, perm);
    return check_launch(who);
  }
  if (n <= kSmallSortMax && !fits32) {  // short list of wide ids: 64-bit composites in one workgroup's LDS
    hipLaunchKernelGGL(segment_sort_small_kernel, dim3(1), dim3(kSmallSortThreads), 0, st, sg, (int)n, sorted_ids, perm);
    return check_launch(who);
  }
  if (n <= kMidSortMax && fits32) {
    if ((size_t)(kMidSortMax + kMidSortMax / kSplitEvery) * 4 > workspace_bytes || ((uintptr_t)workspace & 15)) {
      set_error("%s: workspace %zu bytes too small (or misaligned)", who, workspace_bytes);
      return ESR_EWORKSPACE;
    }
    uint32_t* tiles = (uint32_t*)workspace;
    if (n <= 512 * kMidTiles) launch_tile_sort<9>(sg, (int)n, tiles, sorted_ids, perm, st);
    else if (n <= 1024 * kMidTiles) launch_tile_sort<10>(sg, (int)n, tiles, sorted_ids, perm, st);
    else launch_tile_sort<11>(sg, (int)n, tiles, sorted_ids, perm, st);
    return check_launch(who);
  }
  if (n > kSortMaxN) {
    set_error("%s: n=%lld ids exceed 2^30", who, (long long)n);
    return ESR_EINVAL;
  }
  if (radix_ws_layout(n, nullptr, nullptr) > workspace_bytes || ((uintptr_t)workspace & 15)) {
    set_error("%s: workspace %zu bytes too small (or misaligned)", who, workspace_bytes);
    return ESR_EWORKSPACE;
  }
  radix_any(sg, n, bits_for(V), (char*)workspace, sorted_ids, perm, st);
  return check_launch(who);
}

int esr_segment_sort_ids(const int32_t* ids, int64_t n, int64_t V, int32_t* sorted_ids, int32_t* perm,
                         void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  ESR_REQUIRE(n >= 0 && V > 0 && n < ((int64_t)1 << 31), "esr_segment_sort_ids: bad sizes n=%lld V=%lld",
              (long long)n, (long long)V);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(ids && sorted_ids && perm && workspace, "esr_segment_sort_ids: null pointer");
  SortSegs sg;
  sg.n = 1;
  for (int i = 0; i < kMaxSortSegs; ++i) {
    sg.ids[i] = i == 0 ? ids : nullptr;
    sg.offset[i] = 0;
    sg.start[i] = i == 0 ? 0 : n;
  }
  sg.start[kMaxSortSegs] = n;
  return segment_sort_segs("esr_segment_sort_ids", sg, n, V, sorted_ids, perm, workspace, workspace_bytes,
                           as_stream(stream));
}

int esr_segment_sort_ids_multi(const int32_t* const* ids, const int64_t* counts, const int64_t* offsets, int nseg,
                               int64_t V, int32_t* sorted_ids, int32_t* perm, void* workspace, size_t workspace_bytes,
                               esr_stream_t stream) {
  TraceScope trace_scope_("esr_segment_sort_ids_multi");
  ESR_REQUIRE(nseg >= 1 && nseg <= kMaxSortSegs && ids && counts && offsets && V > 0,
              "esr_segment_sort_ids_multi: nseg=%d not in [1, %d] or null argument", nseg, kMaxSortSegs);
  SortSegs sg;
  sg.n = nseg;
  sg.start[0] = 0;
  for (int i = 0; i < kMaxSortSegs; ++i) {
    ESR_REQUIRE(i >= nseg || (counts[i] >= 0 && (counts[i] == 0 || ids[i])), "esr_segment_sort_ids_multi: bad segment %d",
                i);
    sg.ids[i] = i < nseg ? ids[i] : nullptr;
    sg.offset[i] = i < nseg ? offsets[i] : 0;
    sg.start[i + 1] = sg.start[i] + (i < nseg ? counts[i] : 0);
  }
  const int64_t n = sg.start[nseg];
  ESR_REQUIRE(n < ((int64_t)1 << 31), "esr_segment_sort_ids_multi: n=%lld", (long long)n);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(sorted_ids && perm && workspace, "esr_segment_sort_ids_multi: null pointer");
  return segment_sort_segs("esr_segment_sort_ids_multi", sg, n, V, sorted_ids, perm, workspace, workspace_bytes,
                           as_stream(stream));
}

size_t esr_segment_sort_batched_workspace_bytes(int64_t n, int nbatch) {
  if (n <= 0 || nbatch <= 0) return 256;
  const size_t one = esr_segment_sort_workspace_bytes(n);  // the fallback sorts list after list in this much
  const size_t mid = n <= kMidSortMax ? align_up((size_t)nbatch * kMidWsWords * 4, 256) : 0;
  // (any n: lists of wide ids -- V beyond 2^21 -- take the batched radix passes even when they are short)
  const size_t radix = n <= kRadixLongN ? (size_t)nbatch * radix_ws_layout(n, nullptr, nullptr) : 0;  // (longer: list by list)
  return std::max({one, mid, radix});
}

int esr_segment_sort_ids_batched(const int32_t* const* ids, const int64_t* counts, const int64_t* offsets, int nseg,
                                 int nbatch, int64_t V, int32_t* sorted_ids, int32_t* perm, void* workspace,
                                 size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_segment_sort_ids_batched");
  ESR_REQUIRE(nseg >= 1 && nseg <= kMaxSortSegs && nbatch >= 1 && nbatch <= kMaxSortBatch && ids && counts && offsets &&
                  V > 0,
              "esr_segment_sort_ids_batched: nseg=%d not in [1, %d], nbatch=%d not in [1, %d], or null argument", nseg,
              kMaxSortSegs, nbatch, kMaxSortBatch);
  SortSegsBatch sb;
  int64_t n = 0;
  for (int b = 0; b < nbatch; ++b) {
    SortSegs& sg = sb.b[b];
    sg.n = nseg;
    sg.start[0] = 0;
    for (int i = 0; i < kMaxSortSegs; ++i) {
      const int32_t* src = i < nseg ? ids[(size_t)b * nseg + i] : nullptr;
      ESR_REQUIRE(i >= nseg || (counts[i] >= 0 && (counts[i] == 0 || src)),
                  "esr_segment_sort_ids_batched: bad segment %d of list %d", i, b);
      sg.ids[i] = src;
      sg.offset[i] = i < nseg ? offsets[i] : 0;
      sg.start[i + 1] = sg.start[i] + (i < nseg ? counts[i] : 0);
    }
    n = sg.start[nseg];
  }
  for (int b = nbatch; b < kMaxSortBatch; ++b) sb.b[b] = sb.b[0];
  ESR_REQUIRE(n < ((int64_t)1 << 31), "esr_segment_sort_ids_batched: n=%lld", (long long)n);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(sorted_ids && perm && workspace, "esr_segment_sort_ids_batched: null pointer");
  if (workspace_bytes < esr_segment_sort_batched_workspace_bytes(n, nbatch) || ((uintptr_t)workspace & 15)) {
    set_error("esr_segment_sort_ids_batched: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_segment_sort_batched_workspace_bytes(n, nbatch));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const bool fits32 = V <= ((int64_t)1 << (32 - kMaxTileBits));
  if (fits32 && n <= (1 << kMaxTileBits)) {  // every list is one tile: one launch for all of them
    if (n <= 512)
      hipLaunchKernelGGL(tile_sort_single_batched_kernel<9>, dim3(nbatch), dim3(256), 0, st, sb, (int)n, sorted_ids, perm);
    else if (n <= 1024)
      hipLaunchKernelGGL(tile_sort_single_batched_kernel<10>, dim3(nbatch), dim3(512), 0, st, sb, (int)n, sorted_ids, perm);
    else
      hipLaunchKernelGGL(tile_sort_single_batched_kernel<11>, dim3(nbatch), dim3(1024), 0, st, sb, (int)n, sorted_ids, perm);
    return check_launch("esr_segment_sort_ids_batched");
  }
  if (fits32 && n <= kMidSortMax) {  // two launches for all the lists
    uint32_t* tiles = (uint32_t*)workspace;
    if (n <= 512 * kMidTiles) launch_tile_sort_batched<9>(sb, nbatch, (int)n, tiles, sorted_ids, perm, st);
    else if (n <= 1024 * kMidTiles) launch_tile_sort_batched<10>(sb, nbatch, (int)n, tiles, sorted_ids, perm, st);
    else launch_tile_sort_batched<11>(sb, nbatch, (int)n, tiles, sorted_ids, perm, st);
    return check_launch("esr_segment_sort_ids_batched");
  }
  if (n <= kRadixLongN && bits_for(V) <= kRadixMaxPasses * 11 && nbatch > 1) {  // four launches per pass pair for all lists
    launch_radix_sort_batched<11>(sb, nbatch, (int)n, bits_for(V), (char*)workspace, sorted_ids, perm, st);
    return check_launch("esr_segment_sort_ids_batched");
  }
  for (int b = 0; b < nbatch; ++b)  // longer lists: one after the other (stream order: the workspace is reused)
    if (int rc = segment_sort_segs("esr_segment_sort_ids_batched", sb.b[b], n, V, sorted_ids + (int64_t)b * n,
                                   perm + (int64_t)b * n, workspace, workspace_bytes, st))
      return rc;
  return ESR_OK;
}

// ------------------------------------------------------------------------------------------------
int esr_score_all(const float* emb, int64_t V, int D, const int32_t* token, int T, float* scores,
                  esr_stream_t stream) {
  TraceScope trace_scope_("esr_score_all");
  ESR_REQUIRE(V > 0 && D > 0 && T > 0, "esr_score_all: bad sizes V=%lld D=%d T=%d", (long long)V, D, T);
  ESR_REQUIRE(emb && token && scores, "esr_score_all: null pointer");
  const RowGeom g = row_geom(D);
  ESR_REQUIRE(g.nch <= kMaxChunksPerLane, "esr_score_all: D=%d not supported", D);
  // scores is [V, T] row-major: query stride 1, candidate-row stride T.
  return launch_score_rows(emb, token, T, emb, V, D, scores, 1, T, as_stream(stream));
}


// columns of up to kRadixLongN rows go through this file's own radix sort, kMaxSortBatch columns per launch sequence
static size_t argsort_own_layout(int64_t V, int T, size_t* keysT, size_t* idxT, size_t* ksorted, size_t* sort_ws) {
  size_t off = 0;
  *keysT = off;   off += align_up((size_t)V * T * 4, 256);
  *idxT = off;    off += align_up((size_t)V * T * 4, 256);
  *ksorted = off; off += align_up((size_t)V * 4 * kMaxSortBatch, 256);
  *sort_ws = off;
  off += std::max((size_t)std::min(T, kMaxSortBatch) * radix_ws_layout(V, nullptr, nullptr),
                  esr_segment_sort_workspace_bytes(V));
  return off;
}

size_t esr_argsort_columns_workspace_bytes(int64_t V, int T) {
  if (V <= 0 || T <= 0) return 256;
  size_t a, b, c, d;
  return argsort_own_layout(V, T, &a, &b, &c, &d);
}

int esr_argsort_columns(const float* scores, int64_t V, int T, int32_t* indices, void* workspace,
                        size_t workspace_bytes, esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && T > 0 && V <= kSortMaxN, "esr_argsort_columns: bad sizes V=%lld T=%d", (long long)V, T);
  ESR_REQUIRE(scores && indices && workspace, "esr_argsort_columns: null pointer");
  if (workspace_bytes < esr_argsort_columns_workspace_bytes(V, T) || ((uintptr_t)workspace & 15)) {
    set_error("esr_argsort_columns: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_argsort_columns_workspace_bytes(V, T));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  size_t o_keys, o_idx, o_sorted;
  char* base = (char*)workspace;
  {
    // stable ascending sort of every column by the keys' unsigned images: three 11-bit passes of the radix sort above,
    // eight columns per launch sequence (jnp.argsort(scores, axis=0): wikipedia/train_cooccurence.py:95)
    size_t o_ws;
    const size_t total = argsort_own_layout(V, T, &o_keys, &o_idx, &o_sorted, &o_ws);
    int32_t* keysT = (int32_t*)(base + o_keys);     // [T][V]
    int32_t* idxT = (int32_t*)(base + o_idx);       // [T][V]
    int32_t* ksorted = (int32_t*)(base + o_sorted);  // [<= 8][V] (not read)
    const int grid = (int)std::min<int64_t>(kMaxGrid, cdiv(V * T, kBlock));
    hipLaunchKernelGGL(transpose_cols_keys_kernel, dim3(grid), dim3(kBlock), 0, st, scores, V, T, keysT);
    for (int t0 = 0; t0 < T; t0 += kMaxSortBatch) {
      const int nb = std::min(kMaxSortBatch, T - t0);
      SortSegsBatch sb;
      for (int b = 0; b < kMaxSortBatch; ++b) {
        SortSegs& sg = sb.b[b];
        sg.n = 1;
        for (int i = 0; i < kMaxSortSegs; ++i) {
          sg.ids[i] = i == 0 ? keysT + (int64_t)(t0 + std::min(b, nb - 1)) * V : nullptr;
          sg.offset[i] = 0;
          sg.start[i + 1] = V;
        }
        sg.start[0] = 0;
      }
      if (nb > 1) {
        launch_radix_sort_batched<11>(sb, nb, (int)V, 32, base + o_ws, ksorted, idxT + (int64_t)t0 * V, st);
      } else if (int rc = segment_sort_segs("esr_argsort_columns", sb.b[0], V, (int64_t)1 << 32, ksorted,
                                            idxT + (int64_t)t0 * V, base + o_ws, total - o_ws, st)) {
        return rc;
      }
    }
    hipLaunchKernelGGL(untranspose_idx_kernel, dim3(grid), dim3(kBlock), 0, st, (const int32_t*)idxT, V, T, indices);
    return check_launch("esr_argsort_columns");
  }
}

// ------------------------------------------------------------------------------------------------
size_t esr_score_topk_workspace_bytes(int64_t nq, int64_t N, int k) {
  (void)k;
  if (nq <= 0 || N <= 0) return 256;
  const size_t row = align_up((size_t)N * 4, 256);
  if (k > kSelectMaxK)  // own radix sort, kMaxSortBatch rows per launch sequence (see esr_score_topk)
    return row * (size_t)nq + 3 * row * kMaxSortBatch +
           std::max((size_t)kMaxSortBatch * radix_ws_layout(N, nullptr, nullptr), esr_segment_sort_workspace_bytes(N));
  return row * (size_t)nq + 2 * row;
}

int esr_score_topk(const float* queries, const float* candidates, int64_t nq, int64_t N, int D, int k,
                   float* out_scores, int32_t* out_indices, void* workspace, size_t workspace_bytes,
                   esr_stream_t stream) {
  // (the 2^30 bound belongs to the sort: the select path, k <= kSelectMaxK, takes any 32-bit row length)
  ESR_REQUIRE(nq > 0 && N > 0 && D > 0 && k > 0 && k <= N && N < ((int64_t)1 << 31) && (k <= kSelectMaxK || N <= kSortMaxN),
              "esr_score_topk: bad sizes nq=%lld N=%lld D=%d k=%d", (long long)nq, (long long)N, D, k);
  ESR_REQUIRE(queries && candidates && out_scores && out_indices && workspace, "esr_score_topk: null pointer");
  const RowGeom g = row_geom(D);
  ESR_REQUIRE(g.nch <= kMaxChunksPerLane, "esr_score_topk: D=%d not supported", D);
  if (workspace_bytes < esr_score_topk_workspace_bytes(nq, N, k) || ((uintptr_t)workspace & 15)) {
    set_error("esr_score_topk: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_score_topk_workspace_bytes(nq, N, k));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const size_t row = align_up((size_t)N * 4, 256);
  char* base = (char*)workspace;
  float* scores = (float*)base;  // [nq][row/4]
  if (int rc = launch_score_rows(queries, nullptr, (int)nq, candidates, N, D, scores, (int64_t)(row / 4), 1, st))
    return rc;
  // jax.lax.top_k (pinterest/make_recommendations.py:64) asks for the k best only: a radix select per query row (one
  // launch for all queries; only the k survivors are sorted) instead of a full device sort of all N scores per query
  if (k <= kSelectMaxK)
    return select_topk_dense(scores, (int64_t)(row / 4), nq, (int)N, k, out_scores, out_indices, st);
  {
    // k beyond the select's 1024: every row sorted in full, descending and stable (lower index first among equal scores,
    // as jax.lax.top_k), by this file's radix sort over the complemented images, eight rows per launch sequence
    char* p = base + row * nq;
    int32_t* keys = (int32_t*)p;                                   // [8][row / 4]
    int32_t* ksorted = (int32_t*)(p + row * kMaxSortBatch);        // [8][N] (not read)
    int32_t* idx = (int32_t*)(p + 2 * row * kMaxSortBatch);        // [8][N]
    char* sort_ws = p + 3 * row * kMaxSortBatch;
    const size_t sort_ws_bytes = workspace_bytes - (size_t)(sort_ws - base);
    for (int64_t q0 = 0; q0 < nq; q0 += kMaxSortBatch) {
      const int nb = (int)std::min<int64_t>(kMaxSortBatch, nq - q0);
      const dim3 kg((unsigned)std::min<int64_t>(1024, cdiv(N, kBlock)), nb);
      hipLaunchKernelGGL(desc_keys_kernel, kg, dim3(kBlock), 0, st, (const float*)((char*)scores + row * q0),
                         (int64_t)(row / 4), N, keys);
      SortSegsBatch sb;
      for (int b = 0; b < kMaxSortBatch; ++b) {
        SortSegs& sg = sb.b[b];
        sg.n = 1;
        for (int i = 0; i < kMaxSortSegs; ++i) {
          sg.ids[i] = i == 0 ? keys + (int64_t)std::min(b, nb - 1) * N : nullptr;
          sg.offset[i] = 0;
          sg.start[i + 1] = N;
        }
        sg.start[0] = 0;
      }
      if (nb > 1) {
        launch_radix_sort_batched<11>(sb, nb, (int)N, 32, sort_ws, ksorted, idx, st);
      } else if (int rc = segment_sort_segs("esr_score_topk", sb.b[0], N, (int64_t)1 << 32, ksorted, idx, sort_ws,
                                            sort_ws_bytes, st)) {
        return rc;
      }
      const dim3 tg((unsigned)cdiv(k, kBlock), nb);
      hipLaunchKernelGGL(take_k_rows_kernel, tg, dim3(kBlock), 0, st, (const float*)((char*)scores + row * q0),
                         (int64_t)(row / 4), (const int32_t*)idx, N, k, out_scores + q0 * k, out_indices + q0 * k);
    }
    return check_launch("esr_score_topk");
  }
}

// ------------------------------------------------------------------------------------------------
// Workspace: the larger of the tiled path's counts and the radix path's key columns + sort workspace, behind a column
// for the concatenated ids of the segmented entry point when it has to fall back to the radix path.
size_t esr_bucket_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  const size_t tiled = (size_t)cdiv(n, kBucketThreads) * (kBucketThreads / 64 + 1) * kBucketMaxWorld * sizeof(int);
  return align_up((size_t)n * 4, 256) +
         std::max(tiled, align_up((size_t)n * 4, 256) * 2 + radix_ws_layout(n, nullptr, nullptr));
}

// tiled two-launch path; false when it does not apply (then nothing was launched)
static bool bucket_tiled(const SortSegs& sg, int64_t n, int world, int32_t* local_rows, int32_t* perm, int32_t* inverse,
                         int64_t* counts, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const int64_t ntiles = cdiv(n, kBucketThreads);
  const size_t need = (size_t)ntiles * (kBucketThreads / 64 + 1) * kBucketMaxWorld * sizeof(int);
  if (!(n > 2048 && ntiles <= kBucketTileMax && world <= kBucketMaxWorld && workspace && workspace_bytes >= need &&
        ((uintptr_t)workspace & 15) == 0))
    return false;
  int* wave_cells = (int*)workspace;
  int* tile_tot = wave_cells + ntiles * (kBucketThreads / 64) * kBucketMaxWorld;
  hipLaunchKernelGGL(bucket_count_kernel, dim3((int)ntiles), dim3(kBucketThreads), 0, st, sg, (int)n, world, wave_cells,
                     tile_tot);
  hipLaunchKernelGGL(bucket_scatter_kernel, dim3((int)ntiles), dim3(kBucketThreads), 0, st, sg, (int)n, world,
                     (const int*)wave_cells, (const int*)tile_tot, inverse, local_rows, perm, counts);
  return true;
}

// one-workgroup kernel (n <= 32768) or device radix sort; ids is one device array
static int bucket_plain(const char* who, const int32_t* ids, int64_t n, int world, int32_t* local_rows, int32_t* perm,
                        int32_t* inverse, int64_t* counts, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (n > 0 && n <= kBucketMaxN && world <= kBucketMaxWorld) {
    hipLaunchKernelGGL(bucket_small_kernel, dim3(1), dim3(kBucketThreads), 0, st, ids, (int)n, world, inverse,
                       local_rows, perm, counts);
    return check_launch(who);
  }
  if (hipMemsetAsync(counts, 0, sizeof(int64_t) * world, st) != hipSuccess) return check_launch(who);
  if (n == 0) return ESR_OK;
  const size_t col = align_up((size_t)n * 4, 256);
  const size_t need = 2 * col + radix_ws_layout(n, nullptr, nullptr);
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
    set_error("%s: workspace %zu bytes < %zu required (or misaligned)", who, workspace_bytes, need);
    return ESR_EWORKSPACE;
  }
  char* base = (char*)workspace;
  uint32_t* keys = (uint32_t*)base;
  uint32_t* keys_sorted = (uint32_t*)(base + col);
  const int grid = (int)std::min<int64_t>(kMaxGrid, cdiv(n, kBlock));
  hipLaunchKernelGGL(owner_keys_kernel, dim3(grid), dim3(kBlock), 0, st, ids, n, world, keys,
                     reinterpret_cast<unsigned long long*>(counts));
  // stable sort of the owner keys (the permutation is what is wanted): this file's radix sort, one 11-bit pass for up to
  // 2048 ranks
  if (n > kSortMaxN) {
    set_error("%s: n=%lld ids exceed 2^30", who, (long long)n);
    return ESR_EINVAL;
  }
  SortSegs ks = {};
  ks.n = 1;
  ks.ids[0] = reinterpret_cast<const int32_t*>(keys);
  for (int i = 1; i <= kMaxSortSegs; ++i) ks.start[i] = n;
  radix_any(ks, n, bits_for(world), base + 2 * col, reinterpret_cast<int32_t*>(keys_sorted), perm, st);
  hipLaunchKernelGGL(local_rows_kernel, dim3(grid), dim3(kBlock), 0, st, ids, (const int32_t*)perm, n, world,
                     local_rows, inverse);
  return check_launch(who);
}

int esr_bucket_ids_by_owner(const int32_t* ids, int64_t n, int world, int32_t* local_rows, int32_t* perm,
                            int32_t* inverse, int64_t* counts, void* workspace, size_t workspace_bytes,
                            esr_stream_t stream) {
  ESR_REQUIRE(n >= 0 && world > 0 && n < ((int64_t)1 << 31), "esr_bucket_ids_by_owner: bad sizes n=%lld world=%d",
              (long long)n, world);
  ESR_REQUIRE(counts, "esr_bucket_ids_by_owner: null counts");
  ESR_REQUIRE(n == 0 || (ids && local_rows && perm), "esr_bucket_ids_by_owner: null pointer");
  hipStream_t st = as_stream(stream);
  SortSegs sg = {};
  sg.n = 1;
  sg.ids[0] = ids;
  for (int i = 1; i <= kMaxSortSegs; ++i) sg.start[i] = n;
  if (bucket_tiled(sg, n, world, local_rows, perm, inverse, counts, workspace, workspace_bytes, st))
    return check_launch("esr_bucket_ids_by_owner(tiled)");
  return bucket_plain("esr_bucket_ids_by_owner", ids, n, world, local_rows, perm, inverse, counts, workspace,
                      workspace_bytes, st);
}

int esr_bucket_ids_by_owner_multi(const int32_t* const* ids, const int64_t* seg_counts, const int64_t* offsets, int nseg,
                                  int world, int32_t* local_rows, int32_t* perm, int32_t* inverse, int64_t* counts,
                                  void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  ESR_REQUIRE(nseg >= 1 && nseg <= kMaxSortSegs && ids && seg_counts && offsets && world > 0 && counts,
              "esr_bucket_ids_by_owner_multi: nseg=%d not in [1, %d], world=%d or null argument", nseg, kMaxSortSegs,
              world);
  SortSegs sg;
  sg.n = nseg;
  sg.start[0] = 0;
  for (int i = 0; i < kMaxSortSegs; ++i) {
    ESR_REQUIRE(i >= nseg || (seg_counts[i] >= 0 && (seg_counts[i] == 0 || ids[i])),
                "esr_bucket_ids_by_owner_multi: bad segment %d", i);
    sg.ids[i] = i < nseg ? ids[i] : nullptr;
    sg.offset[i] = i < nseg ? offsets[i] : 0;
    sg.start[i + 1] = sg.start[i] + (i < nseg ? seg_counts[i] : 0);
  }
  const int64_t n = sg.start[nseg];
  ESR_REQUIRE(n < ((int64_t)1 << 31), "esr_bucket_ids_by_owner_multi: n=%lld", (long long)n);
  ESR_REQUIRE(n == 0 || (local_rows && perm), "esr_bucket_ids_by_owner_multi: null pointer");
  hipStream_t st = as_stream(stream);
  if (bucket_tiled(sg, n, world, local_rows, perm, inverse, counts, workspace, workspace_bytes, st))
    return check_launch("esr_bucket_ids_by_owner_multi(tiled)");
  // the other paths want one array: materialise the virtual ids at the head of the workspace
  const size_t col = align_up((size_t)std::max<int64_t>(n, 1) * 4, 256);
  if (n > 0 && (!workspace || workspace_bytes < col || ((uintptr_t)workspace & 15))) {
    set_error("esr_bucket_ids_by_owner_multi: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_bucket_workspace_bytes(n));
    return ESR_EWORKSPACE;
  }
  int32_t* vids = (int32_t*)workspace;
  if (n > 0)
    hipLaunchKernelGGL(concat_segs_kernel, dim3((int)std::min<int64_t>(kMaxGrid, cdiv(n, kBlock))), dim3(kBlock), 0, st, sg,
                       n, vids);
  return bucket_plain("esr_bucket_ids_by_owner_multi", vids, n, world, local_rows, perm, inverse, counts,
                      n > 0 ? (char*)workspace + col : nullptr, n > 0 ? workspace_bytes - col : 0, st);
}

size_t esr_bucket_batched_workspace_bytes(int64_t n, int nbatch) {
  if (n <= 0 || nbatch <= 0) return 256;
  const size_t tiled = (size_t)cdiv(n, kBucketThreads) * (kBucketThreads / 64 + 1) * kBucketMaxWorld * sizeof(int);
  return std::max(esr_bucket_workspace_bytes(n), align_up((size_t)nbatch * tiled, 256));
}

int esr_bucket_ids_by_owner_batched(const int32_t* const* ids, const int64_t* seg_counts, const int64_t* offsets,
                                    int nseg, int nbatch, int world, int32_t* local_rows, int32_t* perm,
                                    int32_t* inverse, int64_t* counts, void* workspace, size_t workspace_bytes,
                                    esr_stream_t stream) {
  ESR_REQUIRE(nseg >= 1 && nseg <= kMaxSortSegs && nbatch >= 1 && nbatch <= kMaxSortBatch && ids && seg_counts &&
                  offsets && world > 0 && counts,
              "esr_bucket_ids_by_owner_batched: nseg=%d not in [1, %d], nbatch=%d not in [1, %d], world=%d or null "
              "argument", nseg, kMaxSortSegs, nbatch, kMaxSortBatch, world);
  SortSegsBatch sb;
  int64_t n = 0;
  for (int b = 0; b < nbatch; ++b) {
    SortSegs& sg = sb.b[b];
    sg.n = nseg;
    sg.start[0] = 0;
    for (int i = 0; i < kMaxSortSegs; ++i) {
      const int32_t* src = i < nseg ? ids[(size_t)b * nseg + i] : nullptr;
      ESR_REQUIRE(i >= nseg || (seg_counts[i] >= 0 && (seg_counts[i] == 0 || src)),
                  "esr_bucket_ids_by_owner_batched: bad segment %d of list %d", i, b);
      sg.ids[i] = src;
      sg.offset[i] = i < nseg ? offsets[i] : 0;
      sg.start[i + 1] = sg.start[i] + (i < nseg ? seg_counts[i] : 0);
    }
    n = sg.start[nseg];
  }
  for (int b = nbatch; b < kMaxSortBatch; ++b) sb.b[b] = sb.b[0];
  ESR_REQUIRE(n < ((int64_t)1 << 31), "esr_bucket_ids_by_owner_batched: n=%lld", (long long)n);
  ESR_REQUIRE(n == 0 || (local_rows && perm), "esr_bucket_ids_by_owner_batched: null pointer");
  if (workspace_bytes < esr_bucket_batched_workspace_bytes(n, nbatch) || (n > 0 && (!workspace || ((uintptr_t)workspace & 15)))) {
    set_error("esr_bucket_ids_by_owner_batched: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_bucket_batched_workspace_bytes(n, nbatch));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int64_t ntiles = cdiv(n, kBucketThreads);
  if (n > 2048 && ntiles <= kBucketTileMax && world <= kBucketMaxWorld) {  // two launches for all the lists
    const int64_t cells = ntiles * (kBucketThreads / 64 + 1) * kBucketMaxWorld;
    int* wave_cells = (int*)workspace;
    int* tile_tot = wave_cells + ntiles * (kBucketThreads / 64) * kBucketMaxWorld;
    hipLaunchKernelGGL(bucket_count_batched_kernel, dim3((int)ntiles, nbatch), dim3(kBucketThreads), 0, st, sb, (int)n,
                       world, wave_cells, tile_tot, cells);
    hipLaunchKernelGGL(bucket_scatter_batched_kernel, dim3((int)ntiles, nbatch), dim3(kBucketThreads), 0, st, sb, (int)n,
                       world, (const int*)wave_cells, (const int*)tile_tot, cells, inverse, local_rows, perm, counts);
    return check_launch("esr_bucket_ids_by_owner_batched");
  }
  for (int b = 0; b < nbatch; ++b) {  // short or very long lists: one after the other (the workspace is reused)
    const int32_t* segs[kMaxSortSegs];
    for (int i = 0; i < nseg; ++i) segs[i] = ids[(size_t)b * nseg + i];
    if (int rc = esr_bucket_ids_by_owner_multi(segs, seg_counts, offsets, nseg, world, local_rows + (int64_t)b * n,
                                               perm + (int64_t)b * n, inverse ? inverse + (int64_t)b * n : nullptr,
                                               counts + (int64_t)b * world, workspace, workspace_bytes, stream))
      return rc;
  }
  return ESR_OK;
}


}  // extern "C"
