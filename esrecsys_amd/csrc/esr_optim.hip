// Optimizer kernels.
//
// Production path (north_star): deterministic row-sparse update.  Occurrence ids have been
// stably sorted (esr_segment_sort_ids), so the occurrences of one row are a contiguous run of
// `sorted_ids`.  One row group (G lanes) is launched per sorted position; the group at the head
// of a run sums the run's gradient rows left to right (== occurrence order, so for a run inside
// one 32-position chunk the result equals a sequential scatter-add bit for bit), then does the
// read-modify-write of the parameter and accumulator rows exactly once.  Runs that cross a chunk
// boundary (hot rows of skewed batches) are summed chunk-wise in parallel and combined in a fixed
// order by a second kernel.  No float atomics, no V x D gradient, no host sync; the result never
// depends on scheduling.  The gradient-row buffer doubles as scratch for the chunk partials.
// HBM traffic per distinct row: D*4 (grad) + 2*D*s (param RMW) + 2*D*4 (accumulator RMW).
//
// Reference-faithful path: esr_rows_to_dense rebuilds the dense V x D gradient that JAX's autodiff
// produces for nn.Embed (wikipedia/train_cooccurence.py:86-87) and esr_dense_adam applies
// optax.adam to every element (train_cooccurence.py:99-101,171) [upstream optax 0.1.2].
#include "esr_common.h"
#include "esr_inbatch_mfma.h"

#include <algorithm>

namespace esr {

enum SegOp { kAdagrad = 0, kSgd = 1, kToDense = 2, kMomentum = 3, kMomentumStep = 4, kMomentumStepLazy = 5 };

template <int VEC, int NCH>
__device__ __forceinline__ void param_load(RowRegs<VEC, NCH>& r, const void* table, int dtype, int64_t row, int D,
                                           int lig, int G, int nvec) {
  if (dtype == ESR_F32) {
    row_load(r, (const float*)table + row * D, lig, G, nvec);
  } else {
    const uint16_t* p = (const uint16_t*)table + row * D;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = lig + k * G;
      if (c < nvec) {
        if constexpr (VEC == 4) {
          const uint2 u = *reinterpret_cast<const uint2*>(p + 4 * c);
          r.v[k][0] = __uint_as_float(u.x << 16);
          r.v[k][1] = __uint_as_float(u.x & 0xffff0000u);
          r.v[k][2] = __uint_as_float(u.y << 16);
          r.v[k][3] = __uint_as_float(u.y & 0xffff0000u);
        } else {
          r.v[k][0] = bf16_to_f32(p[c]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) r.v[k][e] = 0.f;
      }
    }
  }
}

template <int VEC, int NCH>
__device__ __forceinline__ void param_store(const RowRegs<VEC, NCH>& r, void* table, int dtype, int64_t row, int D,
                                            int lig, int G, int nvec) {
  if (dtype == ESR_F32) {
    row_store(r, (float*)table + row * D, lig, G, nvec);
  } else {
    uint16_t* p = (uint16_t*)table + row * D;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = lig + k * G;
      if (c < nvec) {
        if constexpr (VEC == 4) {
          uint2 u;
          u.x = (uint32_t)f32_to_bf16(r.v[k][0]) | ((uint32_t)f32_to_bf16(r.v[k][1]) << 16);
          u.y = (uint32_t)f32_to_bf16(r.v[k][2]) | ((uint32_t)f32_to_bf16(r.v[k][3]) << 16);
          *reinterpret_cast<uint2*>(p + 4 * c) = u;
        } else {
          p[c] = f32_to_bf16(r.v[k][0]);
        }
      }
    }
  }
}

// Several tables may be updated by ONE sorted occurrence list: occurrence ids are then "virtual rows"
// vid = row_offset[t] + id of a concatenation of up to kMaxFusedTables tables with the same D (the two towers of one
// step), so a step needs one sort and one update instead of one per table -- at the reference's batch sizes the step
// is bound by the number of dependent launches, not by bytes.  A single table is the n = 1 case.
constexpr int kMaxFusedTables = 4;
struct FusedTables {
  void* table[kMaxFusedTables];
  float* accum[kMaxFusedTables];
  int64_t row_offset[kMaxFusedTables + 1];
  int n;
};

// the read-modify-write of one table row with the summed gradient g of its occurrences
template <int VEC, int NCH, int OP>
__device__ __forceinline__ void seg_apply(const FusedTables& ft, int dtype, int32_t vid, const RowRegs<VEC, NCH>& g,
                                          int D, int lig, int G, int nvec, float lr, float eps) {
  // select chain with constant indices (a runtime index into the kernarg struct would go through scratch)
  void* table = ft.table[0];
  float* accum = ft.accum[0];
  int64_t base = ft.row_offset[0];
#pragma unroll
  for (int k = 1; k < kMaxFusedTables; ++k)
    if (k < ft.n && (int64_t)vid >= ft.row_offset[k]) {
      table = ft.table[k];
      accum = ft.accum[k];
      base = ft.row_offset[k];
    }
  const int64_t id = (int64_t)vid - base;
  if (OP == kToDense) {
    row_store(g, (float*)table + id * D, lig, G, nvec);
  } else if (OP == kMomentum) {
    // the gradient half of optax.sgd(lr, momentum): trace += g ; p -= lr * g  (the decay half is dense)
    RowRegs<VEC, NCH> w, a;
    param_load(w, table, dtype, id, D, lig, G, nvec);
    row_load(a, accum + id * D, lig, G, nvec);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        a.v[k][e] += g.v[k][e];
        w.v[k][e] -= lr * g.v[k][e];
      }
    row_store(a, accum + id * D, lig, G, nvec);
    param_store(w, table, dtype, id, D, lig, G, nvec);
  } else if (OP == kMomentumStep) {
    // one WHOLE step of optax.sgd(lr, momentum) on a touched row (lazy mode: no dense decay pass runs; the row was
    // brought up to the previous step by momentum_catchup_kernel): trace = g + momentum * trace ; p -= lr * trace, in
    // optax's own order and with explicit roundings.  `eps` carries the momentum.
    RowRegs<VEC, NCH> w, a;
    param_load(w, table, dtype, id, D, lig, G, nvec);
    row_load(a, accum + id * D, lig, G, nvec);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        a.v[k][e] = __fadd_rn(g.v[k][e], __fmul_rn(eps, a.v[k][e]));
        w.v[k][e] = __fsub_rn(w.v[k][e], __fmul_rn(lr, a.v[k][e]));
      }
    row_store(a, accum + id * D, lig, G, nvec);
    param_store(w, table, dtype, id, D, lig, G, nvec);
  } else if (OP == kMomentumStepLazy) {
    // kMomentumStep on a row that is still `now - 1 - last[row]` steps behind (lazy optax.sgd(lr, momentum), see decay_steps
    // in esr_common.h): the catch-up first -- the very operations momentum_catchup_kernel applies -- then the step, and the
    // row is marked current.  Saves the catch-up launch in front of the Spotify step (three dependent memory round trips: 12
    // of its 60 us).  At most two tables; their `last` arrays ride in the unused table slots 2 and 3, `now` in the unused
    // last row offset (sparse_momentum_step_lazy2 below).
    int32_t* last = (int32_t*)ft.table[2];
    if (1 < ft.n && (int64_t)vid >= ft.row_offset[1]) last = (int32_t*)ft.table[3];
    const int now = (int)ft.row_offset[kMaxFusedTables];
    const int steps = now - 1 - last[id];
    RowRegs<VEC, NCH> w, a;
    param_load(w, table, dtype, id, D, lig, G, nvec);
    row_load(a, accum + id * D, lig, G, nvec);
    const DecayCoef dk = decay_coef(steps, eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if (steps > 0) decay_apply(w.v[k][e], a.v[k][e], dk, lr, eps);
        a.v[k][e] = __fadd_rn(g.v[k][e], __fmul_rn(eps, a.v[k][e]));
        w.v[k][e] = __fsub_rn(w.v[k][e], __fmul_rn(lr, a.v[k][e]));
      }
    row_store(a, accum + id * D, lig, G, nvec);
    param_store(w, table, dtype, id, D, lig, G, nvec);
    if (lig == 0) last[id] = now;
  } else if (OP == kSgd) {
    RowRegs<VEC, NCH> w;
    param_load(w, table, dtype, id, D, lig, G, nvec);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) w.v[k][e] -= lr * g.v[k][e];
    param_store(w, table, dtype, id, D, lig, G, nvec);
  } else {
    // optax.adagrad [upstream]: acc += g^2 ; p -= lr * g * rsqrt(acc + eps)  (0 where acc == 0)
    RowRegs<VEC, NCH> w, a;
    param_load(w, table, dtype, id, D, lig, G, nvec);
    row_load(a, accum + id * D, lig, G, nvec);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) adagrad_elem(w.v[k][e], a.v[k][e], g.v[k][e], lr, eps);
    row_store(a, accum + id * D, lig, G, nvec);
    param_store(w, table, dtype, id, D, lig, G, nvec);
  }
}

// Kernel 1.  A run of equal ids is cut into chunks only if it is long: the head chunk extends from the head of the run
// to the first multiple of kSegChunk that is at least kSegChunk positions later (32 - 63 occurrences); further
// chunks start at every following multiple of kSegChunk.  A run that fits its head chunk -- every run of a batch
// with few duplicate ids -- is summed left to right by the head's group and applied at once, exactly as a sequential
// scatter-add would, whatever its alignment.  A longer run -- hot rows of a Zipf-distributed batch: the most
// popular of 10^6 rows draws ~1 200 of 16 384 ids -- leaves one partial sum per chunk in the gradient buffer itself
// (in the row of the chunk's first occurrence, which nobody else reads) for kernel 2.  One group walking such a run
// alone took 0.23 - 5.8 ms per step.
constexpr int kSegChunk = 32;
template <int VEC, int NCH, int OP>
__global__ __launch_bounds__(kBlock) void segment_update_kernel(FusedTables ft, int dtype, int D, int G,
                                                               const int32_t* __restrict__ sorted_ids,
                                                               const int32_t* __restrict__ perm, int64_t n,
                                                               float* __restrict__ grad_rows, float lr, float eps) {
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int nvec = D / VEC;
  // Every group walks a CONTIGUOUS slice of positions.  With a grid-stride walk (stride = a multiple of kSegChunk)
  // all continuation chunks -- positions that are multiples of kSegChunk -- fell to 1/32 of the groups, which then
  // queued ~8 chunk sums each while the rest idled: 338 us instead of 113 us for 131 072 Zipf ids.
  const int64_t per = (n + ngroups - 1) / ngroups;
  const int64_t p_end = min(n, (group + 1) * per);
  for (int64_t p = group * per; p < p_end; ++p) {
    const int32_t id = sorted_ids[p];
    const bool head = p == 0 || sorted_ids[p - 1] != id;
    // a continuation chunk starts at a multiple of kSegChunk whose whole previous block belongs to the run
    if (!head && ((p & (kSegChunk - 1)) != 0 || sorted_ids[p - kSegChunk] != id)) continue;
    const int64_t stop = min(head ? ((p + 2 * kSegChunk - 1) / kSegChunk) * kSegChunk : p + kSegChunk, n);
    RowRegs<VEC, NCH> g;
    row_load(g, grad_rows + (int64_t)perm[p] * D, lig, G, nvec);
    int64_t q = p + 1, e_run = p + 1;
    bool run_over = false;  // the walk met a different id: the run ends inside this chunk (no reload further down)
    while (e_run < stop) {  // end of this chunk (ids are contiguous: one line)
      if (sorted_ids[e_run] != id) {
        run_over = true;
        break;
      }
      ++e_run;
    }
    // rows are added strictly left to right, but four loads are kept in flight: a 32-row chunk walked one dependent
    // load at a time made this kernel 2.5x slower on Zipf ids than on uniform ones
    for (; q + 4 <= e_run; q += 4) {
      RowRegs<VEC, NCH> t0, t1, t2, t3;
      row_load(t0, grad_rows + (int64_t)perm[q] * D, lig, G, nvec);
      row_load(t1, grad_rows + (int64_t)perm[q + 1] * D, lig, G, nvec);
      row_load(t2, grad_rows + (int64_t)perm[q + 2] * D, lig, G, nvec);
      row_load(t3, grad_rows + (int64_t)perm[q + 3] * D, lig, G, nvec);
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) g.v[k][e] = (((g.v[k][e] + t0.v[k][e]) + t1.v[k][e]) + t2.v[k][e]) + t3.v[k][e];
    }
    for (; q < e_run; ++q) {
      RowRegs<VEC, NCH> t;
      row_load(t, grad_rows + (int64_t)perm[q] * D, lig, G, nvec);
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) g.v[k][e] += t.v[k][e];
    }
    const bool ends = run_over || q == n || sorted_ids[q] != id;
    if (head && ends)
      seg_apply<VEC, NCH, OP>(ft, dtype, id, g, D, lig, G, nvec, lr, eps);
    else
      row_store(g, grad_rows + (int64_t)perm[p] * D, lig, G, nvec);  // partial sum of a long run
  }
}

// Kernel 2: the long runs.  Their first continuation chunk is found by screening the chunk boundaries (three loads
// per boundary; in a batch without hot rows that is all this kernel does).  A workgroup then combines one run at a
// time: its row groups sum the run's chunk partials round-robin, the group sums are added in group order through
// LDS, and the row is updated once.  Fixed association: the result does not depend on scheduling.
template <int VEC, int NCH, int OP>
__global__ __launch_bounds__(kBlock) void segment_long_kernel(FusedTables ft, int dtype, int D, int G,
                                                             const int32_t* __restrict__ sorted_ids,
                                                             const int32_t* __restrict__ perm, int64_t n,
                                                             const float* __restrict__ grad_rows, float lr,
                                                             float eps) {
  __shared__ float red[kBlock * VEC * NCH];  // [groups][lanes][NCH][VEC]
  constexpr int kPass = 4;                   // chunk boundaries screened per workgroup pass (long runs are then
                                             // spread over many workgroups instead of queueing in a few)
  __shared__ long long s_long[kPass];        // chunk boundaries at which a long run leaves its first chunk
  __shared__ int s_nlong, s_hoff;
  const int tid = threadIdx.x, lig = tid & (G - 1), gidx = tid / G, NG = kBlock / G;
  const int nvec = D / VEC;
  const int64_t nbound = (n - 1) / kSegChunk;  // boundaries kSegChunk, 2 kSegChunk, ... < n
  for (int64_t b0 = (int64_t)blockIdx.x * kPass; b0 < nbound; b0 += (int64_t)gridDim.x * kPass) {
    __syncthreads();  // red / s_* of the previous pass are no longer read
    if (tid == 0) s_nlong = 0;
    __syncthreads();
    // screening, one thread per chunk boundary B: B is the FIRST continuation chunk of a long run iff the whole
    // block before it belongs to the run (ids at B - 32 and B equal) and the block before that does not
    {
      const int64_t B = (b0 + tid + 1) * kSegChunk;
      if (tid < kPass && b0 + tid < nbound) {
        const int32_t id_b = sorted_ids[B];
        const bool first = B < 2 * kSegChunk || sorted_ids[B - 2 * kSegChunk] != id_b;
        if (sorted_ids[B - kSegChunk] == id_b && first) s_long[atomicAdd(&s_nlong, 1)] = B;
      }
    }
    __syncthreads();
    const int nlong = s_nlong;
    for (int li = 0; li < nlong; ++li) {  // rare: the whole workgroup combines one long run at a time
      // (the order of the list is arbitrary; it affects no result: every run is combined independently)
      const int64_t nxt = s_long[li];
      const int32_t id = sorted_ids[nxt];
      const int64_t win = max<int64_t>(nxt - 2 * kSegChunk + 1, 0);  // the head lies in [nxt - 63, nxt - 32]
      if (tid < 64) {
        const int64_t pos = win + tid;
        const bool is_head = pos <= nxt - kSegChunk && sorted_ids[pos] == id && (pos == 0 || sorted_ids[pos - 1] != id);
        const unsigned long long m = __ballot(is_head);
        if (tid == 0) s_hoff = __ffsll((long long)m) - 1;
      }
      __syncthreads();
      const int64_t h = win + s_hoff;
      // K = number of continuation chunks (chunk starts nxt, nxt + 32, ... that still carry this id)
      int64_t K = 0;
      for (int64_t k0 = 0;; k0 += kBlock) {
        const int64_t pos = nxt + (k0 + tid) * kSegChunk;
        const int cnt = __syncthreads_count(pos < n && sorted_ids[pos] == id);
        K += cnt;
        if (cnt < kBlock) break;
      }
      // partial 0 sits at perm[h]; partial i >= 1 at perm[nxt + (i - 1) * kSegChunk]
      RowRegs<VEC, NCH> acc;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc.v[k][e] = 0.f;
      auto part_row = [&](int64_t i) { return (int64_t)perm[i == 0 ? h : nxt + (i - 1) * kSegChunk] * D; };
      int64_t i = gidx;
      for (; i + 3 * NG <= K; i += 4 * NG) {  // four partials in flight, added in order
        RowRegs<VEC, NCH> t0, t1, t2, t3;
        row_load(t0, grad_rows + part_row(i), lig, G, nvec);
        row_load(t1, grad_rows + part_row(i + NG), lig, G, nvec);
        row_load(t2, grad_rows + part_row(i + 2 * NG), lig, G, nvec);
        row_load(t3, grad_rows + part_row(i + 3 * NG), lig, G, nvec);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            acc.v[k][e] = (((acc.v[k][e] + t0.v[k][e]) + t1.v[k][e]) + t2.v[k][e]) + t3.v[k][e];
      }
      for (; i <= K; i += NG) {
        RowRegs<VEC, NCH> t;
        row_load(t, grad_rows + part_row(i), lig, G, nvec);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc.v[k][e] += t.v[k][e];
      }
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[((gidx * G + lig) * NCH + k) * VEC + e] = acc.v[k][e];
      __syncthreads();
      if (gidx == 0) {
        const int used = (int)min<int64_t>(NG, K + 1);
        for (int gg = 1; gg < used; ++gg)
#pragma unroll
          for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc.v[k][e] += red[((gg * G + lig) * NCH + k) * VEC + e];
        seg_apply<VEC, NCH, OP>(ft, dtype, id, acc, D, lig, G, nvec, lr, eps);
      }
      __syncthreads();  // red is rewritten by the next long run
    }
  }
}

// out[i, :] = tables[t][vid - row_offset[t], :] for virtual rows of up to kMaxFusedTables same-width tables
// (the owner side of a fused two-tower lookup).  16-byte chunks, one row group per output row.
__global__ __launch_bounds__(kBlock) void gather_rows_multi_kernel(FusedTables ft, int nchunk, int G,
                                                                  const int32_t* __restrict__ vids, int64_t n,
                                                                  uint4* __restrict__ out) {
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  for (int64_t r = group; r < n; r += ngroups) {
    const int64_t vid = vids[r];
    const uint4* table = reinterpret_cast<const uint4*>(ft.table[0]);
    int64_t base = ft.row_offset[0];
#pragma unroll
    for (int k = 1; k < kMaxFusedTables; ++k)
      if (k < ft.n && vid >= ft.row_offset[k]) {
        table = reinterpret_cast<const uint4*>(ft.table[k]);
        base = ft.row_offset[k];
      }
    const uint4* src = table + (vid - base) * nchunk;
    for (int c = lig; c < nchunk; c += G) out[r * nchunk + c] = src[c];
  }
}

struct IdSegments {
  const int32_t* ids[kMaxFusedTables];
  int64_t start[kMaxFusedTables + 1];  // output position of each segment
  int64_t offset[kMaxFusedTables];     // added to every id of the segment
  int n;
};
__global__ __launch_bounds__(kBlock) void concat_offset_ids_kernel(IdSegments sg, int32_t* __restrict__ out) {
  const int64_t total = sg.start[sg.n];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int32_t* src = sg.ids[0];
    int64_t start = 0, off = sg.offset[0];
#pragma unroll
    for (int k = 1; k < kMaxFusedTables; ++k)
      if (k < sg.n && i >= sg.start[k]) {
        src = sg.ids[k];
        start = sg.start[k];
        off = sg.offset[k];
      }
    out[i] = (int32_t)((int64_t)src[i - start] + off);
  }
}

template <int OP>
static int launch_segment_tables(const char* who, const FusedTables& ft, int dtype, int D, const int32_t* sorted_ids,
                                 const int32_t* perm, int64_t n, float* grad_rows, float lr, float eps,
                                 hipStream_t st, bool skip_long = false) {
  const RowGeom g = row_geom(D);
  if (g.nch > kMaxChunksPerLane) {
    set_error("%s: D=%d not supported", who, D);
    return ESR_EINVAL;
  }
  const int grid = grid_for_groups(n, g.G);
  const int grid2 = (int)std::min<int64_t>(kMaxGrid, cdiv(cdiv(n, kSegChunk), 4));  // 4 chunk boundaries per pass
  ESR_DISPATCH_ROW(g, {
    ESR_KT("segment_update_kernel", st, hipLaunchKernelGGL((segment_update_kernel<VEC, NCH, OP>), dim3(grid), dim3(kBlock), 0, st, ft, dtype, D, g.G,
                       sorted_ids, perm, n, grad_rows, lr, eps));
    if (n > kSegChunk && !skip_long)
      ESR_KT("segment_long_kernel", st, hipLaunchKernelGGL((segment_long_kernel<VEC, NCH, OP>), dim3(grid2), dim3(kBlock), 0, st, ft, dtype, D, g.G,
                         sorted_ids, perm, n, (const float*)grad_rows, lr, eps));
  });
  return check_launch(who);
}

template <int OP>
static int launch_segment_update(const char* who, void* table, int dtype, float* accum, int D,
                                 const int32_t* sorted_ids, const int32_t* perm, int64_t n, float* grad_rows,
                                 float lr, float eps, hipStream_t st) {
  FusedTables ft;
  ft.n = 1;
  for (int i = 0; i < kMaxFusedTables; ++i) {
    ft.table[i] = i == 0 ? table : nullptr;
    ft.accum[i] = i == 0 ? accum : nullptr;
    ft.row_offset[i] = 0;
  }
  ft.row_offset[kMaxFusedTables] = 0;
  // the gradient rows double as scratch for the partial sums of long runs (see segment_update_kernel)
  return launch_segment_tables<OP>(who, ft, dtype, D, sorted_ids, perm, n, grad_rows, lr, eps, st);
}

// optax.adam, elementwise over the whole table.
__global__ __launch_bounds__(kBlock) void dense_adam_kernel(float* __restrict__ p, float* __restrict__ mu,
                                                           float* __restrict__ nu, const float* __restrict__ g,
                                                           int64_t n4, int64_t numel, float lr, float b1, float b2,
                                                           float eps, float inv_bc1, float inv_bc2) {
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  auto upd = [&](float& pv, float& m, float& v, float gv) {
    m = b1 * m + omb1 * gv;
    v = b2 * v + omb2 * gv * gv;
    pv -= lr * (m * inv_bc1) / (sqrtf(v * inv_bc2) + eps);
  };
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 m = reinterpret_cast<float4*>(mu)[i];
    float4 v = reinterpret_cast<float4*>(nu)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    upd(pv.x, m.x, v.x, gv.x);
    upd(pv.y, m.y, v.y, gv.y);
    upd(pv.z, m.z, v.z, gv.z);
    upd(pv.w, m.w, v.w, gv.w);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(mu)[i] = m;
    reinterpret_cast<float4*>(nu)[i] = v;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
    float pv = p[i], m = mu[i], v = nu[i];
    upd(pv, m, v, g[i]);
    p[i] = pv;
    mu[i] = m;
    nu[i] = v;
  }
}

// the decay half of optax.sgd(lr, momentum) over a whole table: trace *= momentum ; p -= lr * trace
__global__ __launch_bounds__(kBlock) void momentum_decay_kernel(float* __restrict__ p, float* __restrict__ tr,
                                                               int64_t n4, int64_t numel, float lr, float momentum) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 t = reinterpret_cast<float4*>(tr)[i];
    t.x *= momentum; t.y *= momentum; t.z *= momentum; t.w *= momentum;
    pv.x -= lr * t.x; pv.y -= lr * t.y; pv.z -= lr * t.z; pv.w -= lr * t.w;
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(tr)[i] = t;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
    const float t = tr[i] * momentum;
    tr[i] = t;
    p[i] -= lr * t;
  }
}

// ---- lazy momentum (Spotify step, spotify/train_spotify.py:238-241) ----------------------------------------------------
// optax.sgd(lr, momentum) moves EVERY element every step (a row without a gradient keeps coasting on its trace), which
// made the decay half a dense pass over both tables: 80 % of the step's bytes.  A row that gets no gradient for n steps
// only undergoes  trace *= momentum ; p -= lr * trace  n times -- a function of n alone -- so it can be applied when the
// row is next READ: last[row] = the step the row is up to date with.  n <= kLazyExact steps are applied one by one (the
// very operations of the dense pass: bit-identical to it -- rows that are read again soon, the hot part of a playlist
// stream); longer gaps by the closed form  trace *= m^n ; p -= lr * trace0 * m (1 - m^n) / (1 - m)  (1e-7-close: one
// rounding instead of n; a walk of thousands of dependent steps per element made the catch-up launch 10 us).
// (decay_coef / decay_apply / kLazyExact: esr_common.h -- esr_spotify.hip reads rows through the same catch-up)

struct CatchupTables {  // up to two same-width tables caught up by one launch (blockIdx.y = the table)
  float* table[2];
  float* trace[2];
  int32_t* last[2];
  const int32_t* ids[2];
  int modulus[2];
};

// bring the rows ids[i] % modulus (modulus 0: ids[i]) up to step `now - 1` and mark them as handled for step `now`: the
// group whose exchange on last[row] returns an older step owns the row, every other occurrence of it skips
template <int VEC, int NCH>
__global__ __launch_bounds__(kBlock) void momentum_catchup_kernel(CatchupTables ct, int D, int G, int64_t n, int now,
                                                                 float lr, float m) {
  const int y = blockIdx.y;
  float* __restrict__ table = y ? ct.table[1] : ct.table[0];
  float* __restrict__ trace = y ? ct.trace[1] : ct.trace[0];
  int32_t* __restrict__ last = y ? ct.last[1] : ct.last[0];
  const int32_t* __restrict__ ids = y ? ct.ids[1] : ct.ids[0];
  const int modulus = y ? ct.modulus[1] : ct.modulus[0];
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int nvec = D / VEC;
  for (int64_t i = (int64_t)blockIdx.x * gpb + threadIdx.x / G; i < n; i += (int64_t)gridDim.x * gpb) {
    const int32_t row = modulus > 0 ? ids[i] % modulus : ids[i];
    int old = 0;
    if (lig == 0) old = atomicExch(&last[row], now);
    old = __shfl(old, (threadIdx.x & 63) & ~(G - 1), kWave);
    const int steps = now - 1 - old;
    if (steps <= 0) continue;
    RowRegs<VEC, NCH> w, a;
    row_load(w, table + (int64_t)row * D, lig, G, nvec);
    row_load(a, trace + (int64_t)row * D, lig, G, nvec);
    const DecayCoef dk = decay_coef(steps, m);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) decay_apply(w.v[k][e], a.v[k][e], dk, lr, m);
    row_store(w, table + (int64_t)row * D, lig, G, nvec);
    row_store(a, trace + (int64_t)row * D, lig, G, nvec);
  }
}

// every row up to step `now` (before an eval, a checkpoint, or anybody reading the plain tables)
template <int VEC, int NCH>
__global__ __launch_bounds__(kBlock) void momentum_flush_kernel(float* __restrict__ table, float* __restrict__ trace,
                                                               int32_t* __restrict__ last, int64_t V, int D, int G,
                                                               int now, float lr, float m) {
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int nvec = D / VEC;
  for (int64_t row = (int64_t)blockIdx.x * gpb + threadIdx.x / G; row < V; row += (int64_t)gridDim.x * gpb) {
    const int steps = now - last[row];
    if (steps <= 0) continue;
    RowRegs<VEC, NCH> w, a;
    row_load(w, table + row * D, lig, G, nvec);
    row_load(a, trace + row * D, lig, G, nvec);
    const DecayCoef dk = decay_coef(steps, m);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) decay_apply(w.v[k][e], a.v[k][e], dk, lr, m);
    row_store(w, table + row * D, lig, G, nvec);
    row_store(a, trace + row * D, lig, G, nvec);
    if (lig == 0) last[row] = now;
  }
}

}  // namespace esr

using namespace esr;

extern "C" {

int esr_momentum_catchup_rows(float* table, float* trace, int32_t* last, int64_t V, int D, const int32_t* ids, int64_t n,
                              int modulus, int step, float lr, float momentum, esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && D > 0 && n >= 0 && step >= 1 && modulus >= 0, "esr_momentum_catchup_rows: bad arguments");
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(table && trace && last && ids, "esr_momentum_catchup_rows: null pointer");
  const RowGeom g = row_geom(D);
  ESR_REQUIRE(g.nch <= kMaxChunksPerLane, "esr_momentum_catchup_rows: D=%d not supported", D);
  const int grid = grid_for_groups(n, g.G);
  CatchupTables ct{{table, nullptr}, {trace, nullptr}, {last, nullptr}, {ids, nullptr}, {modulus, 0}};
  ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((momentum_catchup_kernel<VEC, NCH>), dim3(grid, 1), dim3(kBlock), 0,
                                         as_stream(stream), ct, D, g.G, n, step, lr, momentum));
  return check_launch("esr_momentum_catchup_rows");
}

// two same-width tables, n ids each, one launch (the Spotify step: albums hashed by modulus0, artists as they are)
int esr_momentum_catchup_rows2(float* table0, float* trace0, int32_t* last0, const int32_t* ids0, int modulus0,
                               float* table1, float* trace1, int32_t* last1, const int32_t* ids1, int modulus1, int D,
                               int64_t n, int step, float lr, float momentum, esr_stream_t stream) {
  ESR_REQUIRE(D > 0 && n >= 0 && step >= 1 && modulus0 >= 0 && modulus1 >= 0, "esr_momentum_catchup_rows2: bad arguments");
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(table0 && trace0 && last0 && ids0 && table1 && trace1 && last1 && ids1,
              "esr_momentum_catchup_rows2: null pointer");
  const RowGeom g = row_geom(D);
  ESR_REQUIRE(g.nch <= kMaxChunksPerLane, "esr_momentum_catchup_rows2: D=%d not supported", D);
  const int grid = grid_for_groups(n, g.G);
  CatchupTables ct{{table0, table1}, {trace0, trace1}, {last0, last1}, {ids0, ids1}, {modulus0, modulus1}};
  ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((momentum_catchup_kernel<VEC, NCH>), dim3(grid, 2), dim3(kBlock), 0,
                                         as_stream(stream), ct, D, g.G, n, step, lr, momentum));
  return check_launch("esr_momentum_catchup_rows2");
}

// the whole momentum step on the touched rows of several same-width tables addressed by virtual rows, one launch pair
int esr_sparse_momentum_step_multi(float* const* tables, float* const* traces, const int64_t* row_offsets, int ntables,
                                   int D, const int32_t* sorted_vids, const int32_t* perm, int64_t n, float* grad_rows,
                                   float lr, float momentum, esr_stream_t stream) {
  ESR_REQUIRE(ntables >= 1 && ntables <= kMaxFusedTables && D > 0 && n >= 0,
              "esr_sparse_momentum_step_multi: ntables=%d not in [1, %d] or bad sizes", ntables, kMaxFusedTables);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(tables && traces && row_offsets && sorted_vids && perm && grad_rows,
              "esr_sparse_momentum_step_multi: null pointer");
  FusedTables ft;
  ft.n = ntables;
  for (int i = 0; i < kMaxFusedTables; ++i) {
    ft.table[i] = i < ntables ? tables[i] : nullptr;
    ft.accum[i] = i < ntables ? traces[i] : nullptr;
    ft.row_offset[i] = i <= ntables ? row_offsets[i] : row_offsets[ntables];
    if (i < ntables) {
      ESR_REQUIRE(tables[i] && traces[i] && row_offsets[i + 1] >= row_offsets[i],
                  "esr_sparse_momentum_step_multi: bad table %d", i);
    }
  }
  ft.row_offset[kMaxFusedTables] = row_offsets[ntables];
  ESR_REQUIRE(row_offsets[ntables] < ((int64_t)1 << 31), "esr_sparse_momentum_step_multi: %lld virtual rows >= 2^31",
              (long long)row_offsets[ntables]);
  return launch_segment_tables<kMomentumStep>("esr_sparse_momentum_step_multi", ft, ESR_F32, D, sorted_vids, perm, n,
                                              grad_rows, lr, momentum, as_stream(stream));
}

}  // extern "C"

namespace esr {
// esr_sparse_adagrad_scatter_multi over a SUB-RANGE of a sorted occurrence list (sorted_vids / perm already advanced to the
// range's first position, n = its length; perm still indexes the whole grad_rows buffer).  A range that starts where the
// virtual ids change table (and at a multiple of kSegChunk: the chunk boundaries of long runs are absolute positions)
// gives every row the bits the whole-list launch gives it.  Internal: the in-batch train step updates the scene tower on
// a side stream while pass C runs (esr_inbatch2h.hip).
int sparse_adagrad_range(void* const* tables, float* const* accums, const int64_t* row_offsets, int ntables, int dtype,
                         int D, const int32_t* sorted_vids, const int32_t* perm, int64_t n, float* grad_rows, float lr,
                         float eps, bool skip_long, hipStream_t st) {
  if (!(ntables >= 1 && ntables <= kMaxFusedTables && D > 0 && n >= 0)) {
    set_error("sparse_adagrad_range: bad arguments");
    return ESR_EINVAL;
  }
  if (n == 0) return ESR_OK;
  FusedTables ft;
  ft.n = ntables;
  for (int i = 0; i < kMaxFusedTables; ++i) {
    ft.table[i] = i < ntables ? tables[i] : nullptr;
    ft.accum[i] = i < ntables ? accums[i] : nullptr;
    ft.row_offset[i] = i <= ntables ? row_offsets[i] : row_offsets[ntables];
  }
  ft.row_offset[kMaxFusedTables] = row_offsets[ntables];
  return launch_segment_tables<kAdagrad>("sparse_adagrad_range", ft, dtype, D, sorted_vids, perm, n, grad_rows, lr, eps, st,
                                         skip_long);
}

// ---------------------------------------------------------------------------------------------------------------------
// The in-batch step's merge launches AND its sparse Adagrad update in one kernel (round 5).  The step used to end with
// merge<Q> (8 partial O rows per query row -> gQ, 16 us at B = 8192), merge<C> (-> gC, 13 us) and the update (gQ / gC rows
// in sorted order -> tables, 13 us): 8 MB of gradient rows written and read back and two launches whose only product
// they were.  Here the group at the head of a run of equal ids produces each occurrence's gradient row on the fly --
// merge_row, the merge kernels' own arithmetic -- adds them left to right and applies Adagrad once: the same bits as
// merge + update.  The partner rows come from copies the op's first launch took (the other tower is being updated by
// this very launch); the owned row is the table row the update reads anyway.  Needs a list without runs that outgrow
// their head chunk (the caller's long-run hint says so; a continuation chunk met anyway poisons the loss) and D = 128.
// ---------------------------------------------------------------------------------------------------------------------
template <bool QSIDE>
__device__ __forceinline__ float4 merged_occurrence(const InbatchMergeArgs& a, int64_t r, float4 x, float oscale, int lig,
                                                    float& row_loss) {
  constexpr int side = QSIDE ? 0 : 1;
  const int nsplit = a.nsplit[side];
  float pm[8], pl[8];
  float4 po[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    pm[s] = -INFINITY; pl[s] = 0.f; po[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < nsplit) {
      if (QSIDE) {
        pm[s] = a.part_m[(int64_t)s * a.B + r];
        pl[s] = a.part_l[(int64_t)s * a.B + r];
      }
      po[s] = *reinterpret_cast<const float4*>(a.part_O[side] + ((int64_t)s * a.B + r) * k3D + 4 * lig);
    }
  }
  const float4 y = *reinterpret_cast<const float4*>(a.partner[side] + r * k3D + 4 * lig);
  float M, L, invL1, wt[8];
  return merge_row<QSIDE>(po, pm, pl, nsplit, false, x, y, oscale, a.scale, a.lam, a.inv_bs, 32, row_loss, M, L, invL1, wt);
}

__global__ __launch_bounds__(kBlock) void inbatch_merge_update_kernel(FusedTables ft, int dtype,
                                                                     const int32_t* __restrict__ sorted_ids,
                                                                     const int32_t* __restrict__ perm, int64_t n,
                                                                     InbatchMergeArgs a, float lr, float eps) {
  constexpr int G = 32, D = k3D, nvec = k3D / 4;
  __shared__ long long sm[4];
  if (a.zero_words && blockIdx.x == 0)
    for (int i = threadIdx.x; i < a.nzero; i += kBlock) a.zero_words[i] = 0ull;
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int64_t per = (n + ngroups - 1) / ngroups;  // contiguous slices, as segment_update_kernel
  const int64_t p_end = min(n, (group + 1) * per);
  const float osc_q = a.oscale[0][0], osc_c = a.oscale[1][0];
  long long acc_loss = 0;
  bool bad = false;
  for (int64_t p = group * per; p < p_end; ++p) {
    const int32_t vid = sorted_ids[p];
    const bool head = p == 0 || sorted_ids[p - 1] != vid;
    if (!head) {
      // a continuation chunk (segment_update_kernel's rule) cannot exist in a list the hint cleared
      if ((p & (kSegChunk - 1)) == 0 && p >= kSegChunk && sorted_ids[p - kSegChunk] == vid) bad = true;
      continue;
    }
    const int64_t stop = min(((p + 2 * kSegChunk - 1) / kSegChunk) * kSegChunk, n);
    void* table = ft.table[0];
    float* accum = ft.accum[0];
    const bool qside = (int64_t)vid < ft.row_offset[1];
    int64_t id = vid;
    if (!qside) {
      table = ft.table[1];
      accum = ft.accum[1];
      id = (int64_t)vid - ft.row_offset[1];
    }
    RowRegs<4, 1> w, ac;
    param_load(w, table, dtype, id, D, lig, G, nvec);
    row_load(ac, accum + id * D, lig, G, nvec);
    const float4 x = make_float4(w.v[0][0], w.v[0][1], w.v[0][2], w.v[0][3]);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t q = p; q < stop; ++q) {
      if (q > p && sorted_ids[q] != vid) break;
      const int64_t o = perm[q];
      float row_loss;
      const float4 t = qside ? merged_occurrence<true>(a, o, x, osc_q, lig, row_loss)
                             : merged_occurrence<false>(a, o - a.B, x, osc_c, lig, row_loss);
      if (q == p) {
        g = t;
      } else {
        // t is a FINISHED gradient row (what merge<Q / C> stores and segment_update_kernel reads back): its last
        // operation, the multiplication by 1 / batch_size, must not be contracted into this addition
        float4 tt = t;
        asm volatile("" : "+v"(tt.x), "+v"(tt.y), "+v"(tt.z), "+v"(tt.w));
        g.x += tt.x; g.y += tt.y; g.z += tt.z; g.w += tt.w;
      }
      if (lig == 0) acc_loss += loss_fixed(row_loss, bad);
      if (q + 1 == stop && q + 1 < n && sorted_ids[q + 1] == vid) bad = true;  // the run outgrew its head chunk
    }
    adagrad_elem(w.v[0][0], ac.v[0][0], g.x, lr, eps);
    adagrad_elem(w.v[0][1], ac.v[0][1], g.y, lr, eps);
    adagrad_elem(w.v[0][2], ac.v[0][2], g.z, lr, eps);
    adagrad_elem(w.v[0][3], ac.v[0][3], g.w, lr, eps);
    row_store(ac, accum + id * D, lig, G, nvec);
    param_store(w, table, dtype, id, D, lig, G, nvec);
  }
  const long long tsum = block_sum_ll(acc_loss, sm);
  const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
  if (threadIdx.x == 0) loss_arrive(tsum, any_bad, 1, a.loss_acc, a.loss_scale, a.loss_out);
}

int inbatch_merge_update(void* const* tables, float* const* accums, const int64_t* row_offsets, int dtype,
                         const int32_t* sorted_vids, const int32_t* perm, const InbatchMergeArgs& a, float lr, float eps,
                         hipStream_t st) {
  FusedTables ft;
  ft.n = 2;
  for (int i = 0; i < kMaxFusedTables; ++i) {
    ft.table[i] = i < 2 ? tables[i] : nullptr;
    ft.accum[i] = i < 2 ? accums[i] : nullptr;
    ft.row_offset[i] = i <= 2 ? row_offsets[i] : row_offsets[2];
  }
  ft.row_offset[kMaxFusedTables] = row_offsets[2];
  const int64_t n = 2 * a.B;
  // two positions per row group on long lists (the second one's id words arrive under the first one's rows): -0.2 .. -1.8 us
  // of 26 at B = 8192 in four same-box pairs; four per group: no better than one.  The slices only divide the work:
  // every row gets the same bits from any grid.
  const int grid = n >= 4096 ? std::max(1, grid_for_groups(n, 32) / 2) : grid_for_groups(n, 32);
  ESR_KT("inbatch_merge_update_kernel", st,
         hipLaunchKernelGGL(inbatch_merge_update_kernel, dim3(grid), dim3(kBlock), 0, st, ft, dtype, sorted_vids, perm, n, a,
                            lr, eps));
  return check_launch("inbatch_merge_update");
}

// esr_sparse_momentum_step_multi for one or two tables whose rows may be behind (last[t][row] = the step the row is current
// with): catch-up + step + mark, in the update kernel itself (kMomentumStepLazy).  Internal: esr_spotify_train_step.
int sparse_momentum_step_lazy2(float* const* tables, float* const* traces, int32_t* const* lasts, const int64_t* row_offsets,
                               int ntables, int D, const int32_t* sorted_vids, const int32_t* perm, int64_t n, float* grad_rows,
                               float lr, float momentum, int now, hipStream_t st) {
  if (!(ntables >= 1 && ntables <= 2 && D > 0 && n >= 0 && now >= 1)) {
    set_error("sparse_momentum_step_lazy2: ntables=%d not in [1, 2] or bad sizes", ntables);
    return ESR_EINVAL;
  }
  if (n == 0) return ESR_OK;
  FusedTables ft;
  ft.n = ntables;
  for (int i = 0; i < kMaxFusedTables; ++i) {
    ft.table[i] = i < ntables ? (void*)tables[i] : nullptr;
    ft.accum[i] = i < ntables ? traces[i] : nullptr;
    ft.row_offset[i] = i <= ntables ? row_offsets[i] : row_offsets[ntables];
  }
  ft.table[2] = lasts[0];
  ft.table[3] = ntables > 1 ? lasts[1] : nullptr;
  ft.row_offset[kMaxFusedTables] = now;
  return launch_segment_tables<kMomentumStepLazy>("sparse_momentum_step_lazy2", ft, ESR_F32, D, sorted_vids, perm, n,
                                                  grad_rows, lr, momentum, st);
}
}  // namespace esr

extern "C" {

int esr_momentum_flush(float* table, float* trace, int32_t* last, int64_t V, int D, int step, float lr, float momentum,
                       esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && D > 0 && step >= 0, "esr_momentum_flush: bad arguments");
  ESR_REQUIRE(table && trace && last, "esr_momentum_flush: null pointer");
  const RowGeom g = row_geom(D);
  ESR_REQUIRE(g.nch <= kMaxChunksPerLane, "esr_momentum_flush: D=%d not supported", D);
  const int grid = grid_for_groups(V, g.G);
  ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((momentum_flush_kernel<VEC, NCH>), dim3(grid), dim3(kBlock), 0, as_stream(stream),
                                         table, trace, last, V, D, g.G, step, lr, momentum));
  return check_launch("esr_momentum_flush");
}

int esr_sparse_momentum_step(float* table, float* trace, int64_t V, int D, const int32_t* sorted_ids,
                             const int32_t* perm, int64_t n, float* grad_rows, float lr, float momentum,
                             esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && D > 0 && n >= 0, "esr_sparse_momentum_step: bad sizes V=%lld D=%d n=%lld", (long long)V, D,
              (long long)n);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(table && trace && sorted_ids && perm && grad_rows, "esr_sparse_momentum_step: null pointer");
  return launch_segment_update<kMomentumStep>("esr_sparse_momentum_step", table, ESR_F32, trace, D, sorted_ids, perm, n,
                                              grad_rows, lr, momentum, as_stream(stream));
}

int esr_dense_momentum_decay(float* param, float* trace, int64_t count, float lr, float momentum,
                             esr_stream_t stream) {
  ESR_REQUIRE(count >= 0, "esr_dense_momentum_decay: bad count %lld", (long long)count);
  if (count == 0) return ESR_OK;
  ESR_REQUIRE(param && trace, "esr_dense_momentum_decay: null pointer");
  ESR_REQUIRE((((uintptr_t)param | (uintptr_t)trace) & 15) == 0, "esr_dense_momentum_decay: pointers must be 16-byte aligned");
  const int64_t n4 = count / 4;
  const int grid = (int)std::min<int64_t>(std::max<int64_t>(cdiv(n4, kBlock), 1), 8192);
  hipLaunchKernelGGL(momentum_decay_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), param, trace, n4, count, lr,
                     momentum);
  return check_launch("esr_dense_momentum_decay");
}

int esr_sparse_momentum_scatter(float* table, float* trace, int64_t V, int D, const int32_t* sorted_ids,
                                const int32_t* perm, int64_t n, float* grad_rows, float lr,
                                esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && D > 0 && n >= 0, "esr_sparse_momentum_scatter: bad sizes V=%lld D=%d n=%lld", (long long)V, D,
              (long long)n);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(table && trace && sorted_ids && perm && grad_rows, "esr_sparse_momentum_scatter: null pointer");
  return launch_segment_update<kMomentum>("esr_sparse_momentum_scatter", table, ESR_F32, trace, D, sorted_ids, perm, n,
                                          grad_rows, lr, 0.f, as_stream(stream));
}

int esr_sparse_adagrad_scatter(void* table, int dtype, float* accum, int64_t V, int D, const int32_t* sorted_ids,
                               const int32_t* perm, int64_t n, float* grad_rows, float lr, float eps,
                               esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && D > 0 && n >= 0, "esr_sparse_adagrad_scatter: bad sizes V=%lld D=%d n=%lld", (long long)V, D,
              (long long)n);
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_sparse_adagrad_scatter: bad dtype %d", dtype);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(table && accum && sorted_ids && perm && grad_rows, "esr_sparse_adagrad_scatter: null pointer");
  return launch_segment_update<kAdagrad>("esr_sparse_adagrad_scatter", table, dtype, accum, D, sorted_ids, perm, n,
                                         grad_rows, lr, eps, as_stream(stream));
}

int esr_sparse_sgd_scatter(void* table, int dtype, int64_t V, int D, const int32_t* sorted_ids, const int32_t* perm,
                           int64_t n, float* grad_rows, float lr, esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && D > 0 && n >= 0, "esr_sparse_sgd_scatter: bad sizes V=%lld D=%d n=%lld", (long long)V, D,
              (long long)n);
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_sparse_sgd_scatter: bad dtype %d", dtype);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(table && sorted_ids && perm && grad_rows, "esr_sparse_sgd_scatter: null pointer");
  return launch_segment_update<kSgd>("esr_sparse_sgd_scatter", table, dtype, nullptr, D, sorted_ids, perm, n,
                                     grad_rows, lr, 0.f, as_stream(stream));
}

int esr_rows_to_dense(float* dense, int64_t V, int D, const int32_t* sorted_ids, const int32_t* perm, int64_t n,
                      float* grad_rows, esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && D > 0 && n >= 0, "esr_rows_to_dense: bad sizes V=%lld D=%d n=%lld", (long long)V, D,
              (long long)n);
  ESR_REQUIRE(dense, "esr_rows_to_dense: null pointer");
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(dense, 0, sizeof(float) * (size_t)V * D, st) != hipSuccess)
    return check_launch("esr_rows_to_dense memset");
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(sorted_ids && perm && grad_rows, "esr_rows_to_dense: null pointer");
  return launch_segment_update<kToDense>("esr_rows_to_dense", dense, ESR_F32, nullptr, D, sorted_ids, perm, n,
                                         grad_rows, 0.f, 0.f, st);
}

int esr_segment_sum_rows(float* out, int64_t rows_out, int D, const int32_t* sorted_ids, const int32_t* perm, int64_t n,
                         float* grad_rows, esr_stream_t stream) {
  ESR_REQUIRE(rows_out > 0 && D > 0 && n >= 0, "esr_segment_sum_rows: bad sizes rows=%lld D=%d n=%lld", (long long)rows_out,
              D, (long long)n);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(out && sorted_ids && perm && grad_rows, "esr_segment_sum_rows: null pointer");
  return launch_segment_update<kToDense>("esr_segment_sum_rows", out, ESR_F32, nullptr, D, sorted_ids, perm, n, grad_rows,
                                         0.f, 0.f, as_stream(stream));
}

int esr_concat_offset_ids(const int32_t* const* ids, const int64_t* counts, const int64_t* offsets, int nseg,
                          int32_t* out, esr_stream_t stream) {
  ESR_REQUIRE(nseg >= 1 && nseg <= kMaxFusedTables, "esr_concat_offset_ids: nseg=%d not in [1, %d]", nseg,
              kMaxFusedTables);
  ESR_REQUIRE(ids && counts && offsets && out, "esr_concat_offset_ids: null pointer");
  IdSegments sg;
  sg.n = nseg;
  sg.start[0] = 0;
  for (int i = 0; i < kMaxFusedTables; ++i) {
    sg.ids[i] = i < nseg ? ids[i] : nullptr;
    sg.offset[i] = i < nseg ? offsets[i] : 0;
    sg.start[i + 1] = sg.start[i] + (i < nseg ? counts[i] : 0);
    if (i < nseg) {
      ESR_REQUIRE(counts[i] >= 0 && (counts[i] == 0 || ids[i]), "esr_concat_offset_ids: bad segment %d", i);
    }
  }
  const int64_t total = sg.start[nseg];
  if (total == 0) return ESR_OK;
  const int grid = (int)std::min<int64_t>(kMaxGrid, cdiv(total, kBlock));
  hipLaunchKernelGGL(concat_offset_ids_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), sg, out);
  return check_launch("esr_concat_offset_ids");
}

int esr_gather_rows_multi(const void* const* tables, const int64_t* row_offsets, int ntables, int dtype, int D,
                          const int32_t* vids, int64_t n, void* out, esr_stream_t stream) {
  ESR_REQUIRE(ntables >= 1 && ntables <= kMaxFusedTables, "esr_gather_rows_multi: ntables=%d not in [1, %d]", ntables,
              kMaxFusedTables);
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_gather_rows_multi: bad dtype %d", dtype);
  ESR_REQUIRE(D > 0 && n >= 0, "esr_gather_rows_multi: bad sizes D=%d n=%lld", D, (long long)n);
  const int64_t row_bytes = (int64_t)D * (dtype == ESR_BF16 ? 2 : 4);
  ESR_REQUIRE(row_bytes % 16 == 0, "esr_gather_rows_multi: row of %lld bytes is not a multiple of 16", (long long)row_bytes);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(tables && row_offsets && vids && out, "esr_gather_rows_multi: null pointer");
  FusedTables ft;
  ft.n = ntables;
  for (int i = 0; i < kMaxFusedTables; ++i) {
    ft.table[i] = i < ntables ? const_cast<void*>(tables[i]) : nullptr;
    ft.accum[i] = nullptr;
    ft.row_offset[i] = i <= ntables ? row_offsets[i] : row_offsets[ntables];
    if (i < ntables) ESR_REQUIRE(tables[i], "esr_gather_rows_multi: null table %d", i);
  }
  ft.row_offset[kMaxFusedTables] = row_offsets[ntables];
  const int nchunk = (int)(row_bytes / 16);
  int G = 1;
  while (G < nchunk && G < kWave) G <<= 1;
  const int grid = grid_for_groups(n, G);
  hipLaunchKernelGGL(gather_rows_multi_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), ft, nchunk, G, vids, n,
                     (uint4*)out);
  return check_launch("esr_gather_rows_multi");
}

int esr_sparse_adagrad_scatter_multi(void* const* tables, float* const* accums, const int64_t* row_offsets,
                                     int ntables, int dtype, int D, const int32_t* sorted_vids, const int32_t* perm,
                                     int64_t n, float* grad_rows, float lr, float eps, int long_runs,
                                     esr_stream_t stream) {
  TraceScope trace_scope_("esr_sparse_adagrad_scatter_multi");
  ESR_REQUIRE(ntables >= 1 && ntables <= kMaxFusedTables, "esr_sparse_adagrad_scatter_multi: ntables=%d not in [1, %d]",
              ntables, kMaxFusedTables);
  ESR_REQUIRE(D > 0 && n >= 0, "esr_sparse_adagrad_scatter_multi: bad sizes D=%d n=%lld", D, (long long)n);
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_sparse_adagrad_scatter_multi: bad dtype %d", dtype);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(tables && accums && row_offsets && sorted_vids && perm && grad_rows,
              "esr_sparse_adagrad_scatter_multi: null pointer");
  FusedTables ft;
  ft.n = ntables;
  for (int i = 0; i < kMaxFusedTables; ++i) {
    ft.table[i] = i < ntables ? tables[i] : nullptr;
    ft.accum[i] = i < ntables ? accums[i] : nullptr;
    ft.row_offset[i] = i <= ntables ? row_offsets[i] : row_offsets[ntables];
    if (i < ntables) {
      ESR_REQUIRE(tables[i] && accums[i] && row_offsets[i + 1] >= row_offsets[i],
                  "esr_sparse_adagrad_scatter_multi: bad table %d", i);
    }
  }
  ft.row_offset[kMaxFusedTables] = row_offsets[ntables];
  ESR_REQUIRE(row_offsets[ntables] < ((int64_t)1 << 31), "esr_sparse_adagrad_scatter_multi: %lld virtual rows >= 2^31",
              (long long)row_offsets[ntables]);
  return launch_segment_tables<kAdagrad>("esr_sparse_adagrad_scatter_multi", ft, dtype, D, sorted_vids, perm, n,
                                         grad_rows, lr, eps, as_stream(stream), long_runs == 0);
}

int esr_dense_adam(float* param, float* mu, float* nu, const float* grad, int64_t numel, float lr, float b1,
                   float b2, float eps, int64_t step, esr_stream_t stream) {
  ESR_REQUIRE(numel >= 0 && step >= 1, "esr_dense_adam: bad numel=%lld step=%lld", (long long)numel, (long long)step);
  if (numel == 0) return ESR_OK;
  ESR_REQUIRE(param && mu && nu && grad, "esr_dense_adam: null pointer");
  ESR_REQUIRE((((uintptr_t)param | (uintptr_t)mu | (uintptr_t)nu | (uintptr_t)grad) & 15) == 0,
              "esr_dense_adam: pointers must be 16-byte aligned");
  // bias corrections in fp64 on the host, applied as fp32 reciprocals
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  const int64_t n4 = numel / 4;
  const int grid = (int)std::min<int64_t>(kMaxGrid, std::max<int64_t>(1, cdiv(n4, kBlock)));
  hipLaunchKernelGGL(dense_adam_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), param, mu, nu, grad, n4,
                     numel, lr, b1, b2, eps, (float)(1.0 / bc1), (float)(1.0 / bc2));
  return check_launch("esr_dense_adam");
}

}  // extern "C"
