// The whole Shop-The-Look train step in one pass over the rows: esr_triplet_train_step.
//
// Reference arithmetic: pinterest/train_shop_the_look.py:93-109 (triplet hinge + norm-excess regulariser,
// value_and_grad, one optimizer update) on the id towers that replace pinterest/models.py:64-70, with the build's
// row-sparse Adagrad.  esr_triplet_fwd_bwd + sort + esr_sparse_adagrad_scatter_multi is six dependent launches that
// write a [3B, D] gradient and read it back; at the reference's batch sizes that chain of launches IS the step time.
// Here the gradient of an occurrence is formed on chip inside the update kernel from the two OTHER rows of its triplet,
// which the double-buffered towers (esr_versioned.h) make safe to read while rows are being rewritten:
//
//   sort      virtual occurrence ids [scene ; Vs + pos ; Vs + neg] -> (sorted, perm)        } ids only: made AHEAD,
//   plan      per sorted position: the occurrence's slot and its two partner rows;           } for several batches
//             "may a run outgrow its head chunk?" (the host's reason to launch `long` at all) } by one launch each
//   update    one row group per sorted position, the head of a run walks it: row locations resolved from the stamped
//             bytes (round 3: no per-step plan launch), both partner rows -> pos / neg score, hinge mask, own norm ->
//             this occurrence's gradient row; summed left to right; Adagrad once per distinct row into the other
//             buffer.  The scene occurrence of a triplet also contributes its loss term; the loss leaves the kernel
//             through an exact fixed-point reduction (no finalize launch).
//   long      only when a run outgrew its head chunk (hot rows): chunk partials combined in a fixed order
//
// Round 2 ran plan -> update -> long per step (4.7 + 18.1 + 4.5 us at B = 8192); a step of uniform ids is now the
// update kernel alone.
#include "esr_common.h"
#include "esr_versioned.h"

namespace esr {

// Cut points of long runs: every 8 positions instead of the 32 of the other segment kernels.  An occurrence costs this
// kernel two dependent round trips (plan record -> two partner rows), so a hot row is a LONG sequential walk per chunk:
// with 32-position chunks a Zipf(1) batch of 8192 triplets took 131 us (99 us before this kernel existed); shorter
// chunks spread the walk over four times as many row groups.
constexpr int kTripChunk = 8;
constexpr uint32_t kSlotShift = 30;  // plan record .x = partner A's virtual row | slot << 30

struct TwoTowers {
  float* s0;  // scene tower, primary buffer            virtual rows [0, Vs)
  float* s1;  //              second buffer
  float* p0;  // product tower, primary buffer          virtual rows [Vs, Vs + Vp)
  float* p1;
  uint8_t* sloc;
  uint8_t* ploc;
  float* sacc;
  float* pacc;
  int64_t Vs;
  uint32_t stamp;  // this step's stamp (esr_versioned.h)
};

__device__ __forceinline__ const float* tower_row(const TwoTowers& tt, uint32_t code, int D) {
  const int64_t vid = code & kIdMask;
  const bool prod = vid >= tt.Vs, second = (code & kLocBit) != 0;
  const float* base = prod ? (second ? tt.p1 : tt.p0) : (second ? tt.s1 : tt.s0);
  return base + (prod ? vid - tt.Vs : vid) * D;
}
__device__ __forceinline__ const float* tower_acc(const TwoTowers& tt, uint32_t vid, int D) {
  return vid >= tt.Vs ? tt.pacc + (int64_t)(vid - tt.Vs) * D : tt.sacc + (int64_t)vid * D;
}
// the location byte of virtual row vid, as it reads now
__device__ __forceinline__ uint32_t loc_byte(const TwoTowers& tt, uint32_t vid) {
  return vid >= tt.Vs ? tt.ploc[vid - tt.Vs] : tt.sloc[vid];
}
// row code (id | buffer bit) of the value the row had when this step began
__device__ __forceinline__ uint32_t code_of(uint32_t vid, uint32_t byte, uint32_t T) {
  return vid | (loc_at_step_begin(byte, T) ? kLocBit : 0u);
}

// ---- the plan of one batch: everything the update kernel needs besides the tables, from the ids alone ----------------
struct TripPlan {
  int* flags;                   // [0] parked: set by the update kernel when it parks a chunk partial; [2..3] direct mode:
                                //     one 8-byte word, plan generation << 32 | number of long runs (= entries of
                                //     long_heads); the rest unused
  unsigned long long* loss_acc; // [kFixAccWords]  fixed-point loss accumulator, zero before the update kernel
  uint2* meta;                  // [n]  stamped mode: {partner A | slot << 30, partner B} per sorted position;
                                //      direct mode: the occurrence codes (see triplet_direct_plan_kernel) per OCCURRENCE
  uint32_t* cnt;                // [n]  direct mode: arrivals per run, at the run's head position; zero before the step
                                //      (zeroed by the plan kernel, and again by the arrival that completes a run)
  int32_t* long_heads;          // [n / 9 + 1]  direct mode: head positions of the runs longer than kDirectMaxRun
  double* loss_part;            // [kMaxGrid]  direct mode: the update kernel's loss partial per workgroup
};
static size_t trip_plan_layout(int64_t B, char* base, TripPlan* out) {
  const int64_t n = 3 * B;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  TripPlan pl;
  pl.flags = (int*)take(sizeof(int) * 64);
  pl.loss_acc = (unsigned long long*)take(sizeof(unsigned long long) * kFixAccWords);
  pl.meta = (uint2*)take(sizeof(uint2) * (size_t)n);
  pl.cnt = (uint32_t*)take(sizeof(uint32_t) * (size_t)n);
  pl.long_heads = (int32_t*)take(sizeof(int32_t) * (size_t)(n / 9 + 1));
  pl.loss_part = (double*)take(sizeof(double) * kMaxGrid);
  if (out) *out = pl;
  return off;
}

struct TripWs {
  int32_t* sorted_ids;  // [n]
  int32_t* perm;        // [n]
  float* chunk_rows;    // [n][D]  stamped mode: [2 * ceil(n / 8)][D] partial sums of long runs; direct mode: the gradient
                        //         row of every occurrence of a DUPLICATED row, at its sorted position
  char* plan;           // trip_plan_layout(B) bytes: the in-line plan of a call that brings none
  void* sort_ws;
  size_t sort_ws_bytes;
};

static size_t trip_ws_layout(int64_t B, int D, char* base, TripWs* ws) {
  const int64_t n = 3 * B;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  TripWs w;
  w.sorted_ids = (int32_t*)take(sizeof(int32_t) * (size_t)n);
  w.perm = (int32_t*)take(sizeof(int32_t) * (size_t)n);
  w.chunk_rows = (float*)take(sizeof(float) * (size_t)n * (size_t)D);
  w.plan = take(trip_plan_layout(B, nullptr, nullptr));
  w.sort_ws_bytes = esr_segment_sort_workspace_bytes(n);
  w.sort_ws = take(w.sort_ws_bytes);
  if (ws) *ws = w;
  return off;
}

constexpr int kMaxPlanBatch = 8;
struct PlanBatch {
  const int32_t* scene[kMaxPlanBatch];
  const int32_t* pos[kMaxPlanBatch];
  const int32_t* neg[kMaxPlanBatch];
};

// plan: one thread per sorted position of list blockIdx.y.  Occurrence o = perm[p]: slot = o / B (0 scene, 1 pos, 2 neg),
// triplet b = o % B.  Partners: scene -> (pos, neg); pos -> (scene, neg); neg -> (scene, pos).  hints[list] = gen when
// some run of equal ids is longer than kTripChunk positions -- only then can the update kernel park a chunk partial and
// need the `long` launch (a head chunk covers at least kTripChunk positions).  `gen` is the caller's call counter: a
// word that need not be cleared.
__global__ __launch_bounds__(kBlock) void triplet_plan_kernel(PlanBatch pb, const int32_t* __restrict__ sorted_all,
                                                             const int32_t* __restrict__ perm_all, int64_t B,
                                                             int64_t Vs, char* __restrict__ plans, size_t plan_stride,
                                                             int* __restrict__ hints, int gen) {
  const int list = blockIdx.y;
  const int64_t n = 3 * B;
  const int32_t* __restrict__ sorted = sorted_all + (int64_t)list * n;
  const int32_t* __restrict__ perm = perm_all + (int64_t)list * n;
  const int32_t* __restrict__ scene_ids = pb.scene[list];
  const int32_t* __restrict__ pos_ids = pb.pos[list];
  const int32_t* __restrict__ neg_ids = pb.neg[list];
  char* base = plans + (size_t)list * plan_stride;
  int* flags = (int*)base;
  unsigned long long* loss_acc = (unsigned long long*)(base + 256);
  uint2* meta = (uint2*)(base + 256 + align_up(sizeof(unsigned long long) * kFixAccWords, 256));
  if (blockIdx.x == 0) {
    if (threadIdx.x < 64) flags[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < kFixAccWords; i += kBlock) loss_acc[i] = 0ull;
  }
  bool long_run = false;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
    const int64_t o = perm[p];
    const int slot = o >= 2 * B ? 2 : (o >= B ? 1 : 0);
    const int64_t b = o - slot * B;
    const uint32_t sc = (uint32_t)scene_ids[b], pc = (uint32_t)(Vs + pos_ids[b]), nc = (uint32_t)(Vs + neg_ids[b]);
    meta[p] = make_uint2((slot == 0 ? pc : sc) | ((uint32_t)slot << kSlotShift), slot == 2 ? pc : nc);
    if (p + kTripChunk < n && sorted[p] == sorted[p + kTripChunk]) long_run = true;
  }
  if (hints && __any(long_run) && (threadIdx.x & 63) == 0) hints[list] = gen;  // (every writer stores the same value)
}

template <int VEC, int NCH>
__device__ __forceinline__ void step_apply2(const TwoTowers& tt, uint32_t code, const RowRegs<VEC, NCH>& own,
                                            RowRegs<VEC, NCH>& a, const RowRegs<VEC, NCH>& g, int D, int lig, int G,
                                            int nvec, float lr, float eps) {
  const int64_t vid = code & kIdMask;
  const bool prod = vid >= tt.Vs, second = (code & kLocBit) != 0;
  const int64_t id = prod ? vid - tt.Vs : vid;
  RowRegs<VEC, NCH> w = own;
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) adagrad_elem(w.v[k][e], a.v[k][e], g.v[k][e], lr, eps);
  row_store(a, (prod ? tt.pacc : tt.sacc) + id * D, lig, G, nvec);
  float* dst = prod ? (second ? tt.p0 : tt.p1) : (second ? tt.s0 : tt.s1);  // the OTHER buffer
  row_store(w, dst + id * D, lig, G, nvec);
  if (lig == 0) (prod ? tt.ploc : tt.sloc)[id] = loc_written(second ? 0u : 1u, tt.stamp);
}

// one position's plan record with the location bytes of its three rows
struct TripRec {
  uint32_t id;      // own virtual row
  uint2 m;          // {A | slot << 30, B}
  uint32_t y0, ya, yb;  // location bytes of own, A, B (as loaded)
};
__device__ __forceinline__ void rec_bytes(const TwoTowers& tt, TripRec& r) {
  r.y0 = loc_byte(tt, r.id);
  r.ya = loc_byte(tt, r.m.x & kIdMask);
  r.yb = loc_byte(tt, r.m.y);
}

template <int VEC, int NCH>
__global__ __launch_bounds__(kBlock) void triplet_step_kernel(TwoTowers tt, int D, int G,
                                                             const int32_t* __restrict__ sorted_ids,
                                                             const uint2* __restrict__ meta, int64_t n, float lam,
                                                             float inv_bs, int with_reg, float lr, float eps,
                                                             float* __restrict__ chunk_rows, int* __restrict__ parked,
                                                             unsigned long long* __restrict__ loss_acc, int frac,
                                                             double inv_batch_size, float* __restrict__ loss) {
  __shared__ double sm[8];
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int nvec = D / VEC;
  const int64_t per = (n + ngroups - 1) / ngroups;
  const int64_t p_begin = group * per, p_end = min(n, (group + 1) * per);
  const uint32_t T = tt.stamp;
  // Pipeline per group: records run THREE positions ahead, location bytes two, rows one.  r0 = position p (bytes
  // landed), r1 = p + 1 (bytes requested an iteration ago), r2 = p + 2 (record requested an iteration ago).
  TripRec r0{0, make_uint2(0, 0), 0, 0, 0}, r1 = r0, r2 = r0;
  uint32_t prev_n = 0xFFFFFFFFu;
  if (p_begin < p_end) {
    r0.id = (uint32_t)sorted_ids[p_begin];
    r0.m = meta[p_begin];
    if (p_begin > 0) prev_n = (uint32_t)sorted_ids[p_begin - 1];
    if (p_begin + 1 < n) {
      r1.id = (uint32_t)sorted_ids[p_begin + 1];
      r1.m = meta[p_begin + 1];
    }
    if (p_begin + 2 < n) {
      r2.id = (uint32_t)sorted_ids[p_begin + 2];
      r2.m = meta[p_begin + 2];
    }
    rec_bytes(tt, r0);
    if (p_begin + 1 < n) rec_bytes(tt, r1);
  }
  double acc_loss = 0.0;
  // rows of the next position requested ahead only while rows are short in registers (four rows of NCH * VEC floats)
  constexpr bool kAhead = NCH <= 2;
  bool have_next = false;
  RowRegs<VEC, NCH> nown, na, nA, nB;

  for (int64_t p = p_begin; p < p_end; ++p) {
    const TripRec cur = r0, nxt = r1;
    const uint32_t prev = prev_n;
    const bool more = p + 1 < n;
    r0 = r1;
    r1 = r2;
    if (p + 2 < n) rec_bytes(tt, r1);  // bytes of position p + 2 (its record arrived an iteration ago)
    if (p + 3 < n) {
      r2.id = (uint32_t)sorted_ids[p + 3];
      r2.m = meta[p + 3];
    }
    prev_n = cur.id;
    const uint32_t id = cur.id;
    const bool head = prev != id;  // prev = all ones at p == 0: no id equals it
    if (!head && ((p & (kTripChunk - 1)) != 0 || (uint32_t)sorted_ids[p - kTripChunk] != id)) continue;
    const int64_t stop = min(head ? ((p + 2 * kTripChunk - 1) / kTripChunk) * kTripChunk : p + kTripChunk, n);
    const uint32_t code = code_of(id, cur.y0, T);
    RowRegs<VEC, NCH> own, a, g, fA, fB;
    if (have_next) {
      own = nown;
      a = na;
      fA = nA;
      fB = nB;
    } else {
      row_load(own, tower_row(tt, code, D), lig, G, nvec);
      row_load(fA, tower_row(tt, code_of(cur.m.x & kIdMask, cur.ya, T), D), lig, G, nvec);
      row_load(fB, tower_row(tt, code_of(cur.m.y, cur.yb, T), D), lig, G, nvec);
      row_load(a, tower_acc(tt, id, D), lig, G, nvec);
    }
    have_next = false;
    int64_t e_run = p + 1;
    if (more && nxt.id == id) {
      ++e_run;
      if (e_run < stop && r1.id == id && p + 2 < n) {  // (position p + 2's record is in r1 by now)
        ++e_run;
        while (e_run < stop && (uint32_t)sorted_ids[e_run] == id) ++e_run;
      }
      if (e_run > stop) e_run = stop;
    } else if (kAhead && p + 1 < p_end) {  // a run of one: position p + 1 heads the next run -- request its rows now
      row_load(nown, tower_row(tt, code_of(nxt.id, nxt.y0, T), D), lig, G, nvec);
      row_load(nA, tower_row(tt, code_of(nxt.m.x & kIdMask, nxt.ya, T), D), lig, G, nvec);
      row_load(nB, tower_row(tt, code_of(nxt.m.y, nxt.yb, T), D), lig, G, nvec);
      row_load(na, tower_acc(tt, nxt.id, D), lig, G, nvec);
      have_next = true;
    }
    row_zero(g);
    // own norm: the same for every occurrence of the run
    float c = 0.f, own_norm = 0.f;
    if (with_reg) {
      own_norm = sqrtf(group_sum(row_dot_partial(own, own), G));
      c = own_norm > 1.f ? lam / own_norm : 0.f;
    }
    auto occ = [&](uint32_t slot, const RowRegs<VEC, NCH>& A, const RowRegs<VEC, NCH>& Bq) {
      // scene: A = pos, B = neg ; pos: A = scene, B = neg ; neg: A = scene, B = pos
      const float d_oa = group_sum(row_dot_partial(own, A), G);
      const float d_ob = group_sum(row_dot_partial(own, Bq), G);
      const float d_ab = group_sum(row_dot_partial(A, Bq), G);
      const float ps = slot == 0 ? d_oa : (slot == 1 ? d_oa : d_ab);  // scene . pos
      const float ns = slot == 0 ? d_ob : (slot == 1 ? d_ab : d_oa);  // scene . neg
      const float margin = 1.0f + ns - ps;
      const float mk = margin > 0.f ? 1.f : 0.f;
      const float coef = slot == 1 ? -mk : mk;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float x = slot == 0 ? __fsub_rn(Bq.v[k][e], A.v[k][e]) : A.v[k][e];
          g.v[k][e] = __fadd_rn(g.v[k][e], trip_grad(coef, x, c, own.v[k][e], inv_bs));
        }
      if (slot == 0) {  // the triplet's loss term, once (train_shop_the_look.py:99-104)
        float loss_b = fmaxf(margin, 0.f);
        if (with_reg) {
          const float pn = sqrtf(group_sum(row_dot_partial(A, A), G));
          const float nn = sqrtf(group_sum(row_dot_partial(Bq, Bq), G));
          loss_b += lam * (fmaxf(own_norm - 1.f, 0.f) + fmaxf(pn - 1.f, 0.f) + fmaxf(nn - 1.f, 0.f));
        }
        if (lig == 0) acc_loss += (double)loss_b;
      }
    };
    occ(cur.m.x >> kSlotShift, fA, fB);
    int64_t q = p + 1;
    // the rest of the run: record -> location bytes -> rows are three dependent round trips, so two occurrences (four
    // partner rows) travel together; position p + 1's record and bytes are in registers already
    for (; q < e_run; q += 2) {
      const bool two = q + 1 < e_run;
      TripRec ra, rb;
      if (q == p + 1) {
        ra = nxt;
      } else {
        ra.id = id;
        ra.m = meta[q];
      }
      rb.id = id;
      rb.m = two ? meta[q + 1] : ra.m;
      if (q != p + 1) {
        ra.ya = loc_byte(tt, ra.m.x & kIdMask);
        ra.yb = loc_byte(tt, ra.m.y);
      }
      rb.ya = loc_byte(tt, rb.m.x & kIdMask);
      rb.yb = loc_byte(tt, rb.m.y);
      RowRegs<VEC, NCH> a0, b0;
      row_load(a0, tower_row(tt, code_of(ra.m.x & kIdMask, ra.ya, T), D), lig, G, nvec);
      row_load(b0, tower_row(tt, code_of(ra.m.y, ra.yb, T), D), lig, G, nvec);
      if (two && kAhead) {
        RowRegs<VEC, NCH> a1, b1;
        row_load(a1, tower_row(tt, code_of(rb.m.x & kIdMask, rb.ya, T), D), lig, G, nvec);
        row_load(b1, tower_row(tt, code_of(rb.m.y, rb.yb, T), D), lig, G, nvec);
        occ(ra.m.x >> kSlotShift, a0, b0);
        occ(rb.m.x >> kSlotShift, a1, b1);
      } else {
        occ(ra.m.x >> kSlotShift, a0, b0);
        if (two) {
          row_load(a0, tower_row(tt, code_of(rb.m.x & kIdMask, rb.ya, T), D), lig, G, nvec);
          row_load(b0, tower_row(tt, code_of(rb.m.y, rb.yb, T), D), lig, G, nvec);
          occ(rb.m.x >> kSlotShift, a0, b0);
        }
      }
    }
    q = e_run;
    // (a run of one -- nearly every run of a uniform batch -- ends at p + 1, whose id is already in a register)
    const bool ends = q == n || (q == p + 1 ? nxt.id : (uint32_t)sorted_ids[q]) != id;
    if (head && ends) {
      step_apply2<VEC, NCH>(tt, code, own, a, g, D, lig, G, nvec, lr, eps);
    } else {
      const int64_t slot = 2 * (p / kTripChunk) + (head ? 1 : 0);
      row_store(g, chunk_rows + slot * D, lig, G, nvec);
      if (lig == 0) *parked = 1;  // (every writer stores the same value)
    }
  }
  const double t = block_sum_d(acc_loss, sm);
  if (threadIdx.x == 0) {
    double total;
    unsigned flags;
    if (fixed_sum_arrive(loss_acc, t, frac, gridDim.x, &total, &flags))
      loss[0] = (flags & 1u) ? __builtin_nanf("") : ((flags & 2u) ? __builtin_inff() : (float)(total * inv_batch_size));
  }
}

template <int VEC, int NCH>
__global__ __launch_bounds__(kBlock) void triplet_step_long_kernel(TwoTowers tt, int D, int G,
                                                                  const int32_t* __restrict__ sorted_ids, int64_t n,
                                                                  float lr, float eps,
                                                                  const float* __restrict__ chunk_rows,
                                                                  const int* __restrict__ parked) {
  // no run of the batch outgrew its head chunk (every batch of uniform ids): nothing to combine -- one load instead of
  // the screening of the chunk boundaries (three dependent loads and three barriers per workgroup)
  if (*parked == 0) return;
  __shared__ float red[kBlock * VEC * NCH];
  constexpr int kPass = 4;
  __shared__ long long s_long[kPass];
  __shared__ int s_nlong, s_hoff;
  const int tid = threadIdx.x, lig = tid & (G - 1), gidx = tid / G, NG = kBlock / G;
  const int nvec = D / VEC;
  auto id_at = [&](int64_t pos) { return (uint32_t)sorted_ids[pos]; };
  const int64_t nbound = (n - 1) / kTripChunk;
  for (int64_t b0 = (int64_t)blockIdx.x * kPass; b0 < nbound; b0 += (int64_t)gridDim.x * kPass) {
    __syncthreads();
    if (tid == 0) s_nlong = 0;
    __syncthreads();
    {
      const int64_t Bd = (b0 + tid + 1) * kTripChunk;
      if (tid < kPass && b0 + tid < nbound) {
        const uint32_t id_b = id_at(Bd);
        const bool first = Bd < 2 * kTripChunk || id_at(Bd - 2 * kTripChunk) != id_b;
        if (id_at(Bd - kTripChunk) == id_b && first) s_long[atomicAdd(&s_nlong, 1)] = Bd;
      }
    }
    __syncthreads();
    const int nlong = s_nlong;
    for (int li = 0; li < nlong; ++li) {
      const int64_t nxt = s_long[li];
      const uint32_t id = id_at(nxt);
      const int64_t win = max<int64_t>(nxt - 2 * kTripChunk + 1, 0);
      if (tid < 64) {
        const int64_t pos = win + tid;
        const bool is_head = pos <= nxt - kTripChunk && id_at(pos) == id && (pos == 0 || id_at(pos - 1) != id);
        const unsigned long long m = __ballot(is_head);
        if (tid == 0) s_hoff = __ffsll((long long)m) - 1;
      }
      __syncthreads();
      const int64_t h = win + s_hoff;
      int64_t K = 0;
      for (int64_t k0 = 0;; k0 += kBlock) {
        const int64_t pos = nxt + (k0 + tid) * kTripChunk;
        const int cnt = __syncthreads_count(pos < n && id_at(pos) == id);
        K += cnt;
        if (cnt < kBlock) break;
      }
      auto part_row = [&](int64_t i) {
        return (i == 0 ? 2 * (h / kTripChunk) + 1 : 2 * ((nxt + (i - 1) * kTripChunk) / kTripChunk)) * (int64_t)D;
      };
      RowRegs<VEC, NCH> acc;
      row_zero(acc);
      int64_t i = gidx;
      for (; i + 3 * NG <= K; i += 4 * NG) {
        RowRegs<VEC, NCH> t0, t1, t2, t3;
        row_load(t0, chunk_rows + part_row(i), lig, G, nvec);
        row_load(t1, chunk_rows + part_row(i + NG), lig, G, nvec);
        row_load(t2, chunk_rows + part_row(i + 2 * NG), lig, G, nvec);
        row_load(t3, chunk_rows + part_row(i + 3 * NG), lig, G, nvec);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            acc.v[k][e] = (((acc.v[k][e] + t0.v[k][e]) + t1.v[k][e]) + t2.v[k][e]) + t3.v[k][e];
      }
      for (; i <= K; i += NG) {
        RowRegs<VEC, NCH> t;
        row_load(t, chunk_rows + part_row(i), lig, G, nvec);
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
        // (nobody has rewritten this row during the step: its head parked its partial instead)
        const uint32_t code = code_of(id, loc_byte(tt, id), tt.stamp);
        RowRegs<VEC, NCH> own, a;
        row_load(own, tower_row(tt, code, D), lig, G, nvec);
        row_load(a, tower_acc(tt, id, D), lig, G, nvec);
        step_apply2<VEC, NCH>(tt, code, own, a, acc, D, lig, G, nvec, lr, eps);
      }
      __syncthreads();
    }
  }
}

// clear the stamps of a location array (bit 0 stays): 16 bytes per thread
__global__ __launch_bounds__(kBlock) void rows_restamp_kernel(uint8_t* __restrict__ loc, int64_t V) {
  const int64_t nv = V / 16;
  uint4* v = reinterpret_cast<uint4*>(loc);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) {
    uint4 x = v[i];
    x.x &= 0x01010101u;
    x.y &= 0x01010101u;
    x.z &= 0x01010101u;
    x.w &= 0x01010101u;
    v[i] = x;
  }
  if (blockIdx.x == 0)
    for (int64_t i = nv * 16 + threadIdx.x; i < V; i += kBlock) loc[i] &= 1;
}

// =====================================================================================================================
// DIRECT mode (round 4, the default): the step walks the TRIPLETS, not the sorted occurrences, and updates rows in place.
//
// The stamped kernel above reads, per occurrence, its own row, its accumulator and BOTH partner rows: 3 x (4 reads + 2
// writes) = 18 row transfers per triplet (measured 9.8 KB per triplet at D = 128), although a triplet only has three rows.
// Here one row group owns one triplet: its three rows are read ONCE, the three gradient rows are formed on chip, and
//   * a row that occurs once in the batch (nearly every row of a uniform batch at the reference's sizes) is stepped on the
//     spot: accumulator read, Adagrad, row + accumulator written IN PLACE -- 4 transfers per row, the fused minimum.
//     Nobody else reads that row during the step (only its own triplet holds it), so there is nothing to double-buffer:
//     no shadow table, no location bytes, no stamp -- and one dependent round trip fewer per row than the stamped walk
//     (ids -> rows instead of record -> location bytes -> rows);
//   * a row with 2 .. kDirectMaxRun occurrences is stepped by whichever of its triplets finishes LAST: every occurrence
//     writes its gradient row to the side buffer at its sorted position (written through to the memory side), then counts
//     itself in at the run's head position with one atomic; the arrival that completes the run has thereby seen every
//     other occurrence finish its READ of the row, sums the run's gradient rows in sorted order (the association of the
//     segment kernels: left to right from zero) and does the row's one read-modify-write;
//   * longer runs (hot rows: Zipfian ids) leave their gradient rows in the side buffer for triplet_direct_long_kernel --
//     launched only when the plan found such a run (the long-run hint, as before).
// The plan (from the sorted ids alone, made ahead for eight batches): per occurrence its sorted position, "duplicated?",
// "long?", the run's head and length; the counters zeroed; the list of long runs.
// Same element arithmetic (trip_grad, adagrad_elem) as esr_triplet_fwd_bwd + esr_sparse_adagrad_scatter_multi; rows with
// up to kDirectMaxRun occurrences get the same bits as that path, longer runs a different (fixed) association.
// =====================================================================================================================
constexpr int kDirectMaxRun = 8;
constexpr uint32_t kDupBit = 0x80000000u, kLongBit = 0x40000000u, kPosMask = 0x3FFFFFFFu;
constexpr uint32_t kHeadMask = 0x1FFFFFFFu;  // code.y = head position | (run length - 1) << 29

__global__ __launch_bounds__(kBlock) void triplet_direct_plan_kernel(const int32_t* __restrict__ sorted_all,
                                                                    const int32_t* __restrict__ perm_all, int64_t B,
                                                                    char* __restrict__ plans, size_t plan_stride,
                                                                    int* __restrict__ hints, int gen) {
  const int list = blockIdx.y;
  const int64_t n = 3 * B;
  const int32_t* __restrict__ sorted = sorted_all + (int64_t)list * n;
  const int32_t* __restrict__ perm = perm_all + (int64_t)list * n;
  char* base = plans + (size_t)list * plan_stride;
  int* flags = (int*)base;
  unsigned long long* loss_acc = (unsigned long long*)(base + 256);
  char* q = base + 256 + align_up(sizeof(unsigned long long) * kFixAccWords, 256);
  uint2* code = (uint2*)q;
  q += align_up(sizeof(uint2) * (size_t)n, 256);
  uint32_t* cnt = (uint32_t*)q;
  q += align_up(sizeof(uint32_t) * (size_t)n, 256);
  int32_t* long_heads = (int32_t*)q;
  if (blockIdx.x == 0) {
    // (flags[2..3], the generation-tagged long-run count, is claimed by whichever workgroup finds the first long run)
    if (threadIdx.x < 64 && (threadIdx.x >> 1) != 1) flags[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < kFixAccWords; i += kBlock) loss_acc[i] = 0ull;
  }
  bool any_long = false;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
    const int32_t id = sorted[p];
    int lb = 0, lf = 0;
    while (lb < kDirectMaxRun && p - lb - 1 >= 0 && sorted[p - lb - 1] == id) ++lb;
    while (lf < kDirectMaxRun && p + lf + 1 < n && sorted[p + lf + 1] == id) ++lf;
    const int len = lb + lf + 1;
    const bool is_long = lb == kDirectMaxRun || lf == kDirectMaxRun || len > kDirectMaxRun;
    uint2 c;
    c.x = (uint32_t)p | (len > 1 ? kDupBit : 0u) | (is_long ? kLongBit : 0u);
    c.y = is_long ? 0u : ((uint32_t)(p - lb) | ((uint32_t)(len - 1) << 29));
    code[perm[p]] = c;
    cnt[p] = 0u;
    if (is_long) {
      any_long = true;
      if (lb == 0) {  // the run's first position joins the (order-free) list of long runs
        // the 8-byte word at flags[2..3] = generation << 32 | count, generation = this plan call's FULL 32-bit `gen`: a
        // count left by an earlier plan in this buffer (another generation) is replaced, not added to -- no fill launch
        // in front of the plan kernel, and no generation of a run (2^32 plan calls) can be mistaken for another (a
        // 12-bit tag, rounds 3-4, came round again after 4096 groups)
        unsigned long long* word = reinterpret_cast<unsigned long long*>(flags + 2);
        const unsigned long long tag = (unsigned long long)(unsigned)gen << 32;
        unsigned slot;
        for (;;) {
          const unsigned long long old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((old >> 32) != (unsigned long long)(unsigned)gen) {
            unsigned long long expect = old;
            if (__hip_atomic_compare_exchange_strong(word, &expect, tag | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
              slot = 0;
              break;
            }
          } else {  // (once the generation is this call's, nobody replaces it)
            slot = (unsigned)__hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        if ((int64_t)slot < n / 9 + 1) long_heads[slot] = (int32_t)p;  // (always, for a buffer zeroed before its first plan)
      }
    }
  }
  if (hints && __any(any_long) && (threadIdx.x & 63) == 0) hints[list] = gen;  // (every writer stores the same value)
}

// T: the table's element type -- float, or uint16_t for bf16 rows (BASELINE config 4's dtype: the register image stays
// f32, loads widen, stores round to nearest even; accumulators are fp32 either way).  A triplet then moves 3 x (2 + 2 + 4
// + 4) D = 5 376 bytes at D = 128 instead of 7 680.
template <class T>
struct DirectTowers {
  T* s;         // scene tower, updated in place           virtual rows [0, Vs)
  T* p;         // product tower                           virtual rows [Vs, Vs + Vp)
  float* sacc;
  float* pacc;
};

// gradient rows written through to the memory side / read from it (another workgroup of the SAME launch is the reader):
// 8-byte agent-scope accesses, two per float4 chunk
template <int VEC, int NCH>
__device__ __forceinline__ void side_store(const RowRegs<VEC, NCH>& r, float* __restrict__ dst, int lig, int G, int nvec) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lig + k * G;
    if (c < nvec) {
      if constexpr (VEC >= 4) {  // (4 or 8 elements per chunk: pairs as 8-byte words)
        unsigned long long* d = reinterpret_cast<unsigned long long*>(dst + VEC * c);
#pragma unroll
        for (int e = 0; e < VEC; e += 2)
          __hip_atomic_store(d + e / 2, (unsigned long long)__float_as_uint(r.v[k][e]) |
                                            ((unsigned long long)__float_as_uint(r.v[k][e + 1]) << 32),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_store(reinterpret_cast<unsigned*>(dst + c), __float_as_uint(r.v[k][0]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
template <int VEC, int NCH>
__device__ __forceinline__ void side_load(RowRegs<VEC, NCH>& r, const float* __restrict__ src, int lig, int G, int nvec) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lig + k * G;
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.v[k][e] = 0.f;
    if (c < nvec) {
      if constexpr (VEC >= 4) {
        const unsigned long long* d = reinterpret_cast<const unsigned long long*>(src + VEC * c);
        unsigned long long w[VEC / 2];
#pragma unroll
        for (int e = 0; e < VEC / 2; ++e) w[e] = __hip_atomic_load(d + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int e = 0; e < VEC / 2; ++e) {
          r.v[k][2 * e] = __uint_as_float((uint32_t)w[e]);
          r.v[k][2 * e + 1] = __uint_as_float((uint32_t)(w[e] >> 32));
        }
      } else {
        r.v[k][0] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(src + c), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT));
      }
    }
  }
}

template <int VEC, int NCH, class T>
__global__ __launch_bounds__(kBlock) void triplet_direct_kernel(DirectTowers<T> tt, int D, int G,
                                                               const int32_t* __restrict__ scene_ids,
                                                               const int32_t* __restrict__ pos_ids,
                                                               const int32_t* __restrict__ neg_ids,
                                                               const uint2* __restrict__ code, int64_t B, float lam,
                                                               float inv_bs, int with_reg, float lr, float eps,
                                                               float* __restrict__ side, uint32_t* __restrict__ cnt,
                                                               int* __restrict__ parked,
                                                               double* __restrict__ loss_part) {
  __shared__ double sm[8];
  const int lig = threadIdx.x & (G - 1);
  const int lane = threadIdx.x & 63;
  const int glane0 = lane & ~(G - 1);  // first lane of this row group inside the wave
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int nvec = D / VEC;
  const int64_t per = (B + ngroups - 1) / ngroups;
  const int64_t b_begin = min(B, group * per), b_end = min(B, (group + 1) * per);
  double acc_loss = 0.0;
  // ids and codes of the triplet AFTER the current one are requested while the current one's rows travel
  int32_t n_sid = 0, n_pid = 0, n_nid = 0;
  uint2 n_cs = make_uint2(0, 0), n_cp = n_cs, n_cn = n_cs;
  auto fetch = [&](int64_t b) {
    n_sid = scene_ids[b];
    n_pid = pos_ids[b];
    n_nid = neg_ids[b];
    n_cs = code[b];
    n_cp = code[B + b];
    n_cn = code[2 * B + b];
  };
  if (b_begin < b_end) fetch(b_begin);
  for (int64_t b = b_begin; b < b_end; ++b) {
    const int64_t sid = n_sid, pid = n_pid, nid = n_nid;
    const uint2 cs = n_cs, cp = n_cp, cn = n_cn;
    T* const srow = tt.s + sid * D;
    T* const prow = tt.p + pid * D;
    T* const nrow = tt.p + nid * D;
    RowRegs<VEC, NCH> S, P, N, aS, aP, aN;
    row_load(S, srow, lig, G, nvec);
    row_load(P, prow, lig, G, nvec);
    row_load(N, nrow, lig, G, nvec);
    // accumulators of the rows this group steps itself (a duplicated row's is read by the arrival that completes its run)
    if (!(cs.x & kDupBit)) row_load(aS, tt.sacc + sid * D, lig, G, nvec);
    if (!(cp.x & kDupBit)) row_load(aP, tt.pacc + pid * D, lig, G, nvec);
    if (!(cn.x & kDupBit)) row_load(aN, tt.pacc + nid * D, lig, G, nvec);
    if (b + 1 < b_end) fetch(b + 1);
    // scores and norms: the expressions (and roundings) of triplet_step_kernel's occ()
    const float d_sp = group_sum(row_dot_partial(S, P), G);
    const float d_sn = group_sum(row_dot_partial(S, N), G);
    float ns = 0.f, np_ = 0.f, nn = 0.f, c_s = 0.f, c_p = 0.f, c_n = 0.f;
    if (with_reg) {
      ns = sqrtf(group_sum(row_dot_partial(S, S), G));
      np_ = sqrtf(group_sum(row_dot_partial(P, P), G));
      nn = sqrtf(group_sum(row_dot_partial(N, N), G));
      c_s = ns > 1.f ? lam / ns : 0.f;
      c_p = np_ > 1.f ? lam / np_ : 0.f;
      c_n = nn > 1.f ? lam / nn : 0.f;
    }
    const float margin = 1.0f + d_sn - d_sp;
    const float mk = margin > 0.f ? 1.f : 0.f;
    RowRegs<VEC, NCH> gS, gP, gN;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        // (0 + g: the segment kernels start every row sum from zero)
        gS.v[k][e] = __fadd_rn(0.f, trip_grad(mk, __fsub_rn(N.v[k][e], P.v[k][e]), c_s, S.v[k][e], inv_bs));
        gP.v[k][e] = __fadd_rn(0.f, trip_grad(-mk, S.v[k][e], c_p, P.v[k][e], inv_bs));
        gN.v[k][e] = __fadd_rn(0.f, trip_grad(mk, S.v[k][e], c_n, N.v[k][e], inv_bs));
      }
    {
      float loss_b = fmaxf(margin, 0.f);
      if (with_reg) loss_b += lam * (fmaxf(ns - 1.f, 0.f) + fmaxf(np_ - 1.f, 0.f) + fmaxf(nn - 1.f, 0.f));
      if (lig == 0) acc_loss += (double)loss_b;
    }
    // the three occurrences.  Unique rows are stepped on the spot.  Duplicated rows: all gradient rows of the triplet go
    // to the side buffer first, ONE wait covers them, the (up to three) arrivals are counted by atomics issued together
    // -- two dependent round trips per triplet however many of its rows are duplicated -- and only then does an arrival
    // that completed its run sum it.
    auto step_here = [&](T* row, float* accrow, RowRegs<VEC, NCH>& own, RowRegs<VEC, NCH>& a,
                         const RowRegs<VEC, NCH>& g) {
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) adagrad_elem(own.v[k][e], a.v[k][e], g.v[k][e], lr, eps);
      row_store(a, accrow, lig, G, nvec);
      row_store(own, row, lig, G, nvec);
    };
    auto park = [&](uint2 c, const RowRegs<VEC, NCH>& g) {  // a duplicated row's gradient row -> side buffer
      const int64_t pos = c.x & kPosMask;
      if (c.x & kLongBit) {
        row_store(g, side + pos * D, lig, G, nvec);
        if (lig == 0) *parked = 1;  // (every writer stores the same value)
      } else {
        side_store(g, side + pos * D, lig, G, nvec);
      }
    };
    const bool dS = (cs.x & kDupBit) != 0, dP = (cp.x & kDupBit) != 0, dN = (cn.x & kDupBit) != 0;
    if (!dS) step_here(srow, tt.sacc + sid * D, S, aS, gS);
    if (!dP) step_here(prow, tt.pacc + pid * D, P, aP, gP);
    if (!dN) step_here(nrow, tt.pacc + nid * D, N, aN, gN);
    if (dS | dP | dN) {
      if (dS) park(cs, gS);
      if (dP) park(cp, gP);
      if (dN) park(cn, gN);
      // this group's gradient rows are at the memory side before it is counted in (the wave waits for its lanes' stores)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const bool sS = dS && !(cs.x & kLongBit), sP = dP && !(cp.x & kLongBit), sN = dN && !(cn.x & kLongBit);
      uint32_t oS = 0xFFFFFFFFu, oP = 0xFFFFFFFFu, oN = 0xFFFFFFFFu;
      if (lig == 0) {
        if (sS) oS = atomicAdd(cnt + (cs.y & kHeadMask), 1u);
        if (sP) oP = atomicAdd(cnt + (cp.y & kHeadMask), 1u);
        if (sN) oN = atomicAdd(cnt + (cn.y & kHeadMask), 1u);
      }
      oS = (uint32_t)__shfl((int)oS, glane0, 64);
      oP = (uint32_t)__shfl((int)oP, glane0, 64);
      oN = (uint32_t)__shfl((int)oN, glane0, 64);
      // an arrival that completed its run: every other occurrence has read the row and left its gradient row -- sum them
      // in sorted order and do the row's one read-modify-write
      auto complete = [&](uint2 c, T* row, float* accrow, RowRegs<VEC, NCH>& own, RowRegs<VEC, NCH>& a) {
        const int64_t head = c.y & kHeadMask;
        const uint32_t len = (c.y >> 29) + 1u;
        RowRegs<VEC, NCH> sum;
        row_zero(sum);
        row_load(a, accrow, lig, G, nvec);
        for (uint32_t j = 0; j < len; ++j) {
          RowRegs<VEC, NCH> t;
          side_load(t, side + (head + j) * D, lig, G, nvec);
#pragma unroll
          for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum.v[k][e] = __fadd_rn(sum.v[k][e], t.v[k][e]);
        }
        // every arrival of the run has been counted (this one completed it): the counter goes back to zero, so the plan
        // can feed another step (the same batch stepped again: ops.triplet_train_step(plan=...))
        if (lig == 0) cnt[head] = 0u;
        step_here(row, accrow, own, a, sum);
      };
      if (sS && oS == (cs.y >> 29)) complete(cs, srow, tt.sacc + sid * D, S, aS);
      if (sP && oP == (cp.y >> 29)) complete(cp, prow, tt.pacc + pid * D, P, aP);
      if (sN && oN == (cn.y >> 29)) complete(cn, nrow, tt.pacc + nid * D, N, aN);
    }
  }
  // the workgroup's loss partial, fire and forget: triplet_direct_loss_kernel adds the partials of a step in a fixed order
  // (the counted integer reduction of the stamped kernel put two dependent atomic round trips at the tail of every
  // workgroup -- ~2 us of an 18 us launch at B = 8192)
  const double t = block_sum_d(acc_loss, sm);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = t;
}

// losses[b] = (sum of step b's workgroup partials, fixed order) / batch_size: one workgroup per step of a group
__global__ __launch_bounds__(kBlock) void triplet_direct_loss_kernel(const char* __restrict__ plans, size_t stride,
                                                                    size_t part_off, int nparts, double inv_batch_size,
                                                                    float* __restrict__ losses) {
  __shared__ double sm[8];
  const double* part = reinterpret_cast<const double*>(plans + (size_t)blockIdx.x * stride + part_off);
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) a += part[i];
  const double t = block_sum_d(a, sm);
  if (threadIdx.x == 0) losses[blockIdx.x] = (float)(t * inv_batch_size);
}

// the runs longer than kDirectMaxRun: one workgroup per run (the plan's list).  The association of the segment kernels
// (esr_optim.hip), cut point for cut point: partial 0 = the head chunk [head, the chunk boundary after the next one),
// partial i >= 1 = the 32 positions from boundary nxt + 32 (i - 1), each summed left to right from zero; row group gi
// adds partials gi, gi + NG, ... in order; the groups' sums are combined in group order.  With the same lanes per row
// (row_geom) a hot row gets the bits esr_sparse_adagrad_scatter_multi gives it.
template <int VEC, int NCH, class T>
__global__ __launch_bounds__(kBlock) void triplet_direct_long_kernel(DirectTowers<T> tt, int D, int G, int64_t Vs,
                                                                    const int32_t* __restrict__ sorted_ids, int64_t n,
                                                                    float lr, float eps, const float* __restrict__ side,
                                                                    const int* __restrict__ flags,
                                                                    const int32_t* __restrict__ long_heads) {
  if (flags[0] == 0) return;  // nothing was parked
  __shared__ float red[kBlock * VEC * NCH];
  const int tid = threadIdx.x, lig = tid & (G - 1), gidx = tid / G, NG = kBlock / G;
  const int nvec = D / VEC;
  // (generation << 32 | count in flags[2..3]: something was parked, so the count is this plan's)
  const int nlong = (int)min<unsigned long long>(*reinterpret_cast<const unsigned long long*>(flags + 2) & 0xFFFFFFFFull,
                                                 (unsigned long long)(n / 9 + 1));
  for (int li = blockIdx.x; li < nlong; li += gridDim.x) {
    const int64_t head = long_heads[li];
    const uint32_t id = (uint32_t)sorted_ids[head];
    int64_t len = 0;  // run length
    for (int64_t k0 = 0;; k0 += kBlock) {
      const int64_t pos = head + k0 + tid;
      const int c = __syncthreads_count(pos < n && (uint32_t)sorted_ids[pos] == id);
      len += c;
      if (c < kBlock) break;
    }
    const int64_t end = head + len;
    const int64_t nxt = min(end, ((head + 2 * kStepChunk - 1) / kStepChunk) * kStepChunk);  // end of the head chunk
    const int64_t K = (end - nxt + kStepChunk - 1) / kStepChunk;                          // continuation chunks
    RowRegs<VEC, NCH> acc;
    row_zero(acc);
    for (int64_t i = gidx; i <= K; i += NG) {
      const int64_t q0 = i == 0 ? head : nxt + (i - 1) * kStepChunk;
      const int64_t q1 = i == 0 ? nxt : min(end, q0 + kStepChunk);
      RowRegs<VEC, NCH> c;
      row_zero(c);
      int64_t q = q0;
      for (; q + 4 <= q1; q += 4) {  // four rows in flight, added in order
        RowRegs<VEC, NCH> t0, t1, t2, t3;
        row_load(t0, side + q * D, lig, G, nvec);
        row_load(t1, side + (q + 1) * D, lig, G, nvec);
        row_load(t2, side + (q + 2) * D, lig, G, nvec);
        row_load(t3, side + (q + 3) * D, lig, G, nvec);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            c.v[k][e] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(c.v[k][e], t0.v[k][e]), t1.v[k][e]), t2.v[k][e]), t3.v[k][e]);
      }
      for (; q < q1; ++q) {
        RowRegs<VEC, NCH> t;
        row_load(t, side + q * D, lig, G, nvec);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e) c.v[k][e] = __fadd_rn(c.v[k][e], t.v[k][e]);
      }
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc.v[k][e] += c.v[k][e];
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
      const bool prod = (int64_t)id >= Vs;
      const int64_t r = prod ? (int64_t)id - Vs : (int64_t)id;
      T* row = (prod ? tt.p : tt.s) + r * D;
      float* accrow = (prod ? tt.pacc : tt.sacc) + r * D;
      RowRegs<VEC, NCH> own, a;
      row_load(own, row, lig, G, nvec);
      row_load(a, accrow, lig, G, nvec);
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) adagrad_elem(own.v[k][e], a.v[k][e], acc.v[k][e], lr, eps);
      row_store(a, accrow, lig, G, nvec);
      row_store(own, row, lig, G, nvec);
    }
    __syncthreads();
  }
}

// ESR_TRIPLET_STEP=stamped keeps the double-buffered walk over the sorted occurrences (rounds 2-3); default: direct
static bool trip_direct_mode() {
  const char* e = getenv("ESR_TRIPLET_STEP");
  return !(e && e[0] == 's');
}

static int launch_trip_plan(const int32_t* const* ids, int nbatch, const int32_t* sorted_ids, const int32_t* perm,
                            int64_t B, int64_t Vs, char* plans, size_t stride, int* hints, int gen, hipStream_t st) {
  PlanBatch pb;
  for (int b = 0; b < kMaxPlanBatch; ++b) {
    const int s = b < nbatch ? b : 0;
    pb.scene[b] = ids[3 * s];
    pb.pos[b] = ids[3 * s + 1];
    pb.neg[b] = ids[3 * s + 2];
  }
  const int64_t n = 3 * B;
  const int gx = (int)std::min<int64_t>(kMaxGrid, cdiv(n, kBlock));
  if (trip_direct_mode()) {
    // The long-run counters carry the plan call's full 32-bit generation (see the kernel): nothing to clear in front of
    // it.  gen == 0 (a plan made in line, inside a step call, in scratch memory that may hold anything -- also a word of
    // generation 0): the word is cleared first.
    // (same box, alternating runs at B = 8192: 23.56-23.62 us per step without the fill, 23.81-23.93 with it)
    if (gen == 0 &&
        hipMemset2DAsync(plans + 2 * sizeof(int), stride ? stride : 2 * sizeof(int), 0, 2 * sizeof(int), (size_t)nbatch, st) != hipSuccess)
      return ESR_ELAUNCH;
    hipLaunchKernelGGL(triplet_direct_plan_kernel, dim3(gx, nbatch), dim3(kBlock), 0, st, sorted_ids, perm, B, plans, stride,
                       hints, gen);
    return ESR_OK;
  }
  hipLaunchKernelGGL(triplet_plan_kernel, dim3(gx, nbatch), dim3(kBlock), 0, st, pb, sorted_ids, perm, B, Vs, plans, stride,
                     hints, gen);
  return ESR_OK;
}

// fraction bits of the fixed-point loss accumulator: the SUM over the batch must stay below 2^(52 - frac); with
// frac = 41 - ceil(log2 B) a mean loss below 2048 fits for every B (beyond: the loss comes out +inf)
static int loss_frac_bits(int64_t B) {
  int lg = 0;
  while (((int64_t)1 << lg) < B) ++lg;
  return std::max(8, std::min(36, 41 - lg));
}

}  // namespace esr

using namespace esr;

extern "C" {

size_t esr_triplet_step_workspace_bytes(int64_t B, int D) {
  if (B <= 0 || D <= 0) return 0;
  return trip_ws_layout(B, D, nullptr, nullptr);
}

size_t esr_triplet_plan_bytes(int64_t B) {
  if (B <= 0) return 0;
  return trip_plan_layout(B, nullptr, nullptr);
}

int esr_triplet_plan(const int32_t* const* ids, int nbatch, int64_t B, int64_t Vs, const int32_t* sorted_ids,
                     const int32_t* perm, void* plans, int32_t* hints, int32_t gen, esr_stream_t stream) {
  TraceScope trace_scope_("esr_triplet_plan");
  ESR_REQUIRE(nbatch >= 1 && nbatch <= kMaxPlanBatch && B > 0 && Vs > 0 && 3 * B < ((int64_t)1 << 31),
              "esr_triplet_plan: nbatch=%d not in [1, %d] or bad sizes B=%lld Vs=%lld", nbatch, kMaxPlanBatch,
              (long long)B, (long long)Vs);
  ESR_REQUIRE(ids && sorted_ids && perm && plans && !((uintptr_t)plans & 255), "esr_triplet_plan: null or misaligned pointer");
  for (int i = 0; i < 3 * nbatch; ++i) ESR_REQUIRE(ids[i], "esr_triplet_plan: null id list %d", i);
  launch_trip_plan(ids, nbatch, sorted_ids, perm, B, Vs, (char*)plans, esr_triplet_plan_bytes(B), hints, gen,
                   as_stream(stream));
  return check_launch("esr_triplet_plan");
}

int esr_rows_restamp(uint8_t* loc, int64_t V, esr_stream_t stream) {
  ESR_REQUIRE(V >= 0, "esr_rows_restamp: V=%lld", (long long)V);
  if (V == 0) return ESR_OK;
  ESR_REQUIRE(loc && !((uintptr_t)loc & 15), "esr_rows_restamp: null or misaligned pointer");
  const int grid = (int)std::min<int64_t>(kMaxGrid, std::max<int64_t>(1, cdiv(V / 16, kBlock)));
  hipLaunchKernelGGL(rows_restamp_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), loc, V);
  return check_launch("esr_rows_restamp");
}

}  // extern "C"

// one step's launches (arguments validated by the callers)
// what a direct step leaves for its caller: where its loss partials are (triplet_direct_loss_kernel turns them into losses)
struct DirectLoss {
  int nparts = 0;
  const char* plan = nullptr;
  size_t part_off = 0;
};
static void launch_direct_losses(const DirectLoss& d, size_t stride, int nb, float batch_size, float* losses,
                                 hipStream_t st) {
  ESR_KT("triplet_direct_loss_kernel", st,
         hipLaunchKernelGGL(triplet_direct_loss_kernel, dim3(nb), dim3(kBlock), 0, st, d.plan, stride, d.part_off, d.nparts,
                            1.0 / (double)batch_size, losses));
}

static int launch_trip_step(const TwoTowers& tt, int dtype, int D, const RowGeom& g, const int32_t* scene_ids, const int32_t* pos_ids,
                            const int32_t* neg_ids, int64_t B, float regularization, float batch_size, float lr, float eps,
                            const int32_t* sorted, const int32_t* perm, void* plan, int long_runs, float* loss,
                            const TripWs& ws, hipStream_t st, DirectLoss* direct = nullptr) {
  const int64_t n = 3 * B;
  if (!plan) {  // no plan made ahead: make it here (and nobody told us whether a run is long: screen for it)
    const int32_t* ids3[3] = {scene_ids, pos_ids, neg_ids};
    launch_trip_plan(ids3, 1, sorted, perm, B, tt.Vs, ws.plan, 0, nullptr, 0, st);
    plan = ws.plan;
    long_runs = -1;
  }
  TripPlan pl;
  trip_plan_layout(B, (char*)plan, &pl);
  int grid = grid_for_groups(n, g.G);
  const int grid2 = (int)std::min<int64_t>(kMaxGrid, cdiv(cdiv(n, kTripChunk), 4));
  const float inv_bs = 1.0f / batch_size;
  if (trip_direct_mode()) {
    // rows are stepped in place in the primary buffers (the second buffers and location bytes are not touched: the
    // caller's rows never leave home)
    // lanes per row: a triplet's five dot products are a small part of its work (the stamped walk had six per occurrence
    // and wanted few lanes); ESR_TRIPLET_DIRECT_LANES=few keeps step_geom_few_lanes
    const char* le = getenv("ESR_TRIPLET_DIRECT_LANES");
    // bf16 rows: 8 elements (16 bytes) per lane when the width allows (row_geom8)
    // -- ESR_BF16_VEC8=1 only: measured slower (136 registers, three waves per SIMD instead of five; the step lives on
    // its occupancy): 0.496 against 0.438 ms per 262 144 triplets
    const char* v8 = getenv("ESR_BF16_VEC8");
    const bool vec8 = v8 && v8[0] == '1' && dtype == ESR_BF16 && D % 8 == 0 && !(((uintptr_t)tt.s0 | (uintptr_t)tt.p0) & 15);
    const RowGeom gd = vec8 ? row_geom8(D) : ((le && le[0] == 'f') ? g : row_geom(D));
    const RowGeom& g = gd;
#define ESR_TRIP_DIRECT_LAUNCH(T, DISPATCH)                                                                            \
    DISPATCH(g, {                                                                                                      \
      const DirectTowers<T> dt{(T*)tt.s0, (T*)tt.p0, tt.sacc, tt.pacc};                                                \
      static const int resident = resident_blocks((const void*)triplet_direct_kernel<VEC, NCH, T>);                    \
      const int gridd = std::min(grid_for_groups(B, g.G), resident);                                                   \
      ESR_KT("triplet_direct_kernel", st, hipLaunchKernelGGL((triplet_direct_kernel<VEC, NCH, T>), dim3(gridd), dim3(kBlock), 0, st, dt, D, g.G, scene_ids, \
                         pos_ids, neg_ids, (const uint2*)pl.meta, B, regularization, inv_bs, 1, lr, eps, ws.chunk_rows, \
                         pl.cnt, pl.flags, pl.loss_part));                                                             \
      if (direct) {  /* the caller adds the partials up (one launch for a whole group of steps) */                     \
        direct->nparts = gridd;                                                                                        \
        direct->plan = (const char*)plan;                                                                              \
        direct->part_off = (size_t)((const char*)pl.loss_part - (const char*)plan);                                    \
      }                                                                                                                \
      if (long_runs != 0)  /* 0 = the caller knows (the plan's hint) that no run is longer than kDirectMaxRun */       \
        ESR_KT("triplet_direct_long_kernel", st, hipLaunchKernelGGL((triplet_direct_long_kernel<VEC, NCH, T>), dim3(256), dim3(kBlock), 0, st, dt, D, g.G, \
                           tt.Vs, sorted, n, lr, eps, (const float*)ws.chunk_rows, (const int*)pl.flags,               \
                           (const int32_t*)pl.long_heads));                                                            \
    })
    if (vec8) {
      ESR_TRIP_DIRECT_LAUNCH(uint16_t, ESR_DISPATCH_ROW8);
    } else if (dtype == ESR_BF16) {
      ESR_TRIP_DIRECT_LAUNCH(uint16_t, ESR_DISPATCH_ROW);
    } else {
      ESR_TRIP_DIRECT_LAUNCH(float, ESR_DISPATCH_ROW);
    }
#undef ESR_TRIP_DIRECT_LAUNCH
    return ESR_OK;
  }
  ESR_DISPATCH_ROW(g, {
    static const int resident = resident_blocks((const void*)triplet_step_kernel<VEC, NCH>);  // (one query per process)
    grid = std::min(grid, resident);
    ESR_KT("triplet_step_kernel", st, hipLaunchKernelGGL((triplet_step_kernel<VEC, NCH>), dim3(grid), dim3(kBlock), 0, st, tt, D, g.G, sorted,
                       (const uint2*)pl.meta, n, regularization, inv_bs, 1, lr, eps, ws.chunk_rows, pl.flags,
                       pl.loss_acc, loss_frac_bits(B), 1.0 / (double)batch_size, loss));
    if (long_runs != 0)  // 0 = the caller knows (esr_triplet_plan's hint) that no run outgrows its head chunk
      ESR_KT("triplet_step_long_kernel", st, hipLaunchKernelGGL((triplet_step_long_kernel<VEC, NCH>), dim3(grid2), dim3(kBlock), 0, st, tt, D, g.G, sorted, n,
                         lr, eps, (const float*)ws.chunk_rows, (const int*)pl.flags));
  });
  return ESR_OK;
}

#define ESR_TRIP_STEP_CHECKS(who)                                                                                      \
  ESR_REQUIRE(B > 0 && D > 0 && Vs > 0 && Vp > 0, who ": bad sizes Vs=%lld Vp=%lld D=%d B=%lld", (long long)Vs,        \
              (long long)Vp, D, (long long)B);                                                                         \
  ESR_REQUIRE(Vs + Vp <= (int64_t)kIdMask, who ": %lld virtual rows exceed 2^30 - 1", (long long)(Vs + Vp));           \
  ESR_REQUIRE(3 * B < ((int64_t)1 << 31), who ": B=%lld too large", (long long)B);                                     \
  ESR_REQUIRE(!trip_direct_mode() || 3 * B < ((int64_t)1 << 29),                                                       \
              who ": B=%lld too large for the direct step's 29-bit run heads (ESR_TRIPLET_STEP=stamped)", (long long)B); \
  ESR_REQUIRE(scene && scene_accum && product && product_accum, who ": null table pointer");                           \
  /* direct mode (the default) steps rows in place: the second buffers and location bytes are not used, may be NULL */ \
  ESR_REQUIRE(trip_direct_mode() || (scene_shadow && scene_loc && product_shadow && product_loc),                      \
              who ": null second buffer / location bytes (ESR_TRIPLET_STEP=stamped needs them)");                      \
  ESR_REQUIRE(trip_direct_mode() || (scene != scene_shadow && product != product_shadow),                              \
              who ": a shadow table must be a second buffer");                                                         \
  ESR_REQUIRE(batch_size != 0.f, who ": batch_size must be non-zero");                                                 \
  ESR_REQUIRE(dtype == ESR_F32 || (dtype == ESR_BF16 && trip_direct_mode()),                                           \
              who ": dtype %d (f32, or bf16 rows in the direct step)", dtype);                                         \
  ESR_REQUIRE(dtype == ESR_F32 || D % 4 != 0 || !(((uintptr_t)scene | (uintptr_t)product) & 7),                        \
              who ": bf16 tables must be 8-byte aligned");                                                             \
  const RowGeom g = step_geom_few_lanes(D);                                                                            \
  ESR_REQUIRE(g.nch <= kMaxChunksPerLane, who ": D=%d not supported", D);                                              \
  if (!workspace || workspace_bytes < esr_triplet_step_workspace_bytes(B, D) || ((uintptr_t)workspace & 15)) {         \
    set_error(who ": workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,                             \
              esr_triplet_step_workspace_bytes(B, D));                                                                 \
    return ESR_EWORKSPACE;                                                                                             \
  }

extern "C" {

int esr_triplet_train_step(void* scene, void* scene_shadow, uint8_t* scene_loc, float* scene_accum, int64_t Vs,
                           void* product, void* product_shadow, uint8_t* product_loc, float* product_accum,
                           int64_t Vp, int dtype, int D, const int32_t* scene_ids, const int32_t* pos_ids,
                           const int32_t* neg_ids, int64_t B, float regularization, float batch_size, float lr,
                           float eps, uint32_t stamp, const int32_t* presorted_ids, const int32_t* presorted_perm,
                           void* plan, int long_runs, float* loss, void* workspace, size_t workspace_bytes,
                           esr_stream_t stream) {
  TraceScope trace_scope_("esr_triplet_train_step");
  ESR_TRIP_STEP_CHECKS("esr_triplet_train_step")
  ESR_REQUIRE(scene_ids && pos_ids && neg_ids && loss, "esr_triplet_train_step: null pointer");
  ESR_REQUIRE(trip_direct_mode() || (stamp >= 1 && stamp <= kStampMax), "esr_triplet_train_step: stamp %u not in [1, %u]",
              stamp, kStampMax);
  ESR_REQUIRE((presorted_ids == nullptr) == (presorted_perm == nullptr),
              "esr_triplet_train_step: presorted_ids and presorted_perm must both be set or both be NULL");
  ESR_REQUIRE(!plan || presorted_ids, "esr_triplet_train_step: a plan goes with the sorted ids it was made from");
  ESR_REQUIRE(!plan || !((uintptr_t)plan & 255), "esr_triplet_train_step: misaligned plan");
  hipStream_t st = as_stream(stream);
  TripWs ws;
  trip_ws_layout(B, D, (char*)workspace, &ws);
  const int32_t* sorted = presorted_ids;
  const int32_t* perm = presorted_perm;
  if (!sorted) {
    const int32_t* segs[3] = {scene_ids, pos_ids, neg_ids};
    const int64_t counts[3] = {B, B, B};
    const int64_t offsets[3] = {0, Vs, Vs};
    if (int rc = esr_segment_sort_ids_multi(segs, counts, offsets, 3, Vs + Vp, ws.sorted_ids, ws.perm, ws.sort_ws,
                                            ws.sort_ws_bytes, stream))
      return rc;
    sorted = ws.sorted_ids;
    perm = ws.perm;
  }
  TwoTowers tt{(float*)scene, (float*)scene_shadow, (float*)product, (float*)product_shadow, scene_loc, product_loc,
               scene_accum, product_accum, Vs, stamp};
  DirectLoss dl;
  launch_trip_step(tt, dtype, D, g, scene_ids, pos_ids, neg_ids, B, regularization, batch_size, lr, eps, sorted, perm, plan,
                   long_runs, loss, ws, st, &dl);
  if (dl.nparts) launch_direct_losses(dl, 0, 1, batch_size, loss, st);
  return check_launch("esr_triplet_train_step");
}

int esr_triplet_train_steps(void* scene, void* scene_shadow, uint8_t* scene_loc, float* scene_accum, int64_t Vs,
                            void* product, void* product_shadow, uint8_t* product_loc, float* product_accum,
                            int64_t Vp, int dtype, int D, int nbatch, const int32_t* const* ids, int64_t B, float regularization,
                            float batch_size, float lr, float eps, uint32_t first_stamp, const int32_t* sorted_ids,
                            const int32_t* perm, void* plans, const int32_t* long_runs, float* losses, void* workspace,
                            size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_triplet_train_steps");
  ESR_TRIP_STEP_CHECKS("esr_triplet_train_steps")
  ESR_REQUIRE(nbatch >= 1 && nbatch <= kMaxPlanBatch && ids && sorted_ids && perm && plans && losses &&
                  !((uintptr_t)plans & 255),
              "esr_triplet_train_steps: nbatch=%d not in [1, %d], or a null / misaligned pointer", nbatch, kMaxPlanBatch);
  ESR_REQUIRE(trip_direct_mode() || (first_stamp >= 1 && first_stamp + (uint32_t)nbatch - 1 <= kStampMax),
              "esr_triplet_train_steps: stamps %u .. %u leave [1, %u]", first_stamp, first_stamp + nbatch - 1, kStampMax);
  for (int i = 0; i < 3 * nbatch; ++i) ESR_REQUIRE(ids[i], "esr_triplet_train_steps: null id list %d", i);
  hipStream_t st = as_stream(stream);
  TripWs ws;
  trip_ws_layout(B, D, (char*)workspace, &ws);
  const size_t stride = esr_triplet_plan_bytes(B);
  DirectLoss first;
  for (int b = 0; b < nbatch; ++b) {
    TwoTowers tt{(float*)scene, (float*)scene_shadow, (float*)product, (float*)product_shadow, scene_loc, product_loc,
                 scene_accum, product_accum, Vs, first_stamp + (uint32_t)b};
    DirectLoss dl;
    launch_trip_step(tt, dtype, D, g, ids[3 * b], ids[3 * b + 1], ids[3 * b + 2], B, regularization, batch_size, lr, eps,
                     sorted_ids + (int64_t)b * 3 * B, perm + (int64_t)b * 3 * B, (char*)plans + (size_t)b * stride,
                     long_runs ? long_runs[b] : -1, losses + b, ws, st, &dl);
    if (b == 0) first = dl;
  }
  // direct mode: the losses of the whole group by ONE launch (plans are `stride` apart, same grid for every step)
  if (first.nparts) launch_direct_losses(first, stride, nbatch, batch_size, losses, st);
  return check_launch("esr_triplet_train_steps");
}

}  // extern "C"
