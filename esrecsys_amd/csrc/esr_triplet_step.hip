// The whole Shop-The-Look train step in one pass over the rows: esr_triplet_train_step.
//
// Reference arithmetic: pinterest/train_shop_the_look.py:93-109 (triplet hinge + norm-excess regulariser,
// value_and_grad, one optimizer update) on the id towers that replace pinterest/models.py:64-70, with the build's
// row-sparse Adagrad.  esr_triplet_fwd_bwd + sort + esr_sparse_adagrad_scatter_multi is six dependent launches that
// write a [3B, D] gradient and read it back; at the reference's batch sizes that chain of launches IS the step time.
// Here the gradient of an occurrence is formed on chip inside the update kernel from the two OTHER rows of its triplet,
// which the double-buffered towers (esr_versioned.h) make safe to read while rows are being rewritten:
//
//   sort      virtual occurrence ids [scene ; Vs + pos ; Vs + neg] -> (sorted, perm)   (ahead, on a second stream, or here)
//   plan      per sorted position: own row code, the two partner row codes, the occurrence's slot
//   update    one row group per sorted position, the head of a run walks it: both partner rows -> pos / neg score,
//             hinge mask, own norm -> this occurrence's gradient row; summed left to right; Adagrad once per distinct
//             row into the other buffer.  The scene occurrence of a triplet also contributes its loss term.
//   long      runs longer than a chunk (hot rows): chunk partials combined in a fixed order; its last workgroup
//             reduces the loss
#include "esr_common.h"
#include "esr_versioned.h"

namespace esr {

constexpr int kTripStepBlocks = kMaxGrid;
// Cut points of long runs: every 8 positions instead of the 32 of the other segment kernels.  An occurrence costs this
// kernel two dependent round trips (plan record -> two partner rows), so a hot row is a LONG sequential walk per chunk:
// with 32-position chunks a Zipf(1) batch of 8192 triplets took 131 us (99 us before this kernel existed); shorter
// chunks spread the walk over four times as many row groups.
constexpr int kTripChunk = 8;

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
};

__device__ __forceinline__ const float* tower_row(const TwoTowers& tt, uint32_t code, int D) {
  const int64_t vid = code & kIdMask;
  const bool prod = vid >= tt.Vs, second = (code & kLocBit) != 0;
  const float* base = prod ? (second ? tt.p1 : tt.p0) : (second ? tt.s1 : tt.s0);
  return base + (prod ? vid - tt.Vs : vid) * D;
}

struct TripWs {
  int32_t* sorted_ids;  // [n]
  int32_t* perm;        // [n]
  uint32_t* own_code;   // [n]
  uint4* meta;          // [n]  {slot, partner a code, partner b code, triplet index}
  double* loss_part;    // [kTripStepBlocks]
  int* long_flag;       // [1]  set by the update kernel when it parks a chunk partial: the batch has a long run
  float* chunk_rows;    // [2 * ceil(n / 32)][D]
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
  w.own_code = (uint32_t*)take(sizeof(uint32_t) * (size_t)n);
  w.meta = (uint4*)take(sizeof(uint4) * (size_t)n);
  w.loss_part = (double*)take(sizeof(double) * kTripStepBlocks);
  w.long_flag = (int*)take(sizeof(int));
  w.chunk_rows = (float*)take(sizeof(float) * 2 * (size_t)cdiv(n, kTripChunk) * (size_t)D);
  w.sort_ws_bytes = esr_segment_sort_workspace_bytes(n);
  w.sort_ws = take(w.sort_ws_bytes);
  if (ws) *ws = w;
  return off;
}

// plan: one thread per sorted position.  Occurrence o = perm[p]: slot = o / B (0 scene, 1 pos, 2 neg), triplet b = o % B.
// Partners: scene -> (pos, neg); pos -> (scene, neg); neg -> (scene, pos).
__global__ __launch_bounds__(kBlock) void triplet_plan_kernel(const int32_t* __restrict__ perm,
                                                             const int32_t* __restrict__ scene_ids,
                                                             const int32_t* __restrict__ pos_ids,
                                                             const int32_t* __restrict__ neg_ids,
                                                             const uint8_t* __restrict__ sloc,
                                                             const uint8_t* __restrict__ ploc, int64_t B, int64_t Vs,
                                                             uint32_t* __restrict__ own_code, uint4* __restrict__ meta,
                                                             int* __restrict__ long_flag) {
  const int64_t n = 3 * B;
  if (blockIdx.x == 0 && threadIdx.x == 0) *long_flag = 0;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
    const int64_t o = perm[p];
    const int slot = o >= 2 * B ? 2 : (o >= B ? 1 : 0);
    const int64_t b = o - slot * B;
    const int32_t sid = scene_ids[b], pid = pos_ids[b], nid = neg_ids[b];
    const uint32_t sc = (uint32_t)sid | (sloc[sid] ? kLocBit : 0u);
    const uint32_t pc = (uint32_t)(Vs + pid) | (ploc[pid] ? kLocBit : 0u);
    const uint32_t nc = (uint32_t)(Vs + nid) | (ploc[nid] ? kLocBit : 0u);
    own_code[p] = slot == 0 ? sc : (slot == 1 ? pc : nc);
    meta[p] = make_uint4((uint32_t)slot, slot == 0 ? pc : sc, slot == 2 ? pc : nc, (uint32_t)b);
  }
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
  if (lig == 0) (prod ? tt.ploc : tt.sloc)[id] = second ? 0 : 1;
}

template <int VEC, int NCH>
__global__ __launch_bounds__(kBlock) void triplet_step_kernel(TwoTowers tt, int D, int G,
                                                             const uint32_t* __restrict__ own_code,
                                                             const uint4* __restrict__ meta, int64_t n, float lam,
                                                             float inv_bs, int with_reg, float lr, float eps,
                                                             float* __restrict__ chunk_rows,
                                                             double* __restrict__ loss_part,
                                                             int* __restrict__ long_flag) {
  __shared__ double sm[8];
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int nvec = D / VEC;
  const int64_t per = (n + ngroups - 1) / ngroups;
  const int64_t p_begin = group * per, p_end = min(n, (group + 1) * per);
  uint32_t c0 = 0, c1 = 0, prev_n = 0xFFFFFFFFu;
  uint4 m0 = make_uint4(0, 0, 0, 0), m1 = m0;
  if (p_begin < p_end) {
    c0 = own_code[p_begin];
    if (p_begin > 0) prev_n = own_code[p_begin - 1];
    m0 = meta[p_begin];
    if (p_begin + 1 < n) {
      c1 = own_code[p_begin + 1];
      m1 = meta[p_begin + 1];
    }
  }
  double acc_loss = 0.0;
  // rows of the next position requested ahead only while rows are short in registers (four rows of NCH * VEC floats)
  constexpr bool kAhead = NCH <= 2;
  bool have_next = false;
  RowRegs<VEC, NCH> nown, na, nA, nB;

  for (int64_t p = p_begin; p < p_end; ++p) {
    const uint32_t code = c0, prev = prev_n, code_n = c1;
    const uint4 m_first = m0, m_next = m1;
    const bool more = p + 1 < n;
    c0 = c1;
    m0 = m1;
    if (p + 2 < n) {
      c1 = own_code[p + 2];
      m1 = meta[p + 2];
    }
    prev_n = code;
    const uint32_t id = code & kIdMask;
    const bool head = (prev & kIdMask) != id;
    if (!head && ((p & (kTripChunk - 1)) != 0 || (own_code[p - kTripChunk] & kIdMask) != id)) continue;
    const int64_t stop = min(head ? ((p + 2 * kTripChunk - 1) / kTripChunk) * kTripChunk : p + kTripChunk, n);
    RowRegs<VEC, NCH> own, a, g, fA, fB;
    if (have_next) {
      own = nown;
      a = na;
      fA = nA;
      fB = nB;
    } else {
      row_load(own, tower_row(tt, code, D), lig, G, nvec);
      row_load(fA, tower_row(tt, m_first.y, D), lig, G, nvec);
      row_load(fB, tower_row(tt, m_first.z, D), lig, G, nvec);
      row_load(a, (id >= tt.Vs ? tt.pacc + (int64_t)(id - tt.Vs) * D : tt.sacc + (int64_t)id * D), lig, G, nvec);
    }
    have_next = false;
    int64_t e_run = p + 1;
    if (more && (code_n & kIdMask) == id) {
      ++e_run;
      if (e_run < stop && (c1 & kIdMask) == id) {  // (position p + 2's record is already on its way into c1)
        ++e_run;
        while (e_run < stop && (own_code[e_run] & kIdMask) == id) ++e_run;
      }
      if (e_run > stop) e_run = stop;
    } else if (kAhead && p + 1 < p_end) {  // a run of one: position p + 1 heads the next run -- request its rows now
      const uint32_t idn = code_n & kIdMask;
      row_load(nown, tower_row(tt, code_n, D), lig, G, nvec);
      row_load(nA, tower_row(tt, m_next.y, D), lig, G, nvec);
      row_load(nB, tower_row(tt, m_next.z, D), lig, G, nvec);
      row_load(na, (idn >= tt.Vs ? tt.pacc + (int64_t)(idn - tt.Vs) * D : tt.sacc + (int64_t)idn * D), lig, G, nvec);
      have_next = true;
    }
    row_zero(g);
    // own norm: the same for every occurrence of the run
    float c = 0.f, own_norm = 0.f;
    if (with_reg) {
      own_norm = sqrtf(group_sum(row_dot_partial(own, own), G));
      c = own_norm > 1.f ? lam / own_norm : 0.f;
    }
    auto occ = [&](const uint4& m, const RowRegs<VEC, NCH>& A, const RowRegs<VEC, NCH>& Bq) {
      const uint32_t slot = m.x;
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
    occ(m_first, fA, fB);
    int64_t q = p + 1;
    auto meta_at = [&](int64_t qq) -> uint4 {  // position p + 1's record is in a register
      if (qq == p + 1) return m_next;
      return meta[qq];
    };
    for (; kAhead && q + 2 <= e_run; q += 2) {  // two occurrences = four partner rows in flight
      const uint4 ma = meta_at(q), mb = meta[q + 1];
      RowRegs<VEC, NCH> a0, b0, a1, b1;
      row_load(a0, tower_row(tt, ma.y, D), lig, G, nvec);
      row_load(b0, tower_row(tt, ma.z, D), lig, G, nvec);
      row_load(a1, tower_row(tt, mb.y, D), lig, G, nvec);
      row_load(b1, tower_row(tt, mb.z, D), lig, G, nvec);
      occ(ma, a0, b0);
      occ(mb, a1, b1);
    }
    if (q < e_run) {  // the plan record of the next occurrence travels while this one's rows do
      uint4 mq = meta_at(q);
      for (; q < e_run; ++q) {
        const uint4 m = mq;
        if (q + 1 < e_run) mq = meta[q + 1];
        RowRegs<VEC, NCH> a0, b0;
        row_load(a0, tower_row(tt, m.y, D), lig, G, nvec);
        row_load(b0, tower_row(tt, m.z, D), lig, G, nvec);
        occ(m, a0, b0);
      }
    }
    // (a run of one -- nearly every run of a uniform batch -- ends at p + 1, whose code is already in a register: the
    // reload was a dependent L2 round trip in front of the stores of a kernel that is one latency chain per group)
    const bool ends = q == n || ((q == p + 1 ? code_n : own_code[q]) & kIdMask) != id;
    if (head && ends) {
      step_apply2<VEC, NCH>(tt, code, own, a, g, D, lig, G, nvec, lr, eps);
    } else {
      const int64_t slot = 2 * (p / kTripChunk) + (head ? 1 : 0);
      row_store(g, chunk_rows + slot * D, lig, G, nvec);
      if (lig == 0) *long_flag = 1;  // (every writer stores the same value)
    }
  }
  const double t = block_sum_d(acc_loss, sm);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = t;
}

template <int VEC, int NCH>
__global__ __launch_bounds__(kBlock) void triplet_step_long_kernel(TwoTowers tt, int D, int G,
                                                                  const uint32_t* __restrict__ own_code, int64_t n,
                                                                  float lr, float eps,
                                                                  const float* __restrict__ chunk_rows, int npart,
                                                                  const double* __restrict__ loss_part,
                                                                  double inv_batch_size, float* __restrict__ loss,
                                                                  const int* __restrict__ long_flag) {
  if (blockIdx.x == gridDim.x - 1) {  // the loss: partials of the update kernel in a fixed order
    __shared__ double smp[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < npart; i += kBlock) a += loss_part[i];
    const double t = block_sum_d(a, smp);
    if (threadIdx.x == 0) loss[0] = (float)(t * inv_batch_size);
  }
  // no run of the batch outgrew its head chunk (every batch of uniform ids): nothing to combine -- one load instead of
  // the screening of the chunk boundaries (three dependent loads and three barriers per workgroup)
  if (*long_flag == 0) return;
  __shared__ float red[kBlock * VEC * NCH];
  constexpr int kPass = 4;
  __shared__ long long s_long[kPass];
  __shared__ int s_nlong, s_hoff;
  const int tid = threadIdx.x, lig = tid & (G - 1), gidx = tid / G, NG = kBlock / G;
  const int nvec = D / VEC;
  auto id_at = [&](int64_t pos) { return own_code[pos] & kIdMask; };
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
        const uint32_t code = own_code[h];
        RowRegs<VEC, NCH> own, a;
        row_load(own, tower_row(tt, code, D), lig, G, nvec);
        row_load(a, (id >= tt.Vs ? tt.pacc + (int64_t)(id - tt.Vs) * D : tt.sacc + (int64_t)id * D), lig, G, nvec);
        step_apply2<VEC, NCH>(tt, code, own, a, acc, D, lig, G, nvec, lr, eps);
      }
      __syncthreads();
    }
  }
}

}  // namespace esr

using namespace esr;

extern "C" {

size_t esr_triplet_step_workspace_bytes(int64_t B, int D) {
  if (B <= 0 || D <= 0) return 0;
  return trip_ws_layout(B, D, nullptr, nullptr);
}

int esr_triplet_train_step(float* scene, float* scene_shadow, uint8_t* scene_loc, float* scene_accum, int64_t Vs,
                           float* product, float* product_shadow, uint8_t* product_loc, float* product_accum,
                           int64_t Vp, int D, const int32_t* scene_ids, const int32_t* pos_ids,
                           const int32_t* neg_ids, int64_t B, float regularization, float batch_size, float lr,
                           float eps, const int32_t* presorted_ids, const int32_t* presorted_perm, float* loss,
                           void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  ESR_REQUIRE(B > 0 && D > 0 && Vs > 0 && Vp > 0, "esr_triplet_train_step: bad sizes Vs=%lld Vp=%lld D=%d B=%lld",
              (long long)Vs, (long long)Vp, D, (long long)B);
  ESR_REQUIRE(Vs + Vp <= (int64_t)kIdMask, "esr_triplet_train_step: %lld virtual rows exceed 2^30 - 1",
              (long long)(Vs + Vp));
  ESR_REQUIRE(3 * B < ((int64_t)1 << 31), "esr_triplet_train_step: B=%lld too large", (long long)B);
  ESR_REQUIRE(scene && scene_shadow && scene_loc && scene_accum && product && product_shadow && product_loc &&
                  product_accum && scene_ids && pos_ids && neg_ids && loss,
              "esr_triplet_train_step: null pointer");
  ESR_REQUIRE(scene != scene_shadow && product != product_shadow,
              "esr_triplet_train_step: a shadow table must be a second buffer");
  ESR_REQUIRE(batch_size != 0.f, "esr_triplet_train_step: batch_size must be non-zero");
  ESR_REQUIRE((presorted_ids == nullptr) == (presorted_perm == nullptr),
              "esr_triplet_train_step: presorted_ids and presorted_perm must both be set or both be NULL");
  const RowGeom g = step_geom_few_lanes(D);
  ESR_REQUIRE(g.nch <= kMaxChunksPerLane, "esr_triplet_train_step: D=%d not supported", D);
  if (!workspace || workspace_bytes < esr_triplet_step_workspace_bytes(B, D) || ((uintptr_t)workspace & 15)) {
    set_error("esr_triplet_train_step: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_triplet_step_workspace_bytes(B, D));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  TripWs ws;
  trip_ws_layout(B, D, (char*)workspace, &ws);
  const int64_t n = 3 * B;
  const int32_t* perm = presorted_perm;
  if (!presorted_ids) {
    const int32_t* segs[3] = {scene_ids, pos_ids, neg_ids};
    const int64_t counts[3] = {B, B, B};
    const int64_t offsets[3] = {0, Vs, Vs};
    if (int rc = esr_segment_sort_ids_multi(segs, counts, offsets, 3, Vs + Vp, ws.sorted_ids, ws.perm, ws.sort_ws,
                                            ws.sort_ws_bytes, stream))
      return rc;
    perm = ws.perm;
  }
  TwoTowers tt{scene, scene_shadow, product, product_shadow, scene_loc, product_loc, scene_accum, product_accum, Vs};
  const int nplan = (int)std::min<int64_t>(kMaxGrid, cdiv(n, kBlock));
  hipLaunchKernelGGL(triplet_plan_kernel, dim3(nplan), dim3(kBlock), 0, st, perm, scene_ids, pos_ids, neg_ids,
                     (const uint8_t*)scene_loc, (const uint8_t*)product_loc, B, Vs, ws.own_code, ws.meta, ws.long_flag);
  int grid = grid_for_groups(n, g.G);
  const int grid2 = (int)std::min<int64_t>(kMaxGrid, cdiv(cdiv(n, kTripChunk), 4));
  const float inv_bs = 1.0f / batch_size;
  ESR_DISPATCH_ROW(g, {
    grid = std::min(grid, resident_blocks((const void*)triplet_step_kernel<VEC, NCH>));
    hipLaunchKernelGGL((triplet_step_kernel<VEC, NCH>), dim3(grid), dim3(kBlock), 0, st, tt, D, g.G,
                       (const uint32_t*)ws.own_code, (const uint4*)ws.meta, n, regularization, inv_bs, 1, lr, eps,
                       ws.chunk_rows, ws.loss_part, ws.long_flag);
    hipLaunchKernelGGL((triplet_step_long_kernel<VEC, NCH>), dim3(grid2), dim3(kBlock), 0, st, tt, D, g.G,
                       (const uint32_t*)ws.own_code, n, lr, eps, (const float*)ws.chunk_rows, grid,
                       (const double*)ws.loss_part, 1.0 / (double)batch_size, loss, (const int*)ws.long_flag);
  });
  return check_launch("esr_triplet_train_step");
}

}  // extern "C"
