// Fused GloVe forward / loss / per-occurrence gradients.
//
// Reference arithmetic: wikipedia/models.py:30-37 (gather x4, row-wise dot, bias add with the
// (B,1) broadcast) and wikipedia/train_cooccurence.py:76-87 (weight, log10, mean of the (B,B)
// squared error, value_and_grad).  The (B,B) double sum is evaluated in O(B) with centred
// statistics:
//     L = (1/B^2) sum_j w_j [ B (r_j - sbar)^2 + SS ],   SS = sum_i (s_i - sbar)^2
//     dL/ddot_j = -(2 w_j / B) (r_j - sbar)
//     dL/ds_i   = -(2 / B^2) (sum_j w_j r_j - s_i sum_j w_j)
// Three launches, no host sync, deterministic (fixed reduction trees, fp64 scalars):
//   K_A bias stats : s_i, block partials of (sum s, sum s^2)
//   K_B pairs      : gather 2 rows, dot, gdot, write 2 gradient rows (rows never leave VGPRs
//                    between the dot and the gradient), block partials of
//                    (sum w, sum w r, sum w (r - center)^2)
//   K_C finalize   : loss scalar + grad_bias (reference mode needs sum w r / sum w, known only now)
#include "esr_common.h"
#include "esr_versioned.h"

namespace esr {

constexpr int kStatBlocks = 256;   // K_A grid cap  -> 2 doubles per block
constexpr int kPairBlocks = 2048;  // K_B grid cap (256 CUs x 8 resident blocks) -> 3 doubles per block

struct GloveWs {
  double* stat_part;  // [kStatBlocks][2]
  double* pair_part;  // [kPairBlocks][3]
  float* s;           // [B]
};

static size_t glove_ws_layout(int64_t B, char* base, GloveWs* ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  double* a = (double*)take(sizeof(double) * 2 * kStatBlocks);
  double* b = (double*)take(sizeof(double) * 3 * kPairBlocks);
  float* s = (float*)take(sizeof(float) * (size_t)B);
  if (ws) *ws = GloveWs{a, b, s};
  return off;
}

// K_A: s_i = Bias[t1_i] + Bias[t2_i]; per-block partial (sum s, sum s^2) in fp64.
__global__ __launch_bounds__(kBlock) void glove_bias_stats_kernel(const float* __restrict__ bias,
                                                                 const int32_t* __restrict__ inputs,
                                                                 int64_t B, float* __restrict__ s_out,
                                                                 double* __restrict__ part) {
  __shared__ double sm[8];
  double a = 0.0, a2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B; i += (int64_t)gridDim.x * kBlock) {
    const float s = bias[inputs[i]] + bias[inputs[B + i]];
    s_out[i] = s;
    a += (double)s;
    a2 += (double)s * (double)s;
  }
  if (part) {
    const double t = block_sum_d(a, sm);
    const double t2 = block_sum_d(a2, sm + 4);
    if (threadIdx.x == 0) {
      part[2 * blockIdx.x] = t;
      part[2 * blockIdx.x + 1] = t2;
    }
  }
}

// Every block re-reduces the <= 256 K_A partials in the same fixed order (4 KB from L2): this
// replaces a separate finalize launch and keeps all blocks bit-identical.
__device__ __forceinline__ void reduce_stat_parts(const double* __restrict__ part, int nparts,
                                                  double* sm /* >= 10 doubles */, double* sum_s,
                                                  double* sum_s2) {
  double a = 0.0, a2 = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) {
    a += part[2 * i];
    a2 += part[2 * i + 1];
  }
  const double t = block_sum_d(a, sm);
  const double t2 = block_sum_d(a2, sm + 4);
  if (threadIdx.x == 0) {
    sm[8] = t;
    sm[9] = t2;
  }
  __syncthreads();
  *sum_s = sm[8];
  *sum_s2 = sm[9];
  __syncthreads();
}

// K_B: one row group (G lanes) per pair.  LOSS=false is the forward-only path (dot only).
template <int VEC, int NCH, bool LOSS, bool GRADS>
__global__ __launch_bounds__(kBlock) void glove_pairs_kernel(
    const float* __restrict__ emb, const int32_t* __restrict__ inputs,
    const float* __restrict__ target, int64_t B, int D, int G, int mode, int nstat,
    const double* __restrict__ stat_part, const float* __restrict__ s_in, float* __restrict__ dot_out,
    float* __restrict__ grad_rows, float* __restrict__ grad_bias, double* __restrict__ pair_part) {
  __shared__ double sm[16];
  const bool at_ids = (mode & ESR_GRADS_AT_IDS) != 0;  // gradient rows go where their table rows came from
  mode &= ~ESR_GRADS_AT_IDS;
  double sum_s = 0.0, sum_s2 = 0.0;
  if (LOSS && mode == ESR_GLOVE_REFERENCE) reduce_stat_parts(stat_part, nstat, sm, &sum_s, &sum_s2);
  const float sbar = (float)(sum_s / (double)B);
  const float two_over_B = 2.0f / (float)B;

  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int nvec = D / VEC;

  double acc_w = 0.0, acc_wr = 0.0, acc_wq = 0.0;
  for (int64_t j = group; j < B; j += ngroups) {
    const int64_t t1 = inputs[j], t2 = inputs[B + j];
    RowRegs<VEC, NCH> e1, e2;
    row_load(e1, emb + t1 * D, lig, G, nvec);
    row_load(e2, emb + t2 * D, lig, G, nvec);
    const float dot = group_sum(row_dot_partial(e1, e2), G);
    if (!LOSS) {
      if (lig == 0) dot_out[j] = dot;
      continue;
    }
    const float c_j = target[j];
    // weight = min(1, c/100)^0.75 ; log_target = log10(1 + c)   (train_cooccurence.py:79-82)
    const float w = powf(fminf(1.0f, c_j / 100.0f), 0.75f);
    const float lt = log10f(1.0f + c_j);
    const float r = lt - dot;
    const float center = (mode == ESR_GLOVE_REFERENCE) ? sbar : s_in[j];
    const float gdot = -(two_over_B * w) * (r - center);
    if (lig == 0) {
      const double q = (double)r - (double)center;
      acc_w += (double)w;
      acc_wr += (double)w * (double)r;
      acc_wq += (double)w * q * q;
      if (GRADS && mode == ESR_GLOVE_DIAGONAL) {
        grad_bias[at_ids ? t1 : j] = gdot;
        grad_bias[at_ids ? t2 : B + j] = gdot;
      }
    }
    if (GRADS) {
      RowRegs<VEC, NCH> g1, g2;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          g1.v[k][e] = gdot * e2.v[k][e];
          g2.v[k][e] = gdot * e1.v[k][e];
        }
      row_store(g1, grad_rows + (at_ids ? t1 : j) * D, lig, G, nvec);
      row_store(g2, grad_rows + (at_ids ? t2 : B + j) * D, lig, G, nvec);
    }
  }
  if (LOSS) {
    const double tw = block_sum_d(acc_w, sm);
    const double twr = block_sum_d(acc_wr, sm + 4);
    const double twq = block_sum_d(acc_wq, sm + 8);
    if (threadIdx.x == 0) {
      pair_part[3 * blockIdx.x] = tw;
      pair_part[3 * blockIdx.x + 1] = twr;
      pair_part[3 * blockIdx.x + 2] = twq;
    }
  }
}

// K_C: every block re-reduces the K_B partials (<= 24 KB) in fixed order; block 0 writes the loss;
// all blocks write grad_bias for their slice (reference mode).
__global__ __launch_bounds__(kBlock) void glove_finalize_kernel(
    int64_t B, int mode, int nstat, int npair, const double* __restrict__ stat_part,
    const double* __restrict__ pair_part, const float* __restrict__ s_in, const int32_t* __restrict__ inputs,
    float* __restrict__ loss, float* __restrict__ grad_bias) {
  __shared__ double sm[20];
  const bool at_ids = (mode & ESR_GRADS_AT_IDS) != 0;
  mode &= ~ESR_GRADS_AT_IDS;
  double sum_s = 0.0, sum_s2 = 0.0;
  if (mode == ESR_GLOVE_REFERENCE) reduce_stat_parts(stat_part, nstat, sm, &sum_s, &sum_s2);
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < npair; i += kBlock) {
    a += pair_part[3 * i];
    b += pair_part[3 * i + 1];
    c += pair_part[3 * i + 2];
  }
  const double tw = block_sum_d(a, sm);
  const double twr = block_sum_d(b, sm + 4);
  const double twq = block_sum_d(c, sm + 8);
  if (threadIdx.x == 0) {
    sm[12] = tw;
    sm[13] = twr;
    sm[14] = twq;
  }
  __syncthreads();
  const double Sw = sm[12], Swr = sm[13], Swq = sm[14];
  const double Bd = (double)B;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double L;
    if (mode == ESR_GLOVE_REFERENCE) {
      double SS = sum_s2 - sum_s * sum_s / Bd;  // sum_i (s_i - sbar)^2, fp64
      if (SS < 0.0) SS = 0.0;
      L = (Bd * Swq + Sw * SS) / (Bd * Bd);
    } else {
      L = Swq / Bd;
    }
    loss[0] = (float)L;
  }
  if (mode == ESR_GLOVE_REFERENCE && grad_bias) {
    const float k = (float)(2.0 / (Bd * Bd));
    const float fSw = (float)Sw, fSwr = (float)Swr;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B; i += (int64_t)gridDim.x * kBlock) {
      const float gs = -k * (fSwr - s_in[i] * fSw);
      grad_bias[at_ids ? (int64_t)inputs[i] : i] = gs;
      grad_bias[at_ids ? (int64_t)inputs[B + i] : B + i] = gs;
    }
  }
}

// =====================================================================================================================
// The whole train step in one pass over the rows: esr_glove_train_step (G3 + G4 + the build's sparse Adagrad).
//
// The three-kernel path above materialises a [2B, D] gradient (written once, read once by the optimizer): 7 row
// transfers per occurrence where the algorithm needs ~4.5 (read the partner row; read, rewrite the own row once per
// DISTINCT row; read-modify-write its accumulator).  Producing the gradient inside the update instead runs into a
// hazard: occurrence (row a, partner b) needs the PRE-step value of b while another workgroup may already have
// rewritten b.  288 GB of HBM buy the way out: the table is kept in TWO buffers with a per-row byte `loc` that says
// which one holds the row's current value and which step last moved it (esr_versioned.h).  The step reads every row
// where it lived when the step began, writes every updated row into the OTHER buffer and stamps its byte.  Readers and
// writers of one step never touch the same bytes, no gradient row and no snapshot ever goes to memory, and untouched
// rows cost nothing.  esr_rows_consolidate copies the rows that live in the second buffer back into the primary one
// when somebody needs a plain [V, D] table (eval, kNN, checkpoint).
//
//   sort      occurrence ids [t1 ; t2] -> (sorted, perm)                       } ids + counts only: made AHEAD
//   plan      per sorted position: partner row (id | side bit), w_j, log10(1 + c_j);  } (esr_glove_plan, with the sort)
//             "may a run outgrow its head chunk?" -- the host's reason to launch `long`
//   update    prologue: the first workgroups sum the bias statistics of K_A into integer words and every workgroup
//             waits for them (round 3: no separate launch); then one row group per sorted position, the head of a run
//             walks it: row locations from the stamped bytes (esr_versioned.h), s_j from the bias table, dot with each
//             partner row, gdot, G += gdot * partner (left to right, products rounded as the gradient rows used to be),
//             Adagrad once per distinct row; loss partials; per-run bias sums
//   long      only when a run outgrew its head chunk (hot tokens): chunk partials combined in a fixed order
//   finalize  loss scalar; bias Adagrad (needs the global sum of w r, known only now)
// Round 2 ran plan -> update -> long -> finalize per step; a step of uniform ids is now update -> finalize.
// =====================================================================================================================

constexpr int kStatBlocksStep = 256;  // workgroups of the update kernel that sum the bias statistics first
// Two ways to learn where rows live.  Lists up to this many ids (the reference's batch sizes: launch-bound steps): the
// update kernel reads the stamped bytes and the bias values itself -- no launch in front of it.  Longer lists (the
// update kernel is tens of microseconds of streaming): a resolve launch in front of every step writes row codes with
// the location bits in them (round 2's plan kernel), because the extra dependent stage in the update kernel's walk and
// the wait for the in-kernel statistics cost that kernel 20 % at V = 465 537 x 256, B = 65 536 (111 -> 135 us) -- more
// than the 9 us launch they replace.
constexpr int64_t kResolveMinIds = 32768;

// the plan of one batch (esr_glove_plan): header + one record per sorted position
struct GlovePlan {
  int* flags;                  // [0] parked: set by the update kernel when it parks a chunk partial
  unsigned long long* stat;    // 16-byte-apart words at 128 B stride: [0] sum s hi, [16] lo, [32] sum s^2 hi, [48] lo,
                               // [64] arrivals, [80] poison; zero before the update kernel
  unsigned long long* fin;     // kFinWords: the update kernel's loss sums (fin_arrive) -- six order-free integer
                               // accumulators, an arrival word and a poison word; zero before the update kernel
  float4* meta;                // [n]  {partner id | side bit (bits), w, log10(1 + c), unused}
};
constexpr int kStatWords = 96;
constexpr int kFinAccs = 6;                                // (sum w, sum w r, sum w q^2) x (hi, lo)
constexpr int kFinWords = kFinAccs * kFixAccWords + 32;    // + [0] arrivals of finished accumulators, [16] poison
constexpr int64_t kFinFuseMaxIds = 8192;  // lists up to this long: the update kernel's last workgroup is the finalize step
static size_t glove_plan_layout(int64_t B, char* base, GlovePlan* out) {
  const int64_t n = 2 * B;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  GlovePlan pl;
  pl.flags = (int*)take(sizeof(int) * 64);
  pl.stat = (unsigned long long*)take(sizeof(unsigned long long) * kStatWords);
  pl.fin = (unsigned long long*)take(sizeof(unsigned long long) * kFinWords);
  pl.meta = (float4*)take(sizeof(float4) * (size_t)n);
  if (out) *out = pl;
  return off;
}

struct StepWs {
  int32_t* sorted_ids;   // [n]
  int32_t* perm;         // [n]
  double2* bias_info;    // [n]  per run head / chunk start: {sum over its occurrences of s (reference) or gdot, count}
  double* pair_part;     // [kPairBlocks][3]
  uint32_t* own_code;    // [n]  resolved mode: id | loc bit of the row that sorted position p updates
  float4* meta_res;      // [n]  resolved mode: {partner code (id | loc bit | side bit), w, log10(1 + c), s}
  double* stat_part;     // [kStatBlocks][2]  resolved mode: K_A's partials
  int* res_flags;        // [64] resolved mode: [0] parked
  double* pair_tot;      // [3]  resolved mode: sum w, sum w r, sum w (r - center)^2 over the batch
  float* chunk_rows;     // [2 * ceil(n / 32)][D]  partial sums of long runs (slot 2c: chunk starting at 32c; 2c + 1:
                         //                        the head chunk whose head lies in block c)
  char* plan;            // glove_plan_layout(B) bytes: the in-line plan of a call that brings none
  void* sort_ws;
  size_t sort_ws_bytes;
};

static size_t step_ws_layout(int64_t B, int D, char* base, StepWs* ws) {
  const int64_t n = 2 * B;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  StepWs w;
  w.sorted_ids = (int32_t*)take(sizeof(int32_t) * (size_t)n);
  w.perm = (int32_t*)take(sizeof(int32_t) * (size_t)n);
  w.bias_info = (double2*)take(sizeof(double2) * (size_t)n);
  w.pair_part = (double*)take(sizeof(double) * 3 * kPairBlocks);
  // the resolved mode's records exist for long lists only (kResolveMinIds); short ones resolve inside the update kernel
  const bool res = n > kResolveMinIds;
  w.own_code = (uint32_t*)take(res ? sizeof(uint32_t) * (size_t)n : 0);
  w.meta_res = (float4*)take(res ? sizeof(float4) * (size_t)n : 0);
  w.stat_part = (double*)take(sizeof(double) * 2 * kStatBlocks);
  w.res_flags = (int*)take(sizeof(int) * 64);
  w.pair_tot = (double*)take(sizeof(double) * 4);
  w.chunk_rows = (float*)take(sizeof(float) * 2 * (size_t)cdiv(n, kStepChunk) * (size_t)D);
  w.plan = take(glove_plan_layout(B, nullptr, nullptr));
  w.sort_ws_bytes = esr_segment_sort_workspace_bytes(n);
  w.sort_ws = take(w.sort_ws_bytes);
  if (ws) *ws = w;
  return off;
}

constexpr int kMaxGlovePlanBatch = 8;
struct GlovePlanBatch {
  const int32_t* inputs[kMaxGlovePlanBatch];
  const float* target[kMaxGlovePlanBatch];
};

// plan: one thread per sorted position of list blockIdx.y; nothing here reads a table.  hints[list] = gen when some run
// of equal ids is longer than kStepChunk positions (only then can the update kernel park a chunk partial).
__global__ __launch_bounds__(kBlock) void glove_plan_kernel(GlovePlanBatch pb, const int32_t* __restrict__ sorted_all,
                                                           const int32_t* __restrict__ perm_all, int64_t B,
                                                           char* __restrict__ plans, size_t plan_stride,
                                                           int* __restrict__ hints, int gen) {
  const int list = blockIdx.y;
  const int64_t n = 2 * B;
  const int32_t* __restrict__ sorted = sorted_all + (int64_t)list * n;
  const int32_t* __restrict__ perm = perm_all + (int64_t)list * n;
  const int32_t* __restrict__ inputs = pb.inputs[list];
  const float* __restrict__ target = pb.target[list];
  char* base = plans + (size_t)list * plan_stride;
  int* flags = (int*)base;
  unsigned long long* stat = (unsigned long long*)(base + 256);
  const size_t kFinOff = 256 + align_up(sizeof(unsigned long long) * kStatWords, 256);
  unsigned long long* fin = (unsigned long long*)(base + kFinOff);
  float4* meta = (float4*)(base + kFinOff + align_up(sizeof(unsigned long long) * kFinWords, 256));
  if (blockIdx.x == 0) {
    if (threadIdx.x < 64) flags[threadIdx.x] = 0;
    if (threadIdx.x < kStatWords) stat[threadIdx.x] = 0ull;
    for (int i = threadIdx.x; i < kFinWords; i += kBlock) fin[i] = 0ull;
  }
  bool long_run = false;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
    const int64_t o = perm[p];
    const bool side = o >= B;
    const int64_t j = side ? o - B : o;
    const int32_t partner = side ? inputs[j] : inputs[B + j];
    const float c_j = target[j];
    float4 m;
    m.x = __uint_as_float((uint32_t)partner | (side ? kSideBit : 0u));
    m.y = powf(fminf(1.0f, c_j / 100.0f), 0.75f);  // weight = min(1, c/100)^0.75   (train_cooccurence.py:79-81)
    m.z = log10f(1.0f + c_j);                      // log10(1 + c)                  (train_cooccurence.py:82)
    m.w = 0.f;
    meta[p] = m;
    if (p + kStepChunk < n && sorted[p] == sorted[p + kStepChunk]) long_run = true;
  }
  if (hints && __any(long_run) && (threadIdx.x & 63) == 0) hints[list] = gen;  // (every writer stores the same value)
}

// resolve (long lists, every step): blocks [0, nstat) first redo K_A's loop over the pairs (same grid, same per-thread
// order: the bias statistics, hence sbar and every gdot, are bit-identical to the three-kernel path); then every block
// resolves its sorted positions: own and partner row codes with the location bit the bytes show NOW (between two steps:
// nobody is writing them), weight, log10(1 + c) and the bias sum s.
__global__ __launch_bounds__(kBlock) void glove_resolve_kernel(const int32_t* __restrict__ sorted_ids,
                                                              const int32_t* __restrict__ perm,
                                                              const int32_t* __restrict__ inputs,
                                                              const float* __restrict__ target,
                                                              const float* __restrict__ bias,
                                                              const uint8_t* __restrict__ loc, int64_t B, int nstat,
                                                              uint32_t* __restrict__ own_code, float4* __restrict__ meta,
                                                              double* __restrict__ stat_part, int* __restrict__ flags) {
  __shared__ double sm[8];
  if (blockIdx.x == 0 && threadIdx.x == 0) flags[0] = 0;  // parked: set by the update kernel
  if ((int)blockIdx.x < nstat) {
    double a = 0.0, a2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B; i += (int64_t)nstat * kBlock) {
      const float s = bias[inputs[i]] + bias[inputs[B + i]];
      a += (double)s;
      a2 += (double)s * (double)s;
    }
    const double t = block_sum_d(a, sm);
    const double t2 = block_sum_d(a2, sm + 4);
    if (threadIdx.x == 0) {
      stat_part[2 * blockIdx.x] = t;
      stat_part[2 * blockIdx.x + 1] = t2;
    }
  }
  const int64_t n = 2 * B;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
    const int32_t id = sorted_ids[p];
    const int64_t o = perm[p];
    const bool side = o >= B;
    const int64_t j = side ? o - B : o;
    const int32_t t1 = inputs[j], t2 = inputs[B + j];
    const int32_t partner = side ? t1 : t2;
    const float c_j = target[j];
    float4 m;
    m.x = __uint_as_float((uint32_t)partner | ((loc[partner] & 1) ? kLocBit : 0u) | (side ? kSideBit : 0u));
    m.y = powf(fminf(1.0f, c_j / 100.0f), 0.75f);  // weight = min(1, c/100)^0.75   (train_cooccurence.py:79-81)
    m.z = log10f(1.0f + c_j);                      // log10(1 + c)                  (train_cooccurence.py:82)
    m.w = bias[t1] + bias[t2];
    meta[p] = m;
    own_code[p] = (uint32_t)id | ((loc[id] & 1) ? kLocBit : 0u);
  }
}

// the versioned read-modify-write of one row with its summed gradient g: the row was read where `code` says it lived
// when the step began (`own`), its new value goes to the OTHER buffer and the byte takes this step's stamp.
// `a` = the row's accumulator, loaded by the caller next to `own`.
// TR: the embedding table's element type -- float, or uint16_t for bf16 rows (round 6; the register image stays f32: loads
// widen, stores round to nearest even; accumulators and the bias table are fp32 either way)
template <int VEC, int NCH, class TR>
__device__ __forceinline__ void step_apply(TR* __restrict__ emb0, TR* __restrict__ emb1, uint8_t* __restrict__ loc,
                                           float* __restrict__ accum, uint32_t code, uint32_t T,
                                           const RowRegs<VEC, NCH>& own, RowRegs<VEC, NCH>& a,
                                           const RowRegs<VEC, NCH>& g, int D, int lig, int G, int nvec, float lr,
                                           float eps) {
  const int64_t id = code & kIdMask;
  RowRegs<VEC, NCH> w = own;
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) adagrad_elem(w.v[k][e], a.v[k][e], g.v[k][e], lr, eps);
  row_store(a, accum + id * D, lig, G, nvec);
  row_store(w, ((code & kLocBit) ? emb0 : emb1) + id * D, lig, G, nvec);
  if (lig == 0) loc[id] = loc_written((code & kLocBit) ? 0u : 1u, T);
}

// one position's plan record with what the tables add to it: location bytes and bias values of own and partner row
struct GloveRec {
  uint32_t id;     // own row
  float4 m;        // {partner | side bit, w, log10(1 + c), -}
  uint32_t y0, y1; // location bytes of own and partner row (as loaded)
  float b0, b1;    // bias of own and partner row
};
__device__ __forceinline__ void glove_rec_tables(const uint8_t* __restrict__ loc, const float* __restrict__ bias,
                                                 GloveRec& r) {
  const uint32_t partner = __float_as_uint(r.m.x) & kIdMask;
  r.y0 = loc[r.id];
  r.y1 = loc[partner];
  r.b0 = bias[r.id];
  r.b1 = bias[partner];
}

// ---- the update kernel's loss sums, order-free and exact, with the arrival of the last workgroup known -----------------
// Thread a < kFinAccs of every workgroup adds its share of sum (a / 2) to accumulator a -- a hi part at 2^-8 and a lo
// part at 2^-50 of what the hi part left (both exact integers: the totals do not depend on the order of arrival) --
// through the two-level counted words of fixed_sum_arrive (esr_versioned.h): one integer atomic per accumulator and
// workgroup, whose return value says whether the word is complete; the completer forwards the word's total to the
// accumulator's master word, whose low 11 bits count the words that have arrived: a reader that sees nwords there has
// the total in the same load (fin_poll).  Data flows through atomic return values and that one word only.
// Range: |sum| < 2^43 (beyond it, or non-finite: the poison word is raised and the loss comes out NaN).
__device__ __forceinline__ unsigned fin_nwords(unsigned nblocks) {
  return nblocks < (unsigned)kFixWords ? nblocks : (unsigned)kFixWords;
}
__device__ __forceinline__ void fin_arrive(unsigned long long* fin, int a, double v, unsigned nblocks) {
  unsigned long long* acc = fin + a * kFixAccWords;
  unsigned long long* tail = fin + kFinAccs * kFixAccWords;
  const unsigned wd = blockIdx.x % kFixWords;
  const unsigned on_word = (nblocks - wd + kFixWords - 1) / kFixWords;
  const double h = rint(ldexp(v, 8));
  const double payload = (a & 1) ? rint(ldexp(v - ldexp(h, -8), 50)) : h;  // |v - h 2^-8| <= 2^-9: lo <= 2^41
  unsigned long long add = ((unsigned long long)__double2ll_rn(payload)) << 11;
  if (!(fabs(h) < 2251799813685248.0)) {  // 2^51: non-finite or out of range
    const unsigned r = atomicOr(reinterpret_cast<unsigned*>(tail + 16), 1u);  // raised BEFORE this workgroup is counted
    add = (unsigned long long)(r >> 1);
  }
  const unsigned long long old = atomicAdd(acc + 16 * (1 + wd), add + 1ull);
  if ((unsigned)(old & 2047ull) != on_word - 1) return;
  const unsigned long long word_total = ((old + add) >> 11) << 11;
  atomicAdd(acc, word_total + 1ull);
}
// accumulator a's master word once all its words have arrived (the payload is the word's upper 53 bits); false after
// `ticks` of the 100 MHz clock
__device__ __forceinline__ bool fin_poll(const unsigned long long* fin, int a, unsigned nblocks, unsigned long long ticks,
                                         long long* payload) {
  const unsigned nwords = fin_nwords(nblocks);
  const unsigned long long t_start = wall_clock64();
  unsigned long long m;
  while ((unsigned)((m = coherent_load(fin + a * kFixAccWords)) & 2047ull) != nwords) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t_start > ticks) return false;
  }
  *payload = ((long long)m) >> 11;  // arithmetic shift: signed
  return true;
}
// sum s (0..2) once every accumulator is complete
__device__ __forceinline__ double fin_value(const unsigned long long* fin, int s) {  // (a later launch: all words have arrived)
  const long long hi = ((long long)coherent_load(fin + (2 * s) * kFixAccWords)) >> 11;       // arithmetic shifts: signed
  const long long lo = ((long long)coherent_load(fin + (2 * s + 1) * kFixAccWords)) >> 11;
  return ldexp((double)hi, -8) + ldexp((double)lo, -50);
}
__device__ __forceinline__ bool fin_poisoned(const unsigned long long* fin) {
  return coherent_load(reinterpret_cast<const unsigned*>(fin + kFinAccs * kFixAccWords + 16)) != 0u;
}

// sum w (r - center)^2 from what the update kernel's third accumulator holds: reference mode -- sum w r^2, and
// sum w (r - sbar)^2 = sum w r^2 - 2 sbar sum w r + sbar^2 sum w with the f32 sbar the gradients used; diagonal mode --
// the sum itself
__device__ __forceinline__ double glove_swq(int mode, double Bd, double Sw, double Swr, double S3, double sum_s) {
  if (mode != ESR_GLOVE_REFERENCE) return S3;
  const double sb = (double)(float)(sum_s / Bd);
  const double v = S3 - 2.0 * sb * Swr + sb * sb * Sw;
  return v < 0.0 ? 0.0 : v;
}
// the loss scalar from the batch sums (glove_finalize_kernel's formula)
__device__ __forceinline__ float glove_loss_value(int mode, double Bd, double Sw, double Swq, double sum_s, double sum_s2,
                                                  bool poisoned) {
  if (mode != ESR_GLOVE_REFERENCE) return (float)(poisoned ? __builtin_nan("") : Swq / Bd);
  double SS = sum_s2 - sum_s * sum_s / Bd;
  if (SS < 0.0) SS = 0.0;
  return (float)(poisoned ? __builtin_nan("") : (Bd * Swq + Sw * SS) / (Bd * Bd));
}

// update: the structure of segment_update_kernel (esr_optim.hip) with the gradient rows produced on the fly.
// A group's critical path per position is ONE memory round trip: plan records run three positions ahead, the bytes and
// bias values they point at two, and the own row, its accumulator and the first partner row of the NEXT position are
// requested before the current one is computed.  Written naively -- record, then bytes, then rows, then accumulator --
// the loop is a chain of dependent round trips and ran at 3.2 TB/s.
template <int VEC, int NCH, class TR>
__global__ __launch_bounds__(kBlock) void glove_step_kernel(
    TR* __restrict__ emb0, TR* __restrict__ emb1, uint8_t* __restrict__ loc, float* __restrict__ accum,
    const float* __restrict__ bias, int D, int G, const int32_t* __restrict__ sorted_ids,
    const float4* __restrict__ meta, const int32_t* __restrict__ inputs, int64_t n, int64_t B, int mode, uint32_t T,
    int nstat, unsigned long long* __restrict__ stat, float lr, float eps, float* __restrict__ chunk_rows,
    double2* __restrict__ bias_info, unsigned long long* __restrict__ fin, int fuse_fin, float* __restrict__ bias_rw,
    float* __restrict__ bias_accum, float* __restrict__ loss, int* __restrict__ parked,
    uint32_t* __restrict__ start_flag, uint32_t start_value) {
  // fuse_fin (the caller knows that no run outgrows its head chunk, and the list is short): the workgroup that arrives
  // last at the loss sums IS the finalize step -- loss scalar and the bias table's Adagrad -- and no launch follows.
  __shared__ double sm[16];
  __shared__ int sm_last;
  announce_start(start_flag, start_value);
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  // The first nstat workgroups (reference mode) ONLY sum the bias statistics (round 4): they used to walk a slice of
  // positions afterwards and, two memory round trips and a reduction behind everybody else, were the launch's tail.
  const int64_t wg = (int64_t)blockIdx.x - nstat;
  const int64_t group = wg * gpb + threadIdx.x / G;
  const int64_t ngroups = ((int64_t)gridDim.x - nstat) * gpb;
  const int nvec = D / VEC;
  const int64_t per = (n + ngroups - 1) / ngroups;  // contiguous slices (see segment_update_kernel)
  const int64_t p_begin = wg < 0 ? 0 : min(n, group * per), p_end = wg < 0 ? 0 : min(n, (group + 1) * per);
  // the records of the first three positions are requested before anything else
  GloveRec r0{0, make_float4(0.f, 0.f, 0.f, 0.f), 0, 0, 0.f, 0.f}, r1 = r0, r2 = r0;
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
  }
  // ---- prologue (reference mode): sum s and sum s^2 over the pairs, s_i = Bias[t1_i] + Bias[t2_i] ------------------
  // The first nstat workgroups add their partials to two-word integer sums and count themselves in; every workgroup
  // (those too) then waits for the count.  Workgroups are dispatched in index order, so the ones everybody waits for
  // are never behind a waiting one; the data travels in the atomics, so no fence is needed.
  if (mode == ESR_GLOVE_REFERENCE) {
    if ((int)blockIdx.x < nstat) {
      double a = 0.0, a2 = 0.0;
      const int64_t stride = (int64_t)nstat * kBlock;
      for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < B; i0 += 4 * stride) {
        // four pairs per trip: all eight ids requested together, then all eight bias values (two round trips, not eight)
        int32_t t1[4], t2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t i = i0 + u * stride;
          t1[u] = i < B ? inputs[i] : -1;
          t2[u] = i < B ? inputs[B + i] : -1;
        }
        float b1[4], b2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          b1[u] = t1[u] >= 0 ? bias[t1[u]] : 0.f;
          b2[u] = t1[u] >= 0 ? bias[t2[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float s = b1[u] + b2[u];
          a += (double)s;
          a2 += (double)s * (double)s;
        }
      }
      const double t = block_sum_d(a, sm);
      const double t2 = block_sum_d(a2, sm + 4);
      if (threadIdx.x == 0) {
        if (!(fabs(t) < 2.7e11) || !(t2 < 2.7e11)) {  // 2^38: beyond the words' range, or not finite
          atomicOr(reinterpret_cast<unsigned*>(stat + 80), 1u);
        } else {
          fixed2_add2(stat + 0, stat + 16, t, stat + 32, stat + 48, t2);
        }
        atomicAdd(stat + 64, 1ull);  // (fixed2_add2 returns after its adds have been performed)
      }
    }
  }
  if (p_begin < p_end) {
    glove_rec_tables(loc, bias, r0);
    if (p_begin + 1 < n) glove_rec_tables(loc, bias, r1);
  }
  // Nobody waits for the statistics here (round 4).  In reference mode a row's gradient is
  //   G = sum_j gdot_j partner_j,  gdot_j = -(2 w_j / B) (r_j - sbar)   =>   G = -(2 / B) (A - sbar C),
  //   A = sum_j (w_j r_j) partner_j,  C = sum_j w_j partner_j:
  // the walk accumulates A and C, which need no statistic, and sbar enters when a run is applied (or parked) -- by then
  // the first workgroups have long published it: the wait (per wave, bounded) no longer sits in front of every row load.
  // The loss sum follows the same way: sum w (r - sbar)^2 = sum w r^2 - 2 sbar sum w r + sbar^2 sum w (fp64, in finalize).
  bool have_stats = mode != ESR_GLOVE_REFERENCE;
  float sbar = 0.f;
  auto need_stats = [&]() {
    if (have_stats) return;
    const unsigned long long t_start = wall_clock64();
    bool ok = true;
    while (coherent_load(stat + 64) < (unsigned long long)nstat) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t_start > 300000000ull) { ok = false; break; }  // 3 s: poison the loss, do not hang the queue
    }
    if (!ok && lig == 0) atomicOr(reinterpret_cast<unsigned*>(stat + 80), 1u);
    const double v = fixed2_value(coherent_load(stat + 0), coherent_load(stat + 16));
    const bool bad = !ok || coherent_load(reinterpret_cast<const unsigned*>(stat + 80)) != 0u;
    sbar = (float)((bad ? __builtin_nan("") : v) / (double)B);
    have_stats = true;
  };
  const float two_over_B = 2.0f / (float)B;
  double acc_w = 0.0, acc_wr = 0.0, acc_wq = 0.0;  // acc_wq: sum w r^2 (reference mode) / sum w (r - s)^2 (diagonal)
  auto emb_row = [&](uint32_t id, uint32_t byte) {
    return (loc_at_step_begin(byte, T) ? emb1 : emb0) + (int64_t)id * D;
  };
  auto part_row = [&](const GloveRec& r) { return emb_row(__float_as_uint(r.m.x) & kIdMask, r.y1); };
  // rows of the NEXT position, requested before this position is computed
  bool have_next = false;
  RowRegs<VEC, NCH> nown, nfirst, na;
  // the first run this group heads, kept for the in-launch finalize (fuse_fin)
  int64_t h_p = -1;
  uint32_t h_id = 0;
  double h_bsum = 0.0, h_cnt = 0.0;
  float h_w = 0.f, h_ac = 0.f;
  bool h_more = false;

  for (int64_t p = p_begin; p < p_end; ++p) {
    const GloveRec cur = r0, nxt = r1;
    const uint32_t prev = prev_n;
    const bool more = p + 1 < n;
    r0 = r1;
    r1 = r2;
    if (p + 2 < n) glove_rec_tables(loc, bias, r1);  // position p + 2's record arrived an iteration ago
    if (p + 3 < n) {
      r2.id = (uint32_t)sorted_ids[p + 3];
      r2.m = meta[p + 3];
    }
    prev_n = cur.id;
    const uint32_t id = cur.id;
    const bool head = prev != id;  // prev = all ones at p == 0: no id equals it
    if (!head && ((p & (kStepChunk - 1)) != 0 || (uint32_t)sorted_ids[p - kStepChunk] != id)) continue;
    const int64_t stop = min(head ? ((p + 2 * kStepChunk - 1) / kStepChunk) * kStepChunk : p + kStepChunk, n);
    const uint32_t code = id | (loc_at_step_begin(cur.y0, T) ? kLocBit : 0u);
    RowRegs<VEC, NCH> own, a, g, first;
    if (have_next) {
      own = nown;
      first = nfirst;
      a = na;
    } else {
      row_load(own, emb_row(id, cur.y0), lig, G, nvec);
      row_load(first, part_row(cur), lig, G, nvec);
      row_load(a, accum + (int64_t)id * D, lig, G, nvec);
    }
    have_next = false;
    int64_t e_run = p + 1;
    if (more && nxt.id == id) {  // a run of several occurrences: find the end of this chunk
      ++e_run;
      while (e_run < stop && (uint32_t)sorted_ids[e_run] == id) ++e_run;
      if (e_run > stop) e_run = stop;
    } else if (p + 1 < p_end) {  // the usual case, a run of one: position p + 1 heads the next run -- request its rows now
      row_load(nown, emb_row(nxt.id, nxt.y0), lig, G, nvec);
      row_load(nfirst, part_row(nxt), lig, G, nvec);
      row_load(na, accum + (int64_t)nxt.id * D, lig, G, nvec);
      have_next = true;
    }
    RowRegs<VEC, NCH> gc;  // reference mode: C = sum w partner (g holds A = sum (w r) partner until the run ends)
    row_zero(g);
    row_zero(gc);
    double bsum = 0.0;  // fp64: a hot token's bias gradient is a sum over thousands of occurrences
    // one occurrence.  Diagonal mode: gdot from the dot with its partner row; G += gdot * partner, the product rounded
    // to f32 first (it used to be stored as a gradient row), additions strictly left to right.  Reference mode: the two
    // sbar-free sums A and C, same association.
    auto occ = [&](const GloveRec& r, const RowRegs<VEC, NCH>& part) {
      const float dot = group_sum(row_dot_partial(own, part), G);
      const float rr = r.m.z - dot;
      const float s = r.b0 + r.b1;  // Bias[t1] + Bias[t2] (an f32 addition commutes: either side gives K_A's bits)
      if (mode == ESR_GLOVE_REFERENCE) {
        const float wr = __fmul_rn(r.m.y, rr);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            g.v[k][e] = __fadd_rn(g.v[k][e], __fmul_rn(wr, part.v[k][e]));
            gc.v[k][e] = __fadd_rn(gc.v[k][e], __fmul_rn(r.m.y, part.v[k][e]));
          }
        bsum += (double)s;
        if (lig == 0 && !(__float_as_uint(r.m.x) & kSideBit)) {  // every pair is seen from both sides: count it once
          acc_w += (double)r.m.y;
          acc_wr += (double)r.m.y * (double)rr;
          acc_wq += (double)r.m.y * (double)rr * (double)rr;
        }
      } else {
        const float gdot = -(two_over_B * r.m.y) * (rr - s);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e) g.v[k][e] = __fadd_rn(g.v[k][e], __fmul_rn(gdot, part.v[k][e]));
        bsum += (double)gdot;
        if (lig == 0 && !(__float_as_uint(r.m.x) & kSideBit)) {
          const double q = (double)rr - (double)s;
          acc_w += (double)r.m.y;
          acc_wr += (double)r.m.y * (double)rr;
          acc_wq += (double)r.m.y * q * q;
        }
      }
    };
    occ(cur, first);
    int64_t q = p + 1;
    auto rec_at = [&](int64_t qq) {  // record of a later occurrence of this run (own row's byte / bias: the run's)
      GloveRec r;
      r.id = id;
      r.m = meta[qq];
      r.y0 = cur.y0;
      r.b0 = cur.b0;
      const uint32_t partner = __float_as_uint(r.m.x) & kIdMask;
      r.y1 = loc[partner];
      r.b1 = bias[partner];
      return r;
    };
    for (; q + 4 <= e_run; q += 4) {  // four partner rows in flight
      const GloveRec g0 = rec_at(q), g1 = rec_at(q + 1), g2 = rec_at(q + 2), g3 = rec_at(q + 3);
      RowRegs<VEC, NCH> t0, t1, t2, t3;
      row_load(t0, part_row(g0), lig, G, nvec);
      row_load(t1, part_row(g1), lig, G, nvec);
      row_load(t2, part_row(g2), lig, G, nvec);
      row_load(t3, part_row(g3), lig, G, nvec);
      occ(g0, t0);
      occ(g1, t1);
      occ(g2, t2);
      occ(g3, t3);
    }
    for (; q < e_run; ++q) {
      const GloveRec gq = rec_at(q);
      RowRegs<VEC, NCH> t;
      row_load(t, part_row(gq), lig, G, nvec);
      occ(gq, t);
    }
    // (a run of one ends at p + 1, whose id is already in a register)
    const bool ends = q == n || (q == p + 1 ? nxt.id : (uint32_t)sorted_ids[q]) != id;
    if (lig == 0) bias_info[p] = make_double2(bsum, (double)(e_run - p));
    if (head) {  // (fuse_fin: no run outgrows its head chunk -- every head holds its whole run)
      if (h_p < 0) {
        h_p = p;
        h_id = id;
        h_bsum = bsum;
        h_cnt = (double)(e_run - p);
        h_w = cur.b0;  // the bias value read at the start: nobody writes the table before the grid-wide wait
        h_ac = (fuse_fin && lig == 0) ? bias_accum[id] : 0.f;
      } else {
        h_more = true;
      }
    }
    if (mode == ESR_GLOVE_REFERENCE) {  // G = -(2 / B) (A - sbar C): the statistics enter here
      need_stats();
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          g.v[k][e] = __fmul_rn(-two_over_B, __fsub_rn(g.v[k][e], __fmul_rn(sbar, gc.v[k][e])));
    }
    if (head && ends) {
      step_apply<VEC, NCH, TR>(emb0, emb1, loc, accum, code, T, own, a, g, D, lig, G, nvec, lr, eps);
    } else {  // a chunk of a long run: park the partial sum for glove_step_long_kernel
      const int64_t slot = 2 * (p / kStepChunk) + (head ? 1 : 0);
      row_store(g, chunk_rows + slot * D, lig, G, nvec);
      if (lig == 0) *parked = 1;  // (every writer stores the same value)
    }
  }
  if (threadIdx.x == 0) sm_last = 0;
  const double tw = block_sum_d(acc_w, sm);
  const double twr = block_sum_d(acc_wr, sm + 4);
  const double twq = block_sum_d(acc_wq, sm + 8);
  if (threadIdx.x == 0) {  // (block_sum_d hands the total to thread 0 only)
    sm[12] = tw;
    sm[13] = twr;
    sm[14] = twq;
  }
  __syncthreads();
  if (threadIdx.x < kFinAccs) fin_arrive(fin, threadIdx.x, sm[12 + (threadIdx.x >> 1)], gridDim.x);
  if (!fuse_fin) return;
  // ---- finalize inside this launch: every workgroup waits until the six accumulators are complete (the grid is one
  // resident wave-set -- the prologue's statistics rely on the same -- so everybody arrives), then steps the bias rows of
  // the runs ITS groups headed: their sums are its own, nothing crosses workgroups but the three totals.  The wait is
  // bounded: on a timeout the loss is poisoned instead of the queue hung.
  if (threadIdx.x < kFinAccs) {
    long long pay = 0;
    const bool ok = fin_poll(fin, threadIdx.x, gridDim.x, 300000000ull, &pay);  // 3 s
    sm[threadIdx.x] = (double)pay;
    if (!ok) sm_last = 1;  // (zeroed above, in front of the reductions' barriers)
  }
  __syncthreads();
  const double Sw = ldexp(sm[0], -8) + ldexp(sm[1], -50), Swr = ldexp(sm[2], -8) + ldexp(sm[3], -50),
               S3 = ldexp(sm[4], -8) + ldexp(sm[5], -50);
  const bool timed_out = sm_last != 0;
  const double Bd = (double)B;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const bool poisoned = timed_out || fin_poisoned(fin) ||
                          (mode == ESR_GLOVE_REFERENCE && coherent_load(reinterpret_cast<const unsigned*>(stat + 80)) != 0u);
    const double s1 = mode == ESR_GLOVE_REFERENCE ? fixed2_value(coherent_load(stat + 0), coherent_load(stat + 16)) : 0.0;
    const double s2 = mode == ESR_GLOVE_REFERENCE ? fixed2_value(coherent_load(stat + 32), coherent_load(stat + 48)) : 0.0;
    loss[0] = glove_loss_value(mode, Bd, Sw, glove_swq(mode, Bd, Sw, Swr, S3, s1), s1, s2, poisoned);
  }
  if (timed_out) {
    // this workgroup skips its bias rows: the step is incomplete WHICHEVER workgroup gave up, so the shared poison word
    // goes up (workgroup 0 reads it in front of its loss store; later launches -- consolidate, the next plan's
    // statistics -- see it too) and the loss is overwritten with NaN in case workgroup 0 has stored it already
    if (threadIdx.x == 0) {
      atomicOr(reinterpret_cast<unsigned*>(fin + kFinAccs * kFixAccWords + 16), 1u);
      __hip_atomic_store(reinterpret_cast<unsigned*>(loss), 0x7FC00000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const double kk = 2.0 / (Bd * Bd);
  if (lig == 0) {
    auto bias_step = [&](uint32_t id, double vx, double vy, float w, float ac) {
      const float gb = (float)((mode == ESR_GLOVE_REFERENCE) ? -kk * (vy * Swr - Sw * vx) : vx);
      adagrad_elem(w, ac, gb, lr, eps);
      bias_rw[id] = w;
      bias_accum[id] = ac;
    };
    // the first run this group headed: everything but the totals was in registers before the wait
    if (h_p >= 0) bias_step(h_id, h_bsum, h_cnt, h_w, h_ac);
    if (h_more) {  // further heads of the slice (several positions per group: larger batches): from memory
      uint32_t prev = (uint32_t)sorted_ids[h_p];
      for (int64_t p = h_p + 1; p < p_end; ++p) {
        const uint32_t id = (uint32_t)sorted_ids[p];
        const bool head = prev != id;
        prev = id;
        if (!head) continue;
        const double2 v = bias_info[p];
        bias_step(id, v.x, v.y, bias_rw[id], bias_accum[id]);
      }
    }
  }
}

// the versioned read-modify-write of one row with its summed gradient g: the row was read where `code` says it lives
// (`own`), its new value goes to the OTHER buffer and the byte flips (nobody reads `loc` during this launch: the plan
// kernel resolved every address).  `a` = the row's accumulator, loaded by the caller next to `own`.
template <int VEC, int NCH, class TR>
__device__ __forceinline__ void step_apply_resolved(TR* __restrict__ emb0, TR* __restrict__ emb1, uint8_t* __restrict__ loc,
                                           float* __restrict__ accum, uint32_t code, const RowRegs<VEC, NCH>& own,
                                           RowRegs<VEC, NCH>& a, const RowRegs<VEC, NCH>& g, int D, int lig, int G,
                                           int nvec, float lr, float eps) {
  const int64_t id = code & kIdMask;
  RowRegs<VEC, NCH> w = own;
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) adagrad_elem(w.v[k][e], a.v[k][e], g.v[k][e], lr, eps);
  row_store(a, accum + id * D, lig, G, nvec);
  row_store(w, ((code & kLocBit) ? emb0 : emb1) + id * D, lig, G, nvec);
  if (lig == 0) loc[id] = (code & kLocBit) ? 0 : 1;
}

// update, RESOLVED records (long lists; round 2's kernel as it was: the templated in-kernel-resolution variant below
// measured 0.169 ms per step at C3 against this one's 0.148 on the same box, alternating runs -- its third record in
// flight and the in-kernel statistics make it more sensitive to the sort that runs beside it on the second stream):
// the structure of segment_update_kernel (esr_optim.hip) with the gradient rows produced on the fly.
// A group's critical path per position is ONE memory round trip: the position's code and plan record are fetched one
// iteration ahead, and the own row, its accumulator and the first partner row are requested together (the accumulator
// speculatively: a chunk of a long run does not need it).  Written naively -- code, then own row and record, then
// partner row, then accumulator -- the same loop was four dependent round trips and ran at 3.2 TB/s.
template <int VEC, int NCH, class TR>
__global__ __launch_bounds__(kBlock) void glove_step_resolved_kernel(
    TR* __restrict__ emb0, TR* __restrict__ emb1, uint8_t* __restrict__ loc, float* __restrict__ accum, int D,
    int G, const uint32_t* __restrict__ own_code, const float4* __restrict__ meta, int64_t n, int64_t B, int mode,
    int nstat, const double* __restrict__ stat_part, float lr, float eps, float* __restrict__ chunk_rows,
    double2* __restrict__ bias_info, double* __restrict__ pair_part, int* __restrict__ long_flag,
    uint32_t* __restrict__ start_flag, uint32_t start_value) {
  __shared__ double sm[16];
  announce_start(start_flag, start_value);
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int nvec = D / VEC;
  const int64_t per = (n + ngroups - 1) / ngroups;  // contiguous slices (see segment_update_kernel)
  const int64_t p_begin = group * per, p_end = min(n, (group + 1) * per);
  // the records of the first two positions are requested before the statistics are reduced
  uint32_t c0 = 0, c1 = 0, prev_n = 0xFFFFFFFFu;
  float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
  if (p_begin < p_end) {
    c0 = own_code[p_begin];
    if (p_begin > 0) prev_n = own_code[p_begin - 1];
    m0 = meta[p_begin];
    if (p_begin + 1 < n) {
      c1 = own_code[p_begin + 1];
      m1 = meta[p_begin + 1];
    }
  }
  double sum_s = 0.0, sum_s2 = 0.0;
  if (mode == ESR_GLOVE_REFERENCE) reduce_stat_parts(stat_part, nstat, sm, &sum_s, &sum_s2);
  const float sbar = (float)(sum_s / (double)B);
  const float two_over_B = 2.0f / (float)B;
  double acc_w = 0.0, acc_wr = 0.0, acc_wq = 0.0;
  auto part_ptr = [&](const float4& m) {
    const uint32_t c = __float_as_uint(m.x);
    return ((c & kLocBit) ? emb1 : emb0) + (int64_t)(c & kIdMask) * D;
  };
  // rows of the NEXT position, requested before this position is computed (records run two positions ahead, rows one):
  // a group always has one position's three rows in flight while it works on another
  bool have_next = false;
  RowRegs<VEC, NCH> nown, nfirst, na;

  for (int64_t p = p_begin; p < p_end; ++p) {
    const uint32_t code = c0, prev = prev_n, code_n = c1;
    const float4 m_first = m0, m_next = m1;
    const bool more = p + 1 < n;
    c0 = c1;
    m0 = m1;
    if (p + 2 < n) {
      c1 = own_code[p + 2];
      m1 = meta[p + 2];
    }
    prev_n = code;
    const uint32_t id = code & kIdMask;
    const bool head = (prev & kIdMask) != id;  // prev = all ones at p == 0: no id equals kIdMask
    if (!head && ((p & (kStepChunk - 1)) != 0 || (own_code[p - kStepChunk] & kIdMask) != id)) continue;
    const int64_t stop = min(head ? ((p + 2 * kStepChunk - 1) / kStepChunk) * kStepChunk : p + kStepChunk, n);
    RowRegs<VEC, NCH> own, a, g, first;
    if (have_next) {
      own = nown;
      first = nfirst;
      a = na;
    } else {
      row_load(own, ((code & kLocBit) ? emb1 : emb0) + (int64_t)id * D, lig, G, nvec);
      row_load(first, part_ptr(m_first), lig, G, nvec);
      row_load(a, accum + (int64_t)id * D, lig, G, nvec);
    }
    have_next = false;
    int64_t e_run = p + 1;
    if (more && (code_n & kIdMask) == id) {  // a run of several occurrences: find the end of this chunk
      ++e_run;
      while (e_run < stop && (own_code[e_run] & kIdMask) == id) ++e_run;
      if (e_run > stop) e_run = stop;
    } else if (p + 1 < p_end) {  // the usual case, a run of one: position p + 1 heads the next run -- request its rows now
      row_load(nown, ((code_n & kLocBit) ? emb1 : emb0) + (int64_t)(code_n & kIdMask) * D, lig, G, nvec);
      row_load(nfirst, part_ptr(m_next), lig, G, nvec);
      row_load(na, accum + (int64_t)(code_n & kIdMask) * D, lig, G, nvec);
      have_next = true;
    }
    row_zero(g);
    double bsum = 0.0;  // fp64: a hot token's bias gradient is a sum over thousands of occurrences
    // one occurrence: gdot from the dot with its partner row; G += gdot * partner, the product rounded to f32 first
    // (it used to be stored as a gradient row) and the additions strictly left to right
    auto occ = [&](const float4& m, const RowRegs<VEC, NCH>& part) {
      const float dot = group_sum(row_dot_partial(own, part), G);
      const float r = m.z - dot;
      const float center = (mode == ESR_GLOVE_REFERENCE) ? sbar : m.w;
      const float gdot = -(two_over_B * m.y) * (r - center);
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) g.v[k][e] = __fadd_rn(g.v[k][e], __fmul_rn(gdot, part.v[k][e]));
      bsum += (double)((mode == ESR_GLOVE_REFERENCE) ? m.w : gdot);
      if (lig == 0 && !(__float_as_uint(m.x) & kSideBit)) {  // every pair is seen from both sides: count it once
        const double q = (double)r - (double)center;
        acc_w += (double)m.y;
        acc_wr += (double)m.y * (double)r;
        acc_wq += (double)m.y * q * q;
      }
    };
    occ(m_first, first);
    int64_t q = p + 1;
    for (; q + 4 <= e_run; q += 4) {  // four partner rows in flight
      const float4 m0 = meta[q], m1 = meta[q + 1], m2 = meta[q + 2], m3 = meta[q + 3];
      RowRegs<VEC, NCH> t0, t1, t2, t3;
      row_load(t0, part_ptr(m0), lig, G, nvec);
      row_load(t1, part_ptr(m1), lig, G, nvec);
      row_load(t2, part_ptr(m2), lig, G, nvec);
      row_load(t3, part_ptr(m3), lig, G, nvec);
      occ(m0, t0);
      occ(m1, t1);
      occ(m2, t2);
      occ(m3, t3);
    }
    for (; q < e_run; ++q) {
      const float4 m = meta[q];
      RowRegs<VEC, NCH> t;
      row_load(t, part_ptr(m), lig, G, nvec);
      occ(m, t);
    }
    // (a run of one ends at p + 1, whose code is already in a register: the reload was a dependent round trip per
    // position in front of the stores -- +7 % on the whole step.  The same shortcut for meta[p + 1] and own_code[p + 2]
    // in the walk of longer runs measured 5 % SLOWER here (same box, alternating runs) and is not used)
    const bool ends = q == n || ((q == p + 1 ? code_n : own_code[q]) & kIdMask) != id;
    if (lig == 0) bias_info[p] = make_double2(bsum, (double)(e_run - p));
    if (head && ends) {
      step_apply_resolved<VEC, NCH, TR>(emb0, emb1, loc, accum, code, own, a, g, D, lig, G, nvec, lr, eps);
    } else {  // a chunk of a long run: park the partial sum for glove_step_long_kernel
      const int64_t slot = 2 * (p / kStepChunk) + (head ? 1 : 0);
      row_store(g, chunk_rows + slot * D, lig, G, nvec);
      if (lig == 0) *long_flag = 1;  // (every writer stores the same value)
    }
  }
  const double tw = block_sum_d(acc_w, sm);
  const double twr = block_sum_d(acc_wr, sm + 4);
  const double twq = block_sum_d(acc_wq, sm + 8);
  if (threadIdx.x == 0) {
    pair_part[3 * blockIdx.x] = tw;
    pair_part[3 * blockIdx.x + 1] = twr;
    pair_part[3 * blockIdx.x + 2] = twq;
  }
}

// long: segment_long_kernel's screening and fixed-order combination over the parked chunk partials; also folds the
// chunks' bias sums into the head's bias_info entry.  Launched only when a run may have outgrown its head chunk.
template <int VEC, int NCH, class TR>
__global__ __launch_bounds__(kBlock) void glove_step_long_kernel(
    TR* __restrict__ emb0, TR* __restrict__ emb1, uint8_t* __restrict__ loc, float* __restrict__ accum, int D,
    int G, const int32_t* __restrict__ sorted_ids, int64_t n, uint32_t T, float lr, float eps,
    const float* __restrict__ chunk_rows, double2* __restrict__ bias_info, const int* __restrict__ parked, int npair,
    const double* __restrict__ pair_part, double* __restrict__ pair_tot) {
  // resolved mode (this launch is always made there): the last workgroup reduces the update kernel's loss partials
  // (fixed order) to three doubles, so that the finalize kernel's workgroups read three numbers instead of re-reducing
  // a thousand partials each
  if (pair_tot && blockIdx.x == gridDim.x - 1) {
    __shared__ double smp[12];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < npair; i += kBlock) {
      a += pair_part[3 * i];
      b += pair_part[3 * i + 1];
      c += pair_part[3 * i + 2];
    }
    const double tw = block_sum_d(a, smp);
    const double twr = block_sum_d(b, smp + 4);
    const double twq = block_sum_d(c, smp + 8);
    if (threadIdx.x == 0) {
      pair_tot[0] = tw;
      pair_tot[1] = twr;
      pair_tot[2] = twq;
    }
  }
  // no run of the batch outgrew its head chunk: nothing to combine -- one load instead of screening the chunk boundaries
  if (*parked == 0) return;
  __shared__ float red[kBlock * VEC * NCH];
  __shared__ double smd[8];
  constexpr int kPass = 4;
  __shared__ long long s_long[kPass];
  __shared__ int s_nlong, s_hoff;
  const int tid = threadIdx.x, lig = tid & (G - 1), gidx = tid / G, NG = kBlock / G;
  const int nvec = D / VEC;
  auto id_at = [&](int64_t pos) { return (uint32_t)sorted_ids[pos]; };
  const int64_t nbound = (n - 1) / kStepChunk;
  for (int64_t b0 = (int64_t)blockIdx.x * kPass; b0 < nbound; b0 += (int64_t)gridDim.x * kPass) {
    __syncthreads();
    if (tid == 0) s_nlong = 0;
    __syncthreads();
    {
      const int64_t Bd = (b0 + tid + 1) * kStepChunk;
      if (tid < kPass && b0 + tid < nbound) {
        const uint32_t id_b = id_at(Bd);
        const bool first = Bd < 2 * kStepChunk || id_at(Bd - 2 * kStepChunk) != id_b;
        if (id_at(Bd - kStepChunk) == id_b && first) s_long[atomicAdd(&s_nlong, 1)] = Bd;
      }
    }
    __syncthreads();
    const int nlong = s_nlong;
    for (int li = 0; li < nlong; ++li) {
      const int64_t nxt = s_long[li];
      const uint32_t id = id_at(nxt);
      const int64_t win = max<int64_t>(nxt - 2 * kStepChunk + 1, 0);
      if (tid < 64) {
        const int64_t pos = win + tid;
        const bool is_head = pos <= nxt - kStepChunk && id_at(pos) == id && (pos == 0 || id_at(pos - 1) != id);
        const unsigned long long m = __ballot(is_head);
        if (tid == 0) s_hoff = __ffsll((long long)m) - 1;
      }
      __syncthreads();
      const int64_t h = win + s_hoff;
      int64_t K = 0;  // continuation chunks
      for (int64_t k0 = 0;; k0 += kBlock) {
        const int64_t pos = nxt + (k0 + tid) * kStepChunk;
        const int cnt = __syncthreads_count(pos < n && id_at(pos) == id);
        K += cnt;
        if (cnt < kBlock) break;
      }
      auto part_row = [&](int64_t i) {
        return (i == 0 ? 2 * (h / kStepChunk) + 1 : 2 * ((nxt + (i - 1) * kStepChunk) / kStepChunk)) * (int64_t)D;
      };
      auto part_pos = [&](int64_t i) { return i == 0 ? h : nxt + (i - 1) * kStepChunk; };
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
      // the run's bias sums: chunk entries added in a fixed tree (fp64 carries them exactly enough to be order-free)
      double bs = 0.0, bc = 0.0;
      for (int64_t c = tid; c <= K; c += kBlock) {
        const double2 v = bias_info[part_pos(c)];
        bs += v.x;
        bc += v.y;
      }
      const double tbs = block_sum_d(bs, smd);  // (its barriers also publish `red`)
      const double tbc = block_sum_d(bc, smd + 4);
      if (tid == 0) bias_info[h] = make_double2(tbs, tbc);
      if (gidx == 0) {
        const int used = (int)min<int64_t>(NG, K + 1);
        for (int gg = 1; gg < used; ++gg)
#pragma unroll
          for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc.v[k][e] += red[((gg * G + lig) * NCH + k) * VEC + e];
        // (nobody has rewritten this row during the step: its head parked its partial instead)
        const uint32_t code = id | (loc_at_step_begin(loc[id], T) ? kLocBit : 0u);
        RowRegs<VEC, NCH> own, a;
        row_load(own, ((code & kLocBit) ? emb1 : emb0) + (int64_t)id * D, lig, G, nvec);
        row_load(a, accum + (int64_t)id * D, lig, G, nvec);
        step_apply<VEC, NCH, TR>(emb0, emb1, loc, accum, code, T, own, a, acc, D, lig, G, nvec, lr, eps);
      }
      __syncthreads();
    }
  }
}

// finalize: the loss (same formula as glove_finalize_kernel) and the bias table's Adagrad step, one thread per sorted
// position; a run head holds its run's sums.  Reference mode: sum over the run of dL/ds = -(2/B^2) (cnt * Swr - Sw *
// sum s); diagonal mode: the sum of gdot itself.  Every workgroup re-reduces the update kernel's <= 2048 x 3 loss
// partials in the same fixed order (48 KB out of L2), so it depends on no other launch.
__global__ __launch_bounds__(kBlock) void glove_step_finalize_kernel(
    int64_t B, int mode, const unsigned long long* __restrict__ stat, int nstat, const double* __restrict__ stat_part,
    int npair, const double* __restrict__ pair_part, const double* __restrict__ pair_tot,
    const unsigned long long* __restrict__ fin, const int32_t* __restrict__ sorted_ids,
    const double2* __restrict__ bias_info, float* __restrict__ bias, float* __restrict__ bias_accum, float lr, float eps,
    float* __restrict__ loss) {
  // fin (short lists): the update kernel left the three loss sums in its integer accumulators (fin_arrive) -- the same
  // totals its own last workgroup uses when it is the finalize step itself
  __shared__ double sm[16];
  // this thread's position: everything it needs from memory is requested before the reductions below
  const int64_t n = 2 * B;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  uint32_t id = 0;
  bool head = false;
  double2 v = make_double2(0.0, 0.0);
  float w = 0.f, acc = 0.f;
  if (p < n) {
    id = (uint32_t)sorted_ids[p];
    head = p == 0 || (uint32_t)sorted_ids[p - 1] != id;
    if (head) {
      v = bias_info[p];
      w = bias[id];
      acc = bias_accum[id];
    }
  }
  double Sw, Swr, Swq;
  if (fin) {
    Sw = fin_value(fin, 0);
    Swr = fin_value(fin, 1);
    Swq = glove_swq(mode, (double)B, Sw, Swr, fin_value(fin, 2),
                    mode == ESR_GLOVE_REFERENCE ? fixed2_value(stat[0], stat[16]) : 0.0);
  } else if (pair_tot) {  // reduced by the long-run launch
    Sw = pair_tot[0];
    Swr = pair_tot[1];
    Swq = pair_tot[2];
  } else {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < npair; i += kBlock) {
      a += pair_part[3 * i];
      b += pair_part[3 * i + 1];
      c += pair_part[3 * i + 2];
    }
    const double tw = block_sum_d(a, sm);
    const double twr = block_sum_d(b, sm + 4);
    const double twq = block_sum_d(c, sm + 8);
    if (threadIdx.x == 0) {
      sm[12] = tw;
      sm[13] = twr;
      sm[14] = twq;
    }
    __syncthreads();
    Sw = sm[12];
    Swr = sm[13];
    Swq = sm[14];
  }
  const double Bd = (double)B;
  double rs = 0.0, rs2 = 0.0;
  if (blockIdx.x == 0 && stat_part && mode == ESR_GLOVE_REFERENCE)  // resolved mode: K_A's partials, K_A's order
    reduce_stat_parts(stat_part, nstat, sm, &rs, &rs2);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double L;
    if (mode == ESR_GLOVE_REFERENCE) {
      const bool poisoned = (!stat_part && *reinterpret_cast<const unsigned*>(stat + 80) != 0u) || (fin && fin_poisoned(fin));
      const double sum_s = stat_part ? rs : fixed2_value(stat[0], stat[16]);
      const double sum_s2 = stat_part ? rs2 : fixed2_value(stat[32], stat[48]);
      double SS = sum_s2 - sum_s * sum_s / Bd;
      if (SS < 0.0) SS = 0.0;
      L = poisoned ? __builtin_nan("") : (Bd * Swq + Sw * SS) / (Bd * Bd);
    } else {
      L = (fin && fin_poisoned(fin)) ? __builtin_nan("") : Swq / Bd;
    }
    loss[0] = (float)L;
  }
  if (head) {
    const double k = 2.0 / (Bd * Bd);
    const float gb = (float)((mode == ESR_GLOVE_REFERENCE) ? -k * (v.y * Swr - Sw * v.x) : v.x);
    adagrad_elem(w, acc, gb, lr, eps);
    bias[id] = w;
    bias_accum[id] = acc;
  }
}

// rows whose byte has bit 0 set live in `shadow`: copy them back into `primary`; every byte is cleared (T = float4 or float)
template <typename T>
__global__ __launch_bounds__(kBlock) void rows_consolidate_kernel(T* __restrict__ primary, const T* __restrict__ shadow,
                                                                 uint8_t* __restrict__ loc, int64_t V, int nchunk,
                                                                 int G) {
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  for (int64_t r = group; r < V; r += ngroups) {
    const uint8_t b = loc[r];
    if (!b) continue;
    if (b & 1)
      for (int c = lig; c < nchunk; c += G) primary[r * nchunk + c] = shadow[r * nchunk + c];
    if (lig == 0) loc[r] = 0;  // (the stamp goes too)
  }
}

static int check_dim(const char* who, int D) {
  const RowGeom g = row_geom(D);
  if (g.nch > kMaxChunksPerLane) {
    set_error("%s: D=%d not supported (max 1024 when D %% 4 == 0, else 256)", who, D);
    return ESR_EINVAL;
  }
  return ESR_OK;
}

}  // namespace esr

using namespace esr;

extern "C" {

size_t esr_glove_workspace_bytes(int64_t B) {
  if (B < 0) return 0;
  return glove_ws_layout(B, nullptr, nullptr);
}

int esr_glove_forward(const float* emb, const float* bias, int64_t V, int D, const int32_t* inputs,
                      int64_t B, float* dot, float* s, esr_stream_t stream) {
  ESR_REQUIRE(B >= 0 && V > 0 && D > 0, "esr_glove_forward: bad sizes V=%lld D=%d B=%lld", (long long)V, D,
              (long long)B);
  if (B == 0) return ESR_OK;
  ESR_REQUIRE(emb && bias && inputs && dot && s, "esr_glove_forward: null pointer");
  if (int rc = check_dim("esr_glove_forward", D)) return rc;
  hipStream_t st = as_stream(stream);
  const RowGeom g = row_geom(D);
  const int nstat = (int)std::min<int64_t>(kStatBlocks, cdiv(B, kBlock));
  const int npair = (int)std::min<int64_t>(kPairBlocks, cdiv(B, kBlock / g.G));
  hipLaunchKernelGGL(glove_bias_stats_kernel, dim3(nstat), dim3(kBlock), 0, st, bias, inputs, B, s,
                     (double*)nullptr);
  ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((glove_pairs_kernel<VEC, NCH, false, false>), dim3(npair),
                                         dim3(kBlock), 0, st, emb, inputs, (const float*)nullptr, B, D,
                                         g.G, ESR_GLOVE_DIAGONAL, 0, (const double*)nullptr,
                                         (const float*)nullptr, dot, (float*)nullptr, (float*)nullptr,
                                         (double*)nullptr));
  return check_launch("esr_glove_forward");
}

int esr_glove_fwd_bwd(const float* emb, const float* bias, int64_t V, int D, const int32_t* inputs,
                      const float* target, int64_t B, int mode, float* loss, float* grad_rows,
                      float* grad_bias, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_glove_fwd_bwd");
  ESR_REQUIRE(B > 0 && V > 0 && D > 0, "esr_glove_fwd_bwd: bad sizes V=%lld D=%d B=%lld", (long long)V, D,
              (long long)B);
  ESR_REQUIRE((mode & ~ESR_GRADS_AT_IDS) == ESR_GLOVE_REFERENCE || (mode & ~ESR_GRADS_AT_IDS) == ESR_GLOVE_DIAGONAL,
              "esr_glove_fwd_bwd: bad mode %d", mode);
  ESR_REQUIRE(emb && bias && inputs && target && loss, "esr_glove_fwd_bwd: null pointer");
  ESR_REQUIRE((grad_rows == nullptr) == (grad_bias == nullptr),
              "esr_glove_fwd_bwd: grad_rows and grad_bias must both be set or both be NULL");
  if (int rc = check_dim("esr_glove_fwd_bwd", D)) return rc;
  if (!workspace || workspace_bytes < esr_glove_workspace_bytes(B) || ((uintptr_t)workspace & 15)) {
    set_error("esr_glove_fwd_bwd: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_glove_workspace_bytes(B));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  GloveWs ws;
  glove_ws_layout(B, (char*)workspace, &ws);
  const RowGeom g = row_geom(D);
  const int nstat = (int)std::min<int64_t>(kStatBlocks, cdiv(B, kBlock));
  const int npair = (int)std::min<int64_t>(kPairBlocks, cdiv(B, kBlock / g.G));
  hipLaunchKernelGGL(glove_bias_stats_kernel, dim3(nstat), dim3(kBlock), 0, st, bias, inputs, B, ws.s,
                     ws.stat_part);
  if (grad_rows) {
    ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((glove_pairs_kernel<VEC, NCH, true, true>), dim3(npair),
                                           dim3(kBlock), 0, st, emb, inputs, target, B, D, g.G, mode, nstat,
                                           (const double*)ws.stat_part, (const float*)ws.s, (float*)nullptr,
                                           grad_rows, grad_bias, ws.pair_part));
  } else {
    ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((glove_pairs_kernel<VEC, NCH, true, false>), dim3(npair),
                                           dim3(kBlock), 0, st, emb, inputs, target, B, D, g.G, mode, nstat,
                                           (const double*)ws.stat_part, (const float*)ws.s, (float*)nullptr,
                                           (float*)nullptr, (float*)nullptr, ws.pair_part));
  }
  const int nfin = (int)std::min<int64_t>(256, cdiv(B, kBlock));
  hipLaunchKernelGGL(glove_finalize_kernel, dim3(nfin), dim3(kBlock), 0, st, B, mode, nstat, npair,
                     (const double*)ws.stat_part, (const double*)ws.pair_part, (const float*)ws.s, inputs, loss,
                     grad_bias);
  return check_launch("esr_glove_fwd_bwd");
}

size_t esr_glove_step_workspace_bytes(int64_t B, int D) {
  if (B <= 0 || D <= 0) return 0;
  return step_ws_layout(B, D, nullptr, nullptr);
}

size_t esr_glove_plan_bytes(int64_t B) {
  if (B <= 0) return 0;
  return glove_plan_layout(B, nullptr, nullptr);
}

static void launch_glove_plan(const int32_t* const* inputs, const float* const* targets, int nbatch,
                              const int32_t* sorted_ids, const int32_t* perm, int64_t B, char* plans, size_t stride,
                              int* hints, int gen, hipStream_t st) {
  GlovePlanBatch pb;
  for (int b = 0; b < kMaxGlovePlanBatch; ++b) {
    pb.inputs[b] = inputs[b < nbatch ? b : 0];
    pb.target[b] = targets[b < nbatch ? b : 0];
  }
  const int gx = (int)std::min<int64_t>(kMaxGrid, cdiv(2 * B, kBlock));
  hipLaunchKernelGGL(glove_plan_kernel, dim3(gx, nbatch), dim3(kBlock), 0, st, pb, sorted_ids, perm, B, plans, stride,
                     hints, gen);
}

// hint[0] = gen when sorted_ids has a run of equal ids longer than `chunk` positions
__global__ __launch_bounds__(kBlock) void long_run_hint_kernel(const int32_t* __restrict__ sorted_ids, int64_t n, int chunk,
                                                              int* __restrict__ hint, int gen) {
  bool long_run = false;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p + chunk < n; p += (int64_t)gridDim.x * kBlock)
    if (sorted_ids[p] == sorted_ids[p + chunk]) long_run = true;
  if (__any(long_run) && (threadIdx.x & 63) == 0) hint[0] = gen;
}

int esr_long_run_hint(const int32_t* sorted_ids, int64_t n, int chunk, int32_t* hint, int32_t gen, esr_stream_t stream) {
  ESR_REQUIRE(n >= 0 && chunk > 0 && hint && (n == 0 || sorted_ids), "esr_long_run_hint: bad arguments");
  if (n <= chunk) return ESR_OK;
  const int grid = (int)std::min<int64_t>(kMaxGrid, cdiv(n, kBlock));
  hipLaunchKernelGGL(long_run_hint_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), sorted_ids, n, chunk, hint, gen);
  return check_launch("esr_long_run_hint");
}

int esr_glove_plan(const int32_t* const* inputs, const float* const* targets, int nbatch, int64_t B,
                   const int32_t* sorted_ids, const int32_t* perm, void* plans, int32_t* hints, int32_t gen,
                   esr_stream_t stream) {
  TraceScope trace_scope_("esr_glove_plan");
  ESR_REQUIRE(nbatch >= 1 && nbatch <= kMaxGlovePlanBatch && B > 0 && 2 * B < ((int64_t)1 << 31),
              "esr_glove_plan: nbatch=%d not in [1, %d] or bad B=%lld", nbatch, kMaxGlovePlanBatch, (long long)B);
  ESR_REQUIRE(inputs && targets && sorted_ids && perm && plans && !((uintptr_t)plans & 255),
              "esr_glove_plan: null or misaligned pointer");
  for (int i = 0; i < nbatch; ++i) ESR_REQUIRE(inputs[i] && targets[i], "esr_glove_plan: null list %d", i);
  launch_glove_plan(inputs, targets, nbatch, sorted_ids, perm, B, (char*)plans, esr_glove_plan_bytes(B), hints, gen,
                    as_stream(stream));
  return check_launch("esr_glove_plan");
}

}  // extern "C"

struct GloveTables {
  void* emb;         // f32 or bf16 rows (dtype)
  void* emb_shadow;
  uint8_t* emb_loc;
  float* emb_accum;
  float* bias;
  float* bias_accum;
  int D;
  int dtype;
};

// one step's launches (arguments validated by the callers); T: the embedding rows' element type (GloveTables::dtype)
template <class T>
static void launch_glove_step_t(const GloveTables& t, const int32_t* inputs, const float* target, int64_t B, int mode,
                              float lr, float eps, uint32_t stamp, const int32_t* sorted_ids, const int32_t* perm,
                              void* plan, int long_runs, int blocks_per_cu, uint32_t* start_flag,
                              uint32_t start_value, float* loss, const StepWs& ws, hipStream_t st) {
  const int64_t n = 2 * B;
  const int D = t.D;
  // bf16 rows: 8 elements (16 bytes) per lane when the width allows (row_geom8)
  // -- ESR_BF16_VEC8=1 only (150 registers, three waves per SIMD instead of four: see esr_triplet_step.hip)
  const char* v8 = getenv("ESR_BF16_VEC8");
  const bool vec8 = v8 && v8[0] == '1' && sizeof(T) == 2 && D % 8 == 0 && !(((uintptr_t)t.emb | (uintptr_t)t.emb_shadow) & 15);
  const RowGeom g = vec8 ? row_geom8(D) : row_geom(D);
  int grid = grid_for_groups(n, g.G);
  const int grid2 = (int)std::min<int64_t>(kMaxGrid, cdiv(cdiv(n, kStepChunk), 4));
  const int nfin = (int)cdiv(n, kBlock);  // one thread per sorted position
  if (n > kResolveMinIds) {
    // long list: a resolve launch (round 2's plan kernel: row codes with the location bits, w, log10(1 + c), s and K_A's
    // statistics) in front of the update kernel, which then walks resolved records -- see kResolveMinIds.  A plan made
    // ahead is not used here (its records carry no locations).
    const int nstat = (int)std::min<int64_t>(kStatBlocks, cdiv(B, kBlock));
    const int nres = (int)std::max<int64_t>(nstat, std::min<int64_t>(kMaxGrid, cdiv(n, kBlock)));
    ESR_KT("glove_resolve_kernel", st, hipLaunchKernelGGL(glove_resolve_kernel, dim3(nres), dim3(kBlock), 0, st, sorted_ids, perm, inputs, target,
                       (const float*)t.bias, (const uint8_t*)t.emb_loc, B, nstat, ws.own_code, ws.meta_res, ws.stat_part,
                       ws.res_flags));
    // start_flag: the update kernel's first workgroup stores start_value there as it starts.  A second stream gated on
    // the word (esr_stream_gate) is released as the update kernel runs, so what it brings (the id sort of a coming
    // batch) arrives after that kernel has taken its wave slots.  Arriving first, the sort's workgroups kept part of
    // the update kernel's single resident wave-set waiting for a slot: 126 against 109 us for the same kernel.  (An
    // event recorded here did the same job, but the marker cost the main queue ~7 us between resolve and update.)
    ESR_DISPATCH_ROW_ANY(g, { if constexpr (VEC != 8 || sizeof(T) == 2) {
      static const int resident_all = resident_blocks((const void*)glove_step_resolved_kernel<VEC, NCH, T>, 0);
      const int resident = blocks_per_cu > 0
                               ? resident_blocks((const void*)glove_step_resolved_kernel<VEC, NCH, T>, blocks_per_cu)
                               : resident_all;
      grid = std::min(grid, resident);
      ESR_KT("glove_step_resolved_kernel", st, hipLaunchKernelGGL((glove_step_resolved_kernel<VEC, NCH, T>), dim3(grid), dim3(kBlock), 0, st, (T*)t.emb, (T*)t.emb_shadow,
                         t.emb_loc, t.emb_accum, D, g.G, (const uint32_t*)ws.own_code, (const float4*)ws.meta_res, n, B,
                         mode, nstat, (const double*)ws.stat_part, lr, eps, ws.chunk_rows, ws.bias_info, ws.pair_part,
                         ws.res_flags, start_flag, start_value));
      // (its last workgroup also reduces the loss partials for the finalize kernel; when the caller knows that no run
      // outgrows its head chunk -- long_runs == 0 -- it is skipped and every finalize workgroup reduces them itself,
      // the same sums in the same order)
      if (long_runs != 0)
        ESR_KT("glove_step_long_kernel", st, hipLaunchKernelGGL((glove_step_long_kernel<VEC, NCH, T>), dim3(grid2), dim3(kBlock), 0, st, (T*)t.emb, (T*)t.emb_shadow,
                           t.emb_loc, t.emb_accum, D, g.G, sorted_ids, n, stamp, lr, eps, (const float*)ws.chunk_rows,
                           ws.bias_info, (const int*)ws.res_flags, grid, (const double*)ws.pair_part, ws.pair_tot));
    }});
    ESR_KT("glove_step_finalize_kernel", st, hipLaunchKernelGGL(glove_step_finalize_kernel, dim3(nfin), dim3(kBlock), 0, st, B, mode,
                       (const unsigned long long*)nullptr, nstat, (const double*)ws.stat_part, grid,
                       (const double*)ws.pair_part, long_runs != 0 ? (const double*)ws.pair_tot : (const double*)nullptr,
                       (const unsigned long long*)nullptr, sorted_ids, (const double2*)ws.bias_info, t.bias, t.bias_accum,
                       lr, eps, loss));
    return;
  }
  if (!plan) {  // no plan made ahead: make it here (and nobody told us whether a run is long: screen for it)
    launch_glove_plan(&inputs, &target, 1, sorted_ids, perm, B, ws.plan, 0, nullptr, 0, st);
    plan = ws.plan;
    long_runs = -1;
  }
  GlovePlan pl;
  glove_plan_layout(B, (char*)plan, &pl);
  const int nstat = mode == ESR_GLOVE_REFERENCE ? (int)std::min<int64_t>(kStatBlocksStep, cdiv(B, kBlock)) : 0;
  // no run outgrows its head chunk (the plan's hint) and the list is short: the update kernel's last workgroup is the
  // finalize step (ESR_GLOVE_FIN_FUSED=0 keeps the launch)
  const char* fin_env = getenv("ESR_GLOVE_FIN_FUSED");
  const bool fin_fused_on = !(fin_env && fin_env[0] == '0');
  const bool fuse_fin = fin_fused_on && long_runs == 0 && n <= kFinFuseMaxIds;
  ESR_DISPATCH_ROW_ANY(g, { if constexpr (VEC != 8 || sizeof(T) == 2) {
    // one resident wave-set: every group walks a contiguous slice, so a grid larger than what the chip holds at once
    // only adds a second, partly filled round (94 VGPRs -> 5 blocks per CU: 2048 blocks ran as 1280 + 768); the
    // prologue's wait also relies on the workgroups it waits for having been dispatched (they have: index order)
    static const int resident_all = resident_blocks((const void*)glove_step_kernel<VEC, NCH, T>, 0);  // (one query)
    const int resident = blocks_per_cu > 0 ? resident_blocks((const void*)glove_step_kernel<VEC, NCH, T>, blocks_per_cu)
                                           : resident_all;
    grid = std::min(grid + nstat, std::max(resident, nstat + 1));  // nstat statistics-only workgroups in front
    ESR_KT("glove_step_kernel", st, hipLaunchKernelGGL((glove_step_kernel<VEC, NCH, T>), dim3(grid), dim3(kBlock), 0, st, (T*)t.emb, (T*)t.emb_shadow, t.emb_loc,
                       t.emb_accum, (const float*)t.bias, D, g.G, sorted_ids, (const float4*)pl.meta, inputs, n, B, mode,
                       stamp, nstat, pl.stat, lr, eps, ws.chunk_rows, ws.bias_info, pl.fin, fuse_fin ? 1 : 0, t.bias,
                       t.bias_accum, loss, pl.flags, start_flag, start_value));
    if (long_runs != 0)  // 0 = the caller knows (esr_glove_plan's hint) that no run outgrows its head chunk
      ESR_KT("glove_step_long_kernel", st, hipLaunchKernelGGL((glove_step_long_kernel<VEC, NCH, T>), dim3(grid2), dim3(kBlock), 0, st, (T*)t.emb, (T*)t.emb_shadow,
                         t.emb_loc, t.emb_accum, D, g.G, sorted_ids, n, stamp, lr, eps, (const float*)ws.chunk_rows,
                         ws.bias_info, (const int*)pl.flags, 0, (const double*)nullptr, (double*)nullptr));
  }});
  if (!fuse_fin)
    ESR_KT("glove_step_finalize_kernel", st, hipLaunchKernelGGL(glove_step_finalize_kernel, dim3(nfin), dim3(kBlock), 0, st, B, mode,
                       (const unsigned long long*)pl.stat, 0, (const double*)nullptr, grid, (const double*)nullptr,
                       (const double*)nullptr, (const unsigned long long*)pl.fin, sorted_ids,
                       (const double2*)ws.bias_info, t.bias, t.bias_accum, lr, eps, loss));
}

static void launch_glove_step(const GloveTables& t, const int32_t* inputs, const float* target, int64_t B, int mode,
                              float lr, float eps, uint32_t stamp, const int32_t* sorted_ids, const int32_t* perm,
                              void* plan, int long_runs, int blocks_per_cu, uint32_t* start_flag,
                              uint32_t start_value, float* loss, const StepWs& ws, hipStream_t st) {
  if (t.dtype == ESR_BF16)
    launch_glove_step_t<uint16_t>(t, inputs, target, B, mode, lr, eps, stamp, sorted_ids, perm, plan, long_runs,
                                  blocks_per_cu, start_flag, start_value, loss, ws, st);
  else
    launch_glove_step_t<float>(t, inputs, target, B, mode, lr, eps, stamp, sorted_ids, perm, plan, long_runs,
                               blocks_per_cu, start_flag, start_value, loss, ws, st);
}

#define ESR_GLOVE_STEP_CHECKS(who)                                                                                    \
  ESR_REQUIRE(B > 0 && V > 0 && D > 0, who ": bad sizes V=%lld D=%d B=%lld", (long long)V, D, (long long)B);          \
  ESR_REQUIRE(V <= (int64_t)kIdMask, who ": V=%lld exceeds 2^30 - 1 rows", (long long)V);                             \
  ESR_REQUIRE(2 * B < ((int64_t)1 << 31), who ": B=%lld too large", (long long)B);                                    \
  ESR_REQUIRE(mode == ESR_GLOVE_REFERENCE || mode == ESR_GLOVE_DIAGONAL, who ": bad mode %d", mode);                  \
  ESR_REQUIRE(emb && emb_shadow && emb_loc && emb_accum && bias && bias_accum, who ": null table pointer");           \
  ESR_REQUIRE(emb != emb_shadow, who ": the shadow table must be a second buffer");                                   \
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, who ": bad dtype %d", dtype);                                    \
  ESR_REQUIRE(dtype == ESR_F32 || D % 4 != 0 || !(((uintptr_t)emb | (uintptr_t)emb_shadow) & 7),                      \
              who ": bf16 tables must be 8-byte aligned");                                                            \
  if (int rc = check_dim(who, D)) return rc;                                                                          \
  if (!workspace || workspace_bytes < esr_glove_step_workspace_bytes(B, D) || ((uintptr_t)workspace & 15)) {          \
    set_error(who ": workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,                            \
              esr_glove_step_workspace_bytes(B, D));                                                                  \
    return ESR_EWORKSPACE;                                                                                            \
  }

extern "C" {

int esr_glove_train_step(void* emb, void* emb_shadow, uint8_t* emb_loc, float* emb_accum, float* bias,
                         float* bias_accum, int64_t V, int dtype, int D, const int32_t* inputs, const float* target, int64_t B,
                         int mode, float lr, float eps, uint32_t stamp, const int32_t* presorted_ids,
                         const int32_t* presorted_perm, void* plan, int long_runs, int blocks_per_cu,
                         uint32_t* start_flag, uint32_t start_value, float* loss, void* workspace,
                         size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_glove_train_step");
  ESR_GLOVE_STEP_CHECKS("esr_glove_train_step")
  ESR_REQUIRE(inputs && target && loss, "esr_glove_train_step: null pointer");
  ESR_REQUIRE(stamp >= 1 && stamp <= kStampMax, "esr_glove_train_step: stamp %u not in [1, %u]", stamp, kStampMax);
  ESR_REQUIRE((presorted_ids == nullptr) == (presorted_perm == nullptr),
              "esr_glove_train_step: presorted_ids and presorted_perm must both be set or both be NULL");
  ESR_REQUIRE(!plan || presorted_ids, "esr_glove_train_step: a plan goes with the sorted ids it was made from");
  ESR_REQUIRE(!plan || !((uintptr_t)plan & 255), "esr_glove_train_step: misaligned plan");
  hipStream_t st = as_stream(stream);
  StepWs ws;
  step_ws_layout(B, D, (char*)workspace, &ws);
  const int32_t* sorted_ids = presorted_ids;
  const int32_t* perm = presorted_perm;
  if (!sorted_ids) {
    if (int rc = esr_segment_sort_ids(inputs, 2 * B, V, ws.sorted_ids, ws.perm, ws.sort_ws, ws.sort_ws_bytes, stream))
      return rc;
    sorted_ids = ws.sorted_ids;
    perm = ws.perm;
  }
  const GloveTables t{emb, emb_shadow, emb_loc, emb_accum, bias, bias_accum, D, dtype};
  launch_glove_step(t, inputs, target, B, mode, lr, eps, stamp, sorted_ids, perm, plan, long_runs, blocks_per_cu,
                    start_flag, start_value, loss, ws, st);
  return check_launch("esr_glove_train_step");
}

int esr_glove_train_steps(void* emb, void* emb_shadow, uint8_t* emb_loc, float* emb_accum, float* bias,
                          float* bias_accum, int64_t V, int dtype, int D, int nbatch, const int32_t* const* inputs,
                          const float* const* targets, int64_t B, int mode, float lr, float eps, uint32_t first_stamp,
                          const int32_t* sorted_ids, const int32_t* perm, void* plans, const int32_t* long_runs,
                          float* losses, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_glove_train_steps");
  ESR_GLOVE_STEP_CHECKS("esr_glove_train_steps")
  ESR_REQUIRE(nbatch >= 1 && nbatch <= kMaxGlovePlanBatch && inputs && targets && sorted_ids && perm && plans && losses &&
                  !((uintptr_t)plans & 255),
              "esr_glove_train_steps: nbatch=%d not in [1, %d], or a null / misaligned pointer", nbatch,
              kMaxGlovePlanBatch);
  ESR_REQUIRE(first_stamp >= 1 && first_stamp + (uint32_t)nbatch - 1 <= kStampMax,
              "esr_glove_train_steps: stamps %u .. %u leave [1, %u]", first_stamp, first_stamp + nbatch - 1, kStampMax);
  for (int i = 0; i < nbatch; ++i) ESR_REQUIRE(inputs[i] && targets[i], "esr_glove_train_steps: null list %d", i);
  hipStream_t st = as_stream(stream);
  StepWs ws;
  step_ws_layout(B, D, (char*)workspace, &ws);
  const size_t stride = esr_glove_plan_bytes(B);
  const GloveTables t{emb, emb_shadow, emb_loc, emb_accum, bias, bias_accum, D, dtype};
  for (int b = 0; b < nbatch; ++b)
    launch_glove_step(t, inputs[b], targets[b], B, mode, lr, eps, first_stamp + (uint32_t)b,
                      sorted_ids + (int64_t)b * 2 * B, perm + (int64_t)b * 2 * B, (char*)plans + (size_t)b * stride,
                      long_runs ? long_runs[b] : -1, 0, nullptr, 0u, losses + b, ws, st);
  return check_launch("esr_glove_train_steps");
}

int esr_rows_consolidate(void* primary, const void* shadow, uint8_t* loc, int64_t V, int dtype, int D,
                         esr_stream_t stream) {
  ESR_REQUIRE(V >= 0 && D > 0, "esr_rows_consolidate: bad sizes V=%lld D=%d", (long long)V, D);
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_rows_consolidate: bad dtype %d", dtype);
  if (V == 0) return ESR_OK;
  ESR_REQUIRE(primary && shadow && loc, "esr_rows_consolidate: null pointer");
  const bool vec = D % 4 == 0;  // four elements per lane: 16 bytes of an f32 row, 8 of a bf16 row
  const int nchunk = vec ? D / 4 : D;
  int G = 1;
  while (G < nchunk && G < kWave) G <<= 1;
  const int grid = grid_for_groups(V, G);
  hipStream_t st = as_stream(stream);
  if (dtype == ESR_BF16) {
    if (vec)
      hipLaunchKernelGGL(rows_consolidate_kernel<uint2>, dim3(grid), dim3(kBlock), 0, st, (uint2*)primary,
                         (const uint2*)shadow, loc, V, nchunk, G);
    else
      hipLaunchKernelGGL(rows_consolidate_kernel<uint16_t>, dim3(grid), dim3(kBlock), 0, st, (uint16_t*)primary,
                         (const uint16_t*)shadow, loc, V, nchunk, G);
  } else if (vec) {
    hipLaunchKernelGGL(rows_consolidate_kernel<float4>, dim3(grid), dim3(kBlock), 0, st, (float4*)primary,
                       (const float4*)shadow, loc, V, nchunk, G);
  } else {
    hipLaunchKernelGGL(rows_consolidate_kernel<float>, dim3(grid), dim3(kBlock), 0, st, (float*)primary,
                       (const float*)shadow, loc, V, nchunk, G);
  }
  return check_launch("esr_rows_consolidate");
}

}  // extern "C"
