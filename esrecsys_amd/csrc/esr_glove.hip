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

}  // extern "C"
