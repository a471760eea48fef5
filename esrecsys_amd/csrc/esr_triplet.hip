// Fused Shop-The-Look score head + triplet hinge loss + norm-excess regulariser + gradients.
//
// Reference arithmetic: pinterest/models.py:67-72 (pos/neg score = sum_d scene*product) and
// pinterest/train_shop_the_look.py:99-104 (train) / :118 (eval):
//     triplet = sum_b relu(1 + neg_b - pos_b)
//     reg     = sum_b [relu(|s_b| - 1) + relu(|p_b| - 1) + relu(|n_b| - 1)]
//     loss    = (triplet + lam * reg) / batch_size
// Gradients (relu'(0) = 0):  m_b = [1 + neg_b - pos_b > 0]
//     g_s = (m (n - p) + lam [|s|>1] s/|s|) / B ; g_p = (-m s + lam [|p|>1] p/|p|) / B ;
//     g_n = ( m s + lam [|n|>1] n/|n|) / B
// One row group (G lanes) per triplet: the three rows are gathered once into VGPRs, five
// group-wide shuffle reductions give the scores and squared norms, and the three gradient rows
// are written from the same registers.  Loss partials are fp64, reduced in a fixed tree.
#include "esr_common.h"

namespace esr {

constexpr int kTripletBlocks = 2048;  // 256 CUs x 8 resident blocks

template <int VEC, int NCH, bool GRADS>
__global__ __launch_bounds__(kBlock) void triplet_kernel(
    const float* __restrict__ scene_table, const float* __restrict__ pos_table,
    const float* __restrict__ neg_table, int D, int G,
    const int32_t* __restrict__ scene_ids, const int32_t* __restrict__ pos_ids,
    const int32_t* __restrict__ neg_ids, int64_t B, float lam, float inv_bs, int with_reg,
    float* __restrict__ pos_score, float* __restrict__ neg_score, float* __restrict__ g_scene,
    float* __restrict__ g_pos, float* __restrict__ g_neg, double* __restrict__ part) {
  __shared__ double sm[8];
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  const int nvec = D / VEC;
  const bool at_ids = (with_reg & ESR_GRADS_AT_IDS) != 0;  // gradient rows go where their table rows came from
  with_reg &= 1;

  double acc = 0.0;
  for (int64_t b = group; b < B; b += ngroups) {
    const int64_t is = scene_ids ? (int64_t)scene_ids[b] : b;
    const int64_t ip = pos_ids ? (int64_t)pos_ids[b] : b;
    const int64_t in = neg_ids ? (int64_t)neg_ids[b] : b;
    RowRegs<VEC, NCH> s, p, n;
    row_load(s, scene_table + is * D, lig, G, nvec);
    row_load(p, pos_table + ip * D, lig, G, nvec);
    row_load(n, neg_table + in * D, lig, G, nvec);
    const float ps = group_sum(row_dot_partial(s, p), G);
    const float ns = group_sum(row_dot_partial(s, n), G);
    const float margin = 1.0f + ns - ps;
    const float m = margin > 0.f ? 1.f : 0.f;
    float loss_b = fmaxf(margin, 0.f);
    float cs = 0.f, cp = 0.f, cn = 0.f;  // lam * [|e|>1] / |e|
    if (with_reg) {
      const float s2 = group_sum(row_dot_partial(s, s), G);
      const float p2 = group_sum(row_dot_partial(p, p), G);
      const float n2 = group_sum(row_dot_partial(n, n), G);
      const float sn = sqrtf(s2), pn = sqrtf(p2), nn = sqrtf(n2);
      loss_b += lam * (fmaxf(sn - 1.f, 0.f) + fmaxf(pn - 1.f, 0.f) + fmaxf(nn - 1.f, 0.f));
      cs = sn > 1.f ? lam / sn : 0.f;
      cp = pn > 1.f ? lam / pn : 0.f;
      cn = nn > 1.f ? lam / nn : 0.f;
    }
    if (lig == 0) {
      acc += (double)loss_b;
      if (pos_score) pos_score[b] = ps;
      if (neg_score) neg_score[b] = ns;
    }
    if (GRADS) {
      RowRegs<VEC, NCH> gs, gp, gn;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float sv = s.v[k][e], pv = p.v[k][e], nv = n.v[k][e];
          gs.v[k][e] = trip_grad(m, __fsub_rn(nv, pv), cs, sv, inv_bs);
          gp.v[k][e] = trip_grad(-m, sv, cp, pv, inv_bs);
          gn.v[k][e] = trip_grad(m, sv, cn, nv, inv_bs);
        }
      row_store(gs, g_scene + (at_ids ? is : b) * D, lig, G, nvec);
      row_store(gp, g_pos + (at_ids ? ip : b) * D, lig, G, nvec);
      row_store(gn, g_neg + (at_ids ? in : b) * D, lig, G, nvec);
    }
  }
  const double t = block_sum_d(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

}  // namespace esr

using namespace esr;

extern "C" {

size_t esr_triplet_workspace_bytes(int64_t B) {
  (void)B;
  return sizeof(double) * kTripletBlocks;
}

int esr_triplet_fwd_bwd(const float* scene_table, int64_t Vs, const float* pos_table, int64_t Vp,
                        const float* neg_table, int64_t Vn, int D, const int32_t* scene_ids, const int32_t* pos_ids,
                        const int32_t* neg_ids, int64_t B, float regularization, float batch_size,
                        int with_reg, float* loss, float* pos_score, float* neg_score,
                        float* g_scene, float* g_pos, float* g_neg, void* workspace,
                        size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_triplet_fwd_bwd");
  ESR_REQUIRE(B > 0 && D > 0 && Vs > 0 && Vp > 0 && Vn > 0,
              "esr_triplet_fwd_bwd: bad sizes Vs=%lld Vp=%lld Vn=%lld D=%d B=%lld", (long long)Vs, (long long)Vp,
              (long long)Vn, D, (long long)B);
  ESR_REQUIRE(scene_table && pos_table && neg_table && loss, "esr_triplet_fwd_bwd: null pointer");
  const bool grads = g_scene || g_pos || g_neg;
  ESR_REQUIRE(!grads || (g_scene && g_pos && g_neg), "esr_triplet_fwd_bwd: set all of g_scene/g_pos/g_neg or none");
  ESR_REQUIRE(batch_size != 0.f, "esr_triplet_fwd_bwd: batch_size must be non-zero");
  const RowGeom g = row_geom(D);
  ESR_REQUIRE(g.nch <= kMaxChunksPerLane, "esr_triplet_fwd_bwd: D=%d not supported", D);
  if (!workspace || workspace_bytes < esr_triplet_workspace_bytes(B) || ((uintptr_t)workspace & 15)) {
    set_error("esr_triplet_fwd_bwd: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_triplet_workspace_bytes(B));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  double* part = (double*)workspace;
  const int nblk = (int)std::min<int64_t>(kTripletBlocks, cdiv(B, kBlock / g.G));
  const float inv_bs = 1.0f / batch_size;
  if (grads) {
    ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((triplet_kernel<VEC, NCH, true>), dim3(nblk), dim3(kBlock), 0, st,
                                           scene_table, pos_table, neg_table, D, g.G, scene_ids, pos_ids, neg_ids, B,
                                           regularization, inv_bs, with_reg, pos_score, neg_score, g_scene,
                                           g_pos, g_neg, part));
  } else {
    ESR_DISPATCH_ROW(g, hipLaunchKernelGGL((triplet_kernel<VEC, NCH, false>), dim3(nblk), dim3(kBlock), 0, st,
                                           scene_table, pos_table, neg_table, D, g.G, scene_ids, pos_ids, neg_ids, B,
                                           regularization, inv_bs, with_reg, pos_score, neg_score,
                                           (float*)nullptr, (float*)nullptr, (float*)nullptr, part));
  }
  // loss = total / batch_size (train_shop_the_look.py:104); eval_step (:118) passes batch_size = 1.
  finalize_scalar(part, nblk, 1.0 / (double)batch_size, loss, st);
  return check_launch("esr_triplet_fwd_bwd");
}

}  // extern "C"
