// esr_spotify.hip -- the Spotify id-embedding two-tower: spotify/models.py:27-90 and the
// loss of spotify/train_spotify.py:77-107, forward + backward, plus the score-every-track eval of :113-119.
//
// One step is ONE playlist: n context tracks (5), m next tracks (a few to ~250), o sampled negatives (64); a
// track's embedding is concat(album_embed[album mod A], artist_embed[artist]) (2F = 64 floats).  The work is a
// few hundred rows, so the step is latency-bound: four small launches, no atomics, fixed summation orders.
//   gather     one wave per row: E[r] = concat(...), l2[r], hashed album row
//   affinity   one workgroup: raw = [next; neg] . ctx^T, row max (+0.1 isin boosts), the mean / extremal triplet
//              terms and d loss / d raw (the VJP of max and min splits evenly over ties -- two context tracks of one
//              album + artist have identical embeddings, so ties are real)
//   rowgrad    one wave per row b: the three self-affinity losses reduce to  mean_{a,b} f(E_a . E_b)  over a
//              group (the flip only permutes the matrix), so  d/dE_b = (2 / R^2) sum_a f'(E_a . E_b) E_a ; plus the
//              affinity terms through d raw and the norm term; writes the per-occurrence gradient rows
//   finalize   fixed-order fp64 sum of the loss partials
#include "esr_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace esr {

constexpr int kSpMaxCtx = 32;     // context rows (reference: 5)
constexpr int kSpMaxDim = 256;    // 2F
constexpr float kSpBoost = 0.1f;  // spotify/models.py:76-81

struct SpShape {
  int n, m, o, F;  // R = n + m + o rows, D2 = 2F
};

// (DPP / ds_swizzle / v_permlane32_swap butterflies: as __shfl_xor steps -- ds_bpermute with a computed address each -- the
// six steps per dot product were most of the row-gradient kernel once its rows came from LDS)
__device__ __forceinline__ float wave_sum_f(float v) { return group_sum(v, 64); }

// E[r][0:F] = album_table[album[r] mod A], E[r][F:2F] = artist_table[artist[r]]; l2[r]; hashed[r]
__global__ __launch_bounds__(kBlock) void spotify_gather_kernel(const float* __restrict__ album_table, int64_t A,
                                                               const float* __restrict__ artist_table, int F,
                                                               const int32_t* __restrict__ album,
                                                               const int32_t* __restrict__ artist, int R,
                                                               float* __restrict__ E, float* __restrict__ l2,
                                                               int32_t* __restrict__ hashed) {
  const int lane = threadIdx.x & 63;
  const int r = (int)((blockIdx.x * (int64_t)kBlock + threadIdx.x) >> 6);
  if (r >= R) return;
  const int64_t ha = (int64_t)album[r] % A, ar = artist[r];
  const int D2 = 2 * F;
  float ss = 0.f;
  for (int d = lane; d < D2; d += 64) {
    const float v = d < F ? album_table[ha * F + d] : artist_table[ar * F + (d - F)];
    E[(int64_t)r * D2 + d] = v;
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum_f(ss);
  if (lane == 0) {
    l2[r] = sqrtf(ss);
    if (hashed) hashed[r] = (int32_t)ha;
  }
}

// The same for tables under LAZY optax.sgd(lr, momentum) (train_spotify.py:238-241; decay_steps in esr_common.h): a row
// whose last[row] is behind step `now` - 1 is read as if the missed decay steps had been applied -- every occurrence
// computes the caught-up parameter row for itself (identical values: a function of the stored row, its trace and the
// gap), nothing is written.  The write-back happens where the row is written anyway, in the momentum step at the end of
// the train step (kMomentumStepLazy, esr_optim.hip).  Replaces the catch-up launch in front of the step: claim by
// atomicExch -> load -> decay -> store, three dependent memory round trips (12 us) ahead of everything else.
struct SpLazy {
  const float *album_trace, *artist_trace;
  const int32_t *album_last, *artist_last;
  int now;
  float lr, momentum;
};
__global__ __launch_bounds__(kBlock) void spotify_gather_lazy_kernel(const float* __restrict__ album_table, int64_t A,
                                                                    const float* __restrict__ artist_table, int F,
                                                                    const int32_t* __restrict__ album,
                                                                    const int32_t* __restrict__ artist, int R, SpLazy lz,
                                                                    float* __restrict__ E, float* __restrict__ l2,
                                                                    int32_t* __restrict__ hashed) {
  const int lane = threadIdx.x & 63;
  const int r = (int)((blockIdx.x * (int64_t)kBlock + threadIdx.x) >> 6);
  if (r >= R) return;
  const int64_t ha = (int64_t)album[r] % A, ar = artist[r];
  const int gap_a = lz.now - 1 - lz.album_last[ha], gap_r = lz.now - 1 - lz.artist_last[ar];
  const int D2 = 2 * F;
  const DecayCoef dk_a = decay_coef(gap_a, lz.momentum), dk_r = decay_coef(gap_r, lz.momentum);
  float ss = 0.f;
  for (int d = lane; d < D2; d += 64) {
    const bool alb = d < F;
    const int64_t at = alb ? ha * F + d : ar * F + (d - F);
    float v = alb ? album_table[at] : artist_table[at];
    const int gap = alb ? gap_a : gap_r;
    if (gap > 0) {
      float tv = alb ? lz.album_trace[at] : lz.artist_trace[at];
      decay_apply(v, tv, alb ? dk_a : dk_r, lz.lr, lz.momentum);
    }
    E[(int64_t)r * D2 + d] = v;
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum_f(ss);
  if (lane == 0) {
    l2[r] = sqrtf(ss);
    if (hashed) hashed[r] = (int32_t)ha;
  }
}

// raw[i][c] = E[n + i] . E[c] for the m + o scored rows; aff = rowmax + boosts; W[i][c] = d loss / d raw[i][c]
// (only when W != null); head[0] = relu(mean triplet) + relu(extremal triplet).
__global__ __launch_bounds__(kBlock) void spotify_affinity_kernel(const float* __restrict__ E, SpShape sh,
                                                                 const int32_t* __restrict__ album,
                                                                 const int32_t* __restrict__ artist,
                                                                 float* __restrict__ raw, float* __restrict__ aff,
                                                                 float* __restrict__ W, double* __restrict__ head) {
  __shared__ double sm[kBlock / 64 + 1];
  __shared__ double s_sum_pos, s_sum_neg;
  __shared__ float s_min_pos, s_max_neg;
  __shared__ int s_cnt_min, s_cnt_max;
  const int n = sh.n, m = sh.m, o = sh.o, D2 = 2 * sh.F, S = m + o, t = threadIdx.x;
  // The context rows (rows 0 .. n-1 of E, <= 32 x 256 floats) are every thread's second operand: staged in LDS once and
  // read as broadcasts; a thread's own row arrives as float4 loads, 16 instead of 64 per 64-float row.  Every dot
  // product still adds its terms in the order d = 0, 1, 2, ...: the values are those of the plain loop, bit for bit
  // (which walked both rows one float at a time out of global memory: 40 us for a 5 + 45 + 64 row playlist, a third of
  // the Spotify step).
  __shared__ __attribute__((aligned(16))) float ctx[kSpMaxCtx * kSpMaxDim];
  for (int k = t; k < n * D2; k += kBlock) ctx[k] = E[k];
  __syncthreads();
  double sum_pos = 0.0, sum_neg = 0.0;
  __shared__ float w_mn[kBlock / 64], w_mx[kBlock / 64];
  __shared__ int w_cmn[kBlock / 64], w_cmx[kBlock / 64];
  float mn = INFINITY, mx = -INFINITY;
  int cmn = 0, cmx = 0;
  for (int i = t; i < S; i += kBlock) {
    const float* z = E + (int64_t)(n + i) * D2;
    float best = -INFINITY;
    if ((D2 & 3) == 0) {
      float sacc[kSpMaxCtx];
#pragma unroll
      for (int c = 0; c < kSpMaxCtx; ++c) sacc[c] = 0.f;
      for (int d0 = 0; d0 < D2; d0 += 4) {
        const float4 zv = *reinterpret_cast<const float4*>(z + d0);
#pragma unroll
        for (int c = 0; c < kSpMaxCtx; ++c)
          if (c < n) {
            const float4 xv = *reinterpret_cast<const float4*>(ctx + c * D2 + d0);
            sacc[c] = fmaf(zv.w, xv.w, fmaf(zv.z, xv.z, fmaf(zv.y, xv.y, fmaf(zv.x, xv.x, sacc[c]))));
          }
      }
#pragma unroll
      for (int c = 0; c < kSpMaxCtx; ++c)
        if (c < n) {
          raw[i * n + c] = sacc[c];
          best = fmaxf(best, sacc[c]);
        }
    } else {
      for (int c = 0; c < n; ++c) {
        const float* x = ctx + c * D2;
        float s = 0.f;
        for (int d = 0; d < D2; ++d) s = fmaf(z[d], x[d], s);
        raw[i * n + c] = s;
        best = fmaxf(best, s);
      }
    }
    bool in_album = false, in_artist = false;
    for (int c = 0; c < n; ++c) {
      in_album |= album[n + i] == album[c];
      in_artist |= artist[n + i] == artist[c];
    }
    const float a = best + (in_album ? kSpBoost : 0.f) + (in_artist ? kSpBoost : 0.f);
    aff[i] = a;
    if (i < m) {
      sum_pos += a;
      if (a < mn) { mn = a; cmn = 1; } else if (a == mn) ++cmn;
    } else {
      sum_neg += a;
      if (a > mx) { mx = a; cmx = 1; } else if (a == mx) ++cmx;
    }
  }
  sum_pos = block_sum_d(sum_pos, sm);  // valid in thread 0
  sum_neg = block_sum_d(sum_neg, sm);
  // smallest positive / largest negative affinity and how many rows tie for it (the VJP of min / max splits evenly over
  // ties): an exact, order-free reduction of (value, count) pairs -- thread 0 used to walk all m + o values alone
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) {
    const float omn = __shfl_xor(mn, sft, 64), omx = __shfl_xor(mx, sft, 64);
    const int ocmn = __shfl_xor(cmn, sft, 64), ocmx = __shfl_xor(cmx, sft, 64);
    if (omn < mn) { mn = omn; cmn = ocmn; } else if (omn == mn) cmn += ocmn;
    if (omx > mx) { mx = omx; cmx = ocmx; } else if (omx == mx) cmx += ocmx;
  }
  if ((t & 63) == 0) {
    w_mn[t >> 6] = mn; w_mx[t >> 6] = mx; w_cmn[t >> 6] = cmn; w_cmx[t >> 6] = cmx;
  }
  __syncthreads();  // aff[] and the wave results visible
  if (t == 0) {
    s_sum_pos = sum_pos;
    s_sum_neg = sum_neg;
    for (int w = 1; w < kBlock / 64; ++w) {
      if (w_mn[w] < mn) { mn = w_mn[w]; cmn = w_cmn[w]; } else if (w_mn[w] == mn) cmn += w_cmn[w];
      if (w_mx[w] > mx) { mx = w_mx[w]; cmx = w_cmx[w]; } else if (w_mx[w] == mx) cmx += w_cmx[w];
    }
    s_min_pos = mn; s_max_neg = mx; s_cnt_min = cmn; s_cnt_max = cmx;
  }
  __syncthreads();
  const float mt_arg = 1.0f + (float)(s_sum_neg / o) - (float)(s_sum_pos / m);
  const float et_arg = 1.0f + s_max_neg - s_min_pos;
  if (t == 0) head[0] = (double)fmaxf(mt_arg, 0.f) + (double)fmaxf(et_arg, 0.f);
  if (!W) return;
  for (int i = t; i < S; i += kBlock) {
    const float a = aff[i];
    float d = 0.f;
    if (i < m) {
      if (mt_arg > 0.f) d -= 1.0f / m;
      if (et_arg > 0.f && a == s_min_pos) d -= 1.0f / s_cnt_min;
    } else {
      if (mt_arg > 0.f) d += 1.0f / o;
      if (et_arg > 0.f && a == s_max_neg) d += 1.0f / s_cnt_max;
    }
    float best = -INFINITY;
    for (int c = 0; c < n; ++c) best = fmaxf(best, raw[i * n + c]);
    int ties = 0;
    for (int c = 0; c < n; ++c) ties += raw[i * n + c] == best;
    for (int c = 0; c < n; ++c) W[i * n + c] = raw[i * n + c] == best ? d / ties : 0.f;
  }
}

// The same, for playlists whose rows fit LDS (round 4): the kernel above gives every scored row one thread, which walks its
// row out of global memory 16 bytes at a time -- two waves, sixteen exposed load latencies: 16 us for 104 rows.  Here all
// 1024 threads stage E into LDS in one sweep (rows padded to 2F + 1 floats: a thread per row and a lane per dimension
// both read without bank conflicts), every (scored row, context row) pair is a thread's dot product (terms added d = 0,
// 1, 2, ...: the same values), and the per-row maxima / boosts / tie counts follow from LDS.
constexpr int kSpAffThreads = 1024;
__host__ __device__ inline size_t sp_aff_lds_bytes(int n, int m, int o, int F) {
  return ((size_t)(n + m + o) * (2 * F + 1) + (size_t)(m + o) * n) * 4 + 1024;
}
__global__ __launch_bounds__(kSpAffThreads) void spotify_affinity_lds_kernel(const float* __restrict__ Eg, SpShape sh,
                                                                            const int32_t* __restrict__ album,
                                                                            const int32_t* __restrict__ artist,
                                                                            float* __restrict__ raw_out, float* __restrict__ aff,
                                                                            float* __restrict__ W, double* __restrict__ head) {
  extern __shared__ __attribute__((aligned(16))) char sp_aff_lds[];
  const int n = sh.n, m = sh.m, o = sh.o, D2 = 2 * sh.F, S = m + o, R = n + S, EP = D2 + 1, t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  constexpr int T = kSpAffThreads, NW = T / 64;
  double* red = reinterpret_cast<double*>(sp_aff_lds);                 // [0, 256): 2 x 16 doubles
  float* redf = reinterpret_cast<float*>(sp_aff_lds + 256);            // 36 floats
  int* redi = reinterpret_cast<int*>(sp_aff_lds + 512);                // 34 ints
  float* E = reinterpret_cast<float*>(sp_aff_lds + 1024);
  float* raw = E + (size_t)R * EP;
  for (int e = t; e < R * D2; e += T) {
    const int r = e / D2;
    E[r * EP + (e - r * D2)] = Eg[e];
  }
  __syncthreads();
  for (int e = t; e < S * n; e += T) {
    const int i = e / n, c = e - i * n;
    const float* z = E + (n + i) * EP;
    const float* x = E + c * EP;
    float sc = 0.f;
    int d = 0;
    for (; d + 8 <= D2; d += 8) {
      float zv[8], xv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { zv[q] = z[d + q]; xv[q] = x[d + q]; }
#pragma unroll
      for (int q = 0; q < 8; ++q) sc = fmaf(zv[q], xv[q], sc);
    }
    for (; d < D2; ++d) sc = fmaf(z[d], x[d], sc);
    raw[e] = sc;
    if (raw_out) raw_out[e] = sc;
  }
  __syncthreads();
  double sum_pos = 0.0, sum_neg = 0.0;
  float mn = INFINITY, mx = -INFINITY;
  int cmn = 0, cmx = 0;
  for (int i = t; i < S; i += T) {
    float best = -INFINITY;
    bool in_album = false, in_artist = false;
    const int32_t al = album[n + i], ar = artist[n + i];
    for (int c = 0; c < n; ++c) {
      best = fmaxf(best, raw[i * n + c]);
      in_album |= al == album[c];
      in_artist |= ar == artist[c];
    }
    const float av = best + (in_album ? kSpBoost : 0.f) + (in_artist ? kSpBoost : 0.f);
    aff[i] = av;
    if (i < m) {
      sum_pos += av;
      if (av < mn) { mn = av; cmn = 1; } else if (av == mn) ++cmn;
    } else {
      sum_neg += av;
      if (av > mx) { mx = av; cmx = 1; } else if (av == mx) ++cmx;
    }
  }
  sum_pos = wave_sum_d(sum_pos);
  sum_neg = wave_sum_d(sum_neg);
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) {
    const float omn = __shfl_xor(mn, sft, 64), omx = __shfl_xor(mx, sft, 64);
    const int ocmn = __shfl_xor(cmn, sft, 64), ocmx = __shfl_xor(cmx, sft, 64);
    if (omn < mn) { mn = omn; cmn = ocmn; } else if (omn == mn) cmn += ocmn;
    if (omx > mx) { mx = omx; cmx = ocmx; } else if (omx == mx) cmx += ocmx;
  }
  if (lane == 0) {
    red[wv] = sum_pos; red[NW + wv] = sum_neg;
    redf[wv] = mn; redf[NW + wv] = mx;
    redi[wv] = cmn; redi[NW + wv] = cmx;
  }
  __syncthreads();
  if (t == 0) {
    double sp = 0.0, sn = 0.0;
    for (int w = 0; w < NW; ++w) { sp += red[w]; sn += red[NW + w]; }
    mn = redf[0]; cmn = redi[0]; mx = redf[NW]; cmx = redi[NW];
    for (int w = 1; w < NW; ++w) {
      if (redf[w] < mn) { mn = redf[w]; cmn = redi[w]; } else if (redf[w] == mn) cmn += redi[w];
      if (redf[NW + w] > mx) { mx = redf[NW + w]; cmx = redi[NW + w]; } else if (redf[NW + w] == mx) cmx += redi[NW + w];
    }
    const float mt = 1.0f + (float)(sn / o) - (float)(sp / m);
    const float et = 1.0f + mx - mn;
    head[0] = (double)fmaxf(mt, 0.f) + (double)fmaxf(et, 0.f);
    redf[2 * NW] = mt; redf[2 * NW + 1] = et; redf[2 * NW + 2] = mn; redf[2 * NW + 3] = mx;
    redi[2 * NW] = cmn; redi[2 * NW + 1] = cmx;
  }
  if (!W) return;
  __syncthreads();
  const float mt_arg = redf[2 * NW], et_arg = redf[2 * NW + 1], s_min_pos = redf[2 * NW + 2], s_max_neg = redf[2 * NW + 3];
  const int s_cnt_min = redi[2 * NW], s_cnt_max = redi[2 * NW + 1];
  for (int i = t; i < S; i += T) {
    const float av = aff[i];
    float d = 0.f;
    if (i < m) {
      if (mt_arg > 0.f) d -= 1.0f / m;
      if (et_arg > 0.f && av == s_min_pos) d -= 1.0f / s_cnt_min;
    } else {
      if (mt_arg > 0.f) d += 1.0f / o;
      if (et_arg > 0.f && av == s_max_neg) d += 1.0f / s_cnt_max;
    }
    float best = -INFINITY;
    for (int c = 0; c < n; ++c) best = fmaxf(best, raw[i * n + c]);
    int ties = 0;
    for (int c = 0; c < n; ++c) ties += raw[i * n + c] == best;
    for (int c = 0; c < n; ++c) W[i * n + c] = raw[i * n + c] == best ? d / ties : 0.f;
  }
}

// one wave per row b; lane holds dims lane, lane + 64, ... (NCH = ceil(2F / 64) of them)
// STAGED (round 4): every workgroup first copies ALL rows of the playlist into its LDS (one sweep, one round trip: 29 KB
// from L2) and takes the partner rows from there.  Read from global memory four rows at a time, a negative's 64 partners
// were sixteen dependent round trips: 16-24 us for a launch whose arithmetic is 2 us.
template <int NCH, bool STAGED>
__global__ __launch_bounds__(kBlock) void spotify_rowgrad_kernel(const float* __restrict__ Eg, SpShape sh,
                                                                const float* __restrict__ l2,
                                                                const float* __restrict__ W, float regularization,
                                                                float* __restrict__ g_album,
                                                                float* __restrict__ g_artist,
                                                                double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char sp_rg_lds[];
  const int n = sh.n, m = sh.m, o = sh.o, F = sh.F, D2 = 2 * F, R = n + m + o;
  const int lane = threadIdx.x & 63;
  const float* E = Eg;
  if (STAGED) {
    float* El = reinterpret_cast<float*>(sp_rg_lds);
    if ((D2 & 3) == 0) {
      for (int e = threadIdx.x; e < R * D2 / 4; e += kBlock)
        reinterpret_cast<float4*>(El)[e] = reinterpret_cast<const float4*>(Eg)[e];
    } else {
      for (int e = threadIdx.x; e < R * D2; e += kBlock) El[e] = Eg[e];
    }
    __syncthreads();
    E = El;
  }
  const int b = (int)((blockIdx.x * (int64_t)kBlock + threadIdx.x) >> 6);
  if (b >= R) return;
  auto load = [&](int r, float (&v)[NCH]) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int d = lane + 64 * k;
      v[k] = d < D2 ? E[r * D2 + d] : 0.f;
    }
  };
  auto dot = [&](const float (&x)[NCH], const float (&y)[NCH]) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) s = fmaf(x[k], y[k], s);
    return wave_sum_f(s);
  };
  float z[NCH], acc[NCH];
  load(b, z);
#pragma unroll
  for (int k = 0; k < NCH; ++k) acc[k] = 0.f;
  // ---- self-affinity of b's group: pull (context, next): f = relu(0.5 - s); push (neg): f = relu(s)
  const int g0 = b < n ? 0 : b < n + m ? n : n + m;
  const int Rg = b < n ? n : b < n + m ? m : o;
  const bool push = b >= n + m;
  double lsum = 0.0;
  for (int a0 = 0; a0 < Rg; a0 += 4) {  // four rows per round: their wave reductions overlap
    float za[4][NCH], s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load(g0 + min(a0 + u, Rg - 1), za[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = dot(za[u], z);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (a0 + u >= Rg) continue;
      const float f = push ? fmaxf(s[u], 0.f) : fmaxf(0.5f - s[u], 0.f);
      const float fp = push ? (s[u] > 0.f ? 1.f : 0.f) : (s[u] < 0.5f ? -1.f : 0.f);
      lsum += (double)f;
#pragma unroll
      for (int k = 0; k < NCH; ++k) acc[k] = fmaf(fp, za[u][k], acc[k]);
    }
  }
  const float inv = 1.0f / ((float)Rg * (float)Rg);
#pragma unroll
  for (int k = 0; k < NCH; ++k) acc[k] *= 2.0f * inv;
  double part = lsum / ((double)Rg * (double)Rg);
  // ---- affinity terms through W = d loss / d raw
  if (b < n) {
    // a context row collects from the scored rows whose maximum it is (about (m + o) / n of them): 64 weights per read, only
    // the nonzero ones visited, in ascending order.  (Probing all m + o one by one was a chain of 104 dependent loads:
    // the n context waves finished 10 us after everybody else -- two thirds of this launch.)
    const int S = m + o;
    for (int i0 = 0; i0 < S; i0 += 64) {
      const float wl = i0 + lane < S ? W[(i0 + lane) * n + b] : 0.f;
      unsigned long long todo = __ballot(wl != 0.f);
      while (todo) {  // eight rows per trip: their loads leave together (one at a time each was an exposed round trip)
        float wq[8], x[8][NCH];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int u = todo ? __builtin_ctzll(todo) : 0;
          wq[q] = todo ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), u)) : 0.f;
          todo &= todo - (todo ? 1 : 0);
          load(n + i0 + u, x[q]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int k = 0; k < NCH; ++k) acc[k] = fmaf(wq[q], x[q][k], acc[k]);  // (w = 0 behind the last: acc unchanged)
      }
    }
  } else {
    for (int c = 0; c < n; ++c) {
      const float w = W[(b - n) * n + c];
      if (w == 0.f) continue;
      float x[NCH];
      load(c, x);
#pragma unroll
      for (int k = 0; k < NCH; ++k) acc[k] = fmaf(w, x[k], acc[k]);
    }
  }
  // ---- norm term: relu(|E_b| - regularization)
  const float nb = l2[b];
  if (nb > regularization) {
    part += (double)(nb - regularization);
#pragma unroll
    for (int k = 0; k < NCH; ++k) acc[k] += z[k] / nb;
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int d = lane + 64 * k;
    if (d < F) g_album[(int64_t)b * F + d] = acc[k];
    else if (d < D2) g_artist[(int64_t)b * F + (d - F)] = acc[k];
  }
  if (lane == 0) partial[b] = part;
}

// out[a][b] = E[g0 + Rg - 1 - a] . E[g0 + b]   (jnp.dot(jnp.flip(x, -2), x.T))
__global__ __launch_bounds__(kBlock) void spotify_self_affinity_kernel(const float* __restrict__ E, int D2, int g0,
                                                                      int Rg, float* __restrict__ out) {
  const int64_t total = (int64_t)Rg * Rg;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int a = (int)(i / Rg), b = (int)(i - (int64_t)a * Rg);
    const float* x = E + (int64_t)(g0 + Rg - 1 - a) * D2;
    const float* y = E + (int64_t)(g0 + b) * D2;
    float s = 0.f;
    for (int d = 0; d < D2; ++d) s = fmaf(x[d], y[d], s);
    out[i] = s;
  }
}

// eval: aff[t] = max_c concat(album_table[album[t] mod A], artist_table[artist[t]]) . ctx[c] + boosts, every track
// 16 lanes per track (float4 chunks); the n context rows are staged in LDS
__global__ __launch_bounds__(kBlock) void spotify_affinity_all_kernel(const float* __restrict__ album_table, int64_t A,
                                                                     const float* __restrict__ artist_table, int F,
                                                                     const int32_t* __restrict__ ctx_album,
                                                                     const int32_t* __restrict__ ctx_artist, int n,
                                                                     const int32_t* __restrict__ all_albums,
                                                                     const int32_t* __restrict__ all_artists,
                                                                     int64_t T, float* __restrict__ aff) {
  __shared__ __attribute__((aligned(16))) float ctx[kSpMaxCtx * kSpMaxDim];
  __shared__ int32_t c_album[kSpMaxCtx], c_artist[kSpMaxCtx];
  const int D2 = 2 * F;
  for (int i = threadIdx.x; i < n * D2; i += kBlock) {
    const int c = i / D2, d = i - c * D2;
    ctx[i] = d < F ? album_table[((int64_t)ctx_album[c] % A) * F + d] : artist_table[(int64_t)ctx_artist[c] * F + d - F];
  }
  if (threadIdx.x < n) {
    c_album[threadIdx.x] = ctx_album[threadIdx.x];
    c_artist[threadIdx.x] = ctx_artist[threadIdx.x];
  }
  __syncthreads();
  const int lig = threadIdx.x & 15;
  const int64_t group = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
  const int64_t ngroups = ((int64_t)gridDim.x * kBlock) >> 4;
  const bool vec = (F & 3) == 0;                       // float4 chunks never straddle the two tables
  const uint32_t A32 = A < ((int64_t)1 << 31) ? (uint32_t)A : 0u;  // album ids are non-negative int32: a 32-bit modulo
  const int nchk = D2 >> 2;
  if (vec) {
    // two tracks per group and trip: the ids and rows of both are requested before either is scored (the trip is a
    // chain ids -> rows -> dots; one track at a time it ran at 1.6 TB/s of gathered rows)
    for (int64_t t0 = group; t0 < T; t0 += 2 * ngroups) {
      int32_t al[2], ar[2];
      float4 row[2][kSpMaxDim / 64];
      bool on[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t t = t0 + j * ngroups;
        on[j] = t < T;
        al[j] = on[j] ? all_albums[t] : 0;
        ar[j] = on[j] ? all_artists[t] : 0;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t ha = A32 ? (int64_t)((uint32_t)al[j] % A32) : (int64_t)al[j] % A;
        const float* pa = album_table + ha * F;
        const float* pr = artist_table + (int64_t)ar[j] * F;
#pragma unroll
        for (int k = 0; k < kSpMaxDim / 64; ++k) {
          const int d = 4 * (lig + 16 * k);
          row[j][k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (on[j] && lig + 16 * k < nchk)
            row[j][k] = d < F ? *reinterpret_cast<const float4*>(pa + d) : *reinterpret_cast<const float4*>(pr + d - F);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float best = -INFINITY;
        for (int c = 0; c < n; ++c) {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < kSpMaxDim / 64; ++k)
            if (lig + 16 * k < nchk) {
              const float4 xv = *reinterpret_cast<const float4*>(ctx + c * D2 + 4 * (lig + 16 * k));
              s = fmaf(row[j][k].w, xv.w, fmaf(row[j][k].z, xv.z, fmaf(row[j][k].y, xv.y, fmaf(row[j][k].x, xv.x, s))));
            }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
          best = fmaxf(best, s);
        }
        if (lig == 0 && on[j]) {
          bool in_album = false, in_artist = false;
          for (int c = 0; c < n; ++c) {
            in_album |= al[j] == c_album[c];
            in_artist |= ar[j] == c_artist[c];
          }
          aff[t0 + j * ngroups] = best + (in_album ? kSpBoost : 0.f) + (in_artist ? kSpBoost : 0.f);
        }
      }
    }
    return;
  }
  for (int64_t t = group; t < T; t += ngroups) {  // F not a multiple of 4: one float at a time
    const int32_t al = all_albums[t], ar = all_artists[t];
    const int64_t ha = A32 ? (int64_t)((uint32_t)al % A32) : (int64_t)al % A;
    const float* pa = album_table + ha * F;
    const float* pr = artist_table + (int64_t)ar * F;
    float best = -INFINITY;
    for (int c = 0; c < n; ++c) {
      float s = 0.f;
      for (int d = lig; d < D2; d += 16) s = fmaf(d < F ? pa[d] : pr[d - F], ctx[c * D2 + d], s);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
      best = fmaxf(best, s);
    }
    if (lig == 0) {
      bool in_album = false, in_artist = false;
      for (int c = 0; c < n; ++c) {
        in_album |= al == c_album[c];
        in_artist |= ar == c_artist[c];
      }
      aff[t] = best + (in_album ? kSpBoost : 0.f) + (in_artist ? kSpBoost : 0.f);
    }
  }
}

struct SpWs {
  float *E, *l2, *raw, *aff, *W;
  double* partial;  // [R] row partials then [1] head
  size_t total;
};
static SpWs sp_layout(char* base, int n, int m, int o, int F) {
  const size_t R = (size_t)n + m + o, S = (size_t)m + o;
  SpWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  w.E = (float*)take(R * 2 * F * 4);
  w.l2 = (float*)take(R * 4);
  w.raw = (float*)take(S * n * 4);
  w.aff = (float*)take(S * 4);
  w.W = (float*)take(S * n * 4);
  w.partial = (double*)take((R + 1) * 8);
  w.total = off;
  return w;
}

// spotify_affinity_lds_kernel when the playlist's rows fit LDS, else spotify_affinity_kernel
static bool sp_affinity_in_lds(SpShape sh) {
  const char* e = getenv("ESR_SPOTIFY_AFFINITY");  // "global": the first kernel (A/B)
  return sp_aff_lds_bytes(sh.n, sh.m, sh.o, sh.F) <= 150 * 1024 && !(e && strcmp(e, "global") == 0);
}
static int sp_launch_affinity(const SpWs& w, SpShape sh, const int32_t* album_ids, const int32_t* artist_ids, float* W,
                              double* head, hipStream_t st) {
  constexpr size_t kMaxLds = 150 * 1024;
  const size_t lds = sp_aff_lds_bytes(sh.n, sh.m, sh.o, sh.F);
  if (sp_affinity_in_lds(sh)) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)spotify_affinity_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kMaxLds) != hipSuccess) {
        set_error("spotify affinity: cannot raise the LDS limit");
        return ESR_ELAUNCH;
      }
      attr_set = true;
    }
    hipLaunchKernelGGL(spotify_affinity_lds_kernel, dim3(1), dim3(kSpAffThreads), lds, st, (const float*)w.E, sh, album_ids,
                       artist_ids, w.raw, w.aff, W, head);
  } else {
    hipLaunchKernelGGL(spotify_affinity_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)w.E, sh, album_ids, artist_ids,
                       w.raw, w.aff, W, head);
  }
  return ESR_OK;
}

static int sp_check(const char* who, int n, int m, int o, int F, int64_t A, int64_t n_artists) {
  if (!(n > 0 && m > 0 && o > 0 && F > 0 && A > 0 && n_artists > 0 && n <= kSpMaxCtx && 2 * F <= kSpMaxDim)) {
    set_error("%s: bad sizes n=%d m=%d o=%d F=%d (n <= %d, 2F <= %d)", who, n, m, o, F, kSpMaxCtx, kSpMaxDim);
    return ESR_EINVAL;
  }
  return ESR_OK;
}

// in esr_optim.hip
int sparse_momentum_step_lazy2(float* const* tables, float* const* traces, int32_t* const* lasts, const int64_t* row_offsets,
                               int ntables, int D, const int32_t* sorted_vids, const int32_t* perm, int64_t n, float* grad_rows,
                               float lr, float momentum, int now, hipStream_t st);

}  // namespace esr

using namespace esr;

extern "C" {

size_t esr_spotify_workspace_bytes(int n, int m, int o, int F) {
  if (n <= 0 || m <= 0 || o <= 0 || F <= 0) return 256;
  return sp_layout(nullptr, n, m, o, F).total;
}

int esr_spotify_get_embeddings(const float* album_table, int64_t n_album_rows, const float* artist_table,
                               int64_t n_artists, int F, const int32_t* album_ids, const int32_t* artist_ids,
                               int64_t count, float* out, float* l2, esr_stream_t stream) {
  ESR_REQUIRE(n_album_rows > 0 && n_artists > 0 && F > 0 && count >= 0 && count < ((int64_t)1 << 24),
              "esr_spotify_get_embeddings: bad sizes");
  if (count == 0) return ESR_OK;
  ESR_REQUIRE(album_table && artist_table && album_ids && artist_ids && out && l2,
              "esr_spotify_get_embeddings: null pointer");
  hipLaunchKernelGGL(spotify_gather_kernel, dim3((int)cdiv(count, kBlock / 64)), dim3(kBlock), 0, as_stream(stream),
                     album_table, n_album_rows, artist_table, F, album_ids, artist_ids, (int)count, out, l2,
                     (int32_t*)nullptr);
  return check_launch("esr_spotify_get_embeddings");
}

int esr_spotify_forward(const float* album_table, int64_t n_album_rows, const float* artist_table, int64_t n_artists,
                        int F, const int32_t* album_ids, const int32_t* artist_ids, int n, int m, int o, float* pos,
                        float* neg, float* ctx_self, float* next_self, float* neg_self, float* l2, void* workspace,
                        size_t workspace_bytes, esr_stream_t stream) {
  if (int rc = sp_check("esr_spotify_forward", n, m, o, F, n_album_rows, n_artists)) return rc;
  ESR_REQUIRE(album_table && artist_table && album_ids && artist_ids && pos && neg && ctx_self && next_self &&
                  neg_self && l2 && workspace, "esr_spotify_forward: null pointer");
  const SpWs w = sp_layout((char*)workspace, n, m, o, F);
  if (workspace_bytes < w.total || ((uintptr_t)workspace & 15)) {
    set_error("esr_spotify_forward: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes, w.total);
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int R = n + m + o, D2 = 2 * F;
  const SpShape sh{n, m, o, F};
  hipLaunchKernelGGL(spotify_gather_kernel, dim3((int)cdiv(R, kBlock / 64)), dim3(kBlock), 0, st, album_table,
                     n_album_rows, artist_table, F, album_ids, artist_ids, R, w.E, l2, (int32_t*)nullptr);
  if (int rc = sp_launch_affinity(w, sh, album_ids, artist_ids, nullptr, w.partial + R, st)) return rc;
  (void)hipMemcpyAsync(pos, w.aff, sizeof(float) * m, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(neg, w.aff + m, sizeof(float) * o, hipMemcpyDeviceToDevice, st);
  const int g0s[3] = {0, n, n + m}, rgs[3] = {n, m, o};
  float* outs[3] = {ctx_self, next_self, neg_self};
  for (int g = 0; g < 3; ++g) {
    const int grid = (int)std::min<int64_t>(cdiv((int64_t)rgs[g] * rgs[g], kBlock), 4096);
    hipLaunchKernelGGL(spotify_self_affinity_kernel, dim3(grid), dim3(kBlock), 0, st, (const float*)w.E, D2, g0s[g],
                       rgs[g], outs[g]);
  }
  return check_launch("esr_spotify_forward");
}

static int sp_fwd_bwd(const float* album_table, int64_t n_album_rows, const float* artist_table, int64_t n_artists, int F,
                      const int32_t* album_ids, const int32_t* artist_ids, int n, int m, int o, float regularization,
                      float* loss, int32_t* album_rows, float* g_album_rows, float* g_artist_rows, void* workspace,
                      size_t workspace_bytes, esr_stream_t stream, const SpLazy* lazy);

int esr_spotify_fwd_bwd(const float* album_table, int64_t n_album_rows, const float* artist_table, int64_t n_artists,
                        int F, const int32_t* album_ids, const int32_t* artist_ids, int n, int m, int o,
                        float regularization, float* loss, int32_t* album_rows, float* g_album_rows,
                        float* g_artist_rows, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  return sp_fwd_bwd(album_table, n_album_rows, artist_table, n_artists, F, album_ids, artist_ids, n, m, o, regularization,
                    loss, album_rows, g_album_rows, g_artist_rows, workspace, workspace_bytes, stream, nullptr);
}

// lazy != null: the tables are under lazy momentum -- rows are read through spotify_gather_lazy_kernel
static int sp_fwd_bwd(const float* album_table, int64_t n_album_rows, const float* artist_table, int64_t n_artists, int F,
                      const int32_t* album_ids, const int32_t* artist_ids, int n, int m, int o, float regularization,
                      float* loss, int32_t* album_rows, float* g_album_rows, float* g_artist_rows, void* workspace,
                      size_t workspace_bytes, esr_stream_t stream, const SpLazy* lazy) {
  if (int rc = sp_check("esr_spotify_fwd_bwd", n, m, o, F, n_album_rows, n_artists)) return rc;
  ESR_REQUIRE(album_table && artist_table && album_ids && artist_ids && loss && album_rows && g_album_rows &&
                  g_artist_rows && workspace, "esr_spotify_fwd_bwd: null pointer");
  const SpWs w = sp_layout((char*)workspace, n, m, o, F);
  if (workspace_bytes < w.total || ((uintptr_t)workspace & 15)) {
    set_error("esr_spotify_fwd_bwd: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes, w.total);
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int R = n + m + o, D2 = 2 * F;
  const SpShape sh{n, m, o, F};
  const int wgrid = (int)cdiv(R, kBlock / 64);
  if (lazy)
    hipLaunchKernelGGL(spotify_gather_lazy_kernel, dim3(wgrid), dim3(kBlock), 0, st, album_table, n_album_rows, artist_table,
                       F, album_ids, artist_ids, R, *lazy, w.E, w.l2, album_rows);
  else
    hipLaunchKernelGGL(spotify_gather_kernel, dim3(wgrid), dim3(kBlock), 0, st, album_table, n_album_rows, artist_table, F,
                       album_ids, artist_ids, R, w.E, w.l2, album_rows);
  if (int rc = sp_launch_affinity(w, sh, album_ids, artist_ids, w.W, w.partial + R, st)) return rc;
  const int nch = (int)cdiv(D2, 64);
  const size_t rg_lds = (size_t)R * D2 * 4;
  const char* rg_env = getenv("ESR_SPOTIFY_ROWGRAD");  // "global": partner rows from global memory (A/B)
  const bool staged = rg_lds <= 64 * 1024 && !(rg_env && strcmp(rg_env, "global") == 0);
#define ESR_SP_ROWGRAD(NCH)                                                                                         \
  do {                                                                                                              \
    if (staged)                                                                                                     \
      hipLaunchKernelGGL((spotify_rowgrad_kernel<NCH, true>), dim3(wgrid), dim3(kBlock), rg_lds, st, (const float*)w.E, \
                         sh, (const float*)w.l2, (const float*)w.W, regularization, g_album_rows, g_artist_rows,   \
                         w.partial);                                                                                \
    else                                                                                                            \
      hipLaunchKernelGGL((spotify_rowgrad_kernel<NCH, false>), dim3(wgrid), dim3(kBlock), 0, st, (const float*)w.E, \
                         sh, (const float*)w.l2, (const float*)w.W, regularization, g_album_rows, g_artist_rows,   \
                         w.partial);                                                                                \
  } while (0)
  if (nch == 1) ESR_SP_ROWGRAD(1);
  else if (nch == 2) ESR_SP_ROWGRAD(2);
  else ESR_SP_ROWGRAD(4);
#undef ESR_SP_ROWGRAD
  finalize_scalar(w.partial, R + 1, 1.0, loss, st);
  return check_launch("esr_spotify_fwd_bwd");
}

// in esr_optim.hip
extern "C" int esr_momentum_catchup_rows2(float* table0, float* trace0, int32_t* last0, const int32_t* ids0, int modulus0,
                                          float* table1, float* trace1, int32_t* last1, const int32_t* ids1, int modulus1,
                                          int D, int64_t n, int step, float lr, float momentum, esr_stream_t stream);
extern "C" int esr_sparse_momentum_step_multi(float* const* tables, float* const* traces, const int64_t* row_offsets,
                                              int ntables, int D, const int32_t* sorted_vids, const int32_t* perm, int64_t n,
                                              float* grad_rows, float lr, float momentum, esr_stream_t stream);

size_t esr_spotify_train_step_workspace_bytes(int n, int m, int o, int F) {
  if (n <= 0 || m <= 0 || o <= 0 || F <= 0) return 256;
  const size_t R = (size_t)(n + m + o);
  return align_up(sp_layout(nullptr, n, m, o, F).total, 256) + align_up(2 * R * F * 4, 256) + 3 * align_up(2 * R * 4, 256) +
         esr_segment_sort_workspace_bytes(2 * (int64_t)R);
}

// train_spotify.py:77-111 + 238-241 in ONE call: the playlist's rows caught up (lazy optax.sgd(lr, momentum)), the
// six-term loss and its per-occurrence gradient rows, one sort of the virtual rows [album mod rows ; n_album_rows + artist]
// and the whole momentum step on the touched rows of both tables -- eight launches, one foreign call (the step is
// launch-bound: fourteen launches from Python took ~100 us of host time for ~60 us of kernels).
int esr_spotify_train_step(float* album_table, float* album_trace, int32_t* album_last, int64_t n_album_rows,
                           float* artist_table, float* artist_trace, int32_t* artist_last, int64_t n_artists, int F,
                           const int32_t* album_ids, const int32_t* artist_ids, int n, int m, int o, float regularization,
                           int step, float lr, float momentum, float* loss, void* workspace, size_t workspace_bytes,
                           esr_stream_t stream) {
  TraceScope trace_scope_("esr_spotify_train_step");
  if (int rc = sp_check("esr_spotify_train_step", n, m, o, F, n_album_rows, n_artists)) return rc;
  ESR_REQUIRE(album_table && album_trace && album_last && artist_table && artist_trace && artist_last && album_ids &&
                  artist_ids && loss && workspace && step >= 1,
              "esr_spotify_train_step: null pointer or step < 1");
  ESR_REQUIRE(n_album_rows < ((int64_t)1 << 30) && n_artists < ((int64_t)1 << 30), "esr_spotify_train_step: tables too large");
  if (workspace_bytes < esr_spotify_train_step_workspace_bytes(n, m, o, F) || ((uintptr_t)workspace & 15)) {
    set_error("esr_spotify_train_step: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_spotify_train_step_workspace_bytes(n, m, o, F));
    return ESR_EWORKSPACE;
  }
  const int64_t R = n + m + o;
  char* base = (char*)workspace;
  size_t off = align_up(sp_layout(nullptr, n, m, o, F).total, 256);
  float* grads = (float*)(base + off);             // [2R, F]: album gradient rows, then artist gradient rows
  off += align_up(2 * (size_t)R * F * 4, 256);
  int32_t* album_rows = (int32_t*)(base + off);    // [R] hashed album ids (spotify_gather_kernel)
  off += align_up(2 * (size_t)R * 4, 256);
  int32_t* sorted = (int32_t*)(base + off);
  off += align_up(2 * (size_t)R * 4, 256);
  int32_t* perm = (int32_t*)(base + off);
  off += align_up(2 * (size_t)R * 4, 256);
  void* sort_ws = base + off;
  // ESR_SPOTIFY_CATCHUP=launch: the catch-up as its own launch in front (round 3's sequence; A/B and the equality test);
  // default: rows are read caught-up by the gather and written caught-up by the momentum step -- one launch and three
  // dependent memory round trips less
  const char* cu_env = getenv("ESR_SPOTIFY_CATCHUP");
  const bool inline_catchup = !(cu_env && strcmp(cu_env, "launch") == 0);
  const SpLazy lz{album_trace, artist_trace, album_last, artist_last, step, lr, momentum};
  if (!inline_catchup)
    if (int rc = esr_momentum_catchup_rows2(album_table, album_trace, album_last, album_ids, (int)n_album_rows, artist_table,
                                            artist_trace, artist_last, artist_ids, 0, F, R, step, lr, momentum, stream))
      return rc;
  if (int rc = sp_fwd_bwd(album_table, n_album_rows, artist_table, n_artists, F, album_ids, artist_ids, n, m, o,
                          regularization, loss, album_rows, grads, grads + R * F, workspace,
                          sp_layout(nullptr, n, m, o, F).total, stream, inline_catchup ? &lz : nullptr))
    return rc;
  const int32_t* segs[2] = {album_rows, artist_ids};
  const int64_t counts[2] = {R, R};
  const int64_t offsets[2] = {0, n_album_rows};
  if (int rc = esr_segment_sort_ids_multi(segs, counts, offsets, 2, n_album_rows + n_artists, sorted, perm, sort_ws,
                                          esr_segment_sort_workspace_bytes(2 * R), stream))
    return rc;
  float* tables[2] = {album_table, artist_table};
  float* traces[2] = {album_trace, artist_trace};
  int32_t* lasts[2] = {album_last, artist_last};
  const int64_t row_offsets[3] = {0, n_album_rows, n_album_rows + n_artists};
  if (inline_catchup)
    return sparse_momentum_step_lazy2(tables, traces, lasts, row_offsets, 2, F, sorted, perm, 2 * R, grads, lr, momentum, step,
                                      as_stream(stream));
  return esr_sparse_momentum_step_multi(tables, traces, row_offsets, 2, F, sorted, perm, 2 * R, grads, lr, momentum, stream);
}

int esr_spotify_affinity_all(const float* album_table, int64_t n_album_rows, const float* artist_table,
                             int64_t n_artists, int F, const int32_t* ctx_album, const int32_t* ctx_artist, int n,
                             const int32_t* all_albums, const int32_t* all_artists, int64_t T, float* affinity,
                             esr_stream_t stream) {
  if (int rc = sp_check("esr_spotify_affinity_all", n, 1, 1, F, n_album_rows, n_artists)) return rc;
  ESR_REQUIRE(T > 0, "esr_spotify_affinity_all: T=%lld", (long long)T);
  ESR_REQUIRE(album_table && artist_table && ctx_album && ctx_artist && all_albums && all_artists && affinity,
              "esr_spotify_affinity_all: null pointer");
  const int grid = (int)std::min<int64_t>(cdiv(T, kBlock / 16), 4096);
  hipLaunchKernelGGL(spotify_affinity_all_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), album_table,
                     n_album_rows, artist_table, F, ctx_album, ctx_artist, n, all_albums, all_artists, T, affinity);
  return check_launch("esr_spotify_affinity_all");
}

}  // extern "C"
