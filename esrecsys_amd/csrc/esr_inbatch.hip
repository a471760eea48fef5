// In-batch-negative sampled softmax over the dense B x B query.candidate score matrix,
// forward + backward, on the FP32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 157 TF peak).
//
// Build-defined (north_star); no reference counterpart (closest precedent: the dense
// next.context^T score matrices of spotify/models.py:74-87).
//     S = scale * Q C^T ;  ce_i = logsumexp_j S_ij - S_ii
//     loss = (sum_i ce_i + lam * sum_i [reg(q_i) + reg(c_i)]) / bs      reg(e) = relu(|e| - 1)
//     gQ = scale (softmax(S) - I) C / bs + dreg(Q) ;  gC = scale (softmax(S) - I)^T Q / bs + dreg(C)
//
// The score matrix never leaves the chip (flash-attention structure, 4 GEMM units of 2 B^2 D flop):
//   pass Q ("owned" = Q, "streamed" = C): online softmax; gQ_i = scale (sum_j p_ij c_j / l_i - c_i)/bs, lse_i
//   pass C ("owned" = C, "streamed" = Q): p_ij = exp(S_ij - lse_i);  gC_j = scale (sum_i p_ij q_i - q_j)/bs
//
// Decomposition.  A workgroup owns 32 rows of the owned matrix and holds them in VGPRs as the
// MFMA B operand for the whole kernel.  Its 8 waves split the streamed rows: wave w takes 32-row
// chunks w, w+8, ...  Each wave stages its chunk in a private LDS tile (no barriers in the main
// loop; the 2 waves per SIMD hide each other's load / softmax phases behind MFMAs).  Per chunk:
//   S^T = Y_chunk X^T        64 MFMAs; D layout lane = owned row, reg = streamed row, so all row
//                            statistics (max, sum, lse) are lane-local ("swapped" product)
//   O^T += Y_chunk^T P^T     64 MFMAs; P feeds the B operand straight from the S^T registers
// The k index of an MFMA is free to permute (A and B agree), which lets both LDS operand reads be
// one conflict-free ds_read_b128: k-step (kk, m) <-> d = 8 kk + 4 (lane>>5) + m, and output row
// i of d-block db <-> d = 4 i + db.  LDS rows are padded by 16 B (odd number of 16-B slots).
// The 8 partial (m, l, O) are merged once per workgroup through LDS.
#include "esr_common.h"

namespace esr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kIbWaves = 8;
constexpr int kIbRows = 32;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// streamed-row index of accumulator register r on lane-half h (32x32 MFMA C/D layout)
__device__ __forceinline__ constexpr int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// DP = the tile width (32, 64 or 128); D = the real embedding width, a multiple of 4 with D <= DP: columns D .. DP - 1
// are zero padding that never touches memory (D = 96, pinterest/sweep.yaml:13-14, runs as DP = 128).
// PAD = false is the D == DP instantiation: no column test survives in its loads (with the test in place the D = 128
// kernel went from 280 us to 400 us per pass).
template <int DP, bool QSIDE, bool PAD>
__global__ __launch_bounds__(kIbWaves * 64, 2) void inbatch_kernel(
    const float* __restrict__ X, const float* __restrict__ Y, int64_t B, int64_t nv, int D_, float scale, float lam,
    float inv_bs, float* __restrict__ lse2, float* __restrict__ lse_nat, float* __restrict__ gX,
    double* __restrict__ loss_part) {
  // B = rows rounded up to a multiple of 32 (tiling); nv = rows that exist.  Rows >= nv are padding: their loads are
  // clamped to the last real row, as streamed rows they are masked out of every softmax (score -inf in pass Q,
  // lse = +inf in pass C), as owned rows they produce no output.
  const int D = PAD ? D_ : DP;     // a compile-time constant without padding
  constexpr int KK = DP / 8;       // S-phase k-groups (4 MFMAs each)
  constexpr int DB = DP / 32;      // output d-blocks
  constexpr int STRIDE = DP + 4;   // LDS row stride in floats (+16 B)
  constexpr int TILE = kIbRows * STRIDE;
  constexpr int NLD = DP / 8;      // float4 staging loads per lane per chunk
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int OBUF = kIbWaves * 64 * (16 * DB);  // merge buffer, floats
  constexpr int LDS_FLOATS = (kIbWaves * TILE > OBUF ? kIbWaves * TILE : OBUF) + kIbWaves * 64 + 64;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* const stat = lds + (kIbWaves * TILE > OBUF ? kIbWaves * TILE : OBUF);  // [8][32][2] m,l ; then [32] norms
  float* const normbuf = stat + kIbWaves * 64;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int64_t x0 = (int64_t)blockIdx.x * kIbRows;
  float* const tile = lds + w * TILE;
  const float sl2 = scale * kLog2e;

  // Owned rows -> B operand registers: xr[kk][m] = X[x0 + j][8 kk + 4 h + m]
  float xr[KK][4];
  {
    const float* xp = X + min(x0 + j, nv - 1) * D + 4 * h;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const float4 v = (!PAD || 8 * kk + 4 * h < D) ? *reinterpret_cast<const float4*>(xp + 8 * kk) : zero4;
      xr[kk][0] = v.x; xr[kk][1] = v.y; xr[kk][2] = v.z; xr[kk][3] = v.w;
    }
  }

  f32x16 acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
  float m2 = -INFINITY, l = 0.f;

  const int nchunks = (int)(B / kIbRows);
  for (int t = w; t < nchunks; t += kIbWaves) {
    const int64_t y0 = (int64_t)t * kIbRows;
    // ---- stage the chunk: coalesced 16 B loads -> padded LDS tile (wave-private) ----
    {
      float4 st[NLD];
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        const int idx = q * 64 + lane;
        const int row = idx / (DP / 4), c4 = idx % (DP / 4);
        st[q] = (!PAD || 4 * c4 < D) ? *reinterpret_cast<const float4*>(Y + min(y0 + row, nv - 1) * D + 4 * c4) : zero4;
      }
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        const int idx = q * 64 + lane;
        const int row = idx / (DP / 4), c4 = idx % (DP / 4);
        *reinterpret_cast<float4*>(tile + row * STRIDE + 4 * c4) = st[q];
      }
    }
    // LDS ops of one wave complete in order; this only stops the compiler moving reads above writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- S^T = Y_chunk X^T : lane (j, h), reg r  <->  S[owned j][streamed mfma_row(r, h)] ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(tile + j * STRIDE + 8 * kk + 4 * h);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[kk][0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[kk][1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[kk][2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[kk][3], s, 0, 0, 0);
    }

    // ---- probabilities (log2 domain) ----
    if (QSIDE) {
      float mloc = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = (y0 + mfma_row(r, h) < nv) ? s[r] * sl2 : -INFINITY;  // padding candidates leave the softmax
        mloc = fmaxf(mloc, s[r]);
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));  // both halves hold the same owned row
      const float mnew = fmaxf(m2, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m2 - mnew);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - mnew);
        psum += s[r];
      }
      l = l * alpha + psum;
      m2 = mnew;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] *= alpha;
    } else {
      // lse2[y0 + mfma_row(4 g + m, h)] = lse2[y0 + 8 g + 4 h + m]
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 lv = *reinterpret_cast<const float4*>(lse2 + y0 + 8 * g + 4 * h);
        s[4 * g + 0] = __builtin_amdgcn_exp2f(s[4 * g + 0] * sl2 - lv.x);
        s[4 * g + 1] = __builtin_amdgcn_exp2f(s[4 * g + 1] * sl2 - lv.y);
        s[4 * g + 2] = __builtin_amdgcn_exp2f(s[4 * g + 2] * sl2 - lv.z);
        s[4 * g + 3] = __builtin_amdgcn_exp2f(s[4 * g + 3] * sl2 - lv.w);
      }
    }

    // ---- O^T += Y_chunk^T P^T : acc[db][r'] <-> O[owned j][d = DB * mfma_row(r', h) + db] ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* ap = tile + mfma_row(r, h) * STRIDE + DB * j;
      if constexpr (DB == 4) {
        const float4 a = *reinterpret_cast<const float4*>(ap);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, s[r], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, s[r], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, s[r], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, s[r], acc[3], 0, 0, 0);
      } else if constexpr (DB == 2) {
        const float2 a = *reinterpret_cast<const float2*>(ap);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, s[r], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, s[r], acc[1], 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[0], s[r], acc[0], 0, 0, 0);
      }
    }
    // next iteration overwrites the tile: all its reads have been consumed by the MFMAs above
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ================= merge the 8 per-wave partials through LDS =================
  float xn2 = 0.f;  // |x_j|^2
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
#pragma unroll
    for (int m = 0; m < 4; ++m) xn2 = fmaf(xr[kk][m], xr[kk][m], xn2);
  xn2 += __shfl_xor(xn2, 32, 64);
  const float xnorm = sqrtf(xn2);

  float diag = 0.f;  // x_j . y_j (the positive pair), QSIDE wave 0 only
  if (QSIDE && w == 0) {
    const float* yp = Y + min(x0 + j, nv - 1) * D + 4 * h;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const float4 v = (!PAD || 8 * kk + 4 * h < D) ? *reinterpret_cast<const float4*>(yp + 8 * kk) : zero4;
      diag = fmaf(xr[kk][0], v.x, diag);
      diag = fmaf(xr[kk][1], v.y, diag);
      diag = fmaf(xr[kk][2], v.z, diag);
      diag = fmaf(xr[kk][3], v.w, diag);
    }
    diag += __shfl_xor(diag, 32, 64);
  }

  __syncthreads();  // every wave is done with its tile
  float f = 1.f;
  if (QSIDE) {
    const float ltot = l + __shfl_xor(l, 32, 64);
    if (h == 0) {
      stat[(w * 32 + j) * 2] = m2;
      stat[(w * 32 + j) * 2 + 1] = ltot;
    }
    __syncthreads();
    float M = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < kIbWaves; ++ww) M = fmaxf(M, stat[(ww * 32 + j) * 2]);
    float L = 0.f;
#pragma unroll
    for (int ww = 0; ww < kIbWaves; ++ww)
      L += stat[(ww * 32 + j) * 2 + 1] * __builtin_amdgcn_exp2f(stat[(ww * 32 + j) * 2] - M);
    f = __builtin_amdgcn_exp2f(m2 - M) / L;
    if (w == 0) {
      const float lse_l2 = M + __builtin_amdgcn_logf(L);  // v_log_f32 = log2
      const bool real = x0 + j < nv;
      if (h == 0) {
        lse2[x0 + j] = real ? lse_l2 : INFINITY;  // pass C: exp2(s - inf) = 0 for a padding query
        if (lse_nat && real) lse_nat[x0 + j] = lse_l2 * kLn2;
      }
      // loss partial: sum_j ce_j + lam * reg(q_j)
      double part = 0.0;
      if (h == 0 && real) part = (double)(lse_l2 * kLn2) - (double)(scale * diag) + (double)(lam * fmaxf(xnorm - 1.f, 0.f));
      part = wave_sum_d(part);
      if (lane == 0) loss_part[blockIdx.x] = part;
    }
  } else if (w == 0) {
    double part = (h == 0 && x0 + j < nv) ? (double)(lam * fmaxf(xnorm - 1.f, 0.f)) : 0.0;
    part = wave_sum_d(part);
    if (lane == 0) loss_part[blockIdx.x] = part;
  }
  if (w == 0 && h == 0) normbuf[j] = xnorm;

  // obuf[w][e = r*DB + db][lane]
  float* const obuf = lds;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) obuf[(w * (16 * DB) + r * DB + db) * 64 + lane] = acc[db][r] * f;
  __syncthreads();

  // 16 * 64 (r, lane') items, DB consecutive d each; 512 threads -> 2 items per thread
#pragma unroll
  for (int it = tid; it < 16 * 64; it += kIbWaves * 64) {
    const int r = it >> 6, ln = it & 63;
    const int row = ln & 31, d0 = DB * mfma_row(r, ln >> 5);
    if (x0 + row >= nv || (PAD && d0 >= D)) continue;  // padding row / padding columns: no gradient exists
    float v[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      float a = 0.f;
#pragma unroll
      for (int ww = 0; ww < kIbWaves; ++ww) a += obuf[(ww * (16 * DB) + r * DB + db) * 64 + ln];
      v[db] = a;
    }
    const float nrm = normbuf[row];
    const float creg = nrm > 1.f ? lam / nrm : 0.f;
    const float* xrow = X + (x0 + row) * D + d0;
    const float* yrow = Y + (x0 + row) * D + d0;
    float* grow = gX + (x0 + row) * D + d0;
    float o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = (scale * (v[db] - yrow[db]) + creg * xrow[db]) * inv_bs;
    if constexpr (DB == 4) {
      *reinterpret_cast<float4*>(grow) = make_float4(o[0], o[1], o[2], o[3]);
    } else if constexpr (DB == 2) {
      *reinterpret_cast<float2*>(grow) = make_float2(o[0], o[1]);
    } else {
      grow[0] = o[0];
    }
  }
}

// Wide rows (128 < D <= 512), DP = 256 or 512: the 8 waves of a workgroup work on the SAME 32-row chunk of the streamed
// matrix (one shared LDS tile) and split the embedding dimension instead: wave w owns the DP / 8 columns of panel w.
//   S^T partial over the panel's columns (K = DP / 8) -> the 8 partials are exchanged through LDS and summed in a fixed
//   order (every wave ends up with the same S^T bits, so the softmax statistics agree without a merge) -> softmax ->
//   O^T of the panel's columns only (DP / 256 accumulator blocks).
// No flop is repeated; the price is three barriers and a 32 KB exchange per chunk.  Same index permutations as above.
template <int DP, bool QSIDE>
__global__ __launch_bounds__(kIbWaves * 64, 1) void inbatch_wide_kernel(
    const float* __restrict__ X, const float* __restrict__ Y, int64_t B, int64_t nv, int D, float scale, float lam,
    float inv_bs, float* __restrict__ lse2, float* __restrict__ lse_nat, float* __restrict__ gX,
    double* __restrict__ loss_part) {
  constexpr int PW = DP / kIbWaves;  // panel width: 32 or 64 columns per wave
  constexpr int KKW = PW / 8;        // S-phase k-groups per wave
  constexpr int DBW = PW / 32;       // output d-blocks per wave
  constexpr int STRIDE = DP + 4;
  constexpr int NLD = DP / 64;       // float4 staging loads per thread per chunk (512 threads)
  __shared__ __attribute__((aligned(16))) float tile[kIbRows * STRIDE];
  __shared__ float sx[kIbWaves * 16 * 64];   // S^T partials [wave][reg][lane]
  __shared__ float red[kIbWaves * 64];       // |x|^2 / x.y panel partials [wave][lane]
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int64_t x0 = (int64_t)blockIdx.x * kIbRows;
  const float sl2 = scale * kLog2e;
  const int pbase = w * PW;

  float xr[KKW][4];  // xr[kk][m] = X[x0 + j][pbase + 8 kk + 4 h + m]
  {
    const float* xp = X + min(x0 + j, nv - 1) * D + pbase + 4 * h;
#pragma unroll
    for (int kk = 0; kk < KKW; ++kk) {
      const float4 v = (pbase + 8 * kk + 4 * h < D) ? *reinterpret_cast<const float4*>(xp + 8 * kk) : zero4;
      xr[kk][0] = v.x; xr[kk][1] = v.y; xr[kk][2] = v.z; xr[kk][3] = v.w;
    }
  }
  f32x16 acc[DBW];
#pragma unroll
  for (int db = 0; db < DBW; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
  float m2 = -INFINITY, l = 0.f;

  const int nchunks = (int)(B / kIbRows);
  for (int t = 0; t < nchunks; ++t) {
    const int64_t y0 = (int64_t)t * kIbRows;
    __syncthreads();  // the previous chunk's tile and exchange buffer are no longer read
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = q * (kIbWaves * 64) + tid;
      const int row = idx / (DP / 4), c4 = idx % (DP / 4);
      const float4 v = (4 * c4 < D) ? *reinterpret_cast<const float4*>(Y + min(y0 + row, nv - 1) * D + 4 * c4) : zero4;
      *reinterpret_cast<float4*>(tile + row * STRIDE + 4 * c4) = v;
    }
    __syncthreads();
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KKW; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(tile + j * STRIDE + pbase + 8 * kk + 4 * h);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[kk][0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[kk][1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[kk][2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[kk][3], s, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sx[(w * 16 + r) * 64 + lane] = s[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = 0.f;
#pragma unroll
      for (int ww = 0; ww < kIbWaves; ++ww) a += sx[(ww * 16 + r) * 64 + lane];
      s[r] = a;
    }
    if (QSIDE) {
      float mloc = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = (y0 + mfma_row(r, h) < nv) ? s[r] * sl2 : -INFINITY;
        mloc = fmaxf(mloc, s[r]);
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mnew = fmaxf(m2, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m2 - mnew);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - mnew);
        psum += s[r];
      }
      l = l * alpha + psum;
      m2 = mnew;
#pragma unroll
      for (int db = 0; db < DBW; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] *= alpha;
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 lv = *reinterpret_cast<const float4*>(lse2 + y0 + 8 * g + 4 * h);
        s[4 * g + 0] = __builtin_amdgcn_exp2f(s[4 * g + 0] * sl2 - lv.x);
        s[4 * g + 1] = __builtin_amdgcn_exp2f(s[4 * g + 1] * sl2 - lv.y);
        s[4 * g + 2] = __builtin_amdgcn_exp2f(s[4 * g + 2] * sl2 - lv.z);
        s[4 * g + 3] = __builtin_amdgcn_exp2f(s[4 * g + 3] * sl2 - lv.w);
      }
    }
    // O^T of this wave's panel: acc[db][r'] <-> O[owned j][d = pbase + DBW * mfma_row(r', h) + db]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* ap = tile + mfma_row(r, h) * STRIDE + pbase + DBW * j;
      if constexpr (DBW == 2) {
        const float2 a = *reinterpret_cast<const float2*>(ap);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, s[r], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, s[r], acc[1], 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[0], s[r], acc[0], 0, 0, 0);
      }
    }
  }

  // |x_j|^2 and (pass Q) x_j . y_j: panel partials -> LDS -> every wave sums the 8 in order
  float xn2 = 0.f, diag = 0.f;
#pragma unroll
  for (int kk = 0; kk < KKW; ++kk)
#pragma unroll
    for (int m = 0; m < 4; ++m) xn2 = fmaf(xr[kk][m], xr[kk][m], xn2);
  xn2 += __shfl_xor(xn2, 32, 64);
  if (QSIDE) {
    const float* yp = Y + min(x0 + j, nv - 1) * D + pbase + 4 * h;
#pragma unroll
    for (int kk = 0; kk < KKW; ++kk) {
      const float4 v = (pbase + 8 * kk + 4 * h < D) ? *reinterpret_cast<const float4*>(yp + 8 * kk) : zero4;
      diag = fmaf(xr[kk][0], v.x, diag);
      diag = fmaf(xr[kk][1], v.y, diag);
      diag = fmaf(xr[kk][2], v.z, diag);
      diag = fmaf(xr[kk][3], v.w, diag);
    }
    diag += __shfl_xor(diag, 32, 64);
  }
  __syncthreads();
  red[w * 64 + lane] = h == 0 ? xn2 : diag;  // lanes 0..31: |x|^2 partial of row j ; lanes 32..63: x.y partial
  __syncthreads();
  float tot_n2 = 0.f, tot_diag = 0.f;
#pragma unroll
  for (int ww = 0; ww < kIbWaves; ++ww) {
    tot_n2 += red[ww * 64 + j];
    tot_diag += red[ww * 64 + 32 + j];
  }
  const float xnorm = sqrtf(tot_n2);
  float f = 1.f;
  if (QSIDE) {
    const float ltot = l + __shfl_xor(l, 32, 64);
    f = 1.0f / ltot;
    if (w == 0) {
      const float lse_l2 = m2 + __builtin_amdgcn_logf(ltot);
      const bool real = x0 + j < nv;
      if (h == 0) {
        lse2[x0 + j] = real ? lse_l2 : INFINITY;
        if (lse_nat && real) lse_nat[x0 + j] = lse_l2 * kLn2;
      }
      double part = 0.0;
      if (h == 0 && real)
        part = (double)(lse_l2 * kLn2) - (double)(scale * tot_diag) + (double)(lam * fmaxf(xnorm - 1.f, 0.f));
      part = wave_sum_d(part);
      if (lane == 0) loss_part[blockIdx.x] = part;
    }
  } else if (w == 0) {
    double part = (h == 0 && x0 + j < nv) ? (double)(lam * fmaxf(xnorm - 1.f, 0.f)) : 0.0;
    part = wave_sum_d(part);
    if (lane == 0) loss_part[blockIdx.x] = part;
  }
  // gradient rows of this wave's panel, straight from the accumulators (the epilogue runs once per 32 owned rows)
  if (x0 + j < nv) {
    const float creg = xnorm > 1.f ? lam / xnorm : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d0 = pbase + DBW * mfma_row(r, h);
      if (d0 >= D) continue;
      const float* xrow = X + (x0 + j) * D + d0;
      const float* yrow = Y + (x0 + j) * D + d0;
      float* grow = gX + (x0 + j) * D + d0;
#pragma unroll
      for (int db = 0; db < DBW; ++db)
        grow[db] = (scale * (acc[db][r] * f - yrow[db]) + creg * xrow[db]) * inv_bs;
    }
  }
}

struct InbatchWs {
  float* lse2;        // [B]
  double* loss_part;  // [2 * B/32]
};
static size_t inbatch_ws_layout(int64_t B, char* base, InbatchWs* ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  float* a = (float*)take(sizeof(float) * (size_t)B);
  double* b = (double*)take(sizeof(double) * 2 * (size_t)(B / kIbRows + 1));
  if (ws) *ws = InbatchWs{a, b};
  return off;
}

template <int DP>
static void inbatch_launch(const float* Q, const float* C, int64_t B, int64_t nv, int D, float scale, float lam,
                           float inv_bs, float* lse_nat, float* gQ, float* gC, const InbatchWs& ws, hipStream_t st) {
  const int nblk = (int)(B / kIbRows);
  if constexpr (DP <= 128) {
    if (D == DP) {
      hipLaunchKernelGGL((inbatch_kernel<DP, true, false>), dim3(nblk), dim3(kIbWaves * 64), 0, st, Q, C, B, nv, D, scale,
                         lam, inv_bs, ws.lse2, lse_nat, gQ, ws.loss_part);
      hipLaunchKernelGGL((inbatch_kernel<DP, false, false>), dim3(nblk), dim3(kIbWaves * 64), 0, st, C, Q, B, nv, D, scale,
                         lam, inv_bs, ws.lse2, (float*)nullptr, gC, ws.loss_part + nblk);
    } else {
      hipLaunchKernelGGL((inbatch_kernel<DP, true, true>), dim3(nblk), dim3(kIbWaves * 64), 0, st, Q, C, B, nv, D, scale,
                         lam, inv_bs, ws.lse2, lse_nat, gQ, ws.loss_part);
      hipLaunchKernelGGL((inbatch_kernel<DP, false, true>), dim3(nblk), dim3(kIbWaves * 64), 0, st, C, Q, B, nv, D, scale,
                         lam, inv_bs, ws.lse2, (float*)nullptr, gC, ws.loss_part + nblk);
    }
  } else {
    hipLaunchKernelGGL((inbatch_wide_kernel<DP, true>), dim3(nblk), dim3(kIbWaves * 64), 0, st, Q, C, B, nv, D, scale,
                       lam, inv_bs, ws.lse2, lse_nat, gQ, ws.loss_part);
    hipLaunchKernelGGL((inbatch_wide_kernel<DP, false>), dim3(nblk), dim3(kIbWaves * 64), 0, st, C, Q, B, nv, D, scale,
                       lam, inv_bs, ws.lse2, (float*)nullptr, gC, ws.loss_part + nblk);
  }
}

}  // namespace esr

using namespace esr;

extern "C" {

size_t esr_inbatch_workspace_bytes(int64_t B, int D) {
  (void)D;
  if (B <= 0) return 256;
  return inbatch_ws_layout(cdiv(B, kIbRows) * kIbRows, nullptr, nullptr);
}

int esr_inbatch_softmax_fwd_bwd(const float* Q, const float* C, int64_t B, int D, float scale, float regularization,
                                float batch_size, float* loss, float* lse, float* gQ, float* gC, void* workspace,
                                size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_inbatch_softmax_fwd_bwd");
  ESR_REQUIRE(B > 0, "esr_inbatch_softmax_fwd_bwd: B=%lld must be positive", (long long)B);
  ESR_REQUIRE(D > 0 && D <= 512 && D % 4 == 0, "esr_inbatch_softmax_fwd_bwd: D=%d not supported (a multiple of 4, at most 512)", D);
  ESR_REQUIRE(Q && C && loss && gQ && gC, "esr_inbatch_softmax_fwd_bwd: null pointer");
  ESR_REQUIRE(batch_size != 0.f, "esr_inbatch_softmax_fwd_bwd: batch_size must be non-zero");
  ESR_REQUIRE((((uintptr_t)Q | (uintptr_t)C | (uintptr_t)gQ | (uintptr_t)gC) & 15) == 0,
              "esr_inbatch_softmax_fwd_bwd: matrices must be 16-byte aligned");
  if (!workspace || workspace_bytes < esr_inbatch_workspace_bytes(B, D) || ((uintptr_t)workspace & 15)) {
    set_error("esr_inbatch_softmax_fwd_bwd: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes,
              esr_inbatch_workspace_bytes(B, D));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  InbatchWs ws;
  const int64_t Bp = cdiv(B, kIbRows) * kIbRows;  // tiles of 32 rows; rows >= B are masked padding
  inbatch_ws_layout(Bp, (char*)workspace, &ws);
  const float inv_bs = 1.0f / batch_size;
  // tile width = the smallest of 32 / 64 / 128 (register-resident rows) or 256 / 512 (column panels) that holds D
  if (D <= 32) inbatch_launch<32>(Q, C, Bp, B, D, scale, regularization, inv_bs, lse, gQ, gC, ws, st);
  else if (D <= 64) inbatch_launch<64>(Q, C, Bp, B, D, scale, regularization, inv_bs, lse, gQ, gC, ws, st);
  else if (D <= 128) inbatch_launch<128>(Q, C, Bp, B, D, scale, regularization, inv_bs, lse, gQ, gC, ws, st);
  else if (D <= 256) inbatch_launch<256>(Q, C, Bp, B, D, scale, regularization, inv_bs, lse, gQ, gC, ws, st);
  else inbatch_launch<512>(Q, C, Bp, B, D, scale, regularization, inv_bs, lse, gQ, gC, ws, st);
  const int nblk = (int)(Bp / kIbRows);
  finalize_scalar(ws.loss_part, 2 * nblk, 1.0 / (double)batch_size, loss, st);
  return check_launch("esr_inbatch_softmax_fwd_bwd");
}

}  // extern "C"
