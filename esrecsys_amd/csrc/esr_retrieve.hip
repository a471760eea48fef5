// esr_retrieve.hip -- batched brute-force retrieval (BASELINE config 5):
//   scores[q, n] = queries[q] . candidates[n]   for nq queries x N candidates, then top-k per query,
//   descending, ties -> lower index (jax.lax.top_k: pinterest/make_recommendations.py:62-65, batched over the
//   scenes of :123-132; spotify/train_spotify.py:120 top_k(500) over all tracks).
//
// The score matrix is MFMA-bound (2 nq N D flop against (nq + N) D bytes) and is never materialised in full:
//   split   f32 rows -> bf16 planes (1 plane = "bf16" mode; 3 exact planes = f32-equivalent mode, six cross terms
//           a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 as in esr_inbatch3.hip), zero-padded to tile multiples;
//   gemm    256 x 128 tiles, 8 waves (64 x 64 each, v_mfma_f32_32x32x16_bf16), K in 32-element stages that are
//           DMA'd global -> LDS (global_load_lds_dwordx4, XOR-swizzled on the source address), XCD-aware tile
//           order (the 32 tiles resident on one XCD form a 4 x 8 block and share their operand rows in L2);
//           epilogue of the FIRST candidate chunk stores the tile densely; every later chunk only appends the
//           (score, index) pairs that reach the query's current k-th best score (tau) to a per-query list;
//   select  one workgroup per query: MSB-first radix select (11-bit digits) on the 64-bit composite
//           (order-preserving score key, ~index) -- unique composites, so ties need no special case -- then a
//           bitonic sort of the k survivors in LDS; it rewrites the list head = running top-k and tau.
// The result is a pure function of the inputs: the append order is not deterministic, the selected SET and its
// final order are (total order on the composite).
#include "esr_common.h"

#include <algorithm>

namespace esr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kGM = 256, kGN = 128, kGK = 16;  // tile rows (queries), tile cols (candidates), k per stage
constexpr int kGThreads = 512;
constexpr int kTileRowBytes = kGK * 2;                      // 32 B of one plane row per stage
constexpr int kPlaneStage = (kGM + kGN) * kTileRowBytes;    // 12288 B: A tile then B tile of one plane
constexpr int kPiecesPerPlane = kPlaneStage / 1024;         // 12 DMA instructions (32 rows x 32 B each)
constexpr int kGroupM = 4;                                  // tile-order group height (tiles)
constexpr int kSelThreads = 256;
constexpr int kSelMaxK = 1024;
constexpr int kFirstChunk = 8192;

// ---------------------------------------------------------------------------------------------------
// P == 2 (mode 2, "f16x2"): two fp16 planes of x * 2^e instead of three bf16 ones -- x * 2^e = x1 + x2 to 2^-24 relative
// with round-to-nearest planes, a.b ~= a2 b1 + a1 b2 + a1 b1 (three MFMAs per product instead of six; see
// esr_inbatch2h.hip).  e is one exponent per matrix, from its largest |element| (max |x * 2^e| in [2^13, 2^14)):
// absmax_part_kernel -> slots, absmax_exp_kernel -> the exponent word the split and the GEMM epilogue read.
// ---------------------------------------------------------------------------------------------------
// Plane code P of the split and GEMM templates: 1 = one bf16 plane, 3 = three exact bf16 planes, 2 = two scaled fp16
// planes, 4 (round 6) = ONE scaled fp16 plane -- the hi plane of code 2 alone: the one-term filter of mode 3.
__host__ __device__ constexpr int plane_count(int P) { return P == 4 ? 1 : P; }
__host__ __device__ constexpr bool plane_f16(int P) { return P == 2 || P == 4; }
constexpr int kAbsBlocks = 1024;
__global__ __launch_bounds__(kBlock) void absmax_part_kernel(const float* __restrict__ X, int64_t n,
                                                            float* __restrict__ slots) {
  __shared__ float red[kBlock / 64];
  float m = 0.f;
  const bool vec = ((uintptr_t)X & 15) == 0;
  const int64_t n4 = vec ? n >> 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 f = reinterpret_cast<const float4*>(X)[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w))));
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    m = fmaxf(m, fabsf(X[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = red[0];
    for (int i = 1; i < kBlock / 64; ++i) t = fmaxf(t, red[i]);
    slots[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(kBlock) void absmax_exp_kernel(const float* __restrict__ slots, int nslots,
                                                           int* __restrict__ exp_out) {
  __shared__ float red[kBlock / 64];
  float m = 0.f;
  for (int i = threadIdx.x; i < nslots; i += kBlock) m = fmaxf(m, slots[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = red[0];
    for (int i = 1; i < kBlock / 64; ++i) t = fmaxf(t, red[i]);
    int e = 0;
    if (t > 0.f && t < INFINITY) {
      int x;
      frexpf(t, &x);  // t = m 2^x, m in [0.5, 1)
      e = 14 - x;
      e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    exp_out[0] = e;
  }
}

// Row statistics for mode 3 (the one-term filter, round 6): one wave per row -- the row's 2-norm (norms[r], optional) and,
// per workgroup, the largest |element| (slots_abs) and the largest row norm (slots_nrm) it saw: what the exponent of the
// scaled planes and the error bound of a one-plane score need from a matrix, in ONE read of it.
__global__ __launch_bounds__(kBlock) void rowstat_kernel(const float* __restrict__ X, int64_t n_rows, int D,
                                                        float* __restrict__ norms, float* __restrict__ slots_abs,
                                                        float* __restrict__ slots_nrm) {
  __shared__ float red[2][kBlock / 64];
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  const bool vec = (D & 3) == 0 && ((uintptr_t)X & 15) == 0;
  float amax = 0.f, nmax = 0.f;
  for (int64_t r = wave; r < n_rows; r += nwaves) {
    const float* x = X + r * D;
    float ss = 0.f;
    if (vec) {
      for (int d = lane * 4; d < D; d += 256) {
        const float4 f = *reinterpret_cast<const float4*>(x + d);
        ss = fmaf(f.x, f.x, fmaf(f.y, f.y, fmaf(f.z, f.z, fmaf(f.w, f.w, ss))));
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w))));
      }
    } else {
      for (int d = lane; d < D; d += 64) {
        ss = fmaf(x[d], x[d], ss);
        amax = fmaxf(amax, fabsf(x[d]));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nr = sqrtf(ss);
    if (norms && lane == 0) norms[r] = nr;
    nmax = fmaxf(nmax, nr);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if (lane == 0) { red[0][threadIdx.x >> 6] = amax; red[1][threadIdx.x >> 6] = nmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = red[0][0], b = red[1][0];
    for (int i = 1; i < kBlock / 64; ++i) { a = fmaxf(a, red[0][i]); b = fmaxf(b, red[1][i]); }
    slots_abs[blockIdx.x] = a;
    slots_nrm[blockIdx.x] = b;
  }
}
// max over slots -> out[0] (one workgroup)
__global__ __launch_bounds__(kBlock) void slots_max_kernel(const float* __restrict__ slots, int nslots, float* __restrict__ out) {
  __shared__ float red[kBlock / 64];
  float m = 0.f;
  for (int i = threadIdx.x; i < nslots; i += kBlock) m = fmaxf(m, slots[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = red[0];
    for (int i = 1; i < kBlock / 64; ++i) t = fmaxf(t, red[i]);
    out[0] = t;
  }
}

// ---------------------------------------------------------------------------------------------------
// f32 rows -> P bf16 planes, zero padded (rows >= n_rows, cols >= D), K-BLOCK-MAJOR: plane[kb][row][16] with
// kb = d / 16 -- the 32 rows x 16 k that one DMA instruction moves are 1 KiB of contiguous global memory
// ---------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(kBlock) void split_planes_kernel(const float* __restrict__ X, int64_t n_rows, int D,
                                                             int64_t rows_pad, int Dp, int64_t plane_elems,
                                                             __bf16* __restrict__ out,
                                                             const int* __restrict__ exp_ptr = nullptr) {
  const float mul = plane_f16(P) ? ldexpf(1.f, exp_ptr[0]) : 1.f;
  const int quads = Dp >> 2;
  const int64_t total = rows_pad * quads;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / quads;
    const int c = (int)(i - r * quads) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < n_rows) {
      if ((D & 3) == 0 && c + 3 < D) {
        const float4 f = *reinterpret_cast<const float4*>(X + r * D + c);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < D) v[e] = X[r * D + c + e];
      }
    }
    __bf16* dst = out + ((int64_t)(c >> 4) * rows_pad + r) * 16 + (c & 15);
    if (plane_f16(P)) {
      f16x4 h1, h2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xs = v[e] * mul;  // exact: power of two
        const _Float16 a = (_Float16)xs;
        h1[e] = a;
        h2[e] = (_Float16)(xs - (float)a);
      }
      *reinterpret_cast<f16x4*>(dst) = h1;
      if (P == 2) *reinterpret_cast<f16x4*>(dst + plane_elems) = h2;
      continue;
    }
    bf16x4 p1, p2, p3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 a = (__bf16)v[e];
      p1[e] = a;
      if (P == 3) {
        const float r1 = v[e] - (float)a;
        const __bf16 b = (__bf16)r1;
        p2[e] = b;
        p3[e] = (__bf16)(r1 - (float)b);
      }
    }
    *reinterpret_cast<bf16x4*>(dst) = p1;
    if (P == 3) {
      *reinterpret_cast<bf16x4*>(dst + plane_elems) = p2;
      *reinterpret_cast<bf16x4*>(dst + 2 * plane_elems) = p3;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// score GEMM
// ---------------------------------------------------------------------------------------------------
struct GemmOut {
  float* S;            // dense: [M][ldS]
  int64_t ldS;
  const float* tau;    // filtered: per-query threshold
  int32_t* cnt;        // per-query list length
  int2* pairs;         // [M][ppitch] (score bits, index)
  int64_t ppitch;
  int32_t gbase, gstep;  // global index of local candidate n = gbase + n * gstep
  const int* exps;       // P == 2: {eq, ec}, the plane exponents (scores = accumulators * 2^-(eq + ec))
};

// Wait until at most N of this wave's DMAs are outstanding, then the workgroup barrier.  Written by hand:
// __syncthreads() carries a release fence that makes hipcc wait vmcnt(0), i.e. for the stages that were issued
// only to be in flight across this barrier.
template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

#define ESR_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#ifdef ESR_GEMM_TIMING  // debug build only (scripts/gemm_timing.py): per-workgroup phase stamps
__device__ unsigned long long esr_gemm_dbg[8 * 1024];
#define ESR_GT(VAR) { __builtin_amdgcn_sched_barrier(0); VAR = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define ESR_GT(VAR)
#endif

template <int P, bool DENSE>
__global__ __launch_bounds__(kGThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void score_gemm_kernel(const __bf16* __restrict__ Ap, int64_t a_plane,
                                                                 int64_t a_rows, const __bf16* __restrict__ Bp,
                                                                 int64_t b_plane, int64_t b_rows, int Dp, int tm,
                                                                 int tn, int M, int nvalid, GemmOut o) {
  constexpr int NP = plane_count(P);              // planes per operand
  constexpr bool F16 = plane_f16(P);              // scaled fp16 planes (else bf16)
  constexpr int NS = (NP == 3) ? 2 : (NP == 2 ? 3 : 4);  // LDS stages (one 16-wide k-step each)
  constexpr int kStage = NP * kPlaneStage;        // 36864 / 24576 / 12288 B -> two workgroups per CU
  constexpr int kPieces = NP * kPiecesPerPlane;   // DMA instructions per stage, dealt round-robin to the 8 waves
  constexpr int NPW = (kPieces + 7) / 8;          // 5 / 3 / 2: waves below kLastWaves issue NPW, the others NPW - 1
  constexpr int kLastWaves = kPieces - 8 * (NPW - 1);  // 4 (P = 3, 1) or 8 (P = 2: every wave issues NPW)
  __shared__ __attribute__((aligned(16))) char lds[NS * kStage];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef ESR_GEMM_TIMING
  unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  ESR_GT(g0);
#endif

  // XCD-aware order: workgroup b runs on XCD b % 8; each XCD walks its own contiguous range of logical tiles,
  // logical tiles are ordered in groups of kGroupM tile-rows x all tile-columns, column-major inside a group.
  const int T = tm * tn, per = (T + 7) >> 3;
  const int L = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (L >= T) return;
  const int g = L / (kGroupM * tn), rem = L - g * kGroupM * tn;
  const int gm = min(kGroupM, tm - g * kGroupM);
  const int n_t = rem / gm, m_t = g * kGroupM + (rem - n_t * gm);
  const int m0 = m_t * kGM, n0 = n_t * kGN;
  const int nk = Dp / kGK;

  // ---- DMA: piece q of a stage = plane q / 12, 32-row group q % 12 of the [A tile; B tile] image.  In the
  // k-block-major planes those 32 rows x 32 B are 1 KiB of contiguous memory.  The LDS image of a group is
  // [half][row][16 B] (lane l of the DMA fetches row l % 32, half l / 32), so an MFMA fragment read -- lane =
  // (half, row) -- is lane-linear: conflict-free without a swizzle.
  const uint32_t lane_src = (uint32_t)((lane & 31) * 32 + (lane >> 5) * 16);
  const char* pbase[NPW];
  uint32_t pstep[NPW];
  int pdst[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int q = min(w + 8 * j, kPieces - 1);
    const int pl = q / kPiecesPerPlane, sub = q - pl * kPiecesPerPlane;
    const bool isA = sub < kGM / 32;
    const int64_t row0 = isA ? m0 + 32 * sub : n0 + 32 * (sub - kGM / 32);
    pbase[j] = reinterpret_cast<const char*>((isA ? Ap + pl * a_plane : Bp + pl * b_plane) + row0 * 16);
    pstep[j] = (uint32_t)((isA ? a_rows : b_rows) * 32);
    pdst[j] = pl * kPlaneStage + sub * 1024;
  }
  const bool has_last = w < kLastWaves;  // these waves own a piece in round NPW - 1
  // Pieces are issued one at a time BETWEEN MFMAs (an LDS-DMA costs the issuing wave ~60-180 cycles).  Past the
  // last stage a piece is still issued (from a valid address) into the ring slot nobody reads any more, so the
  // loop body is branch-free and vmcnt counts uniformly.
  auto issue_piece = [&](int j, int kt_target) {
    if (j == NPW - 1 && !has_last) return;
    const int ktc = min(kt_target, nk - 1);
    __builtin_amdgcn_global_load_lds((gptr_t)(pbase[j] + (uint32_t)ktc * pstep[j] + lane_src),
                                     (lptr_t)(lds + (kt_target % NS) * kStage + pdst[j]), 16, 0, 0);
  };

  // ---- fragment addressing: wave (wm, wn) owns rows wm*64.. of A and wn*64.. of B, two 32-row groups each
  const int wm = w >> 1, wn = w & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int fragA = wm * 64 * kTileRowBytes + lane * 16;
  const int fragB = (kGM + wn * 64) * kTileRowBytes + lane * 16;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
#pragma unroll
    for (int j = 0; j < NPW; ++j) issue_piece(j, s);

  constexpr int kTerms = (NP == 3) ? 6 : (NP == 2 ? 3 : 1);
  constexpr int kPieceEvery = (NP == 1) ? 2 : 4;  // a DMA piece after MFMA 1, 5, 9, ... (3, 2 planes) / 0, 2 (one plane)
  constexpr int kPieceFirst = (NP == 1) ? 0 : 1;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed once at most the NS-2 younger stages' DMAs are outstanding; after the barrier it is
    // visible to every wave and every wave is done with stage kt-1's ring slot (refilled below with kt+NS-1)
    if (has_last) wait_vmcnt_barrier<NPW * (NS - 2)>();
    else wait_vmcnt_barrier<(NPW - 1) * (NS - 2)>();
#ifdef ESR_GEMM_TIMING
    if (kt == 0) ESR_GT(g1);
#endif
    const char* st = lds + (kt % NS) * kStage;
    bf16x8 a[NP][2], b[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#if defined(ESR_PROBE_GEMM_HALF_FRAG)  /* timing probe only (results wrong): half the LDS fragment reads per MFMA -- what a
                                          128 x 64 wave tile would buy at best */
        if (r == 1) {
          a[p][1] = a[p][0];
          b[p][1] = b[p][0];
          continue;
        }
#endif
        a[p][r] = *reinterpret_cast<const bf16x8*>(st + p * kPlaneStage + fragA + r * 32 * kTileRowBytes);
        b[p][r] = *reinterpret_cast<const bf16x8*>(st + p * kPlaneStage + fragB + r * 32 * kTileRowBytes);
      }
    // six cross terms, small ones first; consecutive MFMAs go to different accumulators
#pragma unroll
    for (int term = 0; term < kTerms; ++term) {
      // P = 3: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1;  P = 2 (fp16 planes): a2 b1, a1 b2, a1 b1
      const int pa = (NP == 3) ? (term == 0 ? 2 : term == 2 || term == 3 ? 1 : 0) : (NP == 2 ? (term == 0 ? 1 : 0) : 0);
      const int pb = (NP == 3) ? (term == 1 ? 2 : term == 2 || term == 4 ? 1 : 0) : (NP == 2 ? (term == 1 ? 1 : 0) : 0);
#pragma unroll
      for (int ij = 0; ij < 4; ++ij) {
        if (F16)
          acc[ij >> 1][ij & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
              __builtin_bit_cast(f16x8, a[pa][ij >> 1]), __builtin_bit_cast(f16x8, b[pb][ij & 1]), acc[ij >> 1][ij & 1],
              0, 0, 0);
        else
          acc[ij >> 1][ij & 1] = ESR_MFMA(a[pa][ij >> 1], b[pb][ij & 1], acc[ij >> 1][ij & 1]);
        const int nth = term * 4 + ij;  // MFMA number inside this k-step
        if (nth % kPieceEvery == kPieceFirst && nth / kPieceEvery < NPW) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(nth / kPieceEvery, kt + NS - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

#ifdef ESR_GEMM_TIMING
  ESR_GT(g2);
#endif
  // ---- epilogue: acc[i][j][e] = S[m0 + wm*64 + i*32 + 8*(e/4) + 4*h + e%4][n0 + wn*64 + j*32 + l31]
  // P == 2: the accumulators carry 2^(eq + ec) x the scores.  The factor is undone where a score leaves the kernel
  // and the thresholds are scaled UP for the comparisons instead (exact either way: a power of two) -- scaling the 64
  // accumulators in place made hipcc spill 80 registers in the filtered epilogue (1.3 GB of scratch writes per launch).
  const float sscale = F16 ? ldexpf(1.f, -(o.exps[0] + o.exps[1])) : 1.f;
  const float tscale = F16 ? ldexpf(1.f, o.exps[0] + o.exps[1]) : 1.f;
  if (DENSE) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 64 + j * 32 + l31;
          if (m < M && n < nvalid) o.S[(int64_t)m * o.ldS + n] = acc[i][j][e] * sscale;
        }
      }
  } else {
    // Filtered append.  A half-wave (h) holds, for each of its 32 rows r = i*16 + e, the 64 candidates
    // (j, l31) of ONE query.  Pass 1 counts the survivors of every row with ballots and parks row r's count in
    // lane r; ONE atomic per row (all 64 of a wave in flight together) reserves the slots; pass 2 writes the
    // survivors side by side.  An atomic per (row, ballot) with its returned value needed at once serialised
    // 64 round trips per tile (measured: +45 % on the whole kernel).
    // the tile's 256 thresholds through LDS (the staging ring is idle now): the two passes below used to fetch them
    // from global memory, 32 loads per lane and pass, in front of every ballot
    float* tau_s = reinterpret_cast<float*>(lds);
    __syncthreads();
    if (t < 256) tau_s[t] = m0 + t < M ? o.tau[m0 + t] * tscale : INFINITY;
    __syncthreads();
    const float* tau_w = tau_s + wm * 64 + 4 * h;
    const bool c_ok0 = n0 + wn * 64 + l31 < nvalid, c_ok1 = n0 + wn * 64 + 32 + l31 < nvalid;
    const int mrow = m0 + wm * 64 + 4 * h;  // + i*32 + 8*(e/4) + e%4
    // Most (row pair, wave) cells hold no survivor once the thresholds have tightened (k of N candidates pass): pass 1
    // tests a row pair with two compares and one wave-wide OR, and only a cell with a survivor pays for the ballots and
    // counts; `rows` remembers those cells (bit i*16 + e), so pass 2 revisits them alone.
    int my_cnt = 0;
    uint32_t rows = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float tau_r[16];  // 16 thresholds at a time: the kernel must stay within 128 VGPRs (two workgroups per CU)
#pragma unroll
      for (int e = 0; e < 16; ++e) tau_r[e] = tau_w[i * 32 + 8 * (e >> 2) + (e & 3)];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bool p0 = c_ok0 && acc[i][0][e] >= tau_r[e];
        const bool p1 = c_ok1 && acc[i][1][e] >= tau_r[e];
        if (__ballot(p0 || p1) == 0) continue;  // wave-uniform
        rows |= 1u << (i * 16 + e);
        const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1);
        const int c = __popc((uint32_t)(b0 >> (32 * h))) + __popc((uint32_t)(b1 >> (32 * h)));
        if (l31 == i * 16 + e) my_cnt = c;
      }
    }
    rows = __builtin_amdgcn_readfirstlane(rows);
    if (rows) {
      const int my_m = mrow + (l31 >> 4) * 32 + 8 * ((l31 & 15) >> 2) + (l31 & 3);
      int my_slot = 0;
      if (my_cnt > 0) my_slot = atomicAdd(o.cnt + my_m, my_cnt);
      const uint32_t below = (1u << l31) - 1u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (!((rows >> (i * 16 + e)) & 1u)) continue;  // wave-uniform
          const float tau = tau_w[i * 32 + 8 * (e >> 2) + (e & 3)];
          const bool p0 = c_ok0 && acc[i][0][e] >= tau;
          const bool p1 = c_ok1 && acc[i][1][e] >= tau;
          const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1);
          const uint32_t h0 = (uint32_t)(b0 >> (32 * h)), h1 = (uint32_t)(b1 >> (32 * h));
          const int slot = __shfl(my_slot, 32 * h + i * 16 + e, 64);
          const int m = mrow + i * 32 + 8 * (e >> 2) + (e & 3);
          int2* dst = o.pairs + (int64_t)m * o.ppitch + slot;
          const int n = n0 + wn * 64 + l31;
          if (p0) dst[__popc(h0 & below)] = make_int2(__float_as_int(acc[i][0][e] * sscale), o.gbase + n * o.gstep);
          if (p1)
            dst[__popc(h0) + __popc(h1 & below)] =
                make_int2(__float_as_int(acc[i][1][e] * sscale), o.gbase + (n + 32) * o.gstep);
        }
      }
    }
  }
#ifdef ESR_GEMM_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  ESR_GT(g3);
  if (t == 0 && blockIdx.x < 1024) {
    unsigned long long* d = esr_gemm_dbg + blockIdx.x * 8;
    d[0] = g1 - g0; d[1] = g2 - g1; d[2] = g3 - g2; d[3] = __builtin_amdgcn_s_memrealtime() - r0; d[4] = nk;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// per-query top-k select
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t score_key(float v) {  // larger float <=> larger key; -0 == +0
  v += 0.0f;
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct SelIn {
  const float* vals;   // element c of row r at vals[r * vpitch + c * stride]
  int64_t vpitch;
  const int32_t* idx;  // explicit index (same pitch / stride, in int32 elements) or null
  int stride;
  int32_t ibase, istep;  // implicit index = ibase + c * istep
  const int32_t* n_per_row;  // list length per row, or null -> n_fixed
  int n_fixed;
  int skip_upto = 0;   // > 0: a row whose list holds at most this many records is left as it is (lazy compaction: the
                       // caller's list has room for it to grow by another chunk; its tau stays the older, weaker bound)
  // BAND mode (round 6, mode 3's one-term filter; band_qnorm != null): the scores are one-plane scores whose error is
  // bounded by b = band_coef |q| max|c|.  The true top-k lies among the records with score >= (k-th best score) - 2 b
  // (see esr_retrieve_topk): the call keeps ALL of those in out.pairs (in place when the source is that list), sets
  // out.cnt to their number and out.tau to that threshold; nothing is sorted or delivered.
  // A row whose band holds more than band_cap records cannot stay in band mode (its list must always have room for a
  // whole chunk): it becomes an EXACT row -- band_nexact[row] = -1 here, then (the caller's next two launches) all its
  // records are re-scored in f32 and the plain select cuts the list to its k best; from then on band_nexact[row] = the
  // number of leading records that carry exact scores, the caller re-scores what a chunk appended before every select,
  // and the row's threshold is (exact k-th best) - b.  only_pending: the launch handles rows marked -1 alone.
  const float* band_qnorm = nullptr;  // [rows] |q|
  const float* band_cmax = nullptr;   // [1] the largest candidate norm
  float band_coef = 0.f;
  int32_t* band_nexact = nullptr;     // [rows] 0 = band row, -1 = turns exact now, > 0 = exact row
  int band_cap = 0;
  int only_pending = 0;
  // what the exact rows' re-score needs (done by the row's workgroup itself, in front of its select: rescore_rows)
  const float* rs_Q = nullptr;
  const float* rs_C = nullptr;
  int rs_D = 0;
  int32_t rs_base = 0, rs_step = 1;
};
struct SelOut {
  int2* pairs;         // running top-k list head [rows][ppitch], or null
  int64_t ppitch;
  int32_t* cnt;        // list length after the call = min(n, k)
  float* tau;          // k-th best score (or -inf while fewer than k)
  float* scores;       // [rows][k] or null
  int32_t* indices;    // [rows][k] or null
};

// scores of records [jfirst, n) of one query's (score, index) list := f32 dot products with the query row, by the
// query's workgroup (256 threads: sixteen lanes per candidate row, 16-byte loads); qrow: D floats of LDS
__device__ __forceinline__ void rescore_rows(const float* __restrict__ Qrow, const float* __restrict__ C, int D,
                                             int2* __restrict__ list, int jfirst, int n, int32_t base, int32_t step,
                                             float* qrow) {
  for (int d = threadIdx.x; d < D; d += kBlock) qrow[d] = Qrow[d];
  __syncthreads();
  const int lig = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const bool vec = (D & 3) == 0 && ((uintptr_t)C & 15) == 0;
  for (int j0 = jfirst; j0 < n; j0 += kBlock / 16) {
    const int j = j0 + grp;
    float acc = 0.f;
    if (j < n) {
      const int64_t r = ((int64_t)list[j].y - base) / step;
      const float* ca = C + r * D;
      if (vec) {
        for (int d = lig * 4; d < D; d += 64) {
          const float4 y = *reinterpret_cast<const float4*>(ca + d);
          const float4 x = *reinterpret_cast<const float4*>(qrow + d);
          acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
      } else {
        for (int d = lig; d < D; d += 16) acc = fmaf(qrow[d], ca[d], acc);
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    if (j < n && lig == 0) list[j].x = __float_as_int(acc);
  }
  __syncthreads();  // (the list is read back by the whole workgroup; qrow's LDS is reused)
}

// inclusive scan over the 256 threads of the block (wave shuffles + 4 partials)
__device__ __forceinline__ int block_incl_scan(int v, int* part4, int t) {
  const int lane = t & 63, w = t >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) part4[w] = x;
  __syncthreads();
  int add = 0;
  for (int j = 0; j < w; ++j) add += part4[j];
  __syncthreads();
  return x + add;
}

constexpr int kSelBatch = 8;   // elements per thread of a row that stays in registers (rows <= 2048)
constexpr int kSelStream = 4;  // independent loads in flight per thread when a longer row is streamed
constexpr int kSelLdsWords = 8192;  // 32 KB of dynamic LDS for the row cache of the launches that have long rows

// One workgroup per query.  n <= k: everything is kept.  Otherwise an MSB-first radix select runs on
// D = composite - min(composite), starting at the highest bit in which the row's composites differ: the lists a
// merge sees are the survivors of a threshold, i.e. crowded into a sliver of the float range, and digits taken from
// fixed bit positions put them all into two or three bins (measured: same-address LDS atomics made a 2 500-element
// merge take 180 us).  Range-relative digits spread them over the 2048 bins, so one histogram pass usually isolates
// the bucket of the k-th largest; passes continue (11 bits each) until that bucket is needed whole -- at the latest
// when all bits are used, composites being unique -- which yields exactly the top-k SET.  It is compacted into
// LDS; only a call that delivers the answer (out.scores) sorts it (bitonic, 64-bit keys): the running list of an
// intermediate merge need not be ordered, only tau must be exact.
__global__ __launch_bounds__(kSelThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void topk_select_kernel(
    SelIn in, int k, SelOut out, int lds_words) {
  // lds_words 32-bit words of dynamic LDS (0 or kSelLdsWords): a row that does not fit the registers (> 2048 elements)
  // but fits here -- 8192 dense scores as 32-bit keys, or 4096 (score, index) records as 64-bit composites -- is
  // read from global memory once; every later sweep of the radix select (5 - 6 of them) reads LDS.  Without it the
  // select of the dense first chunk re-read 268 MB per sweep.
  extern __shared__ __attribute__((aligned(16))) uint32_t lrow[];
  __shared__ unsigned long long sel[kSelMaxK];
  __shared__ int hist[2048];
  __shared__ int part4[4];
  __shared__ int s_digit, s_above, s_cnt, s_nsel;
  __shared__ unsigned long long s_min, s_lo, s_hi;
  const int row = blockIdx.x, t = threadIdx.x;
  const int n = in.n_per_row ? in.n_per_row[row] : in.n_fixed;
  if (in.skip_upto > 0 && n <= in.skip_upto) return;  // (uniform over the workgroup)
  const bool final = out.scores != nullptr;
  const int nex = in.band_nexact ? in.band_nexact[row] : 0;
  if (in.only_pending && nex != -1) return;
  const bool band = in.band_qnorm != nullptr && nex == 0 && !final;  // (exact rows take the plain path below)
  const float band_b = in.band_qnorm ? in.band_coef * in.band_qnorm[row] * in.band_cmax[0] : 0.f;
  const bool src_is_list = in.stride == 2 && reinterpret_cast<const int2*>(in.vals) == out.pairs;
  if (in.rs_Q && src_is_list && nex != 0 && !final) {
    // an exact row of mode 3: what the chunk appended behind its nex exact records (nex = -1: the row turns exact now,
    // every record) gets its f32 score before the select looks at the list
    const int jfirst = nex > 0 ? min(nex, n) : 0;
    if (jfirst < n)
      rescore_rows(in.rs_Q + (int64_t)row * in.rs_D, in.rs_C, in.rs_D, out.pairs + (int64_t)row * out.ppitch, jfirst, n,
                   in.rs_base, in.rs_step, reinterpret_cast<float*>(lrow));
  }
  if (n <= k && !final && (!band || src_is_list)) {  // nothing was appended (or the list is still short): the running state stands
    if (t == 0) {
      if (out.cnt) out.cnt[row] = n;
      if (out.tau && n < k) out.tau[row] = -INFINITY;
      if (in.band_nexact && nex != 0) in.band_nexact[row] = n;
    }
    return;
  }
  const float* v = in.vals + (int64_t)row * in.vpitch;
  const int32_t* ix = in.idx ? in.idx + (int64_t)row * in.vpitch : nullptr;
  const bool paired = in.stride == 2 && ix == reinterpret_cast<const int32_t*>(v) + 1;  // (score, index) records
  auto comp = [&](int c) -> unsigned long long {
    uint32_t key, gi;
    if (paired) {
      const int2 pr = *reinterpret_cast<const int2*>(v + 2 * (int64_t)c);
      key = score_key(__int_as_float(pr.x));
      gi = (uint32_t)pr.y;
    } else {
      key = score_key(v[(int64_t)c * in.stride]);
      gi = (uint32_t)(ix ? ix[(int64_t)c * in.stride] : in.ibase + c * in.istep);
    }
    return ((unsigned long long)key << 32) | (0xFFFFFFFFu - gi);
  };
  // a row of up to 2048 elements is read once and stays in registers; longer rows are re-read (L2) every sweep
  const bool cached = n <= kSelThreads * kSelBatch;
  const bool dense = !paired && ix == nullptr && in.stride == 1;
  const bool lds_row = !cached && ((dense && n <= lds_words) || (!dense && 2 * n <= lds_words));
  // the fill also yields the row's value range (the first sweep of the select otherwise): bounds, not necessarily
  // attained -- the range-relative digits only need lo <= every composite <= hi
  unsigned long long fill_lo = ~0ull, fill_hi = 0ull;
  if (lds_row) {
    if (dense) {
      uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
      // 16 bytes per thread and load where the row allows it (a rolled loop of dword loads waited one memory latency
      // per 256 elements, 32 in a row for the dense first chunk)
      const bool vec = (n & 3) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0;
      if (vec) {
        for (int c4 = t; c4 < n / 4; c4 += kSelThreads) {
          const float4 x = reinterpret_cast<const float4*>(v)[c4];
          const uint32_t k0 = score_key(x.x), k1 = score_key(x.y), k2 = score_key(x.z), k3 = score_key(x.w);
          *reinterpret_cast<uint4*>(lrow + 4 * c4) = make_uint4(k0, k1, k2, k3);
          const uint32_t a = k0 < k1 ? k0 : k1, b = k2 < k3 ? k2 : k3, c = k0 > k1 ? k0 : k1, d = k2 > k3 ? k2 : k3;
          const uint32_t mn = a < b ? a : b, mx = c > d ? c : d;
          kmin = mn < kmin ? mn : kmin;
          kmax = mx > kmax ? mx : kmax;
        }
      } else {
        for (int c = t; c < n; c += kSelThreads) {
          const uint32_t key = score_key(v[c]);
          lrow[c] = key;
          kmin = key < kmin ? key : kmin;
          kmax = key > kmax ? key : kmax;
        }
      }
      fill_lo = (unsigned long long)kmin << 32;
      fill_hi = ((unsigned long long)kmax << 32) | 0xFFFFFFFFull;
    } else {
      unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lrow);
      for (int c0 = 0; c0 < n; c0 += kSelThreads * kSelStream) {
        unsigned long long C[kSelStream];
#pragma unroll
        for (int u = 0; u < kSelStream; ++u) {
          const int c = c0 + u * kSelThreads + t;
          C[u] = c < n ? comp(c) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kSelStream; ++u) {
          const int c = c0 + u * kSelThreads + t;
          if (c < n) {
            l64[c] = C[u];
            fill_lo = C[u] < fill_lo ? C[u] : fill_lo;
            fill_hi = C[u] > fill_hi ? C[u] : fill_hi;
          }
        }
      }
    }
    __syncthreads();
  }
  unsigned long long R[kSelBatch];
  if (cached) {
#pragma unroll
    for (int u = 0; u < kSelBatch; ++u) {
      const int c = u * kSelThreads + t;
      R[u] = c < n ? comp(c) : 0ull;
    }
  }
  auto sweep = [&](auto&& fn) {
    if (cached) {
#pragma unroll
      for (int u = 0; u < kSelBatch; ++u) fn(R[u], u * kSelThreads + t < n);
    } else if (lds_row) {
      const unsigned long long* l64 = reinterpret_cast<const unsigned long long*>(lrow);
      for (int c = t; c < n + (kSelThreads - 1 - (n - 1) % kSelThreads); c += kSelThreads) {  // whole waves: fn ballots
        const bool valid = c < n;
        unsigned long long C = 0ull;
        if (valid)
          C = dense ? (((unsigned long long)lrow[c] << 32) | (0xFFFFFFFFu - (uint32_t)(in.ibase + c * in.istep)))
                    : l64[c];
        fn(C, valid);
      }
    } else {
      for (int c0 = 0; c0 < n; c0 += kSelThreads * kSelStream) {
        unsigned long long C[kSelStream];
#pragma unroll
        for (int u = 0; u < kSelStream; ++u) {
          const int c = c0 + u * kSelThreads + t;
          C[u] = c < n ? comp(c) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kSelStream; ++u) fn(C[u], c0 + u * kSelThreads + t < n);
      }
    }
  };
  const int nsel = min(n, k);
  if (t == 0) {
    s_nsel = 0;
    s_min = ~0ull;
    s_lo = ~0ull;
    s_hi = 0ull;
  }
  __syncthreads();
  unsigned long long thr = 0;
  if (n > k) {
    unsigned long long lo = fill_lo, hi = fill_hi;
    if (!lds_row)
      sweep([&](unsigned long long C, bool valid) {
        lo = (valid && C < lo) ? C : lo;
        hi = (valid && C > hi) ? C : hi;
      });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a = __shfl_xor(lo, o, 64), b2 = __shfl_xor(hi, o, 64);
      lo = a < lo ? a : lo;
      hi = b2 > hi ? b2 : hi;
    }
    if ((t & 63) == 0) {
      atomicMin(&s_lo, lo);
      atomicMax(&s_hi, hi);
    }
    __syncthreads();
    lo = s_lo;
    const int L = 64 - __clzll((long long)(s_hi - lo));  // >= 1: n > k >= 1 distinct composites
    unsigned long long prefix = 0;
    int bits_done = 0, need = k;
    while (bits_done < L) {
      const int wbits = min(11, L - bits_done);
      const int shift = L - bits_done - wbits;
      for (int b = t; b < 2048; b += kSelThreads) hist[b] = 0;
      __syncthreads();
      if (lds_row && dense && shift >= 32) {
        // a digit that lies entirely in the score key (the upper word of the composite; lo's lower word is zero, so
        // nothing borrows): 32-bit arithmetic straight on the staged keys, ~6 instructions per element instead of ~20
        const uint32_t lok = (uint32_t)(lo >> 32), pfx = (uint32_t)prefix, msk = (1u << wbits) - 1u;
        const int sh = shift - 32, psh = L - bits_done - 32;
        for (int c0 = 0; c0 < n; c0 += kSelThreads * 4) {
          uint32_t d[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * kSelThreads + t;
            d[u] = c < n ? lrow[c] - lok : 0u;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * kSelThreads + t;
            if (c < n && (bits_done == 0 || (d[u] >> psh) == pfx)) atomicAdd(&hist[(int)((d[u] >> sh) & msk)], 1);
          }
        }
      } else {
        sweep([&](unsigned long long C, bool valid) {
          const unsigned long long Dv = C - lo;
          if (valid && (bits_done == 0 || (Dv >> (L - bits_done)) == prefix))
            atomicAdd(&hist[(int)((Dv >> shift) & ((1u << wbits) - 1))], 1);
        });
      }
      __syncthreads();
      // thread t owns bins 2047-8t .. 2040-8t (descending)
      int local = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) local += hist[2047 - 8 * t - j];
      const int incl = block_incl_scan(local, part4, t), excl = incl - local;
      if (excl < need && need <= incl) {
        int run = excl;
        for (int j = 0; j < 8; ++j) {
          const int hv = hist[2047 - 8 * t - j];
          if (run + hv >= need) {
            s_digit = 2047 - 8 * t - j;
            s_above = run;
            s_cnt = hv;
            break;
          }
          run += hv;
        }
      }
      __syncthreads();
      prefix = (prefix << wbits) | (unsigned long long)s_digit;
      need -= s_above;
      bits_done += wbits;
      const bool whole_bucket = (s_cnt == need);
      __syncthreads();  // s_* consumed before the next pass overwrites them
      if (whole_bucket) break;
    }
    thr = lo + (prefix << (L - bits_done));
  }
  if (band) {
    // the k-th best score = the smallest composite >= thr; then every record within the band of it stays, in list order.
    // In place (source == out.pairs) this is safe block by block: a block's records are all read before any of them is
    // written, and they are written to positions not beyond the block's own start + its size.
    unsigned long long mn = ~0ull;
    if (n > k) {
      sweep([&](unsigned long long C, bool valid) { mn = (valid && C >= thr && C < mn) ? C : mn; });
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(mn, o, 64);
        mn = a < mn ? a : mn;
      }
      if ((t & 63) == 0) atomicMin(&s_min, mn);
    }
    __syncthreads();
    const float kth = n > k ? key_score((uint32_t)(s_min >> 32)) : -INFINITY;
    const float keep_thr = n > k ? kth - 2.0f * band_b : -INFINITY;
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += kSelThreads) {
      const int c = c0 + t;
      const bool valid = c < n;
      const unsigned long long C = valid ? comp(c) : 0ull;
      const float sc = key_score((uint32_t)(C >> 32));
      const bool pass = valid && sc >= keep_thr;
      const unsigned long long mask = __ballot(pass);
      if ((t & 63) == 0) part4[t >> 6] = __popcll(mask);
      __syncthreads();  // (every record of the block has been read)
      int off = base;
      for (int j = 0; j < (t >> 6); ++j) off += part4[j];
      const int total = part4[0] + part4[1] + part4[2] + part4[3];
      if (pass)
        out.pairs[(int64_t)row * out.ppitch + off + __popcll(mask & ((1ull << (t & 63)) - 1ull))] =
            make_int2(__float_as_int(sc), (int32_t)(0xFFFFFFFFu - (uint32_t)C));
      base += total;
      __syncthreads();  // (part4 is reused by the next block)
    }
    if (t == 0) {
      out.cnt[row] = base;
      out.tau[row] = keep_thr;
      if (base > in.band_cap) in.band_nexact[row] = -1;  // too wide a band for the list: the row turns exact
    }
    return;
  }
  // compact the composites >= thr (exactly nsel of them) into LDS: one slot reservation per wave instruction
  // (a same-address LDS atomic per element serialises: 500 survivors cost more than the whole radix select)
  unsigned long long my_min = ~0ull;
  auto keep = [&](unsigned long long C, bool pass) {  // called by whole waves
    const unsigned long long mask = __ballot(pass);
    if (mask) {
      const int lane = t & 63;
      int base = 0;
      if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(&s_nsel, __popcll(mask));
      base = __shfl(base, __ffsll((long long)mask) - 1, 64);
      if (pass) {
        const int p = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (p < kSelMaxK) sel[p] = C;
        my_min = C < my_min ? C : my_min;
      }
    }
  };
  sweep([&](unsigned long long C, bool valid) { keep(C, valid && C >= thr); });
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_xor(my_min, o, 64);
    my_min = a < my_min ? a : my_min;
  }
  if ((t & 63) == 0) atomicMin(&s_min, my_min);
  if (final) {
    int np2 = 1;
    while (np2 < nsel) np2 <<= 1;
    __syncthreads();
    for (int c = nsel + t; c < np2; c += kSelThreads) sel[c] = 0ull;
    for (int size = 2; size <= np2; size <<= 1) {  // bitonic sort, descending
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        __syncthreads();
        for (int i = t; i < (np2 >> 1); i += kSelThreads) {
          const int pos = 2 * i - (i & (stride - 1));
          const int j = pos + stride;
          const bool desc = (pos & size) == 0;
          const unsigned long long x = sel[pos], y = sel[j];
          if ((x < y) == desc) {
            sel[pos] = y;
            sel[j] = x;
          }
        }
      }
    }
  }
  __syncthreads();  // every read of this row's list is done: its head may be overwritten
  for (int c = t; c < k; c += kSelThreads) {
    const bool valid = c < nsel;
    const unsigned long long C = valid ? sel[c] : 0ull;
    const float s = valid ? key_score((uint32_t)(C >> 32)) : -INFINITY;
    const int32_t gi = valid ? (int32_t)(0xFFFFFFFFu - (uint32_t)C) : -1;
    if (out.pairs && valid) out.pairs[(int64_t)row * out.ppitch + c] = make_int2(__float_as_int(s), gi);
    if (out.scores) out.scores[(int64_t)row * k + c] = s;
    if (out.indices) out.indices[(int64_t)row * k + c] = gi;
  }
  if (t == 0) {
    if (in.band_nexact && nex != 0 && !final) in.band_nexact[row] = nsel;  // an exact row of mode 3: its k best, all exact
    if (out.cnt) out.cnt[row] = nsel;
    if (out.tau) out.tau[row] = (n >= k) ? key_score((uint32_t)(s_min >> 32)) - band_b : -INFINITY;  // (band_b: exact rows of mode 3)
  }
}

// ---------------------------------------------------------------------------------------------------
// exact f32 re-score of candidate lists (the second stage of the bf16 "ANN" path)
// scores[q, j] = queries[q] . candidates[(idx[q, j] - base) / step], one wave per (q, j)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void rescore_kernel(const float* __restrict__ Q, const float* __restrict__ C,
                                                        int64_t N, int D, const int32_t* __restrict__ idx,
                                                        int64_t total, int kc, int32_t base, int32_t step,
                                                        float* __restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t p = wave; p < total; p += nwaves) {
    const int64_t q = p / kc;
    const int32_t gi = idx[p];
    float acc = 0.f;
    if (gi >= 0) {
      const int64_t r = ((int64_t)gi - base) / step;
      const float* qa = Q + q * D;
      const float* ca = C + r * D;
      if ((D & 3) == 0) {
        for (int d = lane * 4; d < D; d += 256) {
          const float4 x = *reinterpret_cast<const float4*>(qa + d);
          const float4 y = *reinterpret_cast<const float4*>(ca + d);
          acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
      } else {
        for (int d = lane; d < D; d += 64) acc = fmaf(qa[d], ca[d], acc);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) scores[p] = gi >= 0 ? acc : -INFINITY;
  }
}

// the same over ragged per-query lists of (score, index) records (mode 3, after the last chunk: the survivors of the
// one-term filter): the records' scores are REPLACED by the f32 dot products.  One workgroup per query; rows that turned
// exact earlier (nexact != 0) are exact already.
__global__ __launch_bounds__(kBlock) void rescore_lists_kernel(const float* __restrict__ Q, const float* __restrict__ C,
                                                              int D, int2* __restrict__ pairs, int64_t ppitch,
                                                              const int32_t* __restrict__ cnt, int32_t base,
                                                              int32_t step, const int32_t* __restrict__ nexact) {
  extern __shared__ __attribute__((aligned(16))) float qrow[];  // D floats
  const int64_t q = blockIdx.x;
  if (nexact[q] != 0) return;
  rescore_rows(Q + q * D, C, D, pairs + q * ppitch, 0, cnt[q], base, step, qrow);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct RetrievePlan {
  int P, Dp;
  int64_t Mp, chunk, chunk_pad, first;
  size_t off_A, off_B, off_S, off_pairs, off_cnt, off_tau, off_abs, off_band, off_nexact, total;
  int64_t ppitch;
  int skip_upto;  // running lists up to this long are not compacted between chunks
};

static RetrievePlan retrieve_plan(int64_t nq, int64_t N, int D, int k, int mode) {
  RetrievePlan p;
  p.P = (mode == 0) ? 3 : (mode == 2 ? 2 : (mode == 3 ? 4 : 1));
  p.Dp = (int)(cdiv(D, kGK) * kGK);
  p.Mp = cdiv(nq, kGM) * kGM;
  p.first = std::min<int64_t>(N, std::max<int64_t>(kFirstChunk, 16 * (int64_t)k));
  // later chunks: the per-query append list must hold a whole chunk (worst case every candidate passes tau);
  // keep the list buffer around 2 GiB and the 32-bit DMA offsets inside one plane set
  // (mode 3's one-plane GEMM is short enough for the per-chunk ends -- a GEMM tail, a split and two select launches --
  // to show: with 65 536-row chunks (a 4.4 GB list buffer at 8192 queries) the call is 14.22 against 14.47 ms; 131 072
  // is slower again, the filter's threshold being a chunk old: 14.85.  The three-term modes lose with larger chunks.
  // ESR_RETRIEVE_LIST_LOG2 / ESR_RETRIEVE_CHUNK_CAP: measuring hooks.)
  const char* lb = getenv("ESR_RETRIEVE_LIST_LOG2");
  const int list_log2 = lb ? std::min(31, std::max(20, atoi(lb))) : (mode == 3 ? 29 : 28);
  int64_t chunk = ((int64_t)1 << list_log2) / std::max<int64_t>(nq, 1);
  chunk = std::min<int64_t>(chunk, ((int64_t)1 << 30) / ((int64_t)p.Dp * 2 * plane_count(p.P)));
  const char* cc = getenv("ESR_RETRIEVE_CHUNK_CAP");  // (measuring hook)
  const int64_t chunk_cap = cc ? std::max<int64_t>(kGN, atoll(cc)) : 65536;
  chunk = std::max<int64_t>(kGN, std::min<int64_t>(chunk_cap, chunk / kGN * kGN));
  p.chunk = chunk;
  p.chunk_pad = std::max(cdiv(p.chunk, kGN) * kGN, cdiv(p.first, kGN) * kGN);
  // Lazy compaction (round 4): between chunks a query's list is selected down to its k best -- and its tau raised --
  // only once it holds more than skip_upto records; shorter lists just grow (a stale tau is still a lower bound of the
  // final k-th score: nothing of the answer is dropped, a few more records are appended).  With tau from the first 8192
  // candidates the second chunk fills a list once; after that a list passes the mark every few chunks: ~6 real selects
  // per query at N = 1 M instead of 33 (82 us each).  The list must hold the mark plus a whole chunk.
  p.skip_upto = (int)std::max<int64_t>(3 * (int64_t)k, 1536);
  p.ppitch = std::max<int64_t>(k, p.skip_upto) + p.chunk;
  size_t o = 0;
  p.off_A = o; o += align_up((size_t)plane_count(p.P) * p.Mp * p.Dp * 2, 256);
  p.off_B = o; o += align_up((size_t)plane_count(p.P) * p.chunk_pad * p.Dp * 2, 256);
  p.off_S = o; o += align_up((size_t)nq * p.first * 4, 256);
  p.off_pairs = o; o += align_up((size_t)nq * p.ppitch * 8, 256);
  p.off_cnt = o; o += align_up((size_t)nq * 4, 256);
  p.off_tau = o; o += align_up((size_t)nq * 4, 256);
  p.off_abs = o; o += align_up((size_t)(kAbsBlocks + 64) * 4, 256);  // absmax slots, then the two exponent words
  p.off_band = o; o += align_up((size_t)(kAbsBlocks + 64 + nq) * 4, 256);  // mode 3: norm slots, max |c|, then |q| per query
  p.off_nexact = o; o += align_up((size_t)nq * 4, 256);                    //         exact-row marks
  p.total = o;
  return p;
}

// top-k of dense score rows [rows][pitch] (index = column), descending, ties -> lower index: the radix select with the
// final bitonic sort of the k survivors.  Used by esr_score_topk (esr_sort.hip) instead of a full sort of every row.
static_assert(kSelectMaxK == kSelMaxK, "esr_common.h advertises the select kernel's k limit");
int select_topk_dense(const float* scores, int64_t pitch, int64_t rows, int n, int k, float* out_scores,
                      int32_t* out_indices, hipStream_t st) {
  if (k > kSelMaxK) {
    set_error("top-k select: k=%d exceeds %d", k, kSelMaxK);
    return ESR_EINVAL;
  }
  SelIn in;
  in.vals = scores; in.vpitch = pitch; in.idx = nullptr; in.stride = 1; in.ibase = 0; in.istep = 1;
  in.n_per_row = nullptr; in.n_fixed = n;
  SelOut so;
  so.pairs = nullptr; so.ppitch = 0; so.cnt = nullptr; so.tau = nullptr; so.scores = out_scores; so.indices = out_indices;
  const int lds_words = n > kSelThreads * kSelBatch && n <= kSelLdsWords ? kSelLdsWords : 0;
  hipLaunchKernelGGL(topk_select_kernel, dim3((int)rows), dim3(kSelThreads), lds_words * sizeof(uint32_t), st, in, k, so,
                     lds_words);
  return check_launch("select_topk_dense");
}

int select_topk_head(const float* scores, int64_t pitch, int64_t rows, int n, int k, int2* pairs, int64_t ppitch,
                     int32_t* cnt, float* tau, hipStream_t st) {
  if (k > kSelMaxK || n <= k) {
    set_error("top-k select: k=%d must be below the row length %d and at most %d", k, n, kSelMaxK);
    return ESR_EINVAL;
  }
  SelIn in;
  in.vals = scores; in.vpitch = pitch; in.idx = nullptr; in.stride = 1; in.ibase = 0; in.istep = 1;
  in.n_per_row = nullptr; in.n_fixed = n;
  SelOut so;
  so.pairs = pairs; so.ppitch = ppitch; so.cnt = cnt; so.tau = tau; so.scores = nullptr; so.indices = nullptr;
  const int lds_words = n > kSelThreads * kSelBatch && n <= kSelLdsWords ? kSelLdsWords : 0;
  hipLaunchKernelGGL(topk_select_kernel, dim3((int)rows), dim3(kSelThreads), lds_words * sizeof(uint32_t), st, in, k, so,
                     lds_words);
  return check_launch("select_topk_head");
}

int select_topk_tail(const int2* pairs, int64_t ppitch, const int32_t* cnt, int64_t rows, int k, float* out_scores,
                     int32_t* out_indices, hipStream_t st) {
  if (k > kSelMaxK) {
    set_error("top-k select: k=%d exceeds %d", k, kSelMaxK);
    return ESR_EINVAL;
  }
  SelIn in;
  in.vals = (const float*)pairs; in.vpitch = 2 * ppitch; in.idx = (const int32_t*)pairs + 1; in.stride = 2;
  in.ibase = 0; in.istep = 0; in.n_per_row = cnt; in.n_fixed = 0;
  SelOut so;
  so.pairs = nullptr; so.ppitch = 0; so.cnt = nullptr; so.tau = nullptr; so.scores = out_scores; so.indices = out_indices;
  // (lists longer than the registers hold -- 2048 records -- are cached in LDS up to 4096, streamed beyond)
  hipLaunchKernelGGL(topk_select_kernel, dim3((int)rows), dim3(kSelThreads), kSelLdsWords * sizeof(uint32_t), st, in, k, so,
                     kSelLdsWords);
  return check_launch("select_topk_tail");
}

int select_topk_compact(int2* pairs, int64_t ppitch, int32_t* cnt, int64_t rows, int k, float* tau, hipStream_t st,
                        int skip_upto) {
  if (k > kSelMaxK) {
    set_error("top-k select: k=%d exceeds %d", k, kSelMaxK);
    return ESR_EINVAL;
  }
  SelIn in;
  in.vals = (const float*)pairs; in.vpitch = 2 * ppitch; in.idx = (const int32_t*)pairs + 1; in.stride = 2;
  in.ibase = 0; in.istep = 0; in.n_per_row = cnt; in.n_fixed = 0;
  in.skip_upto = skip_upto;  // lazy compaction: shorter lists are left to grow (their tau stays a valid lower bound)
  SelOut so;
  so.pairs = pairs; so.ppitch = ppitch; so.cnt = cnt; so.tau = tau; so.scores = nullptr; so.indices = nullptr;
  hipLaunchKernelGGL(topk_select_kernel, dim3((int)rows), dim3(kSelThreads), kSelLdsWords * sizeof(uint32_t), st, in, k, so,
                     kSelLdsWords);
  return check_launch("select_topk_compact");
}

// top-k of ragged score rows: row r has n_per_row[r] valid entries at scores[r * pitch + c] (index = column).  Rows
// shorter than k deliver all they have (the caller pre-fills the outputs).  For esr_ivf.hip.
int select_topk_ragged(const float* scores, int64_t pitch, int64_t rows, const int32_t* n_per_row, int max_n, int k,
                       float* out_scores, int32_t* out_indices, hipStream_t st) {
  if (k > kSelMaxK) {
    set_error("top-k select: k=%d exceeds %d", k, kSelMaxK);
    return ESR_EINVAL;
  }
  SelIn in;
  in.vals = scores; in.vpitch = pitch; in.idx = nullptr; in.stride = 1; in.ibase = 0; in.istep = 1;
  in.n_per_row = n_per_row; in.n_fixed = 0;
  SelOut so;
  so.pairs = nullptr; so.ppitch = 0; so.cnt = nullptr; so.tau = nullptr; so.scores = out_scores; so.indices = out_indices;
  const int lds_words = max_n > kSelThreads * kSelBatch && max_n <= kSelLdsWords ? kSelLdsWords : 0;
  hipLaunchKernelGGL(topk_select_kernel, dim3((int)rows), dim3(kSelThreads), lds_words * sizeof(uint32_t), st, in, k, so,
                     lds_words);
  return check_launch("select_topk_ragged");
}


template <int P>
static void launch_split(const float* X, int64_t n_rows, int D, int64_t rows_pad, int Dp, int64_t plane_elems,
                         __bf16* out, hipStream_t st, const int* exp_ptr = nullptr) {
  const int64_t total = rows_pad * (Dp >> 2);
  const int grid = (int)std::min<int64_t>(cdiv(total, kBlock), 8192);
  hipLaunchKernelGGL((split_planes_kernel<P>), dim3(grid), dim3(kBlock), 0, st, X, n_rows, D, rows_pad, Dp, plane_elems,
                     out, exp_ptr);
}
// exponent word of matrix X (n elements) -> exp_out; `slots` holds kAbsBlocks floats
static void launch_absmax(const float* X, int64_t n, float* slots, int* exp_out, hipStream_t st) {
  const int grid = (int)std::min<int64_t>(kAbsBlocks, std::max<int64_t>(1, cdiv(n, (int64_t)kBlock * 16)));
  hipLaunchKernelGGL(absmax_part_kernel, dim3(grid), dim3(kBlock), 0, st, X, n, slots);
  hipLaunchKernelGGL(absmax_exp_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)slots, grid, exp_out);
}

template <int P, bool DENSE>
static void launch_gemm(const __bf16* A, int64_t a_plane, const __bf16* B, int64_t b_plane, int Dp, int64_t Mp,
                        int64_t n_pad, int M, int nvalid, const GemmOut& o, hipStream_t st, int64_t b_rows = 0) {
  // planes are k-block-major over Mp query rows / b_rows candidate rows (default: this chunk's padded row count n_pad;
  // a prepared corpus: ALL its rows, B pointing at the chunk's first row -- esr_retrieve_prepare)
  const int tm = (int)(Mp / kGM), tn = (int)(n_pad / kGN);
  const int per = (tm * tn + 7) / 8;
  ESR_KT("score_gemm_kernel", st, hipLaunchKernelGGL((score_gemm_kernel<P, DENSE>), dim3(per * 8), dim3(kGThreads), 0, st, A, a_plane, Mp, B, b_plane,
                     b_rows ? b_rows : n_pad, Dp, tm, tn, M, nvalid, o));
}


// recall of an approximate top-k against the exact one, O(ka + ke) per query: one workgroup per query puts the approximate
// ids into an open-addressing table in LDS and probes it with the exact ids; the hits of all queries are added to ONE
// 64-bit word (integer: order-free).  (The torch form -- an [Q, ka, ke] boolean cube -- was 2 GB at 8192 x 500 x 500.)
// Lists longer than kRecallChunk go through the table a chunk per launch (32 KB of LDS at most: fits any workgroup
// limit); the ids of one approximate list are distinct (a top-k result), so no exact id is counted in two chunks.
constexpr int kRecallChunk = 4096;
__global__ __launch_bounds__(256) void recall_at_k_kernel(const int32_t* __restrict__ approx, int lda, int ka,
                                                         const int32_t* __restrict__ exact, int ke, int cap,
                                                         unsigned long long* __restrict__ hits) {
  extern __shared__ int32_t tab[];  // cap slots, cap = a power of two >= 2 ka; empty = INT32_MIN
  __shared__ int s_hits;
  const int64_t q = blockIdx.x;
  for (int i = threadIdx.x; i < cap; i += 256) tab[i] = INT32_MIN;
  if (threadIdx.x == 0) s_hits = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < ka; i += 256) {
    const int32_t v = approx[q * lda + i];
    if (v == INT32_MIN) continue;
    unsigned h = ((unsigned)v * 2654435761u) & (unsigned)(cap - 1);
    for (;;) {
      const int32_t old = atomicCAS(&tab[h], INT32_MIN, v);
      if (old == INT32_MIN || old == v) break;
      h = (h + 1) & (unsigned)(cap - 1);
    }
  }
  __syncthreads();
  int mine = 0;
  for (int j = threadIdx.x; j < ke; j += 256) {
    const int32_t v = exact[q * ke + j];
    if (v == INT32_MIN) continue;
    unsigned h = ((unsigned)v * 2654435761u) & (unsigned)(cap - 1);
    for (;;) {
      const int32_t t = tab[h];
      if (t == v) { ++mine; break; }
      if (t == INT32_MIN) break;
      h = (h + 1) & (unsigned)(cap - 1);
    }
  }
  if (mine) atomicAdd(&s_hits, mine);
  __syncthreads();
  if (threadIdx.x == 0 && s_hits) atomicAdd(hits, (unsigned long long)s_hits);
}

}  // namespace esr

using namespace esr;

extern "C" {

// ---- a prepared corpus: what a call does with the candidates before it looks at a query, done once --------------------
// [0, 256): header words -- [0] the candidates' exponent code (int; the scaled fp16 modes), [1] their largest row norm
// (float; mode 3), [2..3] N, [4] D; [256, 256 + 2 * kAbsBlocks * 4): the statistics pass's slots; then the mode's planes
// (1, 2 or 3), each k-block-major over ALL rows (rows_pad = N rounded up to the tile + one tile: a chunk's last tile may
// start anywhere).
static int64_t prepared_rows_pad(int64_t N) { return cdiv(N, kGN) * kGN + kGN; }
static size_t prepared_planes_off() { return align_up(256 + 2 * (size_t)kAbsBlocks * sizeof(float), 256); }
static int mode_planes(int mode) { return (mode == 0) ? 3 : (mode == 2 ? 2 : (mode == 3 ? 4 : 1)); }  // (retrieve_plan's P)
size_t esr_retrieve_prepared_bytes(int64_t N, int D, int mode) {
  if (N <= 0 || D <= 0 || mode < 0 || mode > 3) return 0;
  const int64_t Dp = cdiv(D, kGK) * kGK;
  return prepared_planes_off() + align_up((size_t)plane_count(mode_planes(mode)) * prepared_rows_pad(N) * Dp * 2, 256);
}

int esr_retrieve_prepare(const float* candidates, int64_t N, int D, int mode, void* prepared, size_t prepared_bytes,
                         esr_stream_t stream) {
  TraceScope trace_scope_("esr_retrieve_prepare");
  ESR_REQUIRE(mode >= 0 && mode <= 3, "esr_retrieve_prepare: mode %d", mode);
  ESR_REQUIRE(N > 0 && D > 0 && N < ((int64_t)1 << 31) && (mode != 3 || D <= kSelLdsWords),
              "esr_retrieve_prepare: bad sizes N=%lld D=%d", (long long)N, D);
  ESR_REQUIRE(candidates && prepared && !((uintptr_t)prepared & 255), "esr_retrieve_prepare: null or misaligned pointer");
  const int Dp = (int)(cdiv(D, kGK) * kGK);
  const int64_t rows_pad = prepared_rows_pad(N);
  ESR_REQUIRE((int64_t)Dp / kGK * rows_pad * 32 < ((int64_t)1 << 32),
              "esr_retrieve_prepare: N=%lld x D=%d exceeds the 32-bit tile offsets of one plane (4 GiB)", (long long)N, D);
  if (prepared_bytes < esr_retrieve_prepared_bytes(N, D, mode)) {
    set_error("esr_retrieve_prepare: buffer %zu bytes < %zu required", prepared_bytes, esr_retrieve_prepared_bytes(N, D, mode));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  char* base = (char*)prepared;
  int* hdr = (int*)base;
  float* abs_slots = (float*)(base + 256);
  float* nrm_slots = abs_slots + kAbsBlocks;
  const int64_t meta[2] = {N, (int64_t)D};
  if (hipMemcpyAsync(hdr + 2, meta, sizeof(meta), hipMemcpyHostToDevice, st) != hipSuccess) return check_launch("esr_retrieve_prepare");
  __bf16* planes = (__bf16*)(base + prepared_planes_off());
  const int P = mode_planes(mode);
  if (P == 4) {
    const int grid = (int)std::min<int64_t>(kAbsBlocks, std::max<int64_t>(1, cdiv(N, kBlock / 64)));
    hipLaunchKernelGGL(rowstat_kernel, dim3(grid), dim3(kBlock), 0, st, candidates, N, D, (float*)nullptr, abs_slots, nrm_slots);
    hipLaunchKernelGGL(absmax_exp_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)abs_slots, grid, hdr);
    hipLaunchKernelGGL(slots_max_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)nrm_slots, grid, (float*)(hdr + 1));
    launch_split<4>(candidates, N, D, rows_pad, Dp, rows_pad * Dp, planes, st, hdr);
  } else if (P == 2) {
    launch_absmax(candidates, N * (int64_t)D, abs_slots, hdr, st);
    launch_split<2>(candidates, N, D, rows_pad, Dp, rows_pad * Dp, planes, st, hdr);
  } else if (P == 3) {
    launch_split<3>(candidates, N, D, rows_pad, Dp, rows_pad * Dp, planes, st);
  } else {
    launch_split<1>(candidates, N, D, rows_pad, Dp, rows_pad * Dp, planes, st);
  }
  return check_launch("esr_retrieve_prepare");
}

size_t esr_retrieve_workspace_bytes(int64_t nq, int64_t N, int D, int k, int mode) {
  if (nq <= 0 || N <= 0 || D <= 0 || k <= 0) return 256;
  return retrieve_plan(nq, N, D, k, mode).total;
}

static int retrieve_topk_impl(const float* queries, const float* candidates, const char* prepared, int64_t nq, int64_t N,
                              int D, int k, int mode, int32_t index_base, int32_t index_step, float* out_scores,
                              int32_t* out_indices, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  ESR_REQUIRE(nq > 0 && N > 0 && D > 0 && k > 0 && k <= N && k <= kSelMaxK && N < ((int64_t)1 << 31) &&
                  nq < ((int64_t)1 << 24),
              "esr_retrieve_topk: bad sizes nq=%lld N=%lld D=%d k=%d (k <= min(N, %d))", (long long)nq, (long long)N, D,
              k, kSelMaxK);
  ESR_REQUIRE(mode >= 0 && mode <= 3,
              "esr_retrieve_topk: mode %d (0 = exact bf16x3, 1 = bf16, 2 = exact-grade f16x2, 3 = f16 filter + f32 re-score)",
              mode);
  ESR_REQUIRE(index_step > 0 && (int64_t)index_base + (N - 1) * (int64_t)index_step < ((int64_t)1 << 31),
              "esr_retrieve_topk: index_base/index_step overflow int32");
  ESR_REQUIRE(queries && candidates && out_scores && out_indices && workspace, "esr_retrieve_topk: null pointer");
  ESR_REQUIRE(mode != 3 || D <= kSelLdsWords, "esr_retrieve_topk: mode 3 takes D <= %d", kSelLdsWords);
  const RetrievePlan p = retrieve_plan(nq, N, D, k, mode);
  if (workspace_bytes < p.total || ((uintptr_t)workspace & 15)) {
    set_error("esr_retrieve_topk: workspace %zu bytes < %zu required (or misaligned)", workspace_bytes, p.total);
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  char* base = (char*)workspace;
  __bf16* A = (__bf16*)(base + p.off_A);
  __bf16* B = (__bf16*)(base + p.off_B);
  float* S = (float*)(base + p.off_S);
  int2* pairs = (int2*)(base + p.off_pairs);
  int32_t* cnt = (int32_t*)(base + p.off_cnt);
  float* tau = (float*)(base + p.off_tau);
  const int64_t a_plane = p.Mp * p.Dp, b_plane = p.chunk_pad * p.Dp;
  float* abs_slots = (float*)(base + p.off_abs);
  int* exps = (int*)(abs_slots + kAbsBlocks);  // {eq, ec}
  // mode 3: norm slots [kAbsBlocks], the largest candidate norm [1] (+ padding), |q| per query [nq], nexact [nq]
  float* nrm_slots = (float*)(base + p.off_band);
  float* cmax = nrm_slots + kAbsBlocks;
  float* qnorm = nrm_slots + kAbsBlocks + 64;
  int32_t* nexact = (int32_t*)(base + p.off_nexact);
  const bool band = p.P == 4;
  const float band_coef = 1.02f * 0.0009765625f;  // 2^-10 (1 + slack): see the header's mode-3 note
  if (p.P == 2) {
    // one exponent per matrix; the candidates' covers ALL chunks (one 4 N D-byte read, ~1 % of the call at N = 1 M)
    launch_absmax(queries, nq * (int64_t)D, abs_slots, exps, st);
    if (prepared) {
      if (hipMemcpyAsync(exps + 1, prepared, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return check_launch("esr_retrieve_topk_prepared");
    } else {
      launch_absmax(candidates, N * (int64_t)D, abs_slots, exps + 1, st);
    }
    launch_split<2>(queries, nq, D, p.Mp, p.Dp, a_plane, A, st, exps);
  } else if (band) {
    // exponents as above, and in the same read of each matrix the row norms the error bound of a one-plane score needs
    auto rowstat = [&](const float* X, int64_t rows, float* norms, int* exp_out, float* nmax_out) {
      const int grid = (int)std::min<int64_t>(kAbsBlocks, std::max<int64_t>(1, cdiv(rows, kBlock / 64)));
      hipLaunchKernelGGL(rowstat_kernel, dim3(grid), dim3(kBlock), 0, st, X, rows, D, norms, abs_slots, nrm_slots);
      hipLaunchKernelGGL(absmax_exp_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)abs_slots, grid, exp_out);
      if (nmax_out) hipLaunchKernelGGL(slots_max_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)nrm_slots, grid, nmax_out);
    };
    rowstat(queries, nq, qnorm, exps, nullptr);
    if (prepared) {  // the candidates' exponent and largest norm were found when the corpus was prepared
      if (hipMemcpyAsync(exps + 1, prepared, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpyAsync(cmax, prepared + sizeof(int), sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return check_launch("esr_retrieve_topk_prepared");
    } else {
      rowstat(candidates, N, nullptr, exps + 1, cmax);
    }
    if (hipMemsetAsync(nexact, 0, (size_t)nq * sizeof(int32_t), st) != hipSuccess) return check_launch("esr_retrieve_topk");
    launch_split<4>(queries, nq, D, p.Mp, p.Dp, a_plane, A, st, exps);
  } else if (p.P == 3) launch_split<3>(queries, nq, D, p.Mp, p.Dp, a_plane, A, st);
  else launch_split<1>(queries, nq, D, p.Mp, p.Dp, a_plane, A, st);

  int64_t c0 = 0;
  int ncall = 0;
  bool first = true;
  const char* lz = getenv("ESR_RETRIEVE_LAZY");  // "0": compact after every chunk (round 3)
  const int lazy_mark = (lz && lz[0] == '0') ? 0 : p.skip_upto;
  while (c0 < N) {
    const int64_t nc = std::min<int64_t>(first ? p.first : p.chunk, N - c0);
    const int64_t n_pad = cdiv(nc, kGN) * kGN;
    const bool last = (c0 + nc == N);
    // a prepared corpus: the chunk's rows inside the plane over ALL rows (k-block-major: a row is 16 elements per k-block)
    const __bf16* Bc = B;
    int64_t bc_plane = b_plane, bc_rows = 0;
    if (prepared) {
      bc_rows = prepared_rows_pad(N);
      bc_plane = bc_rows * p.Dp;
      Bc = (const __bf16*)(prepared + prepared_planes_off()) + c0 * kGK;
    } else if (p.P == 2) launch_split<2>(candidates + c0 * D, nc, D, n_pad, p.Dp, b_plane, B, st, exps + 1);
    else if (band) launch_split<4>(candidates + c0 * D, nc, D, n_pad, p.Dp, b_plane, B, st, exps + 1);
    else if (p.P == 3) launch_split<3>(candidates + c0 * D, nc, D, n_pad, p.Dp, b_plane, B, st);
    else launch_split<1>(candidates + c0 * D, nc, D, n_pad, p.Dp, b_plane, B, st);
    GemmOut o;
    o.exps = exps;
    o.S = S; o.ldS = p.first; o.tau = tau; o.cnt = cnt; o.pairs = pairs; o.ppitch = p.ppitch;
    o.gbase = index_base + (int32_t)c0 * index_step; o.gstep = index_step;
    SelIn in;
    SelOut so;
    so.pairs = pairs; so.ppitch = p.ppitch; so.cnt = cnt; so.tau = tau;
    so.scores = (last && !band) ? out_scores : nullptr;   // (mode 3 delivers after its re-score, below)
    so.indices = (last && !band) ? out_indices : nullptr;
    if (first) {
      if (p.P == 2) launch_gemm<2, true>(A, a_plane, Bc, bc_plane, p.Dp, p.Mp, n_pad, (int)nq, (int)nc, o, st, bc_rows);
      else if (band) launch_gemm<4, true>(A, a_plane, Bc, bc_plane, p.Dp, p.Mp, n_pad, (int)nq, (int)nc, o, st, bc_rows);
      else if (p.P == 3) launch_gemm<3, true>(A, a_plane, Bc, bc_plane, p.Dp, p.Mp, n_pad, (int)nq, (int)nc, o, st, bc_rows);
      else launch_gemm<1, true>(A, a_plane, Bc, bc_plane, p.Dp, p.Mp, n_pad, (int)nq, (int)nc, o, st, bc_rows);
      in.vals = S; in.vpitch = p.first; in.idx = nullptr; in.stride = 1; in.ibase = o.gbase; in.istep = index_step;
      in.n_per_row = nullptr; in.n_fixed = (int)nc;
    } else {
      if (p.P == 2) launch_gemm<2, false>(A, a_plane, Bc, bc_plane, p.Dp, p.Mp, n_pad, (int)nq, (int)nc, o, st, bc_rows);
      else if (band) launch_gemm<4, false>(A, a_plane, Bc, bc_plane, p.Dp, p.Mp, n_pad, (int)nq, (int)nc, o, st, bc_rows);
      else if (p.P == 3) launch_gemm<3, false>(A, a_plane, Bc, bc_plane, p.Dp, p.Mp, n_pad, (int)nq, (int)nc, o, st, bc_rows);
      else launch_gemm<1, false>(A, a_plane, Bc, bc_plane, p.Dp, p.Mp, n_pad, (int)nq, (int)nc, o, st, bc_rows);
      in.vals = (const float*)pairs; in.vpitch = 2 * p.ppitch; in.idx = (const int32_t*)pairs + 1; in.stride = 2;
      in.ibase = 0; in.istep = 0; in.n_per_row = cnt; in.n_fixed = 0;
      in.skip_upto = last ? 0 : lazy_mark;  // the last chunk's select delivers the answer: every row
    }
    // rows beyond what a workgroup's registers hold (2048 records) are cached in LDS (up to 4096 records)
    const int lds_words = kSelLdsWords;
    if (band) {
      in.band_qnorm = qnorm; in.band_cmax = cmax; in.band_coef = band_coef; in.band_nexact = nexact;
      in.band_cap = p.skip_upto;
      // (exact rows: their workgroup re-scores what this chunk appended, in front of its select)
      in.rs_Q = queries; in.rs_C = candidates; in.rs_D = D; in.rs_base = index_base; in.rs_step = index_step;
    }
    ESR_KT("topk_select_kernel", st, hipLaunchKernelGGL(topk_select_kernel, dim3((int)nq), dim3(kSelThreads), lds_words * sizeof(uint32_t), st, in, k, so,
                       lds_words));
    if (band) {  // rows whose band outgrew the list turn exact at once: re-score everything they hold, cut to the k best
      SelIn in2 = in;
      in2.vals = (const float*)pairs; in2.vpitch = 2 * p.ppitch; in2.idx = (const int32_t*)pairs + 1; in2.stride = 2;
      in2.ibase = 0; in2.istep = 0; in2.n_per_row = cnt; in2.n_fixed = 0; in2.skip_upto = 0; in2.only_pending = 1;
      ESR_KT("topk_select_kernel", st, hipLaunchKernelGGL(topk_select_kernel, dim3((int)nq), dim3(kSelThreads), lds_words * sizeof(uint32_t), st, in2, k, so,
                         lds_words));
    }
    ++ncall;
    c0 += nc;
    first = false;
  }
  if (band) {
    // the survivors of the one-term filter (band rows: every record; exact rows are exact already) in f32, then the
    // k best of every list, best first
    ESR_KT("rescore_lists_kernel", st,
           hipLaunchKernelGGL(rescore_lists_kernel, dim3((int)nq), dim3(kBlock), (size_t)D * sizeof(float), st, queries,
                              candidates, D, pairs, p.ppitch, (const int32_t*)cnt, index_base, index_step,
                              (const int32_t*)nexact));
    if (int rc = select_topk_tail(pairs, p.ppitch, cnt, nq, k, out_scores, out_indices, st)) return rc;
  }
  return check_launch("esr_retrieve_topk");
}

int esr_retrieve_topk(const float* queries, const float* candidates, int64_t nq, int64_t N, int D, int k, int mode,
                      int32_t index_base, int32_t index_step, float* out_scores, int32_t* out_indices, void* workspace,
                      size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_retrieve_topk");
  return retrieve_topk_impl(queries, candidates, nullptr, nq, N, D, k, mode, index_base, index_step, out_scores, out_indices,
                            workspace, workspace_bytes, stream);
}

int esr_retrieve_topk_prepared(const float* queries, const float* candidates, const void* prepared, int64_t nq, int64_t N,
                               int D, int k, int mode, int32_t index_base, int32_t index_step, float* out_scores,
                               int32_t* out_indices, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_retrieve_topk_prepared");
  ESR_REQUIRE(mode >= 0 && mode <= 3, "esr_retrieve_topk_prepared: mode %d", mode);
  ESR_REQUIRE(prepared && !((uintptr_t)prepared & 255), "esr_retrieve_topk_prepared: null or misaligned prepared corpus");
  ESR_REQUIRE(N > 0 && D > 0 && (int64_t)(cdiv(D, kGK)) * prepared_rows_pad(N) * 32 < ((int64_t)1 << 32),
              "esr_retrieve_topk_prepared: bad sizes N=%lld D=%d", (long long)N, D);
  return retrieve_topk_impl(queries, candidates, (const char*)prepared, nq, N, D, k, mode, index_base, index_step,
                            out_scores, out_indices, workspace, workspace_bytes, stream);
}

#ifdef ESR_GEMM_TIMING
int esr_gemm_debug_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(esr_gemm_dbg), sizeof(unsigned long long) * 8 * 1024);
}
#endif

int esr_rescore_candidates(const float* queries, const float* candidates, int64_t nq, int64_t N, int D,
                           const int32_t* indices, int kc, int32_t index_base, int32_t index_step, float* scores,
                           esr_stream_t stream) {
  ESR_REQUIRE(nq > 0 && N > 0 && D > 0 && kc > 0 && index_step > 0, "esr_rescore_candidates: bad sizes");
  ESR_REQUIRE(queries && candidates && indices && scores, "esr_rescore_candidates: null pointer");
  const int64_t total = nq * kc;
  const int grid = (int)std::min<int64_t>(cdiv(total, kBlock / 64), 16384);
  hipLaunchKernelGGL(rescore_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), queries, candidates, N, D, indices,
                     total, kc, index_base, index_step, scores);
  return check_launch("esr_rescore_candidates");
}

__global__ __launch_bounds__(256) void flip_indices_kernel(int32_t* __restrict__ idx, int64_t n, int32_t last) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) idx[i] = last - idx[i];
}

// The k LAST rows of the stable ascending argsort of every column of scores [V, T] (find_knn / dump_knn,
// wikipedia/train_cooccurence.py:91-97,114-126, read the last 10) without sorting the column: the radix select over the
// column read BACKWARDS -- position c = row V - 1 - c, so among equal scores the select's "lower position first" is the
// stable argsort's "higher row last" -- then the positions are turned back into rows.  Output [T, k], best first:
// out[t][j] = argsort(scores[:, t])[V - 1 - j].
int esr_topk_columns(const float* scores, int64_t V, int T, int k, float* out_scores, int32_t* out_indices,
                     esr_stream_t stream) {
  ESR_REQUIRE(V > 0 && T > 0 && k > 0 && k <= V && k <= kSelMaxK && V < ((int64_t)1 << 31),
              "esr_topk_columns: bad sizes V=%lld T=%d k=%d (k <= %d)", (long long)V, T, k, kSelMaxK);
  ESR_REQUIRE(scores && out_scores && out_indices, "esr_topk_columns: null pointer");
  hipStream_t st = as_stream(stream);
  SelIn in;
  in.vals = scores + (V - 1) * T; in.vpitch = 1; in.idx = nullptr; in.stride = -T; in.ibase = 0; in.istep = 1;
  in.n_per_row = nullptr; in.n_fixed = (int)V;
  SelOut so;
  so.pairs = nullptr; so.ppitch = 0; so.cnt = nullptr; so.tau = nullptr; so.scores = out_scores; so.indices = out_indices;
  hipLaunchKernelGGL(topk_select_kernel, dim3(T), dim3(kSelThreads), 0, st, in, k, so, 0);
  hipLaunchKernelGGL(flip_indices_kernel, dim3((int)std::min<int64_t>(64, cdiv((int64_t)T * k, 256))), dim3(256), 0, st,
                     out_indices, (int64_t)T * k, (int32_t)(V - 1));
  return check_launch("esr_topk_columns");
}

int esr_topk_merge(const float* scores, const int32_t* indices, int64_t nq, int n, int k, float* out_scores,
                   int32_t* out_indices, esr_stream_t stream) {
  ESR_REQUIRE(nq > 0 && n > 0 && k > 0 && k <= n && k <= kSelMaxK, "esr_topk_merge: bad sizes nq=%lld n=%d k=%d",
              (long long)nq, n, k);
  ESR_REQUIRE(scores && indices && out_scores && out_indices, "esr_topk_merge: null pointer");
  SelIn in;
  in.vals = scores; in.vpitch = n; in.idx = indices; in.stride = 1; in.ibase = 0; in.istep = 0;
  in.n_per_row = nullptr; in.n_fixed = n;
  SelOut so;
  so.pairs = nullptr; so.ppitch = 0; so.cnt = nullptr; so.tau = nullptr; so.scores = out_scores; so.indices = out_indices;
  const int lds_words = n > kSelThreads * kSelBatch && 2 * n <= kSelLdsWords ? kSelLdsWords : 0;
  hipLaunchKernelGGL(topk_select_kernel, dim3((int)nq), dim3(kSelThreads), lds_words * sizeof(uint32_t),
                     as_stream(stream), in, k, so, lds_words);
  return check_launch("esr_topk_merge");
}


int esr_recall_at_k(const int32_t* approx, int64_t nq, int ka, const int32_t* exact, int ke, unsigned long long* hits,
                    esr_stream_t stream) {
  TraceScope trace_scope_("esr_recall_at_k");
  ESR_REQUIRE(nq >= 0 && ka > 0 && ke > 0, "esr_recall_at_k: bad sizes nq=%lld ka=%d ke=%d", (long long)nq, ka, ke);
  ESR_REQUIRE(hits && (nq == 0 || (approx && exact)), "esr_recall_at_k: null pointer");
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(hits, 0, sizeof(unsigned long long), st) != hipSuccess) return check_launch("esr_recall_at_k");
  if (nq == 0) return ESR_OK;
  for (int a0 = 0; a0 < ka; a0 += kRecallChunk) {
    const int len = std::min(kRecallChunk, ka - a0);
    int cap = 64;
    while (cap < 2 * len) cap <<= 1;
    hipLaunchKernelGGL(recall_at_k_kernel, dim3((unsigned)nq), dim3(256), (size_t)cap * sizeof(int32_t), st, approx + a0, ka,
                       len, exact, ke, cap, hits);
  }
  return check_launch("esr_recall_at_k");
}

}  // extern "C"
