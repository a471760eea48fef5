// Shared pieces of the split-precision in-batch kernels (esr_inbatch3.hip: bf16 x 3 planes; esr_inbatch2h.hip: fp16 x 2
// planes): tile geometry, the LDS swizzle, the transposing LDS read, the row source (dense matrix or gathered tower
// rows), and the merge kernel that turns the per-split partial (l, O) into gradient rows and the loss.
#pragma once
#include "esr_common.h"
#include <stdlib.h>

namespace esr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int k3D = 128;
constexpr int k3Chunk = 32;
constexpr int k3Waves = 4;
constexpr int k3Owned = 32 * k3Waves;  // owned rows per workgroup
constexpr float k3Log2e = 1.4426950408889634f;
constexpr float k3Ln2 = 0.6931471805599453f;

// LDS image of one 32-row chunk: three row-major planes [32 rows][256 B] (24 KB; without kUseTr three transposed
// planes [128 d][64 B] follow, 48 KB), filled by direct global->LDS DMA (global_load_lds_dwordx4: no staging VGPRs, no
// ds_write).  The DMA destination is lane-linear, so bank conflicts are removed by an XOR swizzle applied
// to the per-lane SOURCE address and again on the ds_read_b128 address (same involution on both sides):
//   row-major   : 16-B segment index ^= swz16(row)
//   transposed  : 16-B segment index ^= ((d >> 2) & 3)
constexpr int kPlaneBytes = 8192;
// kUseTr: the O^T phase takes its A operand (Y^T) from the ROW-MAJOR image with the transposing LDS read
// ds_read_b64_tr_b16 (four k-rows x 16 d-columns per 16-lane group, delivered column-per-lane), so the transposed
// image -- half of every chunk's DMA pieces and half of the LDS ring -- is not needed.
#ifndef ESR_IB3_USE_TR
#define ESR_IB3_USE_TR 1
#endif
constexpr bool kUseTr = ESR_IB3_USE_TR != 0;
constexpr int kTOff = 3 * kPlaneBytes;   // 24576 (transposed planes, only without kUseTr)
constexpr int kLseOff = (kUseTr ? 3 : 6) * kPlaneBytes;  // 128 B of streamed-row lse (pass C) behind the planes
constexpr int kBufBytes = kLseOff + 256;
// 16-B segment swizzle of a row-major row.  j & 15 serves the ds_read_b128 fragment reads (16 distinct rows per lane
// group); the transposing reads touch 4 consecutive rows x 4 consecutive segments per 16-lane group and need the 4
// rows on 4 different segment quads: swap the two bit pairs (still a bijection on 0..15, so b128 stays conflict-free).
__device__ __forceinline__ constexpr int swz16(int row) {
  return kUseTr ? (((row & 3) << 2) | ((row >> 2) & 3)) : (row & 15);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// ds_read_b64_tr_b16 as inline assembly (see ESR_O_LOAD for why not the builtin); result valid after lgkmcnt(0)
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(uint32_t lds_addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
  return v;
}
// A fragment F (0..11: plane (F / 4 + 2) % 3 -- the order the O^T rows use them -- column block F % 4) of k-step G:
// two transposing reads from the per-lane bases of this chunk (tc[db][0 / 1], see the kernel), the plane and
// k-step as the instruction's immediate offset so that no address arithmetic is left between the MFMAs.
template <int G, int F, class TA>
__device__ __forceinline__ void tr_frag(TA& ta, const uint32_t (&tc)[4][2]) {
  constexpr int PL = (F / 4 + 2) % 3, DB = F % 4, OFF = PL * kPlaneBytes + 16 * G * 256;
  const s16x4 lo = tr_read<OFF>(tc[DB][0]), hi = tr_read<OFF>(tc[DB][1]);
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  ta[G][DB][PL] = __builtin_bit_cast(bf16x8, both);
}
template <int G, class TA>
__device__ __forceinline__ void tr_frag_n(int f, TA& ta, const uint32_t (&tc)[4][2]) {  // f is an unrolled constant
  switch (f) {
    case 0: tr_frag<G, 0>(ta, tc); break;
    case 1: tr_frag<G, 1>(ta, tc); break;
    case 2: tr_frag<G, 2>(ta, tc); break;
    case 3: tr_frag<G, 3>(ta, tc); break;
    case 4: tr_frag<G, 4>(ta, tc); break;
    case 5: tr_frag<G, 5>(ta, tc); break;
    case 6: tr_frag<G, 6>(ta, tc); break;
    case 7: tr_frag<G, 7>(ta, tc); break;
    case 8: tr_frag<G, 8>(ta, tc); break;
    case 9: tr_frag<G, 9>(ta, tc); break;
    case 10: tr_frag<G, 10>(ta, tc); break;
    default: tr_frag<G, 11>(ta, tc); break;
  }
}
constexpr int k3Bufs = 3;
constexpr int kLossWords = 64;        // first-level accumulators of the merge kernels' loss reduction
constexpr int k3MergeBlocks = 1024;  // most workgroups of a merge launch (= loss partials per launch)

__device__ __forceinline__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// position of streamed row `row` (0..31) inside a transposed chunk: pos = 16 g + 8 h + k  <->
// row = (k & 3) + 8 (2 g + (k >> 2)) + 4 h   (the S^T accumulator order)
__device__ __forceinline__ int row_to_pos(int row) {
  const int k_lo = row & 3, h = (row >> 2) & 1, k_hi = (row >> 3) & 1, g = row >> 4;
  return 16 * g + 8 * h + 4 * k_hi + k_lo;
}

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// {bf16(lo), bf16(hi)} packed in one dword (v_cvt_pk_bf16_f32, round-to-nearest-even)
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float pk_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float pk_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// ---------------------------------------------------------------------------------------------------
// Where the rows of Q / C come from: a dense [B, 128] f32 matrix (idx == null), or rows idx[i] of a tower table
// (f32 or bf16) -- the gather is then folded into the split pre-pass and the merge kernels, and the step needs no
// materialised Q / C at all.
struct RowSrc {
  const void* base;
  const int32_t* idx;
  int bf16;
  int ld;  // row length of the source = number of valid columns (a multiple of 4, <= 128); columns beyond it read as 0
};
__device__ __forceinline__ float4 rowsrc_load4(const RowSrc& s, int64_t row, int d) {  // elements d .. d+3 of row
  if (d >= s.ld) return make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding up to the 128-column tile
  const int64_t r = s.idx ? (int64_t)s.idx[row] : row;
  if (s.bf16) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(s.base) + r * s.ld + d);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(s.base) + r * s.ld + d);
}


typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------------------------------
// One owned row's merge: partial O rows of the splits -> gradient row, loss term, normaliser.  ONE definition with
// explicit roundings for the merge launches and for the update kernel that merges on the fly
// (esr_optim.hip inbatch_merge_update_kernel): the two produce the same bits.
//   po / pm / pl: the splits' partial O quads (this lane's four columns), exponent references and normalisers (entries at
//   s >= nsplit: 0 / -inf / 0);  x / y: the lane's quad of the owned row and of its partner row;  shared_ref: every split
//   exponentiated against one reference (the bf16 x 3 path; weights 1).
// Returns the gradient quad; row_loss, M, L, invL1 and the split weights wt[] come back by reference.
// ---------------------------------------------------------------------------------------------------
template <bool QSIDE>
__device__ __forceinline__ float4 merge_row(const float4 (&po)[8], const float (&pm)[8], const float (&pl)[8], int nsplit,
                                            bool shared_ref, float4 x, float4 y, float oscale, float scale, float lam,
                                            float inv_bs, int G, float& row_loss, float& M, float& L, float& invL1,
                                            float (&wt)[8]) {
  M = 0.f;
  L = 1.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) wt[s] = 1.f;
  if (QSIDE) {  // fp16 x 2 path: split s used the fixed reference pm[s]; the row's is the largest of them
    M = pm[0];
    L = 0.f;
#pragma unroll
    for (int s = 1; s < 8; ++s) M = fmaxf(M, pm[s]);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      wt[s] = (shared_ref || pm[s] == M) ? 1.f : __builtin_amdgcn_exp2f(pm[s] - M);
      L = __fmaf_rn(pl[s], wt[s], L);  // (fac2h_kernel must agree bit for bit)
    }
  }
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < nsplit) {
      o.x = __fmaf_rn(po[s].x, wt[s], o.x); o.y = __fmaf_rn(po[s].y, wt[s], o.y);
      o.z = __fmaf_rn(po[s].z, wt[s], o.z); o.w = __fmaf_rn(po[s].w, wt[s], o.w);
    }
  }
  invL1 = __fdiv_rn(1.0f, L);
  const float invL = __fmul_rn(invL1, oscale);  // (exact: oscale is a power of two)
  const float xn2 =
      group_sum(__fmaf_rn(x.w, x.w, __fmaf_rn(x.z, x.z, __fmaf_rn(x.y, x.y, __fmul_rn(x.x, x.x)))), G);
  const float xnorm = __fsqrt_rn(xn2);
  const float creg = xnorm > 1.f ? __fdiv_rn(lam, xnorm) : 0.f;
  float4 g;
  g.x = __fmul_rn(__fmaf_rn(scale, __fmaf_rn(o.x, invL, -y.x), __fmul_rn(creg, x.x)), inv_bs);
  g.y = __fmul_rn(__fmaf_rn(scale, __fmaf_rn(o.y, invL, -y.y), __fmul_rn(creg, x.y)), inv_bs);
  g.z = __fmul_rn(__fmaf_rn(scale, __fmaf_rn(o.z, invL, -y.z), __fmul_rn(creg, x.z)), inv_bs);
  g.w = __fmul_rn(__fmaf_rn(scale, __fmaf_rn(o.w, invL, -y.w), __fmul_rn(creg, x.w)), inv_bs);
  row_loss = __fmul_rn(lam, fmaxf(__fsub_rn(xnorm, 1.f), 0.f));
  if (QSIDE) {
    const float diag =
        group_sum(__fmaf_rn(x.w, y.w, __fmaf_rn(x.z, y.z, __fmaf_rn(x.y, y.y, __fmul_rn(x.x, y.x)))), G);
    const float l2v = __fadd_rn(M, __builtin_amdgcn_logf(L));
    row_loss = __fadd_rn(row_loss, __fmaf_rn(l2v, k3Ln2, -__fmul_rn(scale, diag)));
  }
  return g;
}
// the loss term of one row in 2^-28 fixed point: integer sums of these do not depend on how rows are grouped into
// workgroups or launches (the merge launches and the merging update kernel add up to the same word)
__device__ __forceinline__ long long loss_fixed(float row_loss, bool& bad) {
  if (!(fabsf(row_loss) < 16777216.0f)) {
    bad = true;
    return 0ll;
  }
  return __double2ll_rn((double)row_loss * 268435456.0);
}
// sum of one long long per thread over the workgroup (kBlock threads; smem: 4 words)
__device__ __forceinline__ long long block_sum_ll(long long v, long long* smem) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int lo = __shfl_xor((int)(v & 0xffffffffll), o, 64), hi = __shfl_xor((int)(v >> 32), o, 64);
    v += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
  __syncthreads();
  long long t = 0;
  for (int w = 0; w < kBlock / 64; ++w) t += smem[w];
  return t;
}
// One workgroup's loss sum (2^-28 fixed point; `bad`: some row's term was non-finite or out of range) into the counted
// words of loss_acc; the last of `arrivals_per_word`-counted arrivals forwards and the very last writes the loss.
// Integer addition is exact and order-free, so the sum is bit-reproducible, and the atomic's return value tells the
// workgroup whether it was the last to arrive.  Two levels, because 2048 atomics on one address serialise (measured:
// +15 us): kLossWords words 128 B apart take workgroups blockIdx % kLossWords; the last arrival of a word forwards that
// word's total to the master word; the last arrival there writes the loss.  Data flows only through atomic return
// values, so no ordering between addresses is needed.  The words were zeroed by the op's first launch.  (A ticket +
// __threadfence() reduction of double partials was 7 us slower than a finalize launch: the agent-scope release writes
// the XCD's L2 back.)  Range: |sum| < 2^24 = 1.6e7 nats (beyond it, or non-finite: NaN); resolution 3.7e-9 per row.
// launches: how many launches of THIS grid add to the words (the two merge launches: 2; the merging update: 1).
__device__ __forceinline__ void loss_arrive(long long tsum, bool bad, int launches, unsigned long long* loss_acc,
                                            double loss_scale, float* loss_out) {
  const unsigned wd = blockIdx.x % kLossWords;
  const unsigned per_launch = (gridDim.x - wd + kLossWords - 1) / kLossWords;  // workgroups of one launch on word wd
  const unsigned nwords = gridDim.x < (unsigned)kLossWords ? gridDim.x : (unsigned)kLossWords;
  unsigned long long add = ((unsigned long long)tsum << 11);
  if (bad || !(tsum < (1ll << 52) && tsum > -(1ll << 52))) {  // the loss must come out NaN, not a wrapped number
    // raise the poison word BEFORE this workgroup is counted: the add below consumes the atomic's return value, so
    // it cannot be issued until the OR has been performed (r is 0 or 1; r >> 1 is the dependence, not a value)
    const unsigned r = atomicOr(reinterpret_cast<unsigned*>(loss_acc + 8), 1u);
    add = (unsigned long long)(r >> 1);
  }
  const unsigned long long old = atomicAdd(loss_acc + 16 * (1 + wd), add + 1ull);
  if ((unsigned)(old & 2047ull) == (unsigned)launches * per_launch - 1) {
    const unsigned long long word_total = ((old + add) >> 11) << 11;  // this word's sum, count bits cleared
    const unsigned long long m = atomicAdd(loss_acc, word_total + 1ull);
    if ((unsigned)(m & 2047ull) == nwords - 1) {
      const long long tot = ((long long)(m + word_total)) >> 11;  // arithmetic shift: signed sum
      // every workgroup was counted before this branch was taken, hence after its OR (if any) was performed
      const bool poisoned = atomicOr(reinterpret_cast<unsigned*>(loss_acc + 8), 0u) != 0u;
      loss_out[0] = poisoned ? __builtin_nanf("") : (float)((double)tot * (1.0 / 268435456.0) * loss_scale);
    }
  }
}

// What the in-batch train step hands the MERGING update (esr_optim.hip inbatch_merge_update) instead of gradient rows:
// side 0 = the query tower's occurrences (positions of the occurrence list below B), side 1 = the candidate tower's.
struct InbatchMergeArgs {
  const float* part_O[2];   // the side's partial O rows, [nsplit][B][128]
  int nsplit[2];
  const float* oscale[2];   // device: the power of two that undoes the planes' scaling of the side's partial rows
  const float* partner[2];  // the OTHER tower's gathered rows as dense f32 [B][128]: copies taken before any update
  const float* part_m;      // pass Q's exponent references and normalisers, [nsplit[0]][B]
  const float* part_l;
  int64_t B;
  float scale, lam, inv_bs;
  unsigned long long* loss_acc;
  double loss_scale;
  float* loss_out;
  unsigned long long* zero_words;  // cleared for the next call (see inbatch3_merge_kernel)
  int nzero;
};

// ---------------------------------------------------------------------------------------------------
// merge: one 32-lane group per owned row
// ---------------------------------------------------------------------------------------------------
template <bool QSIDE>
__global__ __launch_bounds__(kBlock) void inbatch3_merge_kernel(
    RowSrc X, RowSrc Y, const int32_t* __restrict__ out_idx, int64_t B, int nsplit, const float* __restrict__ part_O,
    const float* __restrict__ part_m, const float* __restrict__ part_l, float scale, float lam, float inv_bs,
    float* __restrict__ lse2, float* __restrict__ lse_nat, float* __restrict__ gX,
    unsigned long long* __restrict__ loss_acc, double loss_scale, float* __restrict__ loss_out,
    float* __restrict__ invl, const float* __restrict__ oscale_ptr = nullptr, float invl_scale = 1.0f,
    float* __restrict__ fac = nullptr, unsigned long long* __restrict__ zero_words = nullptr, int nzero = 0) {
  // zero_words (fp16 x 2 path, the op's LAST launch): prepsplit2h_kernel's tagged per-chunk words are cleared for the next
  // call on this workspace -- a replayed hipGraph repeats the call's token, and a word left by the previous replay would
  // pass for this one's
  // fac (fp16 x 2 path, QSIDE): [nsplit][B] factors invl_scale * 2^(M_split - M) / l for the stored-P pass C, whose
  // probabilities carry the reference of the split that wrote them
  // oscale_ptr (fp16 x 2 path): a power of two that undoes the plane scaling of the partial O rows (device-side: it
  // depends on the largest |element| of the batch); invl_scale: a power of two folded into the stored 1 / l_i
  __shared__ long long sm[4];
  if (zero_words && blockIdx.x == 0)
    for (int i = threadIdx.x; i < nzero; i += kBlock) zero_words[i] = 0ull;
  const float oscale = oscale_ptr ? oscale_ptr[0] : 1.0f;
  constexpr int G = 32;
  const int lig = threadIdx.x & (G - 1);
  const int64_t gpb = kBlock / G;
  const int64_t group = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  long long acc_loss = 0;
  bool bad = false;
  for (int64_t row = group; row < B; row += ngroups) {
    // every load of the row is issued before the first use (nsplit <= 8 is a run-time value: the plain loops waited
    // for one memory latency per split and array, ~12 in a row); the sums keep the split order
    float pm[8], pl[8];
    float4 po[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      pm[s] = -INFINITY; pl[s] = 0.f; po[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s < nsplit) {
        if (QSIDE) {
          pm[s] = part_m[(int64_t)s * B + row];
          pl[s] = part_l[(int64_t)s * B + row];
        }
        po[s] = *reinterpret_cast<const float4*>(part_O + ((int64_t)s * B + row) * k3D + 4 * lig);
      }
    }
    const float4 x = rowsrc_load4(X, row, 4 * lig);
    const float4 y = rowsrc_load4(Y, row, 4 * lig);
    float M, L, invL1, row_loss;
    float wt[8];  // 2^(M_s - M): 1 for every split when they shared one reference (the bf16 x 3 path), 0 for s >= nsplit
    // (fac == null: the bf16 x 3 path -- part_m[s] is split s's share of the row maximum there, and every split
    // exponentiated against the maximum of them: weight 1)
    const float4 g = merge_row<QSIDE>(po, pm, pl, nsplit, fac == nullptr, x, y, oscale, scale, lam, inv_bs, G, row_loss, M, L,
                                      invL1, wt);
    if (4 * lig < X.ld)  // gradient rows have the source's width
      *reinterpret_cast<float4*>(gX + (out_idx ? (int64_t)out_idx[row] : row) * X.ld + 4 * lig) = g;
    if (QSIDE) {
      const float l2v = __fadd_rn(M, __builtin_amdgcn_logf(L));
      if (lig == 0) {
        lse2[row] = l2v;
        if (invl) invl[row] = invL1 * invl_scale;  // the stored-P pass C normalises with it
        if (fac) {
#pragma unroll
          for (int s = 0; s < 8; ++s)
            if (s < nsplit) fac[(int64_t)s * B + row] = __fmul_rn(__fmul_rn(invL1, invl_scale), wt[s]);
        }
        if (lse_nat) lse_nat[row] = l2v * k3Ln2;
      }
    }
    if (lig == 0) acc_loss += loss_fixed(row_loss, bad);
  }
  const long long tsum = block_sum_ll(acc_loss, sm);
  const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
  // both merge launches of a call have this grid and add to the same words (see loss_arrive)
  if (threadIdx.x == 0) loss_arrive(tsum, any_bad, 2, loss_acc, loss_scale, loss_out);
}

}  // namespace esr
