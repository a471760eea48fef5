// In-batch softmax forward + backward with FP32-GRADE products from TWO fp16 planes per operand.
//
// Same contract, flash structure and kernel skeleton as esr_inbatch3.hip (bf16 x 3 planes, read that first), half the
// matrix-core work.  The bf16 x 3 kernels run at the chip's POWER limit (1.7-1.8 GHz under them, profiles/r2): wall
// time follows the number of MFMAs, not the cycles a schedule wastes, so the lever is fewer MFMAs per product:
//
//   x' = x * 2^e (e per matrix, from the batch's largest |element|: max |x'| in [2^13, 2^14))
//   x1 = rn_f16(x'),  x2 = rn_f16(x' - x1)          |x' - x1 - x2| <= 2^-24 |x'|  (11 + 1 bits per plane with
//                                                     round-to-nearest; x2 is subnormal only below 2^-16 of the
//                                                     matrix maximum, absolute error then <= 2^-25 * 2^-e)
//   a.b ~= a2 b1 + a1 b2 + a1 b1                     dropped a2 b2 <= 2^-24 |a||b|
//
// i.e. THREE v_mfma_f32_32x32x16_f16 per product instead of six bf16 ones, error <= ~3 * 2^-24 |a||b| per elementary
// product -- the f32 rounding of the product itself (measured against the fp64 oracle: at or below the exact-f32 MFMA
// path, tests/test_gpu_kernels.py).  What fp16 costs is RANGE, and it is handled where it arises:
//   * operands: the per-matrix power-of-two scale above (absmax pre-pass, 8 MB read); undone by exact power-of-two
//     factors in the exponent argument and in the merge kernels;
//   * probabilities of pass Q: p' = exp2(s - M) must lie in fp16's range for every j of the row, with the row's largest
//     p' >= ~4.  Default: an OPTIMISTIC reference per (128-row block, split) workgroup, M = max(diagonal score, largest
//     score of the workgroup's first chunk) - 4, with an overflow flag and a redo launch that reruns the flagged
//     workgroups against the exact maximum of their range (see inbatch2h_q_kernel).  ESR_IB2H_REF=rowmax: the exact row
//     maxima from a hi-plane GEMM pre-pass instead (rowmax2h_kernel, 28 us at B = 8192; skipped when the
//     Cauchy-Schwarz bound on |s| is <= 14 log2 units, where the bound itself is a safe reference);
//   * probabilities of pass C: true probabilities p / l in [0, 1] with a row maximum >= 1 / B, scaled by 2^14.
// Pass C always reads the probabilities pass Q stored (inbatch2h_pc8_kernel): B <= 16384; larger batches and bf16
// tables take the bf16 x 3 path (bf16 tables are one-plane there already).
#include "esr_inbatch_mfma.h"
#include <atomic>
#include <type_traits>
#include <time.h>

namespace esr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kHLseOff = 2 * kPlaneBytes;   // two row-major planes, then (pass C) one 256-B block of per-streamed-row
constexpr int kHBufBytes = kHLseOff + 4 * 256;  // factors PER WAVE (a 4-byte LDS-DMA writes 64 lanes x 4 B)
constexpr float kHOptHead = 4.0f;           // optimistic reference: p' = 2^4 at the reference score, overflow 12 binades up
constexpr float kHOverflow = 60000.0f;      // a probability beyond it does not fit fp16 (max 65504): redo the block
constexpr int kHBufs = 3;
constexpr float kHPexp = 14.0f;             // probabilities are carried as p * 2^14
constexpr float kHBoundSafe = 14.0f;        // Cauchy-Schwarz bound (log2 units) below which no row maximum is needed
constexpr int kHScaleWords = 8;             // device-side scale block: see split2h_kernel

#define H_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define H_SB() __builtin_amdgcn_sched_barrier(0)
#define H_DMA_BARRIER()                                 \
  {                                                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
    __syncthreads();                                    \
  }
#define H_TR_WAIT() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); H_SB(); }
#ifdef H_TIMING  // debug build (scripts/gpu_ib2h_timing.sh): per-wave phase stamps; H_TIMING=1 pass Q, else pass C
__device__ unsigned long long esr_ib2h_dbg[8192];
#define H_TICK(VAR) { H_SB(); VAR = __builtin_readcyclecounter(); H_SB(); }
#define H_TIMING_Q ((H_TIMING + 0) == 1)
#define H_TIMING_DECL()                                                                     \
  const unsigned long long rentry = __builtin_amdgcn_s_memrealtime();                       \
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tacc0 = 0, tacc1 = 0, tacc2 = 0;   \
  unsigned long long tstart = 0, rstart = 0;
#define H_TIMING_START() { tstart = __builtin_readcyclecounter(); rstart = __builtin_amdgcn_s_memrealtime(); }
#define H_TIMING_ACC() { tacc0 += tk1 - tk0; tacc1 += tk2 - tk1; tacc2 += tk3 - tk2; }
#define H_TIMING_WRITE(ON)                                                                  \
  if (lane == 0 && w < 4 && (blockIdx.x & 1) == 0 && blockIdx.x < 512 && (ON)) { /* every other workgroup */  \
    unsigned long long* d = esr_ib2h_dbg + (((blockIdx.x >> 1) * 4 + w) * 4);               \
    d[0] = tacc0; d[1] = tacc1; d[2] = tacc2; d[3] = __builtin_readcyclecounter() - tstart; \
    unsigned long long* e = esr_ib2h_dbg + 4096 + (((blockIdx.x >> 1) * 4 + w) * 4);        \
    e[0] = rentry; e[1] = rstart; e[2] = __builtin_amdgcn_s_memrealtime(); e[3] = e[2];     \
  }
// (the one-plane kernel: realtime at the end of the sweep's last chunk and after the output stores)
#define H_TIMING_MARK(SLOT, ON)                                                             \
  if (lane == 0 && w < 4 && (blockIdx.x & 1) == 0 && blockIdx.x < 512 && (ON)) {            \
    esr_ib2h_dbg[4096 + (((blockIdx.x >> 1) * 4 + w) * 4) + (SLOT)] = __builtin_amdgcn_s_memrealtime(); \
  }
#else
#define H_TICK(VAR)
#define H_TIMING_MARK(SLOT, ON)
#define H_TIMING_DECL()
#define H_TIMING_START()
#define H_TIMING_ACC()
#define H_TIMING_WRITE(ON)
#endif

__device__ __forceinline__ f16x2 pk_f16(float lo, float hi) {  // v_cvt_pk_f16_f32, round-to-nearest-even
  const f32x2 v = {lo, hi};
  return __builtin_convertvector(v, f16x2);
}

// x - (float)h in one instruction (v_fma_mix_f32: the fp16 operand is widened inside the FMA; exact, the difference of
// an f32 and its own fp16 rounding is representable)
__device__ __forceinline__ float resid_f16(float x, _Float16 hval) { return __builtin_fmaf((float)hval, -1.0f, x); }
// The same on the two halves of a packed pair, written as the instruction itself: hipcc turns the C form back into
// v_cvt_f32_f16 + v_sub (5 VALU instructions per pair of probabilities instead of 3, in a phase bound by VALU issue)
__device__ __forceinline__ float resid_lo(float x, f16x2 pk) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
  return r;
}
__device__ __forceinline__ float resid_hi(float x, f16x2 pk) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
  return r;
}
// Arithmetic on register pairs WITHOUT the packed-f32 instructions.  scripts/mfma_valu_probe.py (round 5): behind every
// MFMA four v_fma_f32 / v_fma_mix_f32 / v_max3_f32 are almost free with two waves per SIMD (a round of four MFMAs 217 ->
// 224 cycles), four v_pk_fma_f32 or v_pk_add_f32 nearly double it (396): the packed pair does not run beside the matrix
// pipe, its two scalar halves do.  These kernels' VALU work is written with scalar instructions, and ESR_NO_PK keeps the
// compiler from pairing them up again.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  float r0, r1;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a[0]), "v"(b[0]), "v"(c[0]));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a[1]), "v"(b[1]), "v"(c[1]));
  return f32x2{r0, r1};
}
// (the leading s_nop: the operands are usually fresh v_exp_f32 results, and a VALU instruction that reads a transcendental's
// result needs one wait state in front of it -- hipcc inserts it for its own instructions, not for inline assembly.  The
// one-plane kernel's first schedule put this add right behind the second v_exp_f32: every row's normaliser was garbage.)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  float r0, r1;
  asm("s_nop 0\n\tv_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %5"
      : "=&v"(r0), "=&v"(r1)
      : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
  return f32x2{r0, r1};
}
__device__ __forceinline__ float fma_s(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// la += e0, lb += e1, m = max(m, e0, e1) for two fresh v_exp_f32 results (the wait state: see pk_add)
__device__ __forceinline__ void sum_max(float& la, float& lb, float& m, float e0, float e1) {
  asm("s_nop 0\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_max3_f32 %2, %2, %3, %4"
      : "+v"(la), "+v"(lb), "+v"(m)
      : "v"(e0), "v"(e1));
}

// A wave's 32 x 128 fp32 output tile (acc[db][4 q + e] = O[row j][32 db + 8 q + 4 h + e]: the MFMA's layout) to global
// rows of stride k3D THROUGH LDS.  Stored straight from the accumulators every instruction touches 32 rows with 32
// contiguous bytes each; the one-plane kernel spent ~14 us of its 55 between its last MFMA and its end on the 33 MB of
// partial rows that way (scripts/gpu_ib1h_timing.sh).  Here the tile crosses a wave-private 8.5 KB of LDS in two halves
// of 64 columns (rows padded to 272 B: the eight lanes a ds_write_b128 serves per cycle then fall on 32 different
// banks) and leaves as 256-byte row segments, four rows per instruction.  The caller has made sure (a workgroup
// barrier) that nobody still reads what `wave_lds` overlays.
constexpr int kTileLdsBytes = 32 * 272;
// PERM: tile row n is global row pi(n) = n with bits 2 and 3 swapped (inbatch2h_pct_kernel's tile columns).
template <bool PERM = false, class ACC>
__device__ __forceinline__ void store_tile_via_lds(char* wave_lds, const ACC& acc, float* __restrict__ tile_base, int lane) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int dbl = 0; dbl < 2; ++dbl)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(wave_lds + j * 272 + (32 * dbl + 8 * q + 4 * h) * 4) =
            make_float4(acc[2 * half + dbl][4 * q], acc[2 * half + dbl][4 * q + 1], acc[2 * half + dbl][4 * q + 2],
                        acc[2 * half + dbl][4 * q + 3]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + (lane >> 4), piece = lane & 15;
      const float4 v = *reinterpret_cast<const float4*>(wave_lds + row * 272 + piece * 16);
      const int grow = PERM ? ((row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1)) : row;
      *reinterpret_cast<float4*>(tile_base + (int64_t)grow * k3D + 64 * half + piece * 4) = v;
    }
  }
}

// A fragment F (0..7: plane (F / 4 + 1) % 2 -- the order the O^T rows use them -- column block F % 4) of k-step G
template <int G, int F, class TA>
__device__ __forceinline__ void trh_frag(TA& ta, const uint32_t (&tc)[4][2]) {
  constexpr int PL = (F / 4 + 1) % 2, DB = F % 4, OFF = PL * kPlaneBytes + 16 * G * 256;
  const s16x4 lo = tr_read<OFF>(tc[DB][0]), hi = tr_read<OFF>(tc[DB][1]);
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  ta[G][DB][PL] = __builtin_bit_cast(f16x8, both);
}
template <int G, class TA>
__device__ __forceinline__ void trh_frag_n(int f, TA& ta, const uint32_t (&tc)[4][2]) {  // f is an unrolled constant
  switch (f) {
    case 0: trh_frag<G, 0>(ta, tc); break;
    case 1: trh_frag<G, 1>(ta, tc); break;
    case 2: trh_frag<G, 2>(ta, tc); break;
    case 3: trh_frag<G, 3>(ta, tc); break;
    case 4: trh_frag<G, 4>(ta, tc); break;
    case 5: trh_frag<G, 5>(ta, tc); break;
    case 6: trh_frag<G, 6>(ta, tc); break;
    default: trh_frag<G, 7>(ta, tc); break;
  }
}

// byte offset of DMA piece K (0..3: plane K / 2, half K % 2) of chunk `chunk` from the base of the plane array
template <int K>
__device__ __forceinline__ uint32_t dmah_off0(int64_t B, int64_t chunk, int t) {
  const int within = (t + 256 * K) & 511;
  const int row = within >> 4, seg = (within & 15) ^ swz16(row);
  return (uint32_t)((((int64_t)(K >> 1) * B + chunk * 32 + row) * k3D + seg * 8) * 2);
}
template <int K>
__device__ __forceinline__ void dmah_piece(const char* __restrict__ base, uint32_t off, char* buf, int w) {
  __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(buf + K * 4096 + w * 1024), 16, 0, 0);
}
#define H_DP(K, G, BUF) dmah_piece<K>(baseY, G, (BUF), w)
#define H_DMA_ADVANCE()                                                                    \
  {                                                                                        \
    const uint32_t step_ = (dpos + 1 == nc) ? (uint32_t)(8192 - nc * 8192) : 8192u;        \
    dpos = (dpos + 1 == nc) ? 0 : dpos + 1;                                                \
    g0 += step_; g1 += step_; g2 += step_; g3 += step_;                                    \
  }
#define H_DMA_CHUNK(BUF) { H_DP(0, g0, BUF); H_DP(1, g1, BUF); H_DP(2, g2, BUF); H_DP(3, g3, BUF); H_DMA_ADVANCE(); }

// -----------------------------------------------------------------------------------------------------------------
// prep pre-pass, one block per 32-row chunk: largest |element| of the chunk's Q rows and of its C rows
// -> amax[matrix][chunk] (the operand scales), and the diagonal scores diag[i] = q_i . c_i in f32 (pass Q's optimistic
// exponent reference: the positive pair is the score most likely to dominate its row).
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep2h_kernel(RowSrc X0, RowSrc X1, float* __restrict__ amax,
                                                    float* __restrict__ diag) {
  __shared__ float red[8];
  const int t = threadIdx.x;
  const int64_t grow = (int64_t)blockIdx.x * 32 + (t >> 3);
  const int d0 = (t & 7) * 16;
  float mq = 0.f, mc = 0.f, dot = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 f = rowsrc_load4(X0, grow, d0 + 4 * q);
    const float4 g = rowsrc_load4(X1, grow, d0 + 4 * q);
    mq = fmaxf(mq, fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w))));
    mc = fmaxf(mc, fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))));
    dot = fmaf(f.x, g.x, fmaf(f.y, g.y, fmaf(f.z, g.z, fmaf(f.w, g.w, dot))));
  }
  dot += __shfl_xor(dot, 1, 64);
  dot += __shfl_xor(dot, 2, 64);
  dot += __shfl_xor(dot, 4, 64);
  if ((t & 7) == 0) diag[grow] = dot;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mq = fmaxf(mq, __shfl_xor(mq, o, 64));
    mc = fmaxf(mc, __shfl_xor(mc, o, 64));
  }
  if ((t & 63) == 0) { red[t >> 6] = mq; red[4 + (t >> 6)] = mc; }
  __syncthreads();
  if (t == 0) {
    amax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    amax[gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  }
}

// exponent e with amax * 2^e in [2^13, 2^14) (0 for an all-zero or non-finite matrix)
__device__ __forceinline__ int scale_exp(float amax) {
  if (!(amax > 0.f) || !(amax < INFINITY)) return 0;
  int x;
  frexpf(amax, &x);  // amax = m 2^x, m in [0.5, 1)
  const int e = 14 - x;
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// split pre-pass: one 256-thread block per 32-row chunk of one matrix (blockIdx.y selects Q or C): two row-major fp16
// planes of x * 2^e, the largest squared row norm per wave (unscaled, as in split3_kernel), and -- block (0, 0) -- the
// scale block the later kernels read:  sc[0] = 2^-(eq + ec) (S' -> S),  sc[1] = 2^-ec (pass Q's O'),  sc[2] = 2^-(eq + 14)
// (pass C's O'), sc[3] / sc[4] = 2^eq / 2^ec.
__global__ __launch_bounds__(256) void split2h_kernel(RowSrc X0, RowSrc X1, int64_t B, _Float16* __restrict__ R0,
                                                     _Float16* __restrict__ R1, const float* __restrict__ amax,
                                                     float* __restrict__ nrm, float* __restrict__ sc,
                                                     unsigned long long* __restrict__ loss_acc,
                                                     int* __restrict__ flags, int nflags) {
  __shared__ float red[8];
  if (blockIdx.x == 0 && blockIdx.y == 1)  // pass Q's "redo this block" flags
    for (int i = threadIdx.x; i < nflags; i += 256) flags[i] = 0;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x <= kLossWords) {  // see inbatch3_merge_kernel
    loss_acc[threadIdx.x * 16] = 0ull;
    if (threadIdx.x == 0) loss_acc[8] = 0ull;  // poison word
  }
  const int t = threadIdx.x, nchunks = gridDim.x;
  float mq = 0.f, mc = 0.f;
  for (int i = t; i < nchunks; i += 256) {
    mq = fmaxf(mq, amax[i]);
    mc = fmaxf(mc, amax[nchunks + i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mq = fmaxf(mq, __shfl_xor(mq, o, 64));
    mc = fmaxf(mc, __shfl_xor(mc, o, 64));
  }
  if ((t & 63) == 0) { red[t >> 6] = mq; red[4 + (t >> 6)] = mc; }
  __syncthreads();
  mq = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  mc = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  const int eq = scale_exp(mq), ec = scale_exp(mc);
  if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) {
    sc[0] = ldexpf(1.f, -(eq + ec));
    sc[1] = ldexpf(1.f, -ec);
    sc[2] = ldexpf(1.f, -(eq + (int)kHPexp));
    sc[3] = ldexpf(1.f, eq);
    sc[4] = ldexpf(1.f, ec);
  }
  const RowSrc X = blockIdx.y ? X1 : X0;
  _Float16* R = blockIdx.y ? R1 : R0;
  const float mul = ldexpf(1.f, blockIdx.y ? ec : eq);
  const int chunk = blockIdx.x;
  const int row = t >> 3, d0 = (t & 7) * 16;
  const int64_t grow = (int64_t)chunk * 32 + row;
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 f = rowsrc_load4(X, grow, d0 + 4 * q);
    v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
  }
  {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) ss = fmaf(v[e], v[e], ss);
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    ss = fmaxf(ss, __shfl_xor(ss, 8, 64));
    ss = fmaxf(ss, __shfl_xor(ss, 16, 64));
    ss = fmaxf(ss, __shfl_xor(ss, 32, 64));
    if ((t & 63) == 0) nrm[((int64_t)blockIdx.y * gridDim.x + chunk) * 4 + (t >> 6)] = ss;
  }
  f16x8 p[2][2];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float xs = v[e] * mul;  // exact (power of two)
    const _Float16 a = (_Float16)xs;
    const _Float16 b = (_Float16)(xs - (float)a);
    p[0][e >> 3][e & 7] = a;
    p[1][e >> 3][e & 7] = b;
  }
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    f16x8* dst = reinterpret_cast<f16x8*>(R + ((int64_t)pl * B + grow) * k3D + d0);
    dst[0] = p[pl][0];
    dst[1] = p[pl][1];
  }
}

// -----------------------------------------------------------------------------------------------------------------
// prep + split in ONE launch (round 4; the two launches above read the same 8 MB twice, 6.6 + 8.0 us at B = 8192).  One
// block per 32-row chunk holds its Q rows AND its C rows in registers; the only thing a block needs from the others is
// the binade of the largest |element| of each matrix (scale_exp uses nothing else), and it gets it without a
// zero-initialised word, a fence or a second launch: every block PUBLISHES one 64-bit word
//     (token << 16) | code(max |q| of the chunk) << 8 | code(max |c| of the chunk)
// with an agent-scope (write-through) store, where `token` is 48 bits the host draws per call -- a word left by an
// earlier call, or whatever a fresh workspace holds, carries this call's token with probability 2^-48 -- and then every
// thread polls the word of one chunk with agent-scope loads.  One word, so there is no ordering between two stores to
// rely on.  All blocks of the grid are resident (B / 32 <= 512 blocks of 256 threads on 256 CUs), so the poll ends when
// the slowest block has published; it is BOUNDED anyway (kPollTicks of the 100 MHz clock): on a timeout the loss is
// poisoned (NaN) rather than the queue hung.  code: 0 = all zero (or NaN: fmaxf drops it, as in split2h_kernel; a NaN
// element shows up in the planes themselves), 255 = infinite, else the frexp exponent + 127.
// The grid also zeroes the words the merge launches count in.
// -----------------------------------------------------------------------------------------------------------------
constexpr unsigned long long kPollTicks = 300000000ull;  // 3 s
__device__ __forceinline__ unsigned amax_code(float amax) {
  if (!(amax > 0.f)) return 0u;
  if (!(amax < INFINITY)) return 255u;
  int x;
  frexpf(amax, &x);
  return (unsigned)min(max(x + 127, 1), 254);
}
__device__ __forceinline__ int scale_exp_code(unsigned code) {  // scale_exp of any value with that code
  if (code == 0u || code == 255u) return 0;
  const int e = 14 - ((int)code - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

__global__ __launch_bounds__(256) void prepsplit2h_kernel(RowSrc X0, RowSrc X1, int64_t B, _Float16* __restrict__ R0,
                                                         _Float16* __restrict__ R1,
                                                         unsigned long long* __restrict__ ent,
                                                         unsigned long long token, float* __restrict__ diag,
                                                         float* __restrict__ nrm, float* __restrict__ sc,
                                                         unsigned long long* __restrict__ loss_acc,
                                                         int* __restrict__ flags, int nflags,
                                                         float* __restrict__ copy0, float* __restrict__ copy1,
                                                         unsigned* __restrict__ czero, int two_level) {
  // copy0 / copy1 (optional): the gathered rows as dense f32 [B, ld] matrices.  The overlapped train step updates a tower
  // while the merge launch of the OTHER side still needs that tower's old rows: the merges then read these copies
  __shared__ float red[8];
  __shared__ unsigned cred[8];
  const int t = threadIdx.x, chunk = blockIdx.x, nchunks = gridDim.x;
  if (chunk == 0 && t <= kLossWords) {  // see inbatch3_merge_kernel
    loss_acc[t * 16] = 0ull;
    if (t == 0) loss_acc[8] = 0ull;  // poison word
  }
  if (chunk == 1 || nchunks == 1)
    for (int i = t; i < nflags; i += 256) flags[i] = 0;
  // czero: the pass-Q splits' flags of pass C's form (facscale2h_kernel), 16 words
  if (czero && chunk == 0 && t < 16) czero[t] = 0u;
  const int row = t >> 3, d0 = (t & 7) * 16;
  const int64_t grow = (int64_t)chunk * 32 + row;
  float vq[16], vc[16];
  {
    // (rowsrc_load4 per quad re-reads the row's id and branches on the source kind every time: eight dependent
    // id -> row round trips in a row; here the two ids are read once and the eight row loads are issued together)
    const int64_t rq = X0.idx ? (int64_t)X0.idx[grow] : grow;
    const int64_t rc = X1.idx ? (int64_t)X1.idx[grow] : grow;
    RowSrc Y0 = X0, Y1 = X1;
    Y0.idx = nullptr; Y1.idx = nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 f = rowsrc_load4(Y0, rq, d0 + 4 * q);
      const float4 g = rowsrc_load4(Y1, rc, d0 + 4 * q);
      vq[4 * q] = f.x; vq[4 * q + 1] = f.y; vq[4 * q + 2] = f.z; vq[4 * q + 3] = f.w;
      vc[4 * q] = g.x; vc[4 * q + 1] = g.y; vc[4 * q + 2] = g.z; vc[4 * q + 3] = g.w;
      if (copy0 && d0 + 4 * q < X0.ld) *reinterpret_cast<float4*>(copy0 + grow * X0.ld + d0 + 4 * q) = f;
      if (copy1 && d0 + 4 * q < X1.ld) *reinterpret_cast<float4*>(copy1 + grow * X1.ld + d0 + 4 * q) = g;
    }
  }
  float mq = 0.f, mc = 0.f, dot = 0.f, ssq = 0.f, ssc = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    mq = fmaxf(mq, fabsf(vq[e]));
    mc = fmaxf(mc, fabsf(vc[e]));
    dot = fmaf(vq[e], vc[e], dot);
    ssq = fmaf(vq[e], vq[e], ssq);
    ssc = fmaf(vc[e], vc[e], ssc);
  }
  dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64);
  ssq += __shfl_xor(ssq, 1, 64); ssq += __shfl_xor(ssq, 2, 64); ssq += __shfl_xor(ssq, 4, 64);
  ssc += __shfl_xor(ssc, 1, 64); ssc += __shfl_xor(ssc, 2, 64); ssc += __shfl_xor(ssc, 4, 64);
  if ((t & 7) == 0) diag[grow] = dot;
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    ssq = fmaxf(ssq, __shfl_xor(ssq, o, 64));
    ssc = fmaxf(ssc, __shfl_xor(ssc, o, 64));
  }
  if ((t & 63) == 0) {  // largest squared row norm per wave, as split2h_kernel leaves it (the row-max pass reads it)
    nrm[((int64_t)0 * nchunks + chunk) * 4 + (t >> 6)] = ssq;
    nrm[((int64_t)1 * nchunks + chunk) * 4 + (t >> 6)] = ssc;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mq = fmaxf(mq, __shfl_xor(mq, o, 64));
    mc = fmaxf(mc, __shfl_xor(mc, o, 64));
  }
  if ((t & 63) == 0) { red[t >> 6] = mq; red[4 + (t >> 6)] = mc; }
  __syncthreads();
  if (t == 0) {
    mq = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    mc = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    __hip_atomic_store(&ent[chunk], (token << 16) | (unsigned long long)(amax_code(mq) << 8 | amax_code(mc)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned cq = 0u, cc = 0u;
  bool timed_out = false;
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  // Two levels (round 6): workgroup 0 gathers the chunks' words -- one poller per word -- and publishes the two maxima in
  // ONE more tagged word, which is all the other workgroups poll (one lane each).  With every workgroup polling every
  // word, 65 536 agent-scope requests a round queued on 32 lines at the memory side: the poll was 5 of the kernel's 16 us.
  const bool gatherer = chunk == 0 || !two_level;
  for (int i = gatherer ? t : (t == 0 ? nchunks : nchunks + 1); i < (gatherer ? nchunks : nchunks + 1); i += 256) {
    unsigned long long v;
    while (((v = __hip_atomic_load(&ent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 16) != token) {
      if (__builtin_amdgcn_s_memrealtime() - t_start > kPollTicks) { timed_out = true; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    cq = max(cq, (unsigned)(v >> 8) & 255u);
    cc = max(cc, (unsigned)v & 255u);
  }
  if (timed_out) atomicOr(reinterpret_cast<unsigned*>(loss_acc + 8), 1u);  // poison: the loss comes out NaN
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cq = max(cq, (unsigned)__shfl_xor((int)cq, o, 64));
    cc = max(cc, (unsigned)__shfl_xor((int)cc, o, 64));
  }
  if ((t & 63) == 0) { cred[t >> 6] = cq; cred[4 + (t >> 6)] = cc; }
  __syncthreads();
  cq = max(max(cred[0], cred[1]), max(cred[2], cred[3]));
  cc = max(max(cred[4], cred[5]), max(cred[6], cred[7]));
  if (two_level && chunk == 0 && t == 0)
    __hip_atomic_store(&ent[nchunks], (token << 16) | (unsigned long long)(cq << 8 | cc), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  const int eq = scale_exp_code(cq), ec = scale_exp_code(cc);
  if (chunk == 0 && t == 0) {
    sc[0] = ldexpf(1.f, -(eq + ec));
    sc[1] = ldexpf(1.f, -ec);
    sc[2] = ldexpf(1.f, -(eq + (int)kHPexp));
    sc[3] = ldexpf(1.f, eq);
    sc[4] = ldexpf(1.f, ec);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float mul = ldexpf(1.f, m ? ec : eq);
    _Float16* R = m ? R1 : R0;
    f16x8 p[2][2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float xs = (m ? vc[e] : vq[e]) * mul;  // exact (power of two)
      const _Float16 a = (_Float16)xs;
      const _Float16 b = (_Float16)(xs - (float)a);
      p[0][e >> 3][e & 7] = a;
      p[1][e >> 3][e & 7] = b;
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      f16x8* dst = reinterpret_cast<f16x8*>(R + ((int64_t)pl * B + grow) * k3D + d0);
      dst[0] = p[pl][0];
      dst[1] = p[pl][1];
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------
// Row reference for pass Q: part_mr[split][row] = M with p' = exp2(s sl2 - M) <= 2^14 (+ the hi-plane product's error).
// bound <= kHBoundSafe: M = bound - 14 (no GEMM).  Otherwise the row maximum of the hi-plane product (error <= 2^-10
// |q||c| sl2 in log2 units: fp16's range above 2^14 absorbs it up to bounds of ~1000; beyond that -- scores whose exp
// over- or underflows f32 anyway -- the second-order planes are added to the product).
//
// The GEMM is 1 / 9 of the main kernels' MFMA work and all about moving the streamed hi plane: each WAVE owns 64 rows
// (two B operands per A fragment read from LDS), a workgroup 256, so the plane is streamed B / 256 times (67 MB at
// B = 8192); the tiles come in GROUPS of kHRmGroup chunks through a ring of kHRmRing slots with kHRmRing - 1 groups in
// flight (LDS-DMAs complete in order: s_waitcnt vmcnt(<DMAs of the younger groups>) waits for the oldest group only).
// With one group in flight the kernel took 40 us: the DMA round trip (~2 us under 256 workgroups reading the same
// lines) was exposed once per group.
// -----------------------------------------------------------------------------------------------------------------
constexpr int kHRmGroup = 4, kHRmRing = 4, kHRmOwned = 256;
template <bool FULL>
__device__ __forceinline__ float rowmax2h_sweep(const char* __restrict__ baseY, char* lds, int64_t B, int64_t c0, int nc,
                                                const f16x8 (&bx0)[2][8], const f16x8 (&bx1)[2][8], int t, int w, int j,
                                                int h) {
  // (the three-term variant also streams plane 2: half as many chunks per group, same slot size and DMA count)
  constexpr int kG = FULL ? kHRmGroup / 2 : kHRmGroup;
  constexpr int kSlotChunk = (FULL ? 2 : 1) * kPlaneBytes, kSlot = kG * kSlotChunk;
  constexpr int kDmaPerGroup = kG * (FULL ? 4 : 2);
  const uint32_t g0 = dmah_off0<0>(B, c0, t), g1 = dmah_off0<1>(B, c0, t);
  const uint32_t g2 = dmah_off0<2>(B, c0, t), g3 = dmah_off0<3>(B, c0, t);
  const int ngroups = (nc + kG - 1) / kG;
  // group GI -> ring slot GI % kHRmRing; chunks past the end re-fetch the last one (the DMA count per group stays
  // constant, which is what the partial vmcnt wait counts on)
#define H_RM_FETCH(GI)                                                                       \
  _Pragma("unroll") for (int k = 0; k < kG; ++k) {                                          \
    const int ch_ = min((GI) * kG + k, nc - 1);                                             \
    char* dst_ = lds + ((GI) % kHRmRing) * kSlot + k * kSlotChunk;                          \
    const uint32_t adv_ = (uint32_t)ch_ * 8192u;                                            \
    H_DP(0, g0 + adv_, dst_); H_DP(1, g1 + adv_, dst_);                                     \
    if (FULL) { H_DP(2, g2 + adv_, dst_); H_DP(3, g3 + adv_, dst_); }                       \
  }
#pragma unroll
  for (int gi = 0; gi < kHRmRing - 1; ++gi) { H_RM_FETCH(gi); }
  float ma = -INFINITY, mb = -INFINITY;  // running maxima of this lane's two owned rows over ITS half of the streamed rows
  for (int gi = 0; gi < ngroups; ++gi) {
    // group gi has landed (the kHRmRing - 2 younger groups may still be in flight) and everyone is done with the slot
    // group gi + kHRmRing - 1 goes to (it held group gi - 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kHRmRing - 2) * kDmaPerGroup) : "memory");
    __syncthreads();
    H_RM_FETCH(gi + kHRmRing - 1);
#pragma unroll
    for (int k = 0; k < kG; ++k) {
      if (gi * kG + k < nc) {
        const char* buf = lds + (gi % kHRmRing) * kSlot + k * kSlotChunk;
        f32x16 sa = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 sb = sa;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int off = j * 256 + (((2 * s + h) ^ swz16(j)) << 4);
          const f16x8 a1 = *reinterpret_cast<const f16x8*>(buf + off);
          if (FULL) {
            const f16x8 a2 = *reinterpret_cast<const f16x8*>(buf + off + kPlaneBytes);
            sa = H_MFMA(a2, bx0[0][s], sa);
            sb = H_MFMA(a2, bx0[1][s], sb);
            sa = H_MFMA(a1, bx1[0][s], sa);
            sb = H_MFMA(a1, bx1[1][s], sb);
          }
          sa = H_MFMA(a1, bx0[0][s], sa);
          sb = H_MFMA(a1, bx0[1][s], sb);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { ma = fmaxf(ma, sa[r]); mb = fmaxf(mb, sb[r]); }
      }
    }
  }
#undef H_RM_FETCH
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped re-fetches of the tail
  // lane (j, h) holds rows j (ma) and 32 + j (mb) of the wave's 64 against the streamed rows of half h: lane h = 0
  // reports row j, lane h = 1 row 32 + j, each taking the other half's value across
  const float mine = h == 0 ? ma : mb, other = h == 0 ? mb : ma;
  return fmaxf(mine, __shfl_xor(other, 32, 64));
}

__global__ __launch_bounds__(256) void rowmax2h_kernel(const _Float16* __restrict__ Xr, const _Float16* __restrict__ Yr,
                                                      int64_t B, int nsplit, float sl2, const float* __restrict__ nrm,
                                                      const float* __restrict__ sc, float* __restrict__ part_mr) {
  __shared__ __attribute__((aligned(16))) char lds[kHRmRing * kHRmGroup * kPlaneBytes];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int ob = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
  // this lane reports row xrow = block base + 64 w + 32 h + j
  const int64_t wrow = (int64_t)ob * kHRmOwned + w * 64;
  const int64_t xrow = wrow + 32 * h + j;
  const bool live = wrow < B;  // B is a multiple of 128: the last block's upper two waves may own nothing
  float bound;
  {
    const int nslots = (int)(B / k3Chunk) * 4;  // per matrix
    float mq = 0.f, mc = 0.f;
    for (int i = t; i < nslots; i += 256) {
      mq = fmaxf(mq, nrm[i]);
      mc = fmaxf(mc, nrm[nslots + i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mq = fmaxf(mq, __shfl_xor(mq, o, 64));
      mc = fmaxf(mc, __shfl_xor(mc, o, 64));
    }
    float* red = reinterpret_cast<float*>(lds);
    if (lane == 0) { red[w] = mq; red[4 + w] = mc; }
    __syncthreads();
    mq = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    mc = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    __syncthreads();  // the slow path reuses lds as the DMA ring
    bound = sqrtf(mq * mc) * fabsf(sl2);
    if (bound <= kHBoundSafe) {
      if (live) part_mr[(int64_t)split * B + xrow] = bound - kHPexp;
      return;
    }
  }
  const bool full = !(bound <= 1024.f);  // absurd score ranges: all three terms (workgroup-uniform)
  const float sl2s = sl2 * sc[0];
  const int nc = (int)(B / k3Chunk) / nsplit;
  const int64_t c0 = (int64_t)split * nc;
  const char* const baseY = reinterpret_cast<const char*>(Yr);
  f16x8 bx0[2][8], bx1[2][8];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int64_t r = live ? wrow + 32 * u + j : 0;
      bx0[u][s] = *reinterpret_cast<const f16x8*>(Xr + r * k3D + 16 * s + 8 * h);
      bx1[u][s] = *reinterpret_cast<const f16x8*>(Xr + ((int64_t)B + r) * k3D + 16 * s + 8 * h);
      // a negative temperature turns the maximum of s sl2 into the minimum of s
      if (sl2 < 0.f) { bx0[u][s] = -bx0[u][s]; bx1[u][s] = -bx1[u][s]; }
    }
  // (plane 1 of the streamed matrix is only fetched -- and the ring only holds it -- in the three-term variant; the
  // ring is sized for the one-term variant and the three-term one runs with half-size groups)
  const float m = full ? rowmax2h_sweep<true>(baseY, lds, B, c0, nc, bx0, bx1, t, w, j, h)
                       : rowmax2h_sweep<false>(baseY, lds, B, c0, nc, bx0, bx1, t, w, j, h);
  if (live) part_mr[(int64_t)split * B + xrow] = m * fabsf(sl2s) - kHPexp;
}

// -----------------------------------------------------------------------------------------------------------------
// Pass Q: owned = Q rows (B operand of both products, 64 VGPRs), streamed = C rows in 32-row chunks through a 3-deep
// LDS ring.  Per chunk and wave: 24 MFMAs of S^T (k = d) + 24 of O^T (k = streamed row), software-pipelined exactly as
// inbatch3_kernel: the S^T MFMAs of chunk t + 1 carry the exp / two-plane split / P store of chunk t between them, the
// O^T MFMAs of chunk t carry the second k-step's A fragments and the DMA of chunk t + 2.  The unnormalised
// probabilities p' = exp2(s sl2 - M_i) go to Pmat (f32, 32 x 32 tiles, transposed) for pass C.
// -----------------------------------------------------------------------------------------------------------------
// The A fragments of the S^T phase are read with ds_read_b128 written as inline assembly and waited for by hand: hipcc's
// own s_waitcnt does not count the transposing reads (also assembly) that are issued between a fragment pair and its
// use, and came out as lgkmcnt(2) / lgkmcnt(0) on alternating k-steps -- every other k-step waited for the pair it had
// JUST issued, one exposed LDS latency each.  LDS operations return in order; before the first MFMA of k-step s the
// outstanding ones are, oldest first: pair s (issued a k-step ago), the two transposing reads of k-step s - 1 (VALU_ON,
// s >= 1) and pair s + 1 (s < 7): the wait lets everything but pair s stay in flight.
#ifndef H_S_ASM_LOADS
#define H_S_ASM_LOADS 1
#endif
template <int OFF>
__device__ __forceinline__ f16x8 lds_b128(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
#if H_S_ASM_LOADS
#define H_S_LD(P, OFFB) lds_b128<(P) * kPlaneBytes>(ap32_ + (uint32_t)(OFFB))
#define H_S_WAIT(N) { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#else
#define H_S_LD(P, OFFB) (*reinterpret_cast<const f16x8*>(ap0_ + (OFFB) + (P) * kPlaneBytes))
#define H_S_WAIT(N)
#endif
#define H_S_PHASE(NBUF, SA, VALU_ON)                                                                      \
  {                                                                                                       \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) SA[r_] = 0.f;                                       \
    const char* ap0_ = (NBUF) + j * 256;                                                                  \
    const uint32_t ap32_ = lds32 + (uint32_t)((NBUF) - lds) + (uint32_t)(j * 256);                        \
    (void)ap0_; (void)ap32_;                                                                              \
    const int sw_ = swz16(j);                                                                             \
    f16x8 a1_ = H_S_LD(0, (h ^ sw_) << 4);                                                                \
    f16x8 a2_ = H_S_LD(1, (h ^ sw_) << 4);                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                                    \
      f16x8 n1_ = a1_, n2_ = a2_;                                                                         \
      if (s_ < 7) {                                                                                       \
        const int off_ = (((2 * (s_ + 1) + h) ^ sw_) << 4);                                               \
        n1_ = H_S_LD(0, off_);                                                                            \
        n2_ = H_S_LD(1, off_);                                                                            \
      }                                                                                                   \
      float e0_ = 0.f, e1_ = 0.f;                                                                         \
      f16x2 pa_ = {0, 0};                                                                                 \
      H_SB();                                                                                             \
      H_S_WAIT((s_ < 7 ? 2 : 0) + (((VALU_ON) && s_ >= 1) ? 2 : 0));                                      \
      H_SB();                                                                                             \
      SA = H_MFMA(a2_, bx[0][s_], SA);                                                                    \
      H_SB();                                                                                             \
      if (VALU_ON) {                                                                                      \
        const f32x2 arg_ = pk_fma(f32x2{p[2 * s_], p[2 * s_ + 1]}, sl2v, nrefv);                          \
        e0_ = __builtin_amdgcn_exp2f(arg_[0]);                                                            \
        e1_ = __builtin_amdgcn_exp2f(arg_[1]);                                                            \
        trh_frag_n<0>(s_, ta2_, trc_); /* one of the 8 G = 0 fragments of the coming O^T phase */          \
      }                                                                                                   \
      H_SB();                                                                                             \
      SA = H_MFMA(a1_, bx[1][s_], SA);                                                                    \
      H_SB();                                                                                             \
      if (VALU_ON) {                                                                                      \
        pa_ = pk_f16(e0_, e1_);                                                                           \
        l2 = pk_add(l2, f32x2{e0_, e1_});                                                                 \
        emax = __builtin_fmaxf(emax, __builtin_fmaxf(e0_, e1_)); /* v_max3_f32 */                          \
        /* the lo-plane piece of k-steps 0..3, complete since the previous k-step: see H_P_ST */            \
        if (s_ == 4) { H_P_ST(1, pw[1][0], pw[1][1], pw[1][2], pw[1][3]); }                                 \
      }                                                                                                   \
      H_SB();                                                                                             \
      SA = H_MFMA(a1_, bx[0][s_], SA);                                                                    \
      H_SB();                                                                                             \
      if (VALU_ON) {                                                                                      \
        const f16x2 pq_ = pk_f16(resid_lo(e0_, pa_), resid_hi(e1_, pa_));                                 \
        pw[0][s_] = __builtin_bit_cast(uint32_t, pa_);                                                    \
        pw[1][s_] = __builtin_bit_cast(uint32_t, pq_);                                                    \
        if (s_ == 3) { H_P_ST(0, pw[0][0], pw[0][1], pw[0][2], pw[0][3]); }                                 \
        if (s_ == 7) {                                                                                    \
          H_P_ST(2, pw[0][4], pw[0][5], pw[0][6], pw[0][7]);                                                \
          H_P_ST(3, pw[1][4], pw[1][5], pw[1][6], pw[1][7]);                                                \
        }                                                                                                 \
      }                                                                                                   \
      H_SB();                                                                                             \
      a1_ = n1_; a2_ = n2_;                                                                               \
    }                                                                                                     \
    if (VALU_ON) pst_u += nch * 4096;                                                                     \
  }
// O^T += Y_chunk^T P^T (see ESR_O_ROW): rows in the order small terms first
#if defined(H_PROBE_NOMFMA)  /* timing probe only: the O^T phase without its MFMAs (what the data movement alone costs) */
#define H_O_ROW(PL_A, PL_P, G)                                                                            \
  _Pragma("unroll") for (int db_ = 0; db_ < 4; ++db_) {                                                   \
    acc[db_][0] += (float)ta2_[G][db_][PL_A][0] * (float)pb[PL_P][G][0];                                  \
  }
#else
#define H_O_ROW(PL_A, PL_P, G)                                                                            \
  _Pragma("unroll") for (int db_ = 0; db_ < 4; ++db_)                                                     \
    acc[db_] = H_MFMA(ta2_[G][db_][PL_A], pb[PL_P][G], acc[db_]);
#endif
#if defined(H_PROBE_PC_NOFRAG)
#define H_O_G1(F0, F1)
#else
#define H_O_G1(F0, F1) { _Pragma("unroll") for (int f_ = (F0); f_ < (F1); ++f_) trh_frag_n<1>(f_, ta2_, trc_); }
#endif
#define H_PB()                                                                                            \
  f16x8 pb[2][2];                                                                                         \
  _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_)                                                        \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                                    \
      const u32x4 u_ = {pw[q_][4 * g_], pw[q_][4 * g_ + 1], pw[q_][4 * g_ + 2], pw[q_][4 * g_ + 3]};      \
      pb[q_][g_] = __builtin_bit_cast(f16x8, u_);                                                         \
    }
#define H_O_PHASE(DMA_ON, DBUF)                                                                           \
  {                                                                                                       \
    H_PB();                                                                                               \
    H_TR_WAIT(); /* the G = 0 fragments were requested during the S^T phase (or by the burst below) */    \
    H_SB(); H_O_ROW(1, 0, 0); H_SB(); H_O_G1(0, 3); if (DMA_ON) { H_DP(0, g0, DBUF); }                    \
    H_SB(); H_O_ROW(0, 1, 0); H_SB(); H_O_G1(3, 6); if (DMA_ON) { H_DP(1, g1, DBUF); }                    \
    H_SB(); H_O_ROW(0, 0, 0); H_SB(); H_O_G1(6, 8); if (DMA_ON) { H_DP(2, g2, DBUF); }                    \
    H_TR_WAIT();                                                                                          \
    H_SB(); H_O_ROW(1, 0, 1); H_SB(); if (DMA_ON) { H_DP(3, g3, DBUF); }                                  \
    H_SB(); H_O_ROW(0, 1, 1); H_SB();                                                                     \
    H_SB(); H_O_ROW(0, 0, 1); H_SB();                                                                     \
    if (DMA_ON) H_DMA_ADVANCE();                                                                          \
  }
#define H_TR_BASES(BUF)                                                                                   \
  {                                                                                                       \
    const uint32_t slot_ = (uint32_t)((BUF) - lds);                                                       \
    _Pragma("unroll") for (int db_ = 0; db_ < 4; ++db_) { trc_[db_][0] = trb_[db_][0] + slot_; trc_[db_][1] = trb_[db_][1] + slot_; } \
  }
#define H_TR_SETUP()                                                                                      \
  const int tr_a = (lane & 15) >> 2;                                                                      \
  const int tr_e = (2 * ((lane >> 4) & 1)) + ((lane & 3) >> 1), tr_low = (lane & 1) * 8;                  \
  const int tr_row0 = (4 * h + tr_a) * 256, tr_row1 = (4 * h + 8 + tr_a) * 256;                           \
  const int tr_l0 = ((tr_e ^ (h & 3)) << 4) | tr_low, tr_l1 = ((tr_e ^ ((h + 2) & 3)) << 4) | tr_low;     \
  uint32_t trb_[4][2], trc_[4][2];                                                                        \
  _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                                      \
    trb_[db][0] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds + tr_row0 + (((db ^ tr_a) << 6) | tr_l0); \
    trb_[db][1] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds + tr_row1 + (((db ^ tr_a) << 6) | tr_l1); \
  }

// The exponent reference M of a (128-row block, split) workgroup, `mode`:
//   0  from the row-max pass (part_mr, its own splits): exact, costs that pass (28 us at B = 8192);
//   1  OPTIMISTIC: M = max(diagonal score, largest score of the workgroup's first chunk) - 4.  Exact whenever no score
//      of the workgroup's range exceeds that reference by 12 binades (a candidate 4000 x more probable than both the
//      positive pair and the best of 32 others); the kernel tracks its largest probability, and a workgroup that saw
//      one beyond fp16's range raises flags[blockIdx.x];
//   FIX (second launch of the same grid, one load and exit for unflagged workgroups): the flagged ones sweep their
//      range for the exact score maximum and run again.  Nothing downstream depends on which launch produced a block:
//      merge<Q> combines splits with different references, and pass C takes its factors per (row, split).
//   KIND 2 (round 4, the default): ONE launch.  A workgroup that saw a probability leave fp16's range redoes itself on the
//      spot (the decision is workgroup-wide: the four waves share the ring and its barriers) -- no flags, no redo launch
//      (6 us for 512 workgroups that load a flag and exit).  [Merging the partials in this kernel's epilogue as well --
//      the nsplit workgroups of a block waiting for each other and merging a slice each, partials written through and
//      read at agent scope -- was built and measured in round 4: correct, but 128 us against 102 + 14 for the sweep and
//      the merge launch (two dependent memory round trips and three atomics per slice at the tail of every workgroup;
//      with cache-wide release / acquire fences instead of per-access scopes 186 us).  DESIGN_HISTORY.md.]
template <int KIND>
__global__ ESR_NO_PK __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void inbatch2h_q_kernel(const _Float16* __restrict__ Xr, const _Float16* __restrict__ Yr,
                                                         int64_t B, int nsplit, float sl2_in,
                                                         const float* __restrict__ sc, const float* __restrict__ ref,
                                                         int nsplit_ref, const float* __restrict__ diag, int mode,
                                                         int* __restrict__ flags, float* __restrict__ part_m,
                                                         float* __restrict__ part_O, float* __restrict__ part_l,
                                                         float* __restrict__ Pmat) {
  __shared__ __attribute__((aligned(16))) char lds[kHBufs * kHBufBytes];
  if (KIND == 1 && flags[blockIdx.x] == 0) return;
  H_TIMING_DECL();
  int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  int j = lane & 31, h = lane >> 5;
  H_TR_SETUP();
  const uint32_t lds32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
  (void)lds32;
  const int ob = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
  const int64_t xrow = (int64_t)ob * k3Owned + w * 32 + j;
  const int nc = (int)(B / k3Chunk) / nsplit;
  const int64_t c0 = (int64_t)split * nc;
  const int64_t nch = B / 32;
  const float sl2 = sl2_in * sc[0];  // the planes carry 2^(eq + ec) S

  // The probabilities leave for pass C as the two fp16 planes this kernel forms for its own O^T product (round 6; until
  // then: the f32 values, multiplied by the row's factor and split again by pass C's VALU).  Lane (a = owned row, h) holds
  // in pw[plane][4 m' .. 4 m' + 3] the plane's values of streamed rows 16 m' + 8 b + 4 h + c (b = 0, 1, c = 0..3): one
  // 16-byte piece.  A 32 x 32 tile (streamed chunk jt, owned block it; 4 KB at (jt * B/32 + it) * 4096) is stored as four
  // 1 KB blocks M = 2 m' + plane, the piece of lane (a, h) at position (a / 8) * 16 + (2 h + a / 4 % 2) * 4 + a % 4 of its
  // block: FOUR fully coalesced 1 KB stores per chunk and wave (every aligned quad of lanes writes one 64-byte segment).
  // Pass C copies the blocks to LDS as they are (contiguous 1 KB requests) and takes its MFMA operand from them with the
  // transposing read: the position formula is what makes those reads bank-conflict-free (inbatch2h_pct_kernel).
  char* pst_u = nullptr;
  const uint32_t pst_v = (uint32_t)((((j >> 3) * 16) + ((2 * h + ((j >> 2) & 1)) * 4) + (j & 3)) * 16);
#if defined(H_PROBE_Q_NOSTORE)  /* timing probe only: pass Q without its P stores */
#define H_P_ST(M, V0, V1, V2, V3)
#else
// streaming (non-temporal) stores: the B x B probabilities are written once and read once, and at 268 MB (B = 8192) do
// not fit the 256 MB Infinity Cache anyway
#define H_P_ST(M, V0, V1, V2, V3) \
  __builtin_nontemporal_store(u32x4{(V0), (V1), (V2), (V3)}, reinterpret_cast<u32x4*>(pst_u + (M) * 1024 + pst_v))
#endif

  f32x16 acc[4];
  f32x2 l2 = {0.f, 0.f};  // even / odd score pairs; added in a fixed order at the end
  int dpos = 0;
  const char* const baseY = reinterpret_cast<const char*>(Yr);
  uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  f16x8 bx[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      bx[p][s] = *reinterpret_cast<const f16x8*>(Xr + ((int64_t)p * B + xrow) * k3D + 16 * s + 8 * h);
  f32x16 sa;
  float p[16];
  uint32_t pw[2][8];
  f16x8 ta2_[2][4][2];
  float emax = 0.f;
  float refv = -INFINITY;
  const f32x2 sl2v = {sl2, sl2};
  f32x2 nrefv = {0.f, 0.f};
  // One sweep of the workgroup's chunks as an inlined lambda: KIND 2 calls it a second time, as the redo (fix = true), when
  // the first sweep overflowed.  (Written as a loop around the sweep the redo cost 37 registers -- what the compiler
  // hoisted out of the loop stayed live through it -- and with them the second wave per SIMD.)
  auto sweep = [&](auto fix_tag) __attribute__((always_inline)) {
  constexpr bool fix = decltype(fix_tag)::value;
  pst_u = reinterpret_cast<char*>(Pmat) + (c0 * nch + (xrow >> 5)) * 4096;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
  l2 = f32x2{0.f, 0.f};
  dpos = 0;
  g0 = dmah_off0<0>(B, c0, t); g1 = dmah_off0<1>(B, c0, t); g2 = dmah_off0<2>(B, c0, t); g3 = dmah_off0<3>(B, c0, t);
  emax = 0.f;
  refv = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r] = 0.f;
  if (fix) {
    // exact maximum of s sl2 over this workgroup's chunks (one tile at a time; this path runs for flagged blocks only).
    // The DMA offsets wrap after nc chunks: the ring state is back at chunk 0 afterwards.
    float m = -INFINITY;
    for (int c = 0; c < nc; ++c) {
      H_DMA_CHUNK(lds);
      H_DMA_BARRIER();
      H_S_PHASE(lds, sa, false);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sa[r] * sl2);
      __syncthreads();  // the next tile overwrites this one
    }
    refv = fmaxf(m, __shfl_xor(m, 32, 64)) - kHPexp;
  }
  // The optimistic reference is the SAME for every split of a row (round 6): max(diagonal score, scores against chunk 0)
  // - 4 -- chunk 0 being split 0's first chunk, the other splits fetch it into the ring's third slot beside their own
  // first two.  (Until round 6 every split took its own first chunk: as robust, but the splits' references differed and
  // with them the factor a probability needs in pass C.  With one reference per row the factor is per row, and pass C
  // streams ONE scaled copy of Q: inbatch2h_pct_kernel.  A workgroup that redoes itself still gets its own reference:
  // facscale2h_kernel flags that split.)
  const bool sample = !fix && mode == 1 && c0 != 0;
  if (sample) {
    char* sbuf = lds + 2 * kHBufBytes;
    H_DP(0, dmah_off0<0>(B, 0, t), sbuf); H_DP(1, dmah_off0<1>(B, 0, t), sbuf);
    H_DP(2, dmah_off0<2>(B, 0, t), sbuf); H_DP(3, dmah_off0<3>(B, 0, t), sbuf);
  }
  H_DMA_CHUNK(lds);
  if (nc > 1) H_DMA_CHUNK(lds + kHBufBytes);
  if (!fix && mode == 0) {
    float rv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) rv[s] = s < nsplit_ref ? ref[(int64_t)s * B + xrow] : -INFINITY;
    refv = rv[0];
#pragma unroll
    for (int s = 1; s < 8; ++s) refv = fmaxf(refv, rv[s]);
  }
  const float dref = (!fix && mode == 1) ? diag[xrow] * sl2_in : -INFINITY;
  H_DMA_BARRIER();

  float msample = dref;
  if (sample) {
    H_S_PHASE(lds + 2 * kHBufBytes, sa, false);
#pragma unroll
    for (int r = 0; r < 16; ++r) msample = fmaxf(msample, sa[r] * sl2);
  }
  H_S_PHASE(lds, sa, false);
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r] = sa[r];
  if (!fix && mode == 1) {
    float m = msample;
    if (!sample) {
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sa[r] * sl2);
    }
    refv = fmaxf(m, __shfl_xor(m, 32, 64)) - kHOptHead;
  }
  if (h == 0) part_m[(int64_t)split * B + xrow] = refv;  // what merge<Q> adds back
  nrefv = f32x2{-refv, -refv};

  H_TIMING_START();
  int cur = 0;
  for (int it = 0; it + 2 < nc; ++it) {
    const int nxt = cur == kHBufs - 1 ? 0 : cur + 1;
    const int nn = nxt == kHBufs - 1 ? 0 : nxt + 1;
    H_TICK(tk0);
    H_DMA_BARRIER();
    H_TICK(tk1);
    const char* buf = lds + cur * kHBufBytes;
    const char* nbuf = lds + nxt * kHBufBytes;
    char* dbuf = lds + nn * kHBufBytes;
    H_TR_BASES(buf);
    H_S_PHASE(nbuf, sa, true);
    H_TICK(tk2);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = sa[r];
    H_O_PHASE(true, dbuf);
    H_TICK(tk3);
    H_TIMING_ACC();
    cur = nxt;
  }
  if (nc >= 2) {
    const int nxt = cur == kHBufs - 1 ? 0 : cur + 1;
    H_DMA_BARRIER();
    const char* buf = lds + cur * kHBufBytes;
    const char* nbuf = lds + nxt * kHBufBytes;
    H_TR_BASES(buf);
    H_S_PHASE(nbuf, sa, true);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = sa[r];
    H_O_PHASE(false, lds);
    cur = nxt;
  }
  {  // last chunk: nothing left to prefetch; run its exp / split alone
    H_DMA_BARRIER();
    const char* buf = lds + cur * kHBufBytes;
    H_TR_BASES(buf);
#pragma unroll
    for (int f = 0; f < 8; ++f) trh_frag_n<0>(f, ta2_, trc_);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x2 arg = f32x2{p[2 * s], p[2 * s + 1]} * sl2v + nrefv;
      const float e0 = __builtin_amdgcn_exp2f(arg[0]);
      const float e1 = __builtin_amdgcn_exp2f(arg[1]);
      l2 += f32x2{e0, e1};
      emax = fmaxf(emax, fmaxf(e0, e1));
      const f16x2 pa = pk_f16(e0, e1);
      const f16x2 pq = pk_f16(resid_lo(e0, pa), resid_hi(e1, pa));
      pw[0][s] = __builtin_bit_cast(uint32_t, pa);
      pw[1][s] = __builtin_bit_cast(uint32_t, pq);
    }
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        H_P_ST(2 * m2 + pl, pw[pl][4 * m2], pw[pl][4 * m2 + 1], pw[pl][4 * m2 + 2], pw[pl][4 * m2 + 3]);
      }
    H_O_PHASE(false, lds);
  }
  };  // sweep
  if (KIND == 1) {
    sweep(std::true_type{});
  } else {
    sweep(std::false_type{});
    // (the barrier also ends the last chunk's LDS reads before a redo's first DMA overwrites the ring)
    if (KIND == 2 && mode != 0 && __syncthreads_or((mode == 2 || !(emax <= kHOverflow)) ? 1 : 0)) {
      // (the redo starts from opaque copies of the thread coordinates: nothing the first sweep derived from them is kept
      // alive for it -- the compiler would rather hold 22 registers through the first sweep than recompute them)
      asm volatile("" : "+v"(t), "+v"(lane), "+v"(j), "+v"(h));
      sweep(std::true_type{});
    }
  }
#undef H_P_ST
  H_TIMING_WRITE(H_TIMING_Q);
  float* orow = part_O + ((int64_t)split * B + xrow) * k3D;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(orow + 32 * db + 8 * q + 4 * h) =
          make_float4(acc[db][4 * q], acc[db][4 * q + 1], acc[db][4 * q + 2], acc[db][4 * q + 3]);
  const float l = l2[0] + l2[1];
  const float ltot = l + __shfl_xor(l, 32, 64);
  if (h == 0) part_l[(int64_t)split * B + xrow] = ltot;
  // a probability that does not fit fp16 (or a forced redo: mode 2 is the test hook): the FIX launch redoes this block
  if (KIND == 0 && (mode == 2 || !(emax <= kHOverflow))) flags[blockIdx.x] = 1;
}

// -----------------------------------------------------------------------------------------------------------------
// ONE fp16 plane per operand (round 5): bf16 tables (BASELINE config 4's dtype).  A bf16 element has 8 significant bits, so
// x * 2^e is EXACT in fp16's 11 (elements below 2^-28 of the matrix maximum fall into fp16's subnormals and are rounded to
// multiples of 2^-38 of that maximum): the second plane of both operands is identically zero and so are two of the three
// cross terms of S^T and one of the three of O^T.  S^T is then ONE MFMA per k-step -- cheap enough that pass C RECOMPUTES
// it instead of reading stored probabilities:
//     pass Q:  S^T (1 term) + O^T = C^T P'^T (P' in two fp16 planes: 2 terms)           3 GEMMs
//     pass C:  S^T (1 term) + O^T = Q^T P   (true probabilities * 2^14, two planes)     3 GEMMs
// six executed fp16 GEMMs (the bf16 x 3 one-plane kernels: eight), no B x B matrix in memory at all (the fp32-table path
// above moves 2 x 268 MB of it per step at B = 8192).  Same skeleton as inbatch2h_q_kernel -- 3-slot LDS ring by
// LDS-DMA, S^T of chunk t + 1 threaded with the exp / split of chunk t, transposing LDS reads for the O^T A fragments --
// one kernel for both sides:
//   QSIDE: owned = Q rows; per-row optimistic exponent reference, overflow -> the workgroup redoes itself against the
//          exact maximum of its range (KIND 2 of inbatch2h_q_kernel); leaves part_m, part_l, part_O.
//   CSIDE: owned = C rows, streamed = Q rows i; p = exp2(s sl2 - lse2_i + 14) with the row's final lse2_i (merge<Q>),
//          DMA'd per chunk beside the plane tile; leaves part_O.
// -----------------------------------------------------------------------------------------------------------------
// DMA offset of thread t of a 512-thread workgroup: plane K, one 16-byte piece of the 32 x 128 tile each
template <int K>
__device__ __forceinline__ uint32_t dmah8_off0(int64_t B, int64_t chunk, int t) {  // piece K = plane K, 512 threads
  const int row = t >> 4, seg = (t & 15) ^ swz16(row);
  return (uint32_t)((((int64_t)K * B + chunk * 32 + row) * k3D + seg * 8) * 2);
}
// One 512-thread workgroup owns 256 rows (32 per wave), one workgroup per CU: the two waves of a SIMD belong to the SAME
// workgroup and meet at its barrier every chunk.  (As two 256-thread workgroups per CU nothing coupled the pair: the
// stamps showed the older workgroup of every CU done 37 us into the kernel and the younger one 53 us in -- the issue
// arbiter prefers the older wave, the younger runs its second half alone on its SIMD.)  The plane tile is fetched once
// for eight waves: one LDS-DMA instruction per wave and chunk.
constexpr int kH1Waves = 8, kH1Owned = 32 * kH1Waves;
constexpr int kH1BufBytes = kPlaneBytes + kH1Waves * 256;  // one plane tile + per-wave 256 B of streamed-row references
#define H1_DP(K, G, BUF) dmah_piece<K>(baseY, G, (BUF), w)
#define H1_DMA_REF(BUF)                                                                                   \
  __builtin_amdgcn_global_load_lds((gptr_t)(ref + (c0 + dpos) * 32 + (lane & 31)),                        \
                                   (lptr_t)((BUF) + kPlaneBytes + w * 256), 4, 0, 0)
#define H1_DMA_ADVANCE()                                                                   \
  {                                                                                        \
    const uint32_t step_ = (dpos + 1 == nc) ? (uint32_t)(8192 - nc * 8192) : 8192u;        \
    dpos = (dpos + 1 == nc) ? 0 : dpos + 1;                                                \
    g0 += step_;                                                                           \
  }
#define H1_DMA_CHUNK(BUF) { if (!QSIDE) { H1_DMA_REF(BUF); } H1_DP(0, g0, BUF); H1_DMA_ADVANCE(); }
// the references of the chunk in BUF for this lane's 16 accumulator registers (streamed rows 8 m + 4 h + 0..3)
// (assembly reads + one explicit wait: hipcc's own lgkmcnt bookkeeping does not count the assembly LDS reads of the S^T
// phase that follow, and would let these four be consumed before they have returned)
#define H1_LOAD_REFS(BUF)                                                                                 \
  if (!QSIDE) {                                                                                           \
    f16x8 raw_[4];                                                                                        \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_)                                                      \
      raw_[m_] = lds_b128<0>(lds32 + (uint32_t)((BUF) - lds) + (uint32_t)(kPlaneBytes + w * 256 + (8 * m_ + 4 * h) * 4)); \
    H_TR_WAIT();                                                                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                                    \
      const float4 lv_ = __builtin_bit_cast(float4, raw_[m_]);                                            \
      rf[4 * m_] = kHPexp - lv_.x; rf[4 * m_ + 1] = kHPexp - lv_.y;                                       \
      rf[4 * m_ + 2] = kHPexp - lv_.z; rf[4 * m_ + 3] = kHPexp - lv_.w;                                   \
    }                                                                                                     \
  }
// (no packed-f32 instructions: see pk_fma)
#define H1_EXP_PAIR(S, PN)                                                                                \
  {                                                                                                       \
    const float a0_ = fma_s(p[2 * (S)], sl2, QSIDE ? nref1 : rf[2 * (S)]);                                \
    const float a1_ = fma_s(p[2 * (S) + 1], sl2, QSIDE ? nref1 : rf[2 * (S) + 1]);                        \
    const float e0_ = __builtin_amdgcn_exp2f(a0_), e1_ = __builtin_amdgcn_exp2f(a1_);                     \
    const f16x2 pa_ = pk_f16(e0_, e1_);                                                                   \
    if (QSIDE) { sum_max(l2a, l2b, emax, e0_, e1_); }                                                     \
    const f16x2 pq_ = pk_f16(resid_lo(e0_, pa_), resid_hi(e1_, pa_));                                     \
    PN[0][(S)] = __builtin_bit_cast(uint32_t, pa_);                                                       \
    PN[1][(S)] = __builtin_bit_cast(uint32_t, pq_);                                                       \
  }
// The S^T phase: all eight A fragments of the chunk are requested at its top (32 registers; one plane leaves room for
// them), then its eight MFMAs run back to back with NOTHING between them.  LDS operations return in order; before the
// MFMA of k-step s the operations younger than fragment s are the 7 - s later fragments: they stay in flight.
// (Round 5, measured with the phase stamps of scripts/gpu_ib1h_timing.sh: the first form of this kernel threaded the
// exp / split of the previous chunk between these eight MFMAs like the two-plane kernel does between its 24 -- 1843 of an
// iteration's 2727 cycles went by in this phase and 648 in the O^T phase with its 16 MFMAs.  Eight MFMAs are 256 pipe
// cycles: no cover for ~130 VALU instructions.  The VALU work now rides in the O^T phase: H1_O_PHASE.)
#define H1_S_PHASE(NBUF, SA)                                                                              \
  {                                                                                                       \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) SA[r_] = 0.f;                                       \
    const uint32_t ap32_ = lds32 + (uint32_t)((NBUF) - lds) + (uint32_t)(j * 256);                        \
    const int sw_ = swz16(j);                                                                             \
    f16x8 af_[8];                                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_)                                                      \
      af_[s_] = lds_b128<0>(ap32_ + (uint32_t)(((2 * s_ + h) ^ sw_) << 4));                               \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                                    \
      H_SB();                                                                                             \
      H_S_WAIT(7 - s_);                                                                                   \
      H_SB();                                                                                             \
      SA = H_MFMA(af_[s_], bx[s_], SA);                                                                   \
      H_SB();                                                                                             \
    }                                                                                                     \
  }
#define H1_O_G0() { _Pragma("unroll") for (int f_ = 4; f_ < 8; ++f_) trh_frag_n<0>(f_, ta2_, trc_); }
#define H1_O_G1(F0, F1) { _Pragma("unroll") for (int f_ = (F0); f_ < (F1); ++f_) trh_frag_n<1>(f_, ta2_, trc_); }
#define H1_PB(PC)                                                                                         \
  f16x8 pb[2][2];                                                                                         \
  _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_)                                                        \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                                    \
      const u32x4 u_ = {PC[q_][4 * g_], PC[q_][4 * g_ + 1], PC[q_][4 * g_ + 2], PC[q_][4 * g_ + 3]};      \
      pb[q_][g_] = __builtin_bit_cast(f16x8, u_);                                                         \
    }
// one row of the O^T phase (four MFMAs, one per 32-column block of the owned rows' output) hosting the exp / split of two
// pairs A = S0, B = S0 + 1 of the NEXT chunk's scores (EXP_ON), the two pairs' dependent chains interleaved and spread
// behind all four MFMAs (the probe: a chain of dependent VALU instructions behind an MFMA costs what four times as many
// independent ones do)
#define H1_O_ROW(PL_P, G, S0, EXP_ON, PN)                                                                 \
  {                                                                                                       \
    float aA0_ = 0.f, aA1_ = 0.f, aB0_ = 0.f, aB1_ = 0.f, eA0_ = 0.f, eA1_ = 0.f, eB0_ = 0.f, eB1_ = 0.f; \
    float rA0_ = 0.f, rA1_ = 0.f;                                                                         \
    f16x2 paA_ = {0, 0}, paB_ = {0, 0};                                                                   \
    H_SB(); acc[0] = H_MFMA(ta2_[G][0][0], pb[PL_P][G], acc[0]); H_SB();                                  \
    if (EXP_ON) {                                                                                         \
      aA0_ = fma_s(p[2 * (S0)], sl2, QSIDE ? nref1 : rf[2 * (S0)]);                                       \
      aA1_ = fma_s(p[2 * (S0) + 1], sl2, QSIDE ? nref1 : rf[2 * (S0) + 1]);                               \
      aB0_ = fma_s(p[2 * (S0) + 2], sl2, QSIDE ? nref1 : rf[2 * (S0) + 2]);                               \
      aB1_ = fma_s(p[2 * (S0) + 3], sl2, QSIDE ? nref1 : rf[2 * (S0) + 3]);                               \
      eA0_ = __builtin_amdgcn_exp2f(aA0_); eA1_ = __builtin_amdgcn_exp2f(aA1_);                           \
    }                                                                                                     \
    H_SB(); acc[1] = H_MFMA(ta2_[G][1][0], pb[PL_P][G], acc[1]); H_SB();                                  \
    if (EXP_ON) {                                                                                         \
      eB0_ = __builtin_amdgcn_exp2f(aB0_); eB1_ = __builtin_amdgcn_exp2f(aB1_);                           \
      if (QSIDE) { sum_max(l2a, l2b, emax, eA0_, eA1_); }                                                 \
      paA_ = pk_f16(eA0_, eA1_);                                                                          \
    }                                                                                                     \
    H_SB(); acc[2] = H_MFMA(ta2_[G][2][0], pb[PL_P][G], acc[2]); H_SB();                                  \
    if (EXP_ON) {                                                                                         \
      if (QSIDE) { sum_max(l2a, l2b, emax, eB0_, eB1_); }                                                 \
      paB_ = pk_f16(eB0_, eB1_);                                                                          \
      rA0_ = resid_lo(eA0_, paA_); rA1_ = resid_hi(eA1_, paA_);                                           \
    }                                                                                                     \
    H_SB(); acc[3] = H_MFMA(ta2_[G][3][0], pb[PL_P][G], acc[3]); H_SB();                                  \
    if (EXP_ON) {                                                                                         \
      const f16x2 pqA_ = pk_f16(rA0_, rA1_);                                                              \
      const f16x2 pqB_ = pk_f16(resid_lo(eB0_, paB_), resid_hi(eB1_, paB_));                              \
      PN[0][(S0)] = __builtin_bit_cast(uint32_t, paA_); PN[1][(S0)] = __builtin_bit_cast(uint32_t, pqA_); \
      PN[0][(S0) + 1] = __builtin_bit_cast(uint32_t, paB_); PN[1][(S0) + 1] = __builtin_bit_cast(uint32_t, pqB_); \
    }                                                                                                     \
    H_SB();                                                                                               \
  }
// O^T += Y_chunk^T P^T of the CURRENT chunk (its probabilities in PC, the G = 0 fragments requested before the S^T
// phase) while the scores in p -- the NEXT chunk's -- become that chunk's probabilities in PN.  G1A / G1B: the requests
// for the G = 1 fragments (first and second half).
#define H1_O_PHASE_X(DMA_ON, DBUF, EXP_ON, PC, PN, G1A, G1B)                                              \
  {                                                                                                       \
    H1_PB(PC);                                                                                            \
    H_TR_WAIT();                                                                                          \
    H1_O_ROW(1, 0, 0, EXP_ON, PN); G1A; if (DMA_ON) { H1_DP(0, g0, DBUF); }                               \
    H1_O_ROW(0, 0, 2, EXP_ON, PN); G1B; if (DMA_ON) { if (!QSIDE) { H1_DMA_REF(DBUF); } }                 \
    H_TR_WAIT();                                                                                          \
    H1_O_ROW(1, 1, 4, EXP_ON, PN);                                                                        \
    H1_O_ROW(0, 1, 6, EXP_ON, PN);                                                                        \
    if (DMA_ON) H1_DMA_ADVANCE();                                                                         \
  }
#define H1_O_PHASE(DMA_ON, DBUF, EXP_ON, PC, PN) H1_O_PHASE_X(DMA_ON, DBUF, EXP_ON, PC, PN, H1_O_G1(4, 6), H1_O_G1(6, 8))
// --- the same iteration with the ring slot a COMPILE-TIME constant (the steady loop is unrolled by the ring's four
// slots): every LDS address of the iteration is then a per-lane base computed once before the loop plus an immediate
// offset, and the probabilities alternate between two register sets instead of being copied.  (The loop with a run-time
// slot spent 20 v_add_u32 and 33 v_mov_b32 per iteration on exactly that, beside the ~80 instructions of the exp / split
// itself; scripts/mfma_valu_probe.py: with two waves per SIMD more than ~6.5 VALU instructions per MFMA are exposed.)
template <int G, int DB, int SLOTOFF, class TA>
__device__ __forceinline__ void trh1_frag(TA& ta, const uint32_t (&tb)[4][2]) {
  constexpr int OFF = SLOTOFF + 16 * G * 256;
  const s16x4 lo = tr_read<OFF>(tb[DB][0]), hi = tr_read<OFF>(tb[DB][1]);
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  ta[G][DB][0] = __builtin_bit_cast(f16x8, both);
}
#define H1C_OFF(SLOT) ((SLOT) * kH1BufBytes)
#define H1C_G0(SLOT)                                                                                      \
  {                                                                                                       \
    trh1_frag<0, 0, H1C_OFF(SLOT)>(ta2_, trb_); trh1_frag<0, 1, H1C_OFF(SLOT)>(ta2_, trb_);               \
    trh1_frag<0, 2, H1C_OFF(SLOT)>(ta2_, trb_); trh1_frag<0, 3, H1C_OFF(SLOT)>(ta2_, trb_);               \
  }
#define H1C_G1A(SLOT) { trh1_frag<1, 0, H1C_OFF(SLOT)>(ta2_, trb_); trh1_frag<1, 1, H1C_OFF(SLOT)>(ta2_, trb_); }
#define H1C_G1B(SLOT) { trh1_frag<1, 2, H1C_OFF(SLOT)>(ta2_, trb_); trh1_frag<1, 3, H1C_OFF(SLOT)>(ta2_, trb_); }
#define H1C_LOAD_REFS(SLOT)                                                                               \
  if (!QSIDE) {                                                                                           \
    f16x8 raw_[4];                                                                                        \
    raw_[0] = lds_b128<H1C_OFF(SLOT) + kPlaneBytes>(rbase);                                               \
    raw_[1] = lds_b128<H1C_OFF(SLOT) + kPlaneBytes + 32>(rbase);                                          \
    raw_[2] = lds_b128<H1C_OFF(SLOT) + kPlaneBytes + 64>(rbase);                                          \
    raw_[3] = lds_b128<H1C_OFF(SLOT) + kPlaneBytes + 96>(rbase);                                          \
    H_TR_WAIT();                                                                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                                    \
      const float4 lv_ = __builtin_bit_cast(float4, raw_[m_]);                                            \
      rf[4 * m_] = kHPexp - lv_.x; rf[4 * m_ + 1] = kHPexp - lv_.y;                                       \
      rf[4 * m_ + 2] = kHPexp - lv_.z; rf[4 * m_ + 3] = kHPexp - lv_.w;                                   \
    }                                                                                                     \
  }
#if defined(H1_PROBE_NO_SLD)
#define H1C_SLD(SLOT, S) bx[S]
#else
#define H1C_SLD(SLOT, S) lds_b128<H1C_OFF(SLOT)>(sbase[S])
#endif
#define H1C_S_PHASE(SLOT, SA)                                                                             \
  {                                                                                                       \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) SA[r_] = 0.f;                                       \
    f16x8 af_[8];                                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) af_[s_] = H1C_SLD(SLOT, s_);                         \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                                    \
      H_SB();                                                                                             \
      H_S_WAIT(7 - s_);                                                                                   \
      H_SB();                                                                                             \
      SA = H_MFMA(af_[s_], bx[s_], SA);                                                                   \
      H_SB();                                                                                             \
    }                                                                                                     \
  }
// iteration on chunk `it` in slot CUR (probabilities in PC) with chunk it + 3 requested into slot CUR + 3
// (timing probes, results wrong: -DH1_PROBE_NO_DMA / NO_TR / NO_EXP / NO_SLD leave out the LDS-DMAs, the transposing
// reads, the exp / split, the S^T phase's fragment reads: scripts/gpu_ib1h_probe.sh)
#if defined(H1_PROBE_NO_DMA)
#define H1C_DMA_ON false
#else
#define H1C_DMA_ON true
#endif
#if defined(H1_PROBE_NO_EXP)
#define H1C_EXP_ON false
#else
#define H1C_EXP_ON true
#endif
#if defined(H1_PROBE_NO_TR)
#define H1C_TR(X)
#else
#define H1C_TR(X) X
#endif
// WAIT_PARTIAL: the previous iteration requested a chunk (its LDS-DMAs may stay in flight across the barrier: the chunk
// needed now is the one before it); DMA_RT: this iteration requests chunk it + 3 (both run-time, wave-uniform)
#define H1C_ITER(CUR, PC, PN, WAIT_PARTIAL, DMA_RT)                                                       \
  {                                                                                                       \
    H_TICK(tk0);                                                                                          \
    if (WAIT_PARTIAL) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kDmaPerChunk) : "memory"); } \
    else { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }                     \
    H_TICK(tk1);                                                                                          \
    H1C_TR(H1C_G0(CUR));                                                                                  \
    H1C_LOAD_REFS(((CUR) + 1) & 3);                                                                       \
    H1C_S_PHASE(((CUR) + 1) & 3, sa);                                                                     \
    H_TICK(tk2);                                                                                          \
    H1_TAKE_SCORES();                                                                                     \
    H1_O_PHASE_X(H1C_DMA_ON && (DMA_RT), lds + H1C_OFF(((CUR) + 3) & 3), H1C_EXP_ON, PC, PN,              \
                 H1C_TR(H1C_G1A(CUR)), H1C_TR(H1C_G1B(CUR)));                                             \
    H_TICK(tk3);                                                                                          \
    H_TIMING_ACC();                                                                                       \
  }
template <bool QSIDE, int DBG = 0>
__global__ ESR_NO_PK __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void inbatch1h_kernel(
    const _Float16* __restrict__ Xr, const _Float16* __restrict__ Yr, int64_t B, int nsplit, float sl2_in,
    const float* __restrict__ sc, const float* __restrict__ diag, const float* __restrict__ ref, int mode,
    float* __restrict__ part_m, float* __restrict__ part_O, float* __restrict__ part_l) {
  // A FOUR-slot ring, tiles fetched THREE chunks ahead: an iteration of this kernel is 24 MFMAs per wave (~1.8 us), about
  // one LDS-DMA round trip under load.  The loop-top wait leaves the youngest chunk's DMAs in flight.
  // Schedule of iteration `it` (chunk it = "current", its probabilities already in pw):
  //   barrier | G = 0 fragments of chunk it, references of chunk it + 1 | S^T of chunk it + 1 (8 MFMAs, nothing else) |
  //   O^T of chunk it (16 MFMAs) with the exp / split of chunk it + 1 between them -> pwn | pw = pwn
  constexpr int kRing = 4;
  constexpr int kDmaPerChunk = QSIDE ? 1 : 2;  // DMA instructions per wave and chunk (C side: + the references)
  constexpr int kLdsBytes = kRing * kH1BufBytes > kH1Waves * kTileLdsBytes ? kRing * kH1BufBytes : kH1Waves * kTileLdsBytes;
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];  // the ring; at the end the waves' output tiles
  int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  int j = lane & 31, h = lane >> 5;
  H_TR_SETUP();
  const uint32_t lds32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
  const int ob = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
  const int64_t wrow = (int64_t)ob * kH1Owned + w * 32;
  const bool live = wrow < B;  // B is a multiple of 128: the last workgroup's upper four waves may own nothing
  const int64_t xrow = (live ? wrow : 0) + j;  // (idle waves work on block 0's rows: valid addresses, results dropped)
  const int nc = (int)(B / k3Chunk) / nsplit;
  const int64_t c0 = (int64_t)split * nc;
  const float sl2 = sl2_in * sc[0];  // the planes carry 2^(eq + ec) S
  f32x16 acc[4];
  float l2a = 0.f, l2b = 0.f;
  int dpos = 0;
  const char* const baseY = reinterpret_cast<const char*>(Yr);
  uint32_t g0 = 0;
  f16x8 bx[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) bx[s] = *reinterpret_cast<const f16x8*>(Xr + xrow * k3D + 16 * s + 8 * h);
  f32x16 sa;
  float p[16], rf[16];
  uint32_t pw[2][8], pwn[2][8];
  f16x8 ta2_[2][4][2];
  float emax = 0.f, refv = -INFINITY;
  H_TIMING_DECL();
  float nref1 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) rf[r] = 0.f;
  // per-lane LDS bases of the constant-slot iterations (slot 0; the slot is an immediate offset there)
  uint32_t sbase[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) sbase[s] = lds32 + (uint32_t)(j * 256) + (uint32_t)(((2 * s + h) ^ swz16(j)) << 4);
  const uint32_t rbase = lds32 + (uint32_t)(w * 256 + 16 * h);
  (void)rbase;
#define H1_TAKE_SCORES() { _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) p[r_] = sa[r_]; }
#define H1_ROTATE_P()                                                                          \
  {                                                                                            \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_)                                           \
      _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) pw[q_][s_] = pwn[q_][s_];               \
  }
  auto sweep = [&](const bool fix) __attribute__((always_inline)) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
    l2a = l2b = 0.f;
    dpos = 0;
    g0 = dmah8_off0<0>(B, c0, t);
    emax = 0.f;
    refv = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = 0.f;
    if (QSIDE && fix) {  // exact maximum of s sl2 over this workgroup's chunks (flagged workgroups only)
      float m = -INFINITY;
      for (int c = 0; c < nc; ++c) {
        H1_DMA_CHUNK(lds);
        H_DMA_BARRIER();
        H1_S_PHASE(lds, sa);
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sa[r] * sl2);
        __syncthreads();  // the next tile overwrites this one
      }
      refv = fmaxf(m, __shfl_xor(m, 32, 64)) - kHPexp;
    }
    // the reference of the owned rows from their first chunk's scores (Q side), what merge<Q> adds back
#define H1_FIRST_CHUNK_REF(DREF)                                                               \
  if (QSIDE) {                                                                                 \
    if (!fix) {                                                                                \
      float m_ = (DREF);                                                                       \
      if (!sample) {                                                                           \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) m_ = fmaxf(m_, sa[r_] * sl2);        \
      }                                                                                        \
      refv = fmaxf(m_, __shfl_xor(m_, 32, 64)) - kHOptHead;                                    \
    }                                                                                          \
    if (h == 0 && live) part_m[(int64_t)split * B + xrow] = refv;                              \
    nref1 = -refv;                                                                             \
  }
    // the optimistic reference of a row is the same for every split: max(diagonal score, scores against chunk 0) - 4
    // (inbatch2h_q_kernel: the two kernels agree bit for bit on bf16-valued rows); splits other than 0 fetch chunk 0 too
    const bool sample = QSIDE && !fix && c0 != 0;
#define H1_SAMPLE_MAX(BUF, M)                                                                  \
  {                                                                                            \
    H1_S_PHASE(BUF, sa);                                                                       \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) M = fmaxf(M, sa[r_] * sl2);              \
  }
    if (DBG == 1) {  // debugging aid: one chunk at a time, nothing pipelined
      float dref0 = (QSIDE && !fix) ? diag[xrow] * sl2_in : -INFINITY;
      if (sample) {
        H1_DP(0, dmah8_off0<0>(B, 0, t), lds);
        H_DMA_BARRIER();
        H1_SAMPLE_MAX(lds, dref0);
        __syncthreads();
      }
      for (int c = 0; c < nc; ++c) {
        H1_DMA_CHUNK(lds);
        H_DMA_BARRIER();
        H1_S_PHASE(lds, sa);
        H1_TAKE_SCORES();
        if (c == 0) { H1_FIRST_CHUNK_REF(dref0); }
        H_TR_BASES(lds);
        H1_LOAD_REFS(lds);
        H1_O_G0();
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) H1_EXP_PAIR(s2, pw);
        H1_O_PHASE(false, lds, false, pw, pwn);
        __syncthreads();
      }
      return;
    }
    H1_DMA_CHUNK(lds);
    if (nc > 1) H1_DMA_CHUNK(lds + kH1BufBytes);
    if (nc > 2) H1_DMA_CHUNK(lds + 2 * kH1BufBytes);
    if (sample) H1_DP(0, dmah8_off0<0>(B, 0, t), lds + 3 * kH1BufBytes);  // (the ring's fourth slot: free until chunk 3)
    float dref = (QSIDE && !fix) ? diag[xrow] * sl2_in : -INFINITY;
    H_DMA_BARRIER();
    if (sample) H1_SAMPLE_MAX(lds + 3 * kH1BufBytes, dref);
    H1_S_PHASE(lds, sa);
    H1_TAKE_SCORES();
    H1_FIRST_CHUNK_REF(dref);
    H1_LOAD_REFS(lds);
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) H1_EXP_PAIR(s2, pw);  // chunk 0's probabilities, alone
    int cur = 0, it = 0;
    H_TIMING_START();
    // steady state: chunk it + 3 is requested while chunk it is worked on -- four iterations, the ring's slots 0..3 in
    // turn as compile-time constants (H1C_ITER), the probabilities alternating between pw and pwn
    for (; it + 6 < nc; it += 4) {
      H1C_ITER(0, pw, pwn, true, true);
      H1C_ITER(1, pwn, pw, true, true);
      H1C_ITER(2, pw, pwn, true, true);
      H1C_ITER(3, pwn, pw, true, true);
    }
    H_TIMING_WRITE(QSIDE && !fix);
    // the 3..6 chunks left (fewer when nc < 3), slot by run-time index (cur = it % 4 = 0 here; ~50 more VALU
    // instructions per iteration on addresses and on moving pwn to pw)
    for (; it + 3 < nc; ++it) {
      const int nxt = (cur + 1) & (kRing - 1);
      const int nn = (cur + 3) & (kRing - 1);  // held chunk it - 1: every wave left it at the barrier below
      // chunk it + 1 has landed (everything but the youngest chunk's DMAs) and everyone is done with iteration it - 1
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kDmaPerChunk) : "memory");
      const char* buf = lds + cur * kH1BufBytes;
      const char* nbuf = lds + nxt * kH1BufBytes;
      char* dbuf = lds + nn * kH1BufBytes;
      H_TR_BASES(buf);
      H1_O_G0();
      H1_LOAD_REFS(nbuf);
      H1_S_PHASE(nbuf, sa);
      H1_TAKE_SCORES();
      H1_O_PHASE(true, dbuf, true, pw, pwn);
      H1_ROTATE_P();
      cur = nxt;
    }
    for (; it + 1 < nc; ++it) {  // the last tiles are on their way or here: nothing left to request
      const int nxt = (cur + 1) & (kRing - 1);
      H_DMA_BARRIER();
      const char* buf = lds + cur * kH1BufBytes;
      const char* nbuf = lds + nxt * kH1BufBytes;
      H_TR_BASES(buf);
      H1_O_G0();
      H1_LOAD_REFS(nbuf);
      H1_S_PHASE(nbuf, sa);
      H1_TAKE_SCORES();
      H1_O_PHASE(false, lds, true, pw, pwn);
      H1_ROTATE_P();
      cur = nxt;
    }
    {  // last chunk: its probabilities are in pw, nothing follows
      const char* buf = lds + cur * kH1BufBytes;
      H_TR_BASES(buf);
      H1_O_G0();
      H1_O_PHASE(false, lds, false, pw, pwn);
    }
    H_TIMING_MARK(3, QSIDE && !fix);
  };
  sweep(false);
  if (QSIDE && __syncthreads_or((mode == 2 || (live && !(emax <= kHOverflow))) ? 1 : 0)) {
    asm volatile("" : "+v"(t), "+v"(lane), "+v"(j), "+v"(h));
    sweep(true);
  }
#if defined(H1_PROBE_NO_STORE)
  if (mode == 77) part_O[xrow] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
#elif defined(H1_PROBE_DIRECT_STORE)
  float* orow = part_O + ((int64_t)split * B + xrow) * k3D;
  if (live)
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(orow + 32 * db + 8 * q + 4 * h) =
          make_float4(acc[db][4 * q], acc[db][4 * q + 1], acc[db][4 * q + 2], acc[db][4 * q + 3]);
#else
  __syncthreads();  // the ring is free: every wave is past its last chunk
  if (live) store_tile_via_lds(lds + w * kTileLdsBytes, acc, part_O + ((int64_t)split * B + wrow) * k3D, lane);
#endif
  if (QSIDE) {
    const float l = l2a + l2b;
    const float ltot = l + __shfl_xor(l, 32, 64);
    if (h == 0 && live) part_l[(int64_t)split * B + xrow] = ltot;
  }
}

// -----------------------------------------------------------------------------------------------------------------
// Pass C (round 6): owned = C rows j, streamed = Q rows i:  dC_j = sum_i p_ij q_i,  p_ij = P'_ij f_{i,s}  with P' the
// probabilities pass Q stored and f_{i,s} = 2^14 2^(M_s - M) / l_i the factor of streamed row i for the pass-Q split s
// that row j lies in (fac2h_kernel).  Until round 6 this pass multiplied every P' by its factor and split the product
// into two fp16 planes again -- 16 ds_read_b32, 16 multiplies and 48 conversion instructions per chunk and wave beside
// 24 MFMAs, matrix pipes 33 % busy.  Now the factor sits on the OTHER operand:
//     dC_j = sum_i P'_ij (f_{i,s} q_i)
// with one reference per row for all pass-Q splits (inbatch2h_q_kernel) f_{i,s} = f_i, facscale2h_kernel leaves ONE scaled
// copy of Q, x' = (f_i q_i) 2^(eq - 11) in two fp16 planes, and the stored planes of P' ARE the MFMA's B operand:
//   * the P' tile of (this wave's 32 owned rows, chunk) -- wave-private, 4 KB -- comes by LDS-DMA into a 3-slot ring of the
//     wave's own as four contiguous 1 KB blocks (plane, m') in pass Q's piece order (with every lane fetching the piece
//     that makes the LDS image a plain row-major matrix -- 16-byte pieces 512 B apart from neighbouring lanes -- the pass
//     took 65 us against 56 with contiguous requests), and is read with ds_read_b64_tr_b16: 8 reads per chunk, no VALU
//     instruction at all.  A transposing read touches 4 rows i x 2 m' x 2 h_q pieces per half wave: pass Q's position
//     formula puts the (h_q, i / 4 % 2) combinations on the four 64-byte columns of a 256-byte row, and the m' = 1 blocks
//     sit 64 bytes further on in LDS -- the 32 lanes cover all 64 banks;
//   * the slot is free as soon as those 8 reads have returned: the tile of chunk it + 3 is requested into it -- three
//     tiles (96 KB per CU) in flight or waiting, which is what keeps the 268 MB stream at the HBM rate;
//   * tile column n holds owned row pi(n) = n with bits 2 and 3 swapped (the order pass Q's accumulators hold them in):
//     the output tile is stored with its rows permuted.
// The partial O' rows are rescaled to the unit the merges expect (2^(eq + 14) sum p q) by one exact power of two (cexp =
// 2^11).
// GENERAL form (GEN, chosen per workgroup at run time): streams the UNSCALED planes of Q and applies the factors to the
// B fragments in registers (hi + lo is exact in f32; multiply, split again -- the arithmetic of the round-5 kernel).  Taken
// for a pass-Q split that facscale2h_kernel flagged: a pass-Q workgroup of it redid itself (its rows carry their own
// reference), or a live row of the copy lies more than ~17 binades under the largest possible element, where its second
// plane would fall into fp16's subnormals (a normaliser beyond ~2^20).
// -----------------------------------------------------------------------------------------------------------------
#if defined(CT_PROBE_P_DUMMY)  /* timing probe only (values wrong): no P' stream, every tile request hits the cache */
#define CT_PROBE_DUMMY 1
#else
#define CT_PROBE_DUMMY 0
#endif
#if defined(CT_PROBE_NO_P) || defined(CT_PROBE_NO_DMA)  /* timing probes only (values wrong): no tile requests / none at all */
#define CT_PROBE_NOP 1
#else
#define CT_PROBE_NOP 0
#endif
#if defined(CT_PROBE_NO_DMA)
#define CT_PROBE_NOY 1
#else
#define CT_PROBE_NOY 0
#endif
#ifndef CT_P_IN_LOAD
#define CT_P_IN_LOAD 4  /* how many of a chunk's four tile requests LOAD issues (the rest: between COMPUTE's rows) */
#endif
constexpr int kCtRing = 3;
constexpr int kCtYSlot = 2 * kPlaneBytes + 8 * 256;  // two planes + 256 B of factors per wave (general form)
constexpr int kCtPOff = kCtRing * kCtYSlot;          // the waves' private P' rings behind the plane ring
constexpr int kCtPSlot = 4096 + 64;                  // a P' tile: blocks at 0, 1024, 2048 + 64, 3072 + 64 (see above)
constexpr int kCtPWave = kCtRing * kCtPSlot;
constexpr int kCtLds = kCtPOff + 8 * kCtPWave;       // 150 KB of the CU's 160
constexpr int kCtOwned = 256;
constexpr float kCtLive = 0.0625f;                   // a row's largest |x'| below 2^-4: second plane subnormal (see above)
#ifndef H_P_LOAD_AUX
#define H_P_LOAD_AUX 2  /* cache-policy bits of the P' tile loads: nt (see H_P_ST); 0 for an A/B build */
#endif

// A fragment F (0..7: plane (F / 4 + 1) % 2, column block F % 4) of k-step G from the plane tile at SLOTOFF
template <int G, int F, int SLOTOFF, class TA>
__device__ __forceinline__ void ct_frag(TA& ta, const uint32_t (&tb)[4][2]) {
  constexpr int PL = (F / 4 + 1) % 2, DB = F % 4, OFF = SLOTOFF + PL * kPlaneBytes + 16 * G * 256;
  const s16x4 lo = tr_read<OFF>(tb[DB][0]), hi = tr_read<OFF>(tb[DB][1]);
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  ta[G][DB][PL] = __builtin_bit_cast(f16x8, both);
}
template <int G, int SLOTOFF, class TA>
__device__ __forceinline__ void ct_frag_n(int f, TA& ta, const uint32_t (&tb)[4][2]) {  // f is an unrolled constant
  switch (f) {
    case 0: ct_frag<G, 0, SLOTOFF>(ta, tb); break;
    case 1: ct_frag<G, 1, SLOTOFF>(ta, tb); break;
    case 2: ct_frag<G, 2, SLOTOFF>(ta, tb); break;
    case 3: ct_frag<G, 3, SLOTOFF>(ta, tb); break;
    case 4: ct_frag<G, 4, SLOTOFF>(ta, tb); break;
    case 5: ct_frag<G, 5, SLOTOFF>(ta, tb); break;
    case 6: ct_frag<G, 6, SLOTOFF>(ta, tb); break;
    default: ct_frag<G, 7, SLOTOFF>(ta, tb); break;
  }
}
// B fragment (plane PL, k-step G) of the P' tile in ring slot offset SLOTOFF: rows 16 G + 4 h + 0..3 and + 8
template <int PL, int G, int SLOTOFF>
__device__ __forceinline__ f16x8 ct_pfrag(uint32_t pbase) {
  constexpr int OFF = SLOTOFF + PL * 1024 + G * 512;  // (rows 8 r + 4 h + 0..3 of k-step G: 256-byte row 2 G + r of a block)
  const s16x4 lo = tr_read<OFF>(pbase), hi = tr_read<OFF + 256>(pbase);
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(f16x8, both);
}

__global__ ESR_NO_PK __launch_bounds__(512) void inbatch2h_pct_kernel(
    const _Float16* __restrict__ Yt, const _Float16* __restrict__ Yh, int64_t B, int nsplit,
    const float* __restrict__ fac, int nc_q, const char* __restrict__ Pt, float cexp,
    const int* __restrict__ cflags, int force_general, float* __restrict__ part_O) {
  constexpr int kLdsBytes = kCtLds > 8 * kTileLdsBytes ? kCtLds : 8 * kTileLdsBytes;
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];  // the rings; at the end the waves' output tiles
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int j = lane & 31, h = lane >> 5;
  (void)j;
  H_TR_SETUP();
  (void)trc_;
  const int ob = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
  const int64_t wrow = (int64_t)ob * kCtOwned + w * 32;
  const bool live = wrow < B;  // B is a multiple of 128: the last block's upper four waves may own nothing
  const int nc = (int)(B / k3Chunk) / nsplit;
  const int64_t c0 = (int64_t)split * nc;
  const int64_t nch = B / 32;
  const int64_t jt = live ? (wrow >> 5) : 0;  // (idle waves fetch block 0's tiles: valid addresses, results dropped)
  // workgroup-uniform: the flags of the pass-Q splits its eight waves' rows lie in
  bool general = force_general != 0;
  if (!general) {
    const int s_lo = (int)(((int64_t)ob * (kCtOwned / 32)) / nc_q);
    const int s_hi = (int)(min((int64_t)ob * (kCtOwned / 32) + (kCtOwned / 32 - 1), nch - 1) / nc_q);
    for (int sq = s_lo; sq <= s_hi; ++sq) general = general || cflags[sq] != 0;
  }
  const float* ref = fac + (int64_t)(jt / nc_q) * B;               // general form: this WAVE's factors
  const char* const pw_base = Pt + (jt * nch + c0) * 4096;
  const uint32_t lds32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
  // tile block G4 = 2 m' + plane: 1 KB, lane-linear
  uint32_t p_off[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) p_off[g4] = (uint32_t)(g4 * 1024 + lane * 16);
  // transposing reads of the tile: lane 16 (2 h + jb) + 4 a + e reads 8 bytes (half e % 2) of piece (row 16 G + 8 r +
  // 4 h + a, m' = jb, h_q = e / 2): block m' (2048 + 64 bytes apart), 256-byte row 2 G + r (the immediate offset),
  // 64-byte column 2 h_q + h, 16-byte piece a
  const uint32_t pbase = lds32 + (uint32_t)(kCtPOff + w * kCtPWave + ((lane >> 4) & 1) * (2048 + 64) +
                                            (2 * ((lane & 3) >> 1) + h) * 64 + tr_a * 16 + (lane & 1) * 8);
  const uint32_t rbase = lds32 + (uint32_t)(2 * kPlaneBytes + w * 256 + 16 * h);  // general form: factors of a slot

  f32x16 acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
  f16x8 ta2_[2][4][2];
  f16x8 pb[2][2];
  float rf[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rf[r] = 0.f;

  auto sweep = [&](auto gen_tag) __attribute__((always_inline)) {
    constexpr bool GEN = decltype(gen_tag)::value;
    // DMA instructions per wave and chunk: 2 (+ 1) for the planes (and factors), 4 for the P' tile.  In front of
    // LOAD(k) the requests younger than chunk k's planes are: tile k + 1, planes k + 1, tile k + 2
    constexpr int kWaitTop = CT_PROBE_NOP ? (CT_PROBE_NOY ? 0 : 2) : 4 + (GEN ? 3 : 2) + 4;
    const char* const baseY = reinterpret_cast<const char*>(GEN ? Yh : Yt);
    const char* const dummy = reinterpret_cast<const char*>(Yh);  // source of requests past the end: cache-resident, unused
    int dpos = 0;
    uint32_t g0 = dmah8_off0<0>(B, c0, t), g1 = dmah8_off0<1>(B, c0, t);
    const int wq = w & 3;
// plane (and factor) requests of the next chunk in the request order, piece by piece (CT_COMPUTE spreads them between
// its MFMA rows): I = 0, 1 the two planes, I = 2 the factors and the advance of the request position
#define CT_DMA_Y_PIECE(SLOT, I)                                                                           \
  if (!CT_PROBE_NOY) {                                                                                    \
    if ((I) == 0)                                                                                         \
      __builtin_amdgcn_global_load_lds((gptr_t)(baseY + g0), (lptr_t)(lds + (SLOT) * kCtYSlot + w * 1024), 16, 0, 0); \
    if ((I) == 1)                                                                                         \
      __builtin_amdgcn_global_load_lds((gptr_t)(baseY + g1),                                              \
                                       (lptr_t)(lds + (SLOT) * kCtYSlot + kPlaneBytes + w * 1024), 16, 0, 0); \
    if ((I) == 2) {                                                                                       \
      if (GEN)                                                                                            \
        __builtin_amdgcn_global_load_lds((gptr_t)(ref + (c0 + dpos) * 32 + (lane & 31)),                 \
                                         (lptr_t)(lds + (SLOT) * kCtYSlot + 2 * kPlaneBytes + w * 256), 4, 0, 0); \
      const uint32_t step_ = (dpos + 1 == nc) ? (uint32_t)(8192 - nc * 8192) : 8192u;                     \
      dpos = (dpos + 1 == nc) ? 0 : dpos + 1;                                                             \
      g0 += step_; g1 += step_;                                                                           \
    }                                                                                                     \
  }
#define CT_DMA_Y(SLOT) { CT_DMA_Y_PIECE(SLOT, 0); CT_DMA_Y_PIECE(SLOT, 1); CT_DMA_Y_PIECE(SLOT, 2); }
#define CT_DMA_P_PIECE(SLOT, CH, G4)                                                                      \
  if (!CT_PROBE_NOP) {                                                                                    \
    const char* src_ = (!CT_PROBE_DUMMY && (CH) < nc) ? pw_base + (int64_t)(CH) * 4096 : dummy;           \
    __builtin_amdgcn_global_load_lds((gptr_t)(src_ + p_off[G4]),                                          \
                                     (lptr_t)(lds + kCtPOff + w * kCtPWave + (SLOT) * kCtPSlot + (G4) * 1024 +  \
                                              ((G4) >> 1) * 64), 16, 0,                                   \
                                     H_P_LOAD_AUX);                                                       \
  }
#define CT_DMA_P(SLOT, CH) \
  { CT_DMA_P_PIECE(SLOT, CH, 0); CT_DMA_P_PIECE(SLOT, CH, 1); CT_DMA_P_PIECE(SLOT, CH, 2); CT_DMA_P_PIECE(SLOT, CH, 3); }
// general form: this lane's 16 factors of the chunk in plane slot SLOT (rf[8 g + e]: row 16 g + 8 (e / 4) + 4 h + e % 4)
#define CT_LOAD_REFS(SLOT)                                                                                \
  if (GEN) {                                                                                              \
    f16x8 raw_[4];                                                                                        \
    raw_[0] = lds_b128<(SLOT) * kCtYSlot>(rbase);                                                         \
    raw_[1] = lds_b128<(SLOT) * kCtYSlot + 32>(rbase);                                                    \
    raw_[2] = lds_b128<(SLOT) * kCtYSlot + 64>(rbase);                                                    \
    raw_[3] = lds_b128<(SLOT) * kCtYSlot + 96>(rbase);                                                    \
    H_TR_WAIT();                                                                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                                    \
      const float4 lv_ = __builtin_bit_cast(float4, raw_[m_]);                                            \
      rf[4 * m_] = lv_.x; rf[4 * m_ + 1] = lv_.y; rf[4 * m_ + 2] = lv_.z; rf[4 * m_ + 3] = lv_.w;         \
    }                                                                                                     \
  }
// general form: p = (hi + lo) f, split into two planes again, in place
#define CT_APPLY_FAC()                                                                                    \
  if (GEN) {                                                                                              \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)                                                      \
      _Pragma("unroll") for (int e_ = 0; e_ < 8; e_ += 2) {                                               \
        const float x0_ = ((float)pb[0][g_][e_] + (float)pb[1][g_][e_]) * rf[8 * g_ + e_];                \
        const float x1_ = ((float)pb[0][g_][e_ + 1] + (float)pb[1][g_][e_ + 1]) * rf[8 * g_ + e_ + 1];    \
        const f16x2 pa_ = pk_f16(x0_, x1_);                                                               \
        const f16x2 pq_ = pk_f16(resid_lo(x0_, pa_), resid_hi(x1_, pa_));                                 \
        pb[0][g_][e_] = pa_[0]; pb[0][g_][e_ + 1] = pa_[1];                                               \
        pb[1][g_][e_] = pq_[0]; pb[1][g_][e_ + 1] = pq_[1];                                               \
      }                                                                                                   \
  }
#define CT_G0(SLOT) { _Pragma("unroll") for (int f_ = 0; f_ < 8; ++f_) ct_frag_n<0, (SLOT) * kCtYSlot>(f_, ta2_, trb_); }
#define CT_G1(SLOT, F0, F1) \
  { _Pragma("unroll") for (int f_ = (F0); f_ < (F1); ++f_) ct_frag_n<1, (SLOT) * kCtYSlot>(f_, ta2_, trb_); }
// The two waves of a SIMD (w and w + 4) take turns on the matrix pipe.  (The first form of this kernel ran all eight
// waves through "barrier, 24 LDS reads, wait, 4 DMA issues, 24 MFMAs" together: with the P' stream replaced by cached
// reads it still took 58 us, 2400 cycles per chunk at 1.38 GHz for 1536 cycles of MFMAs per SIMD -- every SIMD idle
// while both its waves loaded.)  LOAD(k): all 40 transposing reads of chunk k into registers.  COMPUTE(k): the chunk's
// 24 MFMAs, and between their rows the seven requests a chunk costs a wave: planes k + 2 (the slot chunk k - 1 left)
// and tile k + 3 (into the slot LOAD(k) has just emptied).  [With the requests in LOAD -- six LDS-DMA issues of four
// waves at once in front of and behind the reads -- LOAD took 1000 cycles against COMPUTE's 745 (stamps: 278 for the two
// plane requests, 422 for the reads, 310 for the four tile requests), 2470 cycles per chunk.]  Segments are separated
// by workgroup barriers; waves 0-3 run LOAD(k) | COMPUTE(k), waves 4-7 one segment later:
//     segment 2k    : waves 0-3 LOAD(k)     waves 4-7 COMPUTE(k - 1)
//     segment 2k + 1: waves 0-3 COMPUTE(k)  waves 4-7 LOAD(k)
// The plane slot of chunk k - 1 is last read in segment 2k - 1 and refilled from segment 2k + 1 on.  Chunk k's planes
// have landed for every wave before the barrier that opens segment 2k: the requests younger than them are tile k + 1,
// planes k + 1, tile k + 2 for waves 0-3 (vmcnt(kWaitTop)) and tile k + 1 alone for waves 4-7, whose COMPUTE(k - 1)
// comes after that barrier (vmcnt(4)).
#define CT_LOAD(S, IT)                                                                                    \
  {                                                                                                       \
    pb[0][0] = ct_pfrag<0, 0, (S) * kCtPSlot>(pbase); pb[1][0] = ct_pfrag<1, 0, (S) * kCtPSlot>(pbase);   \
    pb[0][1] = ct_pfrag<0, 1, (S) * kCtPSlot>(pbase); pb[1][1] = ct_pfrag<1, 1, (S) * kCtPSlot>(pbase);   \
    CT_G0(S);                                                                                             \
    CT_G1(S, 0, 8);                                                                                       \
    CT_LOAD_REFS(S);                                                                                      \
    H_TR_WAIT();                                                                                          \
    if (CT_P_IN_LOAD >= 1) CT_DMA_P_PIECE(S, (IT) + 3, 0);                                                \
    if (CT_P_IN_LOAD >= 2) CT_DMA_P_PIECE(S, (IT) + 3, 1);                                                \
    if (CT_P_IN_LOAD >= 3) CT_DMA_P_PIECE(S, (IT) + 3, 2);                                                \
    if (CT_P_IN_LOAD >= 4) CT_DMA_P_PIECE(S, (IT) + 3, 3);                                                \
    CT_APPLY_FAC();                                                                                       \
  }
// COMPUTE of the chunk loaded from slots S (chunk IT).  The four waves of a half run it in lockstep behind the barrier: a
// request issued by all four at the same point queues behind the other three in the CU's address unit (16 cycles per
// 1 KB request; with the seven requests after whole rows the stamps showed COMPUTE at 1131 cycles against 745 without
// them).  Wave q = w % 4 issues its request of a row behind the row's MFMA q: the four waves' requests are 32 cycles
// apart and cost the issuing wave nothing but the issue slot.
#define CT_ROW_DMA(PL_A, PL_P, G, DMA)                                                                    \
  _Pragma("unroll") for (int db_ = 0; db_ < 4; ++db_) {                                                   \
    acc[db_] = H_MFMA(ta2_[G][db_][PL_A], pb[PL_P][G], acc[db_]);                                         \
    if (db_ == wq) { DMA; }                                                                               \
  }
#define CT_COMPUTE(S, IT)                                                                                 \
  {                                                                                                       \
    H_SB(); CT_ROW_DMA(1, 0, 0, CT_DMA_Y_PIECE(((S) + 2) % 3, 0));                                        \
    H_SB(); CT_ROW_DMA(0, 1, 0, CT_DMA_Y_PIECE(((S) + 2) % 3, 1); CT_DMA_Y_PIECE(((S) + 2) % 3, 2));      \
    H_SB(); CT_ROW_DMA(0, 0, 0, if (CT_P_IN_LOAD < 1) CT_DMA_P_PIECE(S, (IT) + 3, 0));                    \
    H_SB(); CT_ROW_DMA(1, 0, 1, if (CT_P_IN_LOAD < 2) CT_DMA_P_PIECE(S, (IT) + 3, 1));                    \
    H_SB(); CT_ROW_DMA(0, 1, 1, if (CT_P_IN_LOAD < 3) CT_DMA_P_PIECE(S, (IT) + 3, 2));                    \
    H_SB(); CT_ROW_DMA(0, 0, 1, if (CT_P_IN_LOAD < 4) CT_DMA_P_PIECE(S, (IT) + 3, 3));                    \
    H_SB();                                                                                               \
  }
#define CT_BAR_TOP() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kWaitTop) : "memory")
#define CT_BAR_TOP_B() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(CT_PROBE_NOP ? 0 : 4) : "memory")
#define CT_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    CT_DMA_P(0, 0);
    CT_DMA_Y(0);
    CT_DMA_P(1, 1);
    CT_DMA_Y(1);
    CT_DMA_P(2, 2);
    if (w < 4) {
#ifdef H_TIMING  /* scripts/ct_timing.py, -DH_TIMING=2: cycles in front of the top barrier / LOAD / middle barrier / COMPUTE */
      unsigned long long tk_[5] = {0, 0, 0, 0, 0}, ta_[4] = {0, 0, 0, 0};
      const unsigned long long rentry_ = __builtin_amdgcn_s_memrealtime(), tstart_ = __builtin_readcyclecounter();
#define CT_T(I) { H_SB(); tk_[I] = __builtin_readcyclecounter(); H_SB(); }
#define CT_TACC() { ta_[0] += tk_[1] - tk_[0]; ta_[1] += tk_[2] - tk_[1]; ta_[2] += tk_[3] - tk_[2]; ta_[3] += tk_[4] - tk_[3]; }
#else
#define CT_T(I)
#define CT_TACC()
#endif
#define CT_STEP_A(S, IT) \
  { CT_T(0); CT_BAR_TOP(); CT_T(1); CT_LOAD(S, IT); CT_T(2); CT_BAR(); CT_T(3); CT_COMPUTE(S, IT); CT_T(4); CT_TACC(); }
      for (int it = 0; it < nc; it += 3) {
        CT_STEP_A(0, it);
        if (it + 1 < nc) { CT_STEP_A(1, it + 1); }
        if (it + 2 < nc) { CT_STEP_A(2, it + 2); }
      }
#ifdef H_TIMING
      if (lane == 0 && (blockIdx.x & 1) == 0 && blockIdx.x < 512) {
        unsigned long long* d = esr_ib2h_dbg + (((blockIdx.x >> 1) * 4 + w) * 4);
        d[0] = ta_[0]; d[1] = ta_[1]; d[2] = ta_[2] + ta_[3]; d[3] = __builtin_readcyclecounter() - tstart_;
        unsigned long long* e = esr_ib2h_dbg + 4096 + (((blockIdx.x >> 1) * 4 + w) * 4);
        e[0] = rentry_; e[1] = rentry_; e[2] = __builtin_amdgcn_s_memrealtime(); e[3] = ta_[3];
      }
#endif
#undef CT_STEP_A
#undef CT_T
#undef CT_TACC
      CT_BAR_TOP();  // (the barrier in front of the other half's last COMPUTE)
    } else {
      // (the slot and chunk of the COMPUTE in a step are the previous step's)
      for (int it = 0; it < nc; it += 3) {
        CT_BAR_TOP_B(); if (it > 0) { CT_COMPUTE(2, it - 1); } CT_BAR(); CT_LOAD(0, it);
        if (it + 1 < nc) { CT_BAR_TOP_B(); CT_COMPUTE(0, it); CT_BAR(); CT_LOAD(1, it + 1); }
        if (it + 2 < nc) { CT_BAR_TOP_B(); CT_COMPUTE(1, it + 1); CT_BAR(); CT_LOAD(2, it + 2); }
      }
      CT_BAR_TOP_B();
      switch ((nc - 1) % 3) {  // the last chunk's slot
        case 0: CT_COMPUTE(0, nc - 1); break;
        case 1: CT_COMPUTE(1, nc - 1); break;
        default: CT_COMPUTE(2, nc - 1); break;
      }
    }
#undef CT_ROW_DMA
#undef CT_BAR
#undef CT_BAR_TOP_B
#undef CT_BAR_TOP
#undef CT_COMPUTE
#undef CT_LOAD
#undef CT_G1
#undef CT_G0
#undef CT_APPLY_FAC
#undef CT_LOAD_REFS
#undef CT_DMA_P
#undef CT_DMA_P_PIECE
#undef CT_DMA_Y
#undef CT_DMA_Y_PIECE
  };
  if (general) {
    sweep(std::true_type{});
  } else {
    sweep(std::false_type{});
    const float ce = cexp;  // exact power of two: the copy's exponent against the planes' common one
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[db][r] *= ce;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requests past the end
  __syncthreads();                                   // the rings are free: every wave is past its last chunk
  if (live) store_tile_via_lds<true>(lds + w * kTileLdsBytes, acc, part_O + ((int64_t)split * B + wrow) * k3D, lane);
}

// The factors pass C needs from pass Q's per-split references and normalisers -- fac[s][i] = 2^14 2^(M_s - M) / l_i --
// as a launch of their own (one thread per row, 16 loads each): merge<Q> computes the same numbers with the same
// operations in the same order (the factors are bit-identical), but also reads the 8 x 4 MB of partial O rows and the
// tower rows and writes gQ; with this launch in front of pass C, merge<Q> leaves the critical path and runs beside
// pass C on a second stream (inbatch2h_run, overlapped form).
__global__ __launch_bounds__(256) void fac2h_kernel(int64_t B, int nsplit, const float* __restrict__ part_m,
                                                   const float* __restrict__ part_l, float invl_scale,
                                                   float* __restrict__ fac, float* __restrict__ lse2 = nullptr,
                                                   float* __restrict__ lse_nat = nullptr) {
  // lse2 / lse_nat (the merging update's form of the step: no merge<Q> launch): the row's log-sum-exp in binary and
  // natural units, as merge<Q> leaves them.  (Launched only when pass C takes the general form everywhere --
  // ESR_IB2H_PC=general --, else facscale2h_kernel does this and the scaled copy of Q in one launch.)
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= B) return;
  float pm[8], pl[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    pm[s] = -INFINITY; pl[s] = 0.f;
    if (s < nsplit) {
      pm[s] = part_m[(int64_t)s * B + row];
      pl[s] = part_l[(int64_t)s * B + row];
    }
  }
  float M = pm[0], L = 0.f;
  float wt[8];
#pragma unroll
  for (int s = 1; s < 8; ++s) M = fmaxf(M, pm[s]);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    wt[s] = pm[s] == M ? 1.f : __builtin_amdgcn_exp2f(pm[s] - M);
    L = __fmaf_rn(pl[s], wt[s], L);  // (explicit, as in merge_row: the two must round alike)
  }
  const float invL1 = __fdiv_rn(1.0f, L);
#pragma unroll
  for (int s = 0; s < 8; ++s)
    if (s < nsplit) fac[(int64_t)s * B + row] = __fmul_rn(__fmul_rn(invL1, invl_scale), wt[s]);
  if (lse2) {
    const float l2v = __fadd_rn(M, __builtin_amdgcn_logf(L));
    lse2[row] = l2v;
    if (lse_nat) lse_nat[row] = l2v * k3Ln2;
  }
}

// Factors AND the scaled copy of Q for inbatch2h_pct_kernel in one launch (round 6).  One 256-thread block per 32-row
// chunk, eight lanes per row:
//   * lane s of a row reads split s's reference and normaliser, the eight pairs are passed round by shuffles and every
//     lane runs fac2h_kernel's arithmetic (same operations, same order: the factors, lse and the merges' normalisers agree
//     bit for bit): fac[s][i] = f_i w_s, f_i = 2^14 / l_i ;
//   * x' = (f_i q_i) 2^E in two fp16 planes (Yt) with the FIXED exponent E = eq - 11, eq the exponent of Q's own planes
//     (2^eq max |q| in [2^13, 2^14)): the optimistic reference puts the row's best candidate at p' = 16, so l_i >= 16 (1 -
//     2^-10) and f_i <= 2^10 (1 + 2^-9) -- max |x'| < 2^13.01, no overflow whatever the batch holds, and no grid-wide
//     maximum to wait for (the first form of this step took the copy's exponent from its largest element: a second
//     launch behind an atomic maximum, 5 us).  Pass C's partial rows are 2^(E + 14) sum p q: one multiplication by 2^11
//     brings them to the unit the merges undo;
//   * cflags[s] != 0: pass C's workgroups on pass-Q split s must take the general form, because
//       - some row's factor for split s is not f_i (a pass-Q workgroup of that split redid itself against its own
//         maximum: its probabilities carry another reference), or
//       - (every split) some row with a non-zero factor has its largest |x'| under 2^-4 -- more than 17 binades under the
//         largest possible element: its second plane would lose bits to fp16's subnormals (a row whose normaliser is
//         beyond ~2^20: the reference's 33 samples all lie 2^16 under the bulk of its scores).
constexpr int kCtExpShift = 11;
__global__ __launch_bounds__(256) void facscale2h_kernel(RowSrc X, int64_t B, int nsplit,
                                                        const float* __restrict__ part_m,
                                                        const float* __restrict__ part_l, float invl_scale,
                                                        const float* __restrict__ sc, float* __restrict__ fac,
                                                        float* __restrict__ lse2, float* __restrict__ lse_nat,
                                                        _Float16* __restrict__ Yt, int* __restrict__ cflags) {
  const int t = threadIdx.x, chunk = blockIdx.x;
  const int row = t >> 3, sl = t & 7, d0 = sl * 16;
  const int64_t grow = (int64_t)chunk * 32 + row;
  float v[16];
  {
    const int64_t r = X.idx ? (int64_t)X.idx[grow] : grow;
    RowSrc Y0 = X;
    Y0.idx = nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 f = rowsrc_load4(Y0, r, d0 + 4 * q);
      v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
    }
  }
  const float pm_l = sl < nsplit ? part_m[(int64_t)sl * B + grow] : -INFINITY;
  const float pl_l = sl < nsplit ? part_l[(int64_t)sl * B + grow] : 0.f;
  float pm[8], pl[8];
  const int lane0 = (t & 63) & ~7;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    pm[s] = __shfl(pm_l, lane0 + s, 64);
    pl[s] = __shfl(pl_l, lane0 + s, 64);
  }
  float M = pm[0], L = 0.f;
  float wt[8];
#pragma unroll
  for (int s = 1; s < 8; ++s) M = fmaxf(M, pm[s]);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    wt[s] = pm[s] == M ? 1.f : __builtin_amdgcn_exp2f(pm[s] - M);
    L = __fmaf_rn(pl[s], wt[s], L);  // (explicit, as in merge_row and fac2h_kernel: they must round alike)
  }
  const float invL1 = __fdiv_rn(1.0f, L);
  const float f1 = __fmul_rn(invL1, invl_scale);
  float wmine = 1.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) wmine = s == sl ? wt[s] : wmine;
  const float fs = __fmul_rn(f1, wmine);
  if (sl < nsplit) {
    fac[(int64_t)sl * B + grow] = fs;
    if (fs != f1) cflags[sl] = 1;  // (a NaN factor flags itself)
  }
  if (sl == 0 && lse2) {
    const float l2v = __fadd_rn(M, __builtin_amdgcn_logf(L));
    lse2[grow] = l2v;
    if (lse_nat) lse_nat[grow] = l2v * k3Ln2;
  }
  const float mul = sc[3] * ldexpf(1.f, -kCtExpShift);  // 2^(eq - 11)
  float m = 0.f;
  f16x8 p[2][2];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float xs = __fmul_rn(v[e], f1) * mul;  // (the scaling is exact: a power of two)
    m = fmaxf(m, fabsf(xs));
    const _Float16 a = (_Float16)xs;
    const _Float16 b = (_Float16)(xs - (float)a);
    p[0][e >> 3][e & 7] = a;
    p[1][e >> 3][e & 7] = b;
  }
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  m = fmaxf(m, __shfl_xor(m, 4, 64));
  if (sl < nsplit && (!(m < 60000.f) || (f1 != 0.f && m < kCtLive))) cflags[sl] = 1;
#pragma unroll
  for (int pl2 = 0; pl2 < 2; ++pl2) {
    f16x8* dst = reinterpret_cast<f16x8*>(Yt + ((int64_t)pl2 * B + grow) * k3D + d0);
    dst[0] = p[pl2][0];
    dst[1] = p[pl2][1];
  }
}

struct InbatchHWs {
  _Float16 *Qh, *Ch;
  float *part_O, *part_O2, *fac_side, *Qcopy, *Ccopy, *part_m, *part_mr, *part_l, *lse2, *fac, *Pmat, *nrm, *amax, *sc, *diag;
  int* flags;
  unsigned long long* loss_acc;
  unsigned long long* ent;  // prepsplit2h_kernel's tagged per-chunk maxima
  _Float16* Qt;             // the scaled copy of Q (facscale2h_kernel): [2][B][128]
  unsigned* czero;          // 16 words zeroed by prepsplit2h_kernel, of which
  int* cflags;              // [8, 16) the pass-Q splits' flags of pass C's form
};
constexpr int64_t kHMaxB = 16384;  // B x B x 4 bytes of stored probabilities: 1 GiB

static size_t inbatch2h_ws_layout(int64_t B, char* base, InbatchHWs* ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  const size_t planes = (size_t)2 * B * k3D * 2;
  InbatchHWs w;
  w.Qh = (_Float16*)take(planes);
  w.Ch = (_Float16*)take(planes);
  w.part_O = (float*)take((size_t)8 * B * k3D * 4);
  w.part_O2 = (float*)take((size_t)8 * B * k3D * 4);  // pass C's partials when merge<Q> runs beside pass C (overlapped form)
  w.fac_side = (float*)take((size_t)8 * B * 4);       // merge<Q>'s own copy of the factors there (pass C reads fac2h_kernel's)
  w.Qcopy = (float*)take((size_t)B * k3D * 4);        // the gathered rows as f32 matrices (overlapped train step: the merges
  w.Ccopy = (float*)take((size_t)B * k3D * 4);        // read them instead of tables that are being updated beside them)
  w.part_m = (float*)take((size_t)8 * B * 4);
  w.part_mr = (float*)take((size_t)8 * B * 4);
  w.part_l = (float*)take((size_t)8 * B * 4);
  w.lse2 = (float*)take((size_t)B * 4);
  w.fac = (float*)take((size_t)8 * B * 4);
  w.diag = (float*)take((size_t)B * 4);
  w.flags = (int*)take((size_t)8 * (B / k3Owned) * sizeof(int));
  w.Pmat = (float*)take((size_t)B * B * 4);
  w.loss_acc = (unsigned long long*)take(sizeof(unsigned long long) * 16 * (1 + kLossWords));
  w.nrm = (float*)take((size_t)2 * (B / k3Chunk) * 4 * sizeof(float));
  w.amax = (float*)take((size_t)2 * (B / k3Chunk) * sizeof(float));
  w.sc = (float*)take(kHScaleWords * sizeof(float));
  w.ent = (unsigned long long*)take((size_t)(B / k3Chunk + 1) * sizeof(unsigned long long));  // (+ the maxima's word)
  w.Qt = (_Float16*)take(planes);
  w.czero = (unsigned*)take(16 * sizeof(unsigned));
  w.cflags = reinterpret_cast<int*>(w.czero ? w.czero + 8 : nullptr);
  if (ws) *ws = w;
  return off;
}

// largest split count <= 8 that divides the chunk count and keeps the grid near `per_cu` workgroups per CU
static int inbatch2h_nsplit(int64_t B, int per_cu) {
  const int64_t owned_blocks = B / k3Owned, nchunks = B / k3Chunk;
  int best = 1;
  for (int s = 1; s <= 8; ++s)
    if (nchunks % s == 0 && owned_blocks * s <= 320 * per_cu) best = s;
  return best;
}

// the sparse-Adagrad tail of a whole train step (esr_inbatch_train_step_f16x2): both towers as one virtual table, the
// occurrence list [query ids ; Vq + candidate ids] sorted by virtual row, the gradient rows [gQ ; gC] in occurrence order
struct InbatchUpdate {
  void* tables[2];
  float* accums[2];
  int64_t row_offsets[3];
  int dtype;
  const int32_t* sorted_vids;
  const int32_t* perm;
  float* grad_rows;
  float lr, eps;
  bool skip_long;
};

// in esr_optim.hip
int inbatch_merge_update(void* const* tables, float* const* accums, const int64_t* row_offsets, int dtype,
                         const int32_t* sorted_vids, const int32_t* perm, const InbatchMergeArgs& a, float lr, float eps,
                         hipStream_t st);
int sparse_adagrad_range(void* const* tables, float* const* accums, const int64_t* row_offsets, int ntables, int dtype,
                         int D, const int32_t* sorted_vids, const int32_t* perm, int64_t n, float* grad_rows, float lr,
                         float eps, bool skip_long, hipStream_t st);

// fork / join events of the overlapped form: one pair per host thread and device, made on first use (an event may be
// recorded again while an earlier wait on it is still queued: a wait binds to the record that preceded it)
static bool inbatch_events(hipEvent_t* fork, hipEvent_t* join) {
  constexpr int kMaxDev = 16;
  static thread_local hipEvent_t ev[kMaxDev][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return false;
  for (int i = 0; i < 2; ++i)
    if (!ev[dev][i] && hipEventCreateWithFlags(&ev[dev][i], hipEventDisableTiming) != hipSuccess) return false;
  *fork = ev[dev][0];
  *join = ev[dev][1];
  return true;
}

}  // namespace esr

using namespace esr;

extern "C" {

#ifdef H_TIMING
int esr_ib2h_debug_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(esr_ib2h_dbg), sizeof(unsigned long long) * 8192);
}
#endif

size_t esr_inbatch2h_workspace_bytes(int64_t B, int D) {
  (void)D;
  if (B <= 0 || B > kHMaxB) return 256;
  return inbatch2h_ws_layout(B, nullptr, nullptr);
}

static int inbatch2h_run(const char* who, RowSrc Qs, RowSrc Cs, const int32_t* gq_rows, const int32_t* gc_rows, int64_t B,
                         int D, float scale, float regularization, float batch_size, float* loss, float* lse,
                         float* gQ, float* gC, void* workspace, size_t workspace_bytes, esr_stream_t stream,
                         hipStream_t side = nullptr, const InbatchUpdate* upd = nullptr) {
  if (!(B > 0 && B % k3Owned == 0 && B <= kHMaxB)) {
    set_error("%s: B=%lld must be a positive multiple of 128, at most %lld (larger batches: the bf16x3 entry point)",
              who, (long long)B, (long long)kHMaxB);
    return ESR_EINVAL;
  }
  if (!(D > 0 && D <= k3D && D % 4 == 0)) {  // narrower rows run in the 128-column tile, zero-padded
    set_error("%s: D=%d not supported (a multiple of 4, at most 128; wider rows: the f32 entry point)", who, D);
    return ESR_EINVAL;
  }
  if (!(Qs.base && Cs.base && loss && gQ && gC)) {
    set_error("%s: null pointer", who);
    return ESR_EINVAL;
  }
  if (batch_size == 0.f) {
    set_error("%s: batch_size must be non-zero", who);
    return ESR_EINVAL;
  }
  if ((((uintptr_t)Qs.base | (uintptr_t)Cs.base | (uintptr_t)gQ | (uintptr_t)gC) & 15) != 0) {
    set_error("%s: matrices must be 16-byte aligned", who);
    return ESR_EINVAL;
  }
  if (!workspace || workspace_bytes < esr_inbatch2h_workspace_bytes(B, D) || ((uintptr_t)workspace & 15)) {
    set_error("%s: workspace %zu bytes < %zu required (or misaligned)", who, workspace_bytes,
              esr_inbatch2h_workspace_bytes(B, D));
    return ESR_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  InbatchHWs ws;
  inbatch2h_ws_layout(B, (char*)workspace, &ws);
  const float inv_bs = 1.0f / batch_size, sl2 = scale * k3Log2e;
  const int nchunks = (int)(B / k3Chunk);
  const char* qcs = getenv("ESR_IB2H_Q_PER_CU");
  const int nsplit_q = inbatch2h_nsplit(B, qcs ? std::max(1, atoi(qcs)) : 2);  // 252 registers, 50 KB of LDS: two per CU
  // pass C: 8-wave workgroups, 256 owned rows each
  const int pc_blocks = (int)cdiv(B, kCtOwned);
  int nsplit_c = 1;
  for (int sp = 1; sp <= 8; ++sp)
    if (nchunks % sp == 0 && pc_blocks * sp <= 320) nsplit_c = sp;
  const int grid_c = pc_blocks * nsplit_c;
  const int grid_q = (int)(B / k3Owned) * nsplit_q;
  const int mgrid = (int)std::min<int64_t>(k3MergeBlocks, cdiv(B, kBlock / 32));
  // exponent reference of pass Q: optimistic + redo launch (default), or the row-max pass (ESR_IB2H_REF=rowmax);
  // ESR_IB2H_REF=redo forces every block through the redo launch (test hook)
  const char* refe = getenv("ESR_IB2H_REF");
  const int mode = (refe && refe[0] == 'r' && refe[1] == 'o') ? 0 : ((refe && refe[0] == 'r' && refe[1] == 'e') ? 2 : 1);
  // round 4 (default): prep + split in one launch, pass Q redoes overflowed workgroups itself: five launches.
  // ESR_IB2H_FUSED=0 (or the row-max reference) keeps round 3's seven.
  const char* fe = getenv("ESR_IB2H_FUSED");
  const bool fused = !(fe && fe[0] == '0') && mode != 0;
  int nsplit_r = 1;
  // Overlapped form (`side` given; round 5): pass C needs nothing of merge<Q> but the factors, so fac2h_kernel makes
  // those (2 us) and merge<Q> -- 14 us of partial-O reads at B = 8192 -- runs on `side` beside pass C, followed there by
  // the query tower's Adagrad update when this is a whole train step; pass C then writes its partials to a second buffer,
  // and both merges read the rows from the f32 copies prepsplit2h_kernel left (merge<C> needs the query tower's OLD rows
  // after the side stream may have updated them, and vice versa).  Same kernels, same arithmetic: results are
  // bit-identical to the sequential form (the loss is an order-free integer sum).
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // bf16 tables (both towers): the one-plane kernels (ESR_IB2H_BF16=two keeps two planes, whose second is all zero)
  const char* b16e = getenv("ESR_IB2H_BF16");
  // (ESR_IB2H_BF16=force: timing probes run the one-plane kernels on any input -- values then lack the second plane)
  const bool one_plane = fused && ((Qs.bf16 && Cs.bf16 && !(b16e && b16e[0] == 't')) || (b16e && b16e[0] == 'f'));
  const bool overlapped = fused && !one_plane && side != nullptr && side != st && inbatch_events(&ev_fork, &ev_join);
  // The merging update (round 5; ESR_IB2H_MERGE_UPDATE=0: off): a train step whose occurrence list has no run that
  // outgrows its head chunk ends with ONE launch that merges the partial O rows of both passes on the fly and applies
  // Adagrad (esr_optim.hip inbatch_merge_update_kernel) -- no merge launches, no gradient rows in memory; merge<Q>'s
  // other products (lse, pass C's factors) come from fac2h_kernel.  Same bits as merge + update.
  const char* mue = getenv("ESR_IB2H_MERGE_UPDATE");
  const bool merge_upd = fused && !overlapped && upd != nullptr && upd->skip_long && D == k3D && Qs.ld == k3D &&
                         Cs.ld == k3D && !(mue && mue[0] == '0');
  auto merging_update = [&](int nsplit_of_c, const float* part_O_of_c) -> int {
    InbatchMergeArgs a;
    a.part_O[0] = ws.part_O; a.part_O[1] = part_O_of_c;
    a.nsplit[0] = nsplit_q; a.nsplit[1] = nsplit_of_c;
    a.oscale[0] = ws.sc + 1; a.oscale[1] = ws.sc + 2;
    a.partner[0] = ws.Ccopy; a.partner[1] = ws.Qcopy;
    a.part_m = ws.part_m; a.part_l = ws.part_l;
    a.B = B;
    a.scale = scale; a.lam = regularization; a.inv_bs = inv_bs;
    a.loss_acc = ws.loss_acc; a.loss_scale = 1.0 / (double)batch_size; a.loss_out = loss;
    a.zero_words = ws.ent; a.nzero = nchunks + 1;
    return inbatch_merge_update(upd->tables, upd->accums, upd->row_offsets, upd->dtype, upd->sorted_vids, upd->perm, a,
                                upd->lr, upd->eps, st);
  };
  if (fused) {
    static std::atomic<unsigned long long> call_seq{0};
    static const unsigned long long seed =
        (unsigned long long)(uintptr_t)&call_seq ^ (unsigned long long)time(nullptr) * 0x9E3779B97F4A7C15ull;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (call_seq.fetch_add(1) + 1);  // splitmix64: a token per call
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    const unsigned long long token = ((z ^ (z >> 31)) >> 16) | 1ull;  // 48 bits, never 0
    const char* tle = getenv("ESR_IB2H_POLL");  // "flat": every workgroup polls every chunk's word (rounds 4 - 5)
    const int two_level = (tle && tle[0] == 'f') ? 0 : 1;
    ESR_KT("prepsplit2h_kernel", st,
           hipLaunchKernelGGL(prepsplit2h_kernel, dim3(nchunks), dim3(256), 0, st, Qs, Cs, B, ws.Qh, ws.Ch, ws.ent, token,
                              ws.diag, ws.nrm, ws.sc, ws.loss_acc, ws.flags, grid_q, (overlapped || merge_upd) ? ws.Qcopy : (float*)nullptr,
                              (overlapped || merge_upd) ? ws.Ccopy : (float*)nullptr, ws.czero, two_level));
    if (one_plane) {
      // bf16 tables: one fp16 plane per operand, S^T recomputed by pass C -- six GEMMs, no stored probabilities.
      // 512-thread workgroups of 256 owned rows, one per CU
      const int blocks1 = (int)cdiv(B, kH1Owned);
      int nsplit_1 = 1;
      for (int sp = 1; sp <= 8; ++sp)
        if (nchunks % sp == 0 && blocks1 * sp <= 320) nsplit_1 = sp;
      const int nsplit_q = nsplit_1, grid_q = blocks1 * nsplit_1;  // (shadow the two-plane geometry)
      const char* dbg1h = getenv("ESR_IB1H_DBG");
      if (dbg1h && dbg1h[0] == '1') {
        hipLaunchKernelGGL((inbatch1h_kernel<true, 1>), dim3(grid_q), dim3(512), 0, st, (const _Float16*)ws.Qh,
                           (const _Float16*)ws.Ch, B, nsplit_q, sl2, (const float*)ws.sc, (const float*)ws.diag,
                           (const float*)nullptr, mode, ws.part_m, ws.part_O, ws.part_l);
      } else
      ESR_KT("inbatch1h_kernel_q", st,
             hipLaunchKernelGGL((inbatch1h_kernel<true>), dim3(grid_q), dim3(512), 0, st, (const _Float16*)ws.Qh,
                                (const _Float16*)ws.Ch, B, nsplit_q, sl2, (const float*)ws.sc, (const float*)ws.diag,
                                (const float*)nullptr, mode, ws.part_m, ws.part_O, ws.part_l));
      if (merge_upd) {
        ESR_KT("fac2h_kernel", st,
               hipLaunchKernelGGL(fac2h_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, B, nsplit_q,
                                  (const float*)ws.part_m, (const float*)ws.part_l, ldexpf(1.f, (int)kHPexp), ws.fac,
                                  ws.lse2, lse));
      } else
      ESR_KT("inbatch3_merge_kernel_q", st,
             hipLaunchKernelGGL((inbatch3_merge_kernel<true>), dim3(mgrid), dim3(kBlock), 0, st, Qs, Cs, gq_rows, B,
                                nsplit_q, (const float*)ws.part_O, (const float*)ws.part_m, (const float*)ws.part_l, scale,
                                regularization, inv_bs, ws.lse2, lse, gQ, ws.loss_acc, 1.0 / (double)batch_size, loss,
                                (float*)nullptr, (const float*)(ws.sc + 1), ldexpf(1.f, (int)kHPexp), ws.fac));
      float* const part_O_1c = merge_upd ? ws.part_O2 : ws.part_O;  // (the merging update still needs pass Q's rows)
      if (dbg1h && (dbg1h[0] == '1' || dbg1h[0] == '2')) {
        hipLaunchKernelGGL((inbatch1h_kernel<false, 1>), dim3(grid_q), dim3(512), 0, st, (const _Float16*)ws.Ch,
                           (const _Float16*)ws.Qh, B, nsplit_q, sl2, (const float*)ws.sc, (const float*)nullptr,
                           (const float*)ws.lse2, mode, (float*)nullptr, part_O_1c, (float*)nullptr);
      } else
      ESR_KT("inbatch1h_kernel_c", st,
             hipLaunchKernelGGL((inbatch1h_kernel<false>), dim3(grid_q), dim3(512), 0, st, (const _Float16*)ws.Ch,
                                (const _Float16*)ws.Qh, B, nsplit_q, sl2, (const float*)ws.sc, (const float*)nullptr,
                                (const float*)ws.lse2, mode, (float*)nullptr, part_O_1c, (float*)nullptr));
      if (merge_upd) return merging_update(nsplit_q, part_O_1c);
      ESR_KT("inbatch3_merge_kernel_c", st,
             hipLaunchKernelGGL((inbatch3_merge_kernel<false>), dim3(mgrid), dim3(kBlock), 0, st, Cs, Qs, gc_rows, B,
                                nsplit_q, (const float*)ws.part_O, (const float*)ws.part_m, (const float*)ws.part_l, scale,
                                regularization, inv_bs, ws.lse2, (float*)nullptr, gC, ws.loss_acc,
                                1.0 / (double)batch_size, loss, (float*)nullptr, (const float*)(ws.sc + 2), 1.0f,
                                (float*)nullptr, ws.ent, nchunks + 1));
      if (upd) {
        const int rc = sparse_adagrad_range(upd->tables, upd->accums, upd->row_offsets, 2, upd->dtype, D, upd->sorted_vids,
                                            upd->perm, 2 * B, upd->grad_rows, upd->lr, upd->eps, upd->skip_long, st);
        if (rc != ESR_OK) return rc;
      }
      return check_launch(who);
    }
    ESR_KT("inbatch2h_q_kernel", st,
           hipLaunchKernelGGL((inbatch2h_q_kernel<2>), dim3(grid_q), dim3(256), 0, st, (const _Float16*)ws.Qh,
                              (const _Float16*)ws.Ch, B, nsplit_q, sl2, (const float*)ws.sc, (const float*)ws.part_mr, 1,
                              (const float*)ws.diag, mode, ws.flags, ws.part_m, ws.part_O, ws.part_l, ws.Pmat));
  } else {
  ESR_KT("prep2h_kernel", st, hipLaunchKernelGGL(prep2h_kernel, dim3(nchunks), dim3(256), 0, st, Qs, Cs, ws.amax, ws.diag));
  ESR_KT("split2h_kernel", st,
         hipLaunchKernelGGL(split2h_kernel, dim3(nchunks, 2), dim3(256), 0, st, Qs, Cs, B, ws.Qh, ws.Ch,
                            (const float*)ws.amax, ws.nrm, ws.sc, ws.loss_acc, ws.flags, grid_q));
  if (mode == 0) {
    const int rm_blocks = (int)cdiv(B, kHRmOwned);
    for (int sp = 1; sp <= 8; ++sp)
      if (nchunks % sp == 0 && rm_blocks * sp <= 320) nsplit_r = sp;
    ESR_KT("rowmax2h_kernel", st,
           hipLaunchKernelGGL(rowmax2h_kernel, dim3(rm_blocks * nsplit_r), dim3(256), 0, st, (const _Float16*)ws.Qh,
                              (const _Float16*)ws.Ch, B, nsplit_r, sl2, (const float*)ws.nrm, (const float*)ws.sc,
                              ws.part_mr));
  }
  {
    ESR_KT("inbatch2h_q_kernel", st,
           hipLaunchKernelGGL((inbatch2h_q_kernel<0>), dim3(grid_q), dim3(256), 0, st, (const _Float16*)ws.Qh,
                              (const _Float16*)ws.Ch, B, nsplit_q, sl2, (const float*)ws.sc, (const float*)ws.part_mr,
                              nsplit_r, (const float*)ws.diag, mode, ws.flags, ws.part_m, ws.part_O, ws.part_l, ws.Pmat));
    if (mode != 0)  // redo launch: workgroups whose block is not flagged (normally all of them) exit after one load
      ESR_KT("inbatch2h_q_kernel_redo", st,
             hipLaunchKernelGGL((inbatch2h_q_kernel<1>), dim3(grid_q), dim3(256), 0, st, (const _Float16*)ws.Qh,
                                (const _Float16*)ws.Ch, B, nsplit_q, sl2, (const float*)ws.sc, (const float*)ws.part_mr,
                                nsplit_r, (const float*)ws.diag, mode, ws.flags, ws.part_m, ws.part_O, ws.part_l, ws.Pmat));
  }
  }  // !fused
  // O_Q' = 2^ec sum p' c, l' = sum p': o / l needs 2^-ec (sc[1]); the stored factors carry pass C's 2^14
  hipStream_t st_q = overlapped ? side : st;
  float* const part_O_c = (overlapped || merge_upd) ? ws.part_O2 : ws.part_O;
  // the rows the merges read: the towers themselves, or (overlapped) their gathered copies
  const RowSrc Qm = overlapped ? RowSrc{ws.Qcopy, nullptr, 0, Qs.ld} : Qs;
  const RowSrc Cm = overlapped ? RowSrc{ws.Ccopy, nullptr, 0, Cs.ld} : Cs;
  // Pass C's form (round 6): the scaled copy of Q (facscale2h_kernel); with ESR_IB2H_PC=general (the test
  // hook) and on the unfused path every workgroup takes the general form (unscaled planes, factors applied to the B
  // fragments in registers).
  const int nc_q = nchunks / nsplit_q;
  const char* pcm = getenv("ESR_IB2H_PC");
  const int force_general = (!fused || (pcm && pcm[0] == 'g')) ? 1 : 0;
  if (overlapped) {
    if (hipEventRecord(ev_fork, st) != hipSuccess || hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess) {
      set_error("%s: could not fork the side stream", who);
      return ESR_ELAUNCH;
    }
  }
  if (!force_general)
    ESR_KT("facscale2h_kernel", st,
           hipLaunchKernelGGL(facscale2h_kernel, dim3(nchunks), dim3(256), 0, st,
                              (overlapped || merge_upd) ? RowSrc{ws.Qcopy, nullptr, 0, Qs.ld} : Qs, B, nsplit_q,
                              (const float*)ws.part_m, (const float*)ws.part_l, ldexpf(1.f, (int)kHPexp),
                              (const float*)ws.sc, ws.fac, merge_upd ? ws.lse2 : (float*)nullptr,
                              merge_upd ? lse : (float*)nullptr, ws.Qt, ws.cflags));
  else if (overlapped || merge_upd)
    ESR_KT("fac2h_kernel", st,
           hipLaunchKernelGGL(fac2h_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, B, nsplit_q,
                              (const float*)ws.part_m, (const float*)ws.part_l, ldexpf(1.f, (int)kHPexp), ws.fac,
                              merge_upd ? ws.lse2 : (float*)nullptr, merge_upd ? lse : (float*)nullptr));
  if (!merge_upd)
  ESR_KT("inbatch3_merge_kernel_q", st_q,
         hipLaunchKernelGGL((inbatch3_merge_kernel<true>), dim3(mgrid), dim3(kBlock), 0, st_q, Qm, Cm, gq_rows, B, nsplit_q,
                            (const float*)ws.part_O, (const float*)ws.part_m, (const float*)ws.part_l, scale,
                            regularization, inv_bs, ws.lse2, lse, gQ, ws.loss_acc, 1.0 / (double)batch_size, loss,
                            (float*)nullptr, (const float*)(ws.sc + 1), ldexpf(1.f, (int)kHPexp),
                            overlapped ? ws.fac_side : ws.fac));
  if (overlapped) {
    if (upd) {  // the query tower's rows: positions [0, B) of the sorted occurrence list
      const int rc = sparse_adagrad_range(upd->tables, upd->accums, upd->row_offsets, 2, upd->dtype, D, upd->sorted_vids,
                                          upd->perm, B, upd->grad_rows, upd->lr, upd->eps, upd->skip_long, side);
      if (rc != ESR_OK) return rc;
    }
    if (hipEventRecord(ev_join, side) != hipSuccess) {
      set_error("%s: could not record the join event", who);
      return ESR_ELAUNCH;
    }
  }
  ESR_KT("inbatch2h_pct_kernel", st,
         hipLaunchKernelGGL(inbatch2h_pct_kernel, dim3(grid_c), dim3(512), 0, st, (const _Float16*)ws.Qt,
                            (const _Float16*)ws.Qh, B, nsplit_c, (const float*)ws.fac, nc_q, (const char*)ws.Pmat,
                            ldexpf(1.f, kCtExpShift), (const int*)ws.cflags, force_general, part_O_c));
  if (merge_upd) return merging_update(nsplit_c, part_O_c);
  // O_C' = 2^(eq + 14) sum_i (p_ij / l_i) q_i
  ESR_KT("inbatch3_merge_kernel_c", st,
         hipLaunchKernelGGL((inbatch3_merge_kernel<false>), dim3(mgrid), dim3(kBlock), 0, st, Cm, Qm, gc_rows, B, nsplit_c,
                            (const float*)part_O_c, (const float*)ws.part_m, (const float*)ws.part_l, scale,
                            regularization, inv_bs, ws.lse2, (float*)nullptr, gC, ws.loss_acc,
                            1.0 / (double)batch_size, loss, (float*)nullptr, (const float*)(ws.sc + 2), 1.0f,
                            (float*)nullptr, fused ? ws.ent : (unsigned long long*)nullptr, nchunks + 1));
  if (upd) {
    // the candidate tower's rows: positions [B, 2B) (sequential form: the whole list, one launch for both towers)
    const int64_t off = overlapped ? B : 0;
    const int rc = sparse_adagrad_range(upd->tables, upd->accums, upd->row_offsets, 2, upd->dtype, D,
                                        upd->sorted_vids + off, upd->perm + off, 2 * B - off, upd->grad_rows, upd->lr,
                                        upd->eps, upd->skip_long, st);
    if (rc != ESR_OK) return rc;
  }
  if (overlapped && hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) {
    set_error("%s: could not join the side stream", who);
    return ESR_ELAUNCH;
  }
  return check_launch(who);
}

int esr_inbatch_softmax_fwd_bwd_f16x2(const float* Q, const float* C, int64_t B, int D, float scale,
                                      float regularization, float batch_size, float* loss, float* lse, float* gQ,
                                      float* gC, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_inbatch_softmax_fwd_bwd_f16x2");
  return inbatch2h_run("esr_inbatch_softmax_fwd_bwd_f16x2", RowSrc{Q, nullptr, 0, D}, RowSrc{C, nullptr, 0, D}, nullptr,
                       nullptr, B, D, scale, regularization, batch_size, loss, lse, gQ, gC, workspace, workspace_bytes,
                       stream);
}

int esr_inbatch_towers_fwd_bwd_f16x2(const void* query_table, int64_t Vq, const void* cand_table, int64_t Vc, int dtype,
                                     int D, const int32_t* query_ids, const int32_t* cand_ids, const int32_t* gq_rows,
                                     const int32_t* gc_rows, int64_t B, float scale, float regularization,
                                     float batch_size, float* loss, float* lse, float* gQ, float* gC, void* workspace,
                                     size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_inbatch_towers_fwd_bwd_f16x2");
  ESR_REQUIRE(Vq > 0 && Vc > 0 && query_ids && cand_ids, "esr_inbatch_towers_fwd_bwd_f16x2: bad tables / ids");
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "esr_inbatch_towers_fwd_bwd_f16x2: bad dtype %d", dtype);
  return inbatch2h_run("esr_inbatch_towers_fwd_bwd_f16x2", RowSrc{query_table, query_ids, dtype == ESR_BF16, D},
                       RowSrc{cand_table, cand_ids, dtype == ESR_BF16, D}, gq_rows, gc_rows, B, D, scale, regularization,
                       batch_size, loss, lse, gQ, gC, workspace, workspace_bytes, stream);
}

int esr_inbatch2h_pass_c_forms(const void* workspace, size_t workspace_bytes, int64_t B, int32_t* forms,
                               esr_stream_t stream) {
  ESR_REQUIRE(workspace && forms && B > 0 && B % k3Owned == 0 && B <= kHMaxB &&
                  workspace_bytes >= esr_inbatch2h_workspace_bytes(B, k3D),
              "esr_inbatch2h_pass_c_forms: bad workspace / B");
  InbatchHWs ws;
  inbatch2h_ws_layout(B, (char*)const_cast<void*>(workspace), &ws);
  if (hipMemcpyAsync(forms, ws.cflags, 8 * sizeof(int32_t), hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess) {
    set_error("esr_inbatch2h_pass_c_forms: copy failed");
    return ESR_ELAUNCH;
  }
  return ESR_OK;
}

// ---- the whole in-batch training step as ONE call (round 5) ---------------------------------------------------------------
static size_t inbatch_step_ws_layout(int64_t B, int D, char* base, char** head, float** grads, int32_t** sorted,
                                     int32_t** perm, char** sort_ws, size_t* sort_ws_bytes) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  const size_t hb = esr_inbatch2h_workspace_bytes(B, D);
  const size_t sb = esr_segment_sort_workspace_bytes(2 * B);
  char* h = take(hb);
  float* g = (float*)take((size_t)2 * B * D * sizeof(float));
  int32_t* so = (int32_t*)take((size_t)2 * B * sizeof(int32_t));
  int32_t* pe = (int32_t*)take((size_t)2 * B * sizeof(int32_t));
  char* sw = take(sb);
  if (head) *head = h;
  if (grads) *grads = g;
  if (sorted) *sorted = so;
  if (perm) *perm = pe;
  if (sort_ws) *sort_ws = sw;
  if (sort_ws_bytes) *sort_ws_bytes = sb;
  return off;
}

size_t esr_inbatch_train_step_workspace_bytes(int64_t B, int D) {
  if (B <= 0 || B > kHMaxB || D <= 0) return 256;
  return inbatch_step_ws_layout(B, D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int esr_inbatch_train_step_f16x2(void* query_table, float* query_accum, int64_t Vq, void* cand_table, float* cand_accum,
                                 int64_t Vc, int dtype, int D, const int32_t* query_ids, const int32_t* cand_ids,
                                 int64_t B, float scale, float regularization, float batch_size, float lr, float eps,
                                 const int32_t* presorted_vids, const int32_t* presorted_perm, int long_runs, float* loss,
                                 float* lse, void* workspace, size_t workspace_bytes, esr_stream_t stream,
                                 esr_stream_t side_stream) {
  TraceScope trace_scope_("esr_inbatch_train_step_f16x2");
  const char* who = "esr_inbatch_train_step_f16x2";
  ESR_REQUIRE(Vq > 0 && Vc > 0 && query_table && cand_table && query_accum && cand_accum && query_ids && cand_ids,
              "%s: bad tables / ids", who);
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16, "%s: bad dtype %d", who, dtype);
  ESR_REQUIRE(Vq + Vc < ((int64_t)1 << 31), "%s: %lld virtual rows >= 2^31", who, (long long)(Vq + Vc));
  ESR_REQUIRE((presorted_vids == nullptr) == (presorted_perm == nullptr), "%s: presorted ids and perm come together", who);
  ESR_REQUIRE(B > 0 && B % k3Owned == 0 && B <= kHMaxB, "%s: B=%lld must be a positive multiple of 128, at most %lld", who,
              (long long)B, (long long)kHMaxB);
  ESR_REQUIRE(workspace && !((uintptr_t)workspace & 255) && workspace_bytes >= esr_inbatch_train_step_workspace_bytes(B, D),
              "%s: workspace %zu bytes < %zu required (or not 256-byte aligned)", who, workspace_bytes,
              esr_inbatch_train_step_workspace_bytes(B, D));
  char *head, *sort_ws;
  float* grads;
  int32_t *sorted, *perm;
  size_t sort_ws_bytes;
  inbatch_step_ws_layout(B, D, (char*)workspace, &head, &grads, &sorted, &perm, &sort_ws, &sort_ws_bytes);
  InbatchUpdate upd;
  upd.tables[0] = query_table; upd.tables[1] = cand_table;
  upd.accums[0] = query_accum; upd.accums[1] = cand_accum;
  upd.row_offsets[0] = 0; upd.row_offsets[1] = Vq; upd.row_offsets[2] = Vq + Vc;
  upd.dtype = dtype;
  upd.grad_rows = grads;
  upd.lr = lr; upd.eps = eps;
  upd.skip_long = long_runs == 0;
  if (presorted_vids) {
    upd.sorted_vids = presorted_vids;
    upd.perm = presorted_perm;
  } else {  // the occurrence list [query ids ; Vq + candidate ids] sorted here (a loop sorts eight batches ahead instead)
    const void* ptrs[2] = {query_ids, cand_ids};
    const int64_t cnt[2] = {B, B}, off[2] = {0, Vq};
    const int rc = esr_segment_sort_ids_multi((const int32_t* const*)ptrs, cnt, off, 2, Vq + Vc, sorted, perm, sort_ws,
                                              sort_ws_bytes, stream);
    if (rc != ESR_OK) return rc;
    upd.sorted_vids = sorted;
    upd.perm = perm;
    upd.skip_long = false;
  }
  // gradient rows in occurrence order: gQ = rows [0, B), gC = rows [B, 2B) of one buffer
  return inbatch2h_run(who, RowSrc{query_table, query_ids, dtype == ESR_BF16, D},
                       RowSrc{cand_table, cand_ids, dtype == ESR_BF16, D}, nullptr, nullptr, B, D, scale, regularization,
                       batch_size, loss, lse, grads, grads + (size_t)B * D, head, esr_inbatch2h_workspace_bytes(B, D), stream,
                       as_stream(side_stream), &upd);
}

}  // extern "C"
