// esr_probe.hip -- measurement probes (not on the hot path): what the matrix pipes of THIS box sustain, so that a
// roofline fraction quoted against the data-sheet peak (MI355X_MICROARCH.md) can be read next to the ceiling a
// register-only MFMA loop reaches under the clocks the part actually holds.
#include "esr_common.h"
#include "../../include/esr_probe.h"

namespace esr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// every wave: `iters` rounds of 4 independent accumulator chains, operands in registers, no memory traffic
__global__ __launch_bounds__(256) void mfma_bf16_probe_kernel(int iters, float* __restrict__ sink) {
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(1.0f + 0.001f * (threadIdx.x & 7));
    b[e] = (__bf16)(0.5f);
  }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (s == 12345.678f) sink[0] = s;  // keeps the chains alive; never true
}

// the same loop with operands that look like data: per-lane pseudo-random bf16 in (-1, 1), four sets rotated so that
// every MFMA sees different A and B registers than the one before it
__device__ __forceinline__ uint32_t probe_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void mfma_bf16_live_probe_kernel(int iters, float* __restrict__ sink) {
  bf16x8 a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t ha = probe_hash((blockIdx.x * 256 + threadIdx.x) * 64 + q * 16 + e);
      const uint32_t hb = probe_hash(ha + 0x9e3779b9u);
      a[q][e] = (__bf16)((float)(int)(ha & 0xffff) * (1.0f / 32768.f) - 1.0f);
      b[q][e] = (__bf16)((float)(int)(hb & 0xffff) * (1.0f / 32768.f) - 1.0f);
    }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[3], c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], b[0], c3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (s == 12345.678f) sink[0] = s;
}

// fp16 planes of the two-plane split paths (esr_inbatch2h.hip, esr_retrieve.hip mode 2): the same loop with
// v_mfma_f32_32x32x16_f16 and full-entropy fp16 operands in (-1, 1)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_f16_live_probe_kernel(int iters, float* __restrict__ sink) {
  f16x8 a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t ha = probe_hash((blockIdx.x * 256 + threadIdx.x) * 64 + q * 16 + e);
      const uint32_t hb = probe_hash(ha + 0x9e3779b9u);
      a[q][e] = (_Float16)((float)(int)(ha & 0xffff) * (1.0f / 32768.f) - 1.0f);
      b[q][e] = (_Float16)((float)(int)(hb & 0xffff) * (1.0f / 32768.f) - 1.0f);
    }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[2], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[3], c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3], b[0], c3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void mfma_f32_probe_kernel(int iters, float* __restrict__ sink) {
  const float a = 1.0f + 0.001f * (threadIdx.x & 7), b = 0.5f;
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (s == 12345.678f) sink[0] = s;
}

// Does the vector ALU run beside the matrix pipe?  Every wave: `iters` rounds of 4 MFMAs (v_mfma_f32_32x32x16_f16, four
// independent chains), each followed by NV plain VALU instructions (v_fma_f32 on eight independent registers) and NT
// transcendentals (v_exp_f32); GROUPED: the four MFMAs first, then the 4 (NV + NT) VALU instructions.  Per-wave cycles
// of the loop go to cycles[global wave].  Launched with 256 or 512 threads per workgroup (one or two waves per SIMD).
// (Round 5: the one-plane in-batch kernel spends as long in ~130 VALU instructions per chunk as in its 24 MFMAs, and the
// probes that removed either shortened it by that part's full length.)
template <int NV, int NT, bool GROUPED>
__global__ __launch_bounds__(512) void mfma_valu_probe_kernel(int iters, unsigned long long* __restrict__ cycles,
                                                              float* __restrict__ sink) {
  f16x8 a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t ha = probe_hash((blockIdx.x * 512 + threadIdx.x) * 64 + q * 16 + e);
      const uint32_t hb = probe_hash(ha + 0x9e3779b9u);
      a[q][e] = (_Float16)((float)(int)(ha & 0xffff) * (1.0f / 32768.f) - 1.0f);
      b[q][e] = (_Float16)((float)(int)(hb & 0xffff) * (1.0f / 32768.f) - 1.0f);
    }
  f32x16 c[4] = {};
  float v[8], x[4];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 1.0f + 0.01f * (float)((threadIdx.x + e) & 15);
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] = 0.001f * (float)((threadIdx.x + e) & 15);
  const float m = 0.999f, d = 0.001f;
#define PROBE_VALU(Q)                                                                                      \
  {                                                                                                        \
    _Pragma("unroll") for (int n_ = 0; n_ < NV; ++n_)                                                      \
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[((Q) * NV + n_) & 7]) : "v"(m), "v"(d));            \
    _Pragma("unroll") for (int n_ = 0; n_ < NT; ++n_)                                                      \
      asm volatile("v_exp_f32 %0, %0" : "+v"(x[((Q) * NT + n_) & 3]));                                     \
  }
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_barrier(0);
      c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q], b[(q + 1) & 3], c[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (!GROUPED) PROBE_VALU(q);
    }
    if (GROUPED) {
#pragma unroll
      for (int q = 0; q < 4; ++q) PROBE_VALU(q);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
#undef PROBE_VALU
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c[0][e] + c[1][e] + c[2][e] + c[3][e];
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e];
#pragma unroll
  for (int e = 0; e < 4; ++e) s += x[e];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// The same question per instruction KIND (four of them behind every MFMA, independent registers unless noted):
//   0 v_fma_f32   1 v_pk_fma_f32   2 v_fma_mix_f32   3 v_cvt_pk_f16_f32   4 v_max3_f32   5 v_pk_add_f32   6 v_exp_f32
//   7 v_fma_f32, all four on ONE register (a dependent chain)   8 v_pk_fma_f32 dependent chain   9 v_exp_f32 -> v_fma_f32 chains
template <int KIND>
__global__ __launch_bounds__(512) void mfma_kind_probe_kernel(int iters, unsigned long long* __restrict__ cycles,
                                                              float* __restrict__ sink) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f16x8 a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t ha = probe_hash((blockIdx.x * 512 + threadIdx.x) * 64 + q * 16 + e);
      const uint32_t hb = probe_hash(ha + 0x9e3779b9u);
      a[q][e] = (_Float16)((float)(int)(ha & 0xffff) * (1.0f / 32768.f) - 1.0f);
      b[q][e] = (_Float16)((float)(int)(hb & 0xffff) * (1.0f / 32768.f) - 1.0f);
    }
  f32x16 c[4] = {};
  f32x2 v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = f32x2{1.0f + 0.01f * (float)((threadIdx.x + e) & 15), 0.5f};
  const f32x2 m = {0.999f, 0.999f}, d = {0.001f, 0.001f};
  uint32_t hsink = 0;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_barrier(0);
      c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q], b[(q + 1) & 3], c[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x2& r = v[(q * 4 + n) & 7];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(m[0]), "v"(d[0]));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(d));
        if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r[0]) : "v"(hsink));
        if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hsink) : "v"(r[0]), "v"(r[1]));
        if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(m[0]), "v"(d[0]));
        if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(d));
        if (KIND == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(r[0]));
        if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0][0]) : "v"(m[0]), "v"(d[0]));
        if (KIND == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(m), "v"(d));
        if (KIND == 9) {
          if (n & 1) asm volatile("s_nop 0\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 1][0]) : "v"(m[0]), "v"(d[0]));
          else asm volatile("v_exp_f32 %0, %0" : "+v"(v[q & 1][0]));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = (float)hsink;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c[0][e] + c[1][e] + c[2][e] + c[3][e];
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e][0] + v[e][1];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// The exp / split of the one-plane in-batch kernel as it sits between the MFMAs of an O^T row (esr_inbatch2h.hip H1_O_ROW):
// one round = four MFMAs with the four stages of two pairs of probabilities behind them (22 VALU instructions).
// VARIANT: 0 as in the kernel; 1 the fp16 conversions as v_fma_mixlo / mixhi_f16 instead of v_cvt_pk_f16_f32; 2 v_exp_f32
// replaced by v_fma_f32; 3 without the two sums and the running maximum; 4 the same number of independent v_fma_f32;
// 5 the MFMAs alone; 6 = 0 with all 22 instructions behind the fourth MFMA.
template <int VARIANT>
__global__ __launch_bounds__(512) void mfma_mix_probe_kernel(int iters, unsigned long long* __restrict__ cycles,
                                                             float* __restrict__ sink) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  f16x8 a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t ha = probe_hash((blockIdx.x * 512 + threadIdx.x) * 64 + q * 16 + e);
      const uint32_t hb = probe_hash(ha + 0x9e3779b9u);
      a[q][e] = (_Float16)((float)(int)(ha & 0xffff) * (1.0f / 32768.f) - 1.0f);
      b[q][e] = (_Float16)((float)(int)(hb & 0xffff) * (1.0f / 32768.f) - 1.0f);
    }
  f32x16 c[4] = {};
  float p[4], la = 0.f, lb = 0.f, mx = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) p[e] = 0.25f * (float)((threadIdx.x + e) & 15) - 2.0f;
  const float sl2 = 0.7f, nref = -0.3f;
  uint32_t keep = 0;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 1.0f + 0.01f * (float)e;
  auto ex = [](float x) -> float {
    float r;
    if (VARIANT == 2) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(r) : "v"(x));
    else asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
  };
  auto fm = [&](float x) -> float {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(sl2), "v"(nref));
    return r;
  };
  auto cvt = [](float x, float y) -> uint32_t {
    uint32_t r;
    if (VARIANT == 1) {
      asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, 0" : "=v"(r) : "v"(x));
      asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, 0" : "+v"(r) : "v"(y));
    } else if (VARIANT == 7 || VARIANT == 9) {
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    } else {
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    }
    return r;
  };
  auto mixlo = [](float x, uint32_t pk) -> float {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
    return r;
  };
  auto mixhi = [](float x, uint32_t pk) -> float {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
    return r;
  };
  auto summax = [&](float e0, float e1) {
    if (VARIANT == 3) return;
    asm volatile("s_nop 0\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_max3_f32 %2, %2, %3, %4"
                 : "+v"(la), "+v"(lb), "+v"(mx) : "v"(e0), "v"(e1));
  };
#define MFMA_Q(Q) { __builtin_amdgcn_sched_barrier(0); c[Q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[Q], b[(Q + 1) & 3], c[Q], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
#define PLAIN(N) { _Pragma("unroll") for (int n_ = 0; n_ < (N); ++n_) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n_ & 7]) : "v"(sl2), "v"(nref)); }
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  // variants 8 / 9: every consumer reads what the round BEFORE produced (arguments -> exps -> planes across three rounds)
  float qa[4] = {0.f, 0.f, 0.f, 0.f}, qe[4] = {1.f, 1.f, 1.f, 1.f};
  for (int i = 0; i < iters; ++i) {
    if (VARIANT == 8 || VARIANT == 9) {
      float na[4], ne[4];
      MFMA_Q(0)
      na[0] = fm(p[0]); na[1] = fm(p[1]); na[2] = fm(p[2]); na[3] = fm(p[3]);
      ne[0] = ex(qa[0]); ne[1] = ex(qa[1]);
      MFMA_Q(1)
      ne[2] = ex(qa[2]); ne[3] = ex(qa[3]);
      summax(qe[0], qe[1]);
      const uint32_t paA = cvt(qe[0], qe[1]);
      MFMA_Q(2)
      summax(qe[2], qe[3]);
      const uint32_t paB = cvt(qe[2], qe[3]);
      const float rA0 = mixlo(qe[0], paA), rA1 = mixhi(qe[1], paA);
      MFMA_Q(3)
      const float rB0 = mixlo(qe[2], paB), rB1 = mixhi(qe[3], paB);
      const uint32_t pqA = cvt(rA0, rA1);
      const uint32_t pqB = cvt(rB0, rB1);
      keep ^= paA ^ paB ^ pqA ^ pqB;
#pragma unroll
      for (int e = 0; e < 4; ++e) { qa[e] = na[e]; qe[e] = ne[e]; }
    } else if (VARIANT == 5) {
      MFMA_Q(0) MFMA_Q(1) MFMA_Q(2) MFMA_Q(3)
    } else if (VARIANT == 4) {
      MFMA_Q(0) PLAIN(6) MFMA_Q(1) PLAIN(6) MFMA_Q(2) PLAIN(6) MFMA_Q(3) PLAIN(4)
    } else {
      float aA0, aA1, aB0, aB1, eA0, eA1, eB0, eB1, rA0, rA1;
      uint32_t paA, paB;
      MFMA_Q(0)
      if (VARIANT == 6) { MFMA_Q(1) MFMA_Q(2) MFMA_Q(3) }
      aA0 = fm(p[0]); aA1 = fm(p[1]); aB0 = fm(p[2]); aB1 = fm(p[3]);
      eA0 = ex(aA0); eA1 = ex(aA1);
      if (VARIANT != 6) MFMA_Q(1)
      eB0 = ex(aB0); eB1 = ex(aB1);
      summax(eA0, eA1);
      paA = cvt(eA0, eA1);
      if (VARIANT != 6) MFMA_Q(2)
      summax(eB0, eB1);
      paB = cvt(eB0, eB1);
      rA0 = mixlo(eA0, paA); rA1 = mixhi(eA1, paA);
      if (VARIANT != 6) MFMA_Q(3)
      const uint32_t pqA = cvt(rA0, rA1);
      const uint32_t pqB = cvt(mixlo(eB0, paB), mixhi(eB1, paB));
      keep ^= paA ^ paB ^ pqA ^ pqB;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
#undef MFMA_Q
#undef PLAIN
  float s = (float)keep + la + lb + mx;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c[0][e] + c[1][e] + c[2][e] + c[3][e];
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// Pure-read HBM bandwidth: every workgroup streams its own contiguous slice with `UNROLL` 16-byte loads in flight per
// lane (nt = streaming loads).  What a read-only kernel can reach on this box (pass C of the fp16 in-batch path reads the
// B x B probabilities at 3.8 TB/s).
typedef float pf4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void hbm_read_probe_kernel(const pf4* __restrict__ x, int64_t n16, float* sink) {
  const int64_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = min(n16, lo + per);
  pf4 a = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = lo + threadIdx.x; i < hi; i += (int64_t)256 * UNROLL) {
    pf4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t k = i + (int64_t)256 * u;
      const pf4 z = {0.f, 0.f, 0.f, 0.f};
      v[u] = k < hi ? (NT ? __builtin_nontemporal_load(x + k) : x[k]) : z;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) a += v[u];
  }
  if (a[0] + a[1] + a[2] + a[3] == 12345.678f) sink[0] = a[0];
}

}  // namespace esr

using namespace esr;

extern "C" {

int esr_probe_mfma(int dtype, int workgroups, int iters, float* sink, double* flops_out, esr_stream_t stream) {
  ESR_REQUIRE(workgroups > 0 && iters > 0 && sink && flops_out, "esr_probe_mfma: bad arguments");
  const bool live = (dtype & ESR_PROBE_LIVE_DATA) != 0;
  dtype &= ~ESR_PROBE_LIVE_DATA;
  ESR_REQUIRE(dtype == ESR_F32 || dtype == ESR_BF16 || dtype == ESR_PROBE_F16, "esr_probe_mfma: dtype %d", dtype);
  ESR_REQUIRE(!live || dtype != ESR_F32, "esr_probe_mfma: ESR_PROBE_LIVE_DATA is for ESR_BF16 / ESR_PROBE_F16");
  ESR_REQUIRE(dtype != ESR_PROBE_F16 || live, "esr_probe_mfma: ESR_PROBE_F16 has the live-data form only");
  const double per_mfma = 2.0 * 32 * 32 * (dtype == ESR_F32 ? 2 : 16);
  *flops_out = per_mfma * 4.0 * (double)iters * 4.0 * (double)workgroups;  // 4 chains x iters x 4 waves x grid
  if (dtype == ESR_PROBE_F16)
    hipLaunchKernelGGL(mfma_f16_live_probe_kernel, dim3(workgroups), dim3(256), 0, as_stream(stream), iters, sink);
  else if (live)
    hipLaunchKernelGGL(mfma_bf16_live_probe_kernel, dim3(workgroups), dim3(256), 0, as_stream(stream), iters, sink);
  else if (dtype == ESR_BF16)
    hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3(workgroups), dim3(256), 0, as_stream(stream), iters, sink);
  else
    hipLaunchKernelGGL(mfma_f32_probe_kernel, dim3(workgroups), dim3(256), 0, as_stream(stream), iters, sink);
  return check_launch("esr_probe_mfma");
}

int esr_probe_hbm_read(const void* x, int64_t bytes, int workgroups, int nontemporal, float* sink, esr_stream_t stream) {
  ESR_REQUIRE(x && bytes >= 16 && workgroups > 0 && sink && ((uintptr_t)x & 15) == 0, "esr_probe_hbm_read: bad arguments");
  const int64_t n16 = bytes / 16;
  if (nontemporal)
    hipLaunchKernelGGL((hbm_read_probe_kernel<8, true>), dim3(workgroups), dim3(256), 0, as_stream(stream),
                       (const pf4*)x, n16, sink);
  else
    hipLaunchKernelGGL((hbm_read_probe_kernel<8, false>), dim3(workgroups), dim3(256), 0, as_stream(stream),
                       (const pf4*)x, n16, sink);
  return check_launch("esr_probe_hbm_read");
}

int esr_probe_mfma_valu(int nv, int nt, int grouped, int waves_per_simd, int workgroups, int iters,
                        unsigned long long* cycles, float* sink, esr_stream_t stream) {
  ESR_REQUIRE(workgroups > 0 && iters > 0 && cycles && sink && (waves_per_simd == 1 || waves_per_simd == 2),
              "esr_probe_mfma_valu: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(workgroups), block(256 * waves_per_simd);
#define PROBE_CASE(NV, NT)                                                                                   \
  if (nv == NV && nt == NT) {                                                                                \
    if (grouped) mfma_valu_probe_kernel<NV, NT, true><<<grid, block, 0, st>>>(iters, cycles, sink);          \
    else mfma_valu_probe_kernel<NV, NT, false><<<grid, block, 0, st>>>(iters, cycles, sink);                 \
    return check_launch("esr_probe_mfma_valu");                                                              \
  }
  PROBE_CASE(0, 0) PROBE_CASE(1, 0) PROBE_CASE(2, 0) PROBE_CASE(4, 0) PROBE_CASE(6, 0) PROBE_CASE(7, 0)
  PROBE_CASE(8, 0) PROBE_CASE(12, 0) PROBE_CASE(0, 1) PROBE_CASE(0, 2) PROBE_CASE(4, 1) PROBE_CASE(3, 1)
#undef PROBE_CASE
#define PROBE_MIX(K)                                                                                         \
  if (nv == -2 && nt == K) {                                                                                 \
    mfma_mix_probe_kernel<K><<<grid, block, 0, st>>>(iters, cycles, sink);                                   \
    return check_launch("esr_probe_mfma_valu");                                                              \
  }
  PROBE_MIX(0) PROBE_MIX(1) PROBE_MIX(2) PROBE_MIX(3) PROBE_MIX(4) PROBE_MIX(5) PROBE_MIX(6) PROBE_MIX(7) PROBE_MIX(8)
  PROBE_MIX(9)
#undef PROBE_MIX
#define PROBE_KIND(K)                                                                                        \
  if (nv == -1 && nt == K) {                                                                                 \
    mfma_kind_probe_kernel<K><<<grid, block, 0, st>>>(iters, cycles, sink);                                  \
    return check_launch("esr_probe_mfma_valu");                                                              \
  }
  PROBE_KIND(0) PROBE_KIND(1) PROBE_KIND(2) PROBE_KIND(3) PROBE_KIND(4) PROBE_KIND(5) PROBE_KIND(6) PROBE_KIND(7)
  PROBE_KIND(8) PROBE_KIND(9)
#undef PROBE_KIND
  ESR_REQUIRE(false, "esr_probe_mfma_valu: no instance for nv = %d, nt = %d", nv, nt);
  return 0;
}

}  // extern "C"
