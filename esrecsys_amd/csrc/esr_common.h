// Shared device/host helpers for libesr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <algorithm>

#include "../../include/esr_hip.h"

namespace esr {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kBlock = 256;        // 4 waves, one per SIMD
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 resident blocks: grid-stride beyond this

void set_error(const char* fmt, ...);
int check_launch(const char* what);
// out[0] = (float)(scale * sum(part[0..n))), one block, fixed reduction order (defined in esr_core.hip).
void finalize_scalar(const double* part, int n, double scale, float* out, hipStream_t st);

// top-k of dense score rows by radix select (esr_retrieve.hip); ESR_EINVAL when k exceeds what the select kernel holds
constexpr int kSelectMaxK = 1024;
int select_topk_dense(const float* scores, int64_t pitch, int64_t rows, int n, int k, float* out_scores,
                      int32_t* out_indices, hipStream_t st);

// the same select over ragged rows (row r: n_per_row[r] entries, at most max_n)
int select_topk_ragged(const float* scores, int64_t pitch, int64_t rows, const int32_t* n_per_row, int max_n, int k,
                       float* out_scores, int32_t* out_indices, hipStream_t st);
// the two stages of a filtered search (esr_ivf.hip; esr_retrieve_topk drives the same kernel itself).
// head: dense rows [rows][pitch] of n > k scores (index = column) -> the k best as (score bits, index) records at the
// head of pairs[row * ppitch ..], cnt[row] = k, tau[row] = the k-th best score.
// tail: the lists pairs[row * ppitch .. + cnt[row]) -- that head plus what a filtered pass appended -- -> the k best,
// best first (scores -inf / index -1 where a list holds fewer than k).
int select_topk_head(const float* scores, int64_t pitch, int64_t rows, int n, int k, int2* pairs, int64_t ppitch,
                     int32_t* cnt, float* tau, hipStream_t st);
int select_topk_tail(const int2* pairs, int64_t ppitch, const int32_t* cnt, int64_t rows, int k, float* out_scores,
                     int32_t* out_indices, hipStream_t st);
// between two filtered passes: every list cut back to its k best (in place), cnt = min(cnt, k), tau raised
int select_topk_compact(int2* pairs, int64_t ppitch, int32_t* cnt, int64_t rows, int k, float* tau, hipStream_t st,
                        int skip_upto = 0);

inline hipStream_t as_stream(esr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Per-kernel launch timing (esr_kernel_timing / esr_kernel_timing_read, esr_core.hip): OFF by default and then one
// relaxed load per launch site.  When on, a HIP event pair is recorded around the launch on the launch's own stream.
extern int g_ktimer_on;
void ktimer_begin(const char* name, hipStream_t st);
void ktimer_end(hipStream_t st);
// rocprofv3 markers (tracing row): OFF by default (one relaxed load per site); ESR_ROCTX=1 in the
// environment or esr_trace_markers(1) turns every ESR_KT launch site and every step entry point into a roctx range
// (roctxRangePushA / roctxRangePop of the rocprofiler-sdk roctx library, bound with dlopen on first use), so a
// `rocprofv3 --marker-trace --kernel-trace` timeline shows which step phase each kernel belongs to.
extern int g_trace_on;
void trace_push(const char* name);
void trace_pop();
struct TraceScope {
  bool on;
  explicit TraceScope(const char* name) : on(g_trace_on != 0) {
    if (on) trace_push(name);
  }
  ~TraceScope() {
    if (on) trace_pop();
  }
};
// A kernel whose VALU work runs beside MFMAs must not use the packed-f32 instructions (v_pk_fma_f32, v_pk_add_f32,
// v_pk_mul_f32): they do not run beside the matrix pipe (scripts/mfma_valu_probe.py; esr_inbatch2h.hip pk_fma).  The
// target attribute exists in the device compilation only.
#if defined(__HIP_DEVICE_COMPILE__)
#define ESR_NO_PK __attribute__((target("no-packed-fp32-ops")))
#else
#define ESR_NO_PK
#endif
#define ESR_KT(NAME, ST, ...)                                  \
  do {                                                         \
    if (esr::g_trace_on) esr::trace_push((NAME));              \
    if (esr::g_ktimer_on) esr::ktimer_begin((NAME), (ST));     \
    __VA_ARGS__;                                               \
    if (esr::g_ktimer_on) esr::ktimer_end((ST));               \
    if (esr::g_trace_on) esr::trace_pop();                     \
  } while (0)

#define ESR_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      esr::set_error(__VA_ARGS__);    \
      return ESR_EINVAL;              \
    }                                 \
  } while (0)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Row-group geometry: a row of `nvec` vector chunks is handled by G lanes (power of two,
// <= 64); a 256-thread block holds 256/G groups.
struct RowGeom {
  int vec;    // elements per chunk (4 = float4, 1 = scalar)
  int nvec;   // chunks per row
  int G;      // lanes per row group
  int nch;    // chunks per lane = ceil(nvec / G); register-resident row kernels support <= 4
};
inline RowGeom row_geom(int D) {
  RowGeom g;
  g.vec = (D % 4 == 0) ? 4 : 1;
  g.nvec = D / g.vec;
  int G = 1;
  while (G < g.nvec && G < kWave) G <<= 1;
  g.G = G;
  g.nch = (g.nvec + G - 1) / G;
  return g;
}
// bf16 rows (round 6): EIGHT elements per chunk -- 16 bytes of the bf16 row per lane and load instruction, as a float4
// chunk is of an f32 row (with four elements per chunk a bf16 row moved as 8-byte accesses: the one-pass steps ran
// SLOWER on bf16 tables than on f32 ones, 0.438 against 0.416 ms per 262 144 triplets).  The row's f32 companions
// (accumulator, parked gradient rows) are then 32 contiguous bytes per lane, two 16-byte accesses.
inline RowGeom row_geom8(int D) {
  RowGeom g;
  g.vec = 8;
  g.nvec = D / 8;
  int G = 1;
  while (G < g.nvec && G < kWave) G <<= 1;
  g.G = G;
  g.nch = (g.nvec + G - 1) / G;
  return g;
}
constexpr int kMaxChunksPerLane = 4;  // D <= 1024 (float4 rows) or D <= 256 (scalar rows)
// Expands BODY with constexpr VEC / NCH matching a RowGeom (nch 3 runs as 4 with bounds checks).
#define ESR_DISPATCH_ROW(geom, ...)                                           \
  do {                                                                        \
    if ((geom).vec == 4) {                                                    \
      if ((geom).nch <= 1) { constexpr int VEC = 4, NCH = 1; __VA_ARGS__; }   \
      else if ((geom).nch <= 2) { constexpr int VEC = 4, NCH = 2; __VA_ARGS__; } \
      else { constexpr int VEC = 4, NCH = 4; __VA_ARGS__; }                   \
    } else {                                                                  \
      if ((geom).nch <= 1) { constexpr int VEC = 1, NCH = 1; __VA_ARGS__; }   \
      else if ((geom).nch <= 2) { constexpr int VEC = 1, NCH = 2; __VA_ARGS__; } \
      else { constexpr int VEC = 1, NCH = 4; __VA_ARGS__; }                   \
    }                                                                         \
  } while (0)
// the same for a geometry of 8-element chunks (row_geom8)
#define ESR_DISPATCH_ROW8(geom, ...)                                          \
  do {                                                                        \
    if ((geom).nch <= 1) { constexpr int VEC = 8, NCH = 1; __VA_ARGS__; }     \
    else if ((geom).nch <= 2) { constexpr int VEC = 8, NCH = 2; __VA_ARGS__; } \
    else { constexpr int VEC = 8, NCH = 4; __VA_ARGS__; }                     \
  } while (0)
// either, by the geometry's chunk width
#define ESR_DISPATCH_ROW_ANY(geom, ...)                                       \
  do {                                                                        \
    if ((geom).vec == 8) { ESR_DISPATCH_ROW8(geom, __VA_ARGS__); }            \
    else { ESR_DISPATCH_ROW(geom, __VA_ARGS__); }                             \
  } while (0)
inline int grid_for_groups(int64_t ngroups_needed, int G) {
  int groups_per_block = kBlock / G;
  int64_t blocks = cdiv(ngroups_needed, groups_per_block);
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  return (int)blocks;
}

#ifdef __HIPCC__
// value of lane (lane ^ stride), stride a compile-time power of two < 64 after inlining: DPP where the pattern exists
// on gfx9 (quad_perm for 1 and 2, row_ror:8 for 8), the LDS crossbar without an address for 4 and 16 (ds_swizzle,
// bit mode), v_permlane32_swap for 32.  A __shfl_xor is a ds_bpermute with a computed address per step.
__device__ __forceinline__ uint32_t xor_lane(uint32_t v, int stride) {
  switch (stride) {
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
    case 8: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
    case 4: return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (4 << 10));
    case 16: return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (16 << 10));
    case 32: {  // gfx950: one v_permlane32_swap (upper half of a copy <-> lower half of another) + a select
      const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // r[0] = [lo, lo], r[1] = [hi, hi]
      return (threadIdx.x & 32) ? r[0] : r[1];
    }
    default: return (uint32_t)__shfl_xor((int)v, stride, 64);
  }
}
// NOTE on "explicit" roundings: hipcc's __fmul_rn / __fadd_rn / __fsub_rn are plain `*` / `+` / `-` (the rounded OCML
// forms sit behind OCML_BASIC_ROUNDED_OPERATIONS) and the default -ffp-contract=fast-honor-pragmas may fuse such a
// product into a following sum; only __fmaf_rn (= __builtin_fmaf) is what it says.  Two kernels that must agree bit for bit
// therefore have to present the SAME expression to the compiler -- or cut it where one of them goes through memory
// (asm volatile("" : "+v"(x)) on the finished value: esr_optim.hip inbatch_merge_update_kernel).
// Sum across the G (power of two) lanes of a row group, butterfly from stride 1 up; every lane gets the total.
__device__ __forceinline__ float group_sum(float v, int G) {
  if (G > 1) v += __uint_as_float(xor_lane(__float_as_uint(v), 1));
  if (G > 2) v += __uint_as_float(xor_lane(__float_as_uint(v), 2));
  if (G > 4) v += __uint_as_float(xor_lane(__float_as_uint(v), 4));
  if (G > 8) v += __uint_as_float(xor_lane(__float_as_uint(v), 8));
  if (G > 16) v += __uint_as_float(xor_lane(__float_as_uint(v), 16));
  if (G > 32) v += __uint_as_float(xor_lane(__float_as_uint(v), 32));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
// Deterministic block-wide double sum (fixed tree); result valid in thread 0.
// `smem` must hold >= 4 doubles per quantity (256-thread block = 4 waves).
__device__ __forceinline__ double block_sum_d(double v, double* smem) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) t += smem[i];
  }
  return t;
}

// Lazy momentum (esr_optim.hip momentum_catchup_kernel / momentum_flush_kernel, esr_spotify.hip's fused step): n missed
// steps of  trace *= m ; p -= lr * trace.  n <= kLazyExact: one by one, the dense pass's own operations (bit-identical to
// it); longer gaps: the closed form (one rounding instead of n).
constexpr int kLazyExact = 64;
struct DecayCoef {  // of a gap of n steps: a function of n and m alone -- formed ONCE per row (powf), applied per element
  int n;
  float mn, geo;
};
__device__ __forceinline__ DecayCoef decay_coef(int n, float m) {
  DecayCoef k{n, 0.f, 0.f};
  if (n > kLazyExact) {
    k.mn = powf(m, (float)n);
    k.geo = m * (1.0f - k.mn) / (1.0f - m);
  }
  return k;
}
__device__ __forceinline__ void decay_apply(float& p, float& t, const DecayCoef& k, float lr, float m) {
  if (k.n <= kLazyExact) {
    for (int i = 0; i < k.n; ++i) {
      t = __fmul_rn(t, m);
      p = __fsub_rn(p, __fmul_rn(lr, t));
    }
  } else {
    p = __fsub_rn(p, __fmul_rn(__fmul_rn(lr, t), k.geo));
    t = __fmul_rn(t, k.mn);
  }
}
__device__ __forceinline__ void decay_steps(float& p, float& t, int n, float lr, float m) {
  decay_apply(p, t, decay_coef(n, m), lr, m);
}

// A row held in registers by its G-lane group: chunk k of this lane is row chunk lig + k*G.
// All indices are compile-time so the arrays stay in VGPRs (no scratch).
// -DESR_ROW_STORE_NT=1: row stores (table rows, accumulators, gradient rows) as streaming (non-temporal) stores.  The idea:
// a step's dirty rows sit in the XCD's write-back L2 until the kernel ends and are written back at the release in front of
// the next step's launch.  Measured (round 5, scripts/gpu_nt_ab.sh, alternating runs): GloVe B = 2048 +2.5 %, triplet
// B = 8192 -2 %, GloVe B = 65 536 -1.5 %, triplet B = 262 144 -1 %, in-batch equal -- off.
#ifndef ESR_ROW_STORE_NT
#define ESR_ROW_STORE_NT 0
#endif
template <int VEC, int NCH>
struct RowRegs {
  float v[NCH][VEC];
};
template <int VEC, int NCH>
__device__ __forceinline__ void row_load(RowRegs<VEC, NCH>& r, const float* __restrict__ p, int lig,
                                         int G, int nvec) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lig + k * G;
    if (c < nvec) {
      if constexpr (VEC == 8) {
        const float4 a = *reinterpret_cast<const float4*>(p + 8 * c);
        const float4 b = *reinterpret_cast<const float4*>(p + 8 * c + 4);
        r.v[k][0] = a.x; r.v[k][1] = a.y; r.v[k][2] = a.z; r.v[k][3] = a.w;
        r.v[k][4] = b.x; r.v[k][5] = b.y; r.v[k][6] = b.z; r.v[k][7] = b.w;
      } else if constexpr (VEC == 4) {
        const float4 a = *reinterpret_cast<const float4*>(p + 4 * c);
        r.v[k][0] = a.x; r.v[k][1] = a.y; r.v[k][2] = a.z; r.v[k][3] = a.w;
      } else {
        r.v[k][0] = p[c];
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) r.v[k][e] = 0.f;
    }
  }
}
template <int VEC, int NCH>
__device__ __forceinline__ void row_store(const RowRegs<VEC, NCH>& r, float* __restrict__ p, int lig,
                                          int G, int nvec) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lig + k * G;
    if (c < nvec) {
      if constexpr (VEC == 8) {
        *reinterpret_cast<float4*>(p + 8 * c) = make_float4(r.v[k][0], r.v[k][1], r.v[k][2], r.v[k][3]);
        *reinterpret_cast<float4*>(p + 8 * c + 4) = make_float4(r.v[k][4], r.v[k][5], r.v[k][6], r.v[k][7]);
      } else if constexpr (VEC == 4) {
#if ESR_ROW_STORE_NT
        typedef float esr_f32x4_ __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(esr_f32x4_{r.v[k][0], r.v[k][1], r.v[k][2], r.v[k][3]},
                                    reinterpret_cast<esr_f32x4_*>(p + 4 * c));
#else
        *reinterpret_cast<float4*>(p + 4 * c) = make_float4(r.v[k][0], r.v[k][1], r.v[k][2], r.v[k][3]);
#endif
      } else {
        p[c] = r.v[k][0];
      }
    }
  }
}
// bf16 table rows (config 4's dtype): the register image stays f32 -- a load widens (exact), a store rounds to nearest
// even.  A float4 chunk of a row is 8 bytes here.
__device__ __forceinline__ uint16_t f32_to_bf16(float f);
// {bf16(lo), bf16(hi)} in one dword: v_cvt_pk_bf16_f32 (round to nearest even, NaN kept quiet) -- one instruction per
// pair where the bit arithmetic of f32_to_bf16 is six per element (and four registers more in triplet_direct_kernel:
// 100 instead of 96, i.e. four waves per SIMD instead of five in a kernel that lives on its occupancy)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef float esr_f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 esr_bf16x2_ __attribute__((ext_vector_type(2)));
  const esr_f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, esr_bf16x2_));
}
template <int VEC, int NCH>
__device__ __forceinline__ void row_load(RowRegs<VEC, NCH>& r, const uint16_t* __restrict__ p, int lig,
                                         int G, int nvec) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lig + k * G;
    if (c < nvec) {
      if constexpr (VEC == 8) {
        const uint4 a = *reinterpret_cast<const uint4*>(p + 8 * c);
        r.v[k][0] = __uint_as_float(a.x << 16); r.v[k][1] = __uint_as_float(a.x & 0xFFFF0000u);
        r.v[k][2] = __uint_as_float(a.y << 16); r.v[k][3] = __uint_as_float(a.y & 0xFFFF0000u);
        r.v[k][4] = __uint_as_float(a.z << 16); r.v[k][5] = __uint_as_float(a.z & 0xFFFF0000u);
        r.v[k][6] = __uint_as_float(a.w << 16); r.v[k][7] = __uint_as_float(a.w & 0xFFFF0000u);
      } else if constexpr (VEC == 4) {
        const uint2 a = *reinterpret_cast<const uint2*>(p + 4 * c);
        r.v[k][0] = __uint_as_float(a.x << 16); r.v[k][1] = __uint_as_float(a.x & 0xFFFF0000u);
        r.v[k][2] = __uint_as_float(a.y << 16); r.v[k][3] = __uint_as_float(a.y & 0xFFFF0000u);
      } else {
        r.v[k][0] = __uint_as_float(((uint32_t)p[c]) << 16);
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) r.v[k][e] = 0.f;
    }
  }
}
template <int VEC, int NCH>
__device__ __forceinline__ void row_store(const RowRegs<VEC, NCH>& r, uint16_t* __restrict__ p, int lig,
                                          int G, int nvec) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lig + k * G;
    if (c < nvec) {
      if constexpr (VEC == 8) {
        uint4 a;
        a.x = pack_bf16x2(r.v[k][0], r.v[k][1]);
        a.y = pack_bf16x2(r.v[k][2], r.v[k][3]);
        a.z = pack_bf16x2(r.v[k][4], r.v[k][5]);
        a.w = pack_bf16x2(r.v[k][6], r.v[k][7]);
        *reinterpret_cast<uint4*>(p + 8 * c) = a;
      } else if constexpr (VEC == 4) {
        uint2 a;
        a.x = pack_bf16x2(r.v[k][0], r.v[k][1]);
        a.y = pack_bf16x2(r.v[k][2], r.v[k][3]);
        *reinterpret_cast<uint2*>(p + 4 * c) = a;
      } else {
        p[c] = f32_to_bf16(r.v[k][0]);
      }
    }
  }
}
template <int VEC, int NCH>
__device__ __forceinline__ float row_dot_partial(const RowRegs<VEC, NCH>& a, const RowRegs<VEC, NCH>& b) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc = fmaf(a.v[k][e], b.v[k][e], acc);
  return acc;
}

// optax.adagrad [upstream]: acc += g^2 ; p -= lr * g * rsqrt(acc + eps)  (0 where acc == 0).  ONE definition for every
// kernel that applies it (esr_optim.hip, esr_glove_step.hip), so their results are bit-identical.
__device__ __forceinline__ void adagrad_elem(float& w, float& a, float gv, float lr, float eps) {
  // explicit roundings: no fused multiply-add may be formed here, whatever kernel this is inlined into
  const float acc = __fmaf_rn(gv, gv, a);
  a = acc;
  const float inv = acc > 0.f ? __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(acc, eps))) : 0.f;
  w = __fsub_rn(w, __fmul_rn(__fmul_rn(lr, gv), inv));
}

// One element of a Shop-The-Look gradient row: (a x + c own) / B with a = +-[margin > 0], x = the
// partner term (n - p for the scene row, s for the product rows), c = lam [|own| > 1] / |own|.  ONE definition with
// explicit roundings for esr_triplet.hip and esr_triplet_step.hip: the two paths produce the same bits.
__device__ __forceinline__ float trip_grad(float a, float x, float c, float own, float inv_bs) {
  return __fmul_rn(__fadd_rn(__fmul_rn(a, x), __fmul_rn(c, own)), inv_bs);
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN preserved
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
#endif

}  // namespace esr
