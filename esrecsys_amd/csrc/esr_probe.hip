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

}  // extern "C"
