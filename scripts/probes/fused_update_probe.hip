// PROBE (not product): what would ONE launch cost that does the work of merge<Q>'s O part + merge<C> + the sparse-Adagrad
// update of the in-batch step (C2: B = 8192 pairs, two 1 M x 128 fp32 towers)?  Per sorted occurrence: 8 partial-O rows
// (512 B each) with their weights, the partner row from the fp16 planes, the own table row + accumulator (random rows of
// the tower), the gradient formed in registers, Adagrad, two row stores.  Today: merge<Q> 13.8 + merge<C> 11.7 +
// segment_update 11.4 us (rocprof, profiles/r4/inbatch_kernel_stats.csv); the fused form still needs a ~4 us statistics
// launch in front of pass C.  Build: hipcc --offload-arch=gfx950 -O3 fused_update_probe.hip -o fused_update_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kB = 8192, kD = 128, kSplit = 8, kBlock = 256;
constexpr int64_t kV = 1000000;

__device__ __forceinline__ float group_sum32(float v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__global__ __launch_bounds__(kBlock) void fused_update(const int32_t* __restrict__ sorted_ids, const int32_t* __restrict__ perm,
                                                       const float* __restrict__ partO,   // [2][8][B][128]
                                                       const float* __restrict__ wts,     // [2][8][B]
                                                       const float* __restrict__ invl,    // [2][B]
                                                       const _Float16* __restrict__ planes,  // [2 sides][2][B][128]
                                                       float* __restrict__ tab0, float* __restrict__ tab1,
                                                       float* __restrict__ acc0, float* __restrict__ acc1, float scale,
                                                       float lam, float inv_bs, float lr, float eps) {
  const int lig = threadIdx.x & 31;
  const int64_t p = (int64_t)blockIdx.x * (kBlock / 32) + threadIdx.x / 32;
  if (p >= 2 * kB) return;
  const int32_t o = perm[p];
  const int32_t id = sorted_ids[p];
  const int side = o >= kB, b = o - side * kB;
  float4 po[kSplit];
  float wt[kSplit];
#pragma unroll
  for (int s = 0; s < kSplit; ++s) {
    po[s] = *reinterpret_cast<const float4*>(partO + (((int64_t)side * kSplit + s) * kB + b) * kD + 4 * lig);
    wt[s] = wts[((int64_t)side * kSplit + s) * kB + b];
  }
  const float il = invl[side * kB + b];
  const _Float16* ph = planes + (((int64_t)(1 - side) * 2 + 0) * kB + b) * kD + 4 * lig;
  const _Float16* pl = planes + (((int64_t)(1 - side) * 2 + 1) * kB + b) * kD + 4 * lig;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 yh = *reinterpret_cast<const h4*>(ph), yl = *reinterpret_cast<const h4*>(pl);
  float* trow = (side ? tab1 : tab0) + (int64_t)id * kD + 4 * lig;
  float* arow = (side ? acc1 : acc0) + (int64_t)id * kD + 4 * lig;
  float4 w = *reinterpret_cast<const float4*>(trow);
  float4 a = *reinterpret_cast<const float4*>(arow);
  float4 osum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < kSplit; ++s) {
    osum.x += po[s].x * wt[s]; osum.y += po[s].y * wt[s]; osum.z += po[s].z * wt[s]; osum.w += po[s].w * wt[s];
  }
  const float xn2 = group_sum32(w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w);
  const float xn = sqrtf(xn2);
  const float creg = xn > 1.f ? lam / xn : 0.f;
  const float y0 = (float)yh[0] + (float)yl[0], y1 = (float)yh[1] + (float)yl[1], y2 = (float)yh[2] + (float)yl[2],
              y3 = (float)yh[3] + (float)yl[3];
  float4 g;
  g.x = (scale * (osum.x * il - y0) + creg * w.x) * inv_bs;
  g.y = (scale * (osum.y * il - y1) + creg * w.y) * inv_bs;
  g.z = (scale * (osum.z * il - y2) + creg * w.z) * inv_bs;
  g.w = (scale * (osum.w * il - y3) + creg * w.w) * inv_bs;
  a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
  w.x -= lr * g.x / (sqrtf(a.x) + eps); w.y -= lr * g.y / (sqrtf(a.y) + eps);
  w.z -= lr * g.z / (sqrtf(a.z) + eps); w.w -= lr * g.w / (sqrtf(a.w) + eps);
  *reinterpret_cast<float4*>(trow) = w;
  *reinterpret_cast<float4*>(arow) = a;
}

__global__ void fill(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v + 1e-3f * (float)(i & 1023);
}

int main() {
  std::mt19937_64 rng(1701);
  std::vector<int32_t> ids(2 * kB), perm(2 * kB);
  for (auto& v : ids) v = (int32_t)(rng() % kV);
  std::iota(perm.begin(), perm.end(), 0);
  std::vector<int32_t> order(2 * kB);
  std::iota(order.begin(), order.end(), 0);
  // sorted by (side, id) like the virtual-id sort: scene ids first, then product ids
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    const int sa = a >= kB, sb = b >= kB;
    return sa != sb ? sa < sb : ids[a] < ids[b];
  });
  std::vector<int32_t> sorted(2 * kB);
  for (int i = 0; i < 2 * kB; ++i) { sorted[i] = ids[order[i]]; perm[i] = order[i]; }
  int32_t *d_sorted, *d_perm;
  float *partO, *wts, *invl, *tab0, *tab1, *acc0, *acc1;
  _Float16* planes;
  CK(hipMalloc(&d_sorted, 2 * kB * 4)); CK(hipMalloc(&d_perm, 2 * kB * 4));
  CK(hipMemcpy(d_sorted, sorted.data(), 2 * kB * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_perm, perm.data(), 2 * kB * 4, hipMemcpyHostToDevice));
  const size_t nO = (size_t)2 * kSplit * kB * kD, nT = (size_t)kV * kD;
  CK(hipMalloc(&partO, nO * 4)); CK(hipMalloc(&wts, (size_t)2 * kSplit * kB * 4)); CK(hipMalloc(&invl, 2 * kB * 4));
  CK(hipMalloc(&planes, (size_t)4 * kB * kD * 2));
  CK(hipMalloc(&tab0, nT * 4)); CK(hipMalloc(&tab1, nT * 4)); CK(hipMalloc(&acc0, nT * 4)); CK(hipMalloc(&acc1, nT * 4));
  fill<<<4096, 256>>>(partO, nO, 0.01f); fill<<<256, 256>>>(wts, (size_t)2 * kSplit * kB, 0.1f); fill<<<64, 256>>>(invl, 2 * kB, 1.f);
  CK(hipMemset(planes, 0, (size_t)4 * kB * kD * 2));
  fill<<<8192, 256>>>(tab0, nT, 0.05f); fill<<<8192, 256>>>(tab1, nT, 0.05f); fill<<<8192, 256>>>(acc0, nT, 0.1f); fill<<<8192, 256>>>(acc1, nT, 0.1f);
  CK(hipDeviceSynchronize());
  // something that evicts the partials from the caches between launches, as pass C does in the real step (268 MB stream)
  float* junk; const size_t nJ = (size_t)96 << 20; CK(hipMalloc(&junk, nJ * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (2 * kB * 32 + kBlock - 1) / kBlock;
  for (int mode = 0; mode < 2; ++mode) {
    float tot = 0.f; const int reps = 40;
    for (int r = 0; r < reps + 5; ++r) {
      if (mode == 1) fill<<<8192, 256>>>(junk, nJ, 1.f);
      CK(hipEventRecord(e0));
      fused_update<<<grid, kBlock>>>(d_sorted, d_perm, partO, wts, invl, planes, tab0, tab1, acc0, acc1, 8.f, 0.1f, 1.f / kB, 0.05f, 1e-7f);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 5) tot += ms;
    }
    printf("fused_update probe (%s): %.2f us per launch (events around the launch; ~2 us of event overhead)\n",
           mode ? "caches flushed by a 384 MB stream in front" : "back to back", tot / reps * 1e3);
  }
  return 0;
}
