// Rate probe (gfx950): a chain of v_mfma_f32_32x32x16_bf16 with R LDS reads of one kind issued after each MFMA,
// one wave per SIMD (256 threads), all CUs.  Prints cycles per MFMA for each (kind, R): does the read hide?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int KIND, int R, int NACC>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // conflict-free patterns: b128 lane-linear 16 B; b64 / tr lane-linear 8 B
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const uint32_t a128 = base + lane * 16, a64 = base + lane * 8;
  f32x16 accs[NACC] = {};
  bf16x8 a = {}, b = {};
  uint32_t sink = 0;
  u32x4 v4[16 * (R ? R : 1)];
  u32x2 v2[16 * (R ? R : 1)];
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accs[m % NACC]) : "v"(a), "v"(b));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int off = ((m * R + r) & 7) * 1024;
        if (KIND == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(v4[m * R + r]) : "v"(a128 + off));
        if (KIND == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(v2[m * R + r]) : "v"(a64 + off));
        if (KIND == 3) asm volatile("v_xor_b32 %0, %1, %0\n v_xor_b32 %0, %1, %0" : "+v"(sink) : "v"(off + lane));
        if (KIND == 2) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v2[m * R + r]) : "v"(a64 + off));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16 * R; ++q) sink ^= (KIND == 0) ? v4[q][0] ^ v4[q][3] : v2[q][0] ^ v2[q][1];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = accs[0][0] + accs[NACC - 1][1] + (float)sink;
}
template <int KIND, int R, int NACC>
void run(float* out, long long* cyc) {
  const int iters = 200;
  hipLaunchKernelGGL((k<KIND, R, NACC>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<KIND, R, NACC>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("nacc %d kind %d (0=b128 1=b64 2=tr_b64 3=2xVALU) reads/mfma %d: %.1f memtime ticks per MFMA\n", NACC, KIND, R, (double)c / (iters * 16));
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0, 0, 1>(out, cyc); run<0, 0, 4>(out, cyc);
  run<0, 1, 1>(out, cyc); run<0, 2, 1>(out, cyc); run<0, 1, 4>(out, cyc); run<0, 2, 4>(out, cyc);
  run<2, 1, 1>(out, cyc); run<2, 2, 1>(out, cyc); run<2, 1, 4>(out, cyc); run<2, 2, 4>(out, cyc); run<2, 3, 4>(out, cyc);
  run<3, 1, 1>(out, cyc); run<3, 2, 1>(out, cyc); run<3, 4, 1>(out, cyc); run<3, 1, 4>(out, cyc); run<3, 4, 4>(out, cyc);
  return 0;
}
