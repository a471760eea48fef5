/* libesr_probe.so -- measurement probes of the MI355X build of the ESRecsys hot path.
 *
 * NOT part of the product ABI (include/esr_hip.h): two micro-benchmarks that bench.py and benchmarks/ use to put the
 * data-sheet peaks next to what THIS box sustains -- the matrix pipes under a register-only MFMA loop, the HBM under a
 * pure read stream.  Built from esrecsys_amd/csrc/esr_probe.hip into its own shared object (it resolves the error
 * helpers of libesr_hip.so, which must be loaded first).  Same conventions as esr_hip.h: 0 / negative ESR_E* codes,
 * asynchronous on `stream`. */
#ifndef ESR_PROBE_H
#define ESR_PROBE_H
#include "esr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement probe (not on the hot path): `workgroups` x 4 waves each run `iters` rounds of four independent
 * register-only MFMA chains (dtype ESR_BF16: v_mfma_f32_32x32x16_bf16, ESR_F32: v_mfma_f32_32x32x2_f32).
 * *flops_out (host) = flops the launch executes; time it on `stream` to get the matrix ceiling this box sustains.
 * sink: one device float (never written).  dtype | ESR_PROBE_LIVE_DATA feeds the chains full-entropy operands that
 * change every instruction instead of constants: switching activity, and with it the clock the part holds under
 * its power limit, is that of a real GEMM (constants measured 2.44 PFLOP/s bf16; see profiles/). */
#define ESR_PROBE_LIVE_DATA 0x100
#define ESR_PROBE_F16 2 /* v_mfma_f32_32x32x16_f16 (with ESR_PROBE_LIVE_DATA): the planes of the f16x2 paths */
int esr_probe_mfma(int dtype, int workgroups, int iters, float* sink, double* flops_out, esr_stream_t stream);
/* Measurement probe: `bytes` of x read once by `workgroups` workgroups, each streaming its own contiguous slice with
 * eight 16-byte loads in flight per lane (nontemporal != 0: streaming loads).  Time it with events: the pure-read
 * bandwidth a kernel can reach on this box.  sink: one device float (never written). */
int esr_probe_hbm_read(const void* x, int64_t bytes, int workgroups, int nontemporal, float* sink, esr_stream_t stream);

/* Measurement probe: does the vector ALU run beside the matrix pipe?  `workgroups` workgroups of 4 * waves_per_simd waves
 * (waves_per_simd 1 or 2) run `iters` rounds of four independent v_mfma_f32_32x32x16_f16, each followed by `nv` plain
 * VALU instructions and `nt` v_exp_f32 (grouped != 0: the four MFMAs first, then all the VALU work).  cycles (device,
 * one word per wave): shader-clock cycles of the loop.  Instances: (nv, nt) in {0,1,2,4,6,7,8,12} x {0}, (0,1), (0,2),
 * (4,1), (3,1).  nv = -1: four instructions of ONE kind behind every MFMA, nt = the kind (0 v_fma_f32, 1 v_pk_fma_f32,
 * 2 v_fma_mix_f32, 3 v_cvt_pk_f16_f32, 4 v_max3_f32, 5 v_pk_add_f32, 6 v_exp_f32, 7 / 8 a dependent chain of v_fma_f32 /
 * v_pk_fma_f32, 9 v_exp_f32 -> v_fma_f32 chains).  nv = -2: the exp / split of the one-plane in-batch kernel between the
 * four MFMAs of a round, nt = variant (0 as in the kernel, 1 conversions by v_fma_mixlo / mixhi_f16, 2 no v_exp_f32, 3 no
 * sums / maximum, 4 as many independent v_fma_f32, 5 MFMAs alone, 6 all of it behind the fourth MFMA). */
int esr_probe_mfma_valu(int nv, int nt, int grouped, int waves_per_simd, int workgroups, int iters,
                        unsigned long long* cycles, float* sink, esr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
