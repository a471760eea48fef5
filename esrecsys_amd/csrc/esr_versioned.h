// Shared pieces of the one-pass train steps on DOUBLE-BUFFERED tables (esr_glove.hip, esr_triplet_step.hip).
//
// A table that such a step updates lives in two [V, D] buffers plus one byte per row (`loc`) that says which buffer
// holds the row's current value.  A step reads rows where `loc` pointed when it began, writes every updated row into
// the OTHER buffer and flips the byte: readers and writers of one launch never touch the same bytes, so gradients can
// be formed on chip from rows that other workgroups are rewriting, and no gradient row or snapshot ever goes to memory.
//
// Round 3: the byte is STAMPED.  bit 0 = the buffer that holds the row's current value, bits 1..7 = the stamp of the
// step that last wrote the row (1..127; 0 = "long ago").  A step carries its own stamp T; a reader that finds stamp T
// on a byte knows the row was rewritten DURING this step (a row is rewritten at most once per step) and that the value
// the step began with is in the other buffer.  Either way it reads the buffer nobody is writing, whichever of the two
// byte values it happens to see -- so the update kernel resolves row locations itself and the per-step plan kernel of
// round 2 (which had to snapshot the bytes between two steps) is gone: everything else a plan held depends on the ids
// only and is made ahead, with the sort.  Stamps are reused after 127 steps; the caller clears them (esr_rows_restamp,
// one pass over V bytes) each time its counter wraps.
#pragma once
#include "esr_common.h"

namespace esr {

constexpr int kStepChunk = 32;  // == kSegChunk of esr_optim.hip: same cut points, same association of every sum
constexpr uint32_t kLocBit = 0x80000000u;   // row code: the row's current value is in the second buffer
constexpr uint32_t kSideBit = 0x40000000u;  // (GloVe plan records) the occurrence is the pair's second token
constexpr uint32_t kIdMask = 0x3FFFFFFFu;   // row code: the (virtual) row id, < 2^30 - 1
constexpr uint32_t kStampMax = 127;         // step stamps run 1 .. kStampMax, then esr_rows_restamp and 1 again

// blocks of `kernel` (kBlock threads, no dynamic LDS) the whole device holds at once; kMaxGrid if the query fails.
// The step kernels walk contiguous slices, so a grid beyond one resident wave-set only adds a partly filled round.
inline int resident_blocks(const void* kernel, int cap_per_cu = 0) {
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    return kMaxGrid;
  }
  if (cap_per_cu > 0) per_cu = std::min(per_cu, cap_per_cu);
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
    (void)hipGetLastError();
    return kMaxGrid;
  }
  return std::min(kMaxGrid, per_cu * cus);
}

// Row-group geometry for a step kernel whose per-occurrence work is dominated by group-wide reductions (the triplet
// step: six dot products per occurrence): as FEW lanes per row as the register-resident row helpers allow -- four
// 16-byte chunks per lane, at least 8 lanes -- so that a reduction is 3 cross-lane steps instead of 5 and one wave
// instruction works on 8 rows instead of 2 (D = 128).  Every access is still whole 128-byte lines.
inline RowGeom step_geom_few_lanes(int D) {
  RowGeom g = row_geom(D);
  if (g.vec == 4)
    while (g.G > 8 && g.nch * 2 <= kMaxChunksPerLane) {
      g.G >>= 1;
      g.nch = (g.nvec + g.G - 1) / g.G;
    }
  return g;
}

#ifdef __HIPCC__
// the buffer (0 / 1) that held the row's value when the step with stamp T began, from the byte as it reads now
__device__ __forceinline__ uint32_t loc_at_step_begin(uint32_t byte, uint32_t T) {
  return ((byte >> 1) == T) ? ((byte & 1u) ^ 1u) : (byte & 1u);
}
// the byte a step with stamp T leaves on a row it moved into buffer `buf`
__device__ __forceinline__ uint8_t loc_written(uint32_t buf, uint32_t T) { return (uint8_t)((T << 1) | buf); }

// ---- order-free exact reduction of one double per workgroup, without a second launch and without a fence -------------
// (the scheme of merge_rows_epilogue in esr_inbatch_mfma.h.)  Thread 0 of every workgroup adds (its partial in 2^-frac
// fixed point) << 11 | 1 to one of kFixWords 64-bit words 128 B apart with ONE integer atomic: integer addition is exact
// and order-free, so the total is bit-reproducible, and the atomic's return value tells the workgroup whether it was the
// last of its word; that one forwards the word's total to the master word the same way, and the last arrival there has
// the grand total.  Data flows only through atomic return values: no ordering between addresses, no L2 write-back.
// acc layout (unsigned long long): [0] master, [8] flags (bit 0: a partial was not finite, bit 1: out of range),
// [16 * (1 + w)] word w.  All of it zero before the launch.  Range: |total| < 2^(52 - frac).
constexpr int kFixWords = 16;
constexpr int kFixAccWords = 16 * (1 + kFixWords);  // unsigned long longs per accumulator
__device__ __forceinline__ bool fixed_sum_arrive(unsigned long long* acc, double partial, int frac, unsigned nblocks,
                                                 double* total, unsigned* flags) {
  const unsigned wd = blockIdx.x % kFixWords;
  const unsigned on_word = (nblocks - wd + kFixWords - 1) / kFixWords;
  const unsigned nwords = nblocks < (unsigned)kFixWords ? nblocks : (unsigned)kFixWords;
  const double scaled = ldexp(partial, frac);
  unsigned long long add = ((unsigned long long)__double2ll_rn(scaled)) << 11;
  if (!(fabs(scaled) < 2251799813685248.0)) {  // 2^51: non-finite or out of range
    // raise the flag BEFORE this workgroup is counted: the add below consumes the OR's return value
    const unsigned r = atomicOr(reinterpret_cast<unsigned*>(acc + 8), (scaled != scaled) ? 1u : 2u);
    add = (unsigned long long)(r >> 2);
  }
  const unsigned long long old = atomicAdd(acc + 16 * (1 + wd), add + 1ull);
  if ((unsigned)(old & 2047ull) != on_word - 1) return false;
  const unsigned long long word_total = ((old + add) >> 11) << 11;
  const unsigned long long m = atomicAdd(acc, word_total + 1ull);
  if ((unsigned)(m & 2047ull) != nwords - 1) return false;
  const long long tot = ((long long)(m + word_total)) >> 11;  // arithmetic shift: signed sum
  *flags = atomicOr(reinterpret_cast<unsigned*>(acc + 8), 0u);
  *total = ldexp((double)tot, -frac);
  return true;
}

// ---- a sum of doubles carried by two integer words (hi at 2^-24, lo at 2^-76): exact for every term of magnitude
// 2^-24 .. 2^38, order-free, and readable by a workgroup of the SAME launch (the data travels in atomics only).  Used
// for the GloVe bias statistics that every workgroup of the update kernel needs before its first gradient.
__device__ __forceinline__ void fixed2_add(unsigned long long* hi_word, unsigned long long* lo_word, double v) {
  const double h = rint(ldexp(v, 24));
  const double l = rint(ldexp(v - ldexp(h, -24), 76));  // |v - h 2^-24| <= 2^-25: at most 2^51 after scaling
  const unsigned long long r0 = atomicAdd(hi_word, (unsigned long long)(long long)h);
  const unsigned long long r1 = atomicAdd(lo_word, (unsigned long long)(long long)l);
  // the returned values are demanded here, so both adds HAVE BEEN PERFORMED when this function returns: a workgroup
  // may count itself in afterwards and a reader that sees the count sees the sums
  asm volatile("" : : "v"(r0), "v"(r1) : "memory");
}
// two such sums at once: the four atomics travel together, one wait
__device__ __forceinline__ void fixed2_add2(unsigned long long* hi_a, unsigned long long* lo_a, double va,
                                            unsigned long long* hi_b, unsigned long long* lo_b, double vb) {
  const double ha = rint(ldexp(va, 24)), hb = rint(ldexp(vb, 24));
  const double la = rint(ldexp(va - ldexp(ha, -24), 76)), lb = rint(ldexp(vb - ldexp(hb, -24), 76));
  const unsigned long long r0 = atomicAdd(hi_a, (unsigned long long)(long long)ha);
  const unsigned long long r1 = atomicAdd(lo_a, (unsigned long long)(long long)la);
  const unsigned long long r2 = atomicAdd(hi_b, (unsigned long long)(long long)hb);
  const unsigned long long r3 = atomicAdd(lo_b, (unsigned long long)(long long)lb);
  asm volatile("" : : "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
}
__device__ __forceinline__ double fixed2_value(unsigned long long hi, unsigned long long lo) {
  return ldexp((double)(long long)hi, -24) + ldexp((double)(long long)lo, -76);
}
// agent-scope loads: served by the memory side, not by this XCD's (non-coherent) L2
__device__ __forceinline__ unsigned long long coherent_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned coherent_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// "this kernel runs": its first workgroup stores `value` to *flag (device memory; NULL = nobody asked).  Read by
// stream_gate_kernel (esr_stream_gate) on another stream.
__device__ __forceinline__ void announce_start(uint32_t* flag, uint32_t value) {
  if (flag && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int VEC, int NCH>
__device__ __forceinline__ void row_zero(RowRegs<VEC, NCH>& r) {
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.v[k][e] = 0.f;
}
#endif

}  // namespace esr
