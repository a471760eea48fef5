// Shared pieces of the one-pass train steps on DOUBLE-BUFFERED tables (esr_glove.hip, esr_triplet_step.hip).
//
// A table that such a step updates lives in two [V, D] buffers plus one byte per row (`loc`) that says which buffer
// holds the row's current value.  A step reads rows where `loc` pointed when it began (resolved up front by its plan
// kernel into 32-bit row codes), writes every updated row into the OTHER buffer and flips the byte: readers and writers
// of one launch never touch the same bytes, so gradients can be formed on chip from rows that other workgroups are
// rewriting, and no gradient row or snapshot ever goes to memory.
#pragma once
#include "esr_common.h"

namespace esr {

constexpr int kStepChunk = 32;  // == kSegChunk of esr_optim.hip: same cut points, same association of every sum
constexpr uint32_t kLocBit = 0x80000000u;   // row code: the row's current value is in the second buffer
constexpr uint32_t kSideBit = 0x40000000u;  // (GloVe plan records) the occurrence is the pair's second token
constexpr uint32_t kIdMask = 0x3FFFFFFFu;   // row code: the (virtual) row id, < 2^30 - 1

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
template <int VEC, int NCH>
__device__ __forceinline__ void row_zero(RowRegs<VEC, NCH>& r) {
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.v[k][e] = 0.f;
}
#endif

}  // namespace esr
