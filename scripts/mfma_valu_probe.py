"""Does the vector ALU run beside the matrix pipe on this box?  esr_probe_mfma_valu (include/esr_probe.h): per round four
MFMAs (32 matrix-pipe cycles each) with nv plain VALU instructions + nt v_exp_f32 behind each; cycles per round and wave,
for one and two waves per SIMD.  If the two overlapped fully a round would cost max(128, VALU); fully serial, their sum."""
import sys
import torch
from esrecsys_amd import _lib

lib = _lib.load_probe()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
sink = torch.zeros(1, device=dev)
iters = 2000
print("%-28s %10s %10s" % ("per MFMA: nv VALU + nt exp", "1 wave/SIMD", "2 waves/SIMD"))
for grouped in (0, 1):
    for nv, nt in ((0, 0), (1, 0), (2, 0), (4, 0), (6, 0), (7, 0), (8, 0), (12, 0), (0, 1), (0, 2), (3, 1), (4, 1)):
        row = []
        for wps in (1, 2):
            cyc = torch.zeros(256 * 4 * wps, dtype=torch.int64, device=dev)
            for _ in range(2):
                _lib.check(lib.esr_probe_mfma_valu(nv, nt, grouped, wps, 256, iters, cyc.data_ptr(), sink.data_ptr(), st), "probe")
            torch.cuda.synchronize()
            row.append(cyc.double().mean().item() / iters)
        print("%-28s %10.1f %10.1f   (cycles per round of 4 MFMAs%s)" % (
            "nv=%d nt=%d%s" % (nv, nt, " grouped" if grouped else ""), row[0], row[1],
            "; per wave, two waves share the SIMD" if False else ""))

names = ["v_fma_f32", "v_pk_fma_f32", "v_fma_mix_f32", "v_cvt_pk_f16_f32", "v_max3_f32", "v_pk_add_f32", "v_exp_f32",
         "v_fma_f32 dependent chain", "v_pk_fma_f32 dependent chain", "v_exp_f32 -> v_fma_f32 chains"]
print("four instructions of one kind behind every MFMA:")
for kind, name in enumerate(names):
    row = []
    for wps in (1, 2):
        cyc = torch.zeros(256 * 4 * wps, dtype=torch.int64, device=dev)
        for _ in range(2):
            _lib.check(lib.esr_probe_mfma_valu(-1, kind, 0, wps, 256, iters, cyc.data_ptr(), sink.data_ptr(), st), "probe")
        torch.cuda.synchronize()
        row.append(cyc.double().mean().item() / iters)
    print("%-32s %10.1f %10.1f" % (name, row[0], row[1]))

mix = ["as in inbatch1h_kernel (22 VALU per round)", "conversions by v_fma_mixlo / mixhi_f16", "v_exp_f32 -> v_fma_f32",
       "without the sums and the maximum", "22 independent v_fma_f32", "the four MFMAs alone", "all 22 behind the fourth MFMA",
       "conversions by v_cvt_pkrtz_f16_f32", "every consumer one round behind its producer", "one round behind + pkrtz"]
print("the one-plane kernel's exp / split between the four MFMAs of a round:")
for kind, name in enumerate(mix):
    row = []
    for wps in (1, 2):
        cyc = torch.zeros(256 * 4 * wps, dtype=torch.int64, device=dev)
        for _ in range(2):
            _lib.check(lib.esr_probe_mfma_valu(-2, kind, 0, wps, 256, iters, cyc.data_ptr(), sink.data_ptr(), st), "probe")
        torch.cuda.synchronize()
        row.append(cyc.double().mean().item() / iters)
    print("%-46s %10.1f %10.1f" % (name, row[0], row[1]))
