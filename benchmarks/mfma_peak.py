"""What the matrix pipes of this box sustain: a register-only MFMA loop (esr_probe_mfma), no memory traffic.
One JSON line per dtype; `sustained_TFLOPs` is the ceiling every roofline fraction in this repo should be read
against (the data-sheet peaks assume the maximum clock)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from esrecsys_amd import _lib
    lib = _lib.load_probe()
    dev = torch.device("cuda", 0)
    sink = torch.zeros(1, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    # constants: the pipes' issue ceiling; live data, run long enough (~0.5 s) for the power controller to settle: the
    # ceiling a real bf16 GEMM can reach on this box
    for name, dtype, peak, runs in (
            ("bf16 32x32x16", _lib.ESR_BF16, 2500.0, ((256 * 8, 20000), (256 * 8, 100000))),
            ("bf16 32x32x16 live data", _lib.ESR_BF16 | _lib.ESR_PROBE_LIVE_DATA, 2500.0,
             ((256 * 8, 20000), (256 * 8, 100000), (256 * 8, 1000000))),
            ("f16 32x32x16 live data", _lib.ESR_PROBE_F16 | _lib.ESR_PROBE_LIVE_DATA, 2500.0,
             ((256 * 8, 20000), (256 * 8, 100000), (256 * 8, 1000000))),
            ("f32 32x32x2", _lib.ESR_F32, 157.3, ((256 * 8, 20000), (256 * 8, 100000)))):
        for wgs, iters in runs:
            flops = ctypes.c_double()
            _lib.check(lib.esr_probe_mfma(dtype, wgs, 1000, sink.data_ptr(), ctypes.byref(flops), st), "probe")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.esr_probe_mfma(dtype, wgs, iters, sink.data_ptr(), ctypes.byref(flops), st), "probe")
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            print(json.dumps({"probe": "mfma " + name, "workgroups": wgs, "iters": iters, "ms": ms,
                              "sustained_TFLOPs": flops.value / ms / 1e9, "datasheet_peak_TFLOPs": peak,
                              "frac_of_datasheet": flops.value / ms / 1e9 / peak}), flush=True)


if __name__ == "__main__":
    main()
