"""Per-kernel durations of the in-batch head (esr_kernel_timing: HIP events around every launch of the library) and the
whole-op time, at the headline size.  python scripts/ib_ktime.py [B] [reps]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from esrecsys_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(1701)
D = 128
q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
for _ in range(50):
    out = ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision="f16x2")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision="f16x2")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("op: %.2f us  loss %.6f" % (dt * 1e6, float(out[0])))
lib.esr_kernel_timing(1)
for _ in range(reps):
    out = ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision="f16x2")
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
lib.esr_kernel_timing_read(buf, len(buf))
lib.esr_kernel_timing(0)
tot = 0.0
for line in buf.value.decode().strip().split("\n"):
    name, calls, total, mn, mx = line.split("\t")
    avg = float(total) / int(calls) * 1e3
    tot += avg
    print("  %-28s calls %4s  avg %8.2f us  min %8.2f  max %8.2f" % (name, calls, avg, float(mn) * 1e3, float(mx) * 1e3))
print("  sum of kernel averages: %.2f us" % tot)
