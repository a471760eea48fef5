"""Per-kernel durations of one brute-force retrieval (esr_kernel_timing): python scripts/retr_ktime.py [mode] [N] [k]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from esrecsys_amd import _lib, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_048_576
k = int(sys.argv[3]) if len(sys.argv) > 3 else 500
dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(1701)
D, nq = 512, 8192
q = torch.randn((nq, D), generator=g, device=dev) * D ** -0.5
c = torch.randn((N, D), generator=g, device=dev) * D ** -0.5
for _ in range(2):
    ops.retrieve_topk(q, c, k, mode=mode)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ops.retrieve_topk(q, c, k, mode=mode)
torch.cuda.synchronize()
print("op: %.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
lib.esr_kernel_timing(1)
ops.retrieve_topk(q, c, k, mode=mode)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
lib.esr_kernel_timing_read(buf, len(buf))
lib.esr_kernel_timing(0)
for line in buf.value.decode().strip().split("\n"):
    name, calls, total, mn, mx = line.split("\t")
    print("  %-24s calls %4s  total %8.3f ms  avg %8.1f us  min %8.1f  max %8.1f" % (
        name, calls, float(total), float(total) / int(calls) * 1e3, float(mn) * 1e3, float(mx) * 1e3))
