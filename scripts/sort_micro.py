"""Time ops.segment_sort at the list sizes of the steps (HIP events, 200 reps)."""
import sys, torch
sys.path.insert(0, ".")
from esrecsys_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for n in (8192, 16384, 24576, 32768):
    ids = torch.randint(0, 1_000_000, (n,), generator=g, device=dev, dtype=torch.int32)
    for _ in range(10):
        ops.segment_sort(ids, 1_000_000)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.segment_sort(ids, 1_000_000)
    e1.record()
    torch.cuda.synchronize()
    print(n, round(e0.elapsed_time(e1) / 200 * 1000, 2), "us")
