"""Time ops.sparse_adagrad at the sizes of the steps (HIP events, 100 reps)."""
import sys, torch
sys.path.insert(0, ".")
from esrecsys_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for V, D, n in ((1_000_000, 128, 16384), (1_000_000, 128, 24576), (465_537, 256, 131072), (1_000_000, 128, 131072)):
    table = torch.randn((V, D), generator=g, device=dev)
    accum = torch.full((V, D), 0.1, device=dev)
    ids = torch.randint(0, V, (n,), generator=g, device=dev, dtype=torch.int32)
    grads = torch.randn((n, D), generator=g, device=dev) * 0.01
    sid, perm = ops.segment_sort(ids, V)
    for _ in range(5):
        ops.sparse_adagrad(table, accum, sid, perm, grads, 0.01)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ops.sparse_adagrad(table, accum, sid, perm, grads, 0.01)
    e1.record()
    torch.cuda.synchronize()
    print(V, D, n, round(e0.elapsed_time(e1) / 100 * 1000, 2), "us")
