"""Pure-read HBM bandwidth on this box: torch reductions over buffers larger than the 256 MB Infinity Cache."""
import torch
dev = torch.device("cuda", 0)
for mb, dt in [(1024, torch.float32), (268, torch.float32), (4096, torch.float32), (1024, torch.bfloat16)]:
    n = mb * 1024 * 1024 // torch.tensor([], dtype=dt).element_size()
    x = torch.ones(n, dtype=dt, device=dev)
    for _ in range(3): x.sum()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): x.sum()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print("sum over %5d MB %s: %.1f us, %.2f TB/s" % (mb, str(dt)[6:], t * 1e6, mb * 1.048576e6 / t / 1e12))
    # read after write of the same buffer (what pass C sees: P was just written by pass Q)
    e0.record()
    for _ in range(20):
        x.fill_(1.0); x.sum()
    e1.record(); torch.cuda.synchronize()
    t2 = e0.elapsed_time(e1) / 20 * 1e-3
    print("   fill + sum: %.1f us (fill alone ~%.1f us)" % (t2 * 1e6, (t2 - t) * 1e6))
