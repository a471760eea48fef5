"""Pure-read HBM bandwidth on this box: torch reductions over buffers larger than the 256 MB Infinity Cache."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

# the library's own read probe: contiguous slice per workgroup, eight 16-byte loads in flight per lane
import ctypes
from esrecsys_amd import _lib
lib = _lib.load_probe()
sink = torch.zeros(1, device=dev)
for mb in (268, 1024):
    x = torch.ones(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    for wgs in (256, 512, 1024, 2048, 4096):
        for nt in (0, 1):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                _lib.check(lib.esr_probe_hbm_read(x.data_ptr(), x.numel() * 4, wgs, nt, sink.data_ptr(), st), "probe")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _lib.check(lib.esr_probe_hbm_read(x.data_ptr(), x.numel() * 4, wgs, nt, sink.data_ptr(), st), "probe")
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 20 * 1e-3
            print("esr_probe_hbm_read %5d MB, %4d workgroups, nt=%d: %.1f us, %.2f TB/s" % (mb, wgs, nt, t * 1e6, x.numel() * 4 / t / 1e12))
