"""Phase stamps of the fp16 x 2 in-batch kernels (debug build -DH_TIMING=1: pass Q, =2: pass C)."""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, os.environ["IB2H_LIB"]))
ITERS = int(os.environ["IB2H_ITERS"])
lib.esr_inbatch2h_workspace_bytes.restype = ctypes.c_size_t
lib.esr_inbatch2h_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
dev = torch.device("cuda", 0)
B, D = 8192, 128
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
loss = torch.empty(1, device=dev); lse = torch.empty(B, device=dev); gq = torch.empty_like(q); gc = torch.empty_like(c)
ws = torch.empty(lib.esr_inbatch2h_workspace_bytes(B, D), dtype=torch.uint8, device=dev)
P = ctypes.c_void_p
for _ in range(20):
    rc = lib.esr_inbatch_softmax_fwd_bwd_f16x2(P(q.data_ptr()), P(c.data_ptr()), ctypes.c_int64(B), D, ctypes.c_float(8.0),
        ctypes.c_float(0.1), ctypes.c_float(B), P(loss.data_ptr()), P(lse.data_ptr()), P(gq.data_ptr()), P(gc.data_ptr()),
        P(ws.data_ptr()), ctypes.c_size_t(ws.numel()), None)
    assert rc == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8192)()
lib.esr_ib2h_debug_read(buf)
e = np.array(buf[4096:], dtype=np.float64).reshape(1024, 4)
a = np.array(buf[:4096], dtype=np.float64).reshape(1024, 4)
keep = e[:, 0] > 0   # (slots of workgroups / waves that do not exist stay zero)
e, a = e[keep], a[keep]
print("%d waves stamped" % len(e))
rt = e[:, 2] - e[:, 1]
print("total per iteration %.0f cycles" % (a[..., 3].mean() / ITERS))
print("per-iteration cycles: barrier %.0f  phase1 %.0f  phase2 %.0f  (sum %.0f); loop %.1f us at %.0f MHz; kernel entry->loop end %.1f..%.1f us"
      % (a[..., 0].mean() / ITERS, a[..., 1].mean() / ITERS, a[..., 2].mean() / ITERS, a[..., :3].sum(-1).mean() / ITERS,
         rt.mean() / 100, a[..., 3].mean() / (rt.mean() / 100), (e[:, 2] - e[:, 0].min()).min() / 100,
         (e[:, 2] - e[:, 0].min()).max() / 100))
tail = e[:, 3] - e[:, 2]
if tail.max() > 0:
    print("loop end -> end of the sweep's last chunk: %.1f us (min %.1f, max %.1f); kernel entry -> that point %.1f..%.1f us"
          % (tail.mean() / 100, tail.min() / 100, tail.max() / 100, (e[:, 3] - e[:, 0].min()).min() / 100,
             (e[:, 3] - e[:, 0].min()).max() / 100))
    print("kernel entry -> loop start: %.1f us (mean)" % ((e[:, 1] - e[:, 0]).mean() / 100))
ent = e[:, 0]
print("workgroup entry spread: %.1f us; by quarter of the grid (mean entry - first): %s" % (
    (ent.max() - ent.min()) / 100, ["%.1f" % ((q.mean() - ent.min()) / 100) for q in np.array_split(ent, 4)]))
end = e[:, 3] if tail.max() > 0 else e[:, 2]
print("entry (us after the first): min %.1f  median %.1f  max %.1f;  end: min %.1f  median %.1f  max %.1f" % (
    0.0, (np.median(ent) - ent.min()) / 100, (ent.max() - ent.min()) / 100, (end.min() - ent.min()) / 100,
    (np.median(end) - ent.min()) / 100, (end.max() - ent.min()) / 100))
h, edges = np.histogram((ent - ent.min()) / 100, bins=8)
print("entry histogram (us):", [("%.0f-%.0f" % (edges[i], edges[i + 1]), int(h[i])) for i in range(8)])
if os.environ.get("IB2H_BY_BLOCK"):
    # pass C: is a workgroup whose P' tiles pass Q wrote LAST (still in the Infinity Cache) faster than one whose tiles were
    # written first?  Stamped: waves 0..3 of every even workgroup; workgroup = 256-row block x split (IB2H_BY_BLOCK = nsplit)
    ns = int(os.environ["IB2H_BY_BLOCK"])
    raw = np.array(buf[4096:], dtype=np.float64).reshape(1024, 4)
    idx = np.nonzero(raw[:, 0] > 0)[0]
    blk = 2 * (idx // 4)
    ob, w = blk // ns, idx % 4
    jt = 8 * ob + w                       # the wave's owned 32-row tile = pass Q's streamed chunk index
    dur = (raw[idx, 2] - raw[idx, 1]) / 100
    for cls in range(4):
        m = (jt % 32) // 8 == cls
        if m.any():
            print("tiles pass Q wrote in its iterations %2d..%2d of 32: loop %.1f us (min %.1f max %.1f, %d waves)"
                  % (8 * cls, 8 * cls + 7, dur[m].mean(), dur[m].min(), dur[m].max(), int(m.sum())))
