"""Segment stamps of inbatch2h_pct_kernel (probe build -DH_TIMING=2, IB2H_LIB): per chunk, waves 0-3 of every other
workgroup: cycles waiting at the top barrier / in LOAD / at the middle barrier + in COMPUTE (COMPUTE alone separately)."""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, os.environ["IB2H_LIB"]))
lib.esr_inbatch2h_workspace_bytes.restype = ctypes.c_size_t
lib.esr_inbatch2h_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
dev = torch.device("cuda", 0)
B, D = 8192, 128
ITERS = 32
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
keep = e[:, 0] > 0
e, a = e[keep], a[keep]
rt = (e[:, 2] - e[:, 0]) / 100.0   # us (100 MHz)
print("%s: %d waves; per chunk: top-barrier wait %.0f  LOAD %.0f  mid barrier + COMPUTE %.0f (COMPUTE %.0f)  total %.0f cycles; "
      "kernel entry -> loop end %.1f us (min %.1f max %.1f) at %.0f MHz" % (
          os.environ["IB2H_LIB"], len(e), a[:, 0].mean() / ITERS, a[:, 1].mean() / ITERS, a[:, 2].mean() / ITERS,
          e[:, 3].mean() / ITERS, a[:, 3].mean() / ITERS, rt.mean(), rt.min(), rt.max(), a[:, 3].mean() / rt.mean()))
