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
a = np.array(buf[:4096], dtype=np.float64).reshape(256, 4, 4)
rt = e[:, 2] - e[:, 1]
print("total per iteration %.0f cycles" % (a[..., 3].mean() / ITERS))
print("per-iteration cycles: barrier %.0f  phase1 %.0f  phase2 %.0f  (sum %.0f); loop %.1f us at %.0f MHz; kernel entry->loop end %.1f..%.1f us"
      % (a[..., 0].mean() / ITERS, a[..., 1].mean() / ITERS, a[..., 2].mean() / ITERS, a[..., :3].sum(-1).mean() / ITERS,
         rt.mean() / 100, a[..., 3].mean() / (rt.mean() / 100), (e[:, 2] - e[:, 0].min()).min() / 100,
         (e[:, 2] - e[:, 0].min()).max() / 100))
