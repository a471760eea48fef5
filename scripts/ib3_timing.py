"""Phase timing of inbatch3_kernel (debug build with -DESR_IB3_TIMING): cycles in barrier / S-phase / O-phase."""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libib3dbg.so"))
lib.esr_inbatch3_workspace_bytes.restype = ctypes.c_size_t
lib.esr_inbatch3_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
dev = torch.device("cuda", 0)
B, D = 8192, 128
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
loss = torch.empty(1, device=dev); lse = torch.empty(B, device=dev); gq = torch.empty_like(q); gc = torch.empty_like(c)
ws = torch.empty(lib.esr_inbatch3_workspace_bytes(B, D), dtype=torch.uint8, device=dev)
P = ctypes.c_void_p
for _ in range(3):
    rc = lib.esr_inbatch_softmax_fwd_bwd_bf16x3(P(q.data_ptr()), P(c.data_ptr()), ctypes.c_int64(B), D, ctypes.c_float(8.0),
        ctypes.c_float(0.1), ctypes.c_float(B), P(loss.data_ptr()), P(lse.data_ptr()), P(gq.data_ptr()), P(gc.data_ptr()),
        P(ws.data_ptr()), ctypes.c_size_t(ws.numel()), None)
    assert rc == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 5120)()
lib.esr_ib3_debug_read(buf)
rt = np.array(buf[4096:], dtype=np.float64)
a = np.array(buf[:4096], dtype=np.float64).reshape(256, 4, 4)  # last launch = pass C kernel
print("per-wave mean cycles over 63 pipelined iterations: barrier %.0f  S-phase(+VALU) %.0f  O-phase(+DMA) %.0f  total kernel %.0f"
      % tuple(a[..., k].mean() for k in range(4)))
print("per-iteration: barrier %.0f  S %.0f  O %.0f" % tuple(a[..., k].mean() / 63 for k in range(3)))
print("min/max total", a[..., 3].min(), a[..., 3].max(), " loss", float(loss))
print("loop wall time %.1f us (s_memrealtime, 100 MHz) -> shader clock %.0f MHz" % (rt.mean() / 100, a[..., 3].mean() / (rt.mean() / 100)))
