"""Phase timing of inbatch3_kernel (debug build with -DESR_IB3_TIMING): cycles in barrier / S-phase / O-phase."""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, os.environ.get("IB3_LIB", "libib3dbg.so")))
ITERS = int(os.environ.get("IB3_ITERS", "62"))  # pipelined iterations per workgroup: chunks per split - 2
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
BF16 = os.environ.get("IB3_BF16", "0") == "1"   # bf16 tables -> the one-plane kernels
if BF16:
    qt, ct = q.to(torch.bfloat16), c.to(torch.bfloat16)
    ids = torch.arange(B, dtype=torch.int32, device=dev)
for _ in range(3):
    if BF16:
        rc = lib.esr_inbatch_towers_fwd_bwd_bf16x3(P(qt.data_ptr()), ctypes.c_int64(B), P(ct.data_ptr()), ctypes.c_int64(B), 1, D,
            P(ids.data_ptr()), P(ids.data_ptr()), None, None, ctypes.c_int64(B), ctypes.c_float(8.0), ctypes.c_float(0.1),
            ctypes.c_float(B), P(loss.data_ptr()), P(lse.data_ptr()), P(gq.data_ptr()), P(gc.data_ptr()),
            P(ws.data_ptr()), ctypes.c_size_t(ws.numel()), None)
    else:
        rc = lib.esr_inbatch_softmax_fwd_bwd_bf16x3(P(q.data_ptr()), P(c.data_ptr()), ctypes.c_int64(B), D, ctypes.c_float(8.0),
            ctypes.c_float(0.1), ctypes.c_float(B), P(loss.data_ptr()), P(lse.data_ptr()), P(gq.data_ptr()), P(gc.data_ptr()),
            P(ws.data_ptr()), ctypes.c_size_t(ws.numel()), None)
    assert rc == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8192)()
lib.esr_ib3_debug_read(buf)
e = np.array(buf[4096:], dtype=np.float64).reshape(1024, 4)
rt = e[:, 2] - e[:, 1]
a = np.array(buf[:4096], dtype=np.float64).reshape(256, 4, 4)  # last launch = pass C kernel
print("per-wave mean cycles over the pipelined iterations: barrier %.0f  S-phase(+VALU) %.0f  O-phase(+DMA) %.0f  total kernel %.0f"
      % tuple(a[..., k].mean() for k in range(4)))
print("per-iteration: barrier %.0f  S %.0f  O %.0f" % tuple(a[..., k].mean() / ITERS for k in range(3)))
print("min/max total", a[..., 3].min(), a[..., 3].max(), " loss", float(loss))
print("loop wall time %.1f us (s_memrealtime, 100 MHz) -> shader clock %.0f MHz" % (rt.mean() / 100, a[..., 3].mean() / (rt.mean() / 100)))
t0 = e[:, 0].min()
print("wall clock (us from the first wave's entry): entry %.1f..%.1f  loop start %.1f..%.1f  loop end %.1f..%.1f  stores done %.1f..%.1f"
      % tuple(x / 100 for c in range(4) for x in ((e[:, c] - t0).min(), (e[:, c] - t0).max())))
d = (e[:, 2] - e[:, 1]).reshape(256, 4).mean(1) / 100  # per workgroup loop time, us
print("loop us by XCD (blockIdx % 8):", np.round([d[x::8].mean() for x in range(8)], 1))
print("loop us by split (blockIdx % 4):", np.round([d[x::4].mean() for x in range(4)], 1))
print("loop us min/max over workgroups: %.1f %.1f; by blockIdx // 32:" % (d.min(), d.max()), np.round(d.reshape(8, 32).mean(1), 1))
cyc = a[..., 3].mean(1)
print("shader MHz by XCD:", np.round([(cyc[x::8] / d[x::8]).mean() for x in range(8)], 0))
print("loop cycles by XCD:", np.round([cyc[x::8].mean() for x in range(8)], 0))
