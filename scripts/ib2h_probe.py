"""Time the fp16 x 2 in-batch op of a probe build (IB2H_LIB = a .so built from esr_inbatch2h.hip + esr_core.hip with -D
probe flags; values may be wrong, only the timing is of interest).  Run under rocprofv3 for per-kernel averages."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, os.environ["IB2H_LIB"]))
lib.esr_inbatch2h_workspace_bytes.restype = ctypes.c_size_t
lib.esr_inbatch2h_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
dev = torch.device("cuda", 0)
B, D = int(os.environ.get("IB2H_B", "8192")), 128
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
loss = torch.empty(1, device=dev); lse = torch.empty(B, device=dev); gq = torch.empty_like(q); gc = torch.empty_like(c)
ws = torch.empty(lib.esr_inbatch2h_workspace_bytes(B, D), dtype=torch.uint8, device=dev)
P = ctypes.c_void_p
def run():
    rc = lib.esr_inbatch_softmax_fwd_bwd_f16x2(P(q.data_ptr()), P(c.data_ptr()), ctypes.c_int64(B), D, ctypes.c_float(8.0),
        ctypes.c_float(0.1), ctypes.c_float(B), P(loss.data_ptr()), P(lse.data_ptr()), P(gq.data_ptr()), P(gc.data_ptr()),
        P(ws.data_ptr()), ctypes.c_size_t(ws.numel()), None)
    assert rc == 0
for _ in range(10): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 100
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
print("%s: op %.1f us, loss %.6f" % (os.environ["IB2H_LIB"], e0.elapsed_time(e1) / n * 1e3, float(loss)))
