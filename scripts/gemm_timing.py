"""Phase stamps of score_gemm_kernel (debug build -DESR_GEMM_TIMING -> scripts/libretrdbg.so)."""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libretrdbg.so"))
lib.esr_retrieve_workspace_bytes.restype = ctypes.c_size_t
lib.esr_retrieve_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda", 0)
nq, N, D, k = 8192, 8192 * 2, 512, 10
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((nq, D), generator=g, device=dev) * D ** -0.5
c = torch.randn((N, D), generator=g, device=dev) * D ** -0.5
P = ctypes.c_void_p
for mode in (0, 1):
    ws = torch.empty(lib.esr_retrieve_workspace_bytes(nq, N, D, k, mode), dtype=torch.uint8, device=dev)
    os_ = torch.empty((nq, k), device=dev); oi = torch.empty((nq, k), dtype=torch.int32, device=dev)
    for _ in range(3):
        rc = lib.esr_retrieve_topk(P(q.data_ptr()), P(c.data_ptr()), ctypes.c_int64(nq), ctypes.c_int64(N), D, k, mode,
                                   0, 1, P(os_.data_ptr()), P(oi.data_ptr()), P(ws.data_ptr()), ctypes.c_size_t(ws.numel()), None)
        assert rc == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8192)()
    lib.esr_gemm_debug_read(buf)
    a = np.array(buf[:], dtype=np.float64).reshape(1024, 8)   # the last GEMM launch (filtered chunk)
    nk = a[0, 4]
    tot = a[:, 0] + a[:, 1] + a[:, 2]
    print("mode %d: cycles/WG  prologue %.0f  main loop %.0f (%.0f per k-tile, %d k-tiles)  epilogue %.0f  | realtime ticks(100MHz) %.0f"
          " -> shader clock ~%.2f GHz" % (mode, a[:, 0].mean(), a[:, 1].mean(), a[:, 1].mean() / nk, nk, a[:, 2].mean(),
                                          a[:, 3].mean(), tot.mean() / a[:, 3].mean() / 10))
