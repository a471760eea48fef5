import torch, sys
sys.path.insert(0, "/root/repo")
from esrecsys_amd import ops
dev = torch.device("cuda", 0)
B = 8192
for D in (32, 64, 96, 128):
    g = torch.Generator(device=dev).manual_seed(D)
    q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    for prec in ("f32", "f16x2", "bf16x3"):
        for _ in range(5): ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision=prec)
        e1.record(); torch.cuda.synchronize()
        print("D=%3d %-7s %.1f us" % (D, prec, e0.elapsed_time(e1) / 50 * 1e3))
