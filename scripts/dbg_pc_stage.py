import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from esrecsys_amd import ops
dev = torch.device("cuda", 0)
for B in (256, 512):
    g = torch.Generator(device=dev).manual_seed(B)
    D = 128
    q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
    c[B - 40] = 3.0 * q[7] / q[7].norm()
    outs = {}
    for mode in ("32", "64"):
        os.environ["ESR_IB2H_Q"] = mode
        for pcm in ("stage", "dma"):
            os.environ["ESR_IB2H_PC"] = pcm
            o = [t.clone() for t in ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision="f16x2")]
            outs[(mode, pcm)] = o
            print(B, mode, pcm, "loss", float(o[0]), "finite", [bool(torch.isfinite(t).all()) for t in o])
    ref = outs[("32", "dma")]
    for k, o in outs.items():
        print(B, k, [float((a - b).abs().max() / b.abs().max()) for a, b in zip(o, ref)])
