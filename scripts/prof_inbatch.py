"""Runs only the in-batch score kernels (B = 8192, D = 128) a few times; used under rocprofv3 --pmc."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrecsys_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
B, D = 8192, 128
q = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
c = torch.randn((B, D), generator=g, device=dev) * D ** -0.5
for _ in range(n):
    out = ops.inbatch_softmax_fwd_bwd(q, c, 8.0, 0.1, float(B), precision=prec)
torch.cuda.synchronize()
print("ok", float(out[0]))
