export TMPDIR=/tmp
cat > /tmp/ivf_one.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
import numpy as np
from esrecsys_amd.ivf import IVFIndex
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
N, D, nq = 1 << 20, 512, 8192
cent = torch.randn((1024, D), generator=g, device=dev); cent = cent / cent.norm(dim=1, keepdim=True)
pick = torch.randint(0, 1024, (N,), generator=g, device=dev)
c = cent[pick] + 0.5 * torch.randn((N, D), generator=g, device=dev) / D ** 0.5
q = cent[torch.randint(0, 1024, (nq,), generator=g, device=dev)] + 0.5 * torch.randn((nq, D), generator=g, device=dev) / D ** 0.5
idx = IVFIndex(c, 1024, iters=2)
for _ in range(3):
    s, i = idx.search(q, int(sys.argv[1]), int(sys.argv[2]))
torch.cuda.synchronize()
PY
rm -rf /tmp/ivft; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ivft -o t -- python /tmp/ivf_one.py ${K:-10} ${NP:-32} > /tmp/ivft.log 2>&1
python3 scripts/trace_gaps.py /tmp/ivft ivf_prep 2 ${COUNT:-14} | cut -c1-120
