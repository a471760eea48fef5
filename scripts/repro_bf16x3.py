import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from esrecsys_amd import ops
from oracle import stl_head as o_stl
dev = torch.device("cuda", 0)
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))
def run(B, D, scale, mq, mc, bs, dom, seed=0, precs=("f32", "bf16x3", "f16x2")):
    rng = np.random.default_rng(seed)
    q = (rng.standard_normal((B, D)) * mq / np.sqrt(D)).astype(np.float32)
    c = (rng.standard_normal((B, D)) * mc / np.sqrt(D)).astype(np.float32)
    if dom:
        j = B - 5
        c[j] = (3.0 * mc) * q[7] / max(np.linalg.norm(q[7]), 1e-20) * (1 if scale > 0 else -1)
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(q.astype(np.float64), c.astype(np.float64), 0.1, bs, scale, np.float64)
    out = {}
    for prec in precs:
        if prec != "f32" and ops.inbatch_split_path(prec, B, D, bf16_tables=False) is None:
            continue
        loss, lse, gq, gc = ops.inbatch_softmax_fwd_bwd(torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev), scale, 0.1, bs, precision=prec)
        lse_n = lse.cpu().numpy()
        bad_rows = np.where(~np.isfinite(lse_n))[0]
        out[prec] = ("%.2e" % abs(float(loss) - el), "%.2e" % rel(np.nan_to_num(lse_n, posinf=0, neginf=0), else_), "bad rows", bad_rows[:8].tolist(), len(bad_rows),
                     "gq nan rows", np.where(~np.isfinite(gq.cpu().numpy()).all(1))[0][:6].tolist(), "gc nan rows", int((~np.isfinite(gc.cpu().numpy()).all(1)).sum()))
    return out
for cfg in [(640, 100, -12.0, 2.97, 0.99, 77.0, True), (640, 128, -12.0, 2.97, 0.99, 77.0, True), (640, 100, -12.0, 2.97, 0.99, 77.0, False),
            (640, 100, 12.0, 2.97, 0.99, 77.0, True), (1024, 128, -12.0, 2.97, 0.99, 77.0, True), (640, 100, -12.0, 1.0, 0.99, 77.0, True)]:
    print(cfg, run(*cfg, precs=("bf16x3",)))
