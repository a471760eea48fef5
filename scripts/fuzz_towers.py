"""Randomised check of the in-batch head straight from the tower tables (esr_inbatch_towers_fwd_bwd_f16x2 / _bf16x3;
fp32 and bf16 tables, gradient rows scattered to given positions or not) against the fp64 oracle: B, D, temperature
(both signs), magnitudes, duplicate ids, a far-out candidate.  SEED, CASES."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from esrecsys_amd import ops
from oracle import stl_head as o_stl
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N_ = int(os.environ.get("CASES", "40"))
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))
bad = 0
for case in range(N_):
    B = int(rng.choice([128, 256, 640, 1024, 2048, 4096]))
    D = int(rng.choice([32, 64, 96, 100, 128]))
    V = int(rng.choice([B // 2, 4 * B, 200000]))
    scale = float(rng.choice([-12.0, -2.0, 0.7, 4.0, 8.0, 16.0]))
    mq, mc = float(10 ** rng.uniform(-1.5, 0.3)), float(10 ** rng.uniform(-1.5, 0.3))
    bf16 = rng.random() < 0.4
    st = (rng.standard_normal((V, D)) * mq / np.sqrt(D)).astype(np.float32)
    pt = (rng.standard_normal((V, D)) * mc / np.sqrt(D)).astype(np.float32)
    sid = rng.integers(0, V, B).astype(np.int32); pid = rng.integers(0, V, B).astype(np.int32)
    if case % 3 == 0:
        pt[pid[B - 3]] = (3.0 * mc) * st[sid[7]] / max(np.linalg.norm(st[sid[7]]), 1e-20) * (1 if scale > 0 else -1)
    std, ptd = torch.from_numpy(st).to(dev), torch.from_numpy(pt).to(dev)
    if bf16:
        std, ptd = std.to(torch.bfloat16), ptd.to(torch.bfloat16)
        st, pt = std.float().cpu().numpy(), ptd.float().cpu().numpy()   # the oracle sees the rounded tables
    bs = float(rng.choice([B, 77.0]))
    el, else_, egq, egc = o_stl.inbatch_softmax_loss_and_grads(st[sid].astype(np.float64), pt[pid].astype(np.float64), 0.1, bs, scale, np.float64)
    for prec in ("auto", "f16x2", "bf16x3"):
        try:
            if ops.inbatch_split_path(prec, B, D, bf16_tables=bf16) is None:
                continue
        except ValueError:   # an explicit precision the shape / table dtype does not take
            continue
        scat = rng.random() < 0.5
        gp = None
        if scat:
            perm = rng.permutation(2 * B).astype(np.int32)
            gp = (torch.from_numpy(perm[:B]).to(dev), torch.from_numpy(perm[B:]).to(dev))
        loss, lse, gq, gc = ops.inbatch_towers_fwd_bwd(std, ptd, torch.from_numpy(sid).to(dev), torch.from_numpy(pid).to(dev),
                                                       scale, 0.1, bs, grad_positions=gp, precision=prec)
        if scat:
            buf = gq.cpu().numpy(); gqn, gcn = buf[perm[:B]], buf[perm[B:]]
        else:
            gqn, gcn = gq.cpu().numpy(), gc.cpu().numpy()
        errs = (abs(float(loss) - el) / abs(el), rel(lse.cpu().numpy(), else_), rel(gqn, egq), rel(gcn, egc))
        ok = bool(np.all(np.isfinite(errs))) and max(errs) <= 1e-5
        if os.environ.get("VERBOSE") == "1" or not ok:
            print("ok  " if ok else "MISMATCH", dict(prec=prec, B=B, D=D, V=V, scale=scale, mq=mq, mc=mc, bf16=bf16, scat=scat, bs=bs, case=case),
                  ["%.1e" % e for e in errs], flush=True)
        bad += 0 if ok else 1
print("cases", N_, "mismatches", bad)
