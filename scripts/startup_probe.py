"""Where does a SHORT timed region (the driver's --steps 20 --warmup 5) lose time against a 200-step one?

Replays bench.py's headline leg (train_steps over in-batch batches) with a HIP event + host stamp in front of every
train_step call, for (warmup, steps) = (5, 20), (20, 200) and (5, 20) behind 100 extra untimed steps."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from esrecsys_amd.pinterest import train_shop_the_look as tstl  # noqa: E402


def run(workload, warmup, steps, prewarm=0):
    dev = torch.device("cuda", 0)
    cfg = dict(bench.WORKLOADS[workload], table_dtype="f32", ids="uniform")
    B = cfg["B"]
    state, batches = bench.make_state_and_batches(workload, cfg, dev, warmup + steps + prewarm, 0)
    if workload == "inbatch":
        wb = [(b[0], b[1], None) for b in batches]
        kw = dict(scale=bench.SCALE, precision="auto")
    else:
        wb, kw = batches, {}
    stamps = []
    if workload == "inbatch":
        orig = tstl.train_step

        def stamped(*a, **k):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            stamps.append((time.perf_counter(), e))
            return orig(*a, **k)
        tstl.train_step = stamped
    if prewarm:
        state, _ = tstl.train_steps(state, iter(wb[:prewarm]), prewarm, bench.LAM, B, **kw)
        wb = wb[prewarm:]
    state, _ = tstl.train_steps(state, iter(wb[:warmup]), warmup, bench.LAM, B, **kw)
    torch.cuda.synchronize()
    stamps.clear()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    state, losses = tstl.train_steps(state, iter(wb[warmup:]), steps, bench.LAM, B, **kw)
    t_issued = time.perf_counter()
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    torch.cuda.synchronize()
    t_sync = time.perf_counter()
    fl = float(losses[-1])
    t_end = time.perf_counter()
    if workload == "inbatch":
        tstl.train_step = orig
    out = {"workload": workload, "warmup": warmup, "steps": steps, "prewarm": prewarm,
           "wall_ms": (t_sync - t0) * 1e3, "host_issue_ms": (t_issued - t0) * 1e3, "gpu_ms": e0.elapsed_time(e1),
           "float_loss_ms": (t_end - t_sync) * 1e3, "ms_per_step": (t_sync - t0) * 1e3 / steps, "loss": fl}
    if stamps:
        out["host_stamp_ms"] = [round((s[0] - t0) * 1e3, 3) for s in stamps]
        out["gpu_stamp_ms"] = [round(e0.elapsed_time(s[1]), 3) for s in stamps]
    return out


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "inbatch"
    for (w, k, p) in [(5, 20, 0), (5, 20, 0), (20, 200, 0), (5, 20, 100), (5, 20, 0)]:
        r = run(wl, w, k, p)
        if "gpu_stamp_ms" in r and len(r["gpu_stamp_ms"]) > 40:
            g = r["gpu_stamp_ms"]
            r["gpu_stamp_ms"] = g[:24] + ["..."] + g[-4:]
            h = r["host_stamp_ms"]
            r["host_stamp_ms"] = h[:24] + ["..."] + h[-4:]
        print(json.dumps(r), flush=True)
