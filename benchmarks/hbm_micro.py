"""HBM micro-benchmarks of the memory-bound kernels at launch sizes from the headline batch (B = 8192) up
to saturating (2^22 ids), so the achieved GB/s can be read against the 8 TB/s HBM3E roofline without the
launch-latency floor that dominates an 8 MB gather.  Prints one JSON line per measurement.

    python benchmarks/hbm_micro.py [--reps 20]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from esrecsys_amd import ops  # noqa: E402

PEAK = 8000.0


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def emit(name, n, D, nbytes, t, note=""):
    print(json.dumps({"kernel": name, "n": n, "D": D, "us": t * 1e6, "algorithmic_GB": nbytes / 1e9,
                      "GBps": nbytes / t / 1e9, "frac_of_8TBps": nbytes / t / 1e9 / PEAK, "note": note}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1701)

    # stream-copy ceiling measured on this box (1 GiB read + 1 GiB write)
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_(generator=g)
    b = torch.empty_like(a)
    t = timed(lambda: b.copy_(a), args.reps)
    emit("torch_copy_ceiling", a.numel(), 1, 2 * a.numel() * 4, t, "hipMemcpy D2D of 1 GiB")
    del a, b

    for (V, D) in ((1_000_000, 128), (465_537, 256)):
        table = torch.randn((V, D), generator=g, device=dev)
        accum = torch.full((V, D), 0.1, device=dev)
        for n in (16_384, 131_072, 1 << 20, 1 << 22):
            if n > 4 * V:
                continue
            ids = torch.randint(0, V, (n,), generator=g, device=dev, dtype=torch.int32)
            out = torch.empty((n, D), device=dev)
            t = timed(lambda: ops.gather_rows(table, ids, out=out), args.reps)
            emit("gather_rows", n, D, 2 * n * D * 4, t, "read n rows + write n rows")
            grads = torch.randn((n, D), generator=g, device=dev) * 0.01
            sid, perm = ops.segment_sort(ids, V)
            uniq = int(torch.unique(ids).numel())
            t = timed(lambda: ops.sparse_adagrad(table, accum, sid, perm, grads, 0.01), args.reps)
            emit("sparse_adagrad", n, D, (n + 4 * uniq) * D * 4, t,
                 "grad rows read once + param/accum RMW of %d distinct rows" % uniq)
            t = timed(lambda: ops.segment_sort(ids, V), args.reps)
            emit("segment_sort", n, 1, 0, t, "stable sort of (id, occurrence) pairs: one-launch bitonic up to 2048 ids, tile sort + rank up to 32768, own 11-bit LSD radix sort up to 2^21 ids (rocPRIM only beyond that)")
            if D == 128 and n <= (1 << 20):
                t3 = timed(lambda: ops.triplet_fwd_bwd(table, table, table, ids[: n // 3], ids[n // 3: 2 * (n // 3)],
                                                       ids[2 * (n // 3): 3 * (n // 3)], n // 3, 0.1, n // 3,
                                                       want_scores=False), args.reps)
                emit("triplet_fused", n // 3, D, 6 * (n // 3) * D * 4, t3, "3 rows read + 3 grad rows written per triplet")
            if D == 256 and n <= (1 << 20):
                B = n // 2
                bias = torch.zeros((V, 1), device=dev)
                target = torch.rand(B, generator=g, device=dev) * 300
                inp = ids[: 2 * B].reshape(2, B).contiguous()
                tg = timed(lambda: ops.glove_fwd_bwd(table, bias, inp, target), args.reps)
                emit("glove_fused", B, D, 4 * B * D * 4, tg, "2 rows read + 2 grad rows written per pair")
        del table, accum


if __name__ == "__main__":
    main()
