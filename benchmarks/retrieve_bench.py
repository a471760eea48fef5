"""Config-5 retrieval micro-benchmark: batched score GEMM + top-k on one MI355X's share of the candidates.
One JSON line per (mode, k).  f32-equivalent flops = 2 nq N D; executed bf16 MFMA flops = 6x that in exact mode.
Usage: python benchmarks/retrieve_bench.py [--nq 8192] [--N 131072] [--D 512] [--reps 5]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=8192)
    ap.add_argument("--N", type=int, default=131072)
    ap.add_argument("--D", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ks", default="10,500")
    ap.add_argument("--modes", default="exact,bf16,ann")
    a = ap.parse_args()
    from esrecsys_amd import ops
    from esrecsys_amd.pinterest.make_recommendations import find_top_k_batch, recall_at_k
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1701)
    q = torch.randn((a.nq, a.D), generator=g, device=dev) * a.D ** -0.5
    c = torch.randn((a.N, a.D), generator=g, device=dev) * a.D ** -0.5
    for k in [int(x) for x in a.ks.split(",")]:
        exact_i = None
        for mode in a.modes.split(","):
            if mode == "ann":
                fn = lambda: find_top_k_batch(q, c, k, approximate=True)  # noqa: E731
            else:
                fn = lambda: ops.retrieve_topk(q, c, k, mode=mode)  # noqa: E731
            out = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            if mode == "exact":
                exact_i = out[1]
            flops = 2.0 * a.nq * a.N * a.D
            planes = 6 if mode == "exact" else 1
            rec = {"op": "retrieve_topk", "mode": mode, "nq": a.nq, "N": a.N, "D": a.D, "k": k, "ms": ms,
                   "queries_per_s": a.nq / ms * 1e3, "f32_equivalent_TFLOPs": flops / ms / 1e9,
                   "executed_bf16_TFLOPs": planes * flops / ms / 1e9,
                   "frac_of_bf16_mfma_peak": planes * flops / ms / 1e9 / 2500.0}
            if exact_i is not None and mode != "exact":
                rec["recall_at_k_vs_exact"] = recall_at_k(out[1], exact_i)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
