"""Throughput of the co-occurrence line-file reader (host side, no GPU): pairs/s of the C decoder path
(CooccurrenceGenerator.get_batch) against the item-at-a-time loop the reference's generator has
(tests/_reference_loop.py: test infrastructure, kept out of the product package).  Synthetic file: 4000 rows of 20-200 pairs, ids < 400 000, bz2 level 9.

    python benchmarks/reader_bench.py [--batch 65536] [--shuffle 0]
"""
import argparse
import base64
import bz2
import json
import os
import sys
import tempfile
import time

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
from _reference_loop import batches_item_by_item  # noqa: E402
from esrecsys_amd.wikipedia.cooccurrence_matrix import CooccurrenceGenerator  # noqa: E402


def _vint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--shuffle", type=int, default=0)
    ap.add_argument("--rows", type=int, default=4000)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    lines, npairs = [], 0
    for r in range(args.rows):
        k = int(rng.integers(20, 200))
        body = b"".join(_vint(int(o)) for o in rng.integers(0, 400_000, k))
        counts = (rng.random(k) * 100).astype(np.float32)
        lines.append(base64.b64encode(b"\x08" + _vint(r + 1) + b"\x12" + _vint(len(body)) + body + b"\x1a" +
                                      _vint(4 * k) + counts.tobytes()))
        npairs += k
    d = tempfile.mkdtemp()
    fn = os.path.join(d, "x.cooccur.pb.b64.bz2")
    with open(fn, "wb") as f:
        f.write(bz2.compress(b"\n".join(lines) + b"\n"))
    g = CooccurrenceGenerator(fn)
    for name, it, target in (("c_decoder", g.get_batch(args.batch, args.shuffle), 20 * npairs),
                             ("item_loop", batches_item_by_item(g, min(args.batch, 8192), args.shuffle), npairs)):
        next(it)
        t0, n = time.perf_counter(), 0
        while n < target:
            n += next(it)[1].shape[0]
        dt = time.perf_counter() - t0
        print(json.dumps({"reader": name, "pairs": n, "seconds": dt, "M_pairs_per_s": n / dt / 1e6,
                          "batch": args.batch, "shuffle": args.shuffle, "file_pairs": npairs,
                          "file_bytes": os.path.getsize(fn)}), flush=True)


if __name__ == "__main__":
    main()
