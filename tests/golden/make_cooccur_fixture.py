"""Writes tests/golden/tiny.cooccur.pb.b64.bz2 + tiny_cooccur_expected.npz with the REFERENCE's own generated
protobuf class (wikipedia/nlp_pb2.py, importable in the build container with the pure-Python protobuf
backend), in the reference's line format (base64 of the serialised CooccurrenceRow, one per line, bz2).
Run from the repo root in the build container:

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_cooccur_fixture.py

The fixture is data (wire bytes + the numbers that were put in); nothing of the reference travels.
"""
import base64
import bz2
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/wikipedia")
import nlp_pb2 as nlp_pb  # noqa: E402  (the reference's generated module)

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(7)
rows = []
with bz2.open(os.path.join(OUT, "tiny.cooccur.pb.b64.bz2"), "wb") as f:
    for r in range(40):
        proto = nlp_pb.CooccurrenceRow()
        proto.index = int(rng.integers(1, 70000)) if r else 300
        n = int(rng.integers(0, 9))  # includes empty rows
        others = sorted(int(x) for x in rng.integers(0, proto.index, n)) if proto.index > 0 else []
        counts = [float(np.float32(c)) for c in rng.uniform(0.05, 250.0, len(others))]
        proto.other_index.extend(others)
        proto.count.extend(counts)
        f.write(base64.b64encode(proto.SerializeToString()) + b"\n")
        for o, c in zip(others, counts):
            rows.append((proto.index, o, c))
arr = np.array(rows, dtype=np.float64)
np.savez_compressed(os.path.join(OUT, "tiny_cooccur_expected.npz"), index=arr[:, 0].astype(np.int64),
                    other=arr[:, 1].astype(np.int64), count=arr[:, 2].astype(np.float32))
print("wrote %d (i, j, count) items" % len(rows))
