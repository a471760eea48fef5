import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_glove_step import _make_state, _ids
from esrecsys_amd.wikipedia.train_cooccurence import apply_model, train_step, update_model
dev = torch.device("cuda", 0)
for (V, D, B, kind) in [(5000, 256, 4096, "uniform"), (300, 64, 1000, "uniform"), (1000, 6, 64, "uniform"), (300, 64, 1000, "same")]:
    rng = np.random.default_rng(1)
    a, b = _make_state(V, D, "reference", dev), _make_state(V, D, "reference", dev)
    inputs = _ids(kind, V, (2, B), rng)
    target = np.exp(rng.uniform(np.log(0.1), np.log(1000.0), B)).astype(np.float32)
    a, la = train_step(a, inputs, target)
    grads, lb = apply_model(b, inputs, target)
    b = update_model(b, grads)
    ea, eb = a.params["_token_embedding"]["embedding"], b.params["_token_embedding"]["embedding"]
    d = (ea - eb).abs()
    print(V, D, B, kind, "loss", float(la), float(lb), "max diff", float(d.max()), "nonzero rows", int((d.max(1).values > 0).sum()),
          "of touched", len(np.unique(inputs)), "rel", float(d.max() / eb.abs().max()))
    aa, ab = a.opt_state["sum_of_squares"]["_token_embedding"]["embedding"], b.opt_state["sum_of_squares"]["_token_embedding"]["embedding"]
    print("   accum max diff", float((aa - ab).abs().max()), "bias diff", float((a.params["_bias"]["embedding"] - b.params["_bias"]["embedding"]).abs().max()))
