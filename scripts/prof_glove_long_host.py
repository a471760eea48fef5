"""Is train_epoch at C3 (B = 65 536: side-stream sorts) bound by the host?  host issue time per step against the synced
time, then a cProfile of the loop.  python scripts/prof_glove_long_host.py"""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
K = 200
cfg = dict(bench.WORKLOADS["glove"], table_dtype="f32", ids="uniform")
state, batches = bench.make_state_and_batches("glove", cfg, dev, 3 * K, 0)
from esrecsys_amd.wikipedia.train_cooccurence import train_epoch
def run(lo):
    global state
    state, l = train_epoch(state, K, iter(batches[lo:lo + K]), consolidate=False)
run(0); torch.cuda.synchronize()
import esrecsys_amd.wikipedia.train_cooccurence as tc
# train_epoch's float(mean) syncs: time the issue part through the trace hook
os.environ["ESR_TRACE_HOST"] = "1"
t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); t2 = time.perf_counter()
print("with sync us/step", (t2 - t0) / K * 1e6)
pr = cProfile.Profile(); pr.enable(); run(2 * K); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
