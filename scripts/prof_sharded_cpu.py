"""CPU-side profile of the row-sharded in-batch step at world 1 (RCCL): where the host time per step goes."""
import cProfile, os, pstats, sys, io
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from esrecsys_amd import ops, sharded
V, D, B = 1_000_000, 128, 8192
g = torch.Generator(device=dev).manual_seed(1)
def shard(n):
    return sharded.RowShardedTable(torch.randn((n, D), generator=g, device=dev) * D ** -0.5, torch.full((n, D), 0.1, device=dev), n)
towers = sharded.ShardedTableGroup([shard(V), shard(V)], kernels=ops)
batches = [torch.randint(0, V, (2, B), generator=g, device=dev, dtype=torch.int32) for _ in range(120)]
side = torch.cuda.Stream(device=dev)
W = sys.argv[1] if len(sys.argv) > 1 else "side"
def plan(b):
    return sharded.plan_inbatch(towers, b[0], b[1], stream=side if W == "side" else None)
def run(lo, hi):
    nxt = plan(batches[lo])
    for i in range(lo, hi):
        cur, nxt = nxt, plan(batches[i + 1])
        sharded.sharded_inbatch_step(towers, batches[i][0], batches[i][1], 0.1, float(B), 8.0, 0.05, plan=cur)
    torch.cuda.synchronize()
run(0, 10)
import time
t0 = time.perf_counter(); run(10, 110); dt = time.perf_counter() - t0
print("ms/step", dt / 100 * 1e3)
pr = cProfile.Profile(); pr.enable(); run(10, 110); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
dist.destroy_process_group()
