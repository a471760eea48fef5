"""N > 1 leg of bench.py: one process per GPU, rows sharded owner = id mod N, RCCL all-to-all over xGMI
for the id / row / gradient exchange (esrecsys_amd/sharded.py).  Weak scaling: every rank draws its own B
pairs per step; value = N * B * K / max-over-ranks time."""
import json
import time

import numpy as np
import torch
import torch.distributed as dist

LAM, SCALE, LR, SEED = 0.1, 8.0, 0.05, 1701


def run_sharded(args, cfg, dev, rank, world):
    from esrecsys_amd import ops, sharded
    V, D, B = cfg["V"], cfg["D"], cfg["B"]
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + 17 * rank)

    def shard(num_rows, dim):
        n = sharded.RowShardedTable.local_rows_for(num_rows, world, rank)
        t = torch.randn((n, dim), generator=gen, device=dev, dtype=torch.float32).mul_(dim ** -0.5)
        return sharded.RowShardedTable(t, torch.full((n, dim), 0.1, device=dev), num_rows)

    if args.workload == "glove":
        emb_t, bias_t = shard(V, D), shard(V, 1)
        bias_t.local.zero_()
        emb = sharded.ShardedTableGroup([emb_t], kernels=ops)
        bias = sharded.ShardedTableGroup([bias_t], kernels=ops)
    else:
        towers = sharded.ShardedTableGroup([shard(V, D), shard(V, D)], kernels=ops)  # scene, product
    n_batches = args.steps + args.warmup
    batches = []
    for _ in range(n_batches):
        if args.workload == "glove":
            inputs = torch.randint(0, V, (2, B), generator=gen, device=dev, dtype=torch.int32)
            target = torch.exp(np.log(0.1) + torch.rand(B, generator=gen, device=dev) * np.log(1e4))
            batches.append((inputs, target))
        else:
            ids = torch.randint(0, V, (3, B), generator=gen, device=dev, dtype=torch.int32)
            batches.append((ids[0].contiguous(), ids[1].contiguous(), ids[2].contiguous()))
    gb = float(world * B)

    def plan(b):
        if args.workload == "glove":
            return sharded.plan_glove(emb, b[0])
        if args.workload == "inbatch":
            return sharded.plan_inbatch(towers, b[0], b[1])
        return sharded.plan_triplet(towers, b[0], b[1], b[2])

    def step(b, plans):
        if args.workload == "glove":
            return sharded.sharded_glove_step(emb, bias, b[0], b[1], ops.GLOVE_REFERENCE, LR, plan=plans)
        if args.workload == "inbatch":
            return sharded.sharded_inbatch_step(towers, b[0], b[1], LAM, gb, SCALE, LR, plan=plans)
        return sharded.sharded_triplet_step(towers, b[0], b[1], b[2], LAM, gb, LR, plan=plans)

    # The routing plan of batch k+1 (bucket + counts all-to-all + the one host read-back + ids all-to-all)
    # only needs its ids: it is built right after step k has been enqueued, inside the timed region, so the
    # read-back waits behind step k's kernels instead of idling the GPU in the middle of a step.
    nxt = plan(batches[0])
    for i in range(args.warmup):
        cur, nxt = nxt, None
        loss = step(batches[i], cur)
        nxt = plan(batches[i + 1])
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_batches):
        cur, nxt = nxt, None
        loss = step(batches[i], cur)
        if i + 1 < n_batches:
            nxt = plan(batches[i + 1])
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    total = loss.clone()
    dist.all_reduce(total)
    dt = float(dt)

    # per-kernel HIP-event pass (every rank runs it: the collectives must match; rank 0 reports its own kernels)
    roofline, kernels = None, {}
    if not args.no_kernel_timing:
        from bench import KernelTimer, TIMED_GROUPS, roofline_for
        timer = KernelTimer(ops, TIMED_GROUPS)
        timer.install()
        timer.enabled = True
        nxt = plan(batches[args.warmup])
        for i in range(args.warmup, n_batches):
            cur, nxt = nxt, None
            step(batches[i], cur)
            if i + 1 < n_batches:
                nxt = plan(batches[i + 1])
        torch.cuda.synchronize()
        timer.enabled = False
        for g_, (ms, calls) in timer.totals_ms().items():
            if calls:
                kernels[g_] = {"ms_per_step": ms / args.steps, "launch_groups_per_step": calls / args.steps}
        rows_served = cur.recv_local_rows
        roofline = roofline_for(args.workload, kernels, B, D, cfg["rows_per_unit"], "auto",
                                int(rows_served.numel()), int(torch.unique(rows_served).numel()))
        roofline["rank"] = 0
    if rank == 0:
        K = args.steps
        from bench import emit
        emit({
            "metric": "training pairs/sec", "value": world * B * K / dt, "unit": cfg["unit"] + "s/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: V=%d x D=%d fp32 tables row-sharded id mod %d, B=%d per GPU, sparse Adagrad"
                                   % (args.workload, V, D, world, B),
                       "parallelism": "row-sharded x%d, all-to-all ids/rows/grads over RCCL" % world,
                       "loss": float(total)},
            "roofline": roofline, "kernels": kernels, "cpu_baseline": None,
        })
    dist.destroy_process_group()
