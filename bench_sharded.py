"""N > 1 leg of bench.py: one process per GPU, rows sharded owner = id mod N, RCCL all-to-all over xGMI
for the id / row / gradient exchange (esrecsys_amd/sharded.py).  Weak scaling: every rank draws its own B
pairs per step; value = N * B * K / max-over-ranks time."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

LAM, SCALE, LR, SEED = 0.1, 8.0, 0.05, 1701


class Watchdog:
    """A phase of the N > 1 run that must not hang a GPU node silently (process-group rendezvous, the RCCL bootstrap and its
    self-test): if `arm(what)` is not followed by `disarm()` within the limit, ONE diagnostic JSON line goes to stdout
    and the process exits with status 3 -- the driver then has a record instead of a timeout."""

    def __init__(self, rank, world):
        self.rank, self.world, self.timer = rank, world, None
        self.limit = float(os.environ.get("ESR_BENCH_PREFLIGHT_TIMEOUT", "240"))

    def arm(self, what):
        import threading
        self.disarm()

        def fire():
            print(json.dumps({"preflight": {"ok": False, "rank": self.rank, "world_size": self.world, "hung_in": what,
                                            "after_s": self.limit,
                                            "hint": "a peer never reached this phase, or RCCL could not open its links "
                                                    "(HSA_ENABLE_IPC_MODE_LEGACY=0? MASTER_ADDR=127.0.0.1?)"}}), flush=True)
            os._exit(3)
        self.timer = threading.Timer(self.limit, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def preflight(dev, rank, world, watchdog):
    """Before anything is timed: the exchange communicator is built (rccl.py: bind, ncclCommInitRank, a bounded self-test
    all-to-all with known contents, every phase agreed over the process group) under the watchdog, and rank 0 prints one
    JSON line saying what the run will exchange through.  Returns the record."""
    from esrecsys_amd import rccl
    t0 = time.perf_counter()
    watchdog.arm("RCCL bootstrap + self-test (esrecsys_amd/rccl.py)")
    x = rccl.exchange_for(None, dev)
    x2 = rccl.exchange_for(None, dev, lane=1) if x is not None and os.environ.get("ESR_SHARDED_OVERLAP", "0") == "1" else None
    watchdog.disarm()
    wire = os.environ.get("ESR_RCCL_LIB")
    kind = "torch.distributed" if x is None else ("loopback" if wire else "rccl")
    rec = {"ok": True, "world_size": world, "backend": dist.get_backend(), "exchange": kind,
           "rccl_ranks": x.ranks_seen()[0] if x is not None else None,
           "second_communicator": x2 is not None, "bootstrap_and_selftest_s": round(time.perf_counter() - t0, 3),
           "library": os.path.basename(wire) if wire else "torch's librccl"}
    if x is not None and rec["rccl_ranks"] != world:
        rec["ok"] = False
        rec["error"] = "the communicator reports %s ranks, the launch has %d" % (rec["rccl_ranks"], world)
    if not rec["ok"]:  # the one line of a run that will not produce a bench line
        if rank == 0:
            print(json.dumps({"preflight": rec}), flush=True)
        raise SystemExit(3)
    if rank == 0:  # (a run that goes on prints its ONE JSON line at the end, with this record under config.preflight)
        print(json.dumps({"preflight": rec}), file=sys.stderr, flush=True)
    return rec


def run_sharded(args, cfg, dev, rank, world, watchdog=None):
    from esrecsys_amd import ops, sharded
    watchdog = watchdog or Watchdog(rank, world)
    pre = preflight(dev, rank, world, watchdog) if world > 1 else None
    V, D, B = cfg["V"], cfg["D"], cfg["B"]
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + 17 * rank)

    def shard(num_rows, dim):
        n = sharded.RowShardedTable.local_rows_for(num_rows, world, rank)
        t = torch.randn((n, dim), generator=gen, device=dev, dtype=torch.float32).mul_(dim ** -0.5)
        if cfg.get("table_dtype") == "bf16" and dim > 1:   # the GloVe bias column stays fp32
            t = t.to(torch.bfloat16)
        return sharded.RowShardedTable(t, torch.full((n, dim), 0.1, device=dev), num_rows)

    # ESR_BENCH_PARALLELISM=replicated: every rank holds the FULL tables, gathers all ranks' ids + gradient rows and
    # applies one global sparse update (esrecsys_amd/replicated.py) -- SURVEY 8e's other mode, for tables that fit a GPU
    replicated_mode = os.environ.get("ESR_BENCH_PARALLELISM", "sharded") == "replicated"
    rep = rep_bias = None
    if replicated_mode:
        from esrecsys_amd import replicated
        g_all = torch.Generator(device=dev).manual_seed(SEED)  # the SAME tables on every rank
        if args.workload == "glove":
            full = [torch.randn((V, D), generator=g_all, device=dev).mul_(D ** -0.5)]
            rep = replicated.ReplicatedTables(full, [torch.full((V, D), 0.1, device=dev)], kernels=ops)
            rep_bias = replicated.ReplicatedTables([torch.zeros((V, 1), device=dev)], [torch.full((V, 1), 0.1, device=dev)],
                                                   kernels=ops)
        else:
            full = [torch.randn((V, D), generator=g_all, device=dev).mul_(D ** -0.5) for _ in range(2)]
            rep = replicated.ReplicatedTables(full, [torch.full((V, D), 0.1, device=dev) for _ in range(2)], kernels=ops)
    elif args.workload == "glove":
        emb_t, bias_t = shard(V, D), shard(V, 1)
        bias_t.local.zero_()
        emb = sharded.ShardedTableGroup([emb_t], kernels=ops)
        bias = sharded.ShardedTableGroup([bias_t], kernels=ops)
    else:
        # bf16 towers (BASELINE config 4): the gradient rows cross the exchange as bf16 too -- SURVEY 8d's 910 B/pair budget;
        # a bf16 row keeps 8 significant bits of its update anyway (ESR_SHARDED_GRAD_DTYPE=f32 to send them wide)
        gd = os.environ.get("ESR_SHARDED_GRAD_DTYPE") or ("bf16" if cfg.get("table_dtype") == "bf16" else "f32")
        towers = sharded.ShardedTableGroup([shard(V, D), shard(V, D)], kernels=ops, grad_dtype=gd)  # scene, product
    grp0 = None if replicated_mode else (emb if args.workload == "glove" else towers)
    # no routing plans where nothing is routed: the replicated mode, and a world of one rank taking the single-GPU steps
    no_plans = replicated_mode or grp0.world1_direct
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

    def begin(b):  # the device half of the routing plan of one batch: no host wait
        if args.workload == "glove":
            return sharded.begin_plan_glove(emb, b[0])
        if args.workload == "inbatch":
            return sharded.begin_plan_inbatch(towers, b[0], b[1])
        return sharded.begin_plan_triplet(towers, b[0], b[1], b[2])

    def lookup(b):  # (group, virtual id segments) of one batch, for begin_plans
        if args.workload == "glove":
            return (emb, emb.virtual_id_segments([b[0].reshape(-1)], [0]))
        if args.workload == "inbatch":
            return (towers, towers.virtual_id_segments([b[0], b[1]], [0, 1]))
        return (towers, towers.virtual_id_segments([b[0], b[1], b[2]], [0, 1, 1]))

    # ESR_SHARDED_PLAN_GROUP=K (> 1): the routing plans of K coming batches are made together -- K bucket kernels, ONE
    # counts all-to-all, ONE copy to pinned memory and ONE host wait per K steps instead of one each per step -- and the
    # next group's are enqueued in front of this group's steps, K steps before the host needs them.
    # (1: a plan per step, pipelined two batches deep -- the loop of round 1.  World 1, B = 8192, with the group's bucket
    # kernels and owner-side sorts batched as well: in-batch step 27.6 -> 30.4 M pairs/s, triplet 52.2 -> 82-88 M
    # triplets/s, GloVe 110.9 -> 114 M pairs/s; the counts exchange it saves costs more across ranks than as the
    # self-copy it is there.)
    plan_group = max(1, int(os.environ.get("ESR_SHARDED_PLAN_GROUP", "8")))

    def step(b, plans):
        if replicated_mode:
            if args.workload == "glove":
                return replicated.replicated_glove_step(rep, rep_bias, b[0], b[1], ops.GLOVE_REFERENCE, LR)
            if args.workload == "inbatch":
                return replicated.replicated_inbatch_step(rep, b[0], b[1], LAM, gb, SCALE, LR)
            return replicated.replicated_triplet_step(rep, b[0], b[1], b[2], LAM, gb, LR)
        if args.workload == "glove":
            return sharded.sharded_glove_step(emb, bias, b[0], b[1], ops.GLOVE_REFERENCE, LR, plan=plans)
        if args.workload == "inbatch":
            return sharded.sharded_inbatch_step(towers, b[0], b[1], LAM, gb, SCALE, LR, plan=plans)
        return sharded.sharded_triplet_step(towers, b[0], b[1], b[2], LAM, gb, LR, plan=plans)

    # Routing plans are pipelined two batches deep.  begin(k+2) -- bucket kernel, counts all-to-all, asynchronous copy
    # of the counts to pinned memory -- is enqueued right after step k; the host half of plan k+1 (wait for ITS copy,
    # issued one iteration earlier; then the ids all-to-all and the owner-side sort, whose split sizes the host must
    # know) runs while the GPU still has step k queued.  The host therefore never blocks an idle GPU, and everything
    # stays inside the timed region.
    def run(lo, hi, timed_loss=None):
        from esrecsys_amd.train_state import quiet_gc
        if no_plans:
            with quiet_gc():
                loss = None
                for i in range(lo, hi):
                    loss = step(batches[i], None)
                return loss
        if plan_group > 1:  # the library's loop helper (plans of plan_group coming batches made together)
            grps = (emb, bias) if args.workload == "glove" else (towers,)
            out = sharded.sharded_train_steps(args.workload, grps, batches[lo:hi], regularization=LAM,
                                              global_batch_size=gb, scale=SCALE, lr=LR, mode=ops.GLOVE_REFERENCE,
                                              plan_group=plan_group)
            return out[-1] if out else None
        with quiet_gc():  # as the loop helpers: a full cyclic collection inside the loop is a 40 ms hole in the launches
            cur = begin(batches[lo]).finish()
            pend = begin(batches[lo + 1]) if lo + 1 < hi else None
            loss = None
            for i in range(lo, hi):
                loss = step(batches[i], cur)
                nxt_pend = begin(batches[i + 2]) if i + 2 < hi else None
                cur = pend.finish() if pend is not None else None
                pend = nxt_pend
            return loss

    loss = run(0, args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = run(args.warmup, n_batches)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    if sharded._trace is not None and rank == 0:
        print("sharded loop: %.1f us per step wall, %.1f us of it waiting for the routing-plan copy (%d waits)" % (
            (time.perf_counter() - t0) / args.steps * 1e6, sharded._trace[0] / max(1, sharded._trace[1]) * 1e6,
            sharded._trace[1]), file=sys.stderr)
    # (host tensors under the one-GPU wire dry run, whose process group is gloo: bench.py, ESR_WIRE_ONE_GPU)
    red_dev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    dt = torch.tensor([time.perf_counter() - t0], device=red_dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    total = loss.detach().to(red_dev).clone()
    dist.all_reduce(total)
    dt = float(dt)

    # per-kernel HIP-event pass (every rank runs it: the collectives must match; rank 0 reports its own kernels)
    roofline, kernels = None, {}
    if not args.no_kernel_timing:
        from bench import kernel_timer, roofline_for
        timer = kernel_timer()
        timer.enabled = True
        run(args.warmup, n_batches)
        torch.cuda.synchronize()
        timer.enabled = False
        for g_, (ms, calls) in timer.totals_ms(args.steps).items():
            if calls:
                kernels[g_] = {"ms_per_step": ms / args.steps, "launch_groups_per_step": calls / args.steps}
        if no_plans:
            last = batches[-1]
            rows_served = torch.cat([last[0].reshape(-1)] if args.workload == "glove" else
                                    [last[0], last[1] + V] + ([last[2] + V] if args.workload == "triplet" else []))
        else:
            rows_served = begin(batches[-1]).finish().recv_local_rows
        roofline = roofline_for(args.workload, kernels, B, D, cfg["rows_per_unit"], "auto",
                                int(rows_served.numel()), int(torch.unique(rows_served).numel()),
                                bf16_tables=cfg.get("table_dtype") == "bf16", step_s=dt / args.steps)
        roofline["rank"] = 0
    xch = grp0.exchange() if grp0 is not None else rep.coll.x
    exchange = "direct RCCL: grouped ncclSend/ncclRecv on the compute stream (esr_alltoall_*, esrecsys_amd/rccl.py)" \
        if xch is not None else "fallback: torch.distributed all_to_all_single"
    if os.environ.get("ESR_RCCL_LIB"):
        exchange = ("DRY RUN over %s (every rank on ONE GPU, bytes over sockets): the library's grouped send / recv code "
                    "path, NOT RCCL / xGMI -- the value is not a scaling measurement" % os.path.basename(os.environ["ESR_RCCL_LIB"]))
    rccl_ranks = xch.ranks_seen()[0] if xch is not None else None  # ncclCommCount of the exchange communicator
    exchange_kind = "torch.distributed" if xch is None else ("loopback" if os.environ.get("ESR_RCCL_LIB") else "rccl")
    # what this rank puts on the wire per unit of work, from the LAST batch's routing plan (rows addressed to other ranks)
    wire = None
    if not no_plans:
        p_last = begin(batches[-1]).finish()
        out_rows = sum(c for peer, c in enumerate(p_last.send_counts) if peer != rank)
        s_b = 2 if cfg.get("table_dtype") == "bf16" else 4
        g_name = getattr(grp0, "grad_dtype", "f32") or "f32"
        g_b = 2 if g_name == "bf16" else 4
        wire = {"rows_to_other_ranks_per_step": out_rows, "of_rows_per_step": int(p_last.n_rows),
                "ids_int32": round(4.0 * out_rows / B, 1),
                "rows_%s" % ("bf16" if s_b == 2 else "f32"): round(float(s_b * D * out_rows) / B, 1),
                "grads_%s" % g_name: round(float(g_b * D * out_rows) / B, 1)}
        wire["total_per_unit_each_way"] = round(wire["ids_int32"] + (s_b + g_b) * D * out_rows / B, 1)
        wire["note"] = "bytes this rank sends (ids, gradient rows) or receives (rows) per %s; SURVEY 8d budgets 910 for " \
                       "config 4 (bf16 rows + bf16 gradients, G = 8)" % cfg["unit"]
    if rank == 0:
        K = args.steps
        from bench import emit, sustained_bf16_mfma_tflops, MFMA_BF16_PEAK_TFLOPS
        if roofline is not None and roofline.get("bound") == "mfma" and roofline.get("peak") == MFMA_BF16_PEAK_TFLOPS:
            live = sustained_bf16_mfma_tflops(dev, f16=str(roofline.get("dtype", "")).startswith("fp16"))
            roofline["sustained_live_data_TFLOPs"] = live
            roofline["frac_of_sustained"] = roofline["achieved"] / live
        cpu = None
        if not args.no_cpu_baseline:  # the same host-side restatement the N = 1 line carries (rank 0's host cores)
            from bench import cpu_baseline
            cpu = cpu_baseline(args.workload, cfg, budget_s=6.0, dense_budget_s=4.0)
        if replicated_mode:
            par = "replicated x%d: full tables per rank, all-gather of ids + gradient rows, one global sparse update" % world
        elif grp0.world1_direct:
            par = "row-sharded x1: a world of one rank takes the single-GPU steps on its shard (nothing is exchanged)"
        else:
            par = "row-sharded x%d, all-to-all ids/rows/grads over RCCL%s" % (
                world, {"on": ", every distinct row once (unique plans)", "off": ", one row per occurrence",
                        "auto": ", unique plans while this rank's measured distinct / occurrences < %.2f (now: %s)"
                                % (sharded._UNIQUE_KEEP_BELOW, "unique" if grp0.unique else "per occurrence")}[grp0.unique_mode])
        emit({
            "metric": "training pairs/sec", "value": world * B * K / dt, "unit": cfg["unit"] + "s/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: V=%d x D=%d %s tables row-sharded id mod %d, B=%d per GPU, sparse Adagrad"
                                   % (args.workload, V, D, "bf16" if cfg.get("table_dtype") == "bf16" else "fp32",
                                      world, B),
                       "parallelism": par,
                       "exchange": exchange, "exchange_kind": exchange_kind, "rccl_ranks": rccl_ranks, "world_size": world,
                       "wire_bytes_per_unit": wire, "preflight": pre,
                       "gradient_rows_on_the_wire": getattr(grp0, "grad_dtype", "f32") if grp0 is not None else "f32",
                       "routing_plans": ("made for %d coming batches at a time (one bucket launch pair, one counts "
                                         "all-to-all, one RCCL group of ids exchanges, one owner-side sort per group)"
                                         % plan_group) if plan_group > 1 else "one per step, pipelined two batches deep",
                       "overlap": ("next batch's gather + rows exchange on a side stream / second communicator under this "
                                   "batch's loss kernel and update; stale rows served again (ESR_SHARDED_OVERLAP=1)")
                       if os.environ.get("ESR_SHARDED_OVERLAP", "0") == "1" and plan_group > 1 and not replicated_mode
                       and not grp0.world1_direct else "off",
                       "loss": float(total)},
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        })
    dist.barrier()  # rank 0 may still be measuring its MFMA ceiling: leave the group together
    dist.destroy_process_group()
