"""`bench.py --workload retrieve`: BASELINE config 5 -- D = 512 batch score GEMM + top-k, brute force (exact) and
the bf16 candidate stage + exact re-rank ("ann"), candidates row-sharded over the GPUs (id mod N).
A step = one batch of 8192 queries per GPU against ALL 1 048 576 candidates (each GPU scores every rank's queries
against its own shard, then the per-shard answers are exchanged and merged: esrecsys_amd/sharded.py)."""
import os
import time

import torch
import torch.distributed as dist

NQ, N_TOTAL, D, K = 8192, 1_048_576, 512, 500
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3


def _planes(mode):
    """MFMA cross terms per product and whether the planes are fp16, for a retrieve_topk mode name."""
    m = os.environ.get("ESR_RETRIEVE_EXACT", "bf16x3") if mode in ("exact", "f32") else mode
    # (f16r: one fp16 term per candidate as a FILTER, survivors re-scored in f32 -- exact top-k of the f32 scores)
    return {"f16x2": (3, True), "bf16x3": (6, False), "bf16": (1, False), "f16r": (1, True)}[m] + (m,)


def measure_retrieve(dev, n_local, steps, warmup, mode="f16x2", with_cpu=True, with_ann=True, nq=NQ, k=K, prepared=False):
    """Single-GPU leg: `nq` queries against `n_local` candidates of dimension D, top-k, brute force.  Returns the
    fields of a bench line (value = queries/s, roofline of the score GEMM, cpu_baseline from the oracle)."""
    import numpy as np
    from esrecsys_amd import ops
    from esrecsys_amd.pinterest.make_recommendations import find_top_k_batch, recall_at_k
    g = torch.Generator(device=dev).manual_seed(1701)
    q = torch.randn((nq, D), generator=g, device=dev) * D ** -0.5
    c = torch.randn((n_local, D), generator=g, device=dev) * D ** -0.5
    # prepared (mode f16r): the corpus half of the call -- statistics pass + fp16 plane -- made ONCE, outside the timed
    # region, as a serving loop over a fixed product table would (esr_retrieve_prepare; the time it takes is reported)
    prep, prep_ms = None, None
    if prepared:
        torch.cuda.synchronize()
        tp = time.perf_counter()
        prep = ops.retrieve_prepare(c, mode=mode)
        torch.cuda.synchronize()
        prep_ms = (time.perf_counter() - tp) * 1e3
    for _ in range(max(warmup, 1)):
        out = ops.retrieve_topk(q, c, k, mode=mode, prepared=prep)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = ops.retrieve_topk(q, c, k, mode=mode, prepared=prep)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_op = e0.elapsed_time(e1) * 1e-3 / steps  # HIP events on the launch stream around the timed calls
    planes, f16_planes, exact_path = _planes(mode)  # MFMA cross terms per product
    flops = 2.0 * nq * n_local * D
    from bench import sustained_bf16_mfma_tflops
    live = sustained_bf16_mfma_tflops(dev, f16=f16_planes)
    extra = {}
    if with_ann:
        a_s, a_i = find_top_k_batch(q, c, k, approximate=True)
        extra["ann_recall_at_k_vs_brute_force"] = recall_at_k(a_i, out[1])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        find_top_k_batch(q, c, k, approximate=True)
        torch.cuda.synchronize()
        extra["ann_ms"] = (time.perf_counter() - t1) * 1e3
    cpu = None
    if with_cpu:
        from oracle import topk as o_topk          # CPU baseline leg: the oracle on a bounded sample
        ns = 64 if n_local <= 262144 else 16
        qs, cs = q[:ns].cpu().numpy(), c.cpu().numpy()
        t1 = time.perf_counter()
        es, ei = o_topk.batched_top_k(qs, cs, k, np.float32)
        cpu_dt = time.perf_counter() - t1
        cpu = {"value": ns / cpu_dt, "unit": "queries/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": "%d queries x %d candidates x D=%d, numpy f32 GEMM + stable argsort" % (ns, n_local, D)}
        # the sample doubles as a parity spot-check of the timed answer (fp32 GEMM order differs: compare as sets)
        got = out[1][:ns].cpu().numpy()
        extra["agrees_with_cpu_sample_at_k"] = float(np.mean([len(set(a) & set(b)) / float(k) for a, b in zip(got, ei)]))
    from bench import pmc_traffic
    traffic = pmc_traffic("retrieve|N=%d|nq=%d|%s" % (n_local, nq, exact_path)) if k == K else None
    del q, c
    torch.cuda.empty_cache()
    return {
        "metric": "retrieval queries/sec (top-%d of N candidates, brute force)" % k,
        "value": nq * steps / dt, "unit": "queries/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "dtype": exact_path + (" (f32-grade: two fp16 planes of x 2^e, one exponent per matrix)" if exact_path == "f16x2" else
                               " (exact split: three bf16 planes)" if exact_path == "bf16x3" else
                               " (one fp16 plane as a filter with a proven error band, survivors re-scored as f32 dot "
                               "products: the exact top-k of the f32 scores)" if exact_path == "f16r" else ""),
        "config": {"workload": "retrieve: %d queries x %d candidates x D=%d, k=%d, one GPU" % (nq, n_local, D, k),
                   "mode": mode, **({"corpus": "prepared once (esr_retrieve_prepare: %.2f ms incl. its 1 GB buffer's first "
                                                       "touch), outside the timed calls" % prep_ms} if prepared else {}),
                   **extra},
        "roofline": {"kernel": "score_gemm_kernel (+ split, select)", "bound": "mfma",
                     "achieved": planes * flops / t_op / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": planes * flops / t_op / 1e12 / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                     "algorithmic_bytes": (nq + n_local) * D * 4,
                     "sustained_live_data_TFLOPs": live, "frac_of_sustained": planes * flops / t_op / 1e12 / live,
                     "f32_equivalent_TFLOPs": flops / t_op / 1e12,
                     "f32_equivalent_vs_f32_mfma_peak": flops / t_op / 1e12 / MFMA_F32_PEAK_TFLOPS},
        "cpu_baseline": cpu,
    }


def measure_ivf(dev, n_local=1_048_576, nq=NQ, ks=(10, 500), nlist=1024, nprobes=(8, 32, 128), steps=3, corpus="clustered"):
    """Config 5's ANN leg: the IVF index (esrecsys_amd/ivf.py) against the exact brute force on the SAME corpus and
    queries -- recall@k and queries/s per (k, nprobe).  corpus "clustered": 4096 unit-norm centres + N(0, 0.6^2 / D) noise
    (embedding tables have cluster structure; that is what an inverted file exploits); "iid": the N(0, 1/D) rows of the
    brute-force leg (no structure: the worst case, recall then follows the fraction of lists probed)."""
    from esrecsys_amd import ops
    from esrecsys_amd.ivf import IVFIndex
    from esrecsys_amd.pinterest.make_recommendations import recall_at_k
    g = torch.Generator(device=dev).manual_seed(1701)
    if corpus in ("clustered", "hierarchical"):
        if corpus == "clustered":
            centres = torch.randn((4096, D), generator=g, device=dev)
        else:
            # two scales: 64 topics, 4096 sub-centres scattered around them (unit topic + 0.5-norm offset), then the rows.
            # In the flat corpus a row's cluster has N / 4096 = 256 members: of a query's true top-500 the other ~244 are
            # the far tail of a million unrelated rows, spread evenly over ALL lists -- no inverted file can find them
            # without scanning (recall ceiling at a fraction f of the candidates scored ~ (256 + 244 f) / 500).  Here
            # the neighbours beyond the own sub-cluster are the sibling sub-clusters of the topic: what the lists near
            # a query hold.
            topics = torch.randn((64, D), generator=g, device=dev)
            topics /= topics.norm(dim=1, keepdim=True)
            centres = topics[torch.randint(0, 64, (4096,), generator=g, device=dev)] + \
                torch.randn((4096, D), generator=g, device=dev) * (0.5 * D ** -0.5)
        centres /= centres.norm(dim=1, keepdim=True)
        c = centres[torch.randint(0, 4096, (n_local,), generator=g, device=dev)] + \
            torch.randn((n_local, D), generator=g, device=dev) * (0.6 * D ** -0.5)
        q = centres[torch.randint(0, 4096, (nq,), generator=g, device=dev)] + \
            torch.randn((nq, D), generator=g, device=dev) * (0.6 * D ** -0.5)
    else:
        c = torch.randn((n_local, D), generator=g, device=dev) * D ** -0.5
        q = torch.randn((nq, D), generator=g, device=dev) * D ** -0.5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = IVFIndex(c, nlist)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    out = {"corpus": corpus, "N": n_local, "D": D, "queries": nq, "nlist": nlist, "build_s": build_s,
           "longest_list": index.max_list, "legs": []}
    for k in ks:
        # the yardstick is the library's exact brute force (three bf16 planes); the f32-grade fp16 x 2 one is timed too
        brute_by_mode = {}
        for mode in ("exact", "f16x2"):
            ops.retrieve_topk(q, c, k, mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                _, ans = ops.retrieve_topk(q, c, k, mode=mode)
            torch.cuda.synchronize()
            brute_by_mode[mode] = (time.perf_counter() - t0) / steps
            if mode == "exact":
                exact = ans
        brute = brute_by_mode["exact"]
        for nprobe in nprobes:
            index.search(q, k, nprobe)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                _, got = index.search(q, k, nprobe)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out["legs"].append({"k": k, "nprobe": nprobe, "ms": dt * 1e3, "queries_per_s": nq / dt,
                                "recall_at_k_vs_exact": recall_at_k(got, exact), "brute_force_ms": brute * 1e3,
                                "brute_force_f16x2_ms": brute_by_mode["f16x2"] * 1e3,
                                "speedup_vs_brute_force": brute / dt,
                                "fraction_of_candidates_scored": nprobe / nlist})
    del index, c, q
    torch.cuda.empty_cache()
    return out


def run_retrieve(args, emit):
    from esrecsys_amd import ops, sharded
    from esrecsys_amd.pinterest.make_recommendations import find_top_k_batch, recall_at_k
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (ESR_WIRE_ONE_GPU=1 + ESR_RCCL_LIB: the dry run of bench.py -- every rank on cuda:0, gloo, the loopback wire)
    one_gpu_wire = world > 1 and os.environ.get("ESR_WIRE_ONE_GPU") == "1" and bool(os.environ.get("ESR_RCCL_LIB"))
    dev = torch.device("cuda", 0 if one_gpu_wire else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_wire:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # weak scaling would grow the candidate set with N; config 5 fixes it at 1M rows, so at N = 1 one GPU holds
    # the share it would hold in the 8-GPU job (131 072 rows) and N GPUs hold N such shares
    n_local = int(args.rows) if getattr(args, "rows", None) else N_TOTAL // 8   # (--rows 1048576: the whole config on one GPU)
    g = torch.Generator(device=dev).manual_seed(1701 + rank)
    q = torch.randn((NQ, D), generator=g, device=dev) * D ** -0.5
    c = torch.randn((n_local, D), generator=g, device=dev) * D ** -0.5
    # auto: the f32-grade fp16 x 2 planes (asked for by name); f32: the exact split (three bf16 planes); else one plane
    mode = {"auto": "f16x2", "f16x2": "f16x2", "f32": "exact", "bf16x3": "bf16x3"}.get(args.precision, "bf16")
    mode = os.environ.get("ESR_BENCH_RETRIEVE_MODE", mode)   # (e.g. f16r: the one-term filter + f32 re-score)

    def step():
        if world == 1:
            return ops.retrieve_topk(q, c, K, mode=mode)
        return sharded.sharded_find_top_k(q, c, K, mode=mode)

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    # dominant kernel = the score GEMM; HIP events around the op on the launch stream (rank 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    qq = q if world == 1 else q.repeat(world, 1)
    e0.record()
    ops.retrieve_topk(qq, c, K, mode=mode)
    e1.record()
    torch.cuda.synchronize()
    t_op = e0.elapsed_time(e1) * 1e-3
    planes, f16_planes, exact_path = _planes(mode)  # MFMA cross terms per product
    flops = 2.0 * qq.shape[0] * n_local * D
    from bench import pmc_traffic
    traffic = pmc_traffic("retrieve|N=%d|nq=%d|%s" % (n_local, NQ, exact_path)) if world == 1 else None
    extra = {}
    if world == 1 and not args.no_cpu_baseline:
        a_s, a_i = find_top_k_batch(q, c, K, approximate=True)
        extra["ann_recall_at_k_vs_brute_force"] = recall_at_k(a_i, out[1])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        find_top_k_batch(q, c, K, approximate=True)
        torch.cuda.synchronize()
        extra["ann_ms"] = (time.perf_counter() - t1) * 1e3
        from oracle import topk as o_topk          # CPU baseline leg: the oracle on a bounded sample
        import numpy as np
        ns = 64
        qs, cs = q[:ns].cpu().numpy(), c.cpu().numpy()
        t1 = time.perf_counter()
        o_topk.batched_top_k(qs, cs, K, np.float32)
        cpu_dt = time.perf_counter() - t1
        extra_cpu = {"value": ns / cpu_dt, "unit": "queries/s", "cores": torch.get_num_threads(), "kind": "port",
                     "sample": "%d queries x %d candidates x D=%d, numpy f32 GEMM + stable argsort" % (ns, n_local, D)}
    else:
        extra_cpu = None
    if rank == 0:
        from bench import sustained_bf16_mfma_tflops
        live = sustained_bf16_mfma_tflops(dev, f16=f16_planes)
        emit({
            "metric": "retrieval queries/sec (top-%d of N candidates, brute force)" % K,
            "value": world * NQ * args.steps / dt, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": exact_path + (" (f32-grade: two fp16 planes of x 2^e, one exponent per matrix)" if exact_path == "f16x2"
                                   else " (exact split: three bf16 planes)" if exact_path == "bf16x3" else ""),
            "data": "synthetic",
            "config": {"workload": "retrieve: %d queries/GPU x %d candidates (%d per GPU, id mod N) x D=%d, k=%d"
                                   % (NQ, n_local * world, n_local, D, K), "mode": mode,
                       "parallelism": "single" if world == 1 else "candidates row-sharded x%d, all-gather queries + "
                                                                  "all-to-all partial top-k" % world,
                       **({"exchange": "DRY RUN over the loopback wire (every rank on ONE GPU): not a scaling measurement"}
                          if one_gpu_wire else {}), **extra},
            "roofline": {"kernel": "score_gemm_kernel (+ split, select)", "bound": "mfma",
                         "achieved": planes * flops / t_op / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": planes * flops / t_op / 1e12 / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                         "sustained_live_data_TFLOPs": live, "frac_of_sustained": planes * flops / t_op / 1e12 / live,
                         "f32_equivalent_TFLOPs": flops / t_op / 1e12,
                         "f32_equivalent_vs_f32_mfma_peak": flops / t_op / 1e12 / MFMA_F32_PEAK_TFLOPS},
            "cpu_baseline": extra_cpu,
        })
    if world > 1:
        dist.destroy_process_group()
