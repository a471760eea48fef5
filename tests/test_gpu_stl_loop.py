"""GPU: the Shop-The-Look loop helper (train_steps: one-pass triplet steps, the id lists of the coming batches sorted
together ahead of them) against step-by-step train_step on the same batches -- same towers, accumulators and losses."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _state(dev, Vs, Vp, D, seed):
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.pinterest.models import STLModel
    g = torch.Generator(device=dev).manual_seed(seed)
    params = {"params": {"scene_tower": {"embedding": torch.randn((Vs, D), generator=g, device=dev) * D ** -0.5},
                         "product_tower": {"embedding": torch.randn((Vp, D), generator=g, device=dev) * D ** -0.5}}}
    model = STLModel(output_size=D, num_scenes=Vs, num_products=Vp, device=dev)
    return TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(0.05))


@pytest.mark.parametrize("depth,sort_batch", [(0, 8), (0, 3), (0, 1), (2, 1)])
@pytest.mark.parametrize("B,steps,ids", [(256, 7, "uniform"), (2048, 11, "hot"), (64, 1, "uniform"), (8192, 9, "uniform"),
                                         (128, 150, "hot"), (512, 27, "mixed")])  # 150 steps: the stamps wrap
def test_train_steps_equals_stepwise_train_step(dev, B, steps, ids, depth, sort_batch, monkeypatch):
    import esrecsys_amd.pinterest.train_shop_the_look as stl
    from esrecsys_amd.pinterest.train_shop_the_look import train_step, train_steps
    # depth 0: sorts on the main stream -- the lists of `sort_batch` coming batches by one batched call (default 8; groups
    # of 8 + 3, 3 + 3 + ..., or every step its own); depth 2: two batches ahead on the side stream
    monkeypatch.setattr(stl, "_LOOP_DEPTH", depth)
    monkeypatch.setattr(stl, "_SORT_BATCH", sort_batch)
    Vs, Vp, D = 3000, 5000, 64
    rng = np.random.default_rng(B + steps)

    def draw(V):
        # a few very popular rows: long runs, the chunked hot-row path ("mixed": only some batches have them, so the
        # long-run launch is made for some steps of a group and skipped for others)
        if ids == "hot" or (ids == "mixed" and rng.random() < 0.5):
            return np.where(rng.random(B) < 0.4, rng.integers(0, 3, B), rng.integers(0, V, B)).astype(np.int32)
        return rng.integers(0, V, B).astype(np.int32)
    batches = [(torch.from_numpy(draw(Vs)).to(dev), torch.from_numpy(draw(Vp)).to(dev),
                torch.from_numpy(draw(Vp)).to(dev)) for _ in range(steps)]
    a, b = _state(dev, Vs, Vp, D, 3), _state(dev, Vs, Vp, D, 3)
    a, losses = train_steps(a, iter(batches), steps, 0.1, float(B))
    ref = []
    for scene, pos, neg in batches:
        b, l = train_step(b, scene, pos, neg, 0.1, float(B))
        ref.append(l)
    assert losses.shape == (steps,) and int(a.step) == int(b.step) == steps
    assert torch.equal(losses, torch.stack(ref))
    for tower in ("scene_tower", "product_tower"):
        assert torch.equal(a.params["params"][tower]["embedding"], b.params["params"][tower]["embedding"])
        assert torch.equal(a.opt_state["sum_of_squares"]["params"][tower]["embedding"],
                           b.opt_state["sum_of_squares"]["params"][tower]["embedding"])


def test_train_steps_host_batches_and_other_optimizer(dev):
    """NumPy batches (copied per step) and the fallback for an optimizer without the one-pass step"""
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.pinterest.train_shop_the_look import train_step, train_steps
    Vs, Vp, D, B, steps = 500, 700, 32, 128, 3
    rng = np.random.default_rng(5)
    batches = [(rng.integers(0, Vs, B), rng.integers(0, Vp, B), rng.integers(0, Vp, B)) for _ in range(steps)]
    a, b = _state(dev, Vs, Vp, D, 1), _state(dev, Vs, Vp, D, 1)
    a, losses = train_steps(a, iter(batches), steps, 0.1, float(B))
    for scene, pos, neg in batches:
        b, l = train_step(b, scene, pos, neg, 0.1, float(B))
    assert abs(float(losses[-1]) - float(l)) <= 1e-6 * abs(float(l))
    assert rel_err(a.params["params"]["scene_tower"]["embedding"].cpu().numpy(),
                   b.params["params"]["scene_tower"]["embedding"].cpu().numpy()) <= 1e-6
    c = _state(dev, Vs, Vp, D, 1)
    c = TrainState.create(apply_fn=c.apply_fn, params=c.params, tx=optim.sgd(0.1))
    c, losses_c = train_steps(c, iter(batches), steps, 0.1, float(B))
    assert losses_c.shape == (steps,) and bool(torch.isfinite(losses_c).all())


def test_train_steps_iterator_that_ends_early(dev):
    """The reference's loop raises StopIteration at the step whose next() finds the iterator empty; the loop helper
    prefetches groups of eight but must do the same (after running the steps it was fed), not die inside the prefetch."""
    from esrecsys_amd.pinterest.train_shop_the_look import train_steps
    Vs, Vp, D, B = 500, 700, 32, 256
    rng = np.random.default_rng(9)
    batches = [tuple(torch.from_numpy(rng.integers(0, V, B).astype(np.int32)).to(dev) for V in (Vs, Vp, Vp))
               for _ in range(5)]
    with pytest.raises(StopIteration):
        train_steps(_state(dev, Vs, Vp, D, 2), iter(batches), 9, 0.1, float(B))
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,D,steps", [(256, 128, 11), (128, 64, 3), (100, 32, 9)])
def test_train_steps_inbatch_equals_stepwise_train_step(dev, B, D, steps):
    """Batches with neg = None are in-batch-softmax steps: the loop helper runs train_step per batch with the occurrence
    lists [scene ; Vs + pos] of up to eight coming batches sorted by one batched call -- towers, accumulators and losses
    bit-identical to the stepwise loop (fp16 x 2 head at B % 128 = 0, D >= 64; exact-f32 head otherwise)."""
    from esrecsys_amd.pinterest.train_shop_the_look import train_step, train_steps
    Vs, Vp = 3000, 5000
    rng = np.random.default_rng(B + steps)
    batches = [(torch.from_numpy(rng.integers(0, Vs, B).astype(np.int32)).to(dev),
                torch.from_numpy(rng.integers(0, Vp, B).astype(np.int32)).to(dev), None) for _ in range(steps)]
    a, b = _state(dev, Vs, Vp, D, 4), _state(dev, Vs, Vp, D, 4)
    a, losses = train_steps(a, iter(batches), steps, 0.1, float(B), scale=6.0)
    ref = []
    for scene, pos, _ in batches:
        b, l = train_step(b, scene, pos, None, 0.1, float(B), scale=6.0)
        ref.append(l)
    assert losses.shape == (steps,) and int(a.step) == int(b.step) == steps
    assert torch.equal(losses, torch.stack(ref))
    for tower in ("scene_tower", "product_tower"):
        assert torch.equal(a.params["params"][tower]["embedding"], b.params["params"][tower]["embedding"])
        assert torch.equal(a.opt_state["sum_of_squares"]["params"][tower]["embedding"],
                           b.opt_state["sum_of_squares"]["params"][tower]["embedding"])


@pytest.mark.parametrize("hot", [False, True])
def test_train_steps_inbatch_long_run_hint_and_early_end(dev, hot):
    """In-batch groups are sorted and screened one group ahead; with a hot id (a run far beyond 32 positions) the hint
    says "long" and the optimizer keeps its long-run launch, without one it is skipped -- both bit-identical to the
    stepwise loop, with a ragged last batch (sorted on its own) in between; an iterator that ends early raises
    StopIteration after the steps it fed."""
    from esrecsys_amd.pinterest.train_shop_the_look import train_step, train_steps
    Vs, Vp, D, B, steps = 3000, 5000, 64, 256, 19
    rng = np.random.default_rng(31 + hot)

    def draw(V, n):
        ids = rng.integers(0, V, n).astype(np.int32)
        if hot:
            ids[rng.random(n) < 0.4] = 7
        return torch.from_numpy(ids).to(dev)
    batches = [(draw(Vs, B), draw(Vp, B), None) for _ in range(steps)]
    batches[12] = (draw(Vs, 128), draw(Vp, 128), None)  # ragged: its group falls back to per-step sorts
    a, b = _state(dev, Vs, Vp, D, 4), _state(dev, Vs, Vp, D, 4)
    a, losses = train_steps(a, iter(batches), steps, 0.1, float(B), scale=6.0)
    ref = []
    for scene, pos, _ in batches:
        b, l = train_step(b, scene, pos, None, 0.1, float(B), scale=6.0)
        ref.append(l)
    assert torch.equal(losses, torch.stack(ref))
    for tower in ("scene_tower", "product_tower"):
        assert torch.equal(a.params["params"][tower]["embedding"], b.params["params"][tower]["embedding"])
    with pytest.raises(StopIteration):
        train_steps(_state(dev, Vs, Vp, D, 4), iter(batches[:10]), 15, 0.1, float(B), scale=6.0)
    torch.cuda.synchronize()


def test_train_steps_group_after_a_group_of_larger_batches(dev):
    """Groups are sorted one ahead of the group being stepped: a group of SMALLER batches sorted while a group of larger
    ones waits must not leave that group the smaller workspace (found by scripts/fuzz_loops.py: esr_triplet_train_steps
    refused it)."""
    from esrecsys_amd.pinterest.train_shop_the_look import train_step, train_steps
    Vs, Vp, D = 3000, 5000, 64
    rng = np.random.default_rng(41)
    sizes = [512] * 16 + [129] * 9 + [1024] * 9
    batches = [tuple(torch.from_numpy(rng.integers(0, V, b).astype(np.int32)).to(dev) for V in (Vs, Vp, Vp)) for b in sizes]
    a, b = _state(dev, Vs, Vp, D, 3), _state(dev, Vs, Vp, D, 3)
    a, losses = train_steps(a, iter(batches), len(sizes), 0.1, 512.0)
    ref = []
    for scene, pos, neg in batches:
        b, l = train_step(b, scene, pos, neg, 0.1, 512.0)
        ref.append(l)
    assert torch.equal(losses, torch.stack(ref))
    for tower in ("scene_tower", "product_tower"):
        assert torch.equal(a.params["params"][tower]["embedding"], b.params["params"][tower]["embedding"])


@pytest.mark.parametrize("B,steps,ids", [(256, 7, "uniform"), (2048, 19, "hot"), (8192, 9, "uniform"), (128, 150, "hot"),
                                         (512, 27, "mixed")])
def test_reference_shaped_loop_over_presorted_equals_train_steps(dev, B, steps, ids):
    """The reference's own loop (pinterest/train_shop_the_look.py:190-221) -- ``for scene, pos, neg in it: state, loss =
    train_step(state, scene, pos, neg, reg, bs)`` -- over ``presorted(state, it)``: the id lists of eight coming batches
    are sorted and planned together and train_step steps each handle with one library call.  Bit for bit train_steps; an
    in-batch batch in the middle passes through; a handle steps once."""
    from esrecsys_amd.pinterest.train_shop_the_look import PlannedTriplets, presorted, train_step, train_steps
    Vs, Vp, D = 3000, 5000, 64
    rng = np.random.default_rng(B + steps)

    def draw(V):
        if ids == "hot" or (ids == "mixed" and rng.random() < 0.5):
            return np.where(rng.random(B) < 0.4, rng.integers(0, 3, B), rng.integers(0, V, B)).astype(np.int32)
        return rng.integers(0, V, B).astype(np.int32)
    batches = [(torch.from_numpy(draw(Vs)).to(dev), torch.from_numpy(draw(Vp)).to(dev),
                torch.from_numpy(draw(Vp)).to(dev)) for _ in range(steps)]
    a, b = _state(dev, Vs, Vp, D, 3), _state(dev, Vs, Vp, D, 3)
    a, want = train_steps(a, iter(batches), steps, 0.1, float(B))
    got, handles = [], 0
    last = None
    for scene, pos, neg in presorted(b, iter(batches)):
        handles += isinstance(scene, PlannedTriplets)
        last = scene
        b, loss = train_step(b, scene, pos, neg, 0.1, float(B))
        got.append(loss.clone())
    assert handles >= steps - 1 and int(b.step) == steps     # (a single left-over batch sorts inside its own step)
    assert torch.equal(torch.stack(got), want)
    for tower in ("scene_tower", "product_tower"):
        assert torch.equal(a.params["params"][tower]["embedding"], b.params["params"][tower]["embedding"])
    if isinstance(last, PlannedTriplets):
        with pytest.raises(RuntimeError):
            train_step(b, last, None, None, 0.1, float(B))
    # an in-batch batch between triplet batches goes through unchanged
    mixed = batches[:3] + [(batches[0][0], batches[0][1], None)] + batches[3:5]
    kinds = [isinstance(s, PlannedTriplets) for s, _, _ in presorted(b, iter(mixed))]
    assert kinds.count(False) >= 1 and len(kinds) == len(mixed)


def test_planned_batch_refuses_another_state(dev):
    """A PlannedTriplets handle steps the towers its presorted() context captured: handing train_step a state with other
    tables (a restore, a swap) or another learning rate must raise instead of stepping the old tables, and the handle
    stays usable for the right state."""
    from esrecsys_amd import optim
    from esrecsys_amd.pinterest.train_shop_the_look import PlannedTriplets, presorted, train_step
    Vs, Vp, D, B = 3000, 5000, 64, 256
    rng = np.random.default_rng(1)
    batches = [tuple(torch.from_numpy(rng.integers(0, V, B).astype(np.int32)).to(dev) for V in (Vs, Vp, Vp))
               for _ in range(4)]
    a, other = _state(dev, Vs, Vp, D, 3), _state(dev, Vs, Vp, D, 3)
    it = presorted(a, iter(batches))
    scene, pos, neg = next(it)
    assert isinstance(scene, PlannedTriplets)
    before = other.params["params"]["scene_tower"]["embedding"].clone()
    with pytest.raises(RuntimeError, match="another state"):
        train_step(other, scene, pos, neg, 0.1, float(B))
    with pytest.raises(RuntimeError, match="learning rate"):
        train_step(a.replace(tx=optim.sparse_adagrad(0.01)), scene, pos, neg, 0.1, float(B))
    assert torch.equal(other.params["params"]["scene_tower"]["embedding"], before)
    a, loss = train_step(a, scene, pos, neg, 0.1, float(B))     # the refused calls did not consume the handle
    assert np.isfinite(float(loss)) and int(a.step) == 1


@pytest.mark.parametrize("B,D,ids,dt", [(1024, 128, "uniform", "f32"), (2048, 128, "hot", "f32"), (512, 64, "hot", "f32"),
                                        (8192, 128, "uniform", "f32"), (1024, 128, "hot", "bf16"), (1024, 128, "uniform", "bf16"),
                                        (16384, 128, "uniform", "f32"), (384, 128, "uniform", "bf16"),
                                        (384, 128, "uniform", "f32"), (640, 128, "uniform", "f32"), (640, 64, "hot", "f32")])
def test_inbatch_one_call_step_equals_fwd_bwd_plus_update(dev, monkeypatch, B, D, ids, dt):
    """The in-batch step as ONE library call (esr_inbatch_train_step_f16x2) -- with merge<Q> and the scene tower's
    Adagrad on a second stream beside pass C, and without the second stream -- against rounds 1-4's
    esr_inbatch_towers_fwd_bwd_f16x2 + esr_sparse_adagrad_scatter_multi: same kernels on the same values, so losses, towers
    and accumulators are bit-identical; through train_step alone (sort inside the call) and through train_steps (lists of
    eight batches sorted ahead, long-run hints).  "hot": 40 % of the ids are three rows (runs of hundreds: the update's
    long-run launch on each half of the occurrence list).  B = 384 / 640 with f32 towers: the sizes at which the loop takes
    the merging update and the single step does not -- rows that occur twice in a batch came out one ulp apart until the
    merging update stopped the compiler from contracting a gradient row's last multiplication into the run sum
    (scripts/fuzz_loops.py, seed 717)."""
    import esrecsys_amd.pinterest.train_shop_the_look as stl
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.pinterest.models import STLModel
    Vs, Vp, steps = 5000, 7000, 5
    rng = np.random.default_rng(B + D)

    def draw(V):
        if ids == "hot":
            return np.where(rng.random(B) < 0.4, rng.integers(0, 3, B), rng.integers(0, V, B)).astype(np.int32)
        return rng.integers(0, V, B).astype(np.int32)
    batches = [(torch.from_numpy(draw(Vs)).to(dev), torch.from_numpy(draw(Vp)).to(dev), None) for _ in range(steps)]

    def state():
        g = torch.Generator(device=dev).manual_seed(3)
        tdt = torch.bfloat16 if dt == "bf16" else torch.float32  # (bf16 towers: the fp16 one-plane kernels, both paths)
        params = {"params": {"scene_tower": {"embedding": (torch.randn((Vs, D), generator=g, device=dev) * 0.3).to(tdt)},
                             "product_tower": {"embedding": (torch.randn((Vp, D), generator=g, device=dev) * 0.3).to(tdt)}}}
        model = STLModel(output_size=D, num_scenes=Vs, num_products=Vp, device=dev)
        return TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(0.05))

    def run(onecall, overlap, loop):
        monkeypatch.setattr(stl, "_INBATCH_ONECALL", onecall)
        monkeypatch.setattr(stl, "_INBATCH_OVERLAP", overlap)
        st = state()
        if loop:
            st, losses = stl.train_steps(st, iter(batches), steps, 0.1, float(B), scale=4.0, precision="f16x2")
        else:
            losses = []
            for b in batches:
                st, l = stl.train_step(st, b[0], b[1], None, 0.1, float(B), scale=4.0, precision="f16x2")
                losses.append(l.clone())
            losses = torch.stack(losses)
        torch.cuda.synchronize()
        p, a = st.params["params"], st.opt_state["sum_of_squares"]["params"]
        return [losses] + [t[k]["embedding"] for t in (p, a) for k in ("scene_tower", "product_tower")]
    want = run(False, False, False)
    assert np.isfinite(want[0].cpu().numpy()).all() and int(want[0].numel()) == steps
    bad = []
    names = ("losses", "scene tower", "product tower", "scene accumulator", "product accumulator")
    # (one call, no second stream, loop: the lists' long-run hints are known -- "uniform" steps end with the merging update
    # (inbatch_merge_update_kernel at D = 128: merges + Adagrad in one launch), "hot" ones fall back to merge + update)
    for onecall, overlap, loop in ((True, True, False), (True, False, False), (True, True, True), (True, False, True),
                                   (False, False, True)):
        got = run(onecall, overlap, loop)
        for name, x, y in zip(names, got, want):
            if not torch.equal(x, y):
                bad.append((onecall, overlap, loop, name, float((x.float() - y.float()).abs().max())))
    monkeypatch.setenv("ESR_IB2H_MERGE_UPDATE", "0")   # the same loop without the merging update
    for name, x, y in zip(names, run(True, False, True), want):
        if not torch.equal(x, y):
            bad.append(("merge + update", name, float((x.float() - y.float()).abs().max())))
    monkeypatch.delenv("ESR_IB2H_MERGE_UPDATE")
    assert want[1].dtype == (torch.bfloat16 if dt == "bf16" else torch.float32)
    assert not bad, bad


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_inbatch_merging_update_lse_and_a_false_hint(dev, monkeypatch, dt):
    """esr_inbatch_train_step_f16x2 with long_runs = 0 ends with the merging update (merges + Adagrad in one launch, lse from
    fac2h_kernel): loss, lse, towers and accumulators equal the merge + update launches bit for bit.  A caller whose hint is
    wrong -- a run of 70 equal ids handed over as "no long run" -- gets a NaN loss, not a silently partial update."""
    from esrecsys_amd import ops
    Vq, Vc, B, D = 3000, 4000, 1024, 128
    rng = np.random.default_rng(11)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32

    def towers():
        g = torch.Generator(device=dev).manual_seed(5)
        return [(torch.randn((Vq, D), generator=g, device=dev) * 0.3).to(tdt), torch.full((Vq, D), 0.1, device=dev),
                (torch.randn((Vc, D), generator=g, device=dev) * 0.3).to(tdt), torch.full((Vc, D), 0.1, device=dev)]
    qi = torch.from_numpy(rng.integers(0, Vq, B).astype(np.int32)).to(dev)
    ci = torch.from_numpy(rng.integers(0, Vc, B).astype(np.int32)).to(dev)
    srt, prm = ops.segment_sort_batched([[qi, ci]], (0, Vq), Vq + Vc)

    def step(mu, q=qi, c=ci, pre=(srt[0], prm[0])):
        monkeypatch.setenv("ESR_IB2H_MERGE_UPDATE", mu)
        t = towers()
        loss, lse = ops.inbatch_train_step(t[0], t[1], t[2], t[3], q, c, 4.0, 0.1, float(B), 0.05, presorted=pre,
                                           long_runs=0, want_lse=True)
        torch.cuda.synchronize()
        return [loss, lse] + t
    a, b = step("1"), step("0")
    assert np.isfinite(float(a[0])) and all(torch.equal(x, y) for x, y in zip(a, b))
    hot = qi.clone()
    hot[:70] = 7                                            # a run that outgrows its head chunk
    hs, hp = ops.segment_sort_batched([[hot, ci]], (0, Vq), Vq + Vc)
    lied = step("1", q=hot, pre=(hs[0], hp[0]))
    assert np.isnan(float(lied[0]))
    monkeypatch.delenv("ESR_IB2H_MERGE_UPDATE")
