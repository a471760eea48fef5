"""GPU: the one-pass GloVe train step (esr_glove_train_step: loss + on-chip gradients + sparse Adagrad on a
double-buffered table) against (a) the two-call path apply_model + update_model -- same sort, same association of
every sum: the embedding table and accumulator agree to an f32 rounding -- and (b) the fp64 oracle of wikipedia/train_cooccurence.py:71-101 with sparse Adagrad."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ids(kind, V, shape, rng):
    if kind == "uniform":
        return rng.integers(0, V, shape).astype(np.int32)
    if kind == "same":
        return np.full(shape, 7 % V, np.int32)
    # zipf: a few hot tokens take most occurrences (runs of hundreds: the long-run kernel and its chunk partials)
    w = 1.0 / np.arange(1, V + 1)
    return rng.permutation(V)[rng.choice(V, size=shape, p=w / w.sum())].astype(np.int32)


def _make_state(V, D, mode, dev, seed=5, lr=0.05):
    from esrecsys_amd import TrainState, optim
    from esrecsys_amd.wikipedia.models import Glove
    model = Glove(num_embeddings=V, features=D, loss_mode=mode, device=dev)
    params = model.init(seed, None)["params"]
    g = torch.Generator().manual_seed(seed + 1)
    params["_bias"]["embedding"].copy_((torch.randn((V, 1), generator=g) * 0.05).to(dev))
    return TrainState.create(apply_fn=model.apply, params=params, tx=optim.sparse_adagrad(lr))


@pytest.mark.parametrize("mode", ["reference", "diagonal"])
@pytest.mark.parametrize("kind", ["uniform", "zipf", "same"])
@pytest.mark.parametrize("V,D,B", [(5000, 256, 4096), (300, 64, 1000), (2000, 100, 777), (1000, 6, 64), (50000, 128, 40000)])
def test_fused_step_equals_two_call_path(dev, mode, kind, V, D, B):
    from esrecsys_amd.wikipedia.train_cooccurence import apply_model, fused_step_available, train_step, update_model
    rng = np.random.default_rng(V + B)
    a, b = _make_state(V, D, mode, dev), _make_state(V, D, mode, dev)
    assert fused_step_available(a)
    for step in range(3):
        inputs = _ids(kind, V, (2, B), rng)
        target = np.exp(rng.uniform(np.log(0.1), np.log(1000.0), B)).astype(np.float32)
        a, la = train_step(a, inputs, target)
        grads, lb = apply_model(b, inputs, target)
        b = update_model(b, grads)
        assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb)), (step, float(la), float(lb))
    rv = a.versions[("_token_embedding", "embedding")]
    assert rv.dirty and int((rv.loc & 1).sum()) > 0, "some rows must live in the second buffer before consolidation"
    pa, pb = a.params, b.params            # reading .params consolidates
    assert not rv.dirty and int(rv.loc.sum()) == 0
    assert int(a.step) == int(b.step) == 3
    assert rel_err(pa["_token_embedding"]["embedding"].cpu().numpy(), pb["_token_embedding"]["embedding"].cpu().numpy()) <= 1e-6
    assert rel_err(a.opt_state["sum_of_squares"]["_token_embedding"]["embedding"].cpu().numpy(),
                   b.opt_state["sum_of_squares"]["_token_embedding"]["embedding"].cpu().numpy()) <= 1e-6
    # the bias gradient is associated differently: fp64 run sums of s here, f32 sums of per-occurrence gradients there
    # (thousands of terms for a hot token: the two-call path itself is only good to ~1e-6 on those)
    assert rel_err(pa["_bias"]["embedding"].cpu().numpy(), pb["_bias"]["embedding"].cpu().numpy()) <= 1e-5
    assert rel_err(a.opt_state["sum_of_squares"]["_bias"]["embedding"].cpu().numpy(),
                   b.opt_state["sum_of_squares"]["_bias"]["embedding"].cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("mode", ["reference", "diagonal"])
def test_fused_step_trajectory_vs_fp64_oracle(dev, mode):
    from esrecsys_amd.wikipedia.train_cooccurence import train_step
    from oracle import glove as o_glove
    from oracle import optim as o_optim
    V, D, B, lr = 700, 48, 512, 0.05
    state = _make_state(V, D, mode, dev, lr=lr)
    emb = state.params["_token_embedding"]["embedding"].cpu().numpy().astype(np.float64)
    bias = state.params["_bias"]["embedding"].cpu().numpy().astype(np.float64)
    a_e, a_b = np.full_like(emb, 0.1), np.full_like(bias, 0.1)
    rng = np.random.default_rng(3)
    for step in range(4):
        inputs = _ids("zipf" if step % 2 else "uniform", V, (2, B), rng)
        target = rng.uniform(0.1, 300.0, B).astype(np.float32)
        state, loss = train_step(state, inputs, target)
        el, gdot, gs = o_glove.loss_and_grads(emb, bias, inputs, target.astype(np.float64), mode, np.float64)
        ids, rows, gb = o_glove.row_grads(emb, inputs, gdot, gs, np.float64)
        emb, a_e = o_optim.sparse_adagrad_update(emb, a_e, ids, rows, lr, dtype=np.float64)
        bias, a_b = o_optim.sparse_adagrad_update(bias, a_b, ids, gb[:, None], lr, dtype=np.float64)
        assert abs(float(loss) - el) <= 1e-5 * abs(el)
    p = state.params
    assert rel_err(p["_token_embedding"]["embedding"].cpu().numpy(), emb) <= 1e-5
    assert rel_err(p["_bias"]["embedding"].cpu().numpy(), bias) <= 1e-5
    assert rel_err(state.opt_state["sum_of_squares"]["_token_embedding"]["embedding"].cpu().numpy(), a_e) <= 1e-5


def test_train_epoch_takes_the_fused_step_and_params_stay_plain(dev, monkeypatch):
    """train_epoch (wikipedia/train_cooccurence.py:103-112) runs the one-pass step under sparse Adagrad; whoever reads
    state.params afterwards (find_knn, checkpoints) sees a plain table equal to the two-call path's."""
    from esrecsys_amd import checkpoint
    from esrecsys_amd.wikipedia.train_cooccurence import find_knn, train_epoch
    V, D, B, K = 3000, 64, 1024, 5
    rng = np.random.default_rng(8)
    batches = [(_ids("uniform", V, (2, B), rng), rng.uniform(0.1, 300.0, B).astype(np.float32)) for _ in range(K)]
    a, la = train_epoch(_make_state(V, D, "reference", dev), K, iter(batches))
    assert a.versions
    monkeypatch.setenv("ESR_GLOVE_FUSED", "0")
    b, lb = train_epoch(_make_state(V, D, "reference", dev), K, iter(batches))
    assert not b.versions
    assert abs(la - lb) <= 2e-6 * abs(lb)
    token = torch.tensor([1, 5, 9], dtype=torch.int32, device=dev)
    model = a.apply_fn.__self__
    sa, ia = find_knn(model, a.params, token)
    sb, ib = find_knn(model, b.params, token)
    assert rel_err(sa.cpu().numpy(), sb.cpu().numpy()) <= 1e-6 and ia.shape == ib.shape == (V, 3)
    # a checkpoint written after fused steps holds the consolidated table; restoring into a state that has versions works
    data = checkpoint.to_bytes(a)
    c, _ = train_epoch(_make_state(V, D, "reference", dev), 2, iter(batches))   # leaves rows in the second buffer
    c = checkpoint.from_bytes(c, data)
    assert torch.equal(c.params["_token_embedding"]["embedding"], a.params["_token_embedding"]["embedding"])
    assert torch.equal(c.params["_bias"]["embedding"], a.params["_bias"]["embedding"])
    assert int(c.step) == K


def test_rows_consolidate(dev):
    from esrecsys_amd import ops
    for V, D in ((1000, 256), (77, 6), (5, 4)):
        g = torch.Generator(device=dev).manual_seed(V)
        primary = torch.randn((V, D), generator=g, device=dev)
        shadow = torch.randn((V, D), generator=g, device=dev)
        loc = (torch.rand(V, generator=g, device=dev) < 0.3).to(torch.uint8)
        # stamped bytes (esr_versioned.h): bit 0 = the location, bits 1..7 = the stamp of the step that moved the row
        loc |= (torch.randint(0, 128, (V,), generator=g, device=dev, dtype=torch.int32) << 1).to(torch.uint8)
        want = torch.where((loc & 1).bool()[:, None], shadow, primary)
        keep = loc & 1
        stamped = loc.clone()
        ops.rows_restamp(stamped)
        assert torch.equal(stamped, keep), "restamp clears the stamps and keeps the locations"
        ops.rows_consolidate(primary, shadow, loc)
        assert torch.equal(primary, want) and int(loc.sum()) == 0


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
def test_stamps_wrap_around(dev, kind):
    """140 one-pass steps (the stamp counter passes 127 and the bytes are re-stamped) leave the same bits as 140 steps
    with the table consolidated after every one (stamps always 1): the stamps only say WHERE a row is read, never what
    is computed."""
    from esrecsys_amd.wikipedia.train_cooccurence import train_step
    V, D, B = 400, 64, 150
    rng = np.random.default_rng(9)
    a, b = _make_state(V, D, "reference", dev), _make_state(V, D, "reference", dev)
    stamps = []
    for step in range(140):
        inputs = _ids(kind, V, (2, B), rng)
        target = np.exp(rng.uniform(np.log(0.1), np.log(1000.0), B)).astype(np.float32)
        a, la = train_step(a, inputs, target)
        stamps.append(a.versions[("_token_embedding", "embedding")].stamp)
        b, lb = train_step(b, inputs, target)
        _ = b.params  # consolidates: every byte back to 0
        assert float(la) == float(lb), step
    assert max(stamps) == 127 and stamps[127] == 1, "the counter wrapped"
    assert torch.equal(a.params["_token_embedding"]["embedding"], b.params["_token_embedding"]["embedding"])
    assert torch.equal(a.params["_bias"]["embedding"], b.params["_bias"]["embedding"])


def test_config_c3_full_size_fused_step(dev):
    """BASELINE configs[2] at full size (V = 465 537, D = 256, B = 65 536): fused == two-call path to an f32 rounding on the
    embedding table, and a size-independent property: rows no pair touches keep their bits."""
    from esrecsys_amd.wikipedia.train_cooccurence import apply_model, train_step, update_model
    V, D, B = 465_537, 256, 65_536
    a, b = _make_state(V, D, "reference", dev), _make_state(V, D, "reference", dev)
    before = a.params["_token_embedding"]["embedding"].clone()
    g = torch.Generator(device=dev).manual_seed(2)
    touched = torch.zeros(V, dtype=torch.bool, device=dev)
    for _ in range(2):
        inputs = torch.randint(0, V, (2, B), generator=g, device=dev, dtype=torch.int32)
        target = torch.exp(np.log(0.1) + torch.rand(B, generator=g, device=dev) * np.log(1e4))
        a, la = train_step(a, inputs, target)
        grads, lb = apply_model(b, inputs, target)
        b = update_model(b, grads)
        touched[inputs.reshape(-1).long()] = True
        assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb))
    ea, eb = a.params["_token_embedding"]["embedding"], b.params["_token_embedding"]["embedding"]
    assert rel_err(ea.cpu().numpy(), eb.cpu().numpy()) <= 1e-6
    assert torch.equal(ea[~touched], before[~touched]) and not torch.equal(ea[touched], before[touched])


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
@pytest.mark.parametrize("B,K", [(1024, 11), (2048, 8), (16, 3), (3000, 5), (2048, 27), (64, 140), (20000, 11),
                                 (40000, 5)])
def test_train_epoch_batched_sort_equals_in_line_sort(dev, B, K, kind, monkeypatch):
    """The id lists of eight coming batches are sorted by one batched call (esr_segment_sort_ids_batched); the epoch must
    be bit-identical to the one that sorts every list inside its own step -- groups of 8 + 3, exactly 8, fewer than a
    group; lists with plans (<= 32 768 ids: the reference's default batch of 2048 pairs) and without (40 000 and 80 000
    ids: the steps resolve their own records, only the sort is grouped)."""
    import esrecsys_amd.wikipedia.train_cooccurence as tc
    V, D = 3000, 64
    rng = np.random.default_rng(B + K)
    batches = [(_ids(kind, V, (2, B), rng), rng.uniform(0.1, 300.0, B).astype(np.float32)) for _ in range(K)]
    monkeypatch.setattr(tc, "_SORT_BATCH", 8)
    a, la = tc.train_epoch(_make_state(V, D, "reference", dev), K, iter(batches))
    monkeypatch.setattr(tc, "_SORT_BATCH", 1)
    monkeypatch.setattr(tc, "_PRESORT", False)
    b, lb = tc.train_epoch(_make_state(V, D, "reference", dev), K, iter(batches))
    assert la == lb
    assert torch.equal(a.params["_token_embedding"]["embedding"], b.params["_token_embedding"]["embedding"])
    assert torch.equal(a.params["_bias"]["embedding"], b.params["_bias"]["embedding"])


def test_one_pass_step_is_refused_without_room_for_the_second_buffer(dev, monkeypatch):
    """The one-pass steps double the memory of the tables they update (train_state.shadow_fits documents the largest V):
    without room for the second buffer fused_step_available says no, train_step / train_epoch take the gradient-row path
    (same result), and asking for the second buffer anyway raises ShadowMemoryError instead of running out of memory."""
    import esrecsys_amd.train_state as ts
    from esrecsys_amd.wikipedia.train_cooccurence import fused_step_available, train_step
    V, D, B = 2000, 64, 512
    rng = np.random.default_rng(1)
    inputs = _ids("uniform", V, (2, B), rng)
    target = rng.uniform(0.1, 300.0, B).astype(np.float32)
    a, b = _make_state(V, D, "reference", dev), _make_state(V, D, "reference", dev)
    a, la = train_step(a, inputs, target)                      # one-pass
    monkeypatch.setattr(ts, "shadow_fits", lambda table, reserve=0: False)
    assert not fused_step_available(b)
    with pytest.raises(ts.ShadowMemoryError):
        ts.row_versions(b, ("_token_embedding", "embedding"))
    b, lb = train_step(b, inputs, target)                      # gradient rows: no second buffer was made
    assert not b.versions and abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb))
    assert rel_err(a.params["_token_embedding"]["embedding"].cpu().numpy(),
                   b.params["_token_embedding"]["embedding"].cpu().numpy()) <= 1e-6
    assert fused_step_available(a)   # (a table that HAS its second buffer keeps using it)


@pytest.mark.parametrize("B,K", [(40000, 7), (3000, 6)])
def test_train_epoch_side_stream_sort_equals_in_line_sort(dev, B, K, monkeypatch):
    """Lists beyond the grouped sort's limit go to the side stream one by one (ring of sort buffers, gated on the update
    kernel's start word -- esr_stream_gate): forced here by lowering the limit; bit-identical to the in-line sorts."""
    import esrecsys_amd.wikipedia.train_cooccurence as tc
    V, D = 3000, 64
    rng = np.random.default_rng(B + K)
    batches = [(_ids("zipf", V, (2, B), rng), rng.uniform(0.1, 300.0, B).astype(np.float32)) for _ in range(K)]
    monkeypatch.setattr(tc, "_GROUP_SORT_MAX_IDS", 1024)
    monkeypatch.setattr(tc, "_PRESORT_MIN_IDS", 1024)
    a, la = tc.train_epoch(_make_state(V, D, "reference", dev), K, iter(batches))
    monkeypatch.setattr(tc, "_SORT_BATCH", 1)
    monkeypatch.setattr(tc, "_PRESORT", False)
    b, lb = tc.train_epoch(_make_state(V, D, "reference", dev), K, iter(batches))
    assert la == lb
    assert torch.equal(a.params["_token_embedding"]["embedding"], b.params["_token_embedding"]["embedding"])
    assert torch.equal(a.params["_bias"]["embedding"], b.params["_bias"]["embedding"])


def test_stream_gate_opens_on_the_word_and_on_the_timeout(dev):
    """esr_stream_gate: a stream behind the gate runs on once the word has reached the value (wrap-safe compare), and
    after the timeout when it never does."""
    import time
    from esrecsys_amd import ops
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops.stream_gate(flag, 3, timeout_us=5_000_000)
        out.fill_(7)
    time.sleep(0.05)
    assert not side.query()              # still held: the word is 0
    flag.fill_(3)                        # (main stream)
    side.synchronize()
    assert int(out) == 7
    flag.fill_(-5)                       # 0xFFFFFFFB: "before" 2 in sequence-number order
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        ops.stream_gate(flag, 2, timeout_us=20_000)
    side.synchronize()
    dt = time.perf_counter() - t0
    assert 0.015 < dt < 1.0              # released by the timeout, not at once and not never
    flag.fill_(2 ** 31 - 1)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops.stream_gate(flag, -(2 ** 31) + 5, timeout_us=5_000_000)   # 0x80000005 is AFTER 0x7FFFFFFF: held ...
    time.sleep(0.02)
    assert not side.query()
    flag.fill_(-(2 ** 31) + 5)           # ... until the word wraps past it
    side.synchronize()


def test_train_epoch_long_lists_ragged_batch_and_early_end(dev, monkeypatch):
    """Grouped sort of long lists (no plans): a batch of another size inside the epoch makes its group fall back to
    per-step sorts, still bit-identical to the in-line loop; an iterator that ends early raises StopIteration."""
    import esrecsys_amd.wikipedia.train_cooccurence as tc
    V, D, B, K = 3000, 64, 20000, 12
    rng = np.random.default_rng(77)
    sizes = [B] * K
    sizes[6] = 17001
    batches = [(_ids("zipf", V, (2, b), rng), rng.uniform(0.1, 300.0, b).astype(np.float32)) for b in sizes]
    a, la = tc.train_epoch(_make_state(V, D, "reference", dev), K, iter(batches))
    monkeypatch.setattr(tc, "_SORT_BATCH", 1)
    monkeypatch.setattr(tc, "_PRESORT", False)
    b, lb = tc.train_epoch(_make_state(V, D, "reference", dev), K, iter(batches))
    assert la == lb
    assert torch.equal(a.params["_token_embedding"]["embedding"], b.params["_token_embedding"]["embedding"])
    monkeypatch.setattr(tc, "_SORT_BATCH", 8)
    with pytest.raises(StopIteration):
        tc.train_epoch(_make_state(V, D, "reference", dev), K + 3, iter(batches))
    torch.cuda.synchronize()


def test_train_epoch_mixed_grouped_and_side_stream_batches(dev, monkeypatch):
    """An epoch that mixes lists beyond the grouped sort's limit (side stream, gated on the start word) with short ones
    (grouped, stepped by the group call that announces no start): bit-identical to the in-line loop, and no gate sits
    out its one-second timeout."""
    import time
    import esrecsys_amd.wikipedia.train_cooccurence as tc
    V, D = 3000, 64
    rng = np.random.default_rng(123)
    sizes = [3000, 300, 300, 3000, 300, 300, 300, 3000, 3000, 300, 300, 300, 300, 300, 300, 300, 300, 300, 3000, 300]
    batches = [(_ids("zipf", V, (2, b), rng), rng.uniform(0.1, 300.0, b).astype(np.float32)) for b in sizes]
    monkeypatch.setattr(tc, "_GROUP_SORT_MAX_IDS", 1024)
    monkeypatch.setattr(tc, "_PRESORT_MIN_IDS", 1024)
    tc.train_epoch(_make_state(V, D, "reference", dev), len(sizes), iter(batches))   # (warm: workspaces, events)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a, la = tc.train_epoch(_make_state(V, D, "reference", dev), len(sizes), iter(batches))
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.9
    monkeypatch.setattr(tc, "_SORT_BATCH", 1)
    monkeypatch.setattr(tc, "_PRESORT", False)
    b, lb = tc.train_epoch(_make_state(V, D, "reference", dev), len(sizes), iter(batches))
    assert la == lb
    assert torch.equal(a.params["_token_embedding"]["embedding"], b.params["_token_embedding"]["embedding"])
    assert torch.equal(a.params["_bias"]["embedding"], b.params["_bias"]["embedding"])


@pytest.mark.parametrize("mode", ["reference", "diagonal"])
@pytest.mark.parametrize("V,D,B,K", [(465_537, 256, 2048, 120), (3000, 64, 4096, 40), (100_000, 128, 300, 64)])
def test_last_workgroup_finalize_equals_finalize_launch(dev, mode, V, D, B, K, monkeypatch):
    """Short lists whose plan says that no run outgrows its head chunk: the update kernel's last-arriving workgroup is the
    finalize step (loss scalar + the bias table's Adagrad read the other workgroups' sums through the memory side).  Bit
    for bit the epoch that keeps the finalize launch (ESR_GLOVE_FIN_FUSED=0), over enough steps and workgroups (1024
    across the eight XCDs at the reference's default batch on the C3 table) that a stale read would show."""
    import esrecsys_amd.wikipedia.train_cooccurence as tc
    g = torch.Generator(device=dev).manual_seed(B + K)
    batches = [(torch.randint(0, V, (2, B), generator=g, device=dev, dtype=torch.int32),
                torch.exp(np.log(0.1) + torch.rand(B, generator=g, device=dev) * np.log(1e4))) for _ in range(K)]
    monkeypatch.setenv("ESR_GLOVE_FIN_FUSED", "1")
    a, la = tc.train_epoch(_make_state(V, D, mode, dev), K, iter(batches))
    monkeypatch.setenv("ESR_GLOVE_FIN_FUSED", "0")
    b, lb = tc.train_epoch(_make_state(V, D, mode, dev), K, iter(batches))
    assert la == lb and np.isfinite(la)
    assert torch.equal(a.params["_token_embedding"]["embedding"], b.params["_token_embedding"]["embedding"])
    assert torch.equal(a.params["_bias"]["embedding"], b.params["_bias"]["embedding"])
    assert torch.equal(a.opt_state["sum_of_squares"]["_bias"]["embedding"], b.opt_state["sum_of_squares"]["_bias"]["embedding"])
