"""CPU: host-side logic of the drop-in API that needs no GPU -- S4 generate_triplets' sampling protocol
(pinterest/train_shop_the_look.py:72-91), flag namespaces with the reference's names and defaults."""
import numpy as np
import pytest


def _pairs(count):
    return [("scene%03d" % i, "prod%03d" % i) for i in range(count)]


@pytest.mark.parametrize("count,num_neg", [(57, 5), (10, 1), (11, 3), (200, 7)])
def test_generate_triplets_protocol(count, num_neg):
    """Per positive pair i: exactly num_neg triplets (scene_i, pos_i, neg) in input order; every 10th positive
    (i % 10 == 0) goes to the test split, the rest to train; negatives are the PRODUCT of another positive pair,
    drawn from randint(0, count - 1) -- upper bound exclusive, so the last pair's product is never a negative."""
    from esrecsys_amd.pinterest.train_shop_the_look import generate_triplets
    sp = _pairs(count)
    train, test = generate_triplets(sp, num_neg)
    n_test = len(range(0, count, 10))
    assert len(test) == n_test * num_neg and len(train) == (count - n_test) * num_neg
    products = {p: i for i, (_, p) in enumerate(sp)}
    # order + grouping: positives appear in input order, num_neg consecutive triplets each
    exp_train = [i for i in range(count) if i % 10 != 0]
    exp_test = [i for i in range(count) if i % 10 == 0]
    for out, exp in ((train, exp_train), (test, exp_test)):
        for g, i in enumerate(exp):
            grp = out[g * num_neg:(g + 1) * num_neg]
            assert all(t[0] == sp[i][0] and t[1] == sp[i][1] for t in grp)
            assert all(t[2] in products for t in grp)
    negs = np.array([products[t[2]] for t in train + test])
    assert negs.min() >= 0 and negs.max() <= count - 2, "the last item must never be sampled (exclusive upper bound)"


def test_generate_triplets_negative_distribution_and_determinism():
    from esrecsys_amd.pinterest.train_shop_the_look import generate_triplets
    count, num_neg = 41, 50
    sp = _pairs(count)
    a = generate_triplets(sp, num_neg)
    b = generate_triplets(sp, num_neg)
    assert a == b, "PRNGKey(0)-style fixed seed: the split is reproducible"
    c = generate_triplets(sp, num_neg, seed=1)
    assert c != a
    products = {p: i for i, (_, p) in enumerate(sp)}
    negs = np.array([products[t[2]] for t in a[0] + a[1]])
    hist = np.bincount(negs, minlength=count)
    assert hist[count - 1] == 0 and hist[:count - 1].min() > 0  # uniform over [0, count - 1): all others do occur
    # chi-square against uniform over count - 1 bins, very loose (p ~ 1e-6 at 40 dof is ~ 95)
    exp = len(negs) / (count - 1)
    assert ((hist[:count - 1] - exp) ** 2 / exp).sum() < 95.0


def test_generate_triplets_edge_sizes():
    from esrecsys_amd.pinterest.train_shop_the_look import generate_triplets
    assert generate_triplets([], 5) == ([], [])
    train, test = generate_triplets(_pairs(2), 3)        # randint(0, 1): the only legal negative is item 0
    assert train == [("scene001", "prod001", "prod000")] * 3 and test == [("scene000", "prod000", "prod000")] * 3
    with pytest.raises(ValueError):
        generate_triplets(_pairs(1), 2)                  # randint(0, 0) is an empty range (JAX would return garbage)
    assert generate_triplets(_pairs(5), 0) == ([], [])


def test_flag_defaults_match_the_reference():
    """pinterest/train_shop_the_look.py:46-69 and wikipedia/train_cooccurence.py:34-65."""
    from esrecsys_amd.pinterest.train_shop_the_look import FLAGS as P
    from esrecsys_amd.wikipedia.train_cooccurence import FLAGS as W
    assert (P.num_neg, P.learning_rate, P.regularization, P.output_size, P.batch_size, P.max_steps) == \
        (5, 1e-3, 0.1, 32, 16, 30000)
    assert (P.log_every_steps, P.eval_every_steps, P.checkpoint_every_steps) == (100, 2000, 100000)
    assert (W.embedding_dim, W.batch_size, W.seed, W.shuffle_buffer_size, W.steps_per_epoch, W.num_epochs,
            W.learning_rate, W.checkpoint_every_epochs, W.max_terms) == (64, 2048, 1701, 5000000, 10000, 20, 0.001, 20, 20)


def test_inbatch_split_path_resolution(monkeypatch):
    """which MFMA path a precision string selects (host logic only): the fp16 path needs D <= 128, B % 128 == 0 and
    B <= 16384 (round 5: bf16 tables take it too -- its one-plane kernels; ESR_INBATCH_BF16_TABLES=bf16x3 keeps the older
    path); larger batches fall to the three-plane bf16 path; other shapes to f32"""
    from esrecsys_amd import ops
    monkeypatch.delenv("ESR_INBATCH_AUTO", raising=False)
    monkeypatch.delenv("ESR_INBATCH_BF16_TABLES", raising=False)
    assert ops.inbatch_split_path("auto", 8192, 128) == "f16x2"
    assert ops.inbatch_split_path("auto", 16384, 128) == "f16x2"
    assert ops.inbatch_split_path("auto", 16512, 128) == "bf16x3"
    assert ops.inbatch_split_path("auto", 8192, 128, bf16_tables=True) == "f16x2"
    assert ops.inbatch_split_path("auto", 16512, 128, bf16_tables=True) == "bf16x3"
    monkeypatch.setenv("ESR_INBATCH_BF16_TABLES", "bf16x3")
    assert ops.inbatch_split_path("auto", 8192, 128, bf16_tables=True) == "bf16x3"
    with pytest.raises(ValueError):
        ops.inbatch_split_path("f16x2", 256, 128, bf16_tables=True)
    monkeypatch.delenv("ESR_INBATCH_BF16_TABLES")
    assert ops.inbatch_split_path("f16x2", 256, 128, bf16_tables=True) == "f16x2"
    assert ops.inbatch_split_path("auto", 8200, 128) is None
    # narrower rows ride in the 128-column tiles from D = 64 up; below (and for wider or odd widths) the exact-f32 kernel
    assert ops.inbatch_split_path("auto", 8192, 64) == "f16x2" and ops.inbatch_split_path("auto", 8192, 96) == "f16x2"
    assert ops.inbatch_split_path("auto", 8192, 32) is None and ops.inbatch_split_path("auto", 8192, 256) is None
    assert ops.inbatch_split_path("auto", 8192, 98) is None
    assert ops.inbatch_split_path("f16x2", 256, 32) == "f16x2"   # an explicit request is honoured at any D <= 128
    assert ops.inbatch_split_path("f32", 8192, 128) is None
    assert ops.inbatch_split_path("bf16x3", 256, 128) == "bf16x3"
    with pytest.raises(ValueError):
        ops.inbatch_split_path("f16x2", 32768, 128)
    with pytest.raises(ValueError):
        ops.inbatch_split_path("bf16x3", 100, 128)
    with pytest.raises(ValueError):
        ops.inbatch_split_path("fp8", 256, 128)
    monkeypatch.setenv("ESR_INBATCH_AUTO", "bf16x3")
    assert ops.inbatch_split_path("auto", 8192, 128) == "bf16x3"
    monkeypatch.setenv("ESR_INBATCH_AUTO", "nonsense")
    with pytest.raises(ValueError):
        ops.inbatch_split_path("auto", 8192, 128)


def test_retrieve_mode_resolution(monkeypatch):
    from esrecsys_amd import _lib, ops
    monkeypatch.delenv("ESR_RETRIEVE_EXACT", raising=False)
    # "exact" is the exact split (three bf16 planes); the range-limited fp16 x 2 planes are asked for by name
    assert ops._retrieve_mode("exact") == ops._retrieve_mode("f32") == _lib.RETRIEVE_EXACT
    assert ops._retrieve_mode("f16x2") == _lib.RETRIEVE_F16X2 == 2
    assert ops._retrieve_mode("bf16") == _lib.RETRIEVE_BF16 and ops._retrieve_mode("bf16x3") == _lib.RETRIEVE_EXACT
    monkeypatch.setenv("ESR_RETRIEVE_EXACT", "f16x2")   # round 2's mapping, on request
    assert ops._retrieve_mode("f32") == _lib.RETRIEVE_F16X2
    with pytest.raises(ValueError):
        ops._retrieve_mode("int8")


def test_quiet_gc_parks_and_restores(monkeypatch):
    """The loop helpers run under train_state.quiet_gc: live objects parked in the permanent generation for the duration
    (a full collection inside the loop is then cheap), everything back afterwards, also when the body raises."""
    import gc
    from esrecsys_amd.train_state import quiet_gc
    assert gc.get_freeze_count() == 0
    with quiet_gc():
        assert gc.get_freeze_count() > 0
        gc.collect()           # walks only what was created inside
    assert gc.get_freeze_count() == 0
    with pytest.raises(ValueError):
        with quiet_gc():
            raise ValueError("x")
    assert gc.get_freeze_count() == 0
    monkeypatch.setenv("ESR_LOOP_GC_FREEZE", "0")
    with quiet_gc():
        assert gc.get_freeze_count() == 0


def test_long_run_hint_words_are_unknown_once_their_slot_is_reused():
    """the long-run hint words live in a 16-slot ring and a kernel writes its generation only when it FINDS a
    long run.  A handle whose slot a later hint has been handed must answer "unknown" (-1), never "no long run" (0)."""
    import torch
    from esrecsys_amd.wikipedia import train_cooccurence as tc
    dev = torch.device("cpu")
    key = (dev.type, dev.index)
    tc._hint_ring[key] = [torch.zeros(16, dtype=torch.int32), 0, 1]  # (the product ring is pinned host memory)
    try:
        first = tc._hint_slot(dev)
        assert tc.hint_value(first) == 0           # nothing written: no long run
        first[0][0] = first[1]                     # what the hint kernel stores when it finds one
        assert tc.hint_value(first) == 1
        others = [tc._hint_slot(dev) for _ in range(15)]
        assert tc.hint_value(first) == 1 and all(tc.hint_value(h) == 0 for h in others)
        again = tc._hint_slot(dev)                 # the ring has come round: `first`'s word now belongs to `again`
        assert again[0].data_ptr() == first[0].data_ptr()
        assert tc.hint_value(first) == -1
        assert tc.hint_value(again) == 0           # the stale generation in the word is not `again`'s
        again[0][0] = again[1]
        assert tc.hint_value(again) == 1 and tc.hint_value(first) == -1
    finally:
        tc._hint_ring.pop(key, None)
