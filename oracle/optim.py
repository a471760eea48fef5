"""Oracle: optimizers (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The reference uses ``optax.adam(lr)`` on every table
(wikipedia/train_cooccurence.py:171, pinterest/train_shop_the_look.py:175) via
``TrainState.apply_gradients`` (train_cooccurence.py:101).  optax is not vendored;
the update rules below restate optax==0.1.2's published behaviour [upstream].
Sparse Adagrad is the build's production optimizer (north_star); it is
row-for-row identical to dense optax.adagrad because a zero gradient leaves both
the accumulator and the parameter unchanged.
"""
import numpy as np


def adam_init(param):
    return {"count": 0, "mu": np.zeros_like(param), "nu": np.zeros_like(param)}


def adam_update(param, grad, state, lr, b1=0.9, b2=0.999, eps=1e-8, dtype=None):
    """optax.adam == chain(scale_by_adam(b1, b2, eps, eps_root=0), scale(-lr)) [upstream optax 0.1.2].

    mu = b1 mu + (1-b1) g ; nu = b2 nu + (1-b2) g^2 ; t += 1
    p -= lr * (mu / (1 - b1^t)) / (sqrt(nu / (1 - b2^t)) + eps)
    Applied to EVERY element (rows with g = 0 still move while mu decays).
    """
    dtype = dtype or param.dtype.type
    g = grad.astype(dtype)
    mu = dtype(b1) * state["mu"].astype(dtype) + dtype(1 - b1) * g
    nu = dtype(b2) * state["nu"].astype(dtype) + dtype(1 - b2) * g * g
    t = state["count"] + 1
    bc1 = dtype(1.0 - b1 ** t)
    bc2 = dtype(1.0 - b2 ** t)
    upd = (mu / bc1) / (np.sqrt(nu / bc2) + dtype(eps))
    new_param = param.astype(dtype) - dtype(lr) * upd
    return new_param, {"count": t, "mu": mu, "nu": nu}


def segment_sum_rows(ids, rows, dtype=None):
    """Sum per-occurrence gradient rows that share an id, in occurrence order.

    Returns (unique_ids ascending, summed_rows).  This is the deterministic
    order the HIP scatter uses (stable sort by id, then left-to-right sum).
    """
    dtype = dtype or rows.dtype.type
    ids = np.asarray(ids, np.int64)
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    srows = rows[order].astype(dtype)
    uniq, start = np.unique(sid, return_index=True)
    out = np.zeros((len(uniq),) + rows.shape[1:], dtype=dtype)
    bounds = list(start) + [len(sid)]
    for u in range(len(uniq)):
        acc = srows[bounds[u]].copy()
        for k in range(bounds[u] + 1, bounds[u + 1]):
            acc = acc + srows[k]
        out[u] = acc
    return uniq, out


def sparse_adagrad_update(param, accum, ids, rows, lr, eps=1e-7, dtype=None):
    """Row-sparse optax.adagrad [upstream optax 0.1.2: scale_by_rss(initial_accumulator_value=0.1, eps=1e-7)].

    For every distinct id: G = sum of its occurrence rows; acc += G^2;
    p -= lr * G * rsqrt(acc + eps)   (update is 0 where acc == 0).
    ``param``/``accum`` are updated copies; untouched rows are returned unchanged.
    """
    dtype = dtype or param.dtype.type
    uniq, g = segment_sum_rows(ids, rows, dtype)
    p = param.astype(dtype).copy()
    a = accum.astype(dtype).copy()
    acc = a[uniq] + g * g
    with np.errstate(divide="ignore"):
        inv = np.where(acc > 0, dtype(1.0) / np.sqrt(acc + dtype(eps)), dtype(0.0))
    p[uniq] = p[uniq] - dtype(lr) * g * inv
    a[uniq] = acc
    return p, a


def adagrad_init(param, initial_accumulator_value=0.1):
    return np.full_like(param, initial_accumulator_value)


def sgd_momentum_update(param, trace, grad, lr, momentum=0.9, dtype=None):
    """optax.sgd(lr, momentum) == chain(trace(decay=momentum), scale(-lr)) [upstream]:
    trace = g + momentum * trace ; p -= lr * trace.  (spotify/train_spotify.py:238-241, "next" row N1.)"""
    dtype = dtype or param.dtype.type
    tr = grad.astype(dtype) + dtype(momentum) * trace.astype(dtype)
    return param.astype(dtype) - dtype(lr) * tr, tr


def sparse_adagrad_update_inplace(param, accum, ids, rows, lr, eps=1e-7):
    """``sparse_adagrad_update`` for config-size tables: same rule, the (fp64) ``param`` / ``accum`` arrays are
    stepped IN PLACE on the distinct rows only and the per-id sums are one ``np.add.reduceat`` over the stably
    sorted occurrences (same left-to-right order per id as ``segment_sum_rows``; pairwise inside NumPy, which in
    fp64 is ~1e-16 of the sum).  Returns the distinct ids.  tests/test_oracle.py pins it to the plain function."""
    ids = np.asarray(ids, np.int64)
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    uniq, start = np.unique(sid, return_index=True)
    g = np.add.reduceat(np.asarray(rows, param.dtype)[order], start, axis=0)
    acc = accum[uniq] + g * g
    with np.errstate(divide="ignore"):
        inv = np.where(acc > 0, 1.0 / np.sqrt(acc + param.dtype.type(eps)), 0.0)
    param[uniq] -= param.dtype.type(lr) * g * inv
    accum[uniq] = acc
    return uniq


def round_bf16(x):
    """Round fp64 / fp32 values to the nearest bfloat16 (8 significant bits, ties to even), returned in x's dtype.

    What a bf16 table holds after a step computed at higher precision (BASELINE config 4's dtype: the build steps bf16
    rows in f32 and rounds once, on the row's store).  Directly from the input precision -- no double rounding through
    f32 -- and without a bfloat16 type: m 2^e = x with m in [0.5, 1) (frexp), m rounded to 8 bits (numpy rounds half to
    even).  bf16's subnormals (|x| < 2^-126) do not occur in embedding tables and are not modelled.
    """
    x = np.asarray(x)
    m, e = np.frexp(x)
    return np.ldexp(np.round(m * 256.0) / 256.0, e).astype(x.dtype)
