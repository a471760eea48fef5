"""Tensor-level wrappers over the C ABI (include/esr_hip.h).

PyTorch is used for device memory and the current HIP stream only; every computation is a
libesr_hip.so call.  All functions require CUDA(ROCm) tensors and raise if the library is missing.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import ESR_BF16, ESR_F32, GLOVE_DIAGONAL, GLOVE_REFERENCE, check  # noqa: F401


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream
    object through several Python layers (~2 us, four or more times per step on launch-bound steps); the raw getter
    is one C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _req(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/ROCm tensor (no CPU fallback exists)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _table_dtype(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/ROCm tensor (no CPU fallback exists)" % name)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if t.dtype == torch.float32:
        return ESR_F32
    if t.dtype == torch.bfloat16:
        return ESR_BF16
    raise TypeError("%s must be float32 or bfloat16, got %s" % (name, t.dtype))


def check_device_ids(ids, V):
    """Debug screen (ESR_CHECK_IDS=1): raise IndexError if any id of the device tensor is outside [0, V).
    One small launch + a host read-back, i.e. a sync -- which is why it is not on by default."""
    lib = _lib.load()
    report = torch.tensor([0, 2 ** 63 - 1], dtype=torch.int64, device=ids.device)
    check(lib.esr_check_ids(_p(ids), ids.numel(), int(V), _p(report), _stream()), "esr_check_ids")
    bad, first = (int(v) for v in report.cpu())
    if bad:
        raise IndexError("%d device-resident id(s) out of range [0, %d); first at flat position %d (value %d)"
                         % (bad, V, first, int(ids.reshape(-1)[first])))


def as_ids(x, device, check_range=None):
    """int ids (numpy / list / torch, any int dtype) -> contiguous int32 tensor on `device`.

    Host inputs are range-checked against `check_range` = V.  Device inputs are trusted (checking them forces a
    sync on the hot path) unless ESR_CHECK_IDS=1 is set in the environment: then every device id tensor is screened
    by esr_check_ids and an out-of-range id raises IndexError instead of becoming a wild read / RMW."""
    if isinstance(x, torch.Tensor):
        if x.is_cuda:
            # (an int32 contiguous device tensor -- what a training loop hands over -- passes through untouched: the two
            # no-op torch calls were ~3 us per tensor, a quarter of the host's time per step in the triplet loop at B = 8192)
            if x.dtype is not torch.int32 or not x.is_contiguous():
                x = x.to(torch.int32).contiguous()
            if check_range is not None and os.environ.get("ESR_CHECK_IDS") == "1":
                check_device_ids(x, check_range)
            return x
        x = x.numpy()
    a = np.ascontiguousarray(np.asarray(x), dtype=np.int32)
    if check_range is not None and a.size and (a.min() < 0 or a.max() >= check_range):
        raise IndexError("id out of range [0, %d): min %d max %d" % (check_range, a.min(), a.max()))
    return torch.from_numpy(a).to(device, non_blocking=True)


def as_f32(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).to(device, non_blocking=True)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


_ws_size_cache = {}


def _ws_bytes(fn_name, *args):
    key = (fn_name,) + args
    v = _ws_size_cache.get(key)
    if v is None:
        v = int(getattr(_lib.load(), fn_name)(*args))
        _ws_size_cache[key] = v
    return v


# ---------------------------------------------------------------------------------------------
def gather_rows(table, ids, out=None):
    """out[i] = table[ids[i]]  (bit-exact).  table [V, D] f32/bf16; ids int32 [n]."""
    lib = _lib.load()
    dt = _table_dtype(table, "table")
    ids = _req(ids, torch.int32, "ids")
    V, D = table.shape
    n = ids.numel()
    if out is None:
        out = torch.empty((n, D), dtype=table.dtype, device=table.device)
    check(lib.esr_gather_rows(_p(table), dt, V, D, _p(ids), n, _p(out), _stream()), "esr_gather_rows")
    return out


def unpermute_rows(rows, perm, out=None):
    """out[perm[k]] = rows[k]."""
    lib = _lib.load()
    dt = _table_dtype(rows, "rows")
    perm = _req(perm, torch.int32, "perm")
    n, D = rows.shape
    if out is None:
        out = torch.empty_like(rows)
    check(lib.esr_unpermute_rows(_p(rows), dt, D, _p(perm), n, _p(out), _stream()), "esr_unpermute_rows")
    return out


def unpermute_rows_to_f32(rows, perm, out=None):
    """out[perm[k]] = float32(rows[k]); rows may be bf16 (converted on the fly) or f32."""
    if rows.dtype == torch.float32:
        return rows if perm is None else unpermute_rows(rows, perm, out)
    lib = _lib.load()
    _req(rows, torch.bfloat16, "rows")
    n, D = rows.shape
    if out is None:
        out = torch.empty((n, D), dtype=torch.float32, device=rows.device)
    check(lib.esr_unpermute_rows_bf16_to_f32(_p(rows), D, _p(perm), n, _p(out), _stream()),
          "esr_unpermute_rows_bf16_to_f32")
    return out


def glove_forward(emb, bias, inputs):
    """(dot[B], s[B]) of Glove.__call__; the reference output is dot[None, :] + s[:, None]."""
    lib = _lib.load()
    _req(emb, torch.float32, "emb"), _req(bias, torch.float32, "bias"), _req(inputs, torch.int32, "inputs")
    V, D = emb.shape
    B = inputs.shape[1]
    dot = torch.empty(B, dtype=torch.float32, device=emb.device)
    s = torch.empty(B, dtype=torch.float32, device=emb.device)
    check(lib.esr_glove_forward(_p(emb), _p(bias), V, D, _p(inputs), B, _p(dot), _p(s), _stream()),
          "esr_glove_forward")
    return dot, s


def glove_fwd_bwd(emb, bias, inputs, target, mode=GLOVE_REFERENCE, want_grads=True, grads_at_ids=False):
    """Fused GloVe loss + per-occurrence gradients.  Returns (loss[1], grad_rows[2B, D], grad_bias[2B]).
    grads_at_ids: emb / bias are a [2B, .] copy of the looked-up rows (one row per occurrence, any order) and the
    gradient of the occurrence that read row r is written at row r."""
    lib = _lib.load()
    _req(emb, torch.float32, "emb"), _req(bias, torch.float32, "bias")
    _req(inputs, torch.int32, "inputs"), _req(target, torch.float32, "target")
    V, D = emb.shape
    B = inputs.shape[1]
    if inputs.shape[0] != 2 or target.numel() != B:
        raise ValueError("inputs must be [2, B] and target [B]")
    dev = emb.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    grad_rows = torch.empty((2 * B, D), dtype=torch.float32, device=dev) if want_grads else None
    grad_bias = torch.empty(2 * B, dtype=torch.float32, device=dev) if want_grads else None
    nb = _ws_bytes("esr_glove_workspace_bytes", B)
    ws = _ws(nb, dev)
    if grads_at_ids and want_grads:
        mode = mode | GRADS_AT_IDS
    check(lib.esr_glove_fwd_bwd(_p(emb), _p(bias), V, D, _p(inputs), _p(target), B, mode, _p(loss), _p(grad_rows),
                                _p(grad_bias), _p(ws), ws.numel(), _stream()), "esr_glove_fwd_bwd")
    return loss, grad_rows, grad_bias


GRADS_AT_IDS = 0x100  # include/esr_hip.h ESR_GRADS_AT_IDS


def glove_train_step(emb, shadow, loc, accum, bias, bias_accum, inputs, target, mode, lr, eps=1e-7, presorted=None,
                     blocks_per_cu=0, stamp=None, plan=None, long_runs=-1, start_flag=None, start_value=0):
    """One whole GloVe training step (loss + gradients + sparse Adagrad on both tables) without materialised
    gradients: esr_glove_train_step.  `emb` / `shadow` are the two buffers of the double-buffered embedding table and
    `loc` (uint8 [V]) holds each row's stamped location byte; all three are updated.  `stamp` (1 .. 127): this step's
    stamp on that table (train_state.next_stamp hands them out and clears the bytes' stamps when the counter wraps).
    presorted = (sorted_ids, perm) of inputs.reshape(-1) from segment_sort (computed ahead), else the sort runs here;
    plan = this batch's glove_plan record (needs presorted), long_runs = 0 when its hint said no run is long.
    start_flag (int32 [1] device tensor) / start_value: the update kernel announces its start there (stream_gate).
    Returns loss[1]."""
    lib = _lib.load()
    dt = _table_dtype(emb, "emb")  # f32, or bf16 rows (both buffers) with fp32 accumulator and bias tables
    if _table_dtype(shadow, "shadow") != dt:
        raise TypeError("emb and shadow must have the same dtype")
    for name, t in (("accum", accum), ("bias", bias), ("bias_accum", bias_accum), ("target", target)):
        _req(t, torch.float32, name)
    _req(loc, torch.uint8, "loc"), _req(inputs, torch.int32, "inputs")
    if stamp is None:
        raise ValueError("glove_train_step needs the step's stamp (train_state.next_stamp(row_versions))")
    V, D = emb.shape
    B = inputs.shape[1]
    if inputs.shape[0] != 2 or target.numel() != B:
        raise ValueError("inputs must be [2, B] and target [B]")
    if shadow.shape != emb.shape or accum.shape != emb.shape or loc.numel() != V or bias.numel() != V or \
            bias_accum.numel() != V:
        raise ValueError("shadow / accum / loc / bias shapes do not match the embedding table")
    loss = torch.empty(1, dtype=torch.float32, device=emb.device)
    ws = _ws(_ws_bytes("esr_glove_step_workspace_bytes", B, D), emb.device)
    sid = perm = None
    if presorted is not None:
        sid, perm = _req(presorted[0], torch.int32, "sorted_ids"), _req(presorted[1], torch.int32, "perm")
        if sid.numel() != 2 * B or perm.numel() != 2 * B:
            raise ValueError("presorted ids / perm must have 2 B entries")
    if plan is not None and (plan.dtype != torch.uint8 or plan.numel() < _ws_bytes("esr_glove_plan_bytes", B)):
        raise ValueError("plan must be the uint8 record glove_plan made for a batch of this size")
    check(lib.esr_glove_train_step(_p(emb), _p(shadow), _p(loc), _p(accum), _p(bias), _p(bias_accum), V, dt, D, _p(inputs),
                                   _p(target), B, mode, float(lr), float(eps), int(stamp), _p(sid), _p(perm), _p(plan),
                                   int(long_runs), int(blocks_per_cu), _p(start_flag), int(start_value) & 0xFFFFFFFF,
                                   _p(loss), _p(ws), ws.numel(), _stream()),
          "esr_glove_train_step")
    return loss


def stream_gate(flag, value, timeout_us=1_000_000):
    """Hold the CURRENT stream (one sleeping wave) until the device word flag[0] has reached `value` (sequence numbers,
    wrap-safe) or timeout_us have passed: esr_stream_gate."""
    lib = _lib.load()
    if not (flag.is_cuda and flag.dtype == torch.int32 and flag.numel() >= 1):
        raise ValueError("stream_gate: flag must be an int32 device tensor")
    check(lib.esr_stream_gate(_p(flag), int(value) & 0xFFFFFFFF, int(timeout_us), _stream()), "esr_stream_gate")


def _aligned_bytes(nbytes, device, align=256):
    """ZEROED uint8 tensor of nbytes whose data pointer is `align`-aligned (torch's allocator hands out 512-byte blocks).
    Plan buffers come from here: the direct triplet plan's long-run counter is tagged with the plan call's generation
    (esr_triplet_step.hip) -- a fresh buffer must not hold a word that happens to carry a live tag."""
    t = torch.zeros(nbytes + align, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % align
    return t[off:off + nbytes]


def glove_plan(inputs_list, targets_list, sorted_ids, perm, hints=None, gen=0):
    """Plan records of up to eight coming GloVe batches by one launch (esr_glove_plan): inputs_list[b] int32 [2, B],
    targets_list[b] f32 [B], sorted_ids / perm int32 [nb, 2B].  Returns a uint8 [nb, plan_bytes] tensor (row b feeds
    exactly one glove_train_step).  hints: optional int32 [nb] device tensor that receives `gen` where list b has a run
    longer than a chunk."""
    lib = _lib.load()
    nb = len(inputs_list)
    B = inputs_list[0].shape[1]
    for i, t in zip(inputs_list, targets_list):
        _req(i, torch.int32, "inputs"), _req(t, torch.float32, "target")
        if i.shape != (2, B) or t.numel() != B:
            raise ValueError("every batch of a plan group must be [2, B] / [B] with the same B")
    _req(sorted_ids, torch.int32, "sorted_ids"), _req(perm, torch.int32, "perm")
    if sorted_ids.numel() != nb * 2 * B or perm.numel() != nb * 2 * B:
        raise ValueError("sorted_ids / perm must be [nb, 2B]")
    pb = _ws_bytes("esr_glove_plan_bytes", B)
    plans = _aligned_bytes(nb * pb, inputs_list[0].device).view(nb, pb)
    ip = (ctypes.c_void_p * nb)(*[i.data_ptr() for i in inputs_list])
    tp = (ctypes.c_void_p * nb)(*[t.data_ptr() for t in targets_list])
    check(lib.esr_glove_plan(ip, tp, nb, B, _p(sorted_ids), _p(perm), _p(plans), _p(hints), int(gen), _stream()),
          "esr_glove_plan")
    return plans


def unique_by_owner(id_tensors, world, local_rows, offsets=None):
    """The distinct rows of an occurrence list, routed by owner (esr_unique_by_owner): id_tensors = int32 segments (or one
    tensor), offsets = their virtual-row offsets.  Returns (ulocal [n] -- first sum(ucounts) entries valid --, ucounts
    int64 [world] on the device, uidx [n], sorted_uidx [n], perm [n])."""
    lib = _lib.load()
    if isinstance(id_tensors, torch.Tensor):
        id_tensors = [id_tensors.reshape(-1)]
    offsets = list(offsets) if offsets is not None else [0] * len(id_tensors)
    for t in id_tensors:
        _req(t, torch.int32, "ids")
    dev = id_tensors[0].device
    k = len(id_tensors)
    counts = [int(t.numel()) for t in id_tensors]
    n = sum(counts)
    ulocal = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    uidx, sorted_uidx, perm = torch.empty_like(ulocal), torch.empty_like(ulocal), torch.empty_like(ulocal)
    ucounts = torch.empty(world, dtype=torch.int64, device=dev)
    ws = _ws(_ws_bytes("esr_unique_by_owner_workspace_bytes", n), dev)
    ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in id_tensors])
    cnt = (ctypes.c_int64 * k)(*counts)
    off = (ctypes.c_int64 * k)(*[int(x) for x in offsets])
    check(lib.esr_unique_by_owner(ptrs, cnt, off, k, int(world), int(local_rows), _p(ulocal), _p(uidx), _p(sorted_uidx),
                                  _p(perm), _p(ucounts), _p(ws), ws.numel(), _stream()), "esr_unique_by_owner")
    return ulocal[:n], ucounts, uidx[:n], sorted_uidx[:n], perm[:n]


def segment_sum_rows(rows_out, sorted_ids, perm, grad_rows):
    """[rows_out, D]: row u = the sum of grad_rows[perm[p]] over the positions p with sorted_ids[p] == u (every u in
    [0, rows_out) must occur: the distinct-row index of unique_by_owner).  Overwrites grad_rows."""
    lib = _lib.load()
    _req(sorted_ids, torch.int32, "sorted_ids"), _req(perm, torch.int32, "perm"), _req(grad_rows, torch.float32, "grad_rows")
    grad_rows = grad_rows.reshape(grad_rows.shape[0], -1)
    D = grad_rows.shape[1]
    out = torch.empty((int(rows_out), D), dtype=torch.float32, device=grad_rows.device)
    check(lib.esr_segment_sum_rows(_p(out), int(rows_out), D, _p(sorted_ids), _p(perm), sorted_ids.numel(), _p(grad_rows),
                                   _stream()), "esr_segment_sum_rows")
    return out


def topk_columns(scores, k):
    """(values [T, k], rows [T, k]) of the k largest entries of every column of scores [V, T], best first, ties as the
    stable ascending argsort orders them (the higher row counts as larger): esr_topk_columns, k <= 1024."""
    lib = _lib.load()
    _req(scores, torch.float32, "scores")
    V, T_ = scores.shape
    out_s = torch.empty((T_, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((T_, k), dtype=torch.int32, device=scores.device)
    check(lib.esr_topk_columns(_p(scores), V, T_, int(k), _p(out_s), _p(out_i), _stream()), "esr_topk_columns")
    return out_s, out_i


def long_run_hint(sorted_ids, chunk, hint, gen):
    """hint[0] = gen when sorted_ids has a run of equal ids longer than `chunk` positions (esr_long_run_hint); `hint` is
    an int32 tensor on the device or in pinned host memory."""
    lib = _lib.load()
    _req(sorted_ids, torch.int32, "sorted_ids")
    if hint.dtype != torch.int32 or not (hint.is_cuda or hint.is_pinned()):
        raise TypeError("hint must be an int32 tensor on the device or in pinned host memory")
    check(lib.esr_long_run_hint(_p(sorted_ids), sorted_ids.numel(), int(chunk), hint.data_ptr(), int(gen), _stream()),
          "esr_long_run_hint")


def triplet_plan(id_lists, Vs, sorted_ids, perm, hints=None, gen=0):
    """Plan records of up to eight coming triplet batches by one launch (esr_triplet_plan): id_lists[b] = (scene, pos,
    neg) int32 [B] each, sorted_ids / perm int32 [nb, 3B].  Returns a uint8 [nb, plan_bytes] tensor."""
    lib = _lib.load()
    nb = len(id_lists)
    B = id_lists[0][0].numel()
    flat = []
    for trip in id_lists:
        for t in trip:
            _req(t, torch.int32, "ids")
            if t.numel() != B:
                raise ValueError("every id list of a plan group must have B entries")
            flat.append(t.data_ptr())
    _req(sorted_ids, torch.int32, "sorted_ids"), _req(perm, torch.int32, "perm")
    if sorted_ids.numel() != nb * 3 * B or perm.numel() != nb * 3 * B:
        raise ValueError("sorted_ids / perm must be [nb, 3B]")
    pb = _ws_bytes("esr_triplet_plan_bytes", B)
    plans = _aligned_bytes(nb * pb, id_lists[0][0].device).view(nb, pb)
    ptrs = (ctypes.c_void_p * (3 * nb))(*flat)
    check(lib.esr_triplet_plan(ptrs, nb, B, int(Vs), _p(sorted_ids), _p(perm), _p(plans), _p(hints), int(gen),
                               _stream()), "esr_triplet_plan")
    return plans


def triplet_direct_mode():
    """True (default) when the one-pass triplet step walks the TRIPLETS and steps rows in place (esr_triplet_step.hip,
    "DIRECT mode"); ESR_TRIPLET_STEP=stamped: the double-buffered walk over the sorted occurrences of rounds 2-3."""
    return not os.environ.get("ESR_TRIPLET_STEP", "direct").startswith("s")


def triplet_train_step(scene, scene_shadow, scene_loc, scene_accum, product, product_shadow, product_loc,
                       product_accum, scene_ids, pos_ids, neg_ids, regularization, batch_size, lr, eps=1e-7,
                       presorted=None, stamp=None, plan=None, long_runs=-1):
    """One whole Shop-The-Look training step (triplet loss + on-chip gradients + sparse Adagrad on both double-buffered
    towers): esr_triplet_train_step.  `stamp`: this step's stamp on both towers (train_state.next_stamp).  presorted =
    (sorted virtual ids, perm) of [scene ; Vs + pos ; Vs + neg] from segment_sort_multi, else the sort runs here; plan /
    long_runs as for glove_train_step (triplet_plan).  Returns loss[1]."""
    lib = _lib.load()
    direct = triplet_direct_mode()  # rows stepped in place: no second buffers, location bytes or stamp
    dt = _table_dtype(scene, "scene")   # f32, or bf16 rows with fp32 accumulators (direct mode only)
    if _table_dtype(product, "product") != dt:
        raise TypeError("scene and product towers must have the same dtype")
    if dt != ESR_F32 and not direct:
        raise TypeError("bf16 towers need the direct step (ESR_TRIPLET_STEP=stamped is set)")
    for name, t in (("scene_accum", scene_accum), ("product_accum", product_accum)):
        _req(t, torch.float32, name)
    if not direct or scene_shadow is not None:
        _req(scene_shadow, torch.float32, "scene_shadow"), _req(product_shadow, torch.float32, "product_shadow")
        _req(scene_loc, torch.uint8, "scene_loc"), _req(product_loc, torch.uint8, "product_loc")
    for name, t in (("scene_ids", scene_ids), ("pos_ids", pos_ids), ("neg_ids", neg_ids)):
        _req(t, torch.int32, name)
    if stamp is None:
        if not direct:
            raise ValueError("triplet_train_step needs the step's stamp (train_state.next_stamp(row_versions...))")
        stamp = 1
    Vs, D = scene.shape
    Vp = product.shape[0]
    B = scene_ids.numel()
    if product.shape[1] != D or pos_ids.numel() != B or neg_ids.numel() != B:
        raise ValueError("tower dims / id counts differ")
    if scene_accum.shape != scene.shape or product_accum.shape != product.shape:
        raise ValueError("accumulator shapes do not match their towers")
    if scene_shadow is not None and (scene_shadow.shape != scene.shape or scene_loc.numel() != Vs or
                                     product_shadow.shape != product.shape or product_loc.numel() != Vp):
        raise ValueError("shadow / loc shapes do not match their towers")
    sid = perm = None
    if presorted is not None:
        sid, perm = _req(presorted[0], torch.int32, "sorted_ids"), _req(presorted[1], torch.int32, "perm")
        if sid.numel() != 3 * B or perm.numel() != 3 * B:
            raise ValueError("presorted ids / perm must have 3 B entries")
    if plan is not None and (plan.dtype != torch.uint8 or plan.numel() < _ws_bytes("esr_triplet_plan_bytes", B)):
        raise ValueError("plan must be the uint8 record triplet_plan made for a batch of this size")
    loss = torch.empty(1, dtype=torch.float32, device=scene.device)
    ws = _ws(_ws_bytes("esr_triplet_step_workspace_bytes", B, D), scene.device)
    check(lib.esr_triplet_train_step(_p(scene), _p(scene_shadow), _p(scene_loc), _p(scene_accum), Vs, _p(product),
                                     _p(product_shadow), _p(product_loc), _p(product_accum), Vp, dt, D, _p(scene_ids),
                                     _p(pos_ids), _p(neg_ids), B, float(regularization), float(batch_size), float(lr),
                                     float(eps), int(stamp), _p(sid), _p(perm), _p(plan), int(long_runs), _p(loss),
                                     _p(ws), ws.numel(), _stream()),
          "esr_triplet_train_step")
    return loss


def rows_consolidate(primary, shadow, loc):
    """Copy the rows of a double-buffered table whose current value lives in `shadow` (loc[row] == 1) back into
    `primary` and clear their bytes: afterwards `primary` is the plain [V, D] table."""
    lib = _lib.load()
    dt = _table_dtype(primary, "primary")
    if _table_dtype(shadow, "shadow") != dt:
        raise TypeError("primary and shadow must have the same dtype")
    _req(loc, torch.uint8, "loc")
    V, D = primary.shape
    check(lib.esr_rows_consolidate(_p(primary), _p(shadow), _p(loc), V, dt, D, _stream()), "esr_rows_consolidate")


def rows_restamp(loc):
    """Forget the step stamps of a double-buffered table's location bytes (bit 0, the location, stays): run when the
    caller's stamp counter wraps (esr_rows_restamp; train_state.next_stamp does)."""
    lib = _lib.load()
    _req(loc, torch.uint8, "loc")
    check(lib.esr_rows_restamp(_p(loc), loc.numel(), _stream()), "esr_rows_restamp")


def triplet_fwd_bwd(scene_table, pos_table, neg_table, scene_ids, pos_ids, neg_ids, B, regularization, batch_size,
                    with_reg=True, want_grads=True, want_scores=True, grads_at_ids=False):
    """Fused STL head.  ids may be None (= row b).  Returns (loss[1], pos_score, neg_score, g_s, g_p, g_n);
    the three gradients are consecutive slices of one [3B, D] buffer (``g_s._base``).
    grads_at_ids: the three "tables" are ONE [3B, D] copy of the looked-up rows (one row per occurrence, any order)
    and every gradient row is written at the row its table row came from: returns g_s = that [3B, D] buffer,
    g_p = g_n = None."""
    lib = _lib.load()
    for name, t in (("scene_table", scene_table), ("pos_table", pos_table), ("neg_table", neg_table)):
        _req(t, torch.float32, name)
    for name, t in (("scene_ids", scene_ids), ("pos_ids", pos_ids), ("neg_ids", neg_ids)):
        if t is not None:
            _req(t, torch.int32, name)
    Vs, D = scene_table.shape
    Vp, Vn = pos_table.shape[0], neg_table.shape[0]
    if pos_table.shape[1] != D or neg_table.shape[1] != D:
        raise ValueError("tower dims differ")
    dev = scene_table.device
    f32 = dict(dtype=torch.float32, device=dev)
    loss = torch.empty(1, **f32)
    ps = torch.empty(B, **f32) if want_scores else None
    ns = torch.empty(B, **f32) if want_scores else None
    # scene / pos / neg gradient rows share one [3B, D] buffer ([scene ; pos ; neg]): the optimizer consumes
    # them as a single occurrence list (``g_s._base``; ``g_s._base[B:]`` = the product rows) with no copy.
    gall = torch.empty((3 * B, D), **f32) if want_grads else None
    flags = int(bool(with_reg))
    if grads_at_ids and want_grads:
        flags |= GRADS_AT_IDS
        gs = gp = gn = gall
    else:
        gs = gall[:B] if want_grads else None
        gp = gall[B:2 * B] if want_grads else None
        gn = gall[2 * B:] if want_grads else None
    ws = _ws(_ws_bytes("esr_triplet_workspace_bytes", B), dev)
    check(lib.esr_triplet_fwd_bwd(_p(scene_table), Vs, _p(pos_table), Vp, _p(neg_table), Vn, D, _p(scene_ids),
                                  _p(pos_ids), _p(neg_ids), B, float(regularization), float(batch_size),
                                  flags, _p(loss), _p(ps), _p(ns), _p(gs), _p(gp), _p(gn), _p(ws),
                                  ws.numel(), _stream()), "esr_triplet_fwd_bwd")
    if grads_at_ids and want_grads:
        return loss, ps, ns, gall, None, None
    return loss, ps, ns, gs, gp, gn


INBATCH_PRECISIONS = ("auto", "f32", "bf16x3", "f16x2")
INBATCH_F16X2_MAX_B = 16384  # esr_inbatch2h.hip keeps the B x B probabilities between its two passes


def _inbatch_auto_split():
    """What "auto" resolves to where both split-precision paths apply (ESR_INBATCH_AUTO=bf16x3 keeps the older one)."""
    import os
    v = os.environ.get("ESR_INBATCH_AUTO", "f16x2")
    if v not in ("f16x2", "bf16x3"):
        raise ValueError("ESR_INBATCH_AUTO must be f16x2 or bf16x3")
    return v


def inbatch_split_path(precision, B, D, bf16_tables=False):
    """The split-precision MFMA path `precision` selects for a [B, D] in-batch head: "f16x2", "bf16x3" or None (exact
    f32).  f16x2: two fp16 planes per operand, three MFMAs per product (esr_inbatch2h.hip; fp32 rows, B <= 16384);
    bf16x3: three bf16 planes, six MFMAs (esr_inbatch3.hip; also bf16 tables, where only one plane is live).  Both work
    on 128-column tiles: D <= 128, a multiple of 4 (narrower rows are zero-padded); "auto" takes them from D = 64 up
    (ESR_INBATCH_SPLIT_MIN_D), the exact-f32 kernel's own 32- / 64-column tiles below."""
    if precision not in INBATCH_PRECISIONS:
        raise ValueError("precision must be one of %s" % (INBATCH_PRECISIONS,))
    shape_ok = D % 4 == 0 and 0 < D <= 128 and B % 128 == 0 and B > 0
    # bf16 tables (round 5): the fp16 entry points run their ONE-plane kernels on them (a bf16 element is exact in one
    # fp16 plane of x * 2^e): six GEMMs with S^T recomputed by pass C, against eight on the bf16 x 3 one-plane kernels.
    # ESR_INBATCH_BF16_TABLES=bf16x3 keeps the older path.
    h_ok = shape_ok and B <= INBATCH_F16X2_MAX_B and \
        (not bf16_tables or os.environ.get("ESR_INBATCH_BF16_TABLES", "f16") != "bf16x3")
    if precision == "f32":
        return None
    if precision == "bf16x3":
        if not shape_ok:
            raise ValueError("precision='bf16x3' needs D <= 128 (a multiple of 4) and B %% 128 == 0 (got B=%d, D=%d)"
                             % (B, D))
        return "bf16x3"
    if precision == "f16x2":
        if not h_ok:
            raise ValueError("precision='f16x2' needs D <= 128 (a multiple of 4), B %% 128 == 0 and B <= %d "
                             "(got B=%d, D=%d)" % (INBATCH_F16X2_MAX_B, B, D))
        return "f16x2"
    if not shape_ok or D < int(os.environ.get("ESR_INBATCH_SPLIT_MIN_D", "64")):
        return None
    return "f16x2" if h_ok and _inbatch_auto_split() == "f16x2" else "bf16x3"


def inbatch_softmax_fwd_bwd(Q, C, scale, regularization, batch_size, precision="auto", pass_c_forms=None):
    """In-batch-negative softmax on the matrix cores.  Returns (loss[1], lse[B], gQ, gC).

    precision "f32": exact-f32 MFMA (v_mfma_f32_32x32x2_f32).  "bf16x3": f32-equivalent products from three
    exact bf16 planes per operand (six v_mfma_f32_32x32x16_bf16 per block, error <= 2^-23 per product; needs
    D == 128 and B % 128 == 0).  "f16x2": two scaled fp16 planes per operand (three v_mfma_f32_32x32x16_f16 per block,
    error <= ~2^-22 per product; also B <= 16384).  "auto": f16x2 where it applies, else bf16x3, else f32.  All hold
    the 1e-5 bound."""
    lib = _lib.load()
    _req(Q, torch.float32, "Q"), _req(C, torch.float32, "C")
    B, D = Q.shape
    if C.shape != Q.shape:
        raise ValueError("Q and C must have the same shape")
    path = inbatch_split_path(precision, B, D)
    dev = Q.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    lse = torch.empty(B, dtype=torch.float32, device=dev)
    gQC = torch.empty((2 * B, D), dtype=torch.float32, device=dev)  # [gQ ; gC]: one occurrence list downstream
    gQ, gC = gQC[:B], gQC[B:]
    if path == "f16x2":
        ws = _ws(_ws_bytes("esr_inbatch2h_workspace_bytes", B, D), dev)
        fn, name = lib.esr_inbatch_softmax_fwd_bwd_f16x2, "esr_inbatch_softmax_fwd_bwd_f16x2"
    elif path == "bf16x3":
        ws = _ws(_ws_bytes("esr_inbatch3_workspace_bytes", B, D), dev)
        fn, name = lib.esr_inbatch_softmax_fwd_bwd_bf16x3, "esr_inbatch_softmax_fwd_bwd_bf16x3"
    else:
        ws = _ws(_ws_bytes("esr_inbatch_workspace_bytes", B, D), dev)
        fn, name = lib.esr_inbatch_softmax_fwd_bwd, "esr_inbatch_softmax_fwd_bwd"
    check(fn(_p(Q), _p(C), B, D, float(scale), float(regularization), float(batch_size), _p(loss), _p(lse), _p(gQ),
             _p(gC), _p(ws), ws.numel(), _stream()), name)
    if pass_c_forms is not None:  # diagnostics: which form pass C took per pass-Q split (include/esr_hip.h)
        if path != "f16x2":
            raise ValueError("pass_c_forms: only the f16x2 path has them (this call took %r)" % path)
        _req(pass_c_forms, torch.int32, "pass_c_forms")
        check(lib.esr_inbatch2h_pass_c_forms(_p(ws), ws.numel(), B, _p(pass_c_forms), _stream()),
              "esr_inbatch2h_pass_c_forms")
    return loss, lse, gQ, gC


# ---------------------------------------------------------------------------------------------
def segment_sort(ids, V, out=None):
    """Stable sort of occurrence ids.  Returns (sorted_ids, perm) with sorted_ids == ids[perm]; `out` = a pair of int32
    tensors of ids' size to sort into (a caller that recycles its buffers)."""
    lib = _lib.load()
    ids = _req(ids, torch.int32, "ids")
    n = ids.numel()
    if out is not None:
        sorted_ids, perm = out
        _req(sorted_ids, torch.int32, "out[0]"), _req(perm, torch.int32, "out[1]")
        if sorted_ids.numel() != n or perm.numel() != n:
            raise ValueError("segment_sort: out tensors must have ids' size")
    else:
        sorted_ids = torch.empty_like(ids)
        perm = torch.empty_like(ids)
    ws = _ws(_ws_bytes("esr_segment_sort_workspace_bytes", n), ids.device)
    check(lib.esr_segment_sort_ids(_p(ids), n, V, _p(sorted_ids), _p(perm), _p(ws), ws.numel(), _stream()),
          "esr_segment_sort_ids")
    return sorted_ids, perm


def segment_sort_multi(id_tensors, offsets, V):
    """Stable sort of the virtual list [ids_0 + offsets[0] ; ids_1 + offsets[1] ; ...] without materialising it.
    Returns (sorted virtual ids, perm)."""
    import ctypes
    lib = _lib.load()
    k = len(id_tensors)
    for t in id_tensors:
        _req(t, torch.int32, "ids")
    counts = [int(t.numel()) for t in id_tensors]
    n, dev = sum(counts), id_tensors[0].device
    sorted_ids = torch.empty(n, dtype=torch.int32, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    ws = _ws(_ws_bytes("esr_segment_sort_workspace_bytes", n), dev)
    ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in id_tensors])
    cnt = (ctypes.c_int64 * k)(*counts)
    off = (ctypes.c_int64 * k)(*[int(x) for x in offsets])
    check(lib.esr_segment_sort_ids_multi(ptrs, cnt, off, k, V, _p(sorted_ids), _p(perm), _p(ws), ws.numel(), _stream()),
          "esr_segment_sort_ids_multi")
    return sorted_ids, perm


def sparse_adagrad(table, accum, sorted_ids, perm, grad_rows, lr, eps=1e-7):
    """In-place row-sparse Adagrad on `table` / `accum` for the rows named by sorted_ids."""
    lib = _lib.load()
    dt = _table_dtype(table, "table")
    _req(accum, torch.float32, "accum"), _req(grad_rows, torch.float32, "grad_rows")
    V = table.shape[0]
    D = table.shape[1] if table.dim() > 1 else 1
    n = sorted_ids.numel()
    if grad_rows.numel() != n * D or accum.numel() != table.numel():
        raise ValueError("shape mismatch: grad_rows %s, table %s, accum %s, n %d" %
                         (tuple(grad_rows.shape), tuple(table.shape), tuple(accum.shape), n))
    check(lib.esr_sparse_adagrad_scatter(_p(table), dt, _p(accum), V, D, _p(sorted_ids), _p(perm), n, _p(grad_rows),
                                         float(lr), float(eps), _stream()), "esr_sparse_adagrad_scatter")


def concat_offset_ids(id_tensors, offsets):
    """[ids_0 + offsets[0] ; ids_1 + offsets[1] ; ...] as one int32 tensor (virtual rows of concatenated tables)."""
    import ctypes
    lib = _lib.load()
    n = len(id_tensors)
    for t in id_tensors:
        _req(t, torch.int32, "ids")
    counts = [int(t.numel()) for t in id_tensors]
    out = torch.empty(sum(counts), dtype=torch.int32, device=id_tensors[0].device)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in id_tensors])
    cnt = (ctypes.c_int64 * n)(*counts)
    off = (ctypes.c_int64 * n)(*[int(o) for o in offsets])
    check(lib.esr_concat_offset_ids(ptrs, cnt, off, n, _p(out), _stream()), "esr_concat_offset_ids")
    return out


def gather_rows_multi(tables, row_offsets, vids, out=None):
    """out[i] = tables[t][vids[i] - row_offsets[t]] for virtual rows of several same-width tables."""
    import ctypes
    lib = _lib.load()
    n_t = len(tables)
    dts = {_table_dtype(t, "table") for t in tables}
    if len(dts) != 1:
        raise TypeError("fused tables must share a dtype")
    D = tables[0].shape[1] if tables[0].dim() > 1 else 1
    vids = _req(vids, torch.int32, "vids")
    n = vids.numel()
    if out is None:
        out = torch.empty((n, D), dtype=tables[0].dtype, device=vids.device)
    tp = (ctypes.c_void_p * n_t)(*[t.data_ptr() for t in tables])
    ro = (ctypes.c_int64 * (n_t + 1))(*[int(o) for o in row_offsets])
    check(lib.esr_gather_rows_multi(tp, ro, n_t, dts.pop(), D, _p(vids), n, _p(out), _stream()),
          "esr_gather_rows_multi")
    return out


def sparse_adagrad_multi(tables, accums, row_offsets, sorted_vids, perm, grad_rows, lr, eps=1e-7, long_runs=-1):
    """One launch of row-sparse Adagrad over several same-width tables addressed by virtual rows.  long_runs = 0: the
    caller knows (long_run_hint with chunk 32) that no id has a run the head chunk cannot hold -- the long-run launch
    is skipped."""
    import ctypes
    lib = _lib.load()
    n_t = len(tables)
    dts = {_table_dtype(t, "table") for t in tables}
    if len(dts) != 1:
        raise TypeError("fused tables must share a dtype")
    D = tables[0].shape[1] if tables[0].dim() > 1 else 1
    for t, a in zip(tables, accums):
        _req(a, torch.float32, "accum")
        if (t.shape[1] if t.dim() > 1 else 1) != D or a.numel() != t.numel():
            raise ValueError("fused tables must share D and have matching accumulators")
    _req(grad_rows, torch.float32, "grad_rows")
    n = sorted_vids.numel()
    tp = (ctypes.c_void_p * n_t)(*[t.data_ptr() for t in tables])
    ap = (ctypes.c_void_p * n_t)(*[a.data_ptr() for a in accums])
    ro = (ctypes.c_int64 * (n_t + 1))(*[int(o) for o in row_offsets])
    check(lib.esr_sparse_adagrad_scatter_multi(tp, ap, ro, n_t, dts.pop(), D, _p(sorted_vids), _p(perm), n,
                                               _p(grad_rows), float(lr), float(eps), int(long_runs), _stream()),
          "esr_sparse_adagrad_scatter_multi")


def sparse_sgd(table, sorted_ids, perm, grad_rows, lr):
    lib = _lib.load()
    dt = _table_dtype(table, "table")
    _req(grad_rows, torch.float32, "grad_rows")
    V = table.shape[0]
    D = table.shape[1] if table.dim() > 1 else 1
    n = sorted_ids.numel()
    check(lib.esr_sparse_sgd_scatter(_p(table), dt, V, D, _p(sorted_ids), _p(perm), n, _p(grad_rows), float(lr),
                                     _stream()), "esr_sparse_sgd_scatter")


def dense_momentum_decay(param, trace, lr, momentum):
    """In place over the whole table: trace *= momentum ; param -= lr * trace (the decay half of optax.sgd)."""
    lib = _lib.load()
    _req(param, torch.float32, "param"), _req(trace, torch.float32, "trace")
    check(lib.esr_dense_momentum_decay(_p(param), _p(trace), param.numel(), float(lr), float(momentum), _stream()),
          "esr_dense_momentum_decay")


def sparse_momentum(table, trace, sorted_ids, perm, grad_rows, lr):
    """In place on the touched rows: trace[row] += g ; table[row] -= lr * g (duplicates summed first)."""
    lib = _lib.load()
    _req(table, torch.float32, "table"), _req(trace, torch.float32, "trace"), _req(grad_rows, torch.float32, "grad_rows")
    V = table.shape[0]
    D = table.shape[1] if table.dim() > 1 else 1
    check(lib.esr_sparse_momentum_scatter(_p(table), _p(trace), V, D, _p(sorted_ids), _p(perm), sorted_ids.numel(),
                                          _p(grad_rows), float(lr), _stream()), "esr_sparse_momentum_scatter")


def momentum_catchup_rows(table, trace, last, ids, modulus, step, lr, momentum):
    """Lazy optax.sgd(lr, momentum): bring the rows ids % modulus (modulus 0: ids) up to step - 1 and mark them `step`."""
    lib = _lib.load()
    _req(table, torch.float32, "table"), _req(trace, torch.float32, "trace"), _req(last, torch.int32, "last")
    ids = _req(ids, torch.int32, "ids")
    V, D = table.shape
    check(lib.esr_momentum_catchup_rows(_p(table), _p(trace), _p(last), V, D, _p(ids), ids.numel(), int(modulus), int(step),
                                        float(lr), float(momentum), _stream()), "esr_momentum_catchup_rows")


def sparse_momentum_step(table, trace, sorted_ids, perm, grad_rows, lr, momentum):
    """In place on the touched rows: trace = g + momentum * trace ; table -= lr * trace (duplicates summed first)."""
    lib = _lib.load()
    _req(table, torch.float32, "table"), _req(trace, torch.float32, "trace"), _req(grad_rows, torch.float32, "grad_rows")
    V, D = table.shape
    check(lib.esr_sparse_momentum_step(_p(table), _p(trace), V, D, _p(sorted_ids), _p(perm), sorted_ids.numel(),
                                       _p(grad_rows), float(lr), float(momentum), _stream()), "esr_sparse_momentum_step")


def momentum_flush(table, trace, last, step, lr, momentum):
    """Bring every row of a lazily updated table up to `step`."""
    lib = _lib.load()
    _req(table, torch.float32, "table"), _req(trace, torch.float32, "trace"), _req(last, torch.int32, "last")
    V, D = table.shape
    check(lib.esr_momentum_flush(_p(table), _p(trace), _p(last), V, D, int(step), float(lr), float(momentum), _stream()),
          "esr_momentum_flush")


def spotify_get_embeddings(album_table, artist_table, album_ids, artist_ids):
    """[count, 2F] = concat(album_table[album mod rows], artist_table[artist]) (spotify/models.py:37-51)."""
    lib = _lib.load()
    _req(album_table, torch.float32, "album_table"), _req(artist_table, torch.float32, "artist_table")
    album_ids, artist_ids = _req(album_ids, torch.int32, "album_ids"), _req(artist_ids, torch.int32, "artist_ids")
    count, F = album_ids.numel(), album_table.shape[1]
    out = torch.empty((count, 2 * F), dtype=torch.float32, device=album_table.device)
    l2 = torch.empty(count, dtype=torch.float32, device=album_table.device)
    check(lib.esr_spotify_get_embeddings(_p(album_table), album_table.shape[0], _p(artist_table),
                                         artist_table.shape[0], F, _p(album_ids), _p(artist_ids), count, _p(out),
                                         _p(l2), _stream()), "esr_spotify_get_embeddings")
    return out


def spotify_forward(album_table, artist_table, album_ids, artist_ids, n, m, o):
    """SpotifyModel.__call__ (spotify/models.py:53-90) on occurrence ids ordered context, next, neg."""
    lib = _lib.load()
    _req(album_table, torch.float32, "album_table"), _req(artist_table, torch.float32, "artist_table")
    album_ids, artist_ids = _req(album_ids, torch.int32, "album_ids"), _req(artist_ids, torch.int32, "artist_ids")
    F, dev = album_table.shape[1], album_table.device
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    pos, neg, cs, ns, gs, l2 = f(m), f(o), f(n, n), f(m, m), f(o, o), f(n + m + o)
    ws = _ws(_ws_bytes("esr_spotify_workspace_bytes", n, m, o, F), dev)
    check(lib.esr_spotify_forward(_p(album_table), album_table.shape[0], _p(artist_table), artist_table.shape[0], F,
                                  _p(album_ids), _p(artist_ids), n, m, o, _p(pos), _p(neg), _p(cs), _p(ns), _p(gs),
                                  _p(l2), _p(ws), ws.numel(), _stream()), "esr_spotify_forward")
    return pos, neg, cs, ns, gs, l2


def spotify_fwd_bwd(album_table, artist_table, album_ids, artist_ids, n, m, o, regularization):
    """loss and per-occurrence gradient rows of train_spotify.py:78-109.
    Returns (loss[1], album_rows[R] (hashed ids), g_album_rows[R, F], g_artist_rows[R, F])."""
    lib = _lib.load()
    _req(album_table, torch.float32, "album_table"), _req(artist_table, torch.float32, "artist_table")
    album_ids, artist_ids = _req(album_ids, torch.int32, "album_ids"), _req(artist_ids, torch.int32, "artist_ids")
    F, dev, R = album_table.shape[1], album_table.device, n + m + o
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    rows = torch.empty(R, dtype=torch.int32, device=dev)
    ga = torch.empty((R, F), dtype=torch.float32, device=dev)
    gr = torch.empty((R, F), dtype=torch.float32, device=dev)
    ws = _ws(_ws_bytes("esr_spotify_workspace_bytes", n, m, o, F), dev)
    check(lib.esr_spotify_fwd_bwd(_p(album_table), album_table.shape[0], _p(artist_table), artist_table.shape[0], F,
                                  _p(album_ids), _p(artist_ids), n, m, o, float(regularization), _p(loss), _p(rows),
                                  _p(ga), _p(gr), _p(ws), ws.numel(), _stream()), "esr_spotify_fwd_bwd")
    return loss, rows, ga, gr


def spotify_train_step(album_table, album_trace, album_last, artist_table, artist_trace, artist_last, album_ids,
                       artist_ids, n, m, o, regularization, step, lr, momentum):
    """One whole Spotify train step under lazy optax.sgd(lr, momentum) by ONE library call (esr_spotify_train_step): catch
    the playlist's rows up, loss + gradient rows, one sort, the momentum step on the touched rows of both tables.
    Returns loss[1]."""
    lib = _lib.load()
    for t, name in ((album_table, "album_table"), (album_trace, "album_trace"), (artist_table, "artist_table"),
                    (artist_trace, "artist_trace")):
        _req(t, torch.float32, name)
    _req(album_last, torch.int32, "album_last"), _req(artist_last, torch.int32, "artist_last")
    album_ids, artist_ids = _req(album_ids, torch.int32, "album_ids"), _req(artist_ids, torch.int32, "artist_ids")
    F, dev = album_table.shape[1], album_table.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ws = _ws(_ws_bytes("esr_spotify_train_step_workspace_bytes", n, m, o, F), dev)
    check(lib.esr_spotify_train_step(_p(album_table), _p(album_trace), _p(album_last), album_table.shape[0],
                                     _p(artist_table), _p(artist_trace), _p(artist_last), artist_table.shape[0], F,
                                     _p(album_ids), _p(artist_ids), n, m, o, float(regularization), int(step), float(lr),
                                     float(momentum), _p(loss), _p(ws), ws.numel(), _stream()), "esr_spotify_train_step")
    return loss


def spotify_affinity_all(album_table, artist_table, ctx_album, ctx_artist, all_albums, all_artists):
    """Affinity [T] of every track to the context (train_spotify.py:113-119, result[1])."""
    lib = _lib.load()
    _req(album_table, torch.float32, "album_table"), _req(artist_table, torch.float32, "artist_table")
    for t, name in ((ctx_album, "ctx_album"), (ctx_artist, "ctx_artist"), (all_albums, "all_albums"),
                    (all_artists, "all_artists")):
        _req(t, torch.int32, name)
    T = all_albums.numel()
    out = torch.empty(T, dtype=torch.float32, device=album_table.device)
    check(lib.esr_spotify_affinity_all(_p(album_table), album_table.shape[0], _p(artist_table), artist_table.shape[0],
                                       album_table.shape[1], _p(ctx_album), _p(ctx_artist), ctx_album.numel(),
                                       _p(all_albums), _p(all_artists), T, _p(out), _stream()),
          "esr_spotify_affinity_all")
    return out


def rows_to_dense(V, D, sorted_ids, perm, grad_rows, out=None):
    """The dense [V, D] gradient (zero-filled, segment sums scattered) the reference's autodiff yields."""
    lib = _lib.load()
    _req(grad_rows, torch.float32, "grad_rows")
    if out is None:
        out = torch.empty((V, D), dtype=torch.float32, device=grad_rows.device)
    n = sorted_ids.numel()
    check(lib.esr_rows_to_dense(_p(out), V, D, _p(sorted_ids), _p(perm), n, _p(grad_rows), _stream()),
          "esr_rows_to_dense")
    return out


def dense_adam(param, mu, nu, grad, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """In-place optax.adam on every element; `step` is the 1-based count after this update."""
    lib = _lib.load()
    for name, t in (("param", param), ("mu", mu), ("nu", nu), ("grad", grad)):
        _req(t, torch.float32, name)
    check(lib.esr_dense_adam(_p(param), _p(mu), _p(nu), _p(grad), param.numel(), float(lr), float(b1), float(b2),
                             float(eps), int(step), _stream()), "esr_dense_adam")


# ---------------------------------------------------------------------------------------------
def score_all(emb, token):
    """scores[v, t] = emb[v] . emb[token[t]]  -> [V, T]."""
    lib = _lib.load()
    _req(emb, torch.float32, "emb"), _req(token, torch.int32, "token")
    V, D = emb.shape
    T = token.numel()
    scores = torch.empty((V, T), dtype=torch.float32, device=emb.device)
    check(lib.esr_score_all(_p(emb), V, D, _p(token), T, _p(scores), _stream()), "esr_score_all")
    return scores


def argsort_columns(scores):
    """Stable ascending argsort of every column of scores [V, T] -> int32 [V, T]."""
    lib = _lib.load()
    _req(scores, torch.float32, "scores")
    V, T = scores.shape
    indices = torch.empty((V, T), dtype=torch.int32, device=scores.device)
    ws = _ws(_ws_bytes("esr_argsort_columns_workspace_bytes", V, T), scores.device)
    check(lib.esr_argsort_columns(_p(scores), V, T, _p(indices), _p(ws), ws.numel(), _stream()),
          "esr_argsort_columns")
    return indices


def score_topk(queries, candidates, k):
    """Top-k of queries @ candidates^T per query (descending, ties -> lower index)."""
    lib = _lib.load()
    _req(queries, torch.float32, "queries"), _req(candidates, torch.float32, "candidates")
    nq, D = queries.shape
    N = candidates.shape[0]
    out_s = torch.empty((nq, k), dtype=torch.float32, device=queries.device)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=queries.device)
    ws = _ws(_ws_bytes("esr_score_topk_workspace_bytes", nq, N, k), queries.device)
    check(lib.esr_score_topk(_p(queries), _p(candidates), nq, N, D, k, _p(out_s), _p(out_i), _p(ws), ws.numel(),
                             _stream()), "esr_score_topk")
    return out_s, out_i


_RETRIEVE_MODES = {"bf16x3": _lib.RETRIEVE_EXACT, "f16x2": _lib.RETRIEVE_F16X2, "bf16": _lib.RETRIEVE_BF16,
                   "f16r": _lib.RETRIEVE_F16R}


def _retrieve_mode(mode):
    """"exact" / "f32": the exact-split path -- every f32 operand as three bf16 planes (24 significand bits, whatever the
    dynamic range of the matrix), six MFMA terms per product.  "f16x2" must be asked for by name: two fp16 planes of
    x * 2^e with ONE exponent per matrix (three MFMA terms, 1.6x faster at C5): f32-grade while the matrix's elements
    lie within ~2^16 of its largest one, absolute (not relative) error below that -- near-tie orderings can then differ
    from the f32 brute force.  (ESR_RETRIEVE_EXACT=f16x2 restores round 2's mapping.)"""
    if mode in ("exact", "f32"):
        mode = os.environ.get("ESR_RETRIEVE_EXACT", "bf16x3")
    if mode not in _RETRIEVE_MODES:
        raise ValueError("retrieval mode must be exact / f32 / f16x2 / f16r / bf16x3 / bf16, got %r" % (mode,))
    return _RETRIEVE_MODES[mode]


class PreparedCorpus:
    """What retrieve_prepare made of a candidate matrix: the buffer esr_retrieve_topk_prepared reads, and what it was made
    for (mode, shape, the matrix's storage) so that a call cannot be handed another corpus's or another mode's planes."""
    __slots__ = ("blob", "mode", "N", "D", "data_ptr")

    def __init__(self, blob, mode, N, D, data_ptr):
        self.blob, self.mode, self.N, self.D, self.data_ptr = blob, mode, N, D, data_ptr


def retrieve_prepare(candidates, mode="f16r"):
    """The per-corpus half of retrieve_topk done once (esr_retrieve_prepare): the candidates' scaling statistics and their
    planes in the mode's format, to pass as retrieve_topk(..., prepared=...) for as long as `candidates` is unchanged (a
    product table that serves many scene batches: pinterest/make_recommendations.py:123-132)."""
    lib = _lib.load()
    _req(candidates, torch.float32, "candidates")
    N, D = candidates.shape
    m = _retrieve_mode(mode)
    nbytes = lib.esr_retrieve_prepared_bytes(N, D, m)
    if nbytes == 0:
        raise ValueError("retrieve_prepare: bad shape %s / mode %r" % (tuple(candidates.shape), mode))
    blob = _aligned_bytes(nbytes, candidates.device)
    check(lib.esr_retrieve_prepare(_p(candidates), N, D, m, _p(blob), blob.numel(), _stream()), "esr_retrieve_prepare")
    return PreparedCorpus(blob, m, N, D, candidates.data_ptr())


def retrieve_topk(queries, candidates, k, mode="exact", index_base=0, index_step=1, prepared=None):
    """Batched brute-force top-k of queries @ candidates^T (descending, ties -> lower index) on MFMA.
    mode "exact" = "bf16x3": three exact bf16 planes per operand (exact products in f32 accumulation order);
    "f16x2": two scaled fp16 planes (f32-grade within a 2^16 dynamic range per matrix, 1.6x faster); "f16r": one scaled
    fp16 plane as a filter with a proven error band, its survivors re-scored in f32 -- the exact top-k of the f32 scores
    at a third of f16x2's matrix work (include/esr_hip.h ESR_RETRIEVE_F16R); "bf16": one plane (approximate).
    Reported indices are index_base + n * index_step for local candidate row n.
    prepared: retrieve_prepare(candidates, mode) -- the candidates' statistics and planes made once."""
    lib = _lib.load()
    _req(queries, torch.float32, "queries"), _req(candidates, torch.float32, "candidates")
    nq, D = queries.shape
    N = candidates.shape[0]
    m = _retrieve_mode(mode)
    out_s = torch.empty((nq, k), dtype=torch.float32, device=queries.device)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=queries.device)
    ws = _ws(_ws_bytes("esr_retrieve_workspace_bytes", nq, N, D, k, m), queries.device)
    if prepared is not None:
        if not isinstance(prepared, PreparedCorpus) or (prepared.mode, prepared.N, prepared.D, prepared.data_ptr) != \
                (m, N, D, candidates.data_ptr()):
            raise ValueError("prepared must be retrieve_prepare(candidates, mode) of THIS candidate matrix and mode")
        check(lib.esr_retrieve_topk_prepared(_p(queries), _p(candidates), _p(prepared.blob), nq, N, D, k, m, index_base,
                                             index_step, _p(out_s), _p(out_i), _p(ws), ws.numel(), _stream()),
              "esr_retrieve_topk_prepared")
        return out_s, out_i
    check(lib.esr_retrieve_topk(_p(queries), _p(candidates), nq, N, D, k, m, index_base, index_step, _p(out_s),
                                _p(out_i), _p(ws), ws.numel(), _stream()), "esr_retrieve_topk")
    return out_s, out_i


def rescore_candidates(queries, candidates, indices, index_base=0, index_step=1):
    """scores[q, j] = queries[q] . candidates[(indices[q, j] - index_base) // index_step] in f32."""
    lib = _lib.load()
    _req(queries, torch.float32, "queries"), _req(candidates, torch.float32, "candidates")
    indices = _req(indices, torch.int32, "indices")
    nq, D = queries.shape
    out = torch.empty(indices.shape, dtype=torch.float32, device=queries.device)
    check(lib.esr_rescore_candidates(_p(queries), _p(candidates), nq, candidates.shape[0], D, _p(indices),
                                     indices.shape[1], index_base, index_step, _p(out), _stream()),
          "esr_rescore_candidates")
    return out


def topk_merge(scores, indices, k):
    """Top-k of per-query (score, index) lists [nq, n] (descending, ties -> lower index)."""
    lib = _lib.load()
    scores = _req(scores, torch.float32, "scores")
    indices = _req(indices, torch.int32, "indices")
    nq, n = scores.shape
    out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=scores.device)
    check(lib.esr_topk_merge(_p(scores), _p(indices), nq, n, k, _p(out_s), _p(out_i), _stream()), "esr_topk_merge")
    return out_s, out_i


def inbatch_towers_fwd_bwd(query_table, cand_table, query_ids, cand_ids, scale, regularization, batch_size,
                           grad_positions=None, precision="auto"):
    """In-batch softmax step head straight from the tower tables (no materialised Q / C): rows query_table[query_ids],
    cand_table[cand_ids]; tables f32 or bf16 [V, 128], B % 128 == 0.  Returns (loss[1], lse[B], gQ, gC) like
    inbatch_softmax_fwd_bwd.  precision: "auto" / "f16x2" / "bf16x3" (see inbatch_split_path; bf16 tables always take
    the bf16x3 kernels, where they are one-plane).  grad_positions = (gq_rows, gc_rows) int32 [B] each: the gradient
    rows are scattered into ONE [2B, 128] buffer at those rows instead (returned as gQ, with gC = None)."""
    lib = _lib.load()
    dt = _table_dtype(query_table, "query_table")
    if _table_dtype(cand_table, "cand_table") != dt:
        raise TypeError("both tower tables must have the same dtype")
    query_ids, cand_ids = _req(query_ids, torch.int32, "query_ids"), _req(cand_ids, torch.int32, "cand_ids")
    B, D, dev = query_ids.numel(), query_table.shape[1], query_table.device
    path = inbatch_split_path(precision, B, D, bf16_tables=query_table.dtype == torch.bfloat16)
    if path is None:
        raise ValueError("inbatch_towers_fwd_bwd needs D <= 128 (a multiple of 4) and B %% 128 == 0 (got B=%d, D=%d) and a split precision"
                         % (B, D))
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    lse = torch.empty(B, dtype=torch.float32, device=dev)
    gQC = torch.empty((2 * B, D), dtype=torch.float32, device=dev)
    if path == "f16x2":
        ws = _ws(_ws_bytes("esr_inbatch2h_workspace_bytes", B, D), dev)
        fn, name = lib.esr_inbatch_towers_fwd_bwd_f16x2, "esr_inbatch_towers_fwd_bwd_f16x2"
    else:
        ws = _ws(_ws_bytes("esr_inbatch3_workspace_bytes", B, D), dev)
        fn, name = lib.esr_inbatch_towers_fwd_bwd_bf16x3, "esr_inbatch_towers_fwd_bwd_bf16x3"
    if grad_positions is not None:
        gqr, gcr = _req(grad_positions[0], torch.int32, "gq_rows"), _req(grad_positions[1], torch.int32, "gc_rows")
        gq_ptr = gc_ptr = _p(gQC)
    else:
        gqr = gcr = None
        gq_ptr, gc_ptr = _p(gQC[:B]), _p(gQC[B:])
    check(fn(_p(query_table), query_table.shape[0], _p(cand_table), cand_table.shape[0], dt, D, _p(query_ids),
             _p(cand_ids), _p(gqr), _p(gcr), B, float(scale), float(regularization), float(batch_size), _p(loss),
             _p(lse), gq_ptr, gc_ptr, _p(ws), ws.numel(), _stream()), name)
    if grad_positions is not None:
        return loss, lse, gQC, None
    return loss, lse, gQC[:B], gQC[B:]


def inbatch_train_step(query_table, query_accum, cand_table, cand_accum, query_ids, cand_ids, scale, regularization,
                       batch_size, lr, eps=1e-7, presorted=None, long_runs=-1, side_stream=None, want_lse=False):
    """One whole in-batch training step of the two towers (esr_inbatch_train_step_f16x2): gather + fp16 x 2 score passes +
    row-sparse Adagrad on both towers, in place.  presorted = (sorted virtual ids, perm) of [query_ids ; Vq + cand_ids]
    (segment_sort_multi / segment_sort_batched), else the list is sorted here; long_runs: 0 when long_run_hint said no id
    occurs more than 32 times.  side_stream (a torch.cuda.Stream of the same device): merge<Q> and the query tower's
    update run on it beside pass C (bit-identical results).  Returns loss[1] (and lse[B] with want_lse)."""
    lib = _lib.load()
    dt = _table_dtype(query_table, "query_table")
    if _table_dtype(cand_table, "cand_table") != dt:
        raise TypeError("both tower tables must have the same dtype")
    _req(query_accum, torch.float32, "query_accum"), _req(cand_accum, torch.float32, "cand_accum")
    query_ids, cand_ids = _req(query_ids, torch.int32, "query_ids"), _req(cand_ids, torch.int32, "cand_ids")
    B, D, dev = query_ids.numel(), query_table.shape[1], query_table.device
    if cand_ids.numel() != B or cand_table.shape[1] != D:
        raise ValueError("id counts / tower widths differ")
    if query_accum.shape != query_table.shape or cand_accum.shape != cand_table.shape:
        raise ValueError("accumulator shapes do not match their towers")
    if not (D % 4 == 0 and 0 < D <= 128 and B % 128 == 0 and 0 < B <= INBATCH_F16X2_MAX_B):
        raise ValueError("inbatch_train_step needs D <= 128 (a multiple of 4), B %% 128 == 0 and B <= %d (got B=%d, D=%d)"
                         % (INBATCH_F16X2_MAX_B, B, D))
    sid = perm = None
    if presorted is not None:
        sid, perm = _req(presorted[0], torch.int32, "sorted_ids"), _req(presorted[1], torch.int32, "perm")
        if sid.numel() != 2 * B or perm.numel() != 2 * B:
            raise ValueError("presorted ids / perm must have 2 B entries")
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    lse = torch.empty(B, dtype=torch.float32, device=dev) if want_lse else None
    ws = _ws(_ws_bytes("esr_inbatch_train_step_workspace_bytes", B, D), dev)
    side = side_stream.cuda_stream if side_stream is not None else None
    check(lib.esr_inbatch_train_step_f16x2(_p(query_table), _p(query_accum), query_table.shape[0], _p(cand_table),
                                           _p(cand_accum), cand_table.shape[0], dt, D, _p(query_ids), _p(cand_ids), B,
                                           float(scale), float(regularization), float(batch_size), float(lr), float(eps),
                                           _p(sid), _p(perm), int(long_runs), _p(loss), _p(lse), _p(ws), ws.numel(),
                                           _stream(), side), "esr_inbatch_train_step_f16x2")
    if side_stream is not None:
        ws.record_stream(side_stream)  # (the call joined the streams; this only tells torch's allocator who used the block)
    return (loss, lse) if want_lse else loss


def sorted_membership(cur, seq, sentinel):
    """flags uint8 [L, n]: cur[l, j] != sentinel and cur[l, j] occurs in the ascending row seq[l] (esr_sorted_membership)."""
    lib = _lib.load()
    cur, seq = _req(cur, torch.int32, "cur"), _req(seq, torch.int32, "seq")
    L, n = cur.shape
    flags = torch.empty((L, n), dtype=torch.uint8, device=cur.device)
    check(lib.esr_sorted_membership(_p(cur), n, _p(seq), seq.shape[1], L, int(sentinel), _p(flags), _stream()),
          "esr_sorted_membership")
    return flags


def flagged_first(flags, values, slice_len, counts_out):
    """Stable partition of every row of `values` [L, n] (None: the positions 0 .. n - 1) by flags uint8 [L, n]: returns
    int32 [L, n] whose rows START with the flagged entries in order (the rest is unspecified); counts_out int64 [G, L]
    (any strides) receives, per row, the flagged entries inside each of G consecutive slices of lengths slice_len int64
    [L, G] (esr_flagged_first)."""
    lib = _lib.load()
    flags = _req(flags, torch.uint8, "flags")
    L, n = flags.shape
    slice_len = _req(slice_len, torch.int64, "slice_len")
    G = slice_len.shape[1]
    if values is not None:
        values = _req(values, torch.int32, "values")
    if counts_out.dtype != torch.int64 or tuple(counts_out.shape) != (G, L):
        raise ValueError("counts_out must be an int64 [G, L] view")
    out = torch.empty((L, n), dtype=torch.int32, device=flags.device)
    ws = _ws(_ws_bytes("esr_flagged_first_workspace_bytes", n, L), flags.device)
    check(lib.esr_flagged_first(_p(flags), _p(values), n, L, _p(slice_len), G, _p(out), counts_out.data_ptr(),
                                counts_out.stride(1), counts_out.stride(0), _p(ws), ws.numel(), _stream()),
          "esr_flagged_first")
    return out


def run_offsets(sorted_ids, nvalues, want_max=False):
    """int32 [nvalues + 1]: the first position of each value in the ascending int32 list (esr_run_offsets); with want_max
    also the longest run as an int32 [1] device tensor."""
    lib = _lib.load()
    sorted_ids = _req(sorted_ids, torch.int32, "sorted_ids")
    off = torch.empty(nvalues + 1, dtype=torch.int32, device=sorted_ids.device)
    mx = torch.empty(1, dtype=torch.int32, device=sorted_ids.device) if want_max else None
    check(lib.esr_run_offsets(_p(sorted_ids), sorted_ids.numel(), int(nvalues), _p(off), _p(mx), _stream()), "esr_run_offsets")
    return (off, mx) if want_max else off


def ivf_centroids(sums, list_off=None, train=None, fallback_rows=None):
    """Unit-length centroids from per-list sums [nlist, D]; a list without members (list_off) takes training row
    fallback_rows[v] instead (esr_ivf_centroids)."""
    lib = _lib.load()
    sums = _req(sums, torch.float32, "sums")
    nlist, D = sums.shape
    out = torch.empty_like(sums)
    check(lib.esr_ivf_centroids(_p(sums), _p(list_off), _p(train), _p(fallback_rows), nlist, D, _p(out), _stream()),
          "esr_ivf_centroids")
    return out


def recall_at_k(approx_indices, exact_indices):
    """hits / (Q * ke): the fraction of exact_indices [Q, ke] that also occur in the same row of approx_indices [Q, ka]
    (esr_recall_at_k).  Synchronises to read the one counter back."""
    lib = _lib.load()
    a = approx_indices if approx_indices.dtype == torch.int32 else approx_indices.to(torch.int32)
    e = exact_indices if exact_indices.dtype == torch.int32 else exact_indices.to(torch.int32)
    a, e = _req(a.contiguous(), torch.int32, "approx_indices"), _req(e.contiguous(), torch.int32, "exact_indices")
    if a.dim() != 2 or e.dim() != 2 or a.shape[0] != e.shape[0]:
        raise ValueError("approx_indices [Q, ka] and exact_indices [Q, ke] must have the same number of rows")
    hits = torch.empty(1, dtype=torch.int64, device=a.device)
    check(lib.esr_recall_at_k(_p(a), a.shape[0], a.shape[1], _p(e), e.shape[1], _p(hits), _stream()), "esr_recall_at_k")
    return float(int(hits.item())) / max(1, e.numel())


def bucket_ids_by_owner_batched(id_lists, world, offsets):
    """bucket_ids_by_owner for the lists of several coming batches in one launch pair (esr_bucket_ids_by_owner_batched).
    id_lists: per batch, the int32 segments of its virtual list [ids_k + offsets[k]] (same lengths in every batch).
    Returns (local_rows [L, n], perm [L, n], counts [L, world] int64, inverse [L, n])."""
    import ctypes
    lib = _lib.load()
    nb, nseg = len(id_lists), len(id_lists[0])
    dev = id_lists[0][0].device
    seg_counts = [int(t.numel()) for t in id_lists[0]]
    for segs in id_lists:
        if len(segs) != nseg or [int(t.numel()) for t in segs] != seg_counts:
            raise ValueError("every batch must have the same segment lengths")
    id_lists = [[_req(t, torch.int32, "ids") for t in segs] for segs in id_lists]
    n = sum(seg_counts)
    local_rows = torch.empty((nb, n), dtype=torch.int32, device=dev)
    perm = torch.empty((nb, n), dtype=torch.int32, device=dev)
    inverse = torch.empty((nb, n), dtype=torch.int32, device=dev)
    counts = torch.empty((nb, world), dtype=torch.int64, device=dev)
    ws = _ws(_ws_bytes("esr_bucket_batched_workspace_bytes", n, nb), dev)
    ptrs = (ctypes.c_void_p * (nb * nseg))(*[t.data_ptr() for segs in id_lists for t in segs])
    cnt = (ctypes.c_int64 * nseg)(*seg_counts)
    off = (ctypes.c_int64 * nseg)(*[int(x) for x in offsets])
    check(lib.esr_bucket_ids_by_owner_batched(ptrs, cnt, off, nseg, nb, int(world), _p(local_rows), _p(perm), _p(inverse),
                                              _p(counts), _p(ws), ws.numel(), _stream()),
          "esr_bucket_ids_by_owner_batched")
    return local_rows, perm, counts, inverse


def segment_sort_batched(id_lists, offsets, num_rows, out=None):
    """The occurrence lists of several coming batches sorted in one launch sequence (esr_segment_sort_ids_batched).
    id_lists: per batch, the int32 segments of its virtual list [ids_k + offsets[k]] (same lengths in every batch).
    Returns (sorted_ids, perm), each [len(id_lists), n]: row b is what segment_sort / the multi-segment sort gives for
    batch b alone.  out = (sorted_ids, perm, workspace) to reuse buffers."""
    import ctypes
    lib = _lib.load()
    nb, nseg = len(id_lists), len(id_lists[0])
    dev = id_lists[0][0].device
    counts = [int(t.numel()) for t in id_lists[0]]
    for segs in id_lists:
        if len(segs) != nseg or [int(t.numel()) for t in segs] != counts:
            raise ValueError("every batch must have the same segment lengths")
    id_lists = [[_req(t, torch.int32, "ids") for t in segs] for segs in id_lists]
    n = sum(counts)
    if out is None:
        sorted_ids = torch.empty((nb, n), dtype=torch.int32, device=dev)
        perm = torch.empty((nb, n), dtype=torch.int32, device=dev)
        ws = _ws(_ws_bytes("esr_segment_sort_batched_workspace_bytes", n, nb), dev)
    else:
        sorted_ids, perm, ws = out
    ptrs = (ctypes.c_void_p * (nb * nseg))(*[t.data_ptr() for segs in id_lists for t in segs])
    cnt = (ctypes.c_int64 * nseg)(*counts)
    off = (ctypes.c_int64 * nseg)(*[int(x) for x in offsets])
    check(lib.esr_segment_sort_ids_batched(ptrs, cnt, off, nseg, nb, int(num_rows), _p(sorted_ids), _p(perm), _p(ws),
                                           ws.numel(), _stream()), "esr_segment_sort_ids_batched")
    return sorted_ids, perm


def bucket_ids_by_owner(ids, world, want_inverse=False, offsets=None, counts_out=None):
    """Stable bucket by owner = id % world.  Returns (local_rows, perm, counts[world] int64 on device) and, with
    want_inverse, also inverse with inverse[perm[k]] = k.  `ids` may be a list of int32 tensors with `offsets`: the
    virtual list [ids_0 + offsets[0] ; ids_1 + offsets[1] ; ...] is bucketed in place (no concatenated copy).
    `counts_out` (contiguous int64 [world] on the device) receives the counts instead of a fresh tensor."""
    import ctypes
    lib = _lib.load()
    segs = list(ids) if isinstance(ids, (list, tuple)) else None
    if segs is None:
        ids = _req(ids, torch.int32, "ids")
        n, dev = ids.numel(), ids.device
    else:
        for t in segs:
            _req(t, torch.int32, "ids")
        n, dev = sum(int(t.numel()) for t in segs), segs[0].device
    local_rows = torch.empty(n, dtype=torch.int32, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    inverse = torch.empty(n, dtype=torch.int32, device=dev) if want_inverse else None
    if counts_out is None:
        counts = torch.empty(world, dtype=torch.int64, device=dev)
    else:
        counts = _req(counts_out, torch.int64, "counts_out")
        if counts.numel() != world:
            raise ValueError("counts_out must hold %d int64" % world)
    ws = _ws(_ws_bytes("esr_bucket_workspace_bytes", n), dev)
    if segs is None:
        check(lib.esr_bucket_ids_by_owner(_p(ids), n, world, _p(local_rows), _p(perm), _p(inverse), _p(counts), _p(ws),
                                          ws.numel(), _stream()), "esr_bucket_ids_by_owner")
    else:
        k = len(segs)
        ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in segs])
        cnt = (ctypes.c_int64 * k)(*[int(t.numel()) for t in segs])
        off = (ctypes.c_int64 * k)(*[int(x) for x in (offsets if offsets is not None else [0] * k)])
        check(lib.esr_bucket_ids_by_owner_multi(ptrs, cnt, off, k, world, _p(local_rows), _p(perm), _p(inverse),
                                                _p(counts), _p(ws), ws.numel(), _stream()),
              "esr_bucket_ids_by_owner_multi")
    if want_inverse:
        return local_rows, perm, counts, inverse
    return local_rows, perm, counts


# ---- a row-sharded step's exchange halves, one library call each (esr_shard_step.hip) ----------------------------------
def i64_array(values):
    """host int64 array for the count / offset arguments of the sharded calls (made once per plan / group, reused)."""
    import ctypes
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def sharded_lookup(comm, world, tables_c, loff_c, ntables, dtype, D, asked_rows, asked_c, ask_c, served, back):
    """esr_sharded_lookup: gather the rows asked of this rank + the rows exchange, on the current stream.  `comm`: the
    DirectExchange communicator (ctypes.c_void_p) or None at world 1; served may be None at world 1."""
    lib = _lib.load()
    check(lib.esr_sharded_lookup(comm, world, tables_c, loff_c, ntables, dtype, D, _p(asked_rows), asked_c, ask_c,
                                 _p(served) if served is not None else None, _p(back), _stream()),
          "esr_sharded_lookup")
    return back


def sharded_update(comm, world, tables_c, accums_c, loff_c, ntables, dtype, D, grad_rows, sorted_uidx, occ_perm, summed,
                   ask_c, asked_c, grad_dtype, send_bf16, recv_raw, recv_grads, owner_sorted, owner_perm, lr, eps=1e-7,
                   long_runs=-1):
    """esr_sharded_update: [per-distinct-row sum ->] gradient rows to their owners -> the owner's fused segment-reduce +
    Adagrad, on the current stream."""
    lib = _lib.load()
    q = lambda t: _p(t) if t is not None else None  # noqa: E731
    check(lib.esr_sharded_update(comm, world, tables_c, accums_c, loff_c, ntables, dtype, D, _p(grad_rows),
                                 int(grad_rows.shape[0]), q(sorted_uidx), q(occ_perm), q(summed), ask_c, asked_c,
                                 grad_dtype, q(send_bf16), q(recv_raw), q(recv_grads), q(owner_sorted), q(owner_perm),
                                 float(lr), float(eps), int(long_runs), _stream()), "esr_sharded_update")


def shard_group_struct(comm, world, tables_c, accums_c, loff_c, ntables, dtype, D, grad_dtype):
    """esr_shard_group_t over the host arrays of ptr_array / i64_array (the caller keeps them -- and the tensors -- alive)."""
    import ctypes
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)  # noqa: E731
    comm = getattr(comm, "value", comm)
    return _lib.ShardGroupStruct(comm, world, cast(tables_c), cast(accums_c), cast(loff_c), ntables, dtype, D, grad_dtype)


def routing_plan_struct(asked_rows, asked_c, ask_c, index, sorted_uidx, occ_perm, owner_sorted, owner_perm, long_runs=-1):
    """esr_routing_plan_t of one batch (device tensors + the host count arrays; the caller keeps them alive)."""
    import ctypes
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)  # noqa: E731
    q = lambda t: t.data_ptr() if t is not None and t.numel() else None  # noqa: E731
    return _lib.RoutingPlanStruct(q(asked_rows), cast(asked_c), cast(ask_c), q(index), q(sorted_uidx), q(occ_perm),
                                  q(owner_sorted), q(owner_perm), int(long_runs))


def step_overlap_struct(backs, ready, stale, next_plan_s, next_backs, next_serveds, comm2, side_stream):
    """esr_step_overlap_t of one overlapped step (include/esr_hip.h).  backs: this batch's rows per group as the previous
    call fetched them (None: look up in line) with `ready`, that call's next_ready; stale = (rows, asked host array, pos, ask
    host array) of esrecsys_amd.sharded.StaleRows; next_*: the coming batch's plan struct and its buffers per group (None:
    last step); side_stream: a torch.cuda.Stream.  The caller keeps every tensor / array alive until the NEXT call."""
    import ctypes
    ov = _lib.StepOverlapStruct()
    q = lambda t: t.data_ptr() if t is not None and t.numel() else None  # noqa: E731
    for i, b in enumerate(backs or ()):
        ov.back[i] = b.data_ptr()
    ov.ready = ready
    if backs and stale is not None:
        rows, asked_c, pos, ask_c = stale
        ov.stale_rows, ov.stale_pos = q(rows), q(pos)
        ov.stale_asked, ov.stale_ask = ctypes.cast(asked_c, ctypes.c_void_p), ctypes.cast(ask_c, ctypes.c_void_p)
    if next_plan_s is not None:
        ov.next_plan = ctypes.cast(ctypes.pointer(next_plan_s), ctypes.c_void_p)
        for i, b in enumerate(next_backs):
            ov.next_back[i] = b.data_ptr()
        for i, b in enumerate(next_serveds):
            ov.next_served[i] = q(b)
    ov.comm2 = getattr(comm2, "value", comm2)
    ov.side = side_stream.cuda_stream if side_stream is not None else None
    return ov


def overlap_release(ready):
    """Release a next_ready event that no call will consume (a loop that stopped early)."""
    if ready:
        _lib.load().esr_sharded_overlap_release(ready)


def sharded_triplet_step(group_s, plan_s, B, regularization, batch_size, lr, eps, device, overlap=None):
    """esr_sharded_triplet_step: lookup -> triplet loss on the rows where they landed -> update, one library call.
    Returns loss [1] (this rank's share).  overlap: a step_overlap_struct (esr_sharded_triplet_step_overlapped)."""
    import ctypes
    lib = _lib.load()
    nb = int(lib.esr_sharded_triplet_step_workspace_bytes(ctypes.byref(group_s), ctypes.byref(plan_s), B))
    ov = ctypes.byref(overlap) if overlap is not None else None
    if ov is not None:
        nb += int(lib.esr_sharded_step_overlap_workspace_bytes(ctypes.byref(group_s), ov))
    ws = _ws(nb, device)
    loss = torch.empty(1, dtype=torch.float32, device=device)
    check(lib.esr_sharded_triplet_step_overlapped(ctypes.byref(group_s), ctypes.byref(plan_s), ov, B, float(regularization),
                                                  float(batch_size), float(lr), float(eps), _p(loss), _p(ws), ws.numel(),
                                                  _stream()), "esr_sharded_triplet_step")
    return loss


def sharded_glove_step(emb_s, bias_s, plan_s, target, B, mode, lr, eps, overlap=None):
    """esr_sharded_glove_step: both lookups -> GloVe loss -> both updates, one library call.  Returns loss [1].
    overlap: a step_overlap_struct (esr_sharded_glove_step_overlapped)."""
    import ctypes
    lib = _lib.load()
    _req(target, torch.float32, "target")
    nb = int(lib.esr_sharded_glove_step_workspace_bytes(ctypes.byref(emb_s), ctypes.byref(bias_s), ctypes.byref(plan_s), B))
    ov = ctypes.byref(overlap) if overlap is not None else None
    if ov is not None:
        nb += int(lib.esr_sharded_step_overlap_workspace_bytes(ctypes.byref(emb_s), ov)) + \
            int(lib.esr_sharded_step_overlap_workspace_bytes(ctypes.byref(bias_s), ov))
    ws = _ws(nb, target.device)
    loss = torch.empty(1, dtype=torch.float32, device=target.device)
    check(lib.esr_sharded_glove_step_overlapped(ctypes.byref(emb_s), ctypes.byref(bias_s), ctypes.byref(plan_s), ov,
                                                _p(target), B, int(mode), float(lr), float(eps), _p(loss), _p(ws),
                                                ws.numel(), _stream()), "esr_sharded_glove_step")
    return loss


def rows_f32_to_bf16(rows):
    """[n, D] f32 -> bf16, round to nearest even (esr_rows_f32_to_bf16: what the bf16 gradient exchange sends)."""
    lib = _lib.load()
    _req(rows, torch.float32, "rows")
    out = torch.empty(rows.shape, dtype=torch.bfloat16, device=rows.device)
    check(lib.esr_rows_f32_to_bf16(_p(rows), rows.shape[0], rows.shape[1], _p(out), _stream()), "esr_rows_f32_to_bf16")
    return out
