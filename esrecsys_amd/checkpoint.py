"""TrainState <-> bytes in the Flax msgpack layout ("next" row N4 of SURVEY.md 8f).

The reference writes ``flax.serialization.to_bytes(state)`` to ``checkpoint-%05d.flax`` / ``glove.flax``
(wikipedia/train_cooccurence.py:129-134,188-192; pinterest/train_shop_the_look.py:224-232).  Flax is not
vendored; the wire layout below is restated from flax==0.5.2's documented behaviour [upstream]:

  * ``to_state_dict(TrainState)`` = ``{'step': ..., 'params': <tree>, 'opt_state': <tree>}`` (apply_fn / tx are
    static fields and are not serialised); tuples and lists become dicts keyed '0', '1', ...; NamedTuple
    optimizer states become dicts of their fields (optax.adam: ``{'0': {'count', 'mu', 'nu'}, '1': {}}``;
    optax.adagrad: ``{'0': {'sum_of_squares'}, '1': {}}``);
  * the dict is msgpack-packed; every ndarray is ``ExtType(1, packb((shape, dtype.name, raw C-order bytes)))``;
    NumPy scalars are ``ExtType(3, ...)`` with the same payload; arrays above 2**30 bytes are split into
    ``{'__msgpack_chunked_array__': True, 'shape': {...}, 'chunks': {'0': ..., ...}}``.

``from_bytes`` restores INTO a target state and returns it -- the reference discards the result of
``from_bytes`` (train_cooccurence.py:173-177), which makes its --resume_checkpoint a no-op; that bug is not
reproduced.
"""
import msgpack
import numpy as np
import torch

from .train_state import TrainState

_EXT_NDARRAY, _EXT_NPSCALAR = 1, 3
_MAX_CHUNK_BYTES = 2 ** 30


def _to_numpy(x):
    if isinstance(x, torch.Tensor):
        if x.dtype == torch.bfloat16:  # msgpack payload dtype name as NumPy/ml_dtypes spells it
            return x.detach().cpu().view(torch.int16).numpy().view(np.dtype("V2")), "bfloat16"
        return x.detach().cpu().numpy(), None
    return np.asarray(x), None


def _pack_array(arr, dtype_name=None):
    name = dtype_name or arr.dtype.name
    return msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(arr.shape), name, arr.tobytes("C")), use_bin_type=True))


def _tuple_to_dict(seq):
    return {str(i): v for i, v in enumerate(seq)}


def _encode(obj, keep_order=False):
    if isinstance(obj, dict):
        # key order: a reference state that has taken a step went through jax.tree_util (optax.apply_updates,
        # tree_map(zeros_like) for mu / nu), which rebuilds dicts with SORTED keys; the top level is the dataclass
        # field order of TrainState.  Readers do not care; the golden-bytes test does.
        keys = list(obj) if keep_order else sorted(obj, key=str)
        return {str(k): _encode(obj[k]) for k in keys}
    if isinstance(obj, (list, tuple)):
        return _encode(_tuple_to_dict(obj), keep_order=True)  # index maps stay in NUMERIC order ('10' after '9', not '1')
    if isinstance(obj, (torch.Tensor, np.ndarray)):
        arr, name = _to_numpy(obj)
        if arr.nbytes > _MAX_CHUNK_BYTES:
            per = max(1, _MAX_CHUNK_BYTES // arr.dtype.itemsize)
            flat = arr.reshape(-1)
            chunks = [flat[i:i + per] for i in range(0, flat.size, per)]
            # (a literal dict: the 'shape' / 'chunks' index maps are emitted as built, in numeric order)
            return {"__msgpack_chunked_array__": True, "shape": _tuple_to_dict(list(arr.shape)),
                    "chunks": {str(i): _pack_array(c, name) for i, c in enumerate(chunks)}}
        return _pack_array(arr, name)
    if isinstance(obj, np.generic):
        a = np.asarray(obj)
        return msgpack.ExtType(_EXT_NPSCALAR, msgpack.packb(([], a.dtype.name, a.tobytes()), use_bin_type=True))
    return obj  # int / float / bool / None / str


def _ext_hook(code, data):
    if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
        shape, name, raw = msgpack.unpackb(data, raw=False)
        if name == "bfloat16":
            return torch.frombuffer(bytearray(raw), dtype=torch.bfloat16).reshape(shape)
        arr = np.frombuffer(raw, dtype=np.dtype(name)).reshape(shape)
        return arr[()] if code == _EXT_NPSCALAR else arr
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get("__msgpack_chunked_array__"):
            shape = [tree["shape"][str(i)] for i in range(len(tree["shape"]))]
            chunks = [tree["chunks"][str(i)] for i in range(len(tree["chunks"]))]
            if any(isinstance(c, torch.Tensor) for c in chunks):  # bf16 chunks come back as torch tensors
                return torch.cat([torch.as_tensor(c).reshape(-1) for c in chunks]).reshape(shape)
            return np.concatenate([np.asarray(c).reshape(-1) for c in chunks]).reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def state_dict(state):
    """flax.serialization.to_state_dict(TrainState): step, params, opt_state."""
    tx = state.tx
    if hasattr(tx, "to_optax_state"):
        opt = tx.to_optax_state(state.opt_state)
    else:
        opt = state.opt_state
    # step: the reference's jitted update_model returns it as an int32 0-d array (ExtType 1), not a Python int
    return {"step": np.asarray(int(state.step), np.int32), "params": state.params, "opt_state": opt}


def to_bytes(state):
    """``flax.serialization.to_bytes(state)`` for an esrecsys_amd TrainState (or any nested dict of tensors)."""
    if isinstance(state, TrainState):
        return msgpack.packb(_encode(state_dict(state), keep_order=True), use_bin_type=True)
    return msgpack.packb(_encode(state), use_bin_type=True)


def msgpack_restore(data):
    return _unchunk(msgpack.unpackb(data, ext_hook=_ext_hook, raw=False, strict_map_key=False))


def _restore_into(target, loaded, path=""):
    if isinstance(target, dict):
        if not isinstance(loaded, dict) or set(map(str, target)) != set(loaded):
            raise ValueError("checkpoint tree mismatch at %r: %s vs %s" % (path, sorted(map(str, target)),
                                                                           sorted(loaded) if isinstance(loaded, dict) else type(loaded)))
        return {k: _restore_into(v, loaded[str(k)], path + "/" + str(k)) for k, v in target.items()}
    if isinstance(target, (list, tuple)):
        vals = [_restore_into(v, loaded[str(i)], path + "/" + str(i)) for i, v in enumerate(target)]
        return type(target)(vals)
    if isinstance(target, torch.Tensor):
        src = loaded if isinstance(loaded, torch.Tensor) else torch.from_numpy(np.array(loaded, copy=True))
        if tuple(src.shape) != tuple(target.shape):
            raise ValueError("shape mismatch at %r: %s vs %s" % (path, tuple(src.shape), tuple(target.shape)))
        target.copy_(src.to(target.dtype))
        return target
    if isinstance(target, (int, np.integer)) and not isinstance(target, bool):
        return int(np.asarray(loaded))
    return loaded


def from_bytes(target, data):
    """``flax.serialization.from_bytes(target, data)``: restores the tables IN PLACE into ``target``'s tensors
    (they stay where they are in HBM) and returns the restored TrainState."""
    loaded = msgpack_restore(data)
    if not isinstance(target, TrainState):
        return _restore_into(target, loaded)
    params = _restore_into(target.params, loaded["params"], "params")
    tx = target.tx
    ref_opt = tx.to_optax_state(target.opt_state) if hasattr(tx, "to_optax_state") else target.opt_state
    opt = _restore_into(ref_opt, loaded["opt_state"], "opt_state")
    if hasattr(tx, "from_optax_state"):
        opt = tx.from_optax_state(opt)
    return target.replace(step=int(np.asarray(loaded["step"])), params=params, opt_state=opt)
