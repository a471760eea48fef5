"""CPU: the C-ABI library loads, exports every symbol include/esr_hip.h declares, and rejects bad
arguments with ESR_EINVAL before touching a device (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "esr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(esr_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from esrecsys_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from esrecsys_amd.build import build_library
        build_library()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from esrecsys_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), "libesr_hip.so does not export %s" % name
    # the ctypes table binds exactly the header's entry points
    assert sorted(_lib.SIGNATURES) == declared


def test_probe_library_is_separate_and_exports_its_header(lib):
    """The measurement probes are not part of the product ABI: include/esr_probe.h, libesr_probe.so."""
    from esrecsys_amd import _lib
    from esrecsys_amd.build import build_probe_library
    if not os.path.exists(_lib.PROBE_LIB_PATH):
        build_probe_library()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "esr_probe.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(esr_[a-z0-9_]+)\s*\(", text)))
    probe = _lib.load_probe()
    assert declared == sorted(_lib.PROBE_SIGNATURES) == ["esr_probe_hbm_read", "esr_probe_mfma", "esr_probe_mfma_valu"]
    for name in declared:
        assert hasattr(probe, name) and name not in _lib.SIGNATURES
        with pytest.raises(AttributeError):
            getattr(ctypes.CDLL(_lib.LIB_PATH), name)


def test_version_and_error_string(lib):
    assert lib.esr_version() >= 100
    assert isinstance(lib.esr_last_error(), bytes)


def test_bad_arguments_are_rejected_without_a_device(lib):
    EINVAL, EWORKSPACE = -1, -3
    assert lib.esr_gather_rows(None, 0, 10, 4, None, -1, None, None) == EINVAL
    assert b"esr_gather_rows" in lib.esr_last_error()
    assert lib.esr_gather_rows(None, 7, 10, 4, None, 1, None, None) == EINVAL      # bad dtype
    assert lib.esr_gather_rows(None, 0, 10, 4, None, 0, None, None) == 0           # n == 0 is a no-op
    assert lib.esr_gather_rows(None, 0, 10, 4, None, 5, None, None) == EINVAL      # null pointers
    assert lib.esr_glove_fwd_bwd(None, None, 10, 4, None, None, 0, 0, None, None, None, None, 0, None) == EINVAL
    assert lib.esr_inbatch_softmax_fwd_bwd(16, 16, 0, 128, 1.0, 0.0, 33.0, 16, 16, 16, 16, 16, 1 << 20, None) == EINVAL
    assert b"must be positive" in lib.esr_last_error()
    assert lib.esr_inbatch_softmax_fwd_bwd_bf16x3(16, 16, 33, 128, 1.0, 0.0, 33.0, 16, 16, 16, 16, 16, 1 << 20,
                                                  None) == EINVAL
    assert b"multiple of 128" in lib.esr_last_error()
    # the fp16 x 2 entry points: B a multiple of 128, at most 16384 (B x B probabilities in the workspace), D = 128
    assert lib.esr_inbatch_softmax_fwd_bwd_f16x2(16, 16, 33, 128, 1.0, 0.0, 33.0, 16, 16, 16, 16, 16, 1 << 20,
                                                 None) == EINVAL
    assert b"multiple of 128" in lib.esr_last_error()
    assert lib.esr_inbatch_softmax_fwd_bwd_f16x2(16, 16, 32768, 128, 1.0, 0.0, 1.0, 16, 16, 16, 16, 16, 1 << 20,
                                                 None) == EINVAL
    assert b"at most 16384" in lib.esr_last_error()
    assert lib.esr_inbatch_softmax_fwd_bwd_f16x2(16, 16, 256, 136, 1.0, 0.0, 1.0, 16, 16, 16, 16, 16, 1 << 20,
                                                 None) == EINVAL   # D <= 128, a multiple of 4
    assert lib.esr_inbatch_softmax_fwd_bwd_f16x2(16, 16, 256, 98, 1.0, 0.0, 1.0, 16, 16, 16, 16, 16, 1 << 20,
                                                 None) == EINVAL
    assert lib.esr_inbatch_softmax_fwd_bwd_f16x2(16, 16, 256, 64, 1.0, 0.0, 1.0, 16, 16, 16, 16, 16, 1 << 10,
                                                 None) == EWORKSPACE  # narrower rows are accepted (zero-padded tiles)
    assert lib.esr_inbatch_softmax_fwd_bwd_f16x2(16, 16, 256, 128, 1.0, 0.0, 1.0, 16, 16, 16, 16, 16, 1 << 10,
                                                 None) == EWORKSPACE
    assert lib.esr_inbatch2h_workspace_bytes(32768, 128) == 256 and lib.esr_inbatch2h_workspace_bytes(8192, 128) > 4 * 8192 * 8192
    assert lib.esr_inbatch_towers_fwd_bwd_f16x2(16, 0, 16, 10, 0, 128, 16, 16, None, None, 256, 1.0, 0.0, 1.0, 16, 16, 16,
                                                16, 16, 1 << 20, None) == EINVAL
    assert lib.esr_inbatch_softmax_fwd_bwd(16, 16, 64, 102, 1.0, 0.0, 64.0, 16, 16, 16, 16, 16, 1 << 20, None) == EINVAL
    assert lib.esr_inbatch_softmax_fwd_bwd(16, 16, 64, 516, 1.0, 0.0, 64.0, 16, 16, 16, 16, 16, 1 << 20, None) == EINVAL
    assert lib.esr_dense_adam(16, 16, 16, 16, 8, 1e-3, 0.9, 0.999, 1e-8, 0, None) == EINVAL  # step must be >= 1
    # workspace too small is reported before any launch
    assert lib.esr_glove_fwd_bwd(16, 16, 10, 4, 16, 16, 8, 0, 16, None, None, 16, 8, None) == EWORKSPACE
    assert lib.esr_score_topk(16, 16, 1, 10, 4, 11, 16, 16, 16, 1 << 20, None) == EINVAL     # k > N


def test_exchange_entry_points_validate_before_touching_rccl(lib):
    """8e exchange in the C ABI: bad arguments come back as ESR_EINVAL with a message, before any group is opened."""
    EINVAL, ENODEVICE = -1, -4
    cnt = (ctypes.c_int64 * 2)(1, 1)
    assert lib.esr_alltoall_ids(None, 16, cnt, 16, cnt, None) == EINVAL
    assert b"null communicator" in lib.esr_last_error()
    assert lib.esr_alltoall_rows(None, 16, 7, 128, cnt, 16, cnt, None) == EINVAL
    assert b"dtype" in lib.esr_last_error()
    assert lib.esr_alltoall_grads(None, 16, 0, cnt, 16, cnt, None) == EINVAL
    assert lib.esr_comm_init(None, 2, 0, None) == EINVAL
    uid = (ctypes.c_byte * 128)()
    out = ctypes.c_void_p()
    assert lib.esr_comm_init(uid, 2, 5, ctypes.byref(out)) == EINVAL and b"rank" in lib.esr_last_error()
    assert lib.esr_comm_count(None, None, None) == EINVAL
    assert lib.esr_comm_destroy(None) == 0 and lib.esr_comm_abort(None) == 0     # freeing nothing is fine
    assert lib.esr_check_ids(None, -1, 10, None, None) == EINVAL


def test_workspace_queries_are_monotone(lib):
    a = lib.esr_glove_workspace_bytes(1024)
    b = lib.esr_glove_workspace_bytes(65536)
    assert 0 < a < b
    assert lib.esr_segment_sort_workspace_bytes(1 << 20) >= lib.esr_segment_sort_workspace_bytes(1 << 10) > 0
    assert lib.esr_inbatch_workspace_bytes(8192, 128) > 8192 * 4


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from esrecsys_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.EsrLibraryError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "nope.so"))


def test_ops_refuse_cpu_tensors():
    import torch
    from esrecsys_amd import ops
    with pytest.raises(TypeError, match="no CPU fallback"):
        ops.gather_rows(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int32))


def test_io_library_exports_its_header():
    """libesr_io.so (host-side input decoder, gcc) exports what include/esr_io.h declares"""
    from esrecsys_amd.build import build_io_library
    text = open(os.path.join(ROOT, "include", "esr_io.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(esr_[a-z0-9_]+)\s*\(", text)))
    assert declared == ["esr_cooccur_decode_lines", "esr_io_version"]
    lib = ctypes.CDLL(build_io_library())
    for name in declared:
        assert hasattr(lib, name)
    assert lib.esr_io_version() >= 100


def test_loopback_wire_exports_what_esr_comm_binds():
    """tests/wire's loopback wire (TEST INFRASTRUCTURE: world > 1 on a one-GPU box) stands in for librccl through
    ESR_RCCL_LIB: it must export exactly the symbols esr_comm.hip looks up (ESR_SYM list), no more of RCCL's surface."""
    import importlib.util
    import subprocess
    text = open(os.path.join(ROOT, "esrecsys_amd", "csrc", "esr_comm.hip")).read()
    bound = sorted(set(re.findall(r'ESR_SYM\(\w+,\s*\w+,\s*"(nccl\w+)"\)', text)))
    assert len(bound) == 12
    spec = importlib.util.spec_from_file_location("build_wire", os.path.join(ROOT, "tests", "wire", "build_wire.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib_path = mod.build()
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("nccl"))
    assert exported == bound
    # and nothing under esrecsys_amd/ names it: the product binds torch's librccl unless ESR_RCCL_LIB says otherwise
    for root, _dirs, files in os.walk(os.path.join(ROOT, "esrecsys_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c")):
                assert "loopback_wire" not in open(os.path.join(root, f), errors="ignore").read(), os.path.join(root, f)


def test_trace_markers_switch(lib):
    """SURVEY section 5's tracing row: esr_trace_markers binds a roctx library at run time (ROCm images carry one) and
    turns the launch-site / step-entry ranges on and off; with markers on, an entry point that fails validation still pops
    the range it pushed (a scope object), i.e. the call simply returns its error code."""
    rc = lib.esr_trace_markers(1)
    assert rc in (0, -4), rc   # ESR_OK, or ESR_ENODEVICE on a box without any roctx library
    if rc == 0:
        assert lib.esr_triplet_plan(None, 0, 0, 0, None, None, None, None, 0, None) != 0   # validated, no device touched
        assert lib.esr_trace_markers(0) == 0
