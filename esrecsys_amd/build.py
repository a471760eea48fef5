"""Builds libesr_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo.

Usage: python -m esrecsys_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libesr_hip.so")
SOURCES = ["esr_core.hip", "esr_glove.hip", "esr_triplet.hip", "esr_inbatch.hip", "esr_inbatch3.hip", "esr_inbatch2h.hip", "esr_optim.hip", "esr_sort.hip",
           "esr_retrieve.hip", "esr_spotify.hip", "esr_comm.hip", "esr_triplet_step.hip", "esr_shard.hip", "esr_shard_step.hip", "esr_ivf.hip"]
HEADERS = [os.path.join(HERE, "..", "include", "esr_probe.h"), os.path.join(CSRC, "esr_common.h"), os.path.join(CSRC, "esr_versioned.h"), os.path.join(CSRC, "esr_inbatch_mfma.h"),
           os.path.join(HERE, "..", "include", "esr_hip.h")]
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + CFLAGS + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def build_library(force=False, verbose=False):
    """Compile every .hip translation unit for gfx950 and link libesr_hip.so. Returns its path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
        if os.path.exists(LIB_PATH):
            os.remove(LIB_PATH)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB_PATH, objs):
        # No rpath on purpose: in a torch process libamdhip64.so.7 is already loaded (torch's bundled
        # copy) and the loader binds to it by soname, so device pointers are shared with torch.
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


PROBE_LIB_PATH = os.path.join(HERE, "libesr_probe.so")


def build_probe_library(force=False, verbose=False):
    """The measurement probes (esr_probe.hip: a register-only MFMA loop, an HBM read stream) as their OWN shared object:
    they are what bench.py measures ceilings with, not part of the product ABI.  The error helpers they call are
    libesr_hip.so's (loaded first, RTLD_GLOBAL: esrecsys_amd/_lib.py load_probe)."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    obj = _compile("esr_probe.hip") if not force else None
    if obj is None:
        o = os.path.join(OBJ_DIR, "esr_probe.o")
        if os.path.exists(o):
            os.remove(o)
        obj = _compile("esr_probe.hip")
    if force or _stale(PROBE_LIB_PATH, [obj]):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", PROBE_LIB_PATH, obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", PROBE_LIB_PATH)
    return PROBE_LIB_PATH


IO_LIB_PATH = os.path.join(HERE, "libesr_io.so")


def build_io_library(force=False, verbose=False):
    """Compile the host-side input decoder (plain C, gcc) into libesr_io.so.  Returns its path."""
    src = os.path.join(CSRC, "esr_io.c")
    if force or _stale(IO_LIB_PATH, [src]):
        cmd = [os.environ.get("CC", "gcc"), "-O3", "-std=c11", "-fPIC", "-shared", "-Wall", "-o", IO_LIB_PATH, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed for esr_io.c:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", IO_LIB_PATH)
    return IO_LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
    build_probe_library(force="--force" in sys.argv, verbose=True)
    build_io_library(force="--force" in sys.argv, verbose=True)
