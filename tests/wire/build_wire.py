"""Builds tests/wire/libesr_loopback_wire.so (TEST INFRASTRUCTURE: see loopback_wire.cpp).  hipcc links the HIP runtime
the wire's copies need; the only device code is the one-wave kernel with which an enqueued group (ESR_WIRE_ASYNC=1)
holds its stream until the bytes have moved."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "loopback_wire.cpp")
LIB = os.path.join(HERE, "libesr_loopback_wire.so")


def build(force=False, verbose=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-x", "hip", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread", "-o", LIB,
               SRC]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for loopback_wire.cpp:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
