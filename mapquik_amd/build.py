"""Builds the HIP shared library in-tree (mapquik_amd/lib/libmapquik_hip.so) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in CI containers too.  The .so is git-ignored but travels
to the GPU box with the source snapshot.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libmapquik_hip.so")
SOURCES = [os.path.join(CSRC, "mq_capi.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "mq_device.hpp"), os.path.join(os.path.dirname(_HERE), "include", "mapquik_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def is_fresh():
    return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)


def build(force=False, verbose=False):
    if not force and is_fresh():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-align-mismatch", "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
