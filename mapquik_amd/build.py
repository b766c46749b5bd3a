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
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")


def _deps():
    """Every file the library is compiled from: csrc/*.hip, csrc/*.hpp, include/*.h (a stale .so would make parity and
    perf numbers describe an old kernel)."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h")))


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


HOST_DIR = os.path.join(CSRC, "host")
CLI = os.path.join(LIBDIR, "mapquik")
CLI_SRC = os.path.join(HOST_DIR, "mapquik_main.cc")


def _cli_deps():
    import glob
    return sorted(glob.glob(os.path.join(HOST_DIR, "*.cc")) + glob.glob(os.path.join(HOST_DIR, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h")))


def build_cli(force=False, verbose=False):
    """The native `mapquik` driver (C++ host mirror over the C ABI): g++ -lz, linked against the in-tree library."""
    build(force=False, verbose=verbose)
    if not force and os.path.exists(CLI) and all(os.path.getmtime(CLI) >= os.path.getmtime(d) for d in _cli_deps() + [LIB]):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", CLI, CLI_SRC, "-L" + LIBDIR, "-lmapquik_hip", "-lz", "-lpthread", "-ldl",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + "/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


FEEDER_DUMP = os.path.join(LIBDIR, "feeder_dump")


def build_feeder_dump(force=False):
    """Test tool for the FASTX feeder (host only, no GPU): lib/feeder_dump."""
    src = os.path.join(HOST_DIR, "feeder_dump.cc")
    deps = [src] + [os.path.join(HOST_DIR, h) for h in ("fastx_feeder.hpp", "fastx_records.hpp", "par_gzip.hpp", "ref_loader.hpp")]
    if not force and os.path.exists(FEEDER_DUMP) and all(os.path.getmtime(FEEDER_DUMP) >= os.path.getmtime(d) for d in deps):
        return FEEDER_DUMP
    os.makedirs(LIBDIR, exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", FEEDER_DUMP, src, "-lz", "-lpthread", "-ldl"])
    return FEEDER_DUMP


SEAM_TEST = os.path.join(LIBDIR, "seam_test")


def build_seam_test(force=False):
    """Test program for the C++ mirror of the reference's seam (tests/cpp/seam_test.cc over mapquik_host.hpp): lib/seam_test."""
    build(force=False)
    src = os.path.join(os.path.dirname(_HERE), "tests", "cpp", "seam_test.cc")
    deps = [src, os.path.join(HOST_DIR, "mapquik_host.hpp"), LIB]
    if not force and os.path.exists(SEAM_TEST) and all(os.path.getmtime(SEAM_TEST) >= os.path.getmtime(d) for d in deps):
        return SEAM_TEST
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", SEAM_TEST, src, "-L" + LIBDIR, "-lmapquik_hip", "-lpthread", "-ldl",
                           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + "/opt/rocm/lib"])
    return SEAM_TEST


def is_fresh():
    return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _deps())


def build(force=False, verbose=False):
    if not force and is_fresh():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # -amdgpu-atomic-optimizer-strategy=None: the optimizer rewrites `if (lane == 0) r = atomicAdd(counter, 1)` into a wave reduction
    # whose result is read back (s_waitcnt vmcnt(0) + v_readfirstlane) right behind the atomic -- which turns map_kernel's prefetch of
    # the next work item (issued a whole seed phase before it is needed) into a memory round trip every wave waits for, per read
    # -pragma-unroll-threshold: the loop over a chunk's lane-batches (mq_device.hpp map_seeds) holds one tuple hash per supported k;
    # written on 32-bit halves its unrolled size passes the default 16384, and left rolled its per-batch arrays are indexed dynamically
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-align-mismatch",
           "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-mllvm", "-pragma-unroll-threshold=65536", "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
