"""CPU-only checks of the boundary: the C-ABI library builds/loads and exports every symbol include/mapquik_hip.h
declares, record layouts match the header, and the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import mapquik_amd
    return mapquik_amd.load_library()


def _declared_functions(headers=("mapquik_hip.h", "mapquik_hip_diag.h")):
    names = set()
    for h in headers:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(mq_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_seam_header_holds_no_diagnostics():
    """include/mapquik_hip.h is the seam INTEGRATION.md mirrors; probe statistics, stage clocks and launch timers live in
    include/mapquik_hip_diag.h.  Together they are everything the library exports under the mq_ prefix."""
    seam, diag = set(_declared_functions(("mapquik_hip.h",))), set(_declared_functions(("mapquik_hip_diag.h",)))
    assert not (seam & diag)
    assert diag == {"mq_last_map_path_counts", "mq_map_probe_stats", "mq_probe_rate", "mq_last_stage_clocks", "mq_last_map_ms", "mq_ctx_last_map_ms", "mq_last_read_cycles", "mq_last_map_order", "mq_map_launch_waves", "mq_index_table_alloc_ms"}
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in sorted(seam):
        assert name in integ, "INTEGRATION.md does not bind %s" % name
    import subprocess
    import mapquik_amd
    out = subprocess.run(["nm", "-D", "--defined-only", mapquik_amd.build.LIB], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("mq_")}
    assert exported == seam | diag, sorted(exported ^ (seam | diag))


def test_every_declared_symbol_is_exported(lib):
    import mapquik_amd
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), "libmapquik_hip.so lacks %s" % name
    assert sorted(mapquik_amd.api.EXPORTS) == declared, "api.EXPORTS out of sync with the header"


def test_abi_version_and_defaults(lib):
    import mapquik_amd
    # 3: mq_hit carries columns 3 and 4 as 64 bits (48-byte records); 4: FASTA pieces on the device, mq_index_reserve, page-lock helpers,
    # the diagnostics' own header (mq_last_stage_clocks fills 16 values), seeding variants in mq_params.flags
    assert lib.mq_abi_version() == 4
    p = mapquik_amd.Params()
    q = mapquik_amd.Params(0, 0, 0.0, False, 0, 0, 0)
    lib.mq_params_default(C.byref(q))
    assert (q.k, q.l, q.density, q.use_hpc, q.c, q.s, q.g) == (5, 31, 0.01, 1, 4, 11, 2000)  # src/main.rs:174-188
    assert bytes(p) == bytes(q)
    assert C.sizeof(mapquik_amd.Params) == 40


def test_record_layouts_match_header():
    import mapquik_amd
    assert mapquik_amd.hit_dtype.itemsize == 48 and mapquik_amd.kminmer_dtype.itemsize == 24
    assert [n for n in mapquik_amd.hit_dtype.names] == ["status", "ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end",
                                                       "score", "n_kminmers", "q_start_hi", "q_end_hi"]


def test_no_cpu_fallback_without_gpu(lib):
    import mapquik_amd
    if lib.mq_device_count() > 0:
        pytest.skip("a GPU is visible: the failure path is not reachable here")
    with pytest.raises(mapquik_amd.MapquikError) as e:
        mapquik_amd.Index()
    assert "no HIP device" in str(e.value) or "CPU fallback" in str(e.value)
    with pytest.raises(mapquik_amd.MapquikError):
        mapquik_amd.find_matches("r", 4, b"ACGT", None, mapquik_amd.Index(), mapquik_amd.Params())


def test_invalid_params_rejected(lib):
    import mapquik_amd
    for bad in (dict(l=0), dict(l=65), dict(k=0), dict(k=33)):
        P = mapquik_amd.Params(**bad)
        h = lib.mq_index_new(C.byref(P), 0)
        assert not h
        assert b"k/l" in lib.mq_last_error() or b"unsupported" in lib.mq_last_error()


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mapquik_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("C oracle", ""), "%s mentions oracle/" % f
    hdr = open(os.path.join(ROOT, "include", "mapquik_hip.h")).read()
    assert "oracle" not in hdr


def test_shard_bounds_cover_in_order():
    from mapquik_amd.shard import shard_bounds, shard_reads
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            got = [shard_bounds(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1
    bases = np.arange(100, dtype=np.uint8)
    offs = np.array([0, 10, 10, 35, 100], dtype=np.uint64)
    b, o, lo = shard_reads(bases, offs, 2, 1)
    assert lo == 2 and o.tolist() == [0, 25, 90] and b[0] == 10 and b.size == 90


def test_makefile_rebuilds_the_library_for_every_part():
    """`make` must rebuild libmapquik_hip.so when ANY part of its one translation unit (or either header) changes, and the driver when
    any host header does: a stale .so would make parity and perf numbers describe old kernels."""
    import glob
    import subprocess
    parts = sorted(glob.glob(os.path.join(ROOT, "mapquik_amd", "csrc", "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert len(parts) >= 12
    for f in parts:
        r = subprocess.run(["make", "-n", "-W", os.path.relpath(f, ROOT), "mapquik_amd/lib/libmapquik_hip.so"], capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0 and "hipcc" in r.stdout and "mq_capi.hip" in r.stdout, (f, r.stdout[-300:], r.stderr[-300:])
    for f in sorted(glob.glob(os.path.join(ROOT, "mapquik_amd", "csrc", "host", "*.hpp"))):
        r = subprocess.run(["make", "-n", "-W", os.path.relpath(f, ROOT), "mapquik_amd/lib/mapquik"], capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0 and "mapquik_main.cc" in r.stdout, (f, r.stdout[-300:])
