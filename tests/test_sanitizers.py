"""The threaded host code under AddressSanitizer + UBSan and under ThreadSanitizer (make asan / make tsan): the FASTX feeder
(feeder_dump) and the native driver's whole pipeline -- feeder readers, inflate thread, reference loader, per-GPU submit threads
with three stream slots each, formatter pool, ordered writer -- linked against a host-only stub of the C ABI with canned results
(tests/cpp/stub_mapquik_hip.cc).  The reference has a tenth of this concurrency and no sanitizer job."""
import gzip
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mapquik_amd", "lib")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", ROOT, "asan", "tsan"], capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        pytest.fail("make asan tsan failed:\n" + r.stderr[-2000:])
    return {k: os.path.join(LIB, k) for k in ("feeder_dump_asan", "feeder_dump_tsan", "mapquik_asan", "mapquik_tsan")}


def _inputs(tmp_path):
    rng = random.Random(3)
    recs = [("r%d" % i, "".join(rng.choice("ACGTacgtN") for _ in range(rng.choice([0, 30, 400, 3000, 9000])))) for i in range(400)]
    fa = "".join(">%s d\n%s\n" % (a, "\n".join(b[j:j + 70] for j in range(0, len(b), 70)) or "") for a, b in recs)
    fq = "".join("@%s d\n%s\n+\n%s\n" % (a, b, "I" * len(b)) for a, b in recs)
    ref = ">chrA x\n" + "".join(rng.choice("ACGT") for _ in range(300000)) + "\n>chrB\n" + "\n".join(
        "".join(rng.choice("ACGT") for _ in range(80)) for _ in range(2000)) + "\n"
    p = {}
    ref1 = ">chrA x\n" + "".join(rng.choice("ACGTacgt") for _ in range(300000)) + "\n>chrB\n" + "".join(rng.choice("ACGT") for _ in range(160000)) + "\n"
    for name, text in (("reads.fa", fa), ("reads.fastq", fq), ("ref.fa", ref), ("ref1.fa", ref1)):
        p[name] = str(tmp_path / name)
        with open(p[name], "w") as f:
            f.write(text)
    p["reads.fa.gz"] = str(tmp_path / "reads.fa.gz")
    with gzip.open(p["reads.fa.gz"], "wt") as f:
        f.write(fa)
    return p, recs


_ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1",
            UBSAN_OPTIONS="print_stacktrace=1")


def _clean(r):
    bad = [w for w in ("AddressSanitizer", "ThreadSanitizer", "LeakSanitizer", "runtime error:") if w in r.stderr]
    assert not bad and r.returncode == 0, (r.returncode, r.stderr[-3000:])


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_feeder_and_reference_loader_under_sanitizers(built, tmp_path, san):
    p, recs = _inputs(tmp_path)
    tool = built["feeder_dump_" + san]
    want = [(a, str(len(b))) for a, b in recs]
    for path, kind in ((p["reads.fa"], "fasta"), (p["reads.fastq"], "fastq"), (p["reads.fa.gz"], "fasta")):
        for chunk, th in ((3000, 4), (1 << 20, 3)):
            r = subprocess.run([tool, path, kind, str(chunk), str(th)], capture_output=True, text=True, timeout=300, env=_ENV)
            _clean(r)
            got = [tuple(ln.split("\t")[:2]) for ln in r.stdout.split("\n") if ln]
            assert got == want, (path, chunk, th)
    r = subprocess.run([tool, p["ref.fa"], "ref", "0", "4"], capture_output=True, text=True, timeout=300, env=_ENV)
    _clean(r)
    assert [ln.split("\t")[:2] for ln in r.stdout.split("\n") if ln] == [["chrA", "300000"], ["chrB", "160000"]]
    # a truncated gzip stream: an error exit, still no sanitizer report
    cut = str(tmp_path / "cut.fa.gz")
    blob = open(p["reads.fa.gz"], "rb").read()
    open(cut, "wb").write(blob[:len(blob) // 2])
    r = subprocess.run([tool, cut, "fasta", "3000", "4"], capture_output=True, text=True, timeout=300, env=_ENV)
    assert r.returncode != 0 and "truncated" in r.stderr and "Sanitizer" not in r.stderr
    # the many-thread member inflater (par_gzip.hpp): through the feeder and alone, whole and damaged input
    env = dict(_ENV, MQ_PARGZ_MIN="1000", MQ_PARGZ_SEG="30000", MQ_PARGZ_MINSEG="8000", MQ_FEEDER_TIMING="1")
    r = subprocess.run([tool, p["reads.fa.gz"], "fasta", "3000", "4"], capture_output=True, text=True, timeout=300, env=env)
    _clean(r)
    assert "all threads" in r.stderr and [tuple(ln.split("\t")[:2]) for ln in r.stdout.split("\n") if ln] == want
    r = subprocess.run([tool, p["reads.fa.gz"], "inflate", "20000", "5"], capture_output=True, timeout=300, env=_ENV)
    assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr
    import gzip as _gz
    assert r.stdout == _gz.decompress(blob)
    flipped = str(tmp_path / "flip.fa.gz")
    for pos in (len(blob) // 3, len(blob) // 2, len(blob) - 5):
        open(flipped, "wb").write(blob[:pos] + bytes([blob[pos] ^ 0x40]) + blob[pos + 1:])
        for args, e in ((["inflate", "20000", "4"], _ENV), (["fasta", "3000", "4"], env)):
            r = subprocess.run([tool, flipped] + args, capture_output=True, timeout=300, env=e)
            assert r.returncode != 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, (pos, args)
    r = subprocess.run([tool, cut, "inflate", "20000", "4"], capture_output=True, timeout=300, env=_ENV)
    assert r.returncode != 0 and b"Sanitizer" not in r.stderr


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_driver_pipeline_under_sanitizers(built, tmp_path, san):
    p, recs = _inputs(tmp_path)
    exe = built["mapquik_" + san]
    n_long = sum(1 for _, b in recs if len(b) >= 50)
    # ref.fa: a single-line record, then a line-wrapped one (the streamer gives the file back after it indexed the first: the index is
    # dropped, the loader joins the lines and queues the records' bytes); ref1.fa: single-line all through (the streamer's own path)
    for reads, extra, env in ((p["reads.fa"], [], {}), (p["reads.fastq"], ["--gpus", "2", "--unmapped"], {"MQ_STUB_DEVICES": "2"}),
                              (p["reads.fa.gz"], ["--threads", "3"], {}), (p["reads.fa"], [], {"MQ_DRIVER_PREFETCH": "1"}),
                              (p["reads.fa"], ["--reference", p["ref1.fa"]], {}), (p["reads.fa"], ["--reference", p["ref1.fa"], "--low-memory"], {}),
                              (p["reads.fa"], ["--low-memory"], {}), (p["reads.fa"], [], {"MQ_DRIVER_REF_PRELOAD": "1"}),
                              (p["reads.fa"], ["--reference", p["ref1.fa"]], {"MQ_DRIVER_REF_PRELOAD": "1"}), (p["reads.fa"], [], {"MQ_DRIVER_REF_HOST": "1"}),
                              (p["reads.fastq"], [], {"MQ_DRIVER_FASTQ": "device"}), (p["reads.fastq"], [], {"MQ_DRIVER_HOST_PARSE": "1", "MQ_FEEDER_NO_LEAN_FASTQ": "1"})):
        prefix = str(tmp_path / "out")
        r = subprocess.run([exe, reads, "--reference", p["ref.fa"], "-p", prefix, "--batch-bases", "20000", "--threads", "4"] + extra,
                           capture_output=True, text=True, timeout=600, env=dict(_ENV, **env))
        _clean(r)
        lines = open(prefix + ".paf").read().splitlines()
        assert len(lines) == n_long and [ln.split("\t")[0] for ln in lines] == [a for a, b in recs if len(b) >= 50]  # input order
    # a failure in the middle of the run: exit 101, no hang, no report
    r = subprocess.run([exe, p["reads.fa"], "--reference", p["ref.fa"], "-p", str(tmp_path / "f"), "--batch-bases", "20000", "--threads", "4",
                        "--gpus", "2"], capture_output=True, text=True, timeout=600, env=dict(_ENV, MQ_STUB_DEVICES="2", MQ_DRIVER_FAIL_AT="20"))
    assert r.returncode == 101 and "injected failure" in r.stderr and "Sanitizer" not in r.stderr, (r.returncode, r.stderr[-2000:])


def test_driver_does_not_deadlock_on_a_small_pool(built, tmp_path):
    """Regression: the submit thread used to WAIT for a new chunk while holding submitted ones; when the writer was waiting for
    exactly one of those and every other buffer of the pool sat behind the writer (formatted, out of turn), no new chunk could be
    parsed and the run hung (about one run in five with these sizes: 60 chunks of 20 kB through a pool of 12).  Many runs, each
    under a timeout; the TSan build widens the timing windows."""
    p, recs = _inputs(tmp_path)
    exe = built["mapquik_tsan"]
    n_long = sum(1 for _, b in recs if len(b) >= 50)
    for it in range(30):
        for env, extra in (({}, []), ({"MQ_DRIVER_PREFETCH": "1"}, []), ({"MQ_STUB_DEVICES": "2"}, ["--gpus", "2"])):
            r = subprocess.run([exe, p["reads.fa"], "--reference", p["ref.fa"], "-p", str(tmp_path / "o"), "--batch-bases", "20000", "--threads", "4"] + extra,
                               capture_output=True, text=True, timeout=90, env=dict(_ENV, **env))
            _clean(r)
        assert sum(1 for _ in open(str(tmp_path / "o.paf"))) == n_long
