"""GPU: the reference phase of the native driver -- the file streamed to the device in pieces (mq_index_stage_*, RefStreamer) against
the host loader (MQ_DRIVER_REF_HOST=1) and the oracle; the shapes that send the streamer back to the host loader; the on-disk index
from the command line (--save-index / --index, both drivers).  Reference behaviour: src/closures.rs:24-94 (the index phase and its
println!s), src/main.rs:60-75."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mq():
    import mapquik_amd
    if mapquik_amd.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return mapquik_amd


@pytest.fixture(scope="module")
def world(mq, oracle, simlib, tmp_path_factory):
    from mapquik_amd import build
    exe = build.build_cli()
    wd = tmp_path_factory.mktemp("refstream")
    # contigs around the streamer's 16-MB block size: one that ends a byte before / at / after a block border would need 16-MB contigs;
    # three contigs of 20, 17 and 3 Mbp put line ends inside blocks 1 and 2 and a header across the border region
    g, off, names = simlib.make_genome([20_000_000, 17_000_000, 3_000_000, 40, 1200], seed=41, repeat_frac=0.05, threads=4)
    po = oracle.params()
    ox = oracle.Index()
    counts = [ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po) for r in range(len(names))]
    reads = simlib.make_reads(g, off, 1500, seed=6, threads=4)
    rn = simlib.read_names(reads, names)
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=4)
    want_txt = "".join(x + "\n" for x in oracle.paf_lines(ox, rn, want))
    rd = wd / "reads.fa"
    o = reads["offsets"]
    with open(rd, "wb") as f:
        for i, n in enumerate(rn):
            f.write(b">" + n.encode() + b"\n" + reads["bases"][int(o[i]):int(o[i + 1])].tobytes() + b"\n")
    return dict(exe=exe, wd=wd, g=g, off=off, names=names, counts=counts, unique=ox.count(), reads=str(rd), want_txt=want_txt)


def _write_ref(path, w, nl=b"\n", wrap=0, final=True, lower=False, blank_between=False, wrap_from=None):
    g, off, names = w["g"], w["off"], w["names"]
    with open(path, "wb") as f:
        for r in range(len(names)):
            s = g[int(off[r]):int(off[r + 1])].tobytes()
            if lower:
                s = s[:len(s) // 2].lower() + s[len(s) // 2:]
            f.write(b">" + names[r].encode() + b" contig %d of the test genome" % r + nl)
            if wrap and (wrap_from is None or r >= wrap_from):
                f.write(b"".join(s[i:i + wrap] + nl for i in range(0, max(len(s), 1), wrap)))
            else:
                f.write(s + (nl if (final or r + 1 < len(names)) else b""))
            if blank_between:
                f.write(nl)


def _run(w, ref, env=None, extra=()):
    prefix = str(w["wd"] / ("o%d" % np.random.randint(1 << 30)))
    r = subprocess.run([w["exe"], w["reads"], "--reference", str(ref), "-p", prefix, "--threads", "3"] + list(extra), capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, MQ_DRIVER_TIMING="1", **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    return r, open(prefix + ".paf").read()


def _index_lines(stdout):
    return [ln for ln in stdout.splitlines() if ln.startswith("Indexed reference ")]


@pytest.mark.parametrize("shape", ["plain", "crlf", "nofinal", "lower", "blank_between"])
def test_streamed_reference_equals_host_loader_and_oracle(world, shape):
    """One sequence line per record: the file goes to the device in 16-MB pieces while it is read and every record is indexed from
    there -- per-reference k-min-mer counts (the `Indexed reference` lines), the unique count and the PAF are the oracle's and the host
    loader's, with LF / CR-LF line ends, no final newline, soft-masked lower case and blank lines between records."""
    w = world
    ref = w["wd"] / ("ref_%s.fa" % shape)
    _write_ref(ref, w, nl=b"\r\n" if shape == "crlf" else b"\n", final=shape != "nofinal", lower=shape == "lower", blank_between=shape == "blank_between")
    want_lines = ["Indexed reference %s: %d k-min-mers." % (n, c) for n, c in zip(w["names"], w["counts"])]
    # default (and --low-memory): never in host memory -- streamed through a pool of page-locked 16-MB blocks
    r, paf = _run(w, ref)
    assert "reference streamed: every record handed to ref_extract" in r.stderr, r.stderr[-1500:]
    assert paf == w["want_txt"] and len(paf) > 50000
    assert _index_lines(r.stdout) == want_lines
    assert "Indexed %d unique k-min-mers in " % w["unique"] in r.stdout
    r0, paf0 = _run(w, ref, extra=["--low-memory"])
    assert "reference streamed: every record handed to ref_extract" in r0.stderr and paf0 == paf and _index_lines(r0.stdout) == want_lines
    # MQ_DRIVER_REF_PRELOAD=1: the file read into host memory while the HIP runtime comes up, page-locked, every record's bytes queued for the device
    r1, paf1 = _run(w, ref, env={"MQ_DRIVER_REF_PRELOAD": "1"})
    assert "reference buffer page-locked" in r1.stderr and "every reference record indexed" in r1.stderr, r1.stderr[-1500:]
    assert paf1 == paf and _index_lines(r1.stdout) == want_lines
    # earlier rounds' path: records copied from pageable memory one by one
    r2, paf2 = _run(w, ref, env={"MQ_DRIVER_REF_HOST": "1"})
    assert "reference streamed" not in r2.stderr and "reference buffer page-locked" not in r2.stderr and paf2 == paf and _index_lines(r2.stdout) == want_lines


def test_other_reference_shapes_fall_back_to_the_host_loader(world):
    """A line-wrapped FASTA shows in its first block (nothing was indexed yet: the index stays); a file whose LATER records are wrapped
    is noticed after records were handed over (the index is dropped and rebuilt by the host loader); text before the first '>' is the
    host loader's error.  Same PAF, same log lines."""
    w = world
    want_lines = ["Indexed reference %s: %d k-min-mers." % (n, c) for n, c in zip(w["names"], w["counts"])]
    wrapped = w["wd"] / "ref_wrapped.fa"
    _write_ref(wrapped, w, wrap=80)
    for ref_file in (wrapped,):
        # the streamer gives the file back in its first block; the loader joins the lines of every record in host memory and queues its bytes
        r, paf = _run(w, ref_file)
        assert "reference is not one line per record: host loader" in r.stderr and "reference buffer page-locked" in r.stderr
        assert paf == w["want_txt"] and _index_lines(r.stdout) == want_lines
        # --low-memory: ... through the chunked reader instead (one record in host memory at a time)
        r, paf = _run(w, ref_file, extra=["--low-memory"])
        assert "reference is not one line per record: host loader" in r.stderr and "reference buffer page-locked" not in r.stderr
        assert paf == w["want_txt"] and _index_lines(r.stdout) == want_lines
        # the whole-file loader from the start
        r, paf = _run(w, ref_file, env={"MQ_DRIVER_REF_PRELOAD": "1"})
        assert "reference buffer page-locked" in r.stderr and paf == w["want_txt"] and _index_lines(r.stdout) == want_lines
    late = w["wd"] / "ref_late_wrap.fa"
    _write_ref(late, w, wrap=70, wrap_from=2)
    for extra in ([], ["--low-memory"]):  # noticed after records were indexed: that index is dropped, the file indexed again by the fallback
        r, paf = _run(w, late, extra=extra)
        assert "reference is not one line per record: host loader" in r.stderr and paf == w["want_txt"] and _index_lines(r.stdout) == want_lines
    junk = w["wd"] / "ref_junk.fa"
    junk.write_bytes(b"this is not FASTA\n>a\nACGT\n")
    rr = subprocess.run([w["exe"], w["reads"], "--reference", str(junk), "-p", str(w["wd"] / "junk"), "--threads", "2"], capture_output=True, text=True, timeout=600)
    assert rr.returncode == 101 and "malformed FASTA record" in rr.stderr


def test_stage_api_pieces_in_any_order(mq, oracle, simlib):
    """mq_index_stage_*: pieces issued out of order, of odd sizes, from pageable and page-locked memory; records asked for as soon as their
    pieces are issued; the result is mq_index_add_ref's."""
    g, off, names = simlib.make_genome([700000, 300000, 50], seed=9, repeat_frac=0.1)
    txt = b"".join(b">" + names[r].encode() + b"\n" + g[int(off[r]):int(off[r + 1])].tobytes() + b"\n" for r in range(3))
    buf = np.frombuffer(txt, dtype=np.uint8)
    P = mq.Params()
    a = mq.Index(P)
    want = [a.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])]) for r in range(3)]
    ua = a.finalize()
    b = mq.Index(P)
    b.stage_begin(buf.size)
    cuts = [0, 1, 4097, 300001, 300002, 700123, 999999, buf.size]
    pieces = [(cuts[i], buf[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)]
    pin = mq.PinnedBuffer(max(p.size for _, p in pieces))
    tickets = []
    for at, piece in reversed(pieces):  # last piece first
        if at % 2:
            pin.array[:piece.size] = piece
            t = b.stage_piece(at, pin.array[:piece.size])
            assert b.stage_done(t, wait=True)  # the page-locked buffer is reused for the next odd piece
        else:
            t = b.stage_piece(at, piece.copy())
        tickets.append(t)
    assert tickets == list(range(len(pieces)))
    got, pos = [], 0
    for r in range(3):
        pos += len(names[r]) + 2
        n = int(off[r + 1] - off[r])
        got.append(b.add_ref_staged(r, names[r], pos, n, after_ticket=None if r != 1 else tickets[-1]))  # (every piece issued so far / behind the last one)
        pos += n + 1
    assert got == want
    assert all(b.stage_done(t, wait=True) for t in tickets)
    assert b.finalize() == ua  # (releases the staging buffer and its tickets)
    with pytest.raises(mq.MapquikError):
        b.stage_done(tickets[0], wait=False)
    reads = simlib.make_reads(g, off, 200, seed=3)
    assert np.array_equal(a.map_batch(reads["bases"], reads["offsets"]).view(np.uint8), b.map_batch(reads["bases"], reads["offsets"]).view(np.uint8))
    c = mq.Index(P)
    with pytest.raises(mq.MapquikError):
        c.stage_piece(0, buf[:10])           # before stage_begin
    c.stage_begin(100)
    with pytest.raises(mq.MapquikError):
        c.stage_piece(96, buf[:10])          # outside the buffer
    with pytest.raises(mq.MapquikError):
        c.stage_begin(100)                   # one buffer per index
    with pytest.raises(mq.MapquikError):
        c.add_ref_staged(0, "x", 50, 51)
    with pytest.raises(mq.MapquikError):
        c.add_ref_staged(0, "x", 0, 10, after_ticket=5)   # a ticket nobody was given
    for x in (a, b, c):
        x.close()
    pin.close()


def test_on_disk_index_from_the_command_line(world, mq):
    """--save-index writes the finalized index, --index maps against it without the reference (both drivers): the PAF of the
    FASTA-indexed run, the index phase's last log line kept; other seeding parameters than the file's are refused, other chaining
    thresholds are this run's."""
    w = world
    ref = w["wd"] / "ref_cli.fa"
    _write_ref(ref, w)
    ixf = str(w["wd"] / "genome.mqx")
    r, paf = _run(w, ref, extra=["--save-index", ixf])
    assert paf == w["want_txt"] and "Saved index to %s in " % ixf in r.stdout and os.path.getsize(ixf) > 1000000
    prefix = str(w["wd"] / "from_index")
    r2 = subprocess.run([w["exe"], w["reads"], "--index", ixf, "-p", prefix, "--threads", "3"], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert open(prefix + ".paf").read() == w["want_txt"]
    assert "Loaded index %s: %d references, %d k-min-mers." % (ixf, len(w["names"]), sum(w["counts"])) in r2.stdout
    assert "Indexed %d unique k-min-mers in " % w["unique"] in r2.stdout and "Reference file:" not in r2.stdout
    # the Python driver: the same file, the same PAF
    prefix_py = str(w["wd"] / "from_index_py")
    r3 = subprocess.run([sys.executable, "-m", "mapquik_amd", w["reads"], "--index", ixf, "-p", prefix_py], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r3.returncode == 0, r3.stderr[-2000:]
    assert open(prefix_py + ".paf").read() == w["want_txt"] and "Loaded index " in r3.stdout
    # other seeding parameters: refused, naming the file's
    r4 = subprocess.run([w["exe"], w["reads"], "--index", ixf, "-p", prefix, "-k", "7"], capture_output=True, text=True, timeout=600)
    assert r4.returncode == 101 and "was built with -k 5 -l 31 -d 0.01" in r4.stderr
    # chaining thresholds are this run's: -s 1000 leaves MAPQ 0 everywhere (score >= s or chain length >= c fails) -- as the FASTA-indexed run with -s 1000 -c 1000
    pa, pb = str(w["wd"] / "thr_a"), str(w["wd"] / "thr_b")
    ra = subprocess.run([w["exe"], w["reads"], "--index", ixf, "-p", pa, "-s", "1000", "-c", "1000", "--threads", "3"], capture_output=True, text=True, timeout=600)
    rb = subprocess.run([w["exe"], w["reads"], "--reference", str(ref), "-p", pb, "-s", "1000", "-c", "1000", "--threads", "3"], capture_output=True, text=True, timeout=600)
    assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr[-500:], rb.stderr[-500:])
    ta, tb = open(pa + ".paf").read(), open(pb + ".paf").read()
    assert ta == tb and ta != w["want_txt"] and all(ln.endswith("\t0") for ln in ta.splitlines())
    # a file saved by the Python driver loads in the native one
    ixf2 = str(w["wd"] / "genome_py.mqx")
    r5 = subprocess.run([sys.executable, "-m", "mapquik_amd", w["reads"], "--reference", str(ref), "--save-index", ixf2, "-p", prefix_py + "2"], capture_output=True, text=True,
                        timeout=1800, cwd=ROOT)
    assert r5.returncode == 0, r5.stderr[-2000:]
    r6 = subprocess.run([w["exe"], w["reads"], "--index", ixf2, "-p", prefix + "6", "--threads", "3"], capture_output=True, text=True, timeout=600)
    assert r6.returncode == 0 and open(prefix + "6.paf").read() == w["want_txt"]
