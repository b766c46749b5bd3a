"""GPU: FASTA records found on the device (mq_ctx_submit_fasta / mq_ctx_wait_fasta, mapquik_amd/csrc/mq_fastx.hpp) against the
host parser and the oracle: what closures.rs:100-123 hands to find_matches (id, sequence) must not depend on who found the
record.  Also the native driver's FASTA -> PAF with the records found on the device, on the host, and through the irregular-chunk
fallback (sequences over several lines), all byte-identical to the oracle's PAF."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mq():
    import mapquik_amd
    if mapquik_amd.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return mapquik_amd


@pytest.fixture(scope="module")
def world(mq, oracle, simlib):
    g, off, names = simlib.make_genome([700000, 400000], seed=91, repeat_frac=0.1, tandem_frac=0.02)
    P, po = mq.Params(fold_case=True), oracle.params()
    ix, ox = mq.Index(P), oracle.Index()
    for r in range(2):
        s = g[int(off[r]):int(off[r + 1])]
        assert ix.add_ref(r, names[r], s) == ox.add_ref(r, names[r], s, po)
    assert ix.finalize() == ox.count()
    reads = simlib.make_reads(g, off, 700, seed=12, len_mean=9000, len_sd=5000, len_min=1)
    return dict(ix=ix, ox=ox, po=po, reads=reads, names=simlib.read_names(reads, names))


def _fasta(world, nl=b"\n", final_newline=True, lower_every=0, extra_empty=False):
    rd, rn = world["reads"], world["names"]
    offs = rd["offsets"]
    parts, seqs = [], []
    for i, n in enumerate(rn):
        s = rd["bases"][int(offs[i]):int(offs[i + 1])].tobytes()
        if lower_every and i % lower_every == 0:
            s = s.lower()
        if extra_empty and i == 5:
            parts.append(b">empty some description" + nl + b"" + nl)
            seqs.append(b"")
        parts.append(b">" + n.encode() + b" len=%d" % len(s) + nl + s + nl)
        seqs.append(s)
    txt = b"".join(parts)
    if not final_newline:
        txt = txt[:-len(nl)]
    return txt, seqs


@pytest.mark.parametrize("nl,final_newline", [(b"\n", True), (b"\r\n", True), (b"\n", False), (b"\r\n", False)])
def test_records_found_on_the_device(mq, world, nl, final_newline):
    txt, seqs = _fasta(world, nl, final_newline, lower_every=3, extra_empty=True)
    ix = world["ix"]
    # what the host-parsed path gives for the same sequences
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    want = ix.map_batch(bases, offs)
    for pad in (0, 37):  # the records need not start the buffer (a reader leaves the byte before its first record in front)
        buf = np.frombuffer(b"x" * (pad - 1) + b"\n" if pad else b"", dtype=np.uint8)
        buf = np.concatenate([buf, np.frombuffer(txt, dtype=np.uint8)])
        ctx = ix.context()
        ctx.submit_fasta(buf, begin=pad)
        hits, lines, flags = ctx.wait_fasta()
        ctx.close()
        assert flags == 0 and hits.size == len(seqs) and lines.size == 2 * len(seqs)
        assert np.array_equal(hits.view(np.uint8), want.view(np.uint8))
        # the line ends are where a host parser finds them
        pos = [pad + i for i, c in enumerate(txt) if c == 0x0A]
        if not final_newline:
            pos.append(pad + len(txt))
        assert lines.tolist() == pos
    assert (want["status"] == 1).sum() > 500


def test_irregular_pieces_are_handed_back(mq, world):
    ix = world["ix"]
    txt, _ = _fasta(world)
    recs = txt.split(b">")[1:]
    ctx = ix.context()
    # a sequence over several lines
    h, s = recs[0].split(b"\n")[:2]
    multi = b">" + h + b"\n" + s[:60] + b"\n" + s[60:] + b"\n" + b">".join([b""] + recs[1:20])
    # a blank line between records; a header without its sequence line; a piece that does not start with '>'
    blank = b">" + recs[0] + b"\n>" + recs[1]
    nohdr = b">" + recs[0] + b">lonely header\n"
    nogt = recs[0]
    for piece in (multi, blank, nohdr, nogt):
        ctx.submit_fasta(np.frombuffer(piece, dtype=np.uint8))
        hits, lines, flags = ctx.wait_fasta()
        assert flags & 1 and hits.size == 0
    # and the context is as good as new afterwards
    ok = b">" + b">".join(recs[:50])
    ctx.submit_fasta(np.frombuffer(ok, dtype=np.uint8))
    hits, lines, flags = ctx.wait_fasta()
    assert flags == 0 and hits.size == 50
    ctx.close()


def test_more_line_ends_than_the_scan_holds(mq, world, tmp_path):
    """A well-formed FASTA of very short records (primers, barcodes, k-mers: under ~32 bytes each) has more line ends than the scan's
    list holds (bytes / 16 + 4096): the piece must come back IRREGULAR with NO record reported -- never a record count that indexes
    past the list -- and the driver then parses it on the host: the same PAF (no line: nothing that short maps) and exit code 0.
    Junk input full of newlines likewise."""
    ix = world["ix"]
    ctx = ix.context()
    for piece in (b">a\nAC\n" * 40000, b">a\nAC\n" * 1500000, b"\n" * 300000, b">x\n" + b"\n" * 200001):
        ctx.submit_fasta(np.frombuffer(piece, dtype=np.uint8))
        hits, lines, flags = ctx.wait_fasta()
        assert flags & 1 and hits.size == 0 and lines.size == 0
    # just under the capacity: regular, every record reported, nothing maps
    n = 2400
    piece = b">a\nAC\n" * n  # 2 n = 4,800 line ends <= 6 n / 16 + 4096 = 4,996
    ctx.submit_fasta(np.frombuffer(piece, dtype=np.uint8))
    hits, lines, flags = ctx.wait_fasta()
    assert flags == 0 and hits.size == n and lines.size == 2 * n and not (hits["status"] != 0).any()
    ctx.close()
    # through the driver: short records in front of real reads, one chunk and many
    from mapquik_amd import build
    exe = build.build_cli()
    from tools import sim
    g, off, names = sim.make_genome([700000, 400000], seed=91, repeat_frac=0.1, tandem_frac=0.02)
    ref = tmp_path / "ref.fa"
    with open(ref, "wb") as w:
        for r in range(2):
            w.write(b">" + names[r].encode() + b"\n" + g[int(off[r]):int(off[r + 1])].tobytes() + b"\n")
    txt, _ = _fasta(world)
    rd = tmp_path / "short_then_reads.fa"
    rd.write_bytes(b">p\nACGTACGT\n" * 60000 + txt)
    outs = []
    for chunk, env in (("33554432", {}), ("200000", {}), ("33554432", {"MQ_DRIVER_HOST_PARSE": "1"})):
        prefix = str(tmp_path / ("s%d" % len(outs)))
        r = subprocess.run([exe, str(rd), "--reference", str(ref), "-p", prefix, "--batch-bases", chunk, "--threads", "3"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(open(prefix + ".paf").read())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 10000


def _fastq(world, nl=b"\n", final_newline=True, nasty_quals=True, lower_every=0):
    """The world's reads as four-line FASTQ; nasty_quals: quality lines that begin with '@' or '+' (legal quality characters: what the
    host's validator needs its look-ahead for) and a '+' line that repeats the id."""
    rd, rn = world["reads"], world["names"]
    offs = rd["offsets"]
    parts, seqs = [], []
    for i, n in enumerate(rn):
        s = rd["bases"][int(offs[i]):int(offs[i + 1])].tobytes()
        if lower_every and i % lower_every == 0:
            s = s.lower()
        q = bytearray(b"I" * len(s))
        if nasty_quals and q:
            q[0] = ord("@") if i % 3 == 0 else ord("+") if i % 3 == 1 else ord("I")
            if len(q) > 5:
                q[5] = ord(">")
        plus = b"+" + (n.encode() if i % 4 == 0 else b"")
        parts.append(b"@" + n.encode() + b" np:i:%d" % (i % 30) + nl + s + nl + plus + nl + bytes(q) + nl)
        seqs.append(s)
    txt = b"".join(parts)
    if not final_newline:
        txt = txt[:-len(nl)]
    return txt, seqs


@pytest.mark.parametrize("nl,final_newline", [(b"\n", True), (b"\r\n", True), (b"\n", False), (b"\r\n", False)])
def test_fastq_records_found_on_the_device(mq, world, nl, final_newline):
    """mq_ctx_submit_fastx(MQ_FASTX_FASTQ): record r = lines 4r .. 4r + 3, checked on the device ('@', '+', one quality per base); the
    hits are the host-parsed path's, the line ends are where a host parser finds them; quality lines beginning with '@' / '+' and
    '+id' separator lines change nothing."""
    txt, seqs = _fastq(world, nl, final_newline, lower_every=3)
    ix = world["ix"]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    want = ix.map_batch(bases, offs)
    for pad in (0, 37):
        buf = np.frombuffer(b"x" * (pad - 1) + b"\n" if pad else b"", dtype=np.uint8)
        buf = np.concatenate([buf, np.frombuffer(txt, dtype=np.uint8)])
        ctx = ix.context()
        ctx.submit_fasta(buf, begin=pad, fastq=True)
        hits, lines, flags = ctx.wait_fasta()
        ctx.close()
        assert flags == 0 and hits.size == len(seqs) and lines.size == 4 * len(seqs)
        assert np.array_equal(hits.view(np.uint8), want.view(np.uint8))
        pos = [pad + i for i, c in enumerate(txt) if c == 0x0A]
        if not final_newline:
            pos.append(pad + len(txt))
        assert lines.tolist() == pos
    assert (want["status"] == 1).sum() > 500


def test_irregular_fastq_pieces_are_handed_back(mq, world):
    ix = world["ix"]
    txt, seqs = _fastq(world, nasty_quals=False)
    recs = txt.split(b"\n@")
    recs = [recs[0]] + [b"@" + r for r in recs[1:]]          # whole records (no quality begins with '@' here), the last with its newline
    recs = [r if r.endswith(b"\n") else r + b"\n" for r in recs]
    h, s, p, q = recs[0].split(b"\n")[:4]
    ctx = ix.context()
    bad = [
        h + b"\n" + s[:50] + b"\n" + s[50:] + b"\n" + p + b"\n" + q + b"\n" + b"".join(recs[1:10]),   # a sequence over two lines
        h + b"\n" + s + b"\n" + p + b"\n" + q[:-1] + b"\n" + b"".join(recs[1:10]),                   # one quality short
        h + b"\n" + s + b"\n-\n" + q + b"\n" + b"".join(recs[1:10]),                                 # no '+' line
        b">" + h[1:] + b"\n" + s + b"\n" + p + b"\n" + q + b"\n",                                     # not '@'
        recs[0] + b"\n" + recs[1],                                                                    # a blank line between records
        recs[0] + h + b"\n" + s + b"\n" + p + b"\n",                                                  # the last record cut short
        recs[0] + h + b"\n" + s + b"\n+\n",                                                           # ... with an empty quality line (3 lines + virtual end)
    ]
    for piece in bad:
        ctx.submit_fasta(np.frombuffer(piece, dtype=np.uint8), fastq=True)
        hits, lines, flags = ctx.wait_fasta()
        assert flags & 1 and hits.size == 0, piece[:60]
    ok = b"".join(recs[:50])
    ctx.submit_fasta(np.frombuffer(ok, dtype=np.uint8), fastq=True)
    hits, lines, flags = ctx.wait_fasta()
    assert flags == 0 and hits.size == 50 and lines.size == 200
    # an empty read is a record too (sequence and quality lines both empty)
    ctx.submit_fasta(np.frombuffer(b"@e\n\n+\n\n" + recs[0], dtype=np.uint8), fastq=True)
    hits, lines, flags = ctx.wait_fasta()
    assert flags == 0 and hits.size == 2 and hits["status"][0] == 0
    ctx.close()


def test_native_driver_fastq_device_and_host_parse(mq, oracle, world, tmp_path):
    """FASTQ file -> PAF through the native driver, three readers: the lean one (default: header and sequence lines read with one pread per
    record, qualities never read), records found on the device (MQ_DRIVER_FASTQ=device: the reader threads never look at a base or a
    quality, the whole file crosses the link), and the chunked reader + host parser (MQ_FEEDER_NO_LEAN_FASTQ=1 with MQ_DRIVER_HOST_PARSE=1);
    CR-LF without a final newline, quality lines that begin with '@' / '+', '+id' separator lines -- the oracle's PAF every time, at chunk
    sizes that put boundaries everywhere."""
    from mapquik_amd import build
    exe = build.build_cli()
    ox, po, rd, rn = world["ox"], world["po"], world["reads"], world["names"]
    from tools import sim
    g, off, names = sim.make_genome([700000, 400000], seed=91, repeat_frac=0.1, tandem_frac=0.02)
    ref = tmp_path / "ref.fa"
    with open(ref, "wb") as w:
        for r in range(2):
            w.write(b">" + names[r].encode() + b"\n" + g[int(off[r]):int(off[r + 1])].tobytes() + b"\n")
    want = ox.map_batch(rd["bases"], rd["offsets"], po, threads=4)
    want_txt = "".join(x + "\n" for x in oracle.paf_lines(ox, rn, want))
    plain = tmp_path / "reads.fastq"
    plain.write_bytes(_fastq(world)[0])
    crlf = tmp_path / "reads_crlf.fq"
    crlf.write_bytes(_fastq(world, b"\r\n", False)[0])
    k = 0
    for path in (plain, crlf):
        for chunk in ("30000", "400000", "33554432"):
            for env in ({}, {"MQ_DRIVER_FASTQ": "device"}, {"MQ_DRIVER_HOST_PARSE": "1", "MQ_FEEDER_NO_LEAN_FASTQ": "1"}):
                k += 1
                prefix = str(tmp_path / ("q%d" % k))
                r = subprocess.run([exe, str(path), "--reference", str(ref), "-p", prefix, "--batch-bases", chunk, "--threads", "3", "--unmapped"],
                                   capture_output=True, text=True, env=dict(os.environ, MQ_DRIVER_TIMING="1", **env))
                assert r.returncode == 0, r.stderr
                assert open(prefix + ".paf").read() == want_txt, (str(path), chunk, env)
                assert open(prefix + ".unmapped.out").read().split() == [n for n, w_ in zip(rn, want) if not w_["mapped"]]


def test_native_driver_fasta_device_and_host_parse(mq, oracle, world, tmp_path):
    """FASTA file -> PAF through the native driver: records found on the device (default), parsed by the reader threads
    (MQ_DRIVER_HOST_PARSE=1), and a multi-line FASTA whose chunks all come back irregular -- the oracle's PAF every time, at chunk
    sizes that put boundaries everywhere."""
    from mapquik_amd import build
    exe = build.build_cli()
    ox, po, rd, rn = world["ox"], world["po"], world["reads"], world["names"]
    g_names = ["chr1", "chr2"]
    # reference file from the oracle's own sequences is not kept: write the genome again
    from tools import sim
    g, off, names = sim.make_genome([700000, 400000], seed=91, repeat_frac=0.1, tandem_frac=0.02)
    ref = tmp_path / "ref.fa"
    with open(ref, "wb") as w:
        for r in range(2):
            w.write(b">" + names[r].encode() + b"\n" + g[int(off[r]):int(off[r + 1])].tobytes() + b"\n")
    want = ox.map_batch(rd["bases"], rd["offsets"], po, threads=4)
    want_txt = "".join(x + "\n" for x in oracle.paf_lines(ox, rn, want))
    assert len(want_txt) > 10000
    txt, _ = _fasta(world)
    single = tmp_path / "reads.fa"
    single.write_bytes(txt)
    crlf = tmp_path / "reads_crlf.fa"
    crlf.write_bytes(_fasta(world, b"\r\n", False)[0])
    multi = tmp_path / "reads_multi.fa"
    with open(multi, "wb") as w:
        offs = rd["offsets"]
        for i, n in enumerate(rn):
            s = rd["bases"][int(offs[i]):int(offs[i + 1])].tobytes()
            w.write(b">" + n.encode() + b" d\n")
            for j in range(0, max(len(s), 1), 70):
                w.write(s[j:j + 70] + b"\n")
    k = 0
    for path in (single, crlf, multi):
        for chunk in ("20000", "300000", "33554432"):
            for env in ({}, {"MQ_DRIVER_HOST_PARSE": "1"}):
                k += 1
                prefix = str(tmp_path / ("o%d" % k))
                r = subprocess.run([exe, str(path), "--reference", str(ref), "-p", prefix, "--batch-bases", chunk, "--threads", "3", "--unmapped"],
                                   capture_output=True, text=True, env=dict(os.environ, **env))
                assert r.returncode == 0, r.stderr
                assert open(prefix + ".paf").read() == want_txt, (str(path), chunk, env)
                assert open(prefix + ".unmapped.out").read().split() == [n for n, w_ in zip(rn, want) if not w_["mapped"]]
