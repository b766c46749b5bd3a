"""The native driver's FASTX feeder (mapquik_amd/csrc/host/fastx_feeder.hpp) against a plain Python parser: FASTA (single- and
multi-line), FASTQ, CRLF, missing final newline, gzip and lz4 input, chunk sizes from 64 bytes up, 1 and 4 threads.  Host only
(chunk buffers from malloc): what get_reader + seq_io give the reference (src/main.rs:60-75, src/closures.rs:177-187)."""
import ctypes as C
import gzip
import random
import subprocess

import pytest

from mapquik_amd import build as B


def _make(n, fastq, multiline, crlf, rng):
    recs, out = [], []
    nl = "\r\n" if crlf else "\n"
    for i in range(n):
        L = rng.choice([0, 1, 5, 50, 200, 1000, 3000])
        seq = "".join(rng.choice("ACGTacgtN") for _ in range(L))
        rid = "r%d" % i
        recs.append((rid, seq))
        if fastq:
            q = "".join(rng.choice("@+!IJ>") for _ in range(L))  # quality lines that begin with '@' or '+'
            out.append("@%s desc %d%s%s%s+%s%s%s" % (rid, i, nl, seq, nl, nl, q, nl))
        elif multiline:
            w = rng.choice([60, 80, 7])
            lines = [seq[j:j + w] for j in range(0, len(seq), w)] or [""]
            out.append(">%s desc%s%s%s" % (rid, nl, nl.join(lines), nl))
        else:
            out.append(">%s%s%s%s" % (rid, nl, seq, nl))
    return recs, "".join(out)


def _dump(tool, path, fastq, chunk, threads, env=None):
    import os
    r = subprocess.run([tool, str(path), "fastq" if fastq else "fasta", str(chunk), str(threads)], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    got = [ln.split("\t") for ln in r.stdout.split("\n") if ln != ""]
    return [g if len(g) == 3 else g + [""] for g in got]


def _lz4_frame(data):
    try:
        L = C.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("liblz4.so.1 not present")
    L.LZ4F_compressFrameBound.restype = C.c_size_t
    L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
    L.LZ4F_compressFrame.restype = C.c_size_t
    L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    cap = L.LZ4F_compressFrameBound(len(data), None)
    dst = C.create_string_buffer(cap)
    n = L.LZ4F_compressFrame(dst, cap, data, len(data), None)
    assert n > 0 and n <= cap
    return dst.raw[:n]


@pytest.fixture(scope="module")
def tool():
    return B.build_feeder_dump()


@pytest.mark.parametrize("fastq,multiline", [(False, False), (False, True), (True, False)])
@pytest.mark.parametrize("crlf", [False, True])
def test_feeder_matches_plain_parser(tool, tmp_path, fastq, multiline, crlf):
    rng = random.Random(11 + 2 * fastq + multiline + 4 * crlf)
    recs, text = _make(200, fastq, multiline, crlf, rng)
    want = [[a, str(len(b)), b] for a, b in recs]
    for trailing in (True, False):
        t = text if trailing else text.rstrip("\r\n")
        raw, gz = tmp_path / "t.fx", tmp_path / "t.fx.gz"
        raw.write_text(t, newline="")
        with gzip.open(gz, "wt", newline="") as f:
            f.write(t)
        for path in (raw, gz):
            for chunk, th in ((64, 4), (1000, 1), (5000, 4), (1 << 28, 2)):
                assert _dump(tool, path, fastq, chunk, th) == want, (str(path), chunk, th, trailing)
        # the other reader of each form: gzip streamed through zlib (default: members inflated whole by libdeflate), FASTQ read
        # with pread as it is (default: header and sequence lines copied out of the mapped file)
        for chunk, th in ((64, 3), (5000, 4)):
            assert _dump(tool, gz, fastq, chunk, th, {"MQ_FEEDER_NO_LIBDEFLATE": "1"}) == want
            assert _dump(tool, gz, fastq, chunk, th, {"MQ_GZ_WHOLE_LIMIT": "0"}) == want
            if fastq:
                assert _dump(tool, raw, fastq, chunk, th, {"MQ_FEEDER_NO_LEAN_FASTQ": "1"}) == want


def test_feeder_reads_lz4_frames(tool, tmp_path):
    recs, text = _make(150, True, False, False, random.Random(5))
    p = tmp_path / "reads.fq.lz4"
    p.write_bytes(_lz4_frame(text.encode()))
    assert _dump(tool, p, True, 4096, 3) == [[a, str(len(b)), b] for a, b in recs]


def test_feeder_long_record_spanning_many_chunks(tool, tmp_path):
    rng = random.Random(9)
    seqs = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (10, 3_000_000, 20)]
    p = tmp_path / "long.fa"
    p.write_text("".join(">c%d\n%s\n" % (i, "\n".join(s[j:j + 80] for j in range(0, len(s), 80))) for i, s in enumerate(seqs)))
    got = _dump(tool, p, False, 100_000, 4)
    assert [(g[0], int(g[1])) for g in got] == [("c%d" % i, len(s)) for i, s in enumerate(seqs)]
    assert got[1][2] == seqs[1]


def _bgzf(data, block=3000):
    """BGZF (bgzip) writer: independent deflate blocks, block size in the gzip extra field 'BC', then the empty EOF block."""
    import struct
    import zlib
    out = bytearray()
    for i in list(range(0, len(data), block)) + [None]:
        piece = b"" if i is None else data[i:i + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = c.compress(piece) + c.flush()
        bsize = 18 + len(d) + 8
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + d
        out += struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out)


@pytest.mark.parametrize("fastq", [False, True])
def test_feeder_inflates_bgzf_blocks_in_parallel(tool, tmp_path, fastq):
    """A bgzip'ed file is indexed by its block headers and read like a raw file: every reader thread inflates the blocks of
    its own chunk (plain gzip streams stay on the single inflate thread).  Blocks of 3,000 bytes so that records, chunk
    boundaries and block boundaries fall everywhere relative to each other."""
    recs, text = _make(250, fastq, not fastq, False, random.Random(21 + fastq))
    want = [[a, str(len(b)), b] for a, b in recs]
    p = tmp_path / ("r.fq.gz" if fastq else "r.fa.gz")
    p.write_bytes(_bgzf(text.encode()))
    import gzip as _g
    assert _g.decompress(p.read_bytes()).decode() == text  # a valid multi-member gzip file, too
    for chunk, th in ((64, 4), (2500, 3), (40000, 4), (1 << 28, 2)):
        assert _dump(tool, p, fastq, chunk, th) == want, (chunk, th)
    import os
    r = subprocess.run([tool, str(p), "fastq" if fastq else "fasta", "5000", "2"], capture_output=True, text=True, env=dict(os.environ, FEEDER_DUMP_KIND="1"))
    assert "kind=bgzf" in r.stderr
    plain = tmp_path / "plain.fx.gz"
    plain.write_bytes(_g.compress(text.encode()))
    r = subprocess.run([tool, str(plain), "fastq" if fastq else "fasta", "5000", "2"], capture_output=True, text=True, env=dict(os.environ, FEEDER_DUMP_KIND="1"))
    assert "kind=gzip" in r.stderr and r.returncode == 0


def test_feeder_rejects_truncated_compressed_input(tool, tmp_path):
    """A .gz / .lz4 file cut in the middle of its stream is an error (flate2's GzDecoder returns UnexpectedEof and the reference
    panics, src/main.rs:60-75), never a shorter list of reads."""
    recs, text = _make(400, False, False, False, random.Random(31))
    gz = gzip.compress(text.encode())
    for name, blob in (("cut.fa.gz", gz[:len(gz) // 2]), ("cut2.fa.gz", gz[:-4]), ("cut.fa.lz4", None)):
        if blob is None:
            fr = _lz4_frame(text.encode())
            blob = fr[:len(fr) // 2]
        p = tmp_path / name
        p.write_bytes(blob)
        import os
        for env in ({}, {"MQ_FEEDER_NO_LIBDEFLATE": "1"}):
            r = subprocess.run([tool, str(p), "fasta", "4096", "2"], capture_output=True, text=True, timeout=60, env=dict(os.environ, **env))
            assert r.returncode != 0 and "truncated" in r.stderr, (name, env, r.returncode, r.stderr[-200:])
    # two whole members back to back are fine (concatenated gzip files), and so is an empty file
    two = tmp_path / "two.fa.gz"
    two.write_bytes(gz + gz)
    assert len(_dump(tool, two, False, 4096, 2)) == 2 * len(recs)
    empty = tmp_path / "empty.fa.gz"
    empty.write_bytes(b"")
    assert _dump(tool, empty, False, 4096, 2) == []


def test_feeder_id_is_cut_at_the_first_space_only(tool, tmp_path):
    """seq_io's id() (src/closures.rs:107,135) splits the header at the first SPACE: a TAB stays in the id; the CR of a CR-LF
    header line does not."""
    p = tmp_path / "ids.fa"
    p.write_bytes(b">id1\tfoo bar\nACGT\n>id2\r\nAC\r\n>id3 x\ty\r\nA\r\n")
    got = _dump_raw(tool, p)
    assert got == [(b"id1\tfoo", b"4", b"ACGT"), (b"id2", b"2", b"AC"), (b"id3", b"1", b"A")]
    q = tmp_path / "ids.fq"
    q.write_bytes(b"@r1\tz w\nACG\n+\nIII\n@r2\r\nAC\r\n+\r\nII\r\n")
    assert _dump_raw(tool, q, fastq=True) == [(b"r1\tz", b"3", b"ACG"), (b"r2", b"2", b"AC")]


def _dump_raw(tool, path, fastq=False):
    """feeder_dump's lines split at the LAST two TABs (ids may hold TABs)."""
    r = subprocess.run([tool, str(path), "fastq" if fastq else "fasta", "4096", "2"], capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return [tuple(ln.rsplit(b"\t", 2)) for ln in r.stdout.split(b"\n") if ln != b""]


@pytest.mark.parametrize("multiline,crlf", [(False, False), (True, False), (True, True)])
def test_reference_loader_matches_plain_parser(tool, tmp_path, multiline, crlf):
    """ref_loader.hpp (whole file read once by all threads, records in file order) against the plain parser: single- and
    multi-line records, CR-LF, a record far longer than a read block boundary would allow for, 1 and 5 threads."""
    rng = random.Random(41 + multiline + 2 * crlf)
    recs, text = _make(60, False, multiline, crlf, rng)
    big = "".join(rng.choice("ACGTN") for _ in range(300_000))
    nl = "\r\n" if crlf else "\n"
    text += ">big one%s%s%s" % (nl, nl.join(big[j:j + 70] for j in range(0, len(big), 70)) if multiline else big, nl)
    recs.append(("big", big))
    for trailing in (True, False):
        t = text if trailing else text.rstrip("\r\n")
        p = tmp_path / "ref.fa"
        p.write_text(t, newline="")
        for th in (1, 5):
            r = subprocess.run([tool, str(p), "ref", "0", str(th)], capture_output=True, text=True, timeout=60)
            assert r.returncode == 0, r.stderr
            got = [ln.split("\t") for ln in r.stdout.split("\n") if ln != ""]
            got = [g if len(g) == 3 else g + [""] for g in got]
            assert got == [[a, str(len(b)), b] for a, b in recs], (th, trailing)
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n>r\nAC\n")
    assert subprocess.run([tool, str(bad), "ref", "0", "2"], capture_output=True).returncode != 0


def test_feeder_gzip_members_cut_anywhere(tool, tmp_path):
    """Concatenated gzip members whose boundaries fall in the middle of records and lines (what `cat a.gz b.gz` of arbitrary
    pieces gives): the whole-member reader carries the unfinished record into the next member; identical to the plain parser, to
    the zlib reader, for FASTA and FASTQ."""
    for fastq in (False, True):
        recs, text = _make(300, fastq, not fastq, False, random.Random(51 + fastq))
        data = text.encode()
        rng = random.Random(7)
        cuts = sorted(rng.sample(range(1, len(data)), 9))
        blob = b"".join(gzip.compress(data[a:b]) for a, b in zip([0] + cuts, cuts + [len(data)]))
        p = tmp_path / ("m.fq.gz" if fastq else "m.fa.gz")
        p.write_bytes(blob)
        want = [[a, str(len(b)), b] for a, b in recs]
        for chunk, th in ((64, 4), (3000, 2), (1 << 26, 3)):
            assert _dump(tool, p, fastq, chunk, th) == want, (fastq, chunk)
            assert _dump(tool, p, fastq, chunk, th, {"MQ_FEEDER_NO_LIBDEFLATE": "1"}) == want


@pytest.mark.parametrize("crlf", [False, True])
def test_feeder_last_quality_line_starting_with_at(tool, tmp_path, crlf):
    """A chunk boundary inside the LAST record of a FASTQ file whose quality line begins with '@': at the end of the input an '@'
    line followed by fewer than three further lines starts no record (it used to come out as a phantom read of length 0 with the
    quality string as its id).  Raw (pread), lean (mapped), plain gzip (both readers) and BGZF."""
    rng = random.Random(77 + crlf)
    nl = "\r\n" if crlf else "\n"
    recs, out = [], []
    for i in range(40):
        L = rng.choice([30, 70, 400])
        seq = "".join(rng.choice("ACGT") for _ in range(L))
        q = "@" + "".join(rng.choice("@+IJ") for _ in range(L - 1))  # every quality line begins with '@'
        recs.append(("q%d" % i, seq))
        out.append("@q%d%s%s%s+%s%s%s" % (i, nl, seq, nl, nl, q, nl))
    want = [[a, str(len(b)), b] for a, b in recs]
    for trailing in (True, False):
        text = "".join(out) if trailing else "".join(out).rstrip("\r\n")
        raw, gz, bg = tmp_path / "t.fq", tmp_path / "t.fq.gz", tmp_path / "b.fq.gz"
        raw.write_text(text, newline="")
        with gzip.open(gz, "wt", newline="") as f:
            f.write(text)
        bg.write_bytes(_bgzf(text.encode(), block=500))
        for chunk in (64, 100, 333, 1000, 70000):
            for th in (1, 3):
                assert _dump(tool, raw, True, chunk, th) == want, ("lean", chunk, th, trailing)
                assert _dump(tool, raw, True, chunk, th, {"MQ_FEEDER_NO_LEAN_FASTQ": "1"}) == want, ("raw", chunk, th, trailing)
                assert _dump(tool, gz, True, chunk, th) == want, ("gz", chunk, th, trailing)
                assert _dump(tool, gz, True, chunk, th, {"MQ_GZ_WHOLE_LIMIT": "0"}) == want, ("gz-stream", chunk, th, trailing)
                assert _dump(tool, gz, True, chunk, th, {"MQ_FEEDER_NO_LIBDEFLATE": "1"}) == want, ("gz-zlib", chunk, th, trailing)
                assert _dump(tool, bg, True, chunk, th) == want, ("bgzf", chunk, th, trailing)


@pytest.mark.parametrize("crlf", [False, True])
def test_feeder_unparsed_fasta_chunks(tool, tmp_path, crlf):
    """The driver's FASTA path with the device's share done on the host (feeder_dump FEEDER_DUMP_UNPARSED=1): chunks are handed over
    unparsed -- read with pread, or as views of the mapped file -- their line ends found by a plain scan and turned into spans
    (spans_from_line_ends); chunks that are not two lines per record go through parse_chunk after all.  Same records as the parser."""
    import os
    rng = random.Random(31 + crlf)
    for multiline in (False, True):
        recs, text = _make(300, False, multiline, crlf, rng)
        want = [[a, str(len(b)), b] for a, b in recs]
        for trailing in (True, False):
            t = text if trailing else text.rstrip("\r\n")
            p = tmp_path / "u.fa"
            p.write_text(t, newline="")
            for chunk, th in ((64, 3), (777, 1), (5000, 4), (1 << 20, 2)):
                for extra in ({}, {"MQ_FEEDER_MAPPED_FASTA": "1"}, {"MQ_FEEDER_MAPPED_FASTA": "1", "MQ_FEEDER_PAGE_LOCK": "1"}):
                    env = dict(os.environ, FEEDER_DUMP_UNPARSED="1", **extra)
                    r = subprocess.run([tool, str(p), "fasta", str(chunk), str(th)], capture_output=True, text=True, timeout=60, env=env)
                    assert r.returncode == 0, r.stderr
                    got = [ln.split("\t") for ln in r.stdout.split("\n") if ln != ""]
                    got = [g if len(g) == 3 else g + [""] for g in got]
                    assert got == want, (multiline, trailing, chunk, th, extra)
                    assert ("mapped 1" in r.stderr) == bool(extra)
                    if multiline:
                        assert "irregular 0" not in r.stderr  # sequences over several lines: handed back to the parser


@pytest.mark.parametrize("crlf", [False, True])
def test_feeder_fastq_quality_lines_of_another_length(tool, tmp_path, crlf):
    """FASTQ records whose quality line is shorter or longer than the sequence line (not what a sequencer writes, but what parse_chunk accepts: the
    quality line ends at the next line end).  The lean reader looks at the byte where an equally long quality line would end together with the
    NEXT record's read; when that byte is no line end it searches the real one.  Lean = chunked reader = plain parser, at every chunk size."""
    rng = random.Random(77 + crlf)
    nl = "\r\n" if crlf else "\n"
    recs, out = [], []
    for i in range(300):
        L = rng.choice([1, 5, 50, 200, 1000, 3000])
        seq = "".join(rng.choice("ACGT") for _ in range(L))
        ql = L if i % 3 == 0 else max(0, L + rng.choice([-40, -3, -1, 1, 2, 64]))
        q = "".join(rng.choice("IJ>!") for _ in range(ql))
        recs.append(("q%d" % i, seq))
        out.append("@q%d%s%s%s+%s%s%s" % (i, nl, seq, nl, nl, q, nl))
    text = "".join(out)
    want = [[a, str(len(b)), b] for a, b in recs]
    raw = tmp_path / "odd.fq"
    raw.write_text(text, newline="")
    for chunk, th in ((64, 4), (700, 3), (5000, 1), (1 << 28, 2)):
        assert _dump(tool, raw, True, chunk, th) == want, ("lean", chunk, th)
        assert _dump(tool, raw, True, chunk, th, {"MQ_FEEDER_NO_LEAN_FASTQ": "1"}) == want, ("chunked", chunk, th)
