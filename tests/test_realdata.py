"""bench.py's real-data door (tools/realdata.py): a reference FASTA and a reads FASTA / FASTQ as the arrays the bench works on -- ids up to
the first space, sequences joined and upper-cased (src/closures.rs:60-66,100-110), plain or .gz, pbsim2fq truth from the read names."""
import gzip

import numpy as np
import pytest

from tools import realdata


def _seq(rng, n):
    return bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))


def test_reference_single_line_multi_line_crlf_lowercase_gz(tmp_path):
    rng = np.random.default_rng(3)
    seqs = [_seq(rng, 1000), _seq(rng, 61), _seq(rng, 1), _seq(rng, 240)]
    names = ["chr1", "chr2", "tiny", "chrM"]
    single = b"".join(b">" + n.encode() + b" some description\n" + s + b"\n" for n, s in zip(names, seqs))
    wrapped = b"".join(b">" + n.encode() + b"\n" + b"".join(s[i:i + 60] + b"\n" for i in range(0, len(s), 60)) for n, s in zip(names, seqs))
    crlf = wrapped.replace(b"\n", b"\r\n")
    lower = single.lower().replace(b">CHR", b">chr")
    lower = b"".join(b">" + n.encode() + b"\n" + s.lower() + b"\n" for n, s in zip(names, seqs))
    nofinal = single[:-1]
    blank_first = b"\n\n" + single
    for i, txt in enumerate((single, wrapped, crlf, lower, nofinal, blank_first)):
        p = tmp_path / ("r%d.fa" % i)
        p.write_bytes(txt)
        g, off, nm = realdata.load_reference(p)
        assert nm == names, i
        assert off.tolist() == [0, 1000, 1061, 1062, 1302], i
        assert bytes(g) == b"".join(seqs), i
    pz = tmp_path / "r.fa.gz"
    with gzip.open(pz, "wb") as f:
        f.write(wrapped)
    g, off, nm = realdata.load_reference(pz)
    assert nm == names and bytes(g) == b"".join(seqs)
    bad = tmp_path / "bad.fa"
    bad.write_bytes(b"ACGT\n>x\nACGT\n")
    with pytest.raises(ValueError):
        realdata.load_reference(bad)
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")
    with pytest.raises(ValueError):
        realdata.load_reference(empty)


def test_reads_fasta_fastq_slices_and_truth(tmp_path):
    rng = np.random.default_rng(4)
    seqs = [_seq(rng, int(n)) for n in rng.integers(1, 400, size=23)]
    names = ["S1_%d!chr%d!%d!%d!%s" % (i + 1, 1 + i % 2, 10 * i, 10 * i + len(s), "+-"[i % 2]) for i, s in enumerate(seqs)]
    fa = tmp_path / "reads.fa"
    fa.write_bytes(b"".join(b">" + n.encode() + b" x=1\n" + s.lower() + b"\n" for n, s in zip(names, seqs)))
    fq = tmp_path / "reads.fastq"
    fq.write_bytes(b"".join(b"@" + n.encode() + b"\n" + s + b"\n+\n" + b"@" * len(s) + b"\n" for n, s in zip(names, seqs)))  # '@' as a quality too
    fqz = tmp_path / "reads.fq.gz"
    with gzip.open(fqz, "wb") as f:
        f.write(fq.read_bytes())
    for p in (fa, fq, fqz):
        assert realdata.is_fasta_name(p) == (p == fa)
        r = realdata.load_reads(p, 100)
        assert r["names"] == names and bytes(r["bases"]) == b"".join(seqs)
        assert r["offsets"].tolist() == np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).tolist()
        r = realdata.load_reads(p, 5, skip=7)  # rank 1's slice at 7 reads per rank... any slice
        assert r["names"] == names[7:12] and bytes(r["bases"]) == b"".join(seqs[7:12])
        r = realdata.load_reads(p, 5, skip=21)
        assert r["names"] == names[21:] and r["offsets"].size == 3
        assert realdata.load_reads(p, 5, skip=99)["offsets"].tolist() == [0]
    t = realdata.truth_from_names(names, ["chr1", "chr2"])
    assert t["ctg"].tolist() == [i % 2 for i in range(23)] and t["strand"].tolist() == [i % 2 for i in range(23)]
    assert t["start"].tolist() == [10 * i for i in range(23)] and t["end"][3] == 30 + len(seqs[3])
    assert realdata.truth_from_names(names, ["chr1"]) is None                 # a contig this reference does not have
    assert realdata.truth_from_names(["m64011_190830_220126/1/ccs"], ["chr1"]) is None  # a real read's name


def test_bench_rejects_reads_without_a_reference():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads-fastx", "/nonexistent.fq"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs --reference-fasta" in r.stderr + r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reference-fasta", "/nonexistent.fa"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no such file" in r.stderr + r.stdout
