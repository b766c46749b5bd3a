"""Real inputs for bench.py (--reference-fasta / --reads-fastx): a reference FASTA and a reads FASTA / FASTQ (plain or .gz) as the arrays
the bench works on.  Bench infrastructure, not the product path (the native driver has its own feeder) and not the oracle.

What the reference does with the same files (src/closures.rs:60-66,100-110): every record's id is its header up to the first space,
its sequence is upper-cased (to_ascii_uppercase), multi-line FASTA records are joined; FASTQ records are four lines.
BASELINE configs 3 and 4 are `chm13v2.0.fa` with pbsim reads (experiments/simulate_chm13.sh:12) or the DeepConsensus HG002 FASTQ
(experiments/table1.sh:50, README.md:68-77); pbsim2fq read names carry the truth (`S1_7!chr3!1042!25012!+`), real reads do not.
"""
import gzip
import io
import os
import re

import numpy as np


def _open(path):
    if str(path).endswith(".gz"):
        return gzip.open(path, "rb")
    return open(path, "rb")


def _upper(a):
    """to_ascii_uppercase on a uint8 array, in place."""
    m = (a >= 97) & (a <= 122)
    a[m] -= 32
    return a


def load_reference(path):
    """(genome uint8, contig offsets uint64[n + 1], contig names) of a FASTA file: sequences joined and upper-cased, in file order."""
    with _open(path) as f:
        raw = f.read()
    buf = np.frombuffer(raw, dtype=np.uint8)
    if buf.size == 0:
        raise ValueError("empty reference file: %s" % path)
    nl = np.flatnonzero(buf == 10)
    line_start = np.concatenate([[0], nl + 1]).astype(np.int64)
    line_start = line_start[line_start < buf.size]
    hdr = line_start[buf[line_start] == ord(">")]
    if hdr.size == 0 or (line_start[0] != hdr[0] and np.any((buf[:hdr[0]] != 10) & (buf[:hdr[0]] != 13))):
        raise ValueError("malformed FASTA record in %s" % path)
    # end of every header line
    hdr_end = np.empty(hdr.size, dtype=np.int64)
    idx = np.searchsorted(nl, hdr)
    hdr_end[:] = np.where(idx < nl.size, nl[np.minimum(idx, max(nl.size - 1, 0))] if nl.size else buf.size, buf.size)
    keep = (buf != 10) & (buf != 13)
    names = []
    for a, b in zip(hdr.tolist(), hdr_end.tolist()):
        keep[a:min(b + 1, buf.size)] = False
        h = raw[a + 1:b].rstrip(b"\r")
        names.append(h.split(b" ", 1)[0].decode())
    genome = _upper(buf[keep].copy())
    # bases per record = kept bytes between consecutive headers
    csum = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
    bounds = np.concatenate([hdr, [buf.size]])
    offs = np.zeros(hdr.size + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(csum[bounds[1:]] - csum[bounds[:-1]]).astype(np.uint64)
    assert int(offs[-1]) == genome.size
    return genome, offs, names


def _records(fh, fasta):
    """(id, sequence bytes) of a FASTA (possibly multi-line) or 4-line FASTQ stream."""
    if fasta:
        name, parts = None, []
        for line in fh:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    yield name, b"".join(parts)
                name, parts = line[1:].split(b" ", 1)[0].decode(), []
            elif name is not None:
                parts.append(line)
        if name is not None:
            yield name, b"".join(parts)
    else:
        while True:
            h = fh.readline()
            if not h:
                return
            s = fh.readline().rstrip(b"\r\n")
            fh.readline()
            fh.readline()
            if not h.startswith(b"@"):
                raise ValueError("malformed FASTQ record: %r" % h[:40])
            yield h.rstrip(b"\r\n")[1:].split(b" ", 1)[0].decode(), s


def is_fasta_name(name):
    """src/main.rs:196,202: anything else is read as FASTQ."""
    name = str(name)
    return (".fasta." in name or name.endswith(".fna") or ".fna." in name or ".fa." in name or name.endswith(".fa") or name.endswith(".fasta"))


def load_reads(path, n_reads, skip=0):
    """Reads skip .. skip + n_reads - 1 of a FASTA / FASTQ file (fewer when the file ends): dict(bases, offsets, names), upper-cased."""
    names, seqs = [], []
    with _open(path) as fh:
        fh = io.BufferedReader(fh, 1 << 24) if not isinstance(fh, io.BufferedReader) else fh
        for i, (nm, s) in enumerate(_records(fh, is_fasta_name(path))):
            if i < skip:
                continue
            if len(names) >= n_reads:
                break
            names.append(nm)
            seqs.append(s)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        offs[1:] = np.cumsum([len(s) for s in seqs]).astype(np.uint64)
    bases = _upper(np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()) if seqs else np.zeros(0, dtype=np.uint8)
    return dict(bases=bases, offsets=offs, names=names)


_PBSIM = re.compile(r"^[^!]*!([^!]+)!(\d+)!(\d+)!([+-])$")


def truth_from_names(names, ctg_names):
    """pbsim2fq-style names (`S1_7!chr3!1042!25012!+`, example/nearperfect-ecoli.100.fa:1) -> dict(ctg, start, end, strand) as
    tools/sim.mapeval wants it, or None when any name does not carry a truth on a contig of this reference."""
    where = {n: i for i, n in enumerate(ctg_names)}
    n = len(names)
    ctg = np.zeros(n, dtype=np.uint32)
    start = np.zeros(n, dtype=np.uint64)
    end = np.zeros(n, dtype=np.uint64)
    strand = np.zeros(n, dtype=np.uint8)
    for i, nm in enumerate(names):
        m = _PBSIM.match(nm)
        if not m or m.group(1) not in where:
            return None
        ctg[i] = where[m.group(1)]
        start[i] = int(m.group(2))
        end[i] = int(m.group(3))
        strand[i] = 1 if m.group(4) == "-" else 0
    return dict(ctg=ctg, start=start, end=end, strand=strand)


def describe(path):
    return "%s (%d bytes)" % (os.path.basename(str(path)), os.path.getsize(path))
