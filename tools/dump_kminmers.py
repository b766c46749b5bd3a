#!/usr/bin/env python3
"""dump_kminmers.py -- the k-min-mers of every record of a FASTA file as the CPU oracle yields them, one per line:
    KMM <TAB> start <TAB> end <TAB> offset <TAB> rev <TAB> hash
the same line the patched reference prints under MQ_DUMP (tools/check_against_upstream.sh).  --variant selects one of the
oracle's diagnostic readings of the unpinned seeding decisions (0 = the frozen one; bits 1 `<` on the bound, 2 f32 bound,
4 32-bit ntHash, 8 position = run end, 16 end from the compressed window, 32 rev on `<=`: oracle/mapquik_oracle.c).
Test/diagnostic infrastructure: imports oracle/."""
import argparse
import gzip
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def records(path):
    op = gzip.open if path.endswith(".gz") else open
    name, seq = None, []
    with op(path, "rb") as f:
        for ln in f:
            ln = ln.rstrip(b"\r\n")
            if ln.startswith(b">"):
                if name is not None:
                    yield name, b"".join(seq).upper()
                name, seq = ln[1:].split(b" ", 1)[0].decode(), []
            elif name is not None:
                seq.append(ln)
    if name is not None:
        yield name, b"".join(seq).upper()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("fasta")
    ap.add_argument("-k", type=int, default=5)
    ap.add_argument("-l", type=int, default=31)
    ap.add_argument("-d", "--density", type=float, default=0.01)
    ap.add_argument("--nohpc", action="store_true")
    ap.add_argument("--variant", type=int, default=0)
    a = ap.parse_args(argv)
    from oracle import oracle as O
    O.lib().mqo_set_variant(a.variant)
    p = O.params(k=a.k, l=a.l, density=a.density, use_hpc=not a.nohpc)
    out = sys.stdout
    for name, seq in records(a.fasta):
        if len(seq) < a.l + a.k - 1:
            continue
        for km in O.kminmers(seq, p):
            out.write("KMM\t%d\t%d\t%d\t%s\t%d\n" % (km["start"], km["end"], km["offset"], "true" if km["rev"] else "false", km["hash"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
