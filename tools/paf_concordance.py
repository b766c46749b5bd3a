#!/usr/bin/env python3
"""paf_concordance.py -- concordance of two PAF files, the measure the reference's experiments use
(experiments/intersect_pafs.py:36-73, itself modelled on paftools mapeval): two mappings of the same read agree when they
are on the same target and  overlap / (highest - lowest coordinate) > 0.1.

    python tools/paf_concordance.py a.paf b.paf [--upstream-columns]

Written from the PAF column definition: target name = column 6, target start = column 8, target end = column 9.  The
reference's script reads columns 7 and 8 (target length, target start) as "start, end" (intersect_pafs.py:18-19), which
makes every pair on one chromosome overlap; --upstream-columns reproduces that reading for comparison with its numbers.
Prints the same five summary lines as the reference's script and returns the counts from concordance()."""
import argparse
import sys


def parse_paf(path, upstream_columns=False):
    """read name -> (target, start, end); a read with several lines keeps the last one, as the reference's dict does."""
    p = {}
    with open(path) as f:
        for line in f:
            ls = line.split()
            if len(ls) < 9:
                continue
            a, b = (6, 7) if upstream_columns else (7, 8)
            p[ls[0]] = (ls[5], int(ls[a]), int(ls[b]))
    return p


def overlap_ratio(s1, e1, s2, e2):
    lo1, hi1, lo2, hi2 = min(s1, e1), max(s1, e1), min(s2, e2), max(s2, e2)
    span = max(hi1, hi2) - min(lo1, lo2)
    if hi1 < hi2:
        o = hi1 - lo2 if hi1 >= lo2 else 0
    else:
        o = hi2 - lo1 if hi2 >= lo1 else 0
    return o / span if span > 0 else 1.0


def concordance(paf1, paf2, threshold=0.1):
    """dict(concordant, discordant, different_target, only_in_1, only_in_2); discordant includes different_target,
    as in the reference's counters."""
    c = dict(concordant=0, discordant=0, different_target=0, only_in_1=0, only_in_2=0)
    for read, (t1, s1, e1) in paf1.items():
        if read not in paf2:
            c["only_in_1"] += 1
            continue
        t2, s2, e2 = paf2[read]
        if t1 != t2:
            c["different_target"] += 1
            c["discordant"] += 1
        elif overlap_ratio(s1, e1, s2, e2) > threshold:
            c["concordant"] += 1
        else:
            c["discordant"] += 1
    c["only_in_2"] = sum(1 for r in paf2 if r not in paf1)
    return c


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("paf1")
    ap.add_argument("paf2")
    ap.add_argument("--upstream-columns", action="store_true", help="read columns 7,8 as start,end like experiments/intersect_pafs.py")
    a = ap.parse_args(argv)
    p1, p2 = parse_paf(a.paf1, a.upstream_columns), parse_paf(a.paf2, a.upstream_columns)
    c = concordance(p1, p2)
    n1, n2 = max(len(p1), 1), max(len(p2), 1)
    print(f"Total number of mapped reads in {a.paf1}: {len(p1)}")
    print(f"Total number of mapped reads in {a.paf2}: {len(p2)}")
    print(f"Number of concordant mappings: {c['concordant']} ({c['concordant'] / n1 * 100}% of {a.paf1}, {c['concordant'] / n2 * 100}% of {a.paf2})")
    print(f"Number of discordant mappings on same      chromosome: {c['discordant']} ({c['discordant'] / n1 * 100}% of {a.paf1}, "
          f"{c['discordant'] / n2 * 100}% of {a.paf2})")
    print(f"Number of discordant mappings on different chromosome: {c['different_target']}")
    print(f"Mapped only in {a.paf1}: {c['only_in_1']}; only in {a.paf2}: {c['only_in_2']}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
