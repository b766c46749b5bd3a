"""Writes tests/golden/kat_intree.json: known-answer vectors for the stages whose source IS in the
reference tree (src/match.rs, src/chain.rs, src/mers.rs).  The reference ships no tests and cannot be
built here (Rust + unpinned git crates, no toolchain), so these were derived by hand from the cited
lines -- they are inputs + expected outputs, no reference text.
Also the published ntHash-1 / SipHash known answers that pin the restated third-party arithmetic.
Run: python tests/golden/make_kat_intree.py
"""
import json, os

def Q(start, end, rev): return dict(start=start, end=end, rev=int(rev))
def E(id, start, end, offset, rc): return dict(id=id, start=start, end=end, offset=offset, rc=int(rc))
def M(qs, qe, rs, re, count, rc): return dict(q_start=qs, q_end=qe, r_start=rs, r_end=re, count=count, rc=int(rc))

kat = {}

# ntHash-1 64-bit known answers (nthash crate test-suite / ntHash paper constants)
kat["nthash"] = [
    dict(seq="TGCAG", l=5, fwd="0x0bafa6728fc6dabf", rev="0x8cf2d4072cca480e", canon="0x0bafa6728fc6dabf"),
    dict(seq="ACGTC", l=5, canon="0x480202d54e8ebecd"),
]
kat["nthash_seeds"] = dict(A="0x3c8bfbb395c60474", C="0x3193c18562a02b4c", G="0x20323ed082572324", T="0x295549f54be24456", N="0x0")
# SipHash-2-4 reference vectors (Aumasson & Bernstein, key 00..0f): pins the round function used as 1-3
kat["siphash24"] = [dict(msg_len=0, out="0x726fdb47dd0e0e31"), dict(msg_len=15, out="0xa129ca6149be45e5")]

# C1: forward run broken by a miss (src/match.rs:45-58, src/mers.rs:62-69)
q = [Q(100 + 100 * i, 550 + 100 * i, False) for i in range(8)]
r = [E(0, 5000 + 100 * i, 5450 + 100 * i, 70 + i, False) if i != 3 else None for i in range(8)]
kat["C1"] = dict(q=q, r=r, matches=[dict(ref=0, **M(100, 750, 5000, 5650, 3, False)), dict(ref=0, **M(500, 1250, 5400, 6150, 4, False))],
                 coords=dict(rc=0, q_start=100, q_end=1249, r_start=5000, r_end=6149, score=7, mapq=0),
                 q_len=2000, r_len=100000, q_id="r1", r_name="chrA",
                 paf="r1\t2000\t0\t1999\t+\tchrA\t100000\t4900\t6899\t7\t100000\t0")
# C2: reverse run
q = [Q(100 + 100 * i, 550 + 100 * i, False) for i in range(6)]
r = [E(2, 9000 - 100 * i, 9450 - 100 * i, 50 - i, True) for i in range(6)]
kat["C2"] = dict(q=q, r=r, matches=[dict(ref=2, **M(100, 1050, 8500, 9450, 6, True))],
                 coords=dict(rc=1, q_start=100, q_end=1049, r_start=8500, r_end=9449, score=6, mapq=0),
                 q_len=1500, r_len=20000, q_id="r2", r_name="chrB",
                 paf="r2\t1500\t0\t1499\t-\tchrB\t20000\t8050\t9549\t6\t20000\t0")
# C3: precedence quirk of Match::check (src/match.rs:39-43)
kat["C3a"] = dict(q=[Q(0, 400, False), Q(100, 500, True), Q(200, 600, False)],
                  r=[E(0, 1000, 1400, 10, False), E(7, 777, 1177, 11, False), E(0, 1200, 1600, 12, False)],
                  matches=[dict(ref=0, **M(0, 600, 1000, 1600, 3, False))])
kat["C3b"] = dict(q=[Q(0, 400, False), Q(100, 500, False)],
                  r=[E(0, 1000, 1400, 10, True), E(7, 900, 1300, 9, True)],
                  matches=[dict(ref=0, **M(0, 400, 1000, 1400, 1, True)), dict(ref=7, **M(100, 500, 900, 1300, 1, True))])
# C4: anchor + co-linear filter (src/chain.rs:43-75,93-104,123-169)
six = [M(100, 600, 50100, 50600, 2, False), M(1000, 3000, 51000, 53000, 9, False), M(3500, 4000, 58000, 58500, 3, False),
       M(4200, 4700, 54200, 54700, 1, False), M(5000, 5400, 40000, 40400, 4, False), M(6000, 6500, 56000, 56500, 2, True)]
kat["C4"] = [
    dict(matches=six, c=4, s=11, g=2000, coords=dict(rc=0, q_start=100, q_end=4699, r_start=50100, r_end=54699, score=12, mapq=60)),
    dict(matches=six, c=4, s=11, g=5000, coords=dict(rc=0, q_start=100, q_end=4699, r_start=50100, r_end=54699, score=15, mapq=60)),
    dict(matches=six[:1], c=4, s=11, g=2000, coords=dict(rc=0, q_start=100, q_end=599, r_start=50100, r_end=50599, score=2, mapq=0)),
    dict(matches=six, c=0, s=11, g=2000, coords=dict(rc=0, q_start=100, q_end=4699, r_start=50100, r_end=54699, score=12, mapq=0)),
]
# C5: best reference (src/mers.rs:104-129)
kat["C5"] = [dict(scores=[12, 12, 3], best=-1), dict(scores=[3, 12, 11], best=1), dict(scores=[5], best=0), dict(scores=[], best=-1)]
# C6: find_coords clipping (src/mers.rs:131-183): (rc,q_start,q_end,r_start,r_end,score,mapq) -> fields 3,4,5,8,9
kat["C6"] = [
    dict(q_len=24299, r_len=248387328, coords=dict(rc=0, q_start=100, q_end=24000, r_start=224752893, r_end=224776729, score=132, mapq=60),
         out=dict(q_start=0, q_end=24298, rc=0, r_start=224752793, r_end=224777027)),
    dict(q_len=10000, r_len=50000, coords=dict(rc=0, q_start=500, q_end=9000, r_start=200, r_end=8700, score=20, mapq=60),
         out=dict(q_start=300, q_end=9999, rc=0, r_start=0, r_end=9699)),
    dict(q_len=10000, r_len=9500, coords=dict(rc=0, q_start=500, q_end=9000, r_start=600, r_end=9100, score=20, mapq=60),
         out=dict(q_start=0, q_end=9399, rc=0, r_start=100, r_end=9499)),
    dict(q_len=10000, r_len=50000, coords=dict(rc=1, q_start=500, q_end=9000, r_start=20000, r_end=28500, score=20, mapq=60),
         out=dict(q_start=0, q_end=9999, rc=1, r_start=19001, r_end=29000)),
    dict(q_len=10000, r_len=29000, coords=dict(rc=1, q_start=700, q_end=9000, r_start=300, r_end=28500, score=20, mapq=0),
         out=dict(q_start=201, q_end=9300, rc=1, r_start=0, r_end=28999)),
]
# PAF line shape quoted in the reference tree (experiments/intersect_pafs.py:14)
kat["paf_shape"] = dict(q_id="S1_1!chr1!224752794!224777027!+", r_name="chr1", q_len=24299, r_len=248387328,
                        coords=dict(rc=0, q_start=100, q_end=24000, r_start=224752893, r_end=224776729, score=132, mapq=60),
                        paf="S1_1!chr1!224752794!224777027!+\t24299\t0\t24298\t+\tchr1\t248387328\t224752793\t224777027\t132\t248387328\t60")
# Index semantics (src/index.rs:94-104,118-126)
kat["index"] = dict(ops=[[5, 1, 10, 40, 0, 0], [6, 1, 20, 50, 1, 1], [5, 2, 99, 140, 7, 0], [7, 0, 0, 0, 3, 0], [8, 3, 5, 36, 2, 1],
                         [6, 1, 20, 50, 1, 1], [6, 1, 20, 50, 1, 1]],
                    get={"5": None, "6": None, "7": None, "8": dict(id=3, start=5, end=36, offset=2, rc=1), "9": None},
                    count=1, keys=4)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_intree.json")
json.dump(kat, open(out, "w"), indent=1)
print("wrote", out)
