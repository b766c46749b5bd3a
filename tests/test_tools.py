"""Evaluation tools around the path: the PAF concordance measure of the reference's experiments (experiments/intersect_pafs.py)."""
import subprocess
import sys
import os

from tools import paf_concordance as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(read, tgt, tlen, ts, te, qlen=1000):
    return "\t".join(map(str, [read, qlen, 0, qlen - 1, "+", tgt, tlen, ts, te, 50, tlen, 60])) + "\n"


def test_concordance_counts_and_cli(tmp_path):
    a, b = tmp_path / "a.paf", tmp_path / "b.paf"
    a.write_text(_line("r1", "chr1", 10_000_000, 1000, 2000) + _line("r2", "chr1", 10_000_000, 5000, 6000) +
                 _line("r3", "chr2", 5_000_000, 100, 1100) + _line("r4", "chr1", 10_000_000, 7000, 8000))
    b.write_text(_line("r1", "chr1", 10_000_000, 1100, 2100) +     # overlaps: 900 / 1100 > 0.1
                 _line("r2", "chr1", 10_000_000, 900_000, 901_000) +  # same target, far away
                 _line("r3", "chr1", 10_000_000, 100, 1100) +        # other target
                 _line("r5", "chr1", 10_000_000, 1, 2))
    c = pc.concordance(pc.parse_paf(a), pc.parse_paf(b))
    assert c == dict(concordant=1, discordant=2, different_target=1, only_in_1=1, only_in_2=1)
    # the reference's column reading (target length as "start") calls r2 concordant too
    cu = pc.concordance(pc.parse_paf(a, True), pc.parse_paf(b, True))
    assert cu["concordant"] == 2 and cu["different_target"] == 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "paf_concordance.py"), str(a), str(b)], capture_output=True, text=True)
    assert r.returncode == 0 and "Number of concordant mappings: 1 (25.0% of" in r.stdout
    assert "different chromosome: 1" in r.stdout


def test_overlap_ratio_edges():
    assert pc.overlap_ratio(0, 100, 100, 200) == 0.0
    assert pc.overlap_ratio(0, 100, 50, 150) == 50 / 150
    assert pc.overlap_ratio(200, 100, 150, 250) == 50 / 150  # reversed coordinates
    assert pc.overlap_ratio(5, 5, 5, 5) == 1.0


def _fasta(tmp_path, seed=5):
    import random
    rng = random.Random(seed)

    def seq(n):
        out = []
        while len(out) < n:
            out.extend(rng.choice("ACGT") * rng.choice([1, 1, 2, 3, 5]))
        return "".join(out[:n])
    rep = seq(4000)
    recs = [("chrA", seq(60000) + rep + seq(20000) + rep), ("chrB some text", seq(30000) + "N" * 500 + seq(30000) + rep), ("tiny", "ACGT")]
    p = tmp_path / "ref.fa"
    with open(p, "w") as f:
        for n, s in recs:
            f.write(">" + n + "\n")
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")
    return str(p)


def _dump(fa, variant, extra=()):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dump_kminmers.py"), fa, "--variant", str(variant)] + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    return r.stdout


def test_upstream_compare_is_not_blinded_by_another_tuple_hash(tmp_path):
    """tools/upstream_compare.py (what tools/check_against_upstream.sh decides with): an "upstream" dump made by the oracle at variant 6 (32-bit
    ntHash, f32 bound -- the crate's likeliest other reading) with ANOTHER tuple hash (the oracle's bit 64) must come back as variant 6:
    positions identical, hash partition identical, hash values differ.  The same with the reference's real hash: all three identical.  A dump no
    variant reproduces (one position moved) comes back with variant null and exit code 1; equal tuples merged by a colliding "hash" show as a
    partition that differs."""
    import json
    fa = _fasta(tmp_path)
    cmp_py = os.path.join(ROOT, "tools", "upstream_compare.py")

    def run(dump_text, extra=()):
        d = tmp_path / "up.kmm"
        d.write_text(dump_text)
        r = subprocess.run([sys.executable, cmp_py, str(d), fa, "--json-only"] + list(extra), capture_output=True, text=True)
        return r.returncode, json.loads(r.stdout.strip().splitlines()[-1])

    rc, j = run(_dump(fa, 6 | 64))
    assert rc == 0 and j["variant"] == 6 and j["positions"] == "identical" and j["hash_partition"] == "identical" and j["hash_values"] == "differ"
    assert j["variants_tried"] == 64 and 6 in j["matching_variants"] and 0 not in j["matching_variants"] and j["tuples"] > 500
    rc, j = run(_dump(fa, 6))
    assert rc == 0 and j["variant"] == 6 and j["hash_values"] == "identical"
    rc, j = run(_dump(fa, 0))
    assert rc == 0 and j["variant"] == 0 and j["hash_values"] == "identical"  # several readings may match an input: the frozen one is preferred
    rc, j = run(_dump(fa, 24, ["-k", "7", "-l", "16"]), ["-k", "7", "-l", "16"])
    assert rc == 0 and j["variant"] == 24
    lines = _dump(fa, 0).splitlines()
    p = lines[40].split("\t")
    p[1] = str(int(p[1]) + 1)
    rc, j = run("\n".join(lines[:40] + ["\t".join(p)] + lines[41:]) + "\n")
    assert rc == 1 and j["variant"] is None and j["positions"] == "differ" and j["matching_variants"] == []
    collide = ["\t".join(ln.split("\t")[:5] + [str(int(ln.split("\t")[5]) >> 54)]) for ln in lines]  # a 10-bit "hash": distinct tuples share values
    rc, j = run("\n".join(collide) + "\n")
    assert rc == 1 and j["variant"] == 0 and j["positions"] == "identical" and j["hash_partition"] == "differs"


def test_check_against_upstream_script_is_well_formed():
    r = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "check_against_upstream.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = open(os.path.join(ROOT, "tools", "check_against_upstream.sh")).read()
    assert "upstream_compare.py" in txt and "--nosimd" in txt and "--seeding-variant" in txt


def test_read_slices_are_the_reads_of_make_reads():
    """tools/sim.py read_slices (what bench.py synthesises a batch with, slice by slice straight into device memory) yields the reads of
    make_reads byte for byte -- bases, offsets and truth columns -- whatever the slice size, and from any first read (a strong-scaling shard)."""
    import numpy as np
    from tools import sim
    g, off, names = sim.make_genome([300_000, 200_000], seed=5, repeat_frac=0.05)
    want = sim.make_reads(g, off, 1500, seed=9, threads=4, len_mean=3000, len_sd=900, len_min=50, len_max=6000)
    for slice_reads, first in ((64, 0), (700, 0), (4096, 0), (333, 401)):
        bs, offs, tr = [], [np.zeros(1, dtype=np.uint64)], {}
        for r0, r1, b, o, t in sim.read_slices(g, off, 1500 - first, seed=9, slice_reads=slice_reads, threads=3, len_mean=3000, len_sd=900, len_min=50, len_max=6000,
                                               first_read=first):
            assert r1 - r0 == o.size - 1 and b.size == int(o[-1])
            bs.append(b.copy())
            offs.append(o[1:] + offs[-1][-1])
            for k, v in t.items():
                tr.setdefault(k, []).append(v)
        lo = int(want["offsets"][first])
        assert np.array_equal(np.concatenate(bs), want["bases"][lo:])
        assert np.array_equal(np.concatenate(offs), want["offsets"][first:] - np.uint64(lo))
        for k in ("ctg", "start", "end", "strand"):
            assert np.array_equal(np.concatenate(tr[k]), want[k][first:]), k
