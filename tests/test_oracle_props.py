"""Oracle self-consistency and the committed real-read fixture (CPU only)."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fasta(path):
    recs, name = [], None
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
        elif name is not None:
            recs.append((name, line.encode()))
            name = None
    return recs


def test_rolling_equals_definition(oracle):
    rng = np.random.default_rng(3)
    for hpc in (True, False):
        for l in (1, 2, 16, 31, 32, 63, 64):
            p = oracle.params(l=l, density=0.2, use_hpc=hpc)
            for alphabet in (b"ACGT", b"ACGTN", b"AACCCCGT"):
                s = bytes(rng.choice(list(alphabet), size=int(rng.integers(1, 700))))
                a, b = oracle.minimizers(s, p), oracle.minimizers(s, p, naive=True)
                assert np.array_equal(a, b), (hpc, l, alphabet)


def test_kminmer_structure(oracle):
    rng = np.random.default_rng(4)
    s = bytes(rng.choice(list(b"ACGT"), size=30000))
    for ps in (dict(), dict(k=8, l=16), dict(k=1, l=31), dict(use_hpc=False, density=0.05)):
        p = oracle.params(**ps)
        mz = oracle.minimizers(s, p)
        km = oracle.kminmers(s, p)
        k, l = int(p.k), int(p.l)
        assert len(km) == max(0, len(mz) - k + 1)
        assert np.array_equal(km["offset"], np.arange(len(km), dtype=np.uint64))
        assert np.array_equal(km["start"], mz["pos"][:len(km)])
        assert np.array_equal(km["end"], mz["pos"][k - 1:] + np.uint64(l - 1))
        assert (np.diff(mz["pos"].astype(np.int64)) > 0).all()
        assert (mz["hash"] <= np.uint64(oracle.lib().mqo_density_bound(p.density))).all()


def test_reverse_complement_gives_same_kminmer_hashes(oracle):
    """Canonical hashing: the reverse complement of a sequence yields the same multiset of k-min-mer hashes with rev flipped
    (HPC off so that run heads map one to one)."""
    rng = np.random.default_rng(5)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    s = rng.choice([65, 67, 71, 84], size=20000).astype(np.uint8)
    rc = np.array([comp[int(b)] for b in s[::-1]], dtype=np.uint8)
    p = oracle.params(use_hpc=False, density=0.02)
    a, b = oracle.kminmers(s, p), oracle.kminmers(rc, p)
    assert len(a) == len(b) and len(a) > 100
    assert np.array_equal(a["hash"], b["hash"][::-1])
    pal = a["rev"] == b["rev"][::-1]  # only palindromic tuples keep rev=0 on both strands
    assert pal.sum() <= 1


def test_short_and_degenerate_sequences(oracle):
    p = oracle.params()
    for s in (b"", b"A", b"ACGT" * 8, b"A" * 5000, b"AC" * 20):
        assert len(oracle.kminmers(s, p)) == 0
    ix = oracle.Index()
    rec = ix.find_matches(b"ACGT" * 5, p)
    assert int(rec["mapped"]) == 0


def test_real_read_fixture_matches_oracle(oracle):
    exp = json.load(open(os.path.join(GOLD, "ecoli5_kminmers.json")))
    recs = _fasta(os.path.join(GOLD, "nearperfect-ecoli.5.fa"))
    assert len(recs) == 5
    for case in exp["cases"]:
        p = oracle.params(**case["params"])
        for (name, seq), e in zip(recs, case["reads"]):
            assert name == e["id"] and len(seq) == e["len"]
            km = oracle.kminmers(seq, p)
            assert len(km) == e["n_kminmers"]
            assert hashlib.sha256(km.tobytes()).hexdigest() == e["sha256_of_tuples"]


def _fasta_gz(path):
    import gzip
    recs, name = [], None
    for line in gzip.open(path, "rt"):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
        elif name is not None:
            recs.append((name, line.encode()))
            name = None
    return recs


def test_real_read_fixture_100_matches_oracle(oracle):
    """All 100 reads of the reference's example/nearperfect-ecoli.100.fa (committed as data) vs the committed digests."""
    exp = json.load(open(os.path.join(GOLD, "ecoli100_kminmers.json")))
    recs = _fasta_gz(os.path.join(GOLD, "nearperfect-ecoli.100.fa.gz"))
    assert len(recs) == 100 and sum(len(s) for _, s in recs) == 2289087
    for case in exp["cases"]:
        p = oracle.params(**case["params"])
        for (name, seq), e in zip(recs, case["reads"]):
            assert name == e["id"] and len(seq) == e["len"]
            km = oracle.kminmers(seq, p)
            assert len(km) == e["n_kminmers"]
            assert hashlib.sha256(km.tobytes()).hexdigest() == e["sha256_of_tuples"]


def test_end_to_end_mapeval_and_index_order_independence(oracle, simlib):
    g, off, names = simlib.make_genome([300000, 200000], seed=17, repeat_frac=0.2, tandem_frac=0.05)
    reads = simlib.make_reads(g, off, 120, seed=2, len_mean=9000, len_sd=3000)
    p = oracle.params()
    a, b = oracle.Index(), oracle.Index()
    for r in (0, 1):
        a.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], p)
    b.build_mt(g, off, names, p, threads=2)  # other insertion order / threading: same final map (src/index.rs:94-104)
    assert a.count() == b.count() and a.keys() == b.keys()
    ra = a.map_batch(reads["bases"], reads["offsets"], p, threads=1)
    rb = b.map_batch(reads["bases"], reads["offsets"], p, threads=3)
    assert np.array_equal(ra.view(np.uint8), rb.view(np.uint8))
    n_mapped, n_q60, n_wrong = simlib.mapeval(reads, ra)
    assert n_mapped >= 100 and n_wrong == 0
    rn = simlib.read_names(reads, names)
    lines = oracle.paf_lines(a, rn, ra)
    assert len(lines) == n_mapped and all(len(x.split("\t")) == 12 for x in lines)
    f = lines[0].split("\t")
    assert f[6] == f[10]  # column 11 repeats r_len (src/mers.rs:181)
