"""GPU parity on the reference's sharp edges, with counters asserting that the inputs really reach them:
Match::check's precedence quirk (src/match.rs:39-43), top-two tie => None (src/mers.rs:104-108), `as i32` casts in the gap
tests on a contig longer than 2^31 bases (src/chain.rs:132-142), find_coords clipping (src/mers.rs:131-183)."""
import numpy as np
import pytest

import directed as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mq():
    import mapquik_amd
    if mapquik_amd.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return mapquik_amd


def _build(mq, oracle, g, off, names, ps, threads=8):
    P, po = mq.Params(**ps), oracle.params(**ps)
    ix, ox = mq.Index(P), oracle.Index()
    for r in range(off.size - 1):
        ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])])
    ox.build_mt(g, off, names, po, threads)
    assert ix.finalize() == ox.count()
    return ix, ox, po


def _cmp(hits, want):
    assert np.array_equal(hits["status"] == 1, want["mapped"] != 0)
    m = want["mapped"] != 0
    import mapquik_amd
    for a in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
        assert np.array_equal(mapquik_amd.hit_column(hits, a)[m], want[a][m].astype(np.uint64)), a


def test_usize_wrap_in_find_coords_on_gpu(mq, oracle, simlib):
    """Columns 3 and 4 are usize in the reference and wrap (src/mers.rs:131-183, release build) when a run the precedence quirk
    extended onto another, longer reference ends beyond the end of the reference it is keyed under: mq_hit carries all 64 bits
    and the PAF text is the reference's (found by the fuzz test, seed 22 case 1469)."""
    g, off, names, bases, offs, ps = D.usize_wrap_case(oracle, simlib)
    ix, ox, po = _build(mq, oracle, g, off, names, ps)
    want = ox.map_batch(bases, offs, po, threads=4)
    n_wrapped = int(((want["mapped"] != 0) & (want["q_end"] >= (1 << 63))).sum())
    assert n_wrapped >= 10
    hits = ix.map_batch(bases, offs)
    _cmp(hits, want)
    assert int((hits["q_end_hi"] == 0xFFFFFFFF).sum()) == n_wrapped
    names_q = ["q%d" % i for i in range(offs.size - 1)]
    got = ix.paf_lines(names_q, offs, hits)
    assert got == oracle.paf_lines(ox, names_q, want)
    assert sum(1 for ln in got if int(ln.split("\t")[3]) >= (1 << 63)) == n_wrapped


def test_match_check_precedence_quirk_on_gpu(mq, oracle, simlib):
    g, off, names, bases, offs, ps = D.quirk_case(oracle, simlib)
    ix, ox, po = _build(mq, oracle, g, off, names, ps)
    want, diag = ox.map_batch_diag(bases, offs, po, threads=4)
    assert (diag["quirk_cross_ref"] > 0).sum() >= 10      # forward Matches extended across the reference boundary
    hits = ix.map_batch(bases, offs)
    _cmp(hits, want)
    names_q = ["q%d" % i for i in range(offs.size - 1)]
    assert ix.paf_lines(names_q, offs, hits) == oracle.paf_lines(ox, names_q, want)
    # the general streaming path must reproduce it too (reads with an N take it): damage a base far from the junction
    b2 = bases.copy()
    for i in range(offs.size - 1):
        b2[int(offs[i]) + 3] = ord("N")
    want2, diag2 = ox.map_batch_diag(b2, offs, po, threads=4)
    assert (diag2["quirk_cross_ref"] > 0).sum() >= 8
    hits2 = ix.map_batch(b2, offs)
    _cmp(hits2, want2)
    assert ix.last_map_path_counts()[1] == offs.size - 1


def test_top_two_tie_is_unmapped_on_gpu(mq, oracle, simlib):
    g, off, names, bases, offs, ps = D.tie_case(oracle, simlib)
    ix, ox, po = _build(mq, oracle, g, off, names, ps)
    want, diag = ox.map_batch_diag(bases, offs, po, threads=8)
    ties = np.nonzero(diag["tie"] != 0)[0]
    assert ties.size >= 10 and (diag["n_candidates"] > 1).sum() > 1000
    hits = ix.map_batch(bases, offs)
    _cmp(hits, want)
    assert (hits["status"][ties] == 0).all()
    # near-ties: the same reads with one more / one fewer matching k-min-mer on one side stay mapped and identical
    near = np.nonzero((diag["tie"] == 0) & (diag["n_candidates"] == 2))[0][:200]
    assert (hits["status"][near] == 1).mean() > 0.99


def test_contig_longer_than_2_pow_31_on_gpu(mq, oracle, simlib):
    g, off, names, reads = D.wrap_case(simlib, n_reads=600)
    ix, ox, po = _build(mq, oracle, g, off, names, dict(), threads=2)
    want, diag = ox.map_batch_diag(reads["bases"], reads["offsets"], po, threads=8)
    assert (diag["i32_wrap"] > 0).sum() >= 100            # gap tests saw coordinates >= 2^31
    assert (want["r_end"] >= (1 << 31)).sum() >= 100 and (want["r_start"] < (1 << 31)).sum() >= 100
    hits = ix.map_batch(reads["bases"], reads["offsets"])
    _cmp(hits, want)
    rn = simlib.read_names(reads, names)
    assert ix.paf_lines(rn, reads["offsets"], hits) == oracle.paf_lines(ox, rn, want)
    n_m, n_q60, n_wrong = simlib.mapeval(reads, want)
    assert n_q60 >= 590 and n_wrong == 0
    # a sequence of 2^32 bases or more is refused (the reference's usize has no such limit; ours is documented)
    with pytest.raises(mq.MapquikError):
        ix2 = mq.Index(mq.Params())
        ix2.add_ref_device(0, "huge", 0x1000, 1 << 32)


def test_find_coords_clipping_on_gpu(mq, oracle, simlib):
    """Reads hanging over both ends of short contigs: all four clip branches of find_coords (src/mers.rs:131-183)."""
    g, off, names = simlib.make_genome([30000, 26000, 41000], seed=77)
    rng = np.random.default_rng(3)
    comp = np.zeros(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    seqs = []
    for r in range(3):
        s = g[int(off[r]):int(off[r + 1])]
        for _ in range(40):
            junk_l = rng.choice(list(b"ACGT"), size=int(rng.integers(50, 4000))).astype(np.uint8)
            junk_r = rng.choice(list(b"ACGT"), size=int(rng.integers(50, 4000))).astype(np.uint8)
            a = np.concatenate([junk_l, s[:int(rng.integers(8000, 20000))]])                  # overhangs the contig start
            b = np.concatenate([s[s.size - int(rng.integers(8000, 20000)):], junk_r])         # overhangs the contig end
            c = np.concatenate([junk_l, s, junk_r])                                            # overhangs both
            for x in (a, b, c):
                seqs.append(x if rng.integers(0, 2) else comp[x[::-1]])
    bases = np.concatenate(seqs)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([x.size for x in seqs])
    ix, ox, po = _build(mq, oracle, g, off, names, dict())
    want, diag = ox.map_batch_diag(bases, offs, po, threads=8)
    assert diag["clip_start"].sum() >= 60 and diag["clip_end"].sum() >= 60
    assert ((want["mapped"] != 0) & (want["rc"] != 0)).sum() >= 60 and ((want["mapped"] != 0) & (want["rc"] == 0)).sum() >= 60
    hits = ix.map_batch(bases, offs)
    _cmp(hits, want)
