"""GPU: reads the fast seeder declines (a byte other than A C G T) -- map_declined_kernel / seed_read_hybrid (mq_map_kernels.hpp): the fast
seeder on every stretch of >= 2,048 clean bytes, the general seeder on the rest.  k-min-mer tuples and hits identical to the oracle for
reads with a gap in the middle, at either end, Ns sprinkled, stretch borders inside homopolymer runs, lower case (folded and not), a
gap with the simulator's errors in it, nothing but N; several parameter sets and the 32-bit seeding variant.  Reference semantics:
KminmersIterator over the whole read (src/mers.rs:41-54), a non-ACGT byte hashes as 0."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mq():
    import mapquik_amd
    if mapquik_amd.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return mapquik_amd


def _reads(g, rng):
    def cl(n):
        a = int(rng.integers(0, g.size - n))
        return g[a:a + n].tobytes()
    noisy = bytearray(b"N" * 12000)
    for p in rng.integers(0, 12000, size=120):
        noisy[int(p)] = b"ACGT"[int(rng.integers(0, 4))]
    seqs = [
        cl(10000) + b"N" * 5000 + cl(10000),                       # a gap in the middle
        b"N" * 3000 + cl(20000),                                   # at the start
        cl(20000) + b"N" * 100,                                    # at the end
        b"N".join(cl(3000) for _ in range(7)),                     # sprinkled: every clean stretch is long enough
        b"N".join(cl(700) for _ in range(20)),                     # sprinkled: none is
        cl(6000) + b"A" * 3000 + b"N" + b"A" * 70 + cl(9000),      # stretch borders inside homopolymer runs
        cl(5000) + b"T" * 2113 + cl(2100) + b"NNN" + b"T" * 64 * 40 + cl(3000),
        cl(9000) + cl(2500).lower() + cl(9000),                    # lower case: other bytes unless folded
        bytes(noisy) + cl(12000),                                  # a gap with errors in it, then sequence
        cl(40) + b"N" + cl(59),                                    # short
        b"N" * 24000,                                              # nothing else
        cl(2047) + b"N" + cl(2048) + b"N" + cl(2049),              # around the stretch minimum
        cl(63) + b"N" + cl(64 * 33) + b"N" + cl(64 * 33 + 1),      # whole 64-byte blocks
        cl(30000).replace(b"G", b"R", 1),                          # one IUPAC byte in 30 kb
        cl(8000) + b"n" * 6000 + cl(8000),                         # a lower-case gap: one run when folding, and a run of another byte when not
        cl(5000) + b"nN" * 2500 + cl(5000),                        # one run when folding; 5,000 run heads in 5 KB when not (the walk's dense steps)
        cl(3000) + b"N" * 1023 + b"A" + b"N" * 1024 + b"C" + b"N" * 1025 + cl(3000),  # runs around the walk's 1-KB step
        cl(100) + b"N" * 40000 + cl(100),                          # a run long enough for the jump
    ]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return seqs, bases, offs


@pytest.mark.parametrize("variant", [0, 4, 24])
def test_declined_reads_stretch_by_stretch(mq, oracle, simlib, variant):
    rng = np.random.default_rng(11 + variant)
    g, off, names = simlib.make_genome([600_000], seed=5)
    seqs, bases, offs = _reads(g, rng)
    oracle.lib().mqo_set_variant(variant)
    try:
        for ps in (dict(), dict(use_hpc=False), dict(k=3, l=12, density=0.05), dict(k=7, l=64, density=0.02), dict(k=5, l=2, density=0.3),
                   dict(fold_case=True)):
            fold = ps.pop("fold_case", False)
            P, po = mq.Params(seeding_variant=variant, fold_case=fold, **ps), oracle.params(**ps)
            ix = mq.Index(P)
            ix.add_ref(0, names[0], g)
            ix.finalize()
            ox = oracle.Index()
            ox.add_ref(0, names[0], g, po)
            ref = [s.upper() if fold else s for s in seqs]  # to_ascii_uppercase (src/closures.rs:63,106) is the fold
            got = ix.kminmers_batch(bases, offs)
            n_fast, n_general = ix.last_map_path_counts()
            dirty = sum(1 for s in ref if len(s) >= po.l + po.k - 1 and any(c not in b"ACGT" for c in s))
            assert n_general == dirty and n_fast == sum(1 for s in ref if len(s) >= po.l + po.k - 1) - dirty, (n_fast, n_general, dirty)
            for i, s in enumerate(ref):
                w = oracle.kminmers(s, po) if len(s) >= po.l + po.k - 1 else np.zeros(0, dtype=oracle.kminmer_dtype)
                assert len(got[i]) == len(w), (variant, ps, fold, i, len(got[i]), len(w))
                for f in ("hash", "start", "end", "offset", "rev"):
                    assert np.array_equal(got[i][f].astype(np.uint64), w[f].astype(np.uint64)), (variant, ps, fold, i, f)
            hits = ix.map_batch(bases, offs)
            rb = np.frombuffer(b"".join(ref), dtype=np.uint8)
            want = ox.map_batch(rb, offs, po, threads=4)
            m = want["mapped"] != 0
            assert np.array_equal(hits["status"] == 1, m)
            for f in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
                assert np.array_equal(mq.hit_column(hits, f)[m], want[f][m].astype(np.uint64)), (variant, ps, fold, f)
            ix.close()
    finally:
        oracle.lib().mqo_set_variant(0)
