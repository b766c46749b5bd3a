"""GPU parity of the SEEDING VARIANTS (mq_params.flags bits 8..13, include/mapquik_hip.h): the k-min-mer iterator is a third-party crate
this image cannot build (rust-seq2kminmers, reference Cargo.toml:30; call sites src/mers.rs:22-27,53), so six of its decisions are
switchable in the product exactly as in the oracle (mqo_set_variant).  Every variant, on the same inputs, through the C ABI: k-min-mer
tuples, per-reference counts, index contents, hits and PAF bytes identical to the oracle run with the same variant -- through the
fast seeder, the general seeder (MQ_FORCE_GENERAL=1), the index build's segment views and the split pipeline."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = [0, 1, 2, 4, 6, 8, 16, 32, 24, 63]


@pytest.fixture(scope="module")
def mq():
    import mapquik_amd
    if mapquik_amd.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return mapquik_amd


@pytest.fixture()
def O(oracle):
    yield oracle
    oracle.lib().mqo_set_variant(0)


@pytest.fixture(scope="module")
def ecoli(simlib):
    return simlib.make_genome(simlib.ECOLI_LEN, seed=913)


def _cmp_kmm(got, want, tag):
    assert len(got) == len(want), (tag, len(got), len(want))
    for f in ("hash", "start", "end", "offset", "rev"):
        assert np.array_equal(got[f].astype(np.uint64), want[f].astype(np.uint64)), (tag, f)


def _want_kmm(O, s, po):
    return O.kminmers(s, po) if len(s) >= po.l + po.k - 1 else np.zeros(0, dtype=O.kminmer_dtype)


PARAM_SETS = [dict(), dict(k=8, l=16, g=100), dict(use_hpc=False), dict(k=3, l=12, density=0.05), dict(k=1, l=31), dict(k=7, l=64, density=0.02),
              dict(k=5, l=2, density=0.3)]


@pytest.mark.parametrize("force_general", [False, True])
@pytest.mark.parametrize("v", VARIANTS)
def test_kminmers_match_oracle_per_variant(mq, O, simlib, ecoli, monkeypatch, v, force_general):
    """test_kminmers_match_oracle of test_gpu_parity.py, for every variant: reads of 1..20,000 bases plus the edge sequences (empty, all one
    base, N runs, lower case, dinucleotide repeats, a homopolymer run longer than a tile side), seven parameter sets."""
    if force_general:
        monkeypatch.setenv("MQ_FORCE_GENERAL", "1")
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 16, seed=3, len_mean=6000, len_sd=4000, len_min=1, len_max=20000)
    rng = np.random.default_rng(5)
    edge = [b"", b"A", b"ACGT" * 8, b"A" * 5000, b"AC" * 3000, b"ACGTN" * 700, b"N" * 4000, bytes(rng.choice(list(b"ACGT"), size=35)),
            bytes(rng.choice(list(b"ACGTN"), size=9000)), b"A" * 3000 + bytes(rng.choice(list(b"ACGT"), size=3000)) + b"T" * 3000,
            bytes(rng.choice(list(b"AACCGGTTTT"), size=64 * 67 + 1)), bytes(rng.choice(list(b"AAAACCGT"), size=30000))]
    seqs = [bytes(reads["bases"][int(reads["offsets"][i]):int(reads["offsets"][i + 1])]) for i in range(16)] + edge
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    O.lib().mqo_set_variant(v)
    for ps in PARAM_SETS:
        P, po = mq.Params(seeding_variant=v, **ps), O.params(**ps)
        assert P.seeding_variant == v
        ix = mq.Index(P)
        got = ix.kminmers_batch(bases, offs)
        for i, s in enumerate(seqs):
            _cmp_kmm(got[i], _want_kmm(O, s, po), (v, ps, i))
        ix.close()
    if not force_general:
        n_fast, n_general = 0, 0
        ix = mq.Index(mq.Params(seeding_variant=v))
        ix.kminmers_batch(bases, offs)
        n_fast, n_general = ix.last_map_path_counts()
        assert n_fast >= 16  # the variants run on the fast seeder, not on a fallback
        ix.close()


def _map_both(mq, O, g, off, names, reads, ps, v):
    O.lib().mqo_set_variant(v)
    P, po = mq.Params(seeding_variant=v, **ps), O.params(**ps)
    ix, ox = mq.Index(P), O.Index()
    for r in range(off.size - 1):
        s = g[int(off[r]):int(off[r + 1])]
        assert ix.add_ref(r, names[r], s) == ox.add_ref(r, names[r], s, po), (v, r)
    assert ix.finalize() == ox.count()
    hits = ix.map_batch(reads["bases"], reads["offsets"])
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=4)
    return ix, ox, hits, want


def _cmp_hits(mq, hits, want):
    assert np.array_equal(hits["status"] == 1, want["mapped"] != 0)
    m = want["mapped"] != 0
    for a in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
        assert np.array_equal(mq.hit_column(hits, a)[m], want[a][m].astype(np.uint64)), a


@pytest.mark.parametrize("v", VARIANTS)
def test_map_ecoli_paf_identical_per_variant(mq, O, simlib, ecoli, v):
    """BASELINE config 1 stand-in (100 HiFi-like reads vs the 4,641,652-bp genome) under every variant: per-reference k-min-mer count,
    unique count, every PAF column and the PAF bytes identical to the oracle with the same variant; and the variant matters where it
    should (4, 8, 16 change the index; 1, 2, 32 do not on ordinary input)."""
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 100, seed=1)
    ix, ox, hits, want = _map_both(mq, O, g, off, names, reads, dict(), v)
    _cmp_hits(mq, hits, want)
    rn = simlib.read_names(reads, names)
    assert ix.paf_lines(rn, reads["offsets"], hits) == O.paf_lines(ox, rn, want)
    n_m, n_q60, n_wrong = simlib.mapeval(reads, want)
    assert n_q60 >= 90 and n_wrong == 0
    st = ix.stats()
    assert st["n_keys"] == ox.keys() and st["n_unique"] == ox.count()
    ix.close()


@pytest.mark.parametrize("v", [4, 8, 16, 63])
def test_map_k8_l16_split_pipeline_and_general_per_variant(mq, O, simlib, ecoli, monkeypatch, v):
    """The other kernels that seed: MQ_PIPELINE=split (seed_reads_kernel / seed_general_kernel / map_lists_kernel) and the general
    seeder for everything (reference segments too), at -k 8 -l 16 (example/run_ecoli.sh:26)."""
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 60, seed=2)
    ps = dict(k=8, l=16, g=100)
    ix, ox, hits, want = _map_both(mq, O, g, off, names, reads, ps, v)
    _cmp_hits(mq, hits, want)
    ix.close()
    monkeypatch.setenv("MQ_PIPELINE", "split")
    ix, ox, hits2, want = _map_both(mq, O, g, off, names, reads, ps, v)
    assert np.array_equal(hits2.view(np.uint8), hits.view(np.uint8))
    ix.close()
    monkeypatch.delenv("MQ_PIPELINE")
    monkeypatch.setenv("MQ_FORCE_GENERAL", "1")
    ix, ox, hits3, want = _map_both(mq, O, g, off, names, reads, ps, v)
    assert np.array_equal(hits3.view(np.uint8), hits.view(np.uint8))
    ix.close()


@pytest.mark.parametrize("v", [8, 16, 28, 63])
def test_reference_segment_borders_per_variant(mq, O, simlib, monkeypatch, v):
    """The index build's segment views (22,528 bases + halo through the fast seeder, declined views through the general one) with the
    position variants: runs across the borders, N across a border, a run longer than the halo; tiny list regions (redo path)."""
    SEG = 22528
    g, off, _ = simlib.make_genome([5 * SEG + 777, 3 * SEG, 3 * SEG + 1, SEG + 2048, 40], seed=23, repeat_frac=0.1, tandem_frac=0.05)
    seqs = [g[int(off[r]):int(off[r + 1])].copy() for r in range(off.size - 1)]
    a = seqs[0]
    a[SEG - 3:SEG + 5] = ord("N")
    a[2 * SEG - 40:2 * SEG + 40] = ord("A")
    a[3 * SEG - 1] = a[3 * SEG]
    a[4 * SEG:4 * SEG + 3000] = ord("C")
    a[4 * SEG + 3000] = ord("G")
    b = seqs[1]
    b[SEG - 2:SEG + 1] = ord("T")  # a run that ends ON the border: variant 8's position of the window starting there

    def both(ps):
        O.lib().mqo_set_variant(v)
        P, po = mq.Params(seeding_variant=v, **ps), O.params(**ps)
        ix, ox = mq.Index(P), O.Index()
        hs = []
        for r, s in enumerate(seqs):
            assert ix.add_ref(r, "c%d" % r, s) == ox.add_ref(r, "c%d" % r, s, po), (v, ps, r)
            if s.size >= po.l + po.k - 1:
                hs.append(O.kminmers(s, po)["hash"])
        assert ix.finalize() == ox.count()
        st = ix.stats()
        assert st["n_keys"] == ox.keys()
        keys = np.unique(np.concatenate(hs))
        if keys.size > 20000:
            keys = keys[np.random.default_rng(3).choice(keys.size, 20000, replace=False)]
        found, ent, ids = ix.lookup(keys)
        for i, h in enumerate(keys):
            e = ox.get(int(h))
            assert bool(found[i]) == (e is not None)
            if e is not None:
                assert (int(ids[i]), int(ent[i]["start"]), int(ent[i]["end"]), int(ent[i]["offset"]), int(ent[i]["rev"])) == \
                       (int(e["id"]), int(e["start"]), int(e["end"]), int(e["offset"]), int(e["rc"])), (v, ps, i)
        ix.close()
        return st

    for ps in (dict(), dict(k=3, l=12, density=0.05)):
        st = both(ps)
        monkeypatch.setenv("MQ_FORCE_GENERAL", "1")
        assert both(ps) == st
        monkeypatch.delenv("MQ_FORCE_GENERAL")
        monkeypatch.setenv("MQ_REF_CAP", "8")
        assert both(ps) == st
        monkeypatch.delenv("MQ_REF_CAP")


def _rand_seq(n, seed):
    rng = random.Random(seed)
    return "".join(rng.choice("ACGT") for _ in range(n)).encode()


def _hash_equal_to_its_bound(O, want_f32_below):
    """tests/test_oracle_variants.py: one l-mer (l = 12, k = 1, no HPC) whose canonical hash v has its low 11 bits clear, so that
    density = v / 2^64 is exact in f64 and the 64-bit bound is v itself."""
    l = 12
    for seed in range(400000):
        s = _rand_seq(l, 1000 + seed)
        v = int(O.lib().mqo_ntc64(s, 0, l))
        if v & 0x7FF or v == 0:
            continue
        d = v / 2.0 ** 64
        if int(O.lib().mqo_density_bound(d)) != v:
            continue
        if want_f32_below and float(np.float32(d)) >= d:
            continue
        return s, d, l
    pytest.skip("no suitable l-mer found")


def _count(mq, seq, ps, v):
    ix = mq.Index(mq.Params(seeding_variant=v, **ps))
    b = np.frombuffer(seq, dtype=np.uint8)
    got = ix.kminmers_batch(b, np.array([0, b.size], dtype=np.uint64))
    ix.close()
    return got[0]


def test_bits_1_and_2_move_the_bound_on_the_gpu(mq, O):
    """The measure-zero events the bound variants differ on, built on purpose: a hash EQUAL to its bound is kept by `<=` and dropped by
    `<` (bit 1); an f32 bound that rounds below that hash drops it too (bit 2).  Longer sequences around the l-mer go through the
    fast seeder (>= 16 bases), the bare l-mer through the general one."""
    s, d, l = _hash_equal_to_its_bound(O, False)
    ps = dict(k=1, l=l, density=d, use_hpc=False)
    assert len(_count(mq, s, ps, 0)) == 1 and len(_count(mq, s, ps, 1)) == 0
    pad = _rand_seq(4000, 77)
    for v in (0, 1):
        O.lib().mqo_set_variant(v)
        _cmp_kmm(_count(mq, pad + s + pad, ps, v), O.kminmers(pad + s + pad, O.params(**ps)), v)
    O.lib().mqo_set_variant(0)
    n0 = len(O.kminmers(pad + s + pad, O.params(**ps)))
    O.lib().mqo_set_variant(1)
    assert len(O.kminmers(pad + s + pad, O.params(**ps))) == n0 - 1  # exactly the one hash that sits on the bound
    O.lib().mqo_set_variant(0)
    s2, d2, l2 = _hash_equal_to_its_bound(O, True)
    ps2 = dict(k=1, l=l2, density=d2, use_hpc=False)
    assert len(_count(mq, s2, ps2, 0)) == 1 and len(_count(mq, s2, ps2, 2)) == 0
    # density 0 with the strict test: nothing at all can pass (`hash < 0`)
    assert len(_count(mq, pad, dict(k=1, l=12, density=0.0), 1)) == 0


def test_bit_32_palindromic_tuples_on_the_gpu(mq, O):
    seq = _rand_seq(30000, 6)
    for ps in (dict(k=1), dict(k=1, l=16), dict(k=2, l=8, density=0.5)):
        a, b = _count(mq, seq, ps, 0), _count(mq, seq, ps, 32)
        O.lib().mqo_set_variant(32)
        _cmp_kmm(b, O.kminmers(seq, O.params(**ps)), ps)
        O.lib().mqo_set_variant(0)
        _cmp_kmm(a, O.kminmers(seq, O.params(**ps)), ps)
        if ps["k"] == 1:
            assert len(a) > 100 and not a["rev"].any() and b["rev"].all()
    # a periodic sequence gives palindromic 2- and 3-tuples (equal neighbouring minimizers): the fixed-k and the general tuple hash
    per = (b"ACGGTCA" * 4000)
    for ps in (dict(k=2, l=7, density=1.0, use_hpc=False), dict(k=5, l=7, density=1.0, use_hpc=False), dict(k=7, l=7, density=1.0, use_hpc=False),
               dict(k=8, l=7, density=1.0, use_hpc=False), dict(k=3, l=7, density=1.0, use_hpc=False)):
        for v in (0, 32):
            O.lib().mqo_set_variant(v)
            _cmp_kmm(_count(mq, per, ps, v), O.kminmers(per, O.params(**ps)), (ps, v))
    O.lib().mqo_set_variant(32)
    assert O.kminmers(per, O.params(k=5, l=7, density=1.0, use_hpc=False))["rev"].any()
    O.lib().mqo_set_variant(0)


def test_rejected_flag_combinations(mq):
    with pytest.raises(mq.MapquikError):
        mq.Index(mq.Params(l=1, seeding_variant=8))       # the run's end is read off the window's second base
    p = mq.Params()
    p.flags |= 1 << 20                                    # an undefined bit
    with pytest.raises(mq.MapquikError):
        mq.Index(p)
    with pytest.raises(ValueError):
        mq.Params(seeding_variant=64)
    ix = mq.Index(mq.Params(l=1, seeding_variant=16 | 32 | 4))  # l = 1 is fine without bit 8
    ix.close()


def test_variant_survives_save_load_and_clone(mq, O, simlib, tmp_path):
    g, off, names = simlib.make_genome([300000, 200000], seed=5, repeat_frac=0.1)
    reads = simlib.make_reads(g, off, 200, seed=8)
    v = 4 | 8 | 16
    ix, ox, hits, want = _map_both(mq, O, g, off, names, reads, dict(), v)
    _cmp_hits(mq, hits, want)
    p = str(tmp_path / "v.mqx")
    ix.save(p)
    ix2 = mq.Index.load(p)
    assert np.array_equal(ix2.map_batch(reads["bases"], reads["offsets"]).view(np.uint8), hits.view(np.uint8))
    rep = ix.clone(0)
    assert np.array_equal(rep.map_batch(reads["bases"], reads["offsets"]).view(np.uint8), hits.view(np.uint8))
    for x in (ix, ix2, rep):
        x.close()


def test_native_driver_seeding_variant_flag(mq, O, simlib, tmp_path):
    """`mapquik --seeding-variant v` (what tools/check_against_upstream.sh's step 2 runs with the matched variant): PAF bytes = the oracle's
    under that variant, and different from variant 0's."""
    import subprocess
    from mapquik_amd import build as B
    B.build_cli()
    g, off, names = simlib.make_genome([400000, 250000], seed=11, repeat_frac=0.05)
    reads = simlib.make_reads(g, off, 300, seed=4)
    rn = simlib.read_names(reads, names)
    offs = reads["offsets"]
    ref = tmp_path / "ref.fa"
    with open(ref, "wb") as f:
        for r in range(2):
            f.write(b">" + names[r].encode() + b"\n" + g[int(off[r]):int(off[r + 1])].tobytes() + b"\n")
    rd = tmp_path / "reads.fa"
    with open(rd, "wb") as f:
        for i, n in enumerate(rn):
            f.write(b">" + n.encode() + b"\n" + reads["bases"][int(offs[i]):int(offs[i + 1])].tobytes() + b"\n")
    texts = {}
    for v in (0, 24):
        O.lib().mqo_set_variant(v)
        po = O.params()
        ox = O.Index()
        for r in range(2):
            ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po)
        want = ox.map_batch(reads["bases"], offs, po, threads=4)
        want_txt = "".join(ln + "\n" for ln in O.paf_lines(ox, rn, want))
        prefix = str(tmp_path / ("out%d" % v))
        r_ = subprocess.run([B.CLI, str(rd), "--reference", str(ref), "-p", prefix, "--threads", "2", "--seeding-variant", str(v)],
                            capture_output=True, text=True, timeout=600)
        assert r_.returncode == 0, r_.stderr[-2000:]
        assert open(prefix + ".paf").read() == want_txt and len(want_txt) > 1000
        assert ("Seeding variant 24" in r_.stdout) == (v == 24)
        texts[v] = want_txt
    assert texts[0] != texts[24]
