"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.  Bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mq():
    import mapquik_amd
    if mapquik_amd.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return mapquik_amd


@pytest.fixture(scope="module")
def ecoli(simlib):
    g, off, names = simlib.make_genome(simlib.ECOLI_LEN, seed=913)
    return g, off, names


def _cmp_kmm(got, want, tag):
    assert len(got) == len(want), (tag, len(got), len(want))
    for f in ("hash", "start", "end", "offset", "rev"):
        assert np.array_equal(got[f].astype(np.uint64), want[f].astype(np.uint64)), (tag, f)


PARAM_SETS = [dict(), dict(k=8, l=16, g=100), dict(use_hpc=False), dict(k=3, l=12, density=0.05), dict(k=1, l=31),
              dict(k=7, l=64, density=0.02), dict(k=32, l=5, density=0.2), dict(k=5, l=1, density=0.3)]


@pytest.mark.parametrize("ps", PARAM_SETS)
def test_kminmers_match_oracle(mq, oracle, simlib, ecoli, ps):
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 24, seed=3, len_mean=6000, len_sd=4000, len_min=1, len_max=20000)
    bases, offs = reads["bases"], reads["offsets"]
    P = mq.Params(**ps)
    po = oracle.params(**ps)
    ix = mq.Index(P)
    got = ix.kminmers_batch(bases, offs)
    for i in range(offs.size - 1):
        s = bases[int(offs[i]):int(offs[i + 1])]
        want = oracle.kminmers(s, po) if s.size >= po.l + po.k - 1 else np.zeros(0, dtype=oracle.kminmer_dtype)
        _cmp_kmm(got[i], want, (ps, i))


def test_kminmers_edge_sequences(mq, oracle):
    rng = np.random.default_rng(5)
    seqs = [b"", b"A", b"ACGT" * 8, b"A" * 5000, b"AC" * 3000, b"ACGTN" * 700, b"N" * 4000,
            bytes(rng.choice(list(b"ACGT"), size=35)), bytes(rng.choice(list(b"ACGT"), size=34)),
            bytes(rng.choice(list(b"ACGTN"), size=9000)), bytes(rng.choice(list(b"ACGTacgtRYKM"), size=7000)),
            b"A" * 3000 + bytes(rng.choice(list(b"ACGT"), size=3000)) + b"T" * 3000,
            bytes(rng.choice(list(b"AACCGGTTTT"), size=64 * 67 + 1))]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    for ps in (dict(density=0.1), dict(density=1.0, k=2, l=4), dict(density=0.1, use_hpc=False)):
        P, po = mq.Params(**ps), oracle.params(**ps)
        got = mq.Index(P).kminmers_batch(bases, offs)
        for i, s in enumerate(seqs):
            want = oracle.kminmers(s, po) if len(s) >= po.l + po.k - 1 else np.zeros(0, dtype=oracle.kminmer_dtype)
            _cmp_kmm(got[i], want, (ps, i))


@pytest.mark.parametrize("ps", [dict(), dict(k=8, l=16, g=100), dict(use_hpc=False, density=0.02)])
def test_index_matches_oracle(mq, oracle, simlib, ps):
    g, off, names = simlib.make_genome([300000, 200001, 40, 70000], seed=11, repeat_frac=0.15, tandem_frac=0.03)
    P, po = mq.Params(**ps), oracle.params(**ps)
    ix, ox = mq.Index(P), oracle.Index()
    all_h = []
    for r in range(off.size - 1):
        s = g[int(off[r]):int(off[r + 1])]
        n_gpu = ix.add_ref(r, names[r], s)
        n_cpu = ox.add_ref(r, names[r], s, po)
        assert n_gpu == n_cpu
        all_h.append(oracle.kminmers(s, po)["hash"] if s.size >= po.l + po.k - 1 else np.zeros(0, np.uint64))
    assert ix.finalize() == ox.count()
    st = ix.stats()
    assert st["n_keys"] == ox.keys() and st["n_unique"] == ox.count()
    hs = np.unique(np.concatenate(all_h))
    rng = np.random.default_rng(1)
    q = np.concatenate([hs, rng.integers(0, 2**63, size=1000, dtype=np.uint64)])
    found, ent, ids = ix.lookup(q)
    for i, h in enumerate(q):
        e = ox.get(int(h))
        assert bool(found[i]) == (e is not None)
        if e is not None:
            assert (int(ids[i]), int(ent[i]["start"]), int(ent[i]["end"]), int(ent[i]["offset"]), int(ent[i]["rev"])) == \
                   (int(e["id"]), int(e["start"]), int(e["end"]), int(e["offset"]), int(e["rc"]))


def test_crowded_table_long_probe_walks(mq, oracle, simlib, monkeypatch, tmp_path):
    """The bucket table at half load (MQ_TABLE_FACTOR=2: two slots per inserted k-min-mer, rounded up to a power of two), where a
    lookup often finds both ways of its home bucket taken and walks on for several buckets (the product default, load <= 1/8, walks
    for about one lookup in a hundred, and then one bucket): every key of the reference, absent keys and the key 0 answer like the
    oracle's map (src/index.rs:118-126), the reads map identically, the index survives save -> load, and probes per lookup are
    well above the default's."""
    monkeypatch.setenv("MQ_TABLE_FACTOR", "2")
    ps = dict(k=3, l=12, density=0.05)
    g, off, names = simlib.make_genome([700000, 500000], seed=23, repeat_frac=0.2, tandem_frac=0.03)
    reads = simlib.make_reads(g, off, 500, seed=4, len_mean=9000, len_sd=3000)
    ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads, ps)
    _cmp_hits(hits, want)
    st = ix.stats()
    assert st["n_keys"] == ox.keys() and st["table_slots"] <= 4 * st["n_kminmers"] and st["n_keys"] > 0.2 * st["table_slots"]
    po = oracle.params(**ps)
    hs = np.unique(np.concatenate([oracle.kminmers(g[int(off[r]):int(off[r + 1])], po)["hash"] for r in range(2)]))
    rng = np.random.default_rng(2)
    q = np.concatenate([hs, rng.integers(0, 2**63, size=5000, dtype=np.uint64), np.zeros(1, np.uint64)])
    for index in (ix,):
        found, ent, ids = index.lookup(q)
        want_found = np.array([ox.get(int(h)) is not None for h in q])
        assert np.array_equal(found.astype(bool), want_found)
        for i in np.flatnonzero(want_found)[::37]:
            e = ox.get(int(q[i]))
            assert (int(ids[i]), int(ent[i]["start"]), int(ent[i]["end"]), int(ent[i]["offset"]), int(ent[i]["rev"])) == \
                   (int(e["id"]), int(e["start"]), int(e["end"]), int(e["offset"]), int(e["rc"]))
    p = str(tmp_path / "crowded.mqx")
    ix.save(p)
    ix2 = mq.Index.load(p)
    assert ix2.stats() == st
    assert np.array_equal(ix2.map_batch(reads["bases"], reads["offsets"]).view(np.uint8), hits.view(np.uint8))
    found2, _, _ = ix2.lookup(q)
    assert np.array_equal(found2, found)
    # probes per lookup on this batch (instrumented launch): a crowded table makes the walk visible
    from hipmem import DevBuf
    offs = reads["offsets"]
    d_b, d_o, d_h = DevBuf.from_numpy(reads["bases"]), DevBuf.from_numpy(offs), DevBuf((offs.size - 1) * 48)
    lookups, extra = ix.probe_stats(d_b.ptr, d_o.ptr, offs.size - 1, int(offs[-1]), d_h.ptr)
    assert lookups > 10000 and extra / lookups > 0.15  # the default table: 0.085
    assert np.array_equal(d_h.to_numpy(mq.hit_dtype, offs.size - 1).view(np.uint8), hits.view(np.uint8))


def _map_both(mq, oracle, g, off, names, reads, ps, variant=0, fast_kh=False):
    P, po = mq.Params(seeding_variant=variant, fast_kh=fast_kh, **ps), oracle.params(**ps)
    ix, ox = mq.Index(P), oracle.Index()
    for r in range(off.size - 1):
        s = g[int(off[r]):int(off[r + 1])]
        assert ix.add_ref(r, names[r], s) == ox.add_ref(r, names[r], s, po)
    assert ix.finalize() == ox.count()
    hits = ix.map_batch(reads["bases"], reads["offsets"])
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=4)
    return ix, ox, hits, want


def _cmp_hits(hits, want):
    assert np.array_equal(hits["status"] == 1, want["mapped"] != 0)
    m = want["mapped"] != 0
    for a, b in (("ref_id", "ref_id"), ("rc", "rc"), ("mapq", "mapq"), ("q_start", "q_start"), ("q_end", "q_end"),
                 ("r_start", "r_start"), ("r_end", "r_end"), ("score", "score")):
        import mapquik_amd
        assert np.array_equal(mapquik_amd.hit_column(hits, a)[m], want[b][m].astype(np.uint64)), a


@pytest.mark.parametrize("ps", [dict(), dict(k=8, l=16, g=100)])
def test_map_ecoli_paf_identical(mq, oracle, simlib, ecoli, ps):
    """BASELINE config 1 stand-in: 100 HiFi-like reads vs the 4,641,652-bp genome; PAF bytes identical."""
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 100, seed=1)
    ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads, ps)
    _cmp_hits(hits, want)
    rn = simlib.read_names(reads, names)
    assert ix.paf_lines(rn, reads["offsets"], hits) == oracle.paf_lines(ox, rn, want)
    n_m, n_q60, n_wrong = simlib.mapeval(reads, want)
    assert n_q60 >= 95 and n_wrong == 0


def test_map_repetitive_multi_contig(mq, oracle, simlib):
    """Repeats => tombstones, misses, many short Matches, several candidate references, ties."""
    g, off, names = simlib.make_genome([400000, 350000, 250000, 1000, 20], seed=21, repeat_frac=0.6, tandem_frac=0.1, div=0.005)
    reads = simlib.make_reads(g, off, 600, seed=9, len_mean=9000, len_sd=5000, len_min=10, len_max=25000, err=0.02)
    for ps in (dict(), dict(k=3, l=15, density=0.03, c=2, s=5, g=500), dict(k=2, l=10, density=0.05, g=50)):
        ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads, ps)
        _cmp_hits(hits, want)


def test_map_chain_multichunk_path(mq, oracle, simlib, monkeypatch):
    """MQ_CHAIN_CHUNK=4 builds the chain stage with 4-lane chunks so ordinary reads take the multi-chunk path."""
    monkeypatch.setenv("MQ_CHAIN_CHUNK", "4")
    g, off, names = simlib.make_genome([300000, 300000], seed=33, repeat_frac=0.5, tandem_frac=0.1, div=0.01)
    reads = simlib.make_reads(g, off, 300, seed=2, len_mean=12000, len_sd=4000, err=0.03)
    ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads, dict(k=3, l=15, density=0.03))
    _cmp_hits(hits, want)


def test_match_overflow_is_loud_on_device_form_and_retried_on_host_form(mq, oracle, simlib, monkeypatch):
    """MQ_MATCH_CAP=1: almost every read has more Match runs than the scratch.  The device-resident entry point reports
    MQ_HIT_OVERFLOW (never a wrong line); the host-buffer entry point maps those reads again with a worst-case scratch."""
    import ctypes as C
    monkeypatch.setenv("MQ_MATCH_CAP", "1")
    g, off, names = simlib.make_genome([200000], seed=5, repeat_frac=0.3)
    reads = simlib.make_reads(g, off, 50, seed=2, len_mean=12000, err=0.03)
    ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads, dict(k=3, l=15, density=0.03))
    _cmp_hits(hits, want)  # host form: retried, complete and identical
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    bases, offs = reads["bases"], reads["offsets"]
    db, do, dout = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(db), bases.size) == 0 and hip.hipMalloc(C.byref(do), offs.size * 8) == 0 and hip.hipMalloc(C.byref(dout), 50 * 48) == 0
    hip.hipMemcpy(db, bases.ctypes.data, bases.size, 1)
    hip.hipMemcpy(do, offs.ctypes.data, offs.size * 8, 1)
    ix.map_batch_device(db.value, do.value, 50, int(offs[-1] - offs[0]), dout.value, 0)
    raw = np.zeros(50, dtype=mq.hit_dtype)
    hip.hipMemcpy(raw.ctypes.data, dout, 50 * 48, 2)
    assert (raw["status"] == 2).any()
    ok = raw["status"] != 2
    assert np.array_equal(raw["status"][ok] == 1, want["mapped"][ok] != 0)


def test_fast_path_taken_and_general_path_agrees(mq, oracle, simlib, ecoli, monkeypatch):
    """ACGT-only reads that fit one LDS tile go through the fast seeding path; MQ_FORCE_GENERAL=1 sends the same reads
    through the general streaming path.  Both must equal the oracle."""
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 300, seed=4)
    ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads, dict())
    _cmp_hits(hits, want)
    n_fast, n_gen = ix.last_map_path_counts()
    assert n_fast == 300 and n_gen == 0
    monkeypatch.setenv("MQ_FORCE_GENERAL", "1")
    ix2, ox2, hits2, want2 = _map_both(mq, oracle, g, off, names, reads, dict())
    _cmp_hits(hits2, want2)
    n_fast, n_gen = ix2.last_map_path_counts()
    assert n_fast == 0 and n_gen == 300
    assert np.array_equal(hits.view(np.uint8), hits2.view(np.uint8))


def test_long_and_mixed_reads_take_the_right_path(mq, oracle, simlib, ecoli):
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 40, seed=8, len_mean=30000, len_sd=15000, len_min=20, len_max=90000)
    bases = reads["bases"].copy()
    offs = reads["offsets"]
    for i in range(0, 40, 5):  # sprinkle N into every 5th read
        lo, hi = int(offs[i]), int(offs[i + 1])
        if hi - lo > 100:
            bases[lo + (hi - lo) // 2] = ord("N")
    reads2 = dict(reads)
    reads2["bases"] = bases
    ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads2, dict())
    _cmp_hits(hits, want)
    n_fast, n_gen = ix.last_map_path_counts()
    assert n_fast > 0 and n_gen > 0 and n_fast + n_gen == int(((offs[1:] - offs[:-1]) >= 35).sum())


def test_real_read_fixture(mq):
    """HIP k-min-mers of the reference's example reads vs the committed fixture (tests/golden/ecoli5_kminmers.json)."""
    import hashlib
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = json.load(open(os.path.join(gold, "ecoli5_kminmers.json")))
    recs, name = [], None
    for line in open(os.path.join(gold, "nearperfect-ecoli.5.fa")):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
        elif name is not None:
            recs.append((name, line.encode()))
            name = None
    bases = np.frombuffer(b"".join(s for _, s in recs), dtype=np.uint8)
    offs = np.zeros(len(recs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for _, s in recs])
    odt = np.dtype([("hash", "<u8"), ("start", "<u8"), ("end", "<u8"), ("offset", "<u8"), ("rev", "<i4"), ("_pad", "<i4")])
    for case in exp["cases"]:
        got = mq.Index(mq.Params(**case["params"])).kminmers_batch(bases, offs)
        for km, e in zip(got, case["reads"]):
            assert len(km) == e["n_kminmers"]
            o = np.zeros(len(km), dtype=odt)  # the oracle's record layout, which the digest was taken over
            for f in ("hash", "start", "end", "offset", "rev"):
                o[f] = km[f]
            assert hashlib.sha256(o.tobytes()).hexdigest() == e["sha256_of_tuples"]


def test_real_read_fixture_100(mq):
    """All 100 reads of the reference's example read file (tests/golden/nearperfect-ecoli.100.fa.gz) through the HIP seeder
    vs the committed per-read digests (tests/golden/ecoli100_kminmers.json), three parameter sets."""
    import gzip
    import hashlib
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = json.load(open(os.path.join(gold, "ecoli100_kminmers.json")))
    recs, name = [], None
    for line in gzip.open(os.path.join(gold, "nearperfect-ecoli.100.fa.gz"), "rt"):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
        elif name is not None:
            recs.append((name, line.encode()))
            name = None
    assert len(recs) == 100
    bases = np.frombuffer(b"".join(s for _, s in recs), dtype=np.uint8)
    offs = np.zeros(len(recs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for _, s in recs])
    odt = np.dtype([("hash", "<u8"), ("start", "<u8"), ("end", "<u8"), ("offset", "<u8"), ("rev", "<i4"), ("_pad", "<i4")])
    for case in exp["cases"]:
        got = mq.Index(mq.Params(**case["params"])).kminmers_batch(bases, offs)
        for km, e in zip(got, case["reads"]):
            assert len(km) == e["n_kminmers"], e["id"]
            o = np.zeros(len(km), dtype=odt)
            for f in ("hash", "start", "end", "offset", "rev"):
                o[f] = km[f]
            assert hashlib.sha256(o.tobytes()).hexdigest() == e["sha256_of_tuples"], e["id"]


def test_device_resident_entry_point_and_reuse(mq, oracle, simlib, ecoli):
    """mq_map_batch_device on caller-owned device buffers (plain hipMalloc through ctypes), launched twice on one index."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 500, seed=12)
    ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads, dict())
    bases, offs = reads["bases"], reads["offsets"]
    db, do, dout = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(db), bases.size) == 0 and hip.hipMalloc(C.byref(do), offs.size * 8) == 0
    assert hip.hipMalloc(C.byref(dout), 500 * 48) == 0
    assert hip.hipMemcpy(db, bases.ctypes.data, bases.size, 1) == 0 and hip.hipMemcpy(do, offs.ctypes.data, offs.size * 8, 1) == 0
    ml = int(offs[-1] - offs[0])  # total bases
    got = np.zeros(500, dtype=mq.hit_dtype)
    for _ in range(2):
        ix.map_batch_device(db.value, do.value, 500, ml, dout.value, 0)
        assert hip.hipMemcpy(got.ctypes.data, dout, 500 * 48, 2) == 0  # blocking D2H on the null stream orders after the kernel
        _cmp_hits(got, want)
    assert ix.last_map_ms() > 0
    for p in (db, do, dout):
        hip.hipFree(p)


def test_cli_end_to_end_paf_identical(mq, oracle, simlib, tmp_path, capsys):
    """`python -m mapquik_amd` surface: files in, `<prefix>.paf` out, byte-identical to the oracle's PAF; the reference's log lines."""
    from mapquik_amd import cli
    g, off, names = simlib.make_genome([500000, 300000], seed=41, repeat_frac=0.1)
    reads = simlib.make_reads(g, off, 150, seed=5, len_mean=12000, len_sd=4000)
    rn = simlib.read_names(reads, names)
    ref = tmp_path / "ref.fa"
    with open(ref, "wb") as w:
        for r in range(2):
            s = g[int(off[r]):int(off[r + 1])].tobytes()
            w.write(b">" + names[r].encode() + b" some description\n")
            for i in range(0, len(s), 70):  # multi-line reference, mixed case
                w.write((s[i:i + 70].lower() if (i // 70) % 2 else s[i:i + 70]) + b"\n")
    rd = tmp_path / "reads.fa"
    with open(rd, "wb") as w:
        for i, n in enumerate(rn):
            w.write(b">" + n.encode() + b"\n" + reads["bases"][int(reads["offsets"][i]):int(reads["offsets"][i + 1])].tobytes() + b"\n")
    prefix = str(tmp_path / "out")
    assert cli.main([str(rd), "--reference", str(ref), "-p", prefix, "--batch-bases", "400000"]) == 0
    po = oracle.params()
    ox = oracle.Index()
    for r in range(2):
        ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po)
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=2)
    want_txt = "".join(x + "\n" for x in oracle.paf_lines(ox, rn, want))
    assert open(prefix + ".paf").read() == want_txt and len(want_txt) > 1000
    out = capsys.readouterr().out
    assert "Indexed reference %s: " % names[0] in out and "Indexed %d unique k-min-mers in " % ox.count() in out
    assert "Mapped query sequences in " in out and "Total execution time: " in out and "Maximum RSS: " in out


def test_multi_tile_fast_path_long_reads(mq, oracle, simlib, ecoli):
    """Reads longer than one LDS tile stay on the fast path (tile by tile, carried codes / positions / minimizers)."""
    g, off, names = ecoli
    reads = simlib.make_reads(g, off, 24, seed=31, len_mean=120000, len_sd=80000, len_min=30000, len_max=400000)
    rng = np.random.default_rng(2)
    bases = reads["bases"].copy()
    offs = reads["offsets"]
    for i in (1, 5, 9):  # homopolymer runs longer than a tile, and one ending exactly at the end of a read
        lo, hi = int(offs[i]), int(offs[i + 1])
        a = lo + int(rng.integers(1000, 20000))
        bases[a:min(hi, a + 40000 * (1 + i % 2))] = ord("A")
    bases[int(offs[12]) - 5000:int(offs[12])] = ord("T")
    reads2 = dict(reads)
    reads2["bases"] = bases
    for ps in (dict(), dict(use_hpc=False), dict(k=3, l=12, density=0.05), dict(k=8, l=64, density=0.02)):
        ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads2, ps)
        _cmp_hits(hits, want)
        n_fast, n_gen = ix.last_map_path_counts()
        assert n_gen == 0 and n_fast == 24, (ps, n_fast, n_gen)
        got = ix.kminmers_batch(bases, offs)
        po = oracle.params(**ps)
        for i in range(0, 24, 5):
            _cmp_kmm(got[i], oracle.kminmers(bases[int(offs[i]):int(offs[i + 1])], po), (ps, i))


def test_scale_properties_permutation_and_idempotence(mq, oracle, simlib):
    """Larger case (312 Mbp, 25 contigs, 8192 reads): GPU == oracle, results do not depend on the order of reads in the batch
    (dynamic work distribution), and a second launch gives the same bytes."""
    lens = [max(40, int(x * 0.1)) for x in simlib.CHM13_LIKE]
    g, off, names = simlib.make_genome(lens, seed=77, threads=8, repeat_frac=0.05, tandem_frac=0.01)
    reads = simlib.make_reads(g, off, 8192, seed=78, threads=8)
    P, po = mq.Params(), oracle.params()
    ix, ox = mq.Index(P), oracle.Index()
    for r in range(len(lens)):
        ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])])
    ox.build_mt(g, off, names, po, threads=8)
    assert ix.finalize() == ox.count()
    bases, offs = reads["bases"], reads["offsets"]
    hits = ix.map_batch(bases, offs)
    want = ox.map_batch(bases, offs, po, threads=8)
    _cmp_hits(hits, want)
    assert np.array_equal(hits.view(np.uint8), ix.map_batch(bases, offs).view(np.uint8))
    perm = np.random.default_rng(0).permutation(8192)
    lens_r = (offs[1:] - offs[:-1]).astype(np.int64)
    poffs = np.zeros(8193, dtype=np.uint64)
    poffs[1:] = np.cumsum(lens_r[perm])
    pb = np.concatenate([bases[int(offs[i]):int(offs[i + 1])] for i in perm])
    ph = ix.map_batch(pb, poffs)
    assert np.array_equal(ph.view(np.uint8).reshape(8192, -1), hits.view(np.uint8).reshape(8192, -1)[perm])
    n_mapped, n_q60, n_wrong = simlib.mapeval(reads, want)
    assert n_q60 > 7800 and n_wrong <= 2


def test_index_save_load_roundtrip(mq, oracle, simlib, tmp_path):
    g, off, names = simlib.make_genome([400000, 100000], seed=9, repeat_frac=0.2)
    reads = simlib.make_reads(g, off, 200, seed=3, len_mean=9000)
    ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads, dict(k=4, l=20, density=0.02))
    p = str(tmp_path / "ix.mqx")
    ix.save(p)
    st = ix.stats()
    size = os.path.getsize(p)
    # the file holds the occupied slots only (32 B each) behind a small header, not the table
    assert 32 * st["n_keys"] < size < 32 * st["n_keys"] + 4096 and size < st["table_bytes"] // 4
    ix2 = mq.Index.load(p)
    assert ix2.stats() == ix.stats() and ix2.ref_info(1) == ix.ref_info(1)
    hits2 = ix2.map_batch(reads["bases"], reads["offsets"])
    assert np.array_equal(hits.view(np.uint8), hits2.view(np.uint8))
    rn = simlib.read_names(reads, names)
    assert ix2.paf_lines(rn, reads["offsets"], hits2) == oracle.paf_lines(ox, rn, want)
    blob = open(p, "rb").read()
    body = size - 32 * st["n_keys"]  # header + reference table

    def must_fail(data):
        q = str(tmp_path / "bad.mqx")
        with open(q, "wb") as f:
            f.write(data)
        with pytest.raises(mq.MapquikError):
            mq.Index.load(q)

    must_fail(b"XXXX" + blob[4:])                                   # not an index file
    must_fail(blob[:size - 32 * 5])                                  # truncated: fewer slots than the header says
    must_fail(blob + b"\0" * 32)                                     # bytes after the last slot
    bad = bytearray(blob)                                            # a slot naming a reference the file does not have
    bad[body + 12:body + 16] = (0x00FFFFFE).to_bytes(4, "little")    # id_rc of the first saved slot (Entry: start, end, offset, id_rc)
    must_fail(bytes(bad))
    dup = bytearray(blob)                                            # the same key twice: the table would hold fewer keys than the header says
    dup[body + 32:body + 64] = dup[body:body + 32]
    must_fail(bytes(dup))


def test_cpp_seam_find_matches_matches_oracle_paf(mq, oracle, simlib, tmp_path):
    """The C++ mirror of the reference's seam -- mers::ref_extract, Index::into_read_only, mers::find_matches (one read per call,
    src/mers.rs:77) and find_matches_batch (mapquik_host.hpp) -- driven the way src/closures.rs drives the Rust functions:
    the lines it returns are the oracle's PAF, for the defaults and for -k 7."""
    import subprocess
    from mapquik_amd import build
    exe = build.build_seam_test()
    g, off, names = simlib.make_genome([400000, 250000], seed=77, repeat_frac=0.1)
    reads = simlib.make_reads(g, off, 120, seed=12, len_mean=10000, len_sd=3000)
    rn = simlib.read_names(reads, names)
    ref, rd = tmp_path / "ref.fa", tmp_path / "reads.fa"
    with open(ref, "wb") as w:
        for r in range(2):
            w.write(b">" + names[r].encode() + b"\n" + g[int(off[r]):int(off[r + 1])].tobytes() + b"\n")
    with open(rd, "wb") as w:
        for i, n in enumerate(rn):
            w.write(b">" + n.encode() + b"\n" + reads["bases"][int(reads["offsets"][i]):int(reads["offsets"][i + 1])].tobytes() + b"\n")
    for ps, extra in ((dict(), []), (dict(k=7, l=31, density=0.01), ["7", "31", "0.01"])):
        po = oracle.params(**ps)
        ox = oracle.Index()
        for r in range(2):
            ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po)
        want = oracle.paf_lines(ox, rn, ox.map_batch(reads["bases"], reads["offsets"], po, threads=2))
        assert len(want) > 100
        for mode in ("single", "batch"):
            r = subprocess.run([exe, str(ref), str(rd), mode] + extra, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-300:]
            assert r.stdout.splitlines() == want, (mode, ps)
            assert ("unique %d" % ox.count()) in r.stderr


def test_native_driver_fails_loudly_mid_run(mq, simlib, tmp_path):
    """A failure in the middle of an input far larger than the feeder's chunk pool (here: a worker reports one on chunk 30 of
    ~70, test hook MQ_DRIVER_FAIL_AT) ends the run with the reference's panic code 101 and no PAF -- it used to hang: nobody
    recycled the chunks in flight, so the feeder's workers waited for buffers and the GPU workers for chunks, forever."""
    import subprocess
    from mapquik_amd import build
    exe = build.build_cli()
    g, off, names = simlib.make_genome([300000], seed=5)
    reads = simlib.make_reads(g, off, 800, seed=8, len_mean=9000)
    rn = simlib.read_names(reads, names)
    ref, rd = tmp_path / "ref.fa", tmp_path / "reads.fa"
    ref.write_bytes(b">" + names[0].encode() + b"\n" + g.tobytes() + b"\n")
    with open(rd, "wb") as w:
        for i, n in enumerate(rn):
            w.write(b">" + n.encode() + b"\n" + reads["bases"][int(reads["offsets"][i]):int(reads["offsets"][i + 1])].tobytes() + b"\n")
    for extra, env in (([], {}), (["--gpus", "2"], {"MQ_FAKE_MULTI": "1"})):
        prefix = str(tmp_path / "fail")
        r = subprocess.run([exe, str(rd), "--reference", str(ref), "-p", prefix, "--batch-bases", "100000", "--threads", "2"] + extra,
                           capture_output=True, text=True, timeout=120, env=dict(os.environ, MQ_DRIVER_FAIL_AT="30", **env))
        assert r.returncode == 101 and "injected failure" in r.stderr, (r.returncode, r.stderr[-300:])
        assert not os.path.exists(prefix + ".paf")
    ok = subprocess.run([exe, str(rd), "--reference", str(ref), "-p", str(tmp_path / "ok"), "--batch-bases", "100000", "--threads", "2"],
                        capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and os.path.getsize(str(tmp_path / "ok.paf")) > 0


def test_native_cli_end_to_end_paf_identical(mq, oracle, simlib, tmp_path):
    """The C++ `mapquik` driver: gzip'ed multi-line mixed-case reference, FASTQ reads, `<prefix>.paf` identical to the oracle."""
    import gzip
    import subprocess
    from mapquik_amd import build
    exe = build.build_cli()
    g, off, names = simlib.make_genome([500000, 300000], seed=43, repeat_frac=0.1)
    reads = simlib.make_reads(g, off, 150, seed=6, len_mean=12000, len_sd=4000)
    rn = simlib.read_names(reads, names)
    ref = tmp_path / "ref.fa.gz"
    with gzip.open(ref, "wb") as w:
        for r in range(2):
            s = g[int(off[r]):int(off[r + 1])].tobytes()
            w.write(b">" + names[r].encode() + b" desc\n")
            for i in range(0, len(s), 60):
                w.write((s[i:i + 60].lower() if (i // 60) % 3 == 0 else s[i:i + 60]) + b"\n")
    rd = tmp_path / "reads.fastq"
    with open(rd, "wb") as w:
        for i, n in enumerate(rn):
            s = reads["bases"][int(reads["offsets"][i]):int(reads["offsets"][i + 1])].tobytes()
            w.write(b"@" + n.encode() + b" x\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    prefix = str(tmp_path / "out")
    r = subprocess.run([exe, str(rd), "--reference", str(ref), "-p", prefix, "--batch-bases", "500000", "--unmapped"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # --gpus 3 on this 1-GPU box through the test hook: three workers with their own index replica, batches dealt round-robin,
    # output still in input order and byte-identical
    env = dict(os.environ, MQ_FAKE_MULTI="1")
    r3 = subprocess.run([exe, str(rd), "--reference", str(ref), "-p", prefix + "3", "--batch-bases", "150000", "--gpus", "3"],
                        capture_output=True, text=True, env=env)
    assert r3.returncode == 0, r3.stderr
    assert open(prefix + "3.paf").read() == open(prefix + ".paf").read()
    # the same reads as ONE gzip member, inflated by all feeder threads (par_gzip.hpp; forced at this size) and by one libdeflate call
    rdz = tmp_path / "reads.fastq.gz"
    rdz.write_bytes(gzip.compress(open(rd, "rb").read(), 6))
    for tag, envz in (("z", dict(os.environ, MQ_PARGZ_MIN="1000", MQ_PARGZ_SEG="60000", MQ_PARGZ_MINSEG="15000", MQ_FEEDER_TIMING="1")),
                      ("y", dict(os.environ, MQ_PARGZ="0", MQ_FEEDER_TIMING="1"))):
        rz = subprocess.run([exe, str(rdz), "--reference", str(ref), "-p", prefix + tag, "--batch-bases", "300000", "--threads", "4"],
                            capture_output=True, text=True, env=envz)
        assert rz.returncode == 0, rz.stderr
        assert ("all threads" in rz.stderr) == (tag == "z"), rz.stderr[-400:]
        assert open(prefix + tag + ".paf").read() == open(prefix + ".paf").read()
    r9 = subprocess.run([exe, str(rd), "--reference", str(ref), "-p", prefix + "9", "--gpus", "9"], capture_output=True, text=True)
    assert r9.returncode == 101 and "devices" in r9.stderr
    po = oracle.params()
    ox = oracle.Index()
    for k in range(2):
        ox.add_ref(k, names[k], g[int(off[k]):int(off[k + 1])], po)
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=2)
    want_txt = "".join(x + "\n" for x in oracle.paf_lines(ox, rn, want))
    assert open(prefix + ".paf").read() == want_txt and len(want_txt) > 1000
    unm = open(prefix + ".unmapped.out").read().split()
    assert unm == [n for n, w_ in zip(rn, want) if not w_["mapped"]]
    assert "Indexed %d unique k-min-mers in " % ox.count() in r.stdout and "Mapped query sequences in " in r.stdout
    assert "Maximum RSS: " in r.stdout


def test_fuzz_params_and_sequences(mq, oracle, simlib):
    """Random (k, l, density, hpc, c, s, g) x random genomes/reads incl. N runs, lowercase, low-complexity stretches."""
    rng = np.random.default_rng(int(os.environ.get("MQ_FUZZ_SEED", "20240")))  # MQ_FUZZ_ITERS / MQ_FUZZ_SEED: longer one-off campaigns
    for it in range(int(os.environ.get("MQ_FUZZ_ITERS", "24"))):
        k = int(rng.integers(1, 13))
        l = int(rng.choice([1, 2, 5, 8, 12, 15, 16, 17, 24, 31, 32, 33, 47, 63, 64]))
        dens = float(rng.choice([0.002, 0.01, 0.03, 0.1, 0.3]))
        ps = dict(k=k, l=l, density=dens, use_hpc=bool(rng.integers(0, 2)), c=int(rng.integers(0, 6)), s=int(rng.integers(0, 15)),
                  g=int(rng.choice([0, 50, 500, 2000, 100000])))
        lens = [int(x) for x in rng.integers(30, 120000, size=int(rng.integers(1, 5)))]
        g, off, names = simlib.make_genome(lens, seed=int(rng.integers(1, 1 << 30)), repeat_frac=float(rng.choice([0, 0.2, 0.6])),
                                           tandem_frac=float(rng.choice([0, 0.1])), threads=2)
        reads = simlib.make_reads(g, off, 40, seed=int(rng.integers(1, 1 << 30)), len_mean=float(rng.choice([300, 3000, 20000])),
                                  len_sd=2000, len_min=1, len_max=60000, err=float(rng.choice([0, 0.01, 0.05])), threads=2)
        if it % 2 == 1:  # damage the REFERENCE too (after the reads were drawn): runs of N, homopolymer and dinucleotide stretches, some of them
            g = g.copy()  # on the borders of the index build's 22,528-base segments: fast views, declined views and the general seeder must agree
            for _ in range(int(rng.integers(1, 6))):
                a = int(rng.integers(0, max(1, g.size - 10)))
                if rng.integers(0, 2):
                    a = min(g.size - 1, (a // 22528) * 22528 + int(rng.integers(-40, 40)) % 22528)
                n = int(rng.choice([1, 7, 40, 600, 3000, 9000]))
                kind = int(rng.integers(0, 3))
                seg = g[a:a + n]
                if kind == 0:
                    seg[:] = ord("N")
                elif kind == 1:
                    seg[:] = ord("ACGT"[int(rng.integers(0, 4))])
                else:
                    seg[:] = np.where(np.arange(seg.size) % 2 == 0, ord("T"), ord("G"))
        bases = reads["bases"].copy()
        offs = reads["offsets"]
        for i in range(0, 40, 7):  # damage some reads: N run, lowercase, homopolymer, dinucleotide repeat
            lo, hi = int(offs[i]), int(offs[i + 1])
            if hi - lo < 50:
                continue
            a = lo + int(rng.integers(0, hi - lo - 40))
            kind = (i // 7) % 4
            if kind == 0:
                bases[a:a + 30] = ord("N")
            elif kind == 1:
                bases[a:a + 20] |= 0x20
            elif kind == 2:
                bases[a:min(hi, a + 3000)] = ord("G")
            else:
                seg = bases[a:min(hi, a + 2000)]
                seg[:] = np.where(np.arange(seg.size) % 2 == 0, ord("A"), ord("C"))
        for i in range(3, 40, 7):  # reads from INSIDE a gap of the reference: N with an error every so often (substituted or inserted bases),
            lo, hi = int(offs[i]), int(offs[i + 1])  # whole or from some point on, N or n -- the general seeder's walk finds two run heads per error
            if hi - lo < 20:
                continue
            a = lo if (i // 7) % 2 == 0 else lo + int(rng.integers(0, hi - lo - 10))
            seg = bases[a:hi]
            seg[:] = ord("N") if (i // 7) % 3 else ord("n")
            nerr = int(rng.integers(0, max(2, seg.size // int(rng.choice([30, 140, 1000])))))
            at = rng.integers(0, seg.size, size=nerr)
            seg[at] = rng.choice(np.frombuffer(b"ACGTacgtNR", dtype=np.uint8), size=nerr)
        reads2 = dict(reads)
        reads2["bases"] = bases
        # every third case under another reading of the third-party k-min-mer iterator (mq_params.flags bits 8..13 = the oracle's
        # mqo_set_variant; bit 8 needs l >= 2): the variants' code paths see the same damaged inputs as the frozen reading's
        variant = 0
        if it % 3 == 2:
            variant = int(rng.choice([1, 2, 4, 8, 16, 32, 12, 24, 28, 63]))
        if os.environ.get("MQ_FUZZ_VARIANT"):  # a campaign on one variant's code paths (e.g. 4: stage B on two dwords)
            variant = int(os.environ["MQ_FUZZ_VARIANT"])
        if l < 2:
            variant &= ~8
        fast_kh = bool(os.environ.get("MQ_FUZZ_FAST_KH")) or it % 8 == 5  # MQ_FLAG_FAST_KH = the oracle's bit 64: every eighth case, or a whole campaign
        oracle.lib().mqo_set_variant(variant | (64 if fast_kh else 0))
        try:
            ix, ox, hits, want = _map_both(mq, oracle, g, off, names, reads2, ps, variant, fast_kh)
            _cmp_hits(hits, want)
            got = ix.kminmers_batch(bases, offs)
            po = oracle.params(**ps)
            for i in range(0, 40, 3):
                s = bases[int(offs[i]):int(offs[i + 1])]
                w = oracle.kminmers(s, po) if s.size >= po.l + po.k - 1 else np.zeros(0, dtype=oracle.kminmer_dtype)
                _cmp_kmm(got[i], w, (it, ps, variant, i))
        finally:
            oracle.lib().mqo_set_variant(0)


def test_no_device_memory_growth(mq, simlib):
    """Creating/freeing indexes and mapping repeatedly must not leak device memory."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    g, off, names = simlib.make_genome([300000], seed=3)
    reads = simlib.make_reads(g, off, 64, seed=4, len_mean=8000)

    def cycle():
        ix = mq.Index(mq.Params())
        ix.add_ref(0, names[0], g)
        ix.finalize()
        for _ in range(5):
            ix.map_batch(reads["bases"], reads["offsets"])
        ix.kminmers_batch(reads["bases"], reads["offsets"])
        ix.close()

    cycle()
    before = free_bytes()
    for _ in range(6):
        cycle()
    after = free_bytes()
    assert before - after < (64 << 20), (before, after)


def _index_eq_oracle(mq, oracle, seqs, ps, fold=False):
    """Index the sequences on the GPU and in the oracle: per-reference k-min-mer counts, key / live counts and the answer to every key."""
    P, po = mq.Params(**ps, **({"fold_case": True} if fold else {})), oracle.params(**ps)
    ix, ox = mq.Index(P), oracle.Index()
    all_h = []
    for r, s in enumerate(seqs):
        su = np.frombuffer(bytes(s).upper(), dtype=np.uint8) if fold else s
        assert ix.add_ref(r, "c%d" % r, s) == ox.add_ref(r, "c%d" % r, su, po), r
        if su.size >= po.l + po.k - 1:
            all_h.append(oracle.kminmers(su, po)["hash"])
    assert ix.finalize() == ox.count()
    st = ix.stats()
    assert st["n_keys"] == ox.keys() and st["n_unique"] == ox.count()
    hs = np.unique(np.concatenate(all_h)) if all_h else np.zeros(0, np.uint64)
    if hs.size > 60000:
        hs = hs[np.random.default_rng(3).choice(hs.size, 60000, replace=False)]
    found, ent, ids = ix.lookup(hs)
    for i, h in enumerate(hs):
        e = ox.get(int(h))
        assert bool(found[i]) == (e is not None)
        if e is not None:
            assert (int(ids[i]), int(ent[i]["start"]), int(ent[i]["end"]), int(ent[i]["offset"]), int(ent[i]["rev"])) == \
                   (int(e["id"]), int(e["start"]), int(e["end"]), int(e["offset"]), int(e["rc"]))
    ix.close()
    return st


@pytest.mark.parametrize("ps", [dict(), dict(k=3, l=12, density=0.05), dict(use_hpc=False, density=0.02)])
def test_reference_segments_borders(mq, oracle, simlib, monkeypatch, ps):
    """The index build seeds a reference in segments of 22,528 bases (views of two tiles with a 2,048-base halo, fast seeder) and sends
    what the fast seeder declines to the general one: runs of N, homopolymer runs and lower-case bases placed ON the segment borders,
    a homopolymer run longer than the halo (too few run heads behind the border: declined), contigs that end just before / at / just
    after a border, and the same references with every segment forced through the general seeder -- all equal to the oracle."""
    SEG = 22528
    rng = np.random.default_rng(5)
    g, off, _ = simlib.make_genome([5 * SEG + 777, 3 * SEG, 3 * SEG + 1, 3 * SEG - 1, SEG + 2048, 2 * SEG + 2047, 40, 31 + 4], seed=23, repeat_frac=0.1, tandem_frac=0.05)
    seqs = [g[int(off[r]):int(off[r + 1])].copy() for r in range(off.size - 1)]
    a = seqs[0]
    a[SEG - 3:SEG + 5] = ord("N")              # a run of N across the first border
    a[2 * SEG - 40:2 * SEG + 40] = ord("A")    # a homopolymer run across the second
    a[3 * SEG - 1] = a[3 * SEG]                # the border falls inside a run of two
    a[4 * SEG:4 * SEG + 3000] = ord("C")       # a run longer than the halo right behind a border: the view before it has too few run heads
    a[4 * SEG + 3000] = ord("G")
    b = seqs[1]
    b[SEG - 1] = ord("T")
    b[SEG] = ord("T")
    b[2 * SEG - 2000:2 * SEG - 1990] = ord("N")  # N inside a halo only
    st = _index_eq_oracle(mq, oracle, seqs, ps)
    assert st["n_keys"] > 1000
    # lower case with folding, on and around a border
    lower = [s.copy() for s in seqs[:3]]
    for s in lower:
        m = (s != ord("N"))
        idx = np.arange(s.size)
        sel = m & (((idx // 97) % 2) == 0)
        s[sel] = np.frombuffer(bytes(s[sel]).lower(), dtype=np.uint8)
    _index_eq_oracle(mq, oracle, lower, ps, fold=True)
    monkeypatch.setenv("MQ_FORCE_GENERAL", "1")
    st2 = _index_eq_oracle(mq, oracle, seqs, ps)
    assert st2 == st
    # list regions of 8 entries per segment: nearly every segment's list outgrows its region and is seeded again into its place
    monkeypatch.delenv("MQ_FORCE_GENERAL")
    monkeypatch.setenv("MQ_REF_CAP", "8")
    assert _index_eq_oracle(mq, oracle, seqs, ps) == st


def test_reserve_table_hint_is_only_a_hint(mq, oracle, simlib):
    """mq_index_reserve (DashMap::with_capacity, src/index.rs:83): the table is allocated in the background for the EXPECTED count;
    a hint far too small, far too large, exact, or none at all gives the same index, and an index freed before finalize leaves nothing behind."""
    g, off, names = simlib.make_genome([400000, 250000], seed=77, repeat_frac=0.1)
    po = oracle.params()
    ox = oracle.Index()
    for r in range(2):
        ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po)
    reads = simlib.make_reads(g, off, 300, seed=5)
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=4)
    stats = []
    for hint in (None, 1, 9000, 10**7):
        ix = mq.Index(mq.Params())
        if hint is not None:
            ix.reserve_table(hint)
            ix.reserve_table(hint * 2 + 5)  # a second call is ignored: one reservation per index
        for r in range(2):
            ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])])
        assert ix.finalize() == ox.count()
        stats.append(ix.stats())
        hits = ix.map_batch(reads["bases"], reads["offsets"])
        assert np.array_equal(hits["status"] == 1, want["mapped"] != 0)
        m = want["mapped"] != 0
        for a in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
            assert np.array_equal(mq.hit_column(hits, a)[m], want[a][m].astype(np.uint64)), (hint, a)
        ix.close()
    assert all(s == stats[0] for s in stats)
    ix = mq.Index(mq.Params())
    ix.reserve_table(5 * 10**6)
    ix.close()  # freed with the background allocation possibly still running
