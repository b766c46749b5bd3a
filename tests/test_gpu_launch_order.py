"""GPU: the launch order of map_kernel (order_reads_kernel, mq_map_kernels.hpp).  Reads that look like short-period tandem arrays are
taken up first -- they list thousands of minimizers and would otherwise be a launch's tail -- and nothing but the order may change:
hits identical with the order on and off (MQ_HEAVY_FIRST=0) and identical to the oracle; the periodicity test pinned by a numpy
restatement; more flagged reads than the front of the order holds; reads through the spans form (device-parsed FASTA).
Reference semantics untouched: find_matches has no cross-read state (src/mers.rs:77-102)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mq():
    import mapquik_amd
    if mapquik_amd.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return mapquik_amd


def _periodic(win):
    """window_periodic of mq_map_kernels.hpp: 48 bases, some lag 1..16 matches in >= 27 of the first 32 positions (compared as the
    2-bit codes (c >> 1) & 3, which are distinct for A C G T)"""
    c = (np.asarray(win, dtype=np.uint8) >> 1) & 3
    return any(int((c[:32] == c[lag:lag + 32]).sum()) >= 27 for lag in range(1, 17))


def _flagged(bases, offs):
    out = []
    for i in range(offs.size - 1):
        o, ln = int(offs[i]), int(offs[i + 1] - offs[i])
        if ln < 512:
            out.append(False)
            continue
        out.append(any(_periodic(bases[w:w + 48]) for w in (((o + at) // 64) * 64 for at in (ln // 6, ln // 2, (ln // 6) * 5))))  # windows start at multiples of 64 bytes of the batch
    return np.array(out)


def _world(simlib, oracle, seed=77):
    """a 3-Mbp genome with tandem arrays of period 2..14 (30-60 kb each, 0.5 % divergence between copies) and reads from everywhere"""
    rng = np.random.default_rng(seed)
    g, off, names = simlib.make_genome([2_000_000, 1_000_000], seed=seed)
    g = g.copy()
    arrays = []
    for a in range(12):
        period = int(rng.integers(2, 15))
        ln = int(rng.integers(30_000, 60_000))
        at = int(rng.integers(0, 2_900_000 - ln))
        unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=period)
        arr = np.tile(unit, ln // period + 1)[:ln].copy()
        mut = rng.random(ln) < 0.005
        arr[mut] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(mut.sum()))
        g[at:at + ln] = arr
        arrays.append((at, ln))
    reads = simlib.make_reads(g, off, 3000, seed=seed + 1, len_mean=12000, len_sd=3000, len_min=100, len_max=25000)
    # plus reads cut from inside the arrays, so that whole reads are periodic
    extra = []
    for at, ln in arrays:
        for j in range(6):
            s = at + int(rng.integers(0, ln - 20_000))
            extra.append(g[s:s + int(rng.integers(8_000, 20_000))].tobytes())
    o = reads["offsets"].astype(np.int64)
    seqs = [reads["bases"][o[i]:o[i + 1]].tobytes() for i in range(o.size - 1)]
    order = rng.permutation(len(seqs) + len(extra))
    allseq = seqs + extra
    seqs = [allseq[i] for i in order]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    po = oracle.params()
    ox = oracle.Index()
    for r in range(len(names)):
        ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po)
    want = ox.map_batch(bases, offs, po, threads=8)
    return g, off, names, bases, offs, want


def _index(mq, g, off, names, **kw):
    ix = mq.Index(mq.Params(**kw))
    for r in range(len(names)):
        ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])])
    ix.finalize()
    return ix


def _same(got, want):
    """HIP hits (mq_hit) against the oracle's records: mapped or not, and every column of the mapped ones"""
    import mapquik_amd
    m = want["mapped"] != 0
    assert np.array_equal(got["status"] == 1, m)
    for f in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
        assert np.array_equal(mapquik_amd.hit_column(got, f)[m], want[f][m].astype(np.uint64)), f


def test_tandem_reads_go_first_and_nothing_else_changes(mq, oracle, simlib, monkeypatch):
    g, off, names, bases, offs, want = _world(simlib, oracle)
    flags = _flagged(bases, offs)
    assert 40 <= flags.sum() <= flags.size // 4  # the planted array reads, not everything
    ix = _index(mq, g, off, names)
    got = ix.map_batch(bases, offs)
    n_flagged, n_first = ix.last_map_order()
    assert (n_flagged, n_first) == (int(flags.sum()), int(flags.sum()))  # the device's test is the restatement above, read for read
    _same(got, want)
    # minimizer lists of such reads are what the order is for: the densest reads are among the flagged
    dense = np.argsort(-got["n_kminmers"].astype(np.int64))[:10]
    assert got["n_kminmers"][dense[0]] > 5 * np.median(got["n_kminmers"]) and flags[dense].sum() >= 8
    ix.close()
    monkeypatch.setenv("MQ_HEAVY_FIRST", "0")  # the A/B hook: every read in its own place
    ix0 = _index(mq, g, off, names)
    got0 = ix0.map_batch(bases, offs)
    assert ix0.last_map_order() == (0, 0)
    assert got0.tobytes() == got.tobytes()
    ix0.close()


def test_order_through_the_other_launch_forms(mq, oracle, simlib):
    """the same batch through a context's device-resident entry point, as device-parsed FASTA (the spans form: per-read lengths, headers
    between the reads) and with a seeding variant (the other kernel instantiation)"""
    from hipmem import DevBuf, device_sync
    g, off, names, bases, offs, want = _world(simlib, oracle, seed=91)
    ix = _index(mq, g, off, names)
    n = offs.size - 1
    db, do, out = DevBuf.from_numpy(bases), DevBuf.from_numpy(offs), DevBuf(n * mq.hit_dtype.itemsize)
    ix.reserve(n, int(offs[-1]))
    ix.map_batch_device(db.ptr, do.ptr, n, int(offs[-1]), out.ptr)
    device_sync()
    _same(out.to_numpy(mq.hit_dtype, n), want)
    assert ix.last_map_order()[0] == int(_flagged(bases, offs).sum())
    ctx = ix.context()
    fa = b"".join(b">r%d some text\n" % i + bases[int(offs[i]):int(offs[i + 1])].tobytes() + b"\n" for i in range(n))
    ctx.submit_fasta(np.frombuffer(fa, dtype=np.uint8))
    hits, lines, flags = ctx.wait_fasta()
    assert flags == 0 and len(hits) == n
    _same(hits, want)
    ctx.close()
    ix.close()
    oracle.lib().mqo_set_variant(1)
    try:
        po = oracle.params()
        ox = oracle.Index()
        for r in range(len(names)):
            ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po)
        want1 = ox.map_batch(bases, offs, po, threads=8)
    finally:
        oracle.lib().mqo_set_variant(0)
    ix1 = _index(mq, g, off, names, seeding_variant=1)
    got1 = ix1.map_batch(bases, offs)
    assert ix1.last_map_order()[0] == int(_flagged(bases, offs).sum())
    _same(got1, want1)
    ix1.close()


@pytest.mark.parametrize("n_periodic,n_plain", [(40_000, 500), (1_500, 4_000), (300, 7_900)])
def test_more_flagged_reads_than_the_front_holds(mq, oracle, simlib, n_periodic, n_plain):
    """40,000 periodic reads of 600-900 bases and 500 ordinary ones: 32,768 go first, the others stay where they are, every read is
    mapped exactly once (results by read number, identical to the oracle); reads shorter than 512 bases are never tested.
    5,500 and 8,200 reads: between one and two work items per wave of a full grid (4,096 waves) -- a wave's first two items are its own,
    and a wave whose first is the marked entry of a read that went first must still take its second (a launch of 6,000 reads once lost two)."""
    rng = np.random.default_rng(5 + n_plain)
    g, off, names = simlib.make_genome([400_000], seed=12)
    g = g.copy()
    unit = np.frombuffer(b"ACGGT", dtype=np.uint8)
    g[100_000:150_000] = np.tile(unit, 10_000)
    reads = simlib.make_reads(g, off, n_plain, seed=3, len_mean=5000 if n_plain <= 500 else 900, len_sd=1000 if n_plain <= 500 else 200, len_min=100, len_max=8000)
    o = reads["offsets"].astype(np.int64)
    seqs = [reads["bases"][o[i]:o[i + 1]].tobytes() for i in range(o.size - 1)]
    per = []
    for i in range(n_periodic):
        s = 100_000 + int(rng.integers(0, 49_000))
        per.append(g[s:s + int(rng.integers(600, 900))].tobytes())
    short = [g[s:s + 511].tobytes() for s in range(100_000, 100_050)]
    order = rng.permutation(len(seqs) + len(per) + len(short))
    allseq = seqs + per + short
    seqs = [allseq[i] for i in order]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    po = oracle.params()
    ox = oracle.Index()
    ox.add_ref(0, names[0], g, po)
    want = ox.map_batch(bases, offs, po, threads=8)
    ix = _index(mq, g, off, names)
    got = ix.map_batch(bases, offs)
    n_flagged, n_first = ix.last_map_order()
    assert n_flagged == int(_flagged(bases, offs).sum()) and n_flagged >= n_periodic and n_first == min(n_flagged, 32768)
    _same(got, want)
    ix.close()
