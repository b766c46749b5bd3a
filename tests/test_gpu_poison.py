"""GPU: a lost or stale read cannot pass as 'unmapped'.  Every launch here writes into a result buffer filled with 0xFF first: a read
whose record no wave writes keeps status 0xFFFFFFFF (a cleared buffer would show status 0 = MQ_HIT_UNMAPPED, which is also what many
real reads are).  find_matches returns one result per read and has no cross-read state (src/mers.rs:77-102); map_kernel's work
distribution -- a wave's first two work items are its own, items w and n_waves + w, the atomic counter hands out the rest, the marked
entries of reads that went first are skipped -- is the part that can drop one (round 5's own-work-items change did: commit 8215505).

  * launches of n in {1, n_waves-1, n_waves, n_waves+1, 1.5 n_waves, 2 n_waves-1, 2 n_waves, 2 n_waves+1} reads (n_waves = the persistent waves of a
    full grid, read from the library) with reads that go first (tandem-array reads) placed so that their marked entries ARE the own
    items w and / or n_waves + w of many waves: all records written, every column the oracle's;
  * the bench's step -- 1,572,864 reads in ONE launch -- on a small genome (the read count is what matters): all records written,
    a strided 50 k sample and the first 8 k reads equal to the oracle, the instrumented launch byte-identical.
"""
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


def _ncpu():
    return max(1, len(os.sched_getaffinity(0)))


def _same(mq, got, want_diag):
    """status, every numeric PAF column and the k-min-mer count of every read against the oracle's (map_batch_diag)"""
    want, diag = want_diag
    assert np.array_equal(got["n_kminmers"].astype(np.uint64), diag["n_kminmers"].astype(np.uint64))
    m = want["mapped"] != 0
    assert np.array_equal(got["status"] == 1, m)
    for f in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
        assert np.array_equal(mq.hit_column(got, f)[m], want[f][m].astype(np.uint64)), f


def _launch_poisoned(mq, ix, bases, offs):
    """one launch of the device-resident entry point on a 0xFF-filled result buffer; returns the records as they are in device memory"""
    from hipmem import DevBuf, device_sync, memset
    n = offs.size - 1
    db, do, out = DevBuf.from_numpy(bases), DevBuf.from_numpy(offs), DevBuf(n * mq.hit_dtype.itemsize)
    memset(out, 0xFF)
    ix.reserve(n, int(offs[-1]))
    ix.map_batch_device(db.ptr, do.ptr, n, int(offs[-1]), out.ptr)
    device_sync()
    hits = out.to_numpy(mq.hit_dtype, n)
    for b in (db, do, out):
        b.free()
    return hits


def _assert_all_written(hits):
    bad = np.flatnonzero(hits["status"] > 2)
    assert bad.size == 0, "%d result records were never written (reads lost by the launch), first: %s" % (bad.size, bad[:10].tolist())


@pytest.fixture(scope="module")
def small_world(simlib, oracle):
    """a 600-kb genome with a 60-kb tandem array of period 5; pools of ordinary reads (1.5-3.5 kb) and of reads from inside the
    array (600-900 bases: order_reads_kernel puts them first)"""
    rng = np.random.default_rng(2026)
    g, off, names = simlib.make_genome([600_000], seed=31)
    g = g.copy()
    g[200_000:260_000] = np.tile(np.frombuffer(b"ACGGT", dtype=np.uint8), 12_000)
    plain = simlib.make_reads(g, off, 12000, seed=4, len_mean=2500, len_sd=300, len_min=1500, len_max=3500)
    o = plain["offsets"].astype(np.int64)
    away = (plain["end"] < 195_000) | (plain["start"] > 265_000)  # ordinary reads: none that touches the array (it would go first too)
    plain_seqs = [plain["bases"][o[i]:o[i + 1]].tobytes() for i in range(o.size - 1) if away[i]]
    assert len(plain_seqs) >= 8500
    per_seqs = []
    for _ in range(2000):
        s = 200_000 + int(rng.integers(0, 59_000))
        per_seqs.append(g[s:s + int(rng.integers(600, 900))].tobytes())
    po = oracle.params()
    ox = oracle.Index()
    ox.add_ref(0, names[0], g, po)
    return g, off, names, plain_seqs, per_seqs, ox, po


def _batch(plain_seqs, per_seqs, n, first_positions):
    """n reads; those at `first_positions` are tandem-array reads"""
    fp = set(int(x) for x in first_positions)
    seqs, ip, iq = [], 0, 0
    for r in range(n):
        if r in fp:
            seqs.append(per_seqs[iq % len(per_seqs)])
            iq += 1
        else:
            seqs.append(plain_seqs[ip % len(plain_seqs)])
            ip += 1
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return bases, offs


def _positions_for_items(items, n):
    """Read numbers whose MARKED natural entries sit at the given work items: item i holds read i - nf once nf reads go first, and nf is
    the number of such reads itself -- the largest consistent choice."""
    items = sorted(set(int(i) for i in items))
    for nf in range(len(items), 0, -1):
        pos = [i - nf for i in items if nf <= i < n + nf]
        if len(pos) >= nf:
            return pos[:nf]
    return []


@pytest.mark.parametrize("shape", ["own1", "own2", "both", "scattered"])
def test_wave_count_edges_on_poisoned_output(mq, oracle, simlib, small_world, shape):
    from test_gpu_launch_order import _flagged
    g, off, names, plain_seqs, per_seqs, ox, po = small_world
    ix = mq.Index(mq.Params())
    ix.add_ref(0, names[0], g)
    ix.finalize()
    nw_full = ix.launch_waves(1 << 20)
    assert nw_full >= 64 and nw_full % 8 == 0
    rng = np.random.default_rng(7)
    # (n_waves + n_waves / 2: between one and two items per wave, the counter empty from the start -- where round 5's bug lost one read per wave
    # whose first own item was a marked entry)
    for n in (1, nw_full - 1, nw_full, nw_full + 1, nw_full + nw_full // 2, 2 * nw_full - 1, 2 * nw_full, 2 * nw_full + 1):
        nw = ix.launch_waves(n)
        assert nw == min(nw_full, 8 * ((n + 7) // 8))
        waves = [int(w) for w in rng.choice(nw, size=min(nw, 96), replace=False)]
        if shape == "own1":
            items = waves
        elif shape == "own2":
            items = [nw + w for w in waves]
        elif shape == "both":
            items = waves + [nw + w for w in waves]
        else:  # anywhere, one read in twelve
            items = [int(i) for i in rng.choice(max(n, 1), size=max(1, n // 12), replace=False)]
        pos = _positions_for_items(items, n)
        bases, offs = _batch(plain_seqs, per_seqs, n, pos)
        flags = _flagged(bases, offs)
        assert flags.sum() == len(pos) and all(flags[p] for p in pos)  # exactly the planted reads go first: the items are where they were aimed
        hits = _launch_poisoned(mq, ix, bases, offs)
        assert ix.last_map_order() == (len(pos), len(pos))
        _assert_all_written(hits)
        _same(mq, hits, ox.map_batch_diag(bases, offs, po, threads=_ncpu()))
        assert n < 64 or (hits["status"] == 1).mean() > 0.5
    ix.close()


def test_bench_step_size_launch_on_poisoned_output(mq, oracle, simlib):
    """1,572,864 reads (37 Gbases) in one launch, as bench.py steps, synthesised slice by slice into device memory (tools/sim.py
    read_slices: the reads of make_reads, never all in host memory)."""
    from hipmem import DevBuf, device_sync, memset, hip
    T = _ncpu()
    n = 1572864
    lens = [max(40, int(x * 0.02)) for x in simlib.CHM13_LIKE]
    g, off, names = simlib.make_genome(lens, seed=2013, threads=T, repeat_frac=0.05, tandem_frac=0.01)
    P, po = mq.Params(), oracle.params()
    ix, ox = mq.Index(P), oracle.Index()
    for r in range(len(names)):
        ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])])
    ox.build_mt(g, off, names, po, T)
    assert ix.finalize() == ox.count()
    caps = np.zeros(n + 1, dtype=np.uint64)
    co = np.ascontiguousarray(off, dtype=np.uint64)
    simlib.lib().mqsim_read_caps(co.ctypes.data, co.size - 1, n, 24000.0, 2300.0, 100, 25000, 3013, caps.ctypes.data)
    db = DevBuf(int(caps[-1]) + 64)
    del caps
    offs = np.zeros(n + 1, dtype=np.uint64)
    stride, n_head = 31, 8192
    s_seqs, s_idx, head = [], [], []
    at = 0
    for r0, r1, b, o, t in simlib.read_slices(g, off, n, seed=3013, slice_reads=32768, threads=T):
        assert hip().hipMemcpy(db.ptr + at, b.ctypes.data, b.size, 1) == 0
        offs[r0 + 1:r1 + 1] = o[1:] + np.uint64(at)
        if r0 < n_head:
            head.append(b[:int(o[min(r1, n_head) - r0])].copy())
        for r in range(-(-r0 // stride) * stride, r1, stride):
            s_seqs.append(b[int(o[r - r0]):int(o[r - r0 + 1])].copy())
            s_idx.append(r)
        at += b.size
    assert at == int(offs[-1]) and at > 36 * 10**9
    do, out, out2 = DevBuf.from_numpy(offs), DevBuf(n * mq.hit_dtype.itemsize), DevBuf(n * mq.hit_dtype.itemsize)
    memset(out, 0xFF)
    memset(out2, 0xFF)
    ix.reserve(n, at)
    ix.map_batch_device(db.ptr, do.ptr, n, at, out.ptr)
    device_sync()
    hits = out.to_numpy(mq.hit_dtype, n)
    _assert_all_written(hits)
    assert (hits["status"] == 2).sum() == 0
    # the same batch through the instrumented launch (other kernel instantiation): byte-identical records
    ix.probe_stats(db.ptr, do.ptr, n, at, out2.ptr)
    device_sync()
    assert out2.to_numpy(np.uint8, n * mq.hit_dtype.itemsize).tobytes() == hits.tobytes()
    # the oracle on the first 8,192 reads and on every 31st read of the launch
    hb = np.concatenate(head)
    _same(mq, hits[:n_head], ox.map_batch_diag(hb, offs[:n_head + 1], po, threads=T))
    s_idx = np.asarray(s_idx)
    assert s_idx.size >= 50000
    so = np.zeros(s_idx.size + 1, dtype=np.uint64)
    so[1:] = np.cumsum([x.size for x in s_seqs])
    _same(mq, hits[s_idx], ox.map_batch_diag(np.concatenate(s_seqs), so, po, threads=T))
    assert (hits["status"] == 1).mean() > 0.9
    for b in (db, do, out, out2):
        b.free()
    ix.close()
