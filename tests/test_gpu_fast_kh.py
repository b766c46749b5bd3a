"""GPU: MQ_FLAG_FAST_KH, the opt-in cheap k-min-mer tuple hash (include/mapquik_hip.h).  The reference's index compares k-min-mers by their
tuple hash only (Index::add_with_mer / ReadOnlyIndex::get key on KminmerHash.hash, src/index.rs:100-104,118-126; Match::check looks at
entries, never at the hash, src/match.rs:39-58), so any hash that separates the tuples gives the same PAF.  With the flag: k-min-mer tuples
identical to the oracle under variant bit 64 (mqo_tuple_hash_fast) -- positions, offsets, strands equal to the SipHash run's, hashes not --,
index counts equal to the SipHash run's, hits BYTE-identical to the SipHash run's and PAF bytes the oracle's; through the general seeder,
with seeding variants, k = 5 / 7 / 8 / other, save / load / clone, and the native driver's --fast-kh."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _index(mq, P, g, off, names):
    ix = mq.Index(P)
    per = [ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])]) for r in range(len(names))]
    return ix, per, ix.finalize()


@pytest.mark.parametrize("ps,v", [(dict(), 0), (dict(k=7), 0), (dict(k=8, l=16, g=100), 0), (dict(k=3, l=12, density=0.05), 0), (dict(k=12, l=24, density=0.1), 0),
                                  (dict(), 4), (dict(use_hpc=False), 6), (dict(k=7), 24)])
def test_tuples_index_and_hits(mq, O, simlib, ps, v, monkeypatch):
    g, off, names = simlib.make_genome([700_000, 400_000], seed=17, repeat_frac=0.3, tandem_frac=0.05)
    reads = simlib.make_reads(g, off, 300, seed=2, len_mean=9000, len_sd=4000, len_min=50, len_max=25000)
    bases, offs = reads["bases"].copy(), reads["offsets"]
    bases[int(offs[5]) + 100:int(offs[5]) + 400] = ord("N")  # a read for map_declined_kernel
    P0, P1, po = mq.Params(seeding_variant=v, **ps), mq.Params(seeding_variant=v, fast_kh=True, **ps), O.params(**ps)
    assert P1.fast_kh and not P0.fast_kh and P1.seeding_variant == v
    ix0, per0, u0 = _index(mq, P0, g, off, names)
    ix1, per1, u1 = _index(mq, P1, g, off, names)
    assert per0 == per1 and u0 == u1 and ix0.stats()["n_keys"] == ix1.stats()["n_keys"]  # the same tuples are equal under either hash
    h0, h1 = ix0.map_batch(bases, offs), ix1.map_batch(bases, offs)
    assert h0.tobytes() == h1.tobytes() and (h0["status"] == 1).sum() > 200
    O.lib().mqo_set_variant(v | 64)
    ox = O.Index()
    for r in range(len(names)):
        ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po)
    assert ox.count() == u1
    want = ox.map_batch(bases, offs, po, threads=4)
    rn = simlib.read_names(reads, names)
    assert ix1.paf_lines(rn, offs, h1) == O.paf_lines(ox, rn, want)
    k0, k1 = ix0.kminmers_batch(bases, offs), ix1.kminmers_batch(bases, offs)
    n_differ = 0
    for i in range(0, 300, 7):
        s = bases[int(offs[i]):int(offs[i + 1])]
        w = O.kminmers(s, po) if s.size >= po.l + po.k - 1 else np.zeros(0, dtype=O.kminmer_dtype)
        assert len(k1[i]) == len(w) == len(k0[i])
        for f in ("hash", "start", "end", "offset", "rev"):
            assert np.array_equal(k1[i][f].astype(np.uint64), w[f].astype(np.uint64)), (i, f)
        for f in ("start", "end", "offset", "rev"):
            assert np.array_equal(k1[i][f], k0[i][f])
        n_differ += int((k1[i]["hash"] != k0[i]["hash"]).sum())
        # the two hashes induce the same partition of the read's tuples
        _, inv0 = np.unique(k0[i]["hash"], return_inverse=True)
        _, inv1 = np.unique(k1[i]["hash"], return_inverse=True)
        assert np.array_equal(inv0 == inv0[:, None], inv1 == inv1[:, None]) if len(inv0) < 600 else True
    assert n_differ > 1000
    # the general seeder for everything: the same records once more
    monkeypatch.setenv("MQ_FORCE_GENERAL", "1")
    ix2, per2, u2 = _index(mq, P1, g, off, names)
    assert per2 == per1 and u2 == u1 and ix2.map_batch(bases, offs).tobytes() == h1.tobytes()
    for x in (ix0, ix1, ix2):
        x.close()


def test_flag_survives_save_load_and_clone_and_is_a_seeding_parameter(mq, O, simlib, tmp_path):
    g, off, names = simlib.make_genome([300_000, 200_000], seed=5, repeat_frac=0.1)
    reads = simlib.make_reads(g, off, 200, seed=8)
    ix, _, _ = _index(mq, mq.Params(fast_kh=True), g, off, names)
    hits = ix.map_batch(reads["bases"], reads["offsets"])
    p = str(tmp_path / "f.mqx")
    ix.save(p)
    ix2 = mq.Index.load(p)
    assert ix2.get_params().fast_kh and ix2.map_batch(reads["bases"], reads["offsets"]).tobytes() == hits.tobytes()
    rep = ix.clone(0)
    assert rep.map_batch(reads["bases"], reads["offsets"]).tobytes() == hits.tobytes()
    for x in (ix, ix2, rep):
        x.close()
    # both drivers refuse to map with another tuple hash than the index was built with
    import subprocess
    import sys
    from mapquik_amd import build as B
    B.build_cli()
    rd = tmp_path / "reads.fa"
    offs = reads["offsets"]
    with open(rd, "wb") as f:
        for i in range(50):
            f.write(b">r%d\n" % i + reads["bases"][int(offs[i]):int(offs[i + 1])].tobytes() + b"\n")
    r1 = subprocess.run([B.CLI, str(rd), "--index", p, "-p", str(tmp_path / "o1")], capture_output=True, text=True, timeout=300)
    assert r1.returncode != 0 and "--fast-kh" in (r1.stdout + r1.stderr)
    r2 = subprocess.run([B.CLI, str(rd), "--index", p, "-p", str(tmp_path / "o2"), "--fast-kh"], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-1500:]
    r3 = subprocess.run([sys.executable, "-m", "mapquik_amd", str(rd), "--index", p, "-p", str(tmp_path / "o3")], capture_output=True, text=True, timeout=300)
    assert r3.returncode != 0 and "--fast-kh" in (r3.stdout + r3.stderr)
    r4 = subprocess.run([sys.executable, "-m", "mapquik_amd", str(rd), "--index", p, "-p", str(tmp_path / "o4"), "--fast-kh"], capture_output=True, text=True, timeout=300)
    assert r4.returncode == 0, r4.stderr[-1500:]
    assert open(str(tmp_path / "o2.paf")).read() == open(str(tmp_path / "o4.paf")).read() != ""


def test_native_driver_fast_kh_gives_the_same_paf(mq, O, simlib, tmp_path):
    import subprocess
    from mapquik_amd import build as B
    B.build_cli()
    g, off, names = simlib.make_genome([400_000, 250_000], seed=11, repeat_frac=0.05)
    reads = simlib.make_reads(g, off, 300, seed=4)
    rn = simlib.read_names(reads, names)
    offs = reads["offsets"]
    ref = tmp_path / "ref.fa"
    with open(ref, "wb") as f:
        for r in range(2):
            f.write(b">" + names[r].encode() + b"\n" + g[int(off[r]):int(off[r + 1])].tobytes() + b"\n")
    rd = tmp_path / "reads.fq"
    with open(rd, "wb") as f:
        for i, n in enumerate(rn):
            s = reads["bases"][int(offs[i]):int(offs[i + 1])].tobytes()
            f.write(b"@" + n.encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    out = {}
    for tag, extra in (("sip", []), ("fast", ["--fast-kh"])):
        prefix = str(tmp_path / tag)
        r_ = subprocess.run([B.CLI, str(rd), "--reference", str(ref), "-p", prefix, "--threads", "2", "-k", "7"] + extra, capture_output=True, text=True, timeout=600)
        assert r_.returncode == 0, r_.stderr[-2000:]
        assert ("Fast k-min-mer tuple hash" in r_.stdout) == bool(extra)
        out[tag] = open(prefix + ".paf").read()
    po = O.params(k=7)
    ox = O.Index()
    for r in range(2):
        ox.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])], po)
    want = "".join(ln + "\n" for ln in O.paf_lines(ox, rn, ox.map_batch(reads["bases"], offs, po, threads=4)))
    assert out["sip"] == out["fast"] == want and len(want) > 1000
