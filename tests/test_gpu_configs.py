"""GPU parity at the sizes BASELINE.json's configs name (synthetic stand-ins: the real genomes are not in the image).

  config 2  E. coli full pbsim set: 4,641,652-bp genome, depth 10 (~1.9 k reads), defaults and example/run_ecoli.sh:26 flags
  config 3  CHM13v2.0-like, scale 1.0 (3.117 Gbp, 25 contigs): oracle identity on 50 k reads, order independence and
            idempotence on a 196,608-read batch (rounds 1-4's bench step; the bench's step since round 5 -- 1,572,864 reads in one
            launch -- is tests/test_gpu_poison.py::test_bench_step_size_launch_on_poisoned_output)
  config 5  maize-like: 2.13 Gbp, 10 contigs <= 308 Mbp, >= 80 % of the bases in transposon-like families (copies 1-5 %
            apart), runs of N in the reference, depth-30-shaped reads (experiments/simulate_maize.sh:1-12)
  config 4  its SHAPE (experiments/table1.sh:50-55: uncompressed FASTQ reads, -k 7 -l 31 -d 0.01, human-scale reference) through
            the native driver: CHM13v2.0-like scale 1.0, 50 k reads as a FASTQ file, <prefix>.paf byte-identical to the oracle's.
            The real DeepConsensus HG002 reads are not in the image, and there is no 8-GPU node to run on.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mq():
    import mapquik_amd
    if mapquik_amd.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    return mapquik_amd


def _cmp(hits, want):
    assert np.array_equal(hits["status"] == 1, want["mapped"] != 0)
    m = want["mapped"] != 0
    import mapquik_amd
    for a in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
        assert np.array_equal(mapquik_amd.hit_column(hits, a)[m], want[a][m].astype(np.uint64)), a


def _ncpu():
    import os
    return max(1, len(os.sched_getaffinity(0)))


@pytest.mark.parametrize("ps", [dict(), dict(k=8, l=16, g=100)])
def test_config2_ecoli_depth10_paf_identical(mq, oracle, simlib, ps):
    g, off, names = simlib.make_genome(simlib.ECOLI_LEN, seed=913, prefix="chr000913_")
    n = int(10 * simlib.ECOLI_LEN[0] / 24000)  # depth 10 at mean length 24 kb (example/simulate_pbsim.sh:7-14)
    reads = simlib.make_reads(g, off, n, seed=1, len_mean=24000, len_sd=2300, len_min=100, len_max=25000)
    assert 1900 <= n <= 1960 and abs(int(reads["offsets"][-1]) / simlib.ECOLI_LEN[0] - 10) < 0.5
    P, po = mq.Params(**ps), oracle.params(**ps)
    ix, ox = mq.Index(P), oracle.Index()
    assert ix.add_ref(0, names[0], g) == ox.add_ref(0, names[0], g, po)
    assert ix.finalize() == ox.count()
    hits = ix.map_batch(reads["bases"], reads["offsets"])
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=_ncpu())
    _cmp(hits, want)
    rn = simlib.read_names(reads, names)
    got_txt = "".join(x + "\n" for x in ix.paf_lines(rn, reads["offsets"], hits))
    want_txt = "".join(x + "\n" for x in oracle.paf_lines(ox, rn, want))
    assert got_txt.encode() == want_txt.encode() and len(want_txt) > 100000
    n_m, n_q60, n_wrong = simlib.mapeval(reads, want)
    assert n_q60 >= 0.97 * n and n_wrong == 0


def test_config3_chm13_like_full_scale(mq, oracle, simlib):
    from hipmem import DevBuf, device_sync
    T = _ncpu()
    lens = list(simlib.CHM13_LIKE)
    g, off, names = simlib.make_genome(lens, seed=2013, threads=T, repeat_frac=0.05, tandem_frac=0.01)
    assert g.size > 3_100_000_000
    P, po = mq.Params(), oracle.params()
    ix, ox = mq.Index(P), oracle.Index()
    for r in range(len(lens)):
        seg = DevBuf.from_numpy(g[int(off[r]):int(off[r + 1])])
        ix.add_ref_device(r, names[r], seg.ptr, seg.nbytes)
        seg.free()
    ox.build_mt(g, off, names, po, T)
    assert ix.finalize() == ox.count()
    n = 196608
    reads = simlib.make_reads(g, off, n, seed=3013, threads=T)
    bases, offs = reads["bases"], reads["offsets"]
    ns = 50000
    sb, so = bases[:int(offs[ns])], offs[:ns + 1]
    want, diag = ox.map_batch_diag(sb, so, po, threads=T)
    # the full batch through the device-resident entry point (what bench.py times)
    d_b, d_o, d_h = DevBuf.from_numpy(bases), DevBuf.from_numpy(offs), DevBuf(n * 48)
    ml = int(offs[-1] - offs[0])  # total bases of the batch
    ix.map_batch_device(d_b.ptr, d_o.ptr, n, ml, d_h.ptr, 0)
    hits = d_h.to_numpy(mq.hit_dtype, n)
    assert (hits["status"] == 2).sum() == 0
    _cmp(hits[:ns], want)
    n_m, n_q60, n_wrong = simlib.mapeval({k: v[:ns] for k, v in reads.items() if k not in ("bases", "offsets")}, want)
    assert n_q60 > 0.97 * ns and n_wrong <= 5
    assert (diag["multi_match_refs"] > 0).sum() > 0.9 * ns  # 1 % error: a read's hits break into several Matches
    # idempotence: a second launch gives the same bytes
    ix.map_batch_device(d_b.ptr, d_o.ptr, n, ml, d_h.ptr, 0)
    assert np.array_equal(d_h.to_numpy(np.uint8, n * 48), hits.view(np.uint8))
    # order independence on the full batch: reversed read order (dynamic work distribution, other wave / tile phases)
    perm = np.arange(n)[::-1].copy()
    lens_r = (offs[1:] - offs[:-1]).astype(np.int64)
    poffs = np.zeros(n + 1, dtype=np.uint64)
    poffs[1:] = np.cumsum(lens_r[perm])
    pb = np.concatenate([bases[int(offs[i]):int(offs[i + 1])] for i in perm])
    d_b.free()
    d_pb, d_po = DevBuf.from_numpy(pb), DevBuf.from_numpy(poffs)
    ix.map_batch_device(d_pb.ptr, d_po.ptr, n, ml, d_h.ptr, 0)
    ph = d_h.to_numpy(mq.hit_dtype, n)
    assert np.array_equal(ph.view(np.uint8).reshape(n, -1), hits.view(np.uint8).reshape(n, -1)[perm])
    device_sync()


def test_config5_maize_like_repetitive(mq, oracle, simlib):
    T = _ncpu()
    lens = list(simlib.MAIZE_LIKE)
    assert len(lens) == 10 and max(lens) <= 308_500_000 and sum(lens) > 2_100_000_000
    # 1.9x the genome length pasted as family copies => ~85 % of the bases lie in a family copy; per-copy divergence
    # U[0.5 %, 2.5 %] from the consensus => copies of one family are 1-5 % apart
    g, off, names = simlib.make_genome(lens, seed=5005, threads=T, family_frac=1.9, n_families=400, family_div=(0.005, 0.025),
                                       tandem_frac=0.02, n_runs=300)
    assert (g == ord("N")).sum() > 1_000_000
    P, po = mq.Params(), oracle.params()
    ix, ox = mq.Index(P), oracle.Index()
    per_gpu = [ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])]) for r in range(10)]
    per_cpu = ox.build_mt(g, off, names, po, T)
    assert per_gpu == [int(x) for x in per_cpu]
    assert ix.finalize() == ox.count()
    st = ix.stats()
    assert st["n_keys"] == ox.keys()
    n = 20000
    reads = simlib.make_reads(g, off, n, seed=6006, threads=T)  # depth-30 recipe shape: mean 24 kb (simulate_maize.sh:9)
    bases, offs = reads["bases"], reads["offsets"]
    want, diag = ox.map_batch_diag(bases, offs, po, threads=T)
    hits = ix.map_batch(bases, offs)
    _cmp(hits, want)
    n_fast, n_gen = ix.last_map_path_counts()
    n_with_n = int(sum(1 for i in range(n) if (bases[int(offs[i]):int(offs[i + 1])] == ord("N")).any()))
    assert n_gen >= n_with_n > 0                       # reads over an N run take the general path
    assert (diag["n_candidates"] > 1).sum() > 0          # repeats offer more than one candidate reference
    print("maize-like: unique/keys %d/%d, mapped %.4f, reads with >1 candidate %d, ties %d, general-path reads %d"
          % (ox.count(), ox.keys(), (want["mapped"] != 0).mean(), int((diag["n_candidates"] > 1).sum()), int(diag["tie"].sum()), n_gen))
    rn = simlib.read_names(reads, names)
    assert ix.paf_lines(rn[:2000], offs[:2001], hits[:2000]) == oracle.paf_lines(ox, rn[:2000], want[:2000])
    # low-divergence variant of the same shape at 1/8 scale: young copies (0-0.4 % from the consensus) => many
    # tombstoned k-min-mers, miss-dominated probes, short Matches, overflow_reads reported by the device form
    lens8 = [x // 8 for x in lens]
    g2, off2, names2 = simlib.make_genome(lens8, seed=5006, threads=T, family_frac=1.9, n_families=60, family_div=(0.0, 0.004), n_runs=40)
    ix2, ox2 = mq.Index(P), oracle.Index()
    for r in range(10):
        ix2.add_ref(r, names2[r], g2[int(off2[r]):int(off2[r + 1])])
    inserted = int(ox2.build_mt(g2, off2, names2, po, T).sum())
    assert ix2.finalize() == ox2.count()
    assert ox2.count() < 0.5 * inserted                  # most of the reference's k-min-mers are repeated => tombstoned
    reads2 = simlib.make_reads(g2, off2, 10000, seed=6007, threads=T)
    want2, diag2 = ox2.map_batch_diag(reads2["bases"], reads2["offsets"], po, threads=T)
    hits2 = ix2.map_batch(reads2["bases"], reads2["offsets"])
    _cmp(hits2, want2)
    assert (hits2["status"] == 2).sum() == 0             # the host form retries overflowing reads on the GPU
    assert diag2["tie"].sum() > 0 and (diag2["n_candidates"] > 1).sum() > 1000 and diag2["filtered_out"].sum() > 1000
    assert diag2["n_hits"].sum() < 0.3 * diag2["n_kminmers"].sum()  # miss-dominated probes
    print("maize-like young repeats: unique/keys %d/%d, mapped %.4f, >1 candidate %d, ties %d"
          % (ox2.count(), ox2.keys(), (want2["mapped"] != 0).mean(), int((diag2["n_candidates"] > 1).sum()), int(diag2["tie"].sum())))


def test_config4_shape_fastq_k7_native_driver(mq, oracle, simlib, tmp_path):
    """BASELINE config 4's shape: FASTQ file + -k 7 (experiments/table1.sh:50-55) against a human-scale reference, file -> PAF
    through the native driver (feeder, spans form, stream slots, formatter pool), PAF bytes identical to the oracle's."""
    import os
    import subprocess
    import tempfile
    from mapquik_amd import build
    exe = build.build_cli()
    T = _ncpu()
    lens = list(simlib.CHM13_LIKE)
    g, off, names = simlib.make_genome(lens, seed=2013, threads=T, repeat_frac=0.05, tandem_frac=0.01)
    ps = dict(k=7, l=31, density=0.01)
    po = oracle.params(**ps)
    ox = oracle.Index()
    ox.build_mt(g, off, names, po, T)
    ns = 50000
    reads = simlib.make_reads(g, off, ns, seed=4013, threads=T)
    bases, offs = reads["bases"], reads["offsets"]
    want = ox.map_batch(bases, offs, po, threads=T)
    rn = simlib.read_names(reads, names)
    want_paf = "".join(x + "\n" for x in oracle.paf_lines(ox, rn, want)).encode()
    assert (want["mapped"] != 0).sum() > 0.9 * ns
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=base) as wd:
        ref, rd = os.path.join(wd, "ref.fa"), os.path.join(wd, "reads.fastq")
        with open(ref, "wb") as f:
            for r in range(len(lens)):
                f.write(b">" + names[r].encode() + b"\n")
                g[int(off[r]):int(off[r + 1])].tofile(f)
                f.write(b"\n")
        del g
        qual = np.full(int((offs[1:] - offs[:-1]).max()), ord("I"), dtype=np.uint8)
        with open(rd, "wb") as f:
            for i in range(ns):
                f.write(b"@" + rn[i].encode() + b" np:i:12\n")
                L = int(offs[i + 1] - offs[i])
                bases[int(offs[i]):int(offs[i + 1])].tofile(f)
                f.write(b"\n+\n")
                qual[:L].tofile(f)
                f.write(b"\n")
        prefix = os.path.join(wd, "c4")
        r = subprocess.run([exe, rd, "--reference", ref, "-p", prefix, "-k", "7", "-l", "31", "-d", "0.01", "--threads", str(min(T, 16))],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-400:]
        assert "Format: FASTA" in r.stdout and "Mapped query sequences in" in r.stdout
        got = open(prefix + ".paf", "rb").read()
    assert got == want_paf and len(want_paf) > 4_000_000
    n_m, n_q60, n_wrong = simlib.mapeval({k: v for k, v in reads.items() if k not in ("bases", "offsets")}, want)
    assert n_q60 > 0.95 * ns and n_wrong <= 5
