"""GPU: stream-slot contexts (mq_ctx), the spans form with in-kernel case folding, and the minimizer-list pool -- all against
the CPU oracle, bit-exact.  Reference behaviour: workers map concurrently over one read-only index (src/closures.rs:183,187,
src/index.rs:108-116); sequences are upper-cased before the seam (src/closures.rs:63,106)."""
import threading

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
def small(simlib):
    return simlib.make_genome([900000, 600000], seed=41, repeat_frac=0.1, tandem_frac=0.02)


def _index_both(mq, oracle, small, ps, **kw):
    g, off, names = small
    P, po = mq.Params(**ps, **kw), oracle.params(**ps)
    ix, ox = mq.Index(P), oracle.Index()
    for r in range(off.size - 1):
        s = g[int(off[r]):int(off[r + 1])]
        assert ix.add_ref(r, names[r], s) == ox.add_ref(r, names[r], s, po)
    assert ix.finalize() == ox.count()
    return ix, ox, po


def _cmp(hits, want):
    assert np.array_equal(hits["status"] == 1, want["mapped"] != 0)
    m = want["mapped"] != 0
    import mapquik_amd
    for a in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
        assert np.array_equal(mapquik_amd.hit_column(hits, a)[m], want[a][m].astype(np.uint64)), a


def test_contexts_map_concurrently_on_one_index(mq, oracle, simlib, small):
    """Three contexts of one finalized index driven from three threads at once, several rounds each; plus submit/wait double
    buffering on one thread.  Every batch must equal the oracle's answer."""
    g, off, names = small
    ix, ox, po = _index_both(mq, oracle, small, dict())
    batches = [simlib.make_reads(g, off, 300, seed=100 + i) for i in range(3)]
    wants = [ox.map_batch(b["bases"], b["offsets"], po, threads=4) for b in batches]
    errs = []

    def worker(i):
        try:
            c = ix.context()
            for _ in range(4):
                _cmp(c.map_batch(batches[i]["bases"], batches[i]["offsets"]), wants[i])
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    # index-level entry point from several threads at once: serialised by the index lock, still right
    th = [threading.Thread(target=lambda i=i: _cmp(ix.map_batch(batches[i]["bases"], batches[i]["offsets"]), wants[i])) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    # double buffering: two contexts, submit both, then wait both
    c0, c1 = ix.context(), ix.context()
    c0.submit(batches[0]["bases"], batches[0]["offsets"])
    c1.submit(batches[1]["bases"], batches[1]["offsets"])
    _cmp(c1.wait(), wants[1])
    _cmp(c0.wait(), wants[0])
    with pytest.raises(mq.MapquikError):
        c0.submit(batches[0]["bases"], batches[0]["offsets"])
        c0.submit(batches[0]["bases"], batches[0]["offsets"])  # a second submit before wait is refused
    c0.wait()


@pytest.mark.parametrize("ps", [dict(), dict(k=3, l=12, density=0.05, use_hpc=False)])
def test_spans_of_a_raw_fastq_buffer_with_case_folding(mq, oracle, simlib, small, ps):
    """Reads handed over as spans of raw FASTQ text (headers, '+' and quality lines in between), half of the bases in lower
    case, some reads with N / n: MQ_FLAG_FOLD_CASE makes the kernels do the reference's to_ascii_uppercase."""
    g, off, names = small
    ix, ox, po = _index_both(mq, oracle, small, ps, fold_case=True)
    reads = simlib.make_reads(g, off, 200, seed=77, len_mean=9000, len_sd=5000, len_min=20)
    bases, offs = reads["bases"].copy(), reads["offsets"]
    rng = np.random.default_rng(3)
    for i in range(0, 200, 9):  # a few reads get N runs: those take the general seeder
        a = int(offs[i]) + 5
        bases[a:a + 7] = ord("N")
    want = ox.map_batch(bases, offs, po, threads=4)
    mixed = bases.copy()
    low = rng.random(mixed.size) < 0.5
    mixed[low] |= 0x20  # a-z for half of the bases (N -> n too)
    parts, starts, lens, pos = [], [], [], 0
    for i in range(200):
        s = mixed[int(offs[i]):int(offs[i + 1])].tobytes()
        hdr = b"@read%d some description\n" % i
        rec = hdr + s + b"\n+\n" + b"I" * len(s) + b"\n"
        starts.append(pos + len(hdr))
        lens.append(len(s))
        parts.append(rec)
        pos += len(rec)
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8)
    c = ix.context()
    c.submit_spans(buf, starts, lens)
    hits = c.wait()
    _cmp(hits, want)
    # the same through the plain offsets form: lower case folds there too
    _cmp(ix.map_batch(mixed, offs), want)
    # without the flag, lower-case reads hash as non-ACGT (the seam's contract: already upper-cased): answers differ
    ix2, _, _ = _index_both(mq, oracle, small, ps)
    h2 = ix2.map_batch(mixed, offs)
    assert (h2["status"] == 1).sum() < (want["mapped"] != 0).sum() // 4


def test_dense_lists_move_to_the_pool_and_pool_exhaustion_is_loud(mq, oracle, simlib, small, monkeypatch):
    """MQ_LIST_F16=1 leaves every read a 64-entry region: the lists of all but the shortest reads outgrow it and are written
    again into pool regions (same answers).  With a density so high that the pool runs out too, the device-resident entry
    point reports MQ_HIT_OVERFLOW for the reads that did not fit and the host-buffer form redoes exactly those."""
    import ctypes as C
    g, off, names = small
    monkeypatch.setenv("MQ_LIST_F16", "1")
    ix, ox, po = _index_both(mq, oracle, small, dict())
    reads = simlib.make_reads(g, off, 400, seed=5)
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=4)
    _cmp(ix.map_batch(reads["bases"], reads["offsets"]), want)
    # pool exhaustion: density 0.5 -> ~0.5 entries per base against a pool of 2^20 entries
    ps = dict(k=3, l=14, density=0.5)
    ix, ox, po = _index_both(mq, oracle, small, ps)
    reads = simlib.make_reads(g, off, 400, seed=6, len_mean=12000)
    bases, offs = reads["bases"], reads["offsets"]
    assert bases.size * 0.3 > (1 << 20)
    want = ox.map_batch(bases, offs, po, threads=4)
    _cmp(ix.map_batch(bases, offs), want)  # host form: complete
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    db, do, dout = C.c_void_p(), C.c_void_p(), C.c_void_p()
    n = offs.size - 1
    assert hip.hipMalloc(C.byref(db), bases.size) == 0 and hip.hipMalloc(C.byref(do), offs.size * 8) == 0 and hip.hipMalloc(C.byref(dout), n * 48) == 0
    hip.hipMemcpy(db, bases.ctypes.data, bases.size, 1)
    hip.hipMemcpy(do, offs.ctypes.data, offs.size * 8, 1)
    ix.map_batch_device(db.value, do.value, n, int(offs[-1] - offs[0]), dout.value, 0)
    raw = np.zeros(n, dtype=mq.hit_dtype)
    hip.hipMemcpy(raw.ctypes.data, dout, n * 48, 2)
    over = raw["status"] == 2
    assert over.any() and not over.all()
    ok = ~over
    assert np.array_equal(raw["status"][ok] == 1, want["mapped"][ok] != 0)
    for p in (db, do, dout):
        hip.hipFree(p)


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_line_small_run(mode):
    """bench.py end to end at a small scale in both scaling modes: one JSON line with the contract's keys, the roofline and
    (weak mode) the end-to-end section; strong mode deals one host-resident read set through mapquik_amd.shard."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--genome-scale", "0.01", "--reads", "3000", "--steps", "2", "--warmup", "1",
           "--scaling", mode, "--no-cpu-baseline", "--e2e-file-reads", "1500"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "end_to_end", "q60", "q60_wrong"):
        assert k in j, k
    assert j["scaling"] == mode and j["value"] > 0 and j["overflow_reads"] == 0
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["q60"] > 0.9 * 3000 and j["q60_wrong"] <= 2
    e = j["end_to_end"]
    assert e["host_buffers_hits_identical"] and e["host_buffers_gbases_s"] > 0
    assert e["file_to_paf"].get("paf_lines", 0) > 1300, e


def test_bench_line_with_a_real_reference_and_real_reads(simlib, tmp_path):
    """bench.py's real-data door (BASELINE configs 3 / 4 the day chm13v2.0.fa and the HG002 FASTQ are on the box): a line-wrapped,
    soft-masked reference FASTA and (a) pbsim2fq-named FASTQ reads: data "real", mapeval against the names' truth; (b) reads whose names
    carry no truth: q60_wrong is null, the truth-free counts stay; (c) no reads file: reads simulated from the given reference; the
    roofline, the oracle's column check (cpu_baseline) and the rest of the line unchanged."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g, off, names = simlib.make_genome([900000, 500000, 40], seed=5, repeat_frac=0.05)
    ref = tmp_path / "ref.fa"
    with open(ref, "wb") as f:
        for r in range(3):
            s = g[int(off[r]):int(off[r + 1])].tobytes()
            s = s[:1000].lower() + s[1000:]
            f.write(b">" + names[r].encode() + b" a description\n" + b"".join(s[i:i + 80] + b"\n" for i in range(0, len(s), 80)))
    reads = simlib.make_reads(g, off, 4000, seed=9)
    rn = simlib.read_names(reads, names)
    o = reads["offsets"]
    fq, fa = tmp_path / "reads.fastq", tmp_path / "anon.fa"
    with open(fq, "wb") as f, open(fa, "wb") as h:
        for i, n in enumerate(rn):
            s = reads["bases"][int(o[i]):int(o[i + 1])].tobytes()
            f.write(b"@" + n.encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
            h.write(b">m64011_190830_220126/%d/ccs\n" % i + s + b"\n")
    base = [sys.executable, os.path.join(root, "bench.py"), "--reads", "3000", "--steps", "2", "--warmup", "1", "--no-e2e", "--cpu-sample-reads", "512",
            "--reference-fasta", str(ref)]
    out = {}
    for tag, extra in (("fq", ["--reads-fastx", str(fq)]), ("anon", ["--reads-fastx", str(fa), "--no-cpu-baseline"]), ("sim", ["--no-cpu-baseline"])):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        j = out[tag] = json.loads(r.stdout.strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in j, k
        assert j["value"] > 0 and j["overflow_reads"] == 0 and j["roofline"]["bound"] == "hbm" and j["configs"] is None
        assert "ref.fa" in j["config"]["workload"] and j["config"]["reads_per_step_per_gpu"] == 3000 and j["roofline"]["traffic"] is None
        assert j["config"]["index_unique_kminmers"] > 10000
    assert out["fq"]["data"] == "real" and "reads.fastq" in out["fq"]["config"]["workload"]
    assert out["fq"]["q60"] > 0.9 * 3000 and out["fq"]["q60_wrong"] is not None and out["fq"]["q60_wrong"] <= 2
    cb = out["fq"]["cpu_baseline"]
    assert cb["kind"] == "port" and cb["paf_columns_identical_to_gpu"] and cb["unique_kminmers_equal"]
    assert out["anon"]["data"] == "real" and out["anon"]["q60_wrong"] is None and out["anon"]["q60"] == out["fq"]["q60"]  # the same sequences
    assert out["anon"]["mapped_reads"] == out["fq"]["mapped_reads"]
    assert out["sim"]["data"] == "real reference, simulated reads" and out["sim"]["q60"] > 0.9 * 3000 and out["sim"]["q60_wrong"] <= 2


def test_index_clone_is_a_deep_replica(mq, oracle, simlib, small):
    g, off, names = small
    ix, ox, po = _index_both(mq, oracle, small, dict())
    reads = simlib.make_reads(g, off, 200, seed=8)
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=4)
    rep = ix.clone(0)
    assert rep.stats() == ix.stats()
    ix.close()  # the replica owns its table
    _cmp(rep.map_batch(reads["bases"], reads["offsets"]), want)
    assert rep.ref_info(1)[0] == names[1]


def test_cloned_replicas_map_from_two_threads_at_once(mq, oracle, simlib, small):
    """What `--gpus N` does with N real devices, as far as one device can show it: the finalized index cloned (device-to-device
    copy of the table), the source and the replica mapping the SAME batch from two threads at the same time, each through its own
    stream slots, several rounds -- both give the oracle's result every time, and the source can be freed while the replica works."""
    import threading
    g, off, names = small
    ix, ox, po = _index_both(mq, oracle, small, dict())
    reads = simlib.make_reads(g, off, 600, seed=18)
    want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=4)
    rep = ix.clone(0)
    assert rep.stats() == ix.stats()
    out, errs = {}, []

    def work(tag, index, rounds):
        try:
            ctxs = [index.context() for _ in range(2)]
            res = []
            for k in range(rounds):
                c = ctxs[k % 2]
                c.submit(reads["bases"], reads["offsets"])
                res.append(c.wait().copy())
            for c in ctxs:
                c.close()
            out[tag] = res
        except Exception as e:  # noqa: BLE001
            errs.append((tag, repr(e)))

    th = [threading.Thread(target=work, args=("src", ix, 6)), threading.Thread(target=work, args=("rep", rep, 6))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for tag in ("src", "rep"):
        for h in out[tag]:
            _cmp(h, want)
    ix.close()  # the replica owns its own table
    h = rep.map_batch(reads["bases"], reads["offsets"])
    assert np.array_equal(h.view(np.uint8), out["rep"][0].view(np.uint8))


def test_native_driver_second_pass(mq, oracle, simlib, tmp_path):
    """--second-pass k2,l2,d2 (experiments/chm13/run_chm13_mapquik_unmapped.sh:8-24 in one process): the reads the first pass
    leaves unmapped are written as FASTA and mapped again with the second parameter set; both PAFs equal the oracle's."""
    import subprocess
    from mapquik_amd import build
    exe = build.build_cli()
    g, off, names = simlib.make_genome([700000, 500000], seed=77, repeat_frac=0.1)
    clean = simlib.make_reads(g, off, 120, seed=1, len_mean=9000, len_sd=3000)
    noisy = simlib.make_reads(g, off, 60, seed=2, len_mean=9000, len_sd=3000, err=0.06)  # mostly unmapped at k=5 l=31
    seqs, ids = [], []
    for tag, rd in (("c", clean), ("n", noisy)):
        o = rd["offsets"]
        for i in range(o.size - 1):
            seqs.append(rd["bases"][int(o[i]):int(o[i + 1])])
            ids.append("%s%d" % (tag, i))
    order = np.random.default_rng(4).permutation(len(seqs))
    seqs, ids = [seqs[i] for i in order], [ids[i] for i in order]
    ref, rdp = tmp_path / "ref.fa", tmp_path / "reads.fa"
    with open(ref, "wb") as w:
        for r in range(2):
            w.write(b">" + names[r].encode() + b"\n" + g[int(off[r]):int(off[r + 1])].tobytes() + b"\n")
    with open(rdp, "wb") as w:
        for n, s in zip(ids, seqs):
            w.write(b">" + n.encode() + b"\n" + s.tobytes() + b"\n")
    prefix = str(tmp_path / "p1")
    r = subprocess.run([exe, str(rdp), "--reference", str(ref), "-p", prefix, "--second-pass", "4,14,0.05", "--batch-bases", "300000"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Mapped query sequences in") == 2 and "Second pass:" in r.stdout

    def oracle_paf(ps, names_, seqs_):
        po = oracle.params(**ps)
        ox = oracle.Index()
        for k in range(2):
            ox.add_ref(k, names[k], g[int(off[k]):int(off[k + 1])], po)
        offs = np.zeros(len(seqs_) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([s.size for s in seqs_])
        bases = np.concatenate(seqs_) if seqs_ else np.zeros(0, dtype=np.uint8)
        res = ox.map_batch(bases, offs, po, threads=2)
        return oracle.paf_lines(ox, names_, res), res

    want1, res1 = oracle_paf(dict(), ids, seqs)
    assert open(prefix + ".paf").read() == "".join(x + "\n" for x in want1)
    un = [i for i in range(len(ids)) if res1["mapped"][i] == 0]
    assert 20 < len(un) < 120
    assert open(prefix + ".unmapped.out").read().split() == [ids[i] for i in un]
    p2 = prefix + "-4-14-0.05"
    fa = open(p2 + ".fa").read().split("\n")
    assert fa[0::2][:len(un)] == [">" + ids[i] for i in un] and fa[1] == seqs[un[0]].tobytes().decode()
    want2, res2 = oracle_paf(dict(k=4, l=14, density=0.05), [ids[i] for i in un], [seqs[i] for i in un])
    assert open(p2 + ".paf").read() == "".join(x + "\n" for x in want2)
    assert (res2["mapped"] != 0).sum() > len(un) // 2  # the second parameter set recovers most of them


@pytest.mark.parametrize("ps", [dict(), dict(k=3, l=12, density=0.05), dict(use_hpc=False, k=7, l=64, density=0.02)])
def test_split_pipeline_equals_the_fused_kernel(mq, oracle, simlib, small, monkeypatch, ps):
    """MQ_PIPELINE=split (seed_reads_kernel, seed_general_kernel, map_lists_kernel: the diagnostic form a profiler prices phase by
    phase) must give the product kernel's bytes: same device functions.  Reads with N runs go through the queue to the general
    seeder there; MQ_LIST_F16=1 on top forces the pool path of both seeders; k-min-mer dumps agree too."""
    g, off, names = small
    reads = simlib.make_reads(g, off, 300, seed=31, len_mean=15000, len_sd=8000, len_min=10)
    bases, offs = reads["bases"].copy(), reads["offsets"]
    for i in range(0, 300, 7):
        a = int(offs[i]) + 3
        if a + 5 < int(offs[i + 1]):
            bases[a:a + 5] = ord("N")
    ix, ox, po = _index_both(mq, oracle, small, ps)
    want = ox.map_batch(bases, offs, po, threads=4)
    fused = ix.map_batch(bases, offs)
    _cmp(fused, want)
    kf = ix.kminmers_batch(bases[:int(offs[20])], offs[:21])
    monkeypatch.setenv("MQ_PIPELINE", "split")
    ix2, _, _ = _index_both(mq, oracle, small, ps)
    split = ix2.map_batch(bases, offs)
    assert np.array_equal(split.view(np.uint8), fused.view(np.uint8))
    n_fast, n_gen = ix2.last_map_path_counts()
    assert n_gen >= 40 and n_fast > 200
    ks = ix2.kminmers_batch(bases[:int(offs[20])], offs[:21])
    assert all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(kf, ks))
    monkeypatch.setenv("MQ_LIST_F16", "1")
    ix3, _, _ = _index_both(mq, oracle, small, ps)
    assert np.array_equal(ix3.map_batch(bases, offs).view(np.uint8), fused.view(np.uint8))
