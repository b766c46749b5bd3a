"""Diagnostic: file -> PAF rate of the native driver for several thread counts / chunk sizes / input formats, on the bench's
CHM13-like workload.  usage: python tools/e2e_sweep.py [genome_scale] [n_reads]"""
import gzip, os, re, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapquik_amd import build as B
from tools import sim

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 196608
lens = [max(40, int(x * scale)) for x in sim.CHM13_LIKE]
genome, off, names = sim.make_genome(lens, seed=2013, threads=16, repeat_frac=0.05, tandem_frac=0.01)
reads = sim.make_reads(genome, off, n_reads, seed=3013, threads=16)
exe = B.build_cli()
base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
with tempfile.TemporaryDirectory(dir=base) as wd:
    ref, rd, fq = os.path.join(wd, "ref.fa"), os.path.join(wd, "reads.fa"), os.path.join(wd, "reads.fq")
    with open(ref, "wb") as f:
        for r in range(len(names)):
            f.write(b">" + names[r].encode() + b"\n")
            genome[int(off[r]):int(off[r + 1])].tofile(f)
            f.write(b"\n")
    o = reads["offsets"]
    with open(rd, "wb") as f, open(fq, "wb") as g:
        for i in range(n_reads):
            s = reads["bases"][int(o[i]):int(o[i + 1])]
            f.write(b">r%d\n" % i); s.tofile(f); f.write(b"\n")
            if i < n_reads // 4:
                g.write(b"@r%d\n" % i); s.tofile(g); g.write(b"\n+\n"); g.write(b"I" * s.size); g.write(b"\n")
    bases = int(o[n_reads]); qbases = int(o[n_reads // 4])

    def run(path, nb, extra, tag, env=None):
        t0 = time.perf_counter()
        r = subprocess.run([exe, path, "--reference", ref, "-p", os.path.join(wd, "o")] + extra, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **(env or {})))
        wall = time.perf_counter() - t0
        m = re.search(r"Mapped query sequences in ([0-9.]+)(s|ms|µs|ns)", r.stdout)
        t = float(m.group(1)) * {"s": 1, "ms": 1e-3, "µs": 1e-6, "ns": 1e-9}[m.group(2)] if m else float("nan")
        print("%-44s map phase %.3f s = %.2f Gbases/s   wall %.1f s  rc %d %s" % (tag, t, nb / t / 1e9, wall, r.returncode, r.stderr[-200:] if r.returncode else ""), flush=True)
        for ln in r.stderr.splitlines():
            if ln.startswith("map phase "):
                print("      " + ln, flush=True)

    NP = {"MQ_DRIVER_NO_PREFETCH": "1"}
    if os.environ.get("E2E_R5Q"):  # round 5: FASTQ (-k 7, BASELINE config 4's shape) by reader and thread count; the whole read set as FASTQ
        fqa = os.path.join(wd, "all.fastq")
        sim.write_fastx(fqa, reads["bases"], o, n_reads, fastq=True, threads=16)
        k7 = ["-k", "7", "-l", "31", "-d", "0.01"]
        run(fqa, bases, ["--threads", "8"] + k7, "warm-up")
        for rep in range(2):
            for mode in ("lean", "device"):
                for th in (2, 4, 8, 16):
                    run(fqa, bases, ["--threads", str(th)] + k7, "FASTQ %-6s %2d threads" % (mode, th), {"MQ_DRIVER_FASTQ": mode})
        run(rd, bases, ["--threads", "4"] + k7, "FASTA (for the ratio) 4 threads")
        run(rd, bases, ["--threads", "16"] + k7, "FASTA (for the ratio) 16 threads")
        sys.exit(0)
    if os.environ.get("E2E_R5M"):  # round 5: medians of the job's phases over repeated runs, by driver option
        import statistics
        def once(extra, env):
            t0 = time.perf_counter()
            r = subprocess.run([exe, rd, "--reference", ref, "-p", os.path.join(wd, "o")] + extra, capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, **env))
            wall = time.perf_counter() - t0
            unit = {"s": 1, "ms": 1e-3, "µs": 1e-6, "ns": 1e-9}
            def grab(pat):
                m = re.search(pat + r" ([0-9.]+)(s|ms|µs|ns)", r.stdout)
                return float(m.group(1)) * unit[m.group(2)] if m else float("nan")
            return wall, grab(r"unique k-min-mers in"), grab(r"Mapped query sequences in"), grab(r"Total execution time:"), r.returncode
        once(["--threads", "4"], {})
        reps = int(os.environ["E2E_R5M"])
        for env in ({}, {"LOWMEM": "1"}, {"MQ_DRIVER_LATE_SLOTS": "1"}, {"MQ_DRIVER_NO_RESERVE": "1"}, {"LOWMEM": "1", "MQ_DRIVER_NO_RESERVE": "1"},
                    {"MQ_DRIVER_FAST_EXIT": "1"}, {"MQ_DRIVER_REF_HOST": "1"}, {"MQ_TABLE_FACTOR": "8"}):
            for th in (4, 8):
                rows = [once(["--threads", str(th)] + (["--low-memory"] if env.get("LOWMEM") else []), env) for _ in range(reps)]
                med = lambda i: statistics.median(r[i] for r in rows)
                print("%-62s %d thr: wall %.3f  index %.3f  map %.3f  total-in-main %.3f  (outside main %.3f)  rc %s" %
                      (env, th, med(0), med(1), med(2), med(3), med(0) - med(3), {r[4] for r in rows}), flush=True)
        sys.exit(0)
    if os.environ.get("E2E_R5T"):  # round 5: the whole job's timeline (MQ_DRIVER_TIMING), by table factor and thread count
        def tline(extra, tag, env):
            t0 = time.perf_counter()
            r = subprocess.run([exe, rd, "--reference", ref, "-p", os.path.join(wd, "o")] + extra, capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, MQ_DRIVER_TIMING="1", **env))
            print("== %s: wall %.3f s rc %d" % (tag, time.perf_counter() - t0, r.returncode), flush=True)
            for ln in r.stderr.splitlines():
                print("      " + ln, flush=True)
            for ln in r.stdout.splitlines():
                if "unique k-min-mers" in ln or "Mapped query" in ln or "Total execution" in ln:
                    print("      " + ln, flush=True)
        tline(["--threads", "4"], "warm-up", {})
        for rep in range(2):
            for extra in ([], ["--low-memory"]):
                for th in (4, 8):
                    tline(["--threads", str(th)] + extra, "FASTA %d threads %s" % (th, extra), {})
        for extra_env in ({"MQ_TABLE_FACTOR": "8"}, {"MQ_DRIVER_NO_RESERVE": "1"}, {"MQ_DRIVER_REF_HOST": "1"}):
            for th in (4, 8):
                tline(["--threads", str(th)], "FASTA %d threads %s" % (th, extra_env), extra_env)
        sys.exit(0)
    if os.environ.get("E2E_BGZF"):  # a bgzip'ed FASTA of a quarter of the reads: blocks inflated in parallel by the reader threads
        import struct, zlib
        from concurrent.futures import ThreadPoolExecutor
        nq = n_reads // 4
        raw = b"".join(b">r%d\n" % i + reads["bases"][int(o[i]):int(o[i + 1])].tobytes() + b"\n" for i in range(nq))

        def blk(piece):
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            d = c.compress(piece) + c.flush()
            return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(d) + 8 - 1) + d +
                    struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
        pieces = [raw[i:i + 65280] for i in range(0, len(raw), 65280)] + [b""]
        with ThreadPoolExecutor(16) as ex:
            blocks = list(ex.map(blk, pieces))
        bz = os.path.join(wd, "reads_q.fa.gz")
        with open(bz, "wb") as f:
            for b_ in blocks:
                f.write(b_)
        print("bgzf file: %.2f GB compressed, %.2f GB inflated" % (os.path.getsize(bz) / 1e9, len(raw) / 1e9), flush=True)
        for th in (16, 8, 4):
            run(bz, int(o[nq]), ["--threads", str(th)], "FASTA bgzf (quarter of the reads), %d threads, no prefetch" % th, NP)
        run(bz, int(o[nq]), ["--threads", "16"], "FASTA bgzf (quarter of the reads), 16 threads, prefetch")
        sys.exit(0)
    if os.environ.get("E2E_R4M"):  # the mapped file, its page tables filled while the reference is indexed
        T = {"MQ_DRIVER_TIMING": "1"}
        run(rd, bases, ["--threads", "4"], "warm-up")
        for th in (2, 4, 8):
            run(rd, bases, ["--threads", str(th)], "FASTA %d threads, mapped early, copies from pageable memory" % th, dict(T, MQ_FEEDER_MAPPED_FASTA="1"))
        for th in (2, 4, 8):
            run(rd, bases, ["--threads", str(th)], "FASTA %d threads, mapped early, pages locked by the readers" % th, dict(T, MQ_FEEDER_MAPPED_FASTA="1", MQ_FEEDER_PAGE_LOCK="1"))
        for th in (4, 8):
            run(rd, bases, ["--threads", str(th)], "FASTA %d threads, pread chunks" % th, T)
        sys.exit(0)
    if os.environ.get("E2E_R4T"):  # where the threads' time goes (MQ_DRIVER_TIMING)
        T = {"MQ_DRIVER_TIMING": "1"}
        run(rd, bases, ["--threads", "4"], "warm-up")
        run(rd, bases, ["--threads", "4"], "FASTA 4 threads, mapped file, device records", dict(T, MQ_FEEDER_MAPPED_FASTA="1"))
        run(rd, bases, ["--threads", "8"], "FASTA 8 threads, mapped file, device records", dict(T, MQ_FEEDER_MAPPED_FASTA="1"))
        run(rd, bases, ["--threads", "4"], "FASTA 4 threads, pread chunks, device records", T)
        run(rd, bases, ["--threads", "8"], "FASTA 8 threads, pread chunks, device records", T)
        run(rd, bases, ["--threads", "4"], "FASTA 4 threads, host parse", dict(T, MQ_DRIVER_HOST_PARSE="1"))
        run(rd, bases, ["--threads", "8"], "FASTA 8 threads, host parse", dict(T, MQ_DRIVER_HOST_PARSE="1"))
        sys.exit(0)
    if os.environ.get("E2E_R4"):  # round 4 (profiles/r04_feeder_scaling.txt): records found on the device from views of the mapped file
        run(rd, bases, ["--threads", "4"], "warm-up (files into the page cache)")
        for th in (1, 2, 4, 8, 16):
            run(rd, bases, ["--threads", str(th)], "FASTA %2d threads, records found on the device (mapped file, pages locked)" % th, {"MQ_FEEDER_MAPPED_FASTA": "1"})
        for th in (4, 8, 16):
            run(rd, bases, ["--threads", str(th)], "FASTA %2d threads, records found on the device (pread chunks)" % th, {})
        for th in (4, 8, 16):
            run(rd, bases, ["--threads", str(th)], "FASTA %2d threads, parsed by the reader threads (round 3's path)" % th, {"MQ_DRIVER_HOST_PARSE": "1"})
        for cb in (1 << 23, 1 << 24, 1 << 26, 1 << 27):
            run(rd, bases, ["--threads", "4", "--batch-bases", str(cb)], "FASTA  4 threads, device records, chunks of %d MB" % (cb >> 20))
        for g in (2, 4):
            run(rd, bases, ["--threads", "8", "--gpus", str(g)], "FASTA  8 threads, %d workers on ONE device (MQ_FAKE_MULTI)" % g, {"MQ_FAKE_MULTI": "1"})
        run(fq, qbases, ["--threads", "4"], "FASTQ  4 threads (quarter of the reads; header + sequence lines copied by the readers)")
        run(fq, qbases, ["--threads", "16"], "FASTQ 16 threads (quarter of the reads)")
        sys.exit(0)
    if os.environ.get("E2E_BIG"):  # steady state: an input several times the size of the feeder's buffer pool
        reps = int(os.environ["E2E_BIG"])
        big = os.path.join(wd, "big.fa")
        with open(big, "wb") as f:
            for rep in range(reps):
                for i in range(n_reads):
                    f.write(b">r%d_%d\n" % (i, rep)); reads["bases"][int(o[i]):int(o[i + 1])].tofile(f); f.write(b"\n")
        for th in (16, 8):
            run(big, bases * reps, ["--threads", str(th)], "FASTA x%d (%d Gbases), %d threads, no prefetch" % (reps, bases * reps // 10**9, th), NP)
        run(big, bases * reps, ["--threads", "16"], "FASTA x%d, 16 threads, prefetch" % reps)
        gz = os.path.join(wd, "reads_q.fa.gz")
        with gzip.open(gz, "wb", compresslevel=1) as f:
            for i in range(n_reads // 8):
                f.write(b">r%d\n" % i); f.write(reads["bases"][int(o[i]):int(o[i + 1])].tobytes()); f.write(b"\n")
        run(gz, int(o[n_reads // 8]), ["--threads", "16"], "FASTA.gz (eighth of the reads), 16 threads, no prefetch", NP)
        os.remove(big)
    run(rd, bases, ["--threads", "16"], "FASTA 16 threads, prefetch during indexing")
    if os.environ.get("E2E_ONLY_FIRST"):
        r = subprocess.run([exe, rd, "--reference", ref, "-p", os.path.join(wd, "o"), "--threads", "16"], capture_output=True, text=True)
        print("\n".join(l for l in r.stdout.splitlines() if "Indexed " in l and "unique" in l or "Mapped" in l or "Total" in l))
        sys.exit(0)
    for th in (16, 8, 4, 2):
        run(rd, bases, ["--threads", str(th)], "FASTA %d threads, no prefetch" % th, NP)
    for cb in (1 << 25, 1 << 29):
        run(rd, bases, ["--threads", "16", "--batch-bases", str(cb)], "FASTA 16 threads chunk %d MB, no prefetch" % (cb >> 20), NP)
    run(fq, qbases, ["--threads", "16"], "FASTQ 16 threads (quarter of the reads), no prefetch", NP)
