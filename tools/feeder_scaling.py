#!/usr/bin/env python3
"""Can the host side feed N GPUs?  (1) the FASTX feeder alone (feeder_dump, no GPU work) at 1..16 threads on FASTA, FASTQ and
BGZF input; (2) the native driver with 1, 2 and 4 workers sharing ONE device (MQ_FAKE_MULTI: every worker its own index replica
and stream slots -- what --gpus N does, minus the extra devices) at the bench's size.  Writes what profiles/r03_feeder_scaling.txt
holds.  Diagnostic tool: not part of the product path."""
import os
import re
import struct
import subprocess
import sys
import tempfile
import time
import zlib
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bgzf_piece(piece):
    out = bytearray()
    for i in range(0, len(piece), 65280):
        p = piece[i:i + 65280]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        d = c.compress(p) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(d) + 8 - 1) + d
        out += struct.pack("<II", zlib.crc32(p) & 0xFFFFFFFF, len(p))
    return bytes(out)


def main():
    from mapquik_amd import build as B
    from tools import sim
    tool, exe = B.build_feeder_dump(), B.build_cli()
    ncpu = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            ncpu = min(ncpu, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    n_reads = int(os.environ.get("FS_READS", "196608"))
    lens = [int(x) for x in sim.CHM13_LIKE]
    print("# host: %d CPUs granted; reads: %d (HiFi-like, mean 24 kb); files in /dev/shm (page cache)" % (ncpu, n_reads))
    g, off, names = sim.make_genome(lens, seed=2013, threads=ncpu, repeat_frac=0.05, tandem_frac=0.01)
    reads = sim.make_reads(g, off, n_reads, seed=3013, threads=ncpu)
    bases = int(reads["offsets"][-1])
    with tempfile.TemporaryDirectory(dir="/dev/shm") as wd:
        fa, fq, bg = os.path.join(wd, "r.fa"), os.path.join(wd, "r.fastq"), os.path.join(wd, "q.fa.gz")
        sim.write_fastx(fa, reads["bases"], reads["offsets"], n_reads, fastq=False, threads=ncpu)
        sim.write_fastx(fq, reads["bases"], reads["offsets"], n_reads, fastq=True, threads=ncpu)
        nq = n_reads // 4
        qa = os.path.join(wd, "q.fa")
        sim.write_fastx(qa, reads["bases"], reads["offsets"], nq, fastq=False, threads=ncpu)
        raw = open(qa, "rb").read()
        step = 65280 * 64
        with ProcessPoolExecutor(ncpu) as ex:
            parts = list(ex.map(_bgzf_piece, [raw[i:i + step] for i in range(0, len(raw), step)]))
        with open(bg, "wb") as f:
            for p_ in parts:
                f.write(p_)
            f.write(_bgzf_piece(b""))
        qbases = int(reads["offsets"][nq])
        del raw, parts
        print("\n(1) feeder alone: Gbases/s parsed (chunks of 32 MB, malloc'ed buffers, best of 2)")
        print("%-28s" % "threads" + "".join("%8d" % t for t in (1, 2, 4, 8, 16)))
        for name, path, kind, nb in (("FASTA %.1f GB" % (os.path.getsize(fa) / 1e9), fa, "fasta", bases),
                                     ("FASTQ %.1f GB (lean reader)" % (os.path.getsize(fq) / 1e9), fq, "fastq", bases),
                                     ("BGZF  %.1f GB inflated" % (os.path.getsize(qa) / 1e9), bg, "fasta", qbases)):
            row = []
            for t in (1, 2, 4, 8, 16):
                best = 1e9
                for _ in range(2):
                    t0 = time.perf_counter()
                    r = subprocess.run([tool, path, kind, str(1 << 25), str(t)], capture_output=True, text=True, env=dict(os.environ, FEEDER_DUMP_QUIET="1"))
                    dt = time.perf_counter() - t0
                    assert r.returncode == 0, r.stderr
                    best = min(best, dt)
                row.append(nb / best / 1e9)
            print("%-28s" % name + "".join("%8.1f" % x for x in row))
        ref = os.path.join(wd, "ref.fa")
        with open(ref, "wb") as f:
            for r_ in range(len(names)):
                f.write(b">" + names[r_].encode() + b"\n")
                g[int(off[r_]):int(off[r_ + 1])].tofile(f)
                f.write(b"\n")
        print("\n(2) native driver, N workers on ONE device (MQ_FAKE_MULTI), FASTA %.1f Gbases, --threads %d: map phase, whole job" % (bases / 1e9, ncpu))
        for gpus in (1, 2, 4):
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([exe, fa, "--reference", ref, "-p", os.path.join(wd, "o"), "--threads", str(ncpu), "--gpus", str(gpus)],
                                   capture_output=True, text=True, env=dict(os.environ, MQ_FAKE_MULTI="1", MQ_DRIVER_NO_PREFETCH="1"))
                wall = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr[-300:]
            m = re.search(r"Mapped query sequences in ([0-9.]+)(s|ms)", r.stdout)
            tm = float(m.group(1)) * (1.0 if m.group(2) == "s" else 1e-3)
            mi = re.search(r"Indexed [0-9]+ unique k-min-mers in ([0-9.]+)(s|ms)", r.stdout)
            ti = float(mi.group(1)) * (1.0 if mi.group(2) == "s" else 1e-3)
            print("   --gpus %d: index phase %.2f s (build once + %d table copies), map phase %.3f s = %.1f Gbases/s, wall %.2f s"
                  % (gpus, ti, gpus - 1, tm, bases / tm / 1e9, wall))


if __name__ == "__main__":
    main()
