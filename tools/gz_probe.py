#!/usr/bin/env python3
"""Plain-gzip input rates on this host (no GPU work): the many-thread member inflater alone (feeder_dump inflate) at 1..N threads,
and the whole feeder (inflate + cut + parse into chunk buffers) with it and with the one-thread libdeflate reader, on gzip -1 / -6
FASTA and FASTQ of HiFi-like reads.  Writes what profiles/r03_gz_probe.txt holds.  Diagnostic tool: not part of the product path."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mapquik_amd import build as B
    from tools import sim
    tool = B.build_feeder_dump()
    ncpu = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            ncpu = min(ncpu, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    n_reads = int(os.environ.get("GZ_READS", "49152"))
    g, off, names = sim.make_genome([60_000_000], seed=2013, threads=ncpu, repeat_frac=0.05, tandem_frac=0.01)
    reads = sim.make_reads(g, off, n_reads, seed=3013, threads=ncpu)
    bases = int(reads["offsets"][n_reads])
    print("# host: %d CPUs granted; %d HiFi-like reads, %.2f Gbases; files in /dev/shm; best of 3" % (ncpu, n_reads, bases / 1e9))

    def best(cmd, env=None, n=3):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FEEDER_DUMP_QUIET="1", **(env or {})))
            ts.append(time.perf_counter() - t0)
            assert r.returncode == 0, r.stderr[-500:]
        return min(ts), r.stdout.strip()

    with tempfile.TemporaryDirectory(dir="/dev/shm") as wd:
        for fastq, level, n_reads in ((False, 1, n_reads), (False, 6, n_reads // 4), (True, 1, n_reads)):
            raw = os.path.join(wd, "r.fq" if fastq else "r.fa")
            size = sim.write_fastx(raw, reads["bases"], reads["offsets"], n_reads, fastq=fastq, threads=ncpu)
            bases = int(reads["offsets"][n_reads])
            if True:
                gz = raw + ".%d.gz" % level
                t0 = time.perf_counter()
                with open(gz, "wb") as f:
                    subprocess.run(["gzip", "-%d" % level, "-c", raw], stdout=f, check=True)
                csize = os.path.getsize(gz)
                print("%s, %d reads, gzip -%d: %.2f GB -> %.3f GB (%.1f s to compress)" % ("FASTQ" if fastq else "FASTA", n_reads, level, size / 1e9, csize / 1e9, time.perf_counter() - t0))
                ths = [t for t in (1, 2, 4, 8, 12, 16, 24, 32) if t <= ncpu]
                for th in ths:
                    t, out = best([tool, gz, "inflate", str(6 << 20), str(th)])
                    inner = float(out.split()[1])
                    print("  inflater alone, %2d threads: %.3f s inside (%.2f GB/s out, %.2f Gbases/s); process %.3f s" % (th, inner, size / inner / 1e9, bases / inner / 1e9, t))
                kind = "fastq" if fastq else "fasta"
                for name, env in (("all threads", {}), ("libdeflate, one thread", {"MQ_PARGZ": "0"}), ("zlib stream", {"MQ_FEEDER_NO_LIBDEFLATE": "1"})):
                    t, out = best([tool, gz, kind, str(64 << 20), str(ncpu)], env, n=2 if env else 3)
                    assert out.split() == [str(n_reads), str(bases)], out
                    print("  feeder (%d threads), %-24s %.3f s = %.2f Gbases/s" % (ncpu, name + ":", t, bases / t / 1e9))
                os.remove(gz)
            os.remove(raw)


if __name__ == "__main__":
    main()
