"""Diagnostic: FASTA file -> PAF through the native driver at 4 and 8 threads by chunk size (--batch-bases), medians of five runs each.
    python tools/chunk_probe.py"""
import os, re, statistics, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapquik_amd import build as B
from tools import sim

lens = list(sim.CHM13_LIKE)
genome, off, names = sim.make_genome(lens, seed=2013, threads=16, repeat_frac=0.05, tandem_frac=0.01)
reads = sim.make_reads(genome, off, 196608, seed=3013, threads=16)
exe = B.build_cli()
with tempfile.TemporaryDirectory(dir="/dev/shm") as wd:
    ref, rd = os.path.join(wd, "ref.fa"), os.path.join(wd, "reads.fa")
    with open(ref, "wb") as f:
        for r in range(len(names)):
            f.write(b">" + names[r].encode() + b"\n")
            genome[int(off[r]):int(off[r + 1])].tofile(f)
            f.write(b"\n")
    sim.write_fastx(rd, reads["bases"], reads["offsets"], 196608, fastq=False, threads=16)
    bases = int(reads["offsets"][-1])

    def run(extra):
        r = subprocess.run([exe, rd, "--reference", ref, "-p", os.path.join(wd, "o")] + extra, capture_output=True, text=True, timeout=900)
        m = re.search(r"Mapped query sequences in ([0-9.]+)(s|ms|µs|ns)", r.stdout)
        return float(m.group(1)) * {"s": 1, "ms": 1e-3, "µs": 1e-6, "ns": 1e-9}[m.group(2)]
    run(["--threads", "8"])
    for th in (4, 8):
        for mb in (8, 16, 32, 64, 128, 256):
            ts = sorted(run(["--threads", str(th), "--batch-bases", str(mb << 20)]) for _ in range(5))
            print("%d threads, %3d-MB chunks: map phase median %.4f s = %.1f Gbases/s  (min %.4f max %.4f)" % (th, mb, ts[2], bases / ts[2] / 1e9, ts[0], ts[-1]), flush=True)
