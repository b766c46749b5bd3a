#!/usr/bin/env python3
"""The FASTX feeder on ONE large gzip member (the 196,608-read bench batch as FASTA, gzip -1: 4.6 GB in 1.5 GB): time and peak RSS
with the many-thread inflater and with the one-call libdeflate reader (no GPU work).  Diagnostic tool: not part of the product path."""
import os, sys, subprocess, time, resource, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mapquik_amd import build as B
from tools import sim
tool = B.build_feeder_dump()
n = 196608
g, off, names = sim.make_genome([60_000_000], seed=2013, threads=16, repeat_frac=0.05, tandem_frac=0.01)
reads = sim.make_reads(g, off, n, seed=3013, threads=16)
bases = int(reads["offsets"][n])
with tempfile.TemporaryDirectory(dir="/dev/shm") as wd:
    raw = os.path.join(wd, "r.fa")
    size = sim.write_fastx(raw, reads["bases"], reads["offsets"], n, fastq=False, threads=16)
    t0 = time.time(); subprocess.run(["gzip", "-1", raw], check=True); gz = raw + ".gz"
    print("compressed %.2f GB -> %.2f GB in %.0f s" % (size / 1e9, os.path.getsize(gz) / 1e9, time.time() - t0), flush=True)
    runner = ("import resource, subprocess, sys, time; t0=time.time(); r = subprocess.run(sys.argv[1:], capture_output=True, text=True); "
              "print(r.returncode, resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss, round(time.time()-t0,3), r.stdout.strip(), r.stderr[-200:])")
    for env in ({}, {"MQ_PARGZ": "0"}):
        r = subprocess.run([sys.executable, "-c", runner, tool, gz, "fasta", str(64 << 20), "16"], capture_output=True, text=True,
                           env=dict(os.environ, FEEDER_DUMP_QUIET="1", **env))
        print(env, r.stdout.strip(), r.stderr[-300:], flush=True)
    print("bases", bases)
