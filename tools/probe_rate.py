"""Diagnostic: random-probe rate of the full-scale index table on this GPU (tools/probe_rate.py [genome_scale]).
Builds bench.py's CHM13-like index, then times mq_probe_rate at several grid sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mapquik_amd as mq
from tools import sim

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
lens = [max(40, int(x * scale)) for x in sim.CHM13_LIKE]
genome, off, names = sim.make_genome(lens, seed=1, threads=16, repeat_frac=0.05, tandem_frac=0.01)
ix = mq.Index(mq.Params())
for r in range(len(lens)):
    seg = torch.from_numpy(genome[int(off[r]):int(off[r + 1])]).cuda()
    ix.add_ref_device(r, names[r], seg.data_ptr(), seg.numel())
    del seg
print("unique", ix.finalize(), ix.stats())
for blocks, per in ((256 * 4, 256), (256 * 8, 128), (256 * 32, 32)):
    ms, n, extra = ix.probe_rate(blocks, per)
    print("table only            blocks %6d x 256 threads x %4d lookups: %.3f ms  %.1f G lookups/s  p-bar %.3f" % (blocks, per, ms, n / ms / 1e6, 1 + extra / n))
for lg in (24, 26, 28, 29, 30):
    for tt in (0, 1):
        ms, n, extra = ix.probe_rate(256 * 8, 128, lg, tt)
        print("bitmap 2^%d bits (%4d MB) %s: %.3f ms  %.1f G lookups/s" % (lg, (1 << lg) >> 23, "then table for 1/8" if tt else "only             ", ms, n / ms / 1e6))
