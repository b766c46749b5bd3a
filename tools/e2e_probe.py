"""PCIe-inclusive rate of the host-buffer entry point (never the bench value): pageable vs page-locked input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mapquik_amd as mq
from tools import sim
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
n = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
lens = [max(40, int(x * scale)) for x in sim.CHM13_LIKE]
g, off, names = sim.make_genome(lens, seed=2013, threads=16, repeat_frac=0.05, tandem_frac=0.01)
ix = mq.Index(mq.Params())
for r in range(len(lens)):
    ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])])
ix.finalize()
reads = sim.make_reads(g, off, n, seed=3013, threads=16)
bases, offs = reads["bases"], reads["offsets"]
nb = int(offs[-1])
pin = mq.PinnedBuffer(nb)
pin.array[:] = bases
for name, buf in (("pageable", bases), ("page-locked", pin.array)):
    ix.map_batch(buf, offs)
    t = time.time()
    for _ in range(3):
        hits = ix.map_batch(buf, offs)
    dt = (time.time() - t) / 3
    print("%-12s %.1f ms per batch of %d reads / %.3f Gbases -> %.1f Gbases/s end to end (kernel %.2f ms)" % (name, dt * 1e3, n, nb / 1e9, nb / dt / 1e9, ix.last_map_ms()))
