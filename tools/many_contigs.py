import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import mapquik_amd as mq
from oracle import oracle as O
from tools import sim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
g, off, names = sim.make_genome([5000] * n, seed=3, repeat_frac=0.0, tandem_frac=0.0)
ix = mq.Index(mq.Params())
t = time.time()
tot = 0
for r in range(n):
    tot += ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])])
u = ix.finalize()
dt = time.time() - t
print("%d contigs of 5 kb: %d k-min-mers, %d unique, %.2f s = %.0f us per contig" % (n, tot, u, dt, dt / n * 1e6))
ox = O.Index(); po = O.params()
ox.build_mt(g, off, names, po, 8)
assert ox.count() == u, (ox.count(), u)
reads = sim.make_reads(g, off, 200, seed=1, len_mean=3000, len_sd=500, len_max=4500)
h = ix.map_batch(reads["bases"], reads["offsets"]); w = ox.map_batch(reads["bases"], reads["offsets"], po, threads=4)
assert np.array_equal(h["status"] == 1, w["mapped"] != 0) and np.array_equal(h["r_start"][w["mapped"] != 0], w["r_start"][w["mapped"] != 0].astype(np.uint32))
print("parity ok, mapped", int((h["status"] == 1).sum()))
