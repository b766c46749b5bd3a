import sys, time; sys.path.insert(0,'.')
import numpy as np
import mapquik_amd as mq
from oracle import oracle as O
from tools import sim
print("devices", mq.device_count())
g, off, names = sim.make_genome(sim.ECOLI_LEN, seed=913)
reads = sim.make_reads(g, off, 2000, seed=1)
P = mq.Params(); po = O.params()
ix = mq.Index(P); ox = O.Index()
t=time.time(); n1 = ix.add_ref(0, names[0], g); nu = ix.finalize(); t1=time.time()-t
t=time.time(); n2 = ox.add_ref(0, names[0], g, po); t2=time.time()-t
print("index gpu", n1, nu, "%.3fs"%t1, "cpu", n2, ox.count(), "%.3fs"%t2, ix.stats())
t=time.time(); hits = ix.map_batch(reads["bases"], reads["offsets"]); t1=time.time()-t
t=time.time(); hits = ix.map_batch(reads["bases"], reads["offsets"]); t1b=time.time()-t
print("kernel ms", ix.last_map_ms())
t=time.time(); want = ox.map_batch(reads["bases"], reads["offsets"], po, threads=1); t2=time.time()-t
nb = int(reads["offsets"][-1])
print("map gpu(host api) %.3fs / %.3fs cpu1 %.3fs bases %d -> kernel Gbases/s %.2f cpu Mbases/s %.1f" % (t1, t1b, t2, nb, nb/ix.last_map_ms()/1e6, nb/t2/1e6))
print("mapped", (hits["status"]==1).sum(), (want["mapped"]!=0).sum(), "equal", all(np.array_equal(hits[a].astype(np.uint64)[want["mapped"]!=0], want[a].astype(np.uint64)[want["mapped"]!=0]) for a in ("ref_id","rc","mapq","q_start","q_end","r_start","r_end","score")))
print(hits[:3])
print(sim.mapeval(reads, want))
