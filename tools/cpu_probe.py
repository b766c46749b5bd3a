import sys, time, os; sys.path.insert(0,'.')
from oracle import oracle as O
from tools import sim
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)'")
g, off, names = sim.make_genome([200_000_000], seed=913, threads=32)
reads = sim.make_reads(g, off, 20000, seed=1, threads=32)
p = O.params(); ix = O.Index()
t=time.time(); ix.build_mt(g, off, names, p, 8); print("build", time.time()-t)
nb = int(reads["offsets"][-1])
for th in (1, 8, 16, 32, 64, 128, 256):
    t=time.time(); out = ix.map_batch(reads["bases"], reads["offsets"], p, threads=th); dt=time.time()-t
    print(th, "threads: %.3fs %.1f Mbases/s" % (dt, nb/dt/1e6), flush=True)
