"""Diagnostic: per-stage cycle shares of map_kernel (MQ_STAGE_TIMING=1 build).  Shares only; never a reported time."""
import os, sys, time
os.environ["MQ_STAGE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mapquik_amd as mq
from tools import sim
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
lens = [max(40, int(x * scale)) for x in sim.CHM13_LIKE]
g, off, names = sim.make_genome(lens, seed=2013, threads=64, repeat_frac=0.05, tandem_frac=0.01)
ix = mq.Index(mq.Params())
for r in range(len(lens)):
    ix.add_ref(r, names[r], g[int(off[r]):int(off[r+1])])
ix.finalize()
reads = sim.make_reads(g, off, nreads, seed=3013, threads=64)
dev = torch.device("cuda", 0)
db = torch.from_numpy(reads["bases"]).to(dev); do = torch.from_numpy(reads["offsets"].astype(np.int64)).to(dev)
out = torch.zeros(nreads * 40, dtype=torch.uint8, device=dev)
ml = int(reads["offsets"][-1] - reads["offsets"][0])  # total bases
for _ in range(3):
    ix.map_batch_device(db.data_ptr(), do.data_ptr(), nreads, ml, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
cyc = ix.last_stage_cycles().astype(np.float64)
names_ = ["A decode+HPC", "B rolling hash", "C gather+consume", "finish", "chain", "wave total"]
print("kernel ms (instrumented):", ix.last_map_ms())
for n_, c in zip(names_, cyc[:6]):
    print("%-18s %14.0f cycles  %5.1f %% of wave total" % (n_, c, 100 * c / cyc[5]))
print("other (work pull, epilogue): %.1f %%" % (100 * (cyc[5] - cyc[:5].sum()) / cyc[5]))
print("in-kernel clock (s_memtime / s_memrealtime x 100 MHz): %.3f GHz" % (cyc[5] / max(cyc[8], 1) * 0.1))
print("index lookups %d, extra probe steps %d -> mean probes per lookup %.3f" % (cyc[7], cyc[6], 1 + cyc[6] / max(cyc[7], 1)))
