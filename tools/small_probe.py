"""Diagnostic: what a 49,152-read launch costs by what came before it (cold start, a row of launches, right after the big batch, after 2 s of idle): an idle GPU
clocks down and the first dozen milliseconds after a pause run ~20 % slow -- why bench.py warms its `smaller_batches` launches up.   python tools/small_probe.py [reads]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench, mapquik_amd as mq
from tools import sim
dev = torch.device("cuda", 0)
P = mq.Params()
lens = list(sim.CHM13_LIKE)
ix0 = mq.Index(P, device=0); ix0.reserve_table(bench.expected_kminmers(sum(lens), P))
g, off, names = sim.make_genome(lens, seed=2013, threads=16, repeat_frac=0.05, tandem_frac=0.01, div=0.01)
ix, per_ref, n_unique, t_up, t_idx = bench.build_index_device(mq, torch, dev, 0, P, g, off, names, ix=ix0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1572864
B = bench.load_batch(torch, dev, sim, g, off, n, 2013 + 1000, 16)
d_out = torch.empty(n * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
ix.reserve(B.n, B.total_bases)
st = torch.cuda.current_stream(dev)
def run(m, reps, tag):
    tb = int(B.offsets[m])
    for _ in range(2): ix.map_batch_device(B.d_bases.data_ptr(), B.d_offs.data_ptr(), m, tb, d_out.data_ptr(), st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): ix.map_batch_device(B.d_bases.data_ptr(), B.d_offs.data_ptr(), m, tb, d_out.data_ptr(), st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    print("%-40s m=%7d  %.4f ms" % (tag, m, e0.elapsed_time(e1) / reps), flush=True)
run(49152, 10, "cold start, small first")
run(49152, 200, "small, 200 launches in a row")
run(n, 10, "big x 10")
run(49152, 10, "small right after big")
run(49152, 10, "small again")
time.sleep(2.0)
run(49152, 10, "small after 2 s idle")
torch.cuda.synchronize(); x = d_out.clone(); torch.cuda.synchronize()
run(49152, 10, "small after a clone")
