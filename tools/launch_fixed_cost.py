#!/usr/bin/env python3
"""launch_fixed_cost.py -- what a map launch costs beyond its reads: time of back-to-back launches of the first n reads of the bench batch for a
range of n, and the straight line through them (intercept = fixed cost per launch, slope = time per read).
    python tools/launch_fixed_cost.py [--genome-scale S] [--max-reads N]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-reads", type=int, default=786432)
    ap.add_argument("--genome-scale", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import torch
    import mapquik_amd as mq
    from tools import sim
    dev = torch.device("cuda", 0)
    lens = [max(40, int(x * a.genome_scale)) for x in sim.CHM13_LIKE]
    g, off, names = sim.make_genome(lens, seed=2013, threads=16, repeat_frac=0.05, tandem_frac=0.01, div=0.01)
    ix = mq.Index(mq.Params(), device=0)
    for r in range(len(lens)):
        d = torch.from_numpy(g[int(off[r]):int(off[r + 1])]).to(dev)
        ix.add_ref_device(r, names[r], d.data_ptr(), int(off[r + 1] - off[r]))
        del d
    ix.finalize()
    reads = sim.make_reads(g, off, a.max_reads, seed=3013, threads=16)
    offs = reads["offsets"]
    db = torch.from_numpy(reads["bases"]).to(dev)
    do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    out = torch.zeros(a.max_reads * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    ix.reserve(a.max_reads, int(offs[-1]))
    ns, ts = [], []
    n = 4096
    sizes = []
    while n < a.max_reads:
        sizes.append(n)
        n *= 2
    sizes.append(a.max_reads)
    for n in sizes:
        total = int(offs[n])
        for _ in range(3):
            ix.map_batch_device(db.data_ptr(), do.data_ptr(), n, total, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(a.reps):
            ix.map_batch_device(db.data_ptr(), do.data_ptr(), n, total, out.data_ptr(), st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        ns.append(n)
        ts.append(ms)
        print("%8d reads  %9.4f ms per launch  %8.1f Gbases/s  %7.3f us per 4096 reads" % (n, ms, total / ms / 1e6, ms * 1e3 * 4096 / n), flush=True)
    ns, ts = np.array(ns, dtype=np.float64), np.array(ts)
    big = ns >= 65536
    b, c = np.polyfit(ns[big], ts[big], 1)
    print("line through n >= 65536: %.4f ms + %.4f ms per 196,608 reads (asymptote %.1f Gbases/s)" % (c, b * 196608, float(offs[int(ns[-1])]) / ns[-1] / b / 1e6))


if __name__ == "__main__":
    main()
