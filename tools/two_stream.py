#!/usr/bin/env python3
"""Diagnostic for VERDICT r02 item 1(a): do the seed phase (issue-bound) and the map phase (latency-bound) overlap better as two
kernels on two streams than inside the fused persistent kernel?  The bench batch is mapped (device-resident, steady state)
  fused, one stream            : map_kernel over the whole batch                          (the product path)
  split, one stream            : seed_reads_kernel then map_lists_kernel, whole batch     (MQ_PIPELINE=split)
  split, two / four streams    : the batch in 2 / 4 parts, each part's seed + map launches on its own stream slot (mq_ctx), so that
                                 part i's map_lists_kernel can run beside part i+1's seed_reads_kernel
  fused, two streams           : two half batches of the fused kernel side by side (control: what splitting the batch alone costs)
Prints ms per whole batch (wall clock around all streams, min of 5).  Run once per MQ_PIPELINE value (the mode is read when the index
is created): tools/two_stream.py  and  MQ_PIPELINE=split tools/two_stream.py.  Not part of the product path."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import mapquik_amd as mq
    from tools import sim
    dev = torch.device("cuda", 0)
    lens = [int(x) for x in sim.CHM13_LIKE]
    g, off, names = sim.make_genome(lens, seed=2013, threads=8, repeat_frac=0.05, tandem_frac=0.01, div=0.01)
    ix = mq.Index(mq.Params(), device=0)
    for r in range(len(lens)):
        d = torch.from_numpy(g[int(off[r]):int(off[r + 1])]).to(dev)
        ix.add_ref_device(r, names[r], d.data_ptr(), int(off[r + 1] - off[r]))
        del d
    ix.finalize()
    n = 196608
    reads = sim.make_reads(g, off, n, seed=3013, threads=8)
    offs = reads["offsets"].astype(np.int64)
    db = torch.from_numpy(reads["bases"]).to(dev)
    mode = os.environ.get("MQ_PIPELINE", "fused")
    for parts in (1, 2, 4):
        cuts = [n * i // parts for i in range(parts + 1)]
        ctxs = [ix.context() for _ in range(parts)]
        streams = [torch.cuda.Stream(dev) for _ in range(parts)]
        d_offs, outs, totals = [], [], []
        for i in range(parts):
            o = offs[cuts[i]:cuts[i + 1] + 1]
            d_offs.append(torch.from_numpy(o.copy()).to(dev))  # absolute offsets into db
            outs.append(torch.zeros((cuts[i + 1] - cuts[i]) * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev))
            totals.append(int(o[-1] - o[0]))
        best = 1e9
        for it in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(parts):
                ctxs[i].map_batch_device(db.data_ptr(), d_offs[i].data_ptr(), cuts[i + 1] - cuts[i], totals[i], outs[i].data_ptr(), streams[i].cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if it >= 3:
                best = min(best, dt)
        hits = np.concatenate([o.cpu().numpy().view(mq.hit_dtype) for o in outs])
        print("%-6s pipeline, %d stream(s): %.3f ms per %d-read batch = %.0f Gbases/s; Q60 %d" % (
            mode, parts, best, n, int(offs[-1]) / best / 1e6, int((hits["mapq"] == 60).sum())))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
