#!/usr/bin/env python3
"""Diagnostic: which reads make map_kernel's tail.  One instrumented launch (mq_map_probe_stats) of a bench workload, then what every read
cost its wave (mq_last_read_cycles): the distribution, the most expensive reads with what they are (k-min-mers, score, mapped or not, where
they come from), and how much later than the average wave the last waves finish.

    python tools/read_tail.py [--genome-preset human-like] [--reads N] [--genome-scale S] [--seeding-variant v]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=196608)
    ap.add_argument("--genome-scale", type=float, default=1.0)
    ap.add_argument("--genome-preset", choices=("planted-repeats", "human-like", "maize-like"), default="planted-repeats")
    ap.add_argument("--seeding-variant", type=int, default=0)
    a = ap.parse_args()
    import torch
    import mapquik_amd as mq
    from tools import sim
    dev = torch.device("cuda", 0)
    lens = [max(40, int(x * a.genome_scale)) for x in (sim.MAIZE_LIKE if a.genome_preset == "maize-like" else sim.CHM13_LIKE)]
    kw = (sim.HUMAN_LIKE if a.genome_preset == "human-like" else
          dict(family_frac=1.9, n_families=400, family_div=(0.005, 0.025), tandem_frac=0.02, n_runs=300) if a.genome_preset == "maize-like" else
          dict(repeat_frac=0.05, tandem_frac=0.01, div=0.01))
    g, off, names = sim.make_genome(lens, seed=2013, threads=16, **kw)
    ix = mq.Index(mq.Params(seeding_variant=a.seeding_variant), device=0)
    for r in range(len(lens)):
        d = torch.from_numpy(g[int(off[r]):int(off[r + 1])]).to(dev)
        ix.add_ref_device(r, names[r], d.data_ptr(), int(off[r + 1] - off[r]))
        del d
    ix.finalize()
    reads = sim.make_reads(g, off, a.reads, seed=3013, threads=16)
    offs = reads["offsets"]
    n, total = offs.size - 1, int(offs[-1])
    db = torch.from_numpy(reads["bases"]).to(dev)
    do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    out = torch.zeros(n * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    for _ in range(2):
        ix.map_batch_device(db.data_ptr(), do.data_ptr(), n, total, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    print("plain launch: %.3f ms" % ix.last_map_ms())
    ix.probe_stats(db.data_ptr(), do.data_ptr(), n, total, out.data_ptr())
    torch.cuda.synchronize()
    print("instrumented launch: %.3f ms" % ix.last_map_ms())
    cyc, start = ix.last_read_cycles(n)
    # reads map_declined_kernel took: bit 63 of the start word, and the seeding split (units of 256 cycles) in its place
    declined = (start >> np.uint64(63)) != 0
    if declined.any():
        sv = start[declined]
        find, fast, gen = [((sv >> np.uint64(sh)) & np.uint64(0x1FFFFF)).astype(np.float64) * 256 for sh in (0, 21, 42)]
        cd = cyc[declined].astype(np.float64)
        print("declined reads (map_declined_kernel): %d; cycles per read: mean %.0f median %.0f; of which stretch finder %.0f, fast seeder on clean stretches %.0f, "
              "general seeder %.0f, the rest (map phase, list stores) %.0f" % (int(declined.sum()), cd.mean(), np.median(cd), find.mean(), fast.mean(), gen.mean(),
                                                                              (cd - find - fast - gen).mean()))
        start = start.copy()
        start[declined] = np.median(start[~declined]) if (~declined).any() else 0  # (they carry no start tick)
    hits = np.frombuffer(out.cpu().numpy().tobytes(), dtype=mq.hit_dtype)
    c = cyc.astype(np.float64)
    print("cycles per read: mean %.0f  median %.0f  p99 %.0f  p99.9 %.0f  max %.0f  (sum %.4g)" % (c.mean(), np.median(c), np.percentile(c, 99), np.percentile(c, 99.9), c.max(), c.sum()))
    for thr in (2, 4, 8, 16, 64):
        m = c > thr * np.median(c)
        print("  reads costing > %2dx the median: %6d  = %.2f %% of all wave-cycles" % (thr, int(m.sum()), 100.0 * c[m].sum() / c.sum()))
    t0 = int(start.min())
    rel_end = (start.astype(np.int64) - t0) / 100.0 + c / 2100.0  # us since the first read was taken up (100-MHz ticks; cycles at ~2.1 GHz)
    order = np.argsort(-c)[:25]
    lens_r = (offs[1:] - offs[:-1]).astype(np.int64)
    print("the 25 most expensive reads (cycles, x median, bases, k-min-mers, status, score, taken up at [us], true origin):")
    for i in order:
        rb = reads["bases"][int(offs[i]):int(offs[i + 1])]
        print("  %9d %6.1fx %6d %5d  st %d score %4d  at %7.1f us  %s:%d%s  other bytes %d" % (cyc[i], c[i] / np.median(c), lens_r[i], hits["n_kminmers"][i], hits["status"][i], hits["score"][i],
                                                                           (int(start[i]) - t0) / 100.0, names[int(reads["ctg"][i])], int(reads["start"][i]), "-" if reads["strand"][i] else "+",
                                                                           int((~np.isin(rb, np.frombuffer(b"ACGT", dtype=np.uint8))).sum())))
    nn = np.array([int((~np.isin(reads["bases"][int(offs[i]):int(offs[i + 1])], np.frombuffer(b"ACGT", dtype=np.uint8))).sum()) for i in np.nonzero(c > 3 * np.median(c))[0]])
    dirty = np.nonzero(c > 3 * np.median(c))[0][nn > 0]
    if dirty.size:
        print("reads with other bytes among those > 3x the median: %d; their cycles: mean %.0f median %.0f; other bytes per read: median %d" % (dirty.size, c[dirty].mean(), np.median(c[dirty]), int(np.median(nn[nn > 0]))))
    # what a read costs by when it was taken up: the launch's start (every wave in the same stage), its steady state, its end
    t_us = (start.astype(np.int64) - t0) / 100.0
    edges = [0, 25, 100, 200, 400, 800, 1600, 2400, 3200, 3600, 3700, 3800, 3900, 4000, 4200, 1e9]
    print("mean cycles per read by the time it was taken up [us]:")
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (t_us >= lo) & (t_us < hi)
        if m.any():
            print("  [%6.0f, %6.0f)  %7d reads  mean %8.0f  median %8.0f" % (lo, min(hi, 99999), int(m.sum()), c[m].mean(), np.median(c[m])))
    print("last read finished %.1f us after the first was taken up; the median wave's last read finished at about %.1f us" % (rel_end.max(), np.median(np.sort(rel_end)[-4096:])))


if __name__ == "__main__":
    main()
