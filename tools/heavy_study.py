#!/usr/bin/env python3
"""heavy_study.py -- CPU study behind the heavy-first launch order: per-read minimizer counts of the bench batch (oracle, test
infrastructure: this is a study tool, not a product path) against the periodicity test the device pre-pass uses, and a
list-scheduling estimate of a launch's tail in the natural order and with the flagged reads first.
usage: python tools/heavy_study.py [--genome-preset human-like] [--reads 196608] [--genome-scale 1.0]"""
import argparse
import os
import sys
import time
from multiprocessing.pool import ThreadPool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import sim  # noqa: E402
from oracle import oracle  # noqa: E402


def periodic_flag(win, max_lag=32, span=64, min_match=52):
    """win: uint8 array of span + max_lag bases: True when some lag 1..max_lag matches in >= min_match of span positions"""
    a = win[:span]
    for p in range(1, max_lag + 1):
        if int((a == win[p:p + span]).sum()) >= min_match:
            return p
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-preset", default="planted-repeats")
    ap.add_argument("--genome-scale", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=196608)
    ap.add_argument("--seed", type=int, default=913)
    ap.add_argument("--repeat-frac", type=float, default=None)
    ap.add_argument("--save", default=None)
    ap.add_argument("--min-match", type=int, default=52)
    ap.add_argument("--span", type=int, default=64)
    ap.add_argument("--max-lag", type=int, default=32)
    ap.add_argument("--counts", default=None, help="minimizer counts saved by an earlier run (--save) of the same workload")
    args = ap.parse_args()
    class d:  # bench.py's defaults
        repeat_frac, tandem_frac, repeat_div = 0.05, 0.01, 0.01
    lens = [max(40, int(x * args.genome_scale)) for x in sim.CHM13_LIKE]
    t0 = time.time()
    if args.genome_preset == "human-like":
        g, off, names = sim.make_genome(lens, seed=args.seed, threads=8, **sim.HUMAN_LIKE)
    else:
        g, off, names = sim.make_genome(lens, seed=args.seed, threads=8, repeat_frac=d.repeat_frac, tandem_frac=d.tandem_frac, div=d.repeat_div)
    rd = sim.make_reads(g, off, args.reads, seed=args.seed + 1000, threads=8)
    print("genome + reads in %.1f s" % (time.time() - t0), flush=True)
    del g
    b, o = rd["bases"], rd["offsets"].astype(np.int64)
    n = o.size - 1
    p = oracle.params()
    lib = oracle.lib()
    import ctypes as C

    def count(i):
        s = b[o[i]:o[i + 1]]
        return lib.mqo_minimizers(oracle._ptr(s), s.size, C.byref(p), None, 0)
    t0 = time.time()
    if args.counts:
        cnt = np.load(args.counts)
    else:
        with ThreadPool(8) as tp:
            cnt = np.array(tp.map(count, range(n), chunksize=256), dtype=np.int64)
    print("minimizer counts in %.1f s: median %d, p99 %d, p99.9 %d, max %d" % (time.time() - t0, np.median(cnt), np.percentile(cnt, 99), np.percentile(cnt, 99.9), cnt.max()), flush=True)
    if args.save:
        np.save(args.save, cnt)
    # the periodicity test at three places of every read
    t0 = time.time()
    flag = np.zeros(n, dtype=np.int32)
    ln = o[1:] - o[:-1]
    ok = ln >= 400
    for fr in (1, 3, 5):
        at = o[:-1] + (ln * fr) // 6
        at = np.where(ok, at, 0)
        W = b[at[:, None] + np.arange(96)[None, :]]  # n x 96
        hit = np.zeros(n, dtype=bool)
        for lag in range(1, args.max_lag + 1):
            hit |= (W[:, :args.span] == W[:, lag:lag + args.span]).sum(axis=1) >= args.min_match
        flag += (hit & ok).astype(np.int32)
    print("periodicity test in %.1f s: flagged (any window) %d, (>= 2 windows) %d of %d" % (time.time() - t0, (flag >= 1).sum(), (flag >= 2).sum(), n))
    med = np.median(cnt)
    for thr in (2, 3, 5, 10):
        heavy = cnt > thr * med
        print("  reads with > %2d x the median count: %6d; of them flagged (any) %6d, (>= 2) %6d" % (thr, heavy.sum(), (heavy & (flag >= 1)).sum(), (heavy & (flag >= 2)).sum()))
    # list scheduling on 4096 wave slots, a read's cost = 1 + 0.9 (count / median - 1) (a launch's measured mixed-batch ratio: 14 x the minimizers = 3.9 x the cycles)
    cost = 1.0 + 0.22 * (cnt / med - 1.0)
    cost = np.maximum(cost, 0.3)

    def sched(order, slots=4096):
        import heapq
        h = [0.0] * slots
        heapq.heapify(h)
        for i in order:
            t = heapq.heappop(h)
            heapq.heappush(h, t + cost[i])
        a = np.array(h)
        return a.max(), np.median(a), cost.sum() / slots
    nat = np.arange(n)
    for name, order in (("natural order", nat), ("flagged (any) first", np.concatenate([nat[flag >= 1], nat[flag < 1]])),
                        ("flagged (>= 2) first", np.concatenate([nat[flag >= 2], nat[flag < 2]])),
                        ("true heaviest 1 % first (unknowable)", np.concatenate([nat[np.argsort(-cnt)[:n // 100]], np.setdiff1d(nat, nat[np.argsort(-cnt)[:n // 100]])]))):
        mx, md, ideal = sched(order)
        print("  %-40s makespan %.2f, median slot %.2f, ideal %.2f: tail %.1f %%" % (name, mx, md, ideal, 100.0 * (mx - ideal) / mx))


if __name__ == "__main__":
    main()
