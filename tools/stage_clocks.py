#!/usr/bin/env python3
"""Diagnostic: where a wave's time goes in map_kernel, by stage (s_memtime stamps at the stage boundaries).

    MQ_LIB=mapquik_amd/lib/<build with -DMQ_STAGE_CLOCKS>.so python tools/stage_clocks.py [--reads N] [--genome-scale S]

Prints the share of wave time per stage for one launch of the bench workload.  The instrumented build is slower (a stamp drains
the wave's LDS queue); shares, not absolute times, are what it is for.  Not part of the product path."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["stage A decode+HPC", "stage B rolling hash", "stage R candidates", "tile carry", "list stores acknowledged", "list -> LDS",
         "tuple hash + probe issue", "probe resolve + runs", "runs done, Match records in L2", "chain + result", "general seeder", "next read",
         "A*: wait for bases, super-rows 2, 3", "A*: before the loop, later tiles", "A*: wait for bases, later tiles", "A*: work"]
# -DMQ_STAGE_MAP_SPLIT builds instead: 12 = keys arrived + compared, payloads requested; 13 = lookups walking on; 14 = payloads arrived;
# "probe resolve + runs" = the runs alone.
# -DMQ_STAGE_R_SPLIT builds instead: 12 = stage R's flags read, masked, counted and scanned; 13 = its listing passes; 14 = its window hashes;
# 15 = its raw positions and list stores; "stage R candidates" = the rest of it.
# A*: builds with -DMQ_STAGE_A_SPLIT only; then "stage A" = wait for the bases of a sequence's first super-row, "general seeder" = what
# precedes the loop in a sequence's first tile (every stamp costs the wave an s_memtime round trip: ~400 cycles)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=196608)
    ap.add_argument("--genome-scale", type=float, default=1.0)
    ap.add_argument("--seeding-variant", type=int, default=0)
    a = ap.parse_args()
    import torch
    import mapquik_amd as mq
    from tools import sim
    dev = torch.device("cuda", 0)
    lens = [max(40, int(x * a.genome_scale)) for x in sim.CHM13_LIKE]
    g, off, names = sim.make_genome(lens, seed=2013, threads=8, repeat_frac=0.05, tandem_frac=0.01, div=0.01)
    ix = mq.Index(mq.Params(seeding_variant=a.seeding_variant), device=0)
    for r in range(len(lens)):
        d = torch.from_numpy(g[int(off[r]):int(off[r + 1])]).to(dev)
        ix.add_ref_device(r, names[r], d.data_ptr(), int(off[r + 1] - off[r]))
        del d
    ix.finalize()
    reads = sim.make_reads(g, off, a.reads, seed=3013, threads=8)
    offs = reads["offsets"]
    n, total = offs.size - 1, int(offs[-1])
    db = torch.from_numpy(reads["bases"]).to(dev)
    do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    out = torch.zeros(n * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    for _ in range(3):
        ix.map_batch_device(db.data_ptr(), do.data_ptr(), n, total, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    clk = ix.last_stage_clocks()
    tot = float(sum(clk)) or 1.0
    print("fast-path / general-path reads: %s" % (ix.last_map_path_counts(),))
    print("launch %.3f ms; wave-cycles per stage (share of the stamped total %.4g):" % (ix.last_map_ms(), tot))
    for nm, c in zip(NAMES, clk):
        print("  %-32s %6.2f %%  %.4g" % (nm, 100.0 * c / tot, c))


if __name__ == "__main__":
    main()
