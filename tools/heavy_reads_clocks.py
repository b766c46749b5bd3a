#!/usr/bin/env python3
"""Diagnostic: where the most expensive reads of a workload spend their time.  Finds the `--top` most expensive reads of the human-like bench batch
(one instrumented launch, mq_last_read_cycles), makes a batch of copies of just those, and prints the stage clocks of a launch over it
(MQ_LIB = a -DMQ_STAGE_CLOCKS build) next to their k-min-mer and Match-run counts.

    MQ_LIB=mapquik_amd/lib/clk4.so python tools/heavy_reads_clocks.py [--top 64] [--genome-preset human-like]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.stage_clocks import NAMES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=64)
    ap.add_argument("--copies", type=int, default=3)
    ap.add_argument("--genome-preset", choices=("planted-repeats", "human-like"), default="human-like")
    a = ap.parse_args()
    import torch
    import mapquik_amd as mq
    from tools import sim
    dev = torch.device("cuda", 0)
    kw = sim.HUMAN_LIKE if a.genome_preset == "human-like" else dict(repeat_frac=0.05, tandem_frac=0.01, div=0.01)
    g, off, names = sim.make_genome(list(sim.CHM13_LIKE), seed=2013, threads=16, **kw)
    ix = mq.Index(mq.Params(), device=0)
    for r in range(len(names)):
        d = torch.from_numpy(g[int(off[r]):int(off[r + 1])]).to(dev)
        ix.add_ref_device(r, names[r], d.data_ptr(), int(off[r + 1] - off[r]))
        del d
    ix.finalize()
    reads = sim.make_reads(g, off, 196608, seed=3013, threads=16)
    offs = reads["offsets"]

    def launch(bases, o, stats=False):
        n, total = o.size - 1, int(o[-1])
        db = torch.from_numpy(bases).to(dev)
        do = torch.from_numpy(o.astype(np.int64)).to(dev)
        out = torch.zeros(n * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev)
        if stats:
            ix.probe_stats(db.data_ptr(), do.data_ptr(), n, total, out.data_ptr())
        else:
            for _ in range(2):
                ix.map_batch_device(db.data_ptr(), do.data_ptr(), n, total, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        return np.frombuffer(out.cpu().numpy().tobytes(), dtype=mq.hit_dtype), n

    hits, n = launch(reads["bases"], offs, stats=True)
    cyc, _ = ix.last_read_cycles(n)
    top = np.argsort(-cyc.astype(np.int64))[:a.top]
    print("the %d most expensive reads: cycles %s ...; k-min-mers %s ...; mapped %d" % (a.top, cyc[top][:8].tolist(), hits["n_kminmers"][top][:8].tolist(), int((hits["status"][top] == 1).sum())))
    seqs = [reads["bases"][int(offs[i]):int(offs[i + 1])] for i in top]
    bases = np.concatenate(seqs * a.copies)
    o = np.zeros(len(seqs) * a.copies + 1, dtype=np.uint64)
    o[1:] = np.cumsum([s.size for s in seqs] * a.copies)
    # the expensive reads INSIDE a full batch (every 100th read replaced by one of them): the difference to the plain batch is what they cost under load
    mix_seqs = [reads["bases"][int(offs[i]):int(offs[i + 1])] for i in range(offs.size - 1)]
    n_mix = 0
    for j in range(0, len(mix_seqs), 100):
        mix_seqs[j] = seqs[(j // 100) % len(seqs)]
        n_mix += 1
    mb = np.concatenate(mix_seqs)
    mo = np.zeros(len(mix_seqs) + 1, dtype=np.uint64)
    mo[1:] = np.cumsum([s.size for s in mix_seqs])
    del mix_seqs
    for tag, (b_, o_) in (("the expensive reads x %d" % a.copies, (bases, o)), ("the whole batch", (reads["bases"], offs)),
                          ("the whole batch with %d reads replaced by expensive ones" % n_mix, (mb, mo))):
        h_, _ = launch(b_, o_)
        print("  (reads that came back MQ_HIT_OVERFLOW: %d; lists moved to the pool are seeded twice)" % int((h_["status"] == 2).sum()))
        clk = ix.last_stage_clocks()
        tot = float(sum(clk)) or 1.0
        print("%s: launch %.3f ms, %d reads; wave-cycles per read %.0f; shares:" % (tag, ix.last_map_ms(), o_.size - 1, tot / (o_.size - 1)))
        for nm, c in zip(NAMES, clk):
            if c:
                print("  %-32s %6.2f %%  %.0f cycles per read" % (nm, 100.0 * c / tot, c / (o_.size - 1)))


if __name__ == "__main__":
    main()
