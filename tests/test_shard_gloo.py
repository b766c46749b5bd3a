"""N>1 path on CPU: two gloo ranks each map their contiguous shard (the CPU oracle stands in for the per-rank engine:
the test is about sharding, ordering and the merge, not the kernel) and the merged result equals the single-rank run."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from mapquik_amd.shard import gather_hits, shard_reads
    from oracle import oracle as O
    from tools import sim
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, off, names = sim.make_genome([250000, 150000], seed=23, repeat_frac=0.1, threads=2)
    reads = sim.make_reads(g, off, 61, seed=6, len_mean=8000, len_sd=3000, threads=2)
    p = O.params()
    ix = O.Index()
    ix.build_mt(g, off, names, p, threads=2)  # index replicated: every rank builds its own copy
    b, o, lo = shard_reads(reads["bases"], reads["offsets"], world, rank)
    local = ix.map_batch(b, o, p, threads=1)
    merged = gather_hits(local)
    dist.barrier()
    if rank == 0:
        full = ix.map_batch(reads["bases"], reads["offsets"], p, threads=1)
        q.put((bool(np.array_equal(merged.view(np.uint8), full.view(np.uint8))), int(merged.size), int((full["mapped"] != 0).sum())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, n, n_mapped = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same and n == 61 and n_mapped > 40
