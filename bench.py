#!/usr/bin/env python3
"""bench.py -- Gbases/s mapped by the HIP hot path on simulated CHM13-like HiFi reads (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the fused hot path (seeding -> index probe -> Match runs -> pseudo-chain) over one batch of
synthetic reads that is already resident in HBM (ASCII bases + offsets); the index is resident too.  Reads are sharded
across ranks with the index replicated per GPU, no data-path collective (weak scaling: every rank maps its own batch).
Rank 0 prints ONE JSON line.  The only collectives are the barrier and the MAX of the elapsed time.

Workload: CHM13v2.0 itself is not in this image, so the genome is a seeded synthetic stand-in with the same contig
lengths (tools/sim.py CHM13_LIKE, ~3.117 Gbp, 25 contigs) with planted repeats; reads follow the reference's pbsim
recipe shape (example/simulate_pbsim.sh:7-14: mean 24 kb, 1 % error).  --genome-scale shrinks the contigs for quick runs
(the JSON names the scale; only scale 1.0 is the BASELINE configuration).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=196608,
                    help="reads per step per GPU (a launch carries ~0.2 ms of start-up + tail, so small batches under-report)")
    ap.add_argument("--genome-scale", type=float, default=1.0, help="1.0 = CHM13-like 3.117 Gbp")
    ap.add_argument("--seed", type=int, default=2013)
    ap.add_argument("--repeat-frac", type=float, default=0.05, help="fraction of the genome overwritten by copied 1-20 kb segments "
                    "(0.8 = maize-like stress: most k-min-mers are tombstoned, lookups miss, Matches are short)")
    ap.add_argument("--tandem-frac", type=float, default=0.01)
    ap.add_argument("--repeat-div", type=float, default=0.01, help="per-base divergence of the planted copies")
    ap.add_argument("--genome-preset", choices=("planted-repeats", "human-like"), default="planted-repeats",
                    help="planted-repeats (default, the BASELINE stand-in of rounds 1-3): --repeat-frac / --tandem-frac / --repeat-div; human-like: "
                         "tools/sim.py HUMAN_LIKE (6 %% satellite arrays, 5 %% segmental duplications, young interspersed copies) -- reads from "
                         "inside satellites and recent duplications do not map uniquely, as on a real genome")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the BASELINE metric): every rank maps its own HBM-resident batch of --reads reads; strong: ONE fixed "
                         "read set of --reads reads in host memory is dealt to the ranks (mapquik_amd.shard) and mapped through the "
                         "host-buffer stream slots, PCIe included")
    ap.add_argument("--k", type=int, default=5, help="k-min-mer order (BASELINE config 4, experiments/table1.sh:50, runs -k 7)")
    ap.add_argument("--l", type=int, default=31)
    ap.add_argument("--density", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end measurements (host buffers -> hits, file -> PAF)")
    ap.add_argument("--e2e-file-reads", type=int, default=196608, help="reads written to the FASTA the native driver maps")
    ap.add_argument("--e2e-fastq-reads", type=int, default=786432,
                    help="reads of the FASTQ + -k 7 leg (BASELINE config 4's shape, experiments/table1.sh:50-55): 786,432 reads = 18.5 Gbases; 0 = skip")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    return ap.parse_args()


def effective_cpus():
    """Host threads this process may really use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)


class HostPipeline:
    """End to end from host memory: page-locked read buffers -> mq_ctx_submit/wait on n_ctx stream slots (copy-in, kernels and
    copy-out of consecutive sub-batches overlap) -> hits on the host."""

    def __init__(self, mq, ix, reads, n_ctx=3, sub_reads=16384):
        self.offs = reads["offsets"]
        self.n = self.offs.size - 1
        self.pin = mq.PinnedBuffer(max(1, int(self.offs[-1])))
        self.pin.array[:int(self.offs[-1])] = reads["bases"]
        self.ctxs = [ix.context() for _ in range(n_ctx)]
        self.subs = [(a, min(a + sub_reads, self.n)) for a in range(0, self.n, sub_reads)]

    def run_pass(self):
        offs, n_ctx = self.offs, len(self.ctxs)
        busy = [None] * n_ctx
        got = []
        for k, (a, b) in enumerate(self.subs):
            sl = k % n_ctx
            if busy[sl] is not None:
                got.append((busy[sl], self.ctxs[sl].wait()))
            self.ctxs[sl].submit(self.pin.array[int(offs[a]):int(offs[b])], offs[a:b + 1] - offs[a])
            busy[sl] = a
        for j in range(n_ctx):
            sl = (len(self.subs) + j) % n_ctx
            if busy[sl] is not None:
                got.append((busy[sl], self.ctxs[sl].wait()))
                busy[sl] = None
        return np.concatenate([h for _, h in sorted(got, key=lambda x: x[0])]) if got else np.zeros(0)

    def close(self):
        for c in self.ctxs:
            c.close()
        self.pin.close()


def measure_host_buffers(mq, ix, reads, passes=2):
    """Gbases/s of HostPipeline over `passes` passes of the batch, and the hits."""
    hp = HostPipeline(mq, ix, reads)
    hp.run_pass()  # warm-up: staging buffers grow to size
    t0 = time.perf_counter()
    for _ in range(passes):
        hits = hp.run_pass()
    dt = (time.perf_counter() - t0) / passes
    hp.close()
    return float(reads["offsets"][-1]) / dt / 1e9, hits


def write_reference(genome, ctg_off, ctg_names, path):
    with open(path, "wb") as f:
        for r in range(len(ctg_names)):
            f.write(b">" + ctg_names[r].encode() + b"\n")
            genome[int(ctg_off[r]):int(ctg_off[r + 1])].tofile(f)
            f.write(b"\n")


def measure_file_to_paf(ref, reads, n_reads, threads, workdir, fastq=False, extra_args=(), compress=None):
    """End to end through the native driver (mapquik_amd/lib/mapquik): reference FASTA + reads file on disk -> <prefix>.paf.
    fastq: the reads as a FASTQ file; compress="gz": as a plain gzip stream.  Three runs: one to bring the files into the page cache,
    the driver's default = the strict run (nothing of the reads is touched before the index is ready; also reported under the
    no_prefetch_* keys of earlier rounds) and one with MQ_DRIVER_PREFETCH=1 (the read feeder starts while the reference is
    indexed).  Rates over the driver's own 'Mapped query sequences' phase."""
    import re
    import subprocess
    from mapquik_amd import build as B
    from tools import sim
    exe = B.build_cli()
    offs = reads["offsets"]
    n_reads = min(n_reads, offs.size - 1)
    rd = os.path.join(workdir, "reads.fastq" if fastq else "reads.fa")
    file_bytes = sim.write_fastx(rd, reads["bases"], offs, n_reads, fastq=fastq, threads=threads)
    if compress == "gz":
        subprocess.run(["gzip", "-1", "-f", rd], check=True)
        rd += ".gz"
    bases = int(offs[n_reads])
    out = {"reads": n_reads, "bases": bases, "file_bytes": file_bytes, "threads": threads, "format": ("FASTQ" if fastq else "FASTA") + (".gz" if compress else ""),
           "args": " ".join(extra_args)}
    prefix = os.path.join(workdir, "e2e")

    def run(env):
        t0 = time.perf_counter()
        r = subprocess.run([exe, rd, "--reference", ref, "-p", prefix, "--threads", str(threads)] + list(extra_args),
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError((r.stderr or r.stdout)[-300:])
        m = re.search(r"Mapped query sequences in ([0-9.]+)(s|ms|\u00b5s|ns)", r.stdout)
        unit = {"s": 1.0, "ms": 1e-3, "\u00b5s": 1e-6, "ns": 1e-9}
        t_map = float(m.group(1)) * unit.get(m.group(2), 1.0) if m else float("nan")
        mi = re.search(r"Indexed [0-9]+ unique k-min-mers in ([0-9.]+)(s|ms|\u00b5s|ns)", r.stdout)
        t_idx = float(mi.group(1)) * unit.get(mi.group(2), 1.0) if mi else float("nan")
        return t_map, t_idx, wall

    try:
        run({})  # first run: files enter the page cache
        t_map, t_idx, wall = run({})
        out.update(gbases_s=round(bases / t_map / 1e9, 3), map_phase_s=round(t_map, 4), index_phase_s=round(t_idx, 3), driver_wall_s=round(wall, 2),
                   whole_job_gbases_s=round(bases / wall / 1e9, 3))
        out.update(no_prefetch_gbases_s=out["gbases_s"], no_prefetch_map_phase_s=out["map_phase_s"], no_prefetch_index_phase_s=out["index_phase_s"],
                   no_prefetch_driver_wall_s=out["driver_wall_s"])
        t_map2, t_idx2, wall2 = run({"MQ_DRIVER_PREFETCH": "1"})
        out.update(prefetch_gbases_s=round(bases / t_map2 / 1e9, 3), prefetch_map_phase_s=round(t_map2, 4), prefetch_index_phase_s=round(t_idx2, 3),
                   prefetch_driver_wall_s=round(wall2, 2))
        with open(prefix + ".paf", "rb") as f:
            out["paf_lines"] = sum(1 for _ in f)
    finally:
        for fn in (rd, prefix + ".paf"):
            try:
                os.remove(fn)
            except OSError:
                pass
    return out


def measure_index_file(mq, ix, P, device, workdir):
    """The on-disk index (occupied slots only): save, load (file in the page cache -> finalized table on the device), sizes."""
    p = os.path.join(workdir, "index.mqx")
    t0 = time.perf_counter()
    ix.save(p)
    t_save = time.perf_counter() - t0
    size = os.path.getsize(p)
    t0 = time.perf_counter()
    ix2 = mq.Index.load(p, device=device)
    t_load = time.perf_counter() - t0
    same = ix2.stats() == ix.stats()
    ix2.close()
    os.remove(p)
    return dict(file_bytes=size, save_s=round(t_save, 3), load_s=round(t_load, 3), stats_identical_after_load=bool(same))


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the mapquik HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import mapquik_amd as mq
    from tools import sim

    ncpu = effective_cpus()
    threads = max(1, ncpu // world)
    P = mq.Params(k=args.k, l=args.l, density=args.density)  # defaults k=5 l=31 d=0.01, HPC on, c=4 s=11 g=2000 (src/main.rs:174-188)

    # ---- genome (same on every rank: the index is replicated)
    t0 = time.time()
    lens = [max(40, int(x * args.genome_scale)) for x in sim.CHM13_LIKE]
    if args.genome_preset == "human-like":
        genome, ctg_off, ctg_names = sim.make_genome(lens, seed=args.seed, threads=threads, **sim.HUMAN_LIKE)
    else:
        genome, ctg_off, ctg_names = sim.make_genome(lens, seed=args.seed, threads=threads, repeat_frac=args.repeat_frac,
                                                     tandem_frac=args.tandem_frac, div=args.repeat_div)
    t_genome = time.time() - t0

    # ---- index on this rank's GPU (Index::add_with_mer + into_read_only on device)
    t0 = time.time()
    ix = mq.Index(P, device=local_rank)
    per_ref = []
    for r in range(len(lens)):
        seg = genome[int(ctg_off[r]):int(ctg_off[r + 1])]
        d_seg = torch.from_numpy(seg).to(dev)
        per_ref.append(ix.add_ref_device(r, ctg_names[r], d_seg.data_ptr(), seg.size))
        del d_seg
    n_unique = ix.finalize()
    torch.cuda.synchronize()
    t_index = time.time() - t0
    st = ix.stats()

    # ---- this rank's batch of reads, resident in HBM
    t0 = time.time()
    strong = args.scaling == "strong"
    if strong:  # the same read set on every rank; this rank's contiguous shard of it
        from mapquik_amd.shard import shard_bounds
        full = sim.make_reads(genome, ctg_off, args.reads, seed=args.seed + 1000, threads=threads)
        lo, hi = shard_bounds(args.reads, world, rank)
        fo = full["offsets"]
        reads = {k: v[lo:hi] for k, v in full.items() if k not in ("bases", "offsets")}
        reads["bases"] = full["bases"][int(fo[lo]):int(fo[hi])]
        reads["offsets"] = (fo[lo:hi + 1] - fo[lo]).astype(np.uint64)
        del full
    else:
        reads = sim.make_reads(genome, ctg_off, args.reads, seed=args.seed + 1000 + rank, threads=threads)
    offs = reads["offsets"]
    n = offs.size - 1
    total_bases = int(offs[-1])
    max_len = int((offs[1:] - offs[:-1]).max())
    d_bases = torch.from_numpy(reads["bases"]).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_out = torch.zeros(n * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
    t_reads = time.time() - t0
    ix.reserve(n, total_bases)
    stream = torch.cuda.current_stream(dev)

    pipe = HostPipeline(mq, ix, reads) if strong else None

    def step():
        if strong:
            pipe.run_pass()
        else:
            ix.map_batch_device(d_bases.data_ptr(), d_offs.data_ptr(), n, total_bases, d_out.data_ptr(), stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        step()
        ev[i][1].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([float(total_bases), float(n)], dtype=torch.float64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        all_bases, all_reads = float(tb[0].item()), float(tb[1].item())
    else:
        all_bases, all_reads = float(total_bases), float(n)
    if strong:  # the kernel's own launch time for the roofline: one resident launch of this rank's shard, outside the timed region
        pipe.close()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ix.map_batch_device(d_bases.data_ptr(), d_offs.data_ptr(), n, total_bases, d_out.data_ptr(), stream.cuda_stream)
        e0.record(stream)
        ix.map_batch_device(d_bases.data_ptr(), d_offs.data_ptr(), n, total_bases, d_out.data_ptr(), stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        kern_ms = [e0.elapsed_time(e1)]
    else:
        kern_ms = [a.elapsed_time(b) for a, b in ev]
    avg_kern_s = float(np.mean(kern_ms)) / 1e3

    hits = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=mq.hit_dtype)
    n_kmm = int(hits["n_kminmers"].astype(np.int64).sum())
    n_mapped = int((hits["status"] == 1).sum())
    n_over = int((hits["status"] == 2).sum())

    # ---- roofline of the dominant (only) kernel of a step: map_kernel
    # algorithmic bytes per launch = SURVEY.md 8(d): L*b_in + n_kmm*S_slot*p_bar + S_out per read, with b_in = 1 B (ASCII in
    # HBM), S_slot = 32 B, p_bar = mean probes per lookup MEASURED on this batch by one instrumented launch outside the timed
    # region, S_out = 40 B (+ 8 B offset)
    d_out2 = torch.zeros_like(d_out)
    lookups, extra = ix.probe_stats(d_bases.data_ptr(), d_offs.data_ptr(), n, total_bases, d_out2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(d_out, d_out2), "instrumented launch disagrees with the timed one"
    p_bar = 1.0 + extra / max(lookups, 1)
    alg_bytes = total_bases * 1 + n_kmm * st["slot_bytes"] * p_bar + n * (8 + 40)  # SURVEY 8(d)'s S_out = 40 B (the result record is 48 B since ABI 3)
    achieved = alg_bytes / avg_kern_s / 1e9
    traffic = None
    traffic_commit = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("reads") == n and abs(tj.get("genome_scale", -1) - args.genome_scale) < 1e-9:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_commit = tj.get("commit")
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=8000.0, unit="GB/s", frac=round(achieved / 8000.0, 4),
                    traffic=traffic, traffic_taken_at_commit=traffic_commit, kernel="map_kernel", avg_launch_ms=round(avg_kern_s * 1e3, 4),
                    algorithmic_bytes_per_launch=int(alg_bytes), mean_probes_per_lookup=round(p_bar, 4))

    # ---- accuracy on the whole batch (BASELINE metric: "Q60 mapeval parity"): paftools-mapeval-style counts
    truth = {k: v for k, v in reads.items() if k not in ("bases", "offsets")}
    pafs = {"mapped": (hits["status"] == 1).astype(np.uint32)}
    for a_ in ("ref_id", "rc", "mapq", "r_start", "r_end"):
        pafs[a_] = hits[a_]
    n_m, n_q60, n_q60_wrong = sim.mapeval(truth, pafs)

    # ---- CPU baseline: the C oracle ("port") on a bounded sample of the same reads, rank 0 at N=1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        po = O.params(k=args.k, l=args.l, density=args.density)
        t0 = time.time()
        ox = O.Index()
        ox.build_mt(genome, ctg_off, ctg_names, po, ncpu)
        t_cpu_index = time.time() - t0
        ns = args.cpu_sample_reads or min(n, 49152)
        sb = reads["bases"][:int(offs[ns])]
        so = offs[:ns + 1]

        def timed(nthreads, min_wall_s=5.0, min_passes=3):
            """>= min_passes passes and >= min_wall_s of wall time at this thread count: (median s/pass, min, max, passes, last result)."""
            ts, w = [], None
            ox.map_batch(sb, so, po, threads=nthreads)  # untimed: page in, thread start-up
            while len(ts) < min_passes or sum(ts) < min_wall_s:
                t0 = time.perf_counter()
                w = ox.map_batch(sb, so, po, threads=nthreads)
                ts.append(time.perf_counter() - t0)
                if len(ts) >= 60:
                    break
            return float(np.median(ts)), min(ts), max(ts), len(ts), w

        # all granted cores, and 10 threads (the reference's own benchmark setting, experiments/figure-k-l/get_mapstats.sh:6)
        t_cpu, t_lo, t_hi, reps, want = timed(ncpu)
        t10, t10_lo, t10_hi, reps10, _ = timed(10)
        m = want["mapped"] != 0
        same = bool(np.array_equal(hits["status"][:ns] == 1, m)) and all(
            np.array_equal(mq.hit_column(hits[:ns], a)[m], want[a][m].astype(np.uint64))
            for a in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"))
        sb_bases = int(so[-1])
        cpu = dict(value=round(sb_bases / t_cpu / 1e9, 4), unit="Gbases/s", cores=ncpu, kind="port",
                   sample="first %d reads (%d bases) of the step batch, C oracle with %d pthreads (cgroup CPU quota of the box), median of %d "
                          "passes (%.1f s), index build (%.1f s) excluded" % (ns, sb_bases, ncpu, reps, t_cpu * reps, t_cpu_index),
                   passes=reps, spread=dict(best=round(sb_bases / t_lo / 1e9, 4), worst=round(sb_bases / t_hi / 1e9, 4)),
                   seconds=round(t_cpu * reps, 2), paf_columns_identical_to_gpu=same, unique_kminmers_equal=bool(ox.count() == n_unique),
                   at_10_threads=dict(value=round(sb_bases / t10 / 1e9, 4), threads=10, passes=reps10,
                                      spread=dict(best=round(sb_bases / t10_lo / 1e9, 4), worst=round(sb_bases / t10_hi / 1e9, 4)),
                                      note="median; 10 pthreads on %d granted cores; mirrors --threads 10 of experiments/figure-k-l/get_mapstats.sh:6" % ncpu),
                   published=dict(value=1.56, unit="Gbases/s", threads=10, seconds=19.98,
                                  source="experiments/figure-k-l/k_perf.csv:5 (k=5: 19.98 s for CHM13 10X, ~31.2 Gbases; the reference's own "
                                         "run on its authors' machine, Rust path, not measured here)"))
        del ox

    # ---- end to end (rank 0, N=1): host buffers -> hits through three stream slots, and FASTA files -> PAF through the native driver
    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e:
        import tempfile
        e2e = {}
        gb, h_e2e = measure_host_buffers(mq, ix, reads)
        e2e["host_buffers_gbases_s"] = round(gb, 2)
        e2e["host_buffers_hits_identical"] = bool(np.array_equal(h_e2e.view(np.uint8), hits.view(np.uint8)))
        e2e["host_buffers_note"] = "page-locked reads -> mq_ctx_submit/wait on 3 stream slots, 16,384-read sub-batches -> hits in host memory; PCIe-bound at 1 B/base"
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        kargs = ["-k", str(args.k), "-l", str(args.l), "-d", repr(args.density)]
        with tempfile.TemporaryDirectory(dir=base) as wd:
            ref = os.path.join(wd, "ref.fa")
            write_reference(genome, ctg_off, ctg_names, ref)

            def leg(name, **kw):
                try:
                    e2e[name] = measure_file_to_paf(ref, **kw)
                except Exception as ex:  # noqa: BLE001
                    e2e[name] = {"error": repr(ex)[:300]}

            # the step batch as a FASTA file, at the bench's own parameters
            leg("file_to_paf", reads=reads, n_reads=args.e2e_file_reads, threads=ncpu, workdir=wd, extra_args=kargs)
            e2e["file_to_paf_gbases_s"] = e2e["file_to_paf"].get("gbases_s")
            # a plain gzip stream of a quarter of it (get_reader's .gz branch, src/main.rs:60-75)
            leg("gz_to_paf", reads=reads, n_reads=max(1, args.e2e_file_reads // 4), threads=ncpu, workdir=wd, extra_args=kargs, compress="gz")
            try:
                e2e["index_file"] = measure_index_file(mq, ix, P, local_rank, wd)
            except Exception as ex:  # noqa: BLE001
                e2e["index_file"] = {"error": repr(ex)[:300]}
            # BASELINE config 4's shape (experiments/table1.sh:50-55): uncompressed FASTQ reads, -k 7 -l 31 -d 0.01, and the same
            # reads as FASTA for the ratio
            if args.e2e_fastq_reads > 0:
                try:
                    big = reads if args.e2e_fastq_reads <= n else sim.make_reads(genome, ctg_off, args.e2e_fastq_reads, seed=args.seed + 5000, threads=threads)
                    k7 = ["-k", "7", "-l", "31", "-d", "0.01"]
                    leg("fastq_k7_to_paf", reads=big, n_reads=args.e2e_fastq_reads, threads=ncpu, workdir=wd, fastq=True, extra_args=k7)
                    leg("fasta_k7_to_paf", reads=big, n_reads=args.e2e_fastq_reads, threads=ncpu, workdir=wd, fastq=False, extra_args=k7)
                    a_, b_ = e2e["fastq_k7_to_paf"].get("no_prefetch_gbases_s"), e2e["fasta_k7_to_paf"].get("no_prefetch_gbases_s")
                    e2e["fastq_over_fasta_rate"] = round(a_ / b_, 3) if a_ and b_ else None
                    del big
                except Exception as ex:  # noqa: BLE001
                    e2e["fastq_k7_to_paf"] = {"error": repr(ex)[:300]}

    if rank == 0:
        value = all_bases * args.steps / elapsed / 1e9
        line = {
            "metric": "Gbases/s mapped (sim CHM13v2-like HiFi, k=%d l=%d d=%g)" % (args.k, args.l, args.density),
            "value": round(value, 3),
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "genome_preset": args.genome_preset,
                "workload": "CHM13v2.0-like synthetic genome (25 contigs, %.3f Gbp, scale %.3g, %g%% planted repeats) "
                            "x pbsim-like HiFi reads (mean 24 kb, 1%% error); k=%d l=%d d=%g HPC"
                            % (sum(lens) / 1e9, args.genome_scale, 100 * args.repeat_frac, args.k, args.l, args.density),
                "reads_per_step_per_gpu": n,
                "bases_per_step_per_gpu": total_bases,
                "index_unique_kminmers": int(n_unique),
                "index_table_bytes": int(st["table_bytes"]),
                "parallelism": ("one fixed read set in host memory dealt to %d GPU(s) (contiguous shards), index replicated, no data-path "
                                "collective; host buffers -> stream slots -> hits, PCIe included" if strong else
                                "reads sharded over %d GPU(s) (every rank its own HBM-resident batch), index replicated, no data-path collective") % world,
            },
            "mreads_per_s": round(all_reads * args.steps / elapsed / 1e6, 4),
            "mapped_frac": round(n_mapped / max(n, 1), 4),
            "overflow_reads": n_over,
            "kminmers_per_step": n_kmm,
            "setup_s": {"genome": round(t_genome, 1), "gpu_index": round(t_index, 2), "reads": round(t_reads, 1)},
            "q60": n_q60,
            "q60_wrong": n_q60_wrong,
            "mapped_reads": n_m,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "end_to_end": e2e,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
