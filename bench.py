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
    ap.add_argument("--reads", type=int, default=1572864,
                    help="reads per step per GPU: 1,572,864 HiFi reads = 37 Gbases = 12x of the 30x of BASELINE config 3, resident in HBM (a launch "
                         "carries 0.15-0.3 ms of start-up + tail -- its waves start in step and finish one read apart -- which is 7 %% of a "
                         "196,608-read launch and 1 %% of this one: the line's `smaller_batches` gives the rates of 49,152 / 196,608 / 786,432 reads)")
    ap.add_argument("--genome-scale", type=float, default=1.0, help="1.0 = CHM13-like 3.117 Gbp")
    ap.add_argument("--seed", type=int, default=2013)
    ap.add_argument("--repeat-frac", type=float, default=0.05, help="fraction of the genome overwritten by copied 1-20 kb segments "
                    "(0.8 = maize-like stress: most k-min-mers are tombstoned, lookups miss, Matches are short)")
    ap.add_argument("--tandem-frac", type=float, default=0.01)
    ap.add_argument("--repeat-div", type=float, default=0.01, help="per-base divergence of the planted copies")
    ap.add_argument("--genome-preset", choices=("planted-repeats", "human-like", "maize-like"), default="planted-repeats",
                    help="planted-repeats (default, the BASELINE stand-in of rounds 1-3): --repeat-frac / --tandem-frac / --repeat-div; human-like: "
                         "tools/sim.py HUMAN_LIKE (6 %% satellite arrays, 5 %% segmental duplications, young interspersed copies) -- reads from "
                         "inside satellites and recent duplications do not map uniquely, as on a real genome; maize-like: the maize-B73-shaped "
                         "genome of the `configs.maize_like` leg (BASELINE config 5) as the step's workload")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the BASELINE metric): every rank maps its own HBM-resident batch of --reads reads; strong: ONE fixed "
                         "read set of --reads reads in host memory is dealt to the ranks (mapquik_amd.shard) and mapped through the "
                         "host-buffer stream slots, PCIe included")
    ap.add_argument("--reference-fasta", default=os.environ.get("MQ_BENCH_REFERENCE") or None,
                    help="a real reference FASTA (plain or .gz; also MQ_BENCH_REFERENCE), e.g. chm13v2.0.fa (BASELINE config 3, "
                         "experiments/simulate_chm13.sh:12): indexed instead of the synthetic genome; the JSON says data: real")
    ap.add_argument("--reads-fastx", default=os.environ.get("MQ_BENCH_READS") or None,
                    help="real reads, FASTA or FASTQ by the reference's naming rule (src/main.rs:196-205), plain or .gz (also MQ_BENCH_READS), "
                         "e.g. pbsim2fq output or the DeepConsensus HG002 FASTQ (BASELINE config 4, experiments/table1.sh:50): rank r maps reads "
                         "r*--reads .. (r+1)*--reads-1 of the file from HBM; without it reads are simulated from the given reference")
    ap.add_argument("--k", type=int, default=5, help="k-min-mer order (BASELINE config 4, experiments/table1.sh:50, runs -k 7)")
    ap.add_argument("--l", type=int, default=31)
    ap.add_argument("--density", type=float, default=0.01)
    ap.add_argument("--seeding-variant", type=int, default=0,
                    help="reading of the third-party k-min-mer iterator's unpinned decisions (mq_params.flags bits 8..13, include/mapquik_hip.h; "
                         "0 = the frozen reading, the BASELINE configuration): the oracle is switched to the same variant for the column checks")
    ap.add_argument("--fast-kh", action="store_true",
                    help="MQ_FLAG_FAST_KH for the step itself (profiling the opt-in tuple hash; the line then says so in `config.fast_kh` and in `metric`: "
                         "not the BASELINE configuration, whose tuple hash is SipHash-1-3); the oracle is switched to its bit 64 for the column checks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end measurements (host buffers -> hits, file -> PAF)")
    ap.add_argument("--e2e-file-reads", type=int, default=196608, help="reads written to the FASTA the native driver maps")
    ap.add_argument("--e2e-fastq-reads", type=int, default=786432,
                    help="reads of the FASTQ + -k 7 leg (BASELINE config 4's shape, experiments/table1.sh:50-55): 786,432 reads = 18.5 Gbases; 0 = skip")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    ap.add_argument("--no-smaller-batches", action="store_true", help="skip the `smaller_batches` launches (profiling runs: a profiler's per-kernel "
                    "averages then cover launches of the step batch only)")
    ap.add_argument("--no-configs", action="store_true", help="skip the kernel-only legs of the other configurations (human-like, maize-like, -k 7)")
    ap.add_argument("--config-reads", type=int, default=0, help="reads per step of the kernel-only legs (0 = --reads)")
    ap.add_argument("--config-sample-reads", type=int, default=8192, help="reads of each leg that the C oracle maps for the column-by-column check")
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


def files_on(path):
    """Where the e2e legs' files live: /dev/shm is tmpfs (page cache, no device behind it) -- the rates say what the host side and the
    link do, not what a disk does."""
    p = os.path.realpath(path)
    return "/dev/shm (tmpfs: memory, no block device)" if p.startswith("/dev/shm") else "%s (temporary directory, files read once before the timed run: page cache)" % os.path.dirname(p)


def measure_file_to_paf(ref, reads, n_reads, threads, workdir, fastq=False, extra_args=(), compress=None, env=None, more_threads=(), prefetch_run=True):
    """End to end through the native driver (mapquik_amd/lib/mapquik): reference FASTA + reads file on disk -> <prefix>.paf.
    fastq: the reads as a FASTQ file; compress="gz": as a plain gzip stream.  One run to bring the files into the page cache, then the
    driver's default = the strict run (nothing of the reads is touched before the index is ready; also reported under the
    no_prefetch_* keys of earlier rounds) THREE times -- the leg's numbers are the run with the median map phase, min / max beside
    them (a single shot on a shared box is not a measurement) -- and one run with MQ_DRIVER_PREFETCH=1 (the read feeder starts while
    the reference is indexed).  Rates over the driver's own 'Mapped query sequences' phase."""
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
           "args": " ".join(extra_args), "files_on": files_on(workdir)}
    prefix = os.path.join(workdir, "e2e")

    base_env = dict(os.environ, **(env or {}))

    def run(env, threads=threads):
        t0 = time.perf_counter()
        r = subprocess.run([exe, rd, "--reference", ref, "-p", prefix, "--threads", str(threads)] + list(extra_args),
                           capture_output=True, text=True, timeout=900, env=dict(base_env, **env))
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError((r.stderr or r.stdout)[-300:])
        m = re.search(r"Mapped query sequences in ([0-9.]+)(s|ms|\u00b5s|ns)", r.stdout)
        unit = {"s": 1.0, "ms": 1e-3, "\u00b5s": 1e-6, "ns": 1e-9}
        t_map = float(m.group(1)) * unit.get(m.group(2), 1.0) if m else float("nan")
        mi = re.search(r"Indexed [0-9]+ unique k-min-mers in ([0-9.]+)(s|ms|\u00b5s|ns)", r.stdout)
        t_idx = float(mi.group(1)) * unit.get(mi.group(2), 1.0) if mi else float("nan")
        return t_map, t_idx, wall

    def median_of(env, threads=threads, reps=3):
        """`reps` runs of the driver: the run with the median map phase, and min / max of the map-phase rate and of the job's wall time"""
        rs = sorted((run(env, threads=threads) for _ in range(reps)), key=lambda r: r[0])
        t_map, t_idx, wall = rs[len(rs) // 2]
        return t_map, t_idx, wall, dict(runs=reps, gbases_s_min=round(bases / rs[-1][0] / 1e9, 3), gbases_s_max=round(bases / rs[0][0] / 1e9, 3),
                                        driver_wall_s_min=round(min(r[2] for r in rs), 2), driver_wall_s_max=round(max(r[2] for r in rs), 2))

    try:
        run({})  # first run: files enter the page cache
        t_map, t_idx, wall, spread = median_of({})
        out.update(gbases_s=round(bases / t_map / 1e9, 3), map_phase_s=round(t_map, 4), index_phase_s=round(t_idx, 3), driver_wall_s=round(wall, 2),
                   whole_job_gbases_s=round(bases / wall / 1e9, 3), spread=spread,
                   note="median of %d runs of the driver (the run with the median map phase); spread = min / max over the runs" % spread["runs"])
        out.update(no_prefetch_gbases_s=out["gbases_s"], no_prefetch_map_phase_s=out["map_phase_s"], no_prefetch_index_phase_s=out["index_phase_s"],
                   no_prefetch_driver_wall_s=out["driver_wall_s"])
        if prefetch_run:
            t_map2, t_idx2, wall2 = run({"MQ_DRIVER_PREFETCH": "1"})
            out.update(prefetch_gbases_s=round(bases / t_map2 / 1e9, 3), prefetch_map_phase_s=round(t_map2, 4), prefetch_index_phase_s=round(t_idx2, 3),
                       prefetch_driver_wall_s=round(wall2, 2))
        with open(prefix + ".paf", "rb") as f:
            out["paf_lines"] = sum(1 for _ in f)
        for t in more_threads:  # the same file at other thread counts (strict run), medians too
            tm, ti, w, sp = median_of({}, threads=t)
            out["at_%d_threads" % t] = dict(gbases_s=round(bases / tm / 1e9, 3), map_phase_s=round(tm, 4), index_phase_s=round(ti, 3), driver_wall_s=round(w, 2),
                                            whole_job_gbases_s=round(bases / w / 1e9, 3), spread=sp)
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


def expected_kminmers(n_bases, P):
    """What Index::new sizes its map for (src/index.rs:83 hardcodes CHM13's 39,821,990): canonical selection keeps 1 - (1 - d)^2 of the
    l-mers, homopolymer compression about three quarters of the bases."""
    d = min(1.0, max(0.0, P.density))
    return int(n_bases * (1.0 - (1.0 - d) ** 2) * (0.75 if P.use_hpc else 1.0)) + 1


def build_index_device(mq, torch, dev, local_rank, P, genome, ctg_off, ctg_names, ix=None):
    """Reference contigs to the device first (one copy of the genome, timed by itself), then the index build proper from
    device-resident contigs: mq_index_add_ref_device per contig + mq_index_finalize (mers::ref_extract + Index::add_with_mer +
    into_read_only, src/mers.rs:15-38, src/index.rs:94-116).  Returns (index, per-contig k-min-mer counts, unique, upload s, build s)."""
    t0 = time.time()
    d_genome = torch.from_numpy(genome).to(dev)
    torch.cuda.synchronize()
    t_up = time.time() - t0
    t0 = time.time()
    if ix is None:  # (the main configuration creates its index -- and reserves its table -- before the genome exists)
        ix = mq.Index(P, device=local_rank)
        ix.reserve_table(expected_kminmers(genome.size, P))
    per_ref = []
    for r in range(len(ctg_names)):
        a, b = int(ctg_off[r]), int(ctg_off[r + 1])
        per_ref.append(ix.add_ref_device(r, ctg_names[r], d_genome.data_ptr() + a, b - a))
    n_unique = ix.finalize()
    torch.cuda.synchronize()
    t_build = time.time() - t0
    del d_genome
    return ix, per_ref, n_unique, t_up, t_build


def columns_equal(mq, hits, want):
    """status and every numeric PAF column of the GPU's hits against the oracle's records of the same reads"""
    m = want["mapped"] != 0
    return bool(np.array_equal(hits["status"] == 1, m)) and all(
        np.array_equal(mq.hit_column(hits, a)[m], want[a][m].astype(np.uint64))
        for a in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"))


def oracle_sample_check(mq, O, genome, ctg_off, ctg_names, po, ncpu, B, hits, ns, variant=0):
    """The batch's first ns reads and its strided sample through the C oracle (index built with all granted threads, the oracle switched
    to the leg's seeding variant): are status and every numeric PAF column of the GPU's hits the oracle's?
    Returns (identical, oracle unique k-min-mers, reads compared)."""
    O.lib().mqo_set_variant(variant)
    try:
        ox = O.Index()
        ox.build_mt(genome, ctg_off, ctg_names, po, ncpu)
        ns = min(ns, B.keep_first)
        rd = host_reads(B, ns)
        same = columns_equal(mq, hits[:ns], ox.map_batch(rd["bases"], rd["offsets"], po, threads=ncpu))
        n_cmp = ns
        idx = B.strided_idx[B.strided_idx < hits.size]  # (a leg on the batch's first reads only: the strided reads among them)
        if idx.size:
            so = B.strided_offsets[:idx.size + 1]
            same = same and columns_equal(mq, hits[idx], ox.map_batch(B.strided_bases[:int(so[-1])], so, po, threads=ncpu))
            n_cmp += int(idx.size)
        uniq = int(ox.count())
        del ox
    finally:
        O.lib().mqo_set_variant(0)
    return same, uniq, n_cmp


def poison(d_out):
    """0xFF in every byte of the result buffer: a read whose record no wave writes keeps status 0xFFFFFFFF -- not 0 = 'unmapped'."""
    d_out.fill_(0xFF)


def records_written(torch, d_out, n):
    """How many of the n result records carry a status a wave wrote (0 unmapped, 1 mapped, 2 overflow): n, or a read was lost."""
    st = d_out.view(torch.int32).view(n, -1)[:, 0]
    return int(((st >= 0) & (st <= 2)).sum().item())


def kernel_leg(mq, sim, O, torch, dev, local_rank, P, po, genome, ctg_off, ctg_names, B, steps, warmup, ncpu, sample_reads, workload, n_reads=None, variant=0,
               same_as=None):
    """One kernel-only leg of another configuration: index on the GPU, the batch resident in HBM, `steps` timed launches of
    map_kernel on a poisoned result buffer, mapeval counts over the batch, the oracle's columns on a sample."""
    ix, _, n_unique, _, t_build = build_index_device(mq, torch, dev, local_rank, P, genome, ctg_off, ctg_names)
    n = B.n if n_reads is None else min(n_reads, B.n)
    offs = B.offsets[:n + 1]
    total_bases = int(offs[-1])
    d_out = torch.empty(n * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
    poison(d_out)
    ix.reserve(n, total_bases)
    stream = torch.cuda.current_stream(dev)
    for _ in range(warmup):
        ix.map_batch_device(B.d_bases.data_ptr(), B.d_offs.data_ptr(), n, total_bases, d_out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        ix.map_batch_device(B.d_bases.data_ptr(), B.d_offs.data_ptr(), n, total_bases, d_out.data_ptr(), stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    n_written = records_written(torch, d_out, n)
    hits = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=mq.hit_dtype)
    truth = {k: v[:n] for k, v in B.truth.items()}
    pafs = {"mapped": (hits["status"] == 1).astype(np.uint32)}
    for a_ in ("ref_id", "rc", "mapq", "r_start", "r_end"):
        pafs[a_] = hits[a_]
    n_m, n_q60, n_q60_wrong = sim.mapeval(truth, pafs)
    n_fast, n_general = ix.last_map_path_counts()
    n_flagged, n_first = ix.last_map_order()
    same, ouniq, n_cmp = oracle_sample_check(mq, O, genome, ctg_off, ctg_names, po, ncpu, B, hits, min(sample_reads, n), variant=variant)
    st = ix.stats()
    ix.close()
    del d_out
    extra = {} if same_as is None else {"hits_byte_identical_to_the_step": bool(hits.tobytes() == same_as[:n].tobytes())}
    return dict(extra, workload=workload, value=round(total_bases / (ms * 1e-3) / 1e9, 3), unit="Gbases/s", ms_per_launch=round(ms, 4), steps=steps,
                reads_per_step=n, bases_per_step=total_bases, records_written=n_written, kminmers_per_step=int(hits["n_kminmers"].astype(np.int64).sum()),
                mapped_frac=round(n_m / max(n, 1), 4), q60=n_q60, q60_wrong=n_q60_wrong, overflow_reads=int((hits["status"] == 2).sum()), reads_first=n_first,
                general_path_reads=int(n_general), index_unique_kminmers=int(n_unique), index_keys=int(st["n_keys"]), index_build_s=round(t_build, 3),
                paf_columns_identical_to_oracle=same, oracle_sample_reads=n_cmp, unique_kminmers_equal_oracle=bool(ouniq == n_unique), seeding_variant=variant)


class Batch:
    """A batch of reads resident in HBM: device tensors (bases, offsets), the host's offsets and truth columns, and the first
    `keep_first` reads (plus every `keep_stride`-th read) in host memory for the CPU-side checks and the file legs."""
    pass


def load_batch(torch, dev, sim, genome, ctg_off, n_reads, seed, threads, keep_first=0, keep_stride=0, stride_from=0, slice_reads=32768):
    """tools/sim.py's reads [0, n_reads) of `seed`, synthesised slice by slice (<= 0.9 GB of page-locked work buffer, two of them) and
    copied straight into the device batch while the next slice is made: no 39-GB capacity layout and no 37-GB host copy per rank
    (the batch a rank maps exists only in HBM).  Same reads, byte for byte, as sim.make_reads(genome, ctg_off, n_reads, seed)."""
    ctg_off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
    caps = np.zeros(n_reads + 1, dtype=np.uint64)
    sim.lib().mqsim_read_caps(ctg_off.ctypes.data, ctg_off.size - 1, n_reads, 24000.0, 2300.0, 100, 25000, seed, caps.ctypes.data)
    d_buf = torch.empty(int(caps[-1]) + 64, dtype=torch.uint8, device=dev)  # an upper bound (template + room for insertions); the batch is a view of it
    del caps
    pinned = {}

    def alloc(nbytes):
        t = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        a = t.numpy()
        pinned[a.ctypes.data] = t
        return a

    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    truth = dict(ctg=np.zeros(n_reads, dtype=np.uint32), start=np.zeros(n_reads, dtype=np.uint64), end=np.zeros(n_reads, dtype=np.uint64),
                 strand=np.zeros(n_reads, dtype=np.uint8))
    head, strided, strided_idx = [], [], []
    copy_stream = torch.cuda.Stream(device=dev)
    done = [None, None]
    at = 0
    for k, (r0, r1, b, o, t) in enumerate(sim.read_slices(genome, ctg_off, n_reads, seed=seed, slice_reads=slice_reads, threads=threads, buffers=alloc)):
        src = pinned[b.ctypes.data][:b.size]
        with torch.cuda.stream(copy_stream):
            d_buf[at:at + b.size].copy_(src, non_blocking=True)
            done[k & 1] = torch.cuda.Event()
            done[k & 1].record(copy_stream)
        offsets[r0 + 1:r1 + 1] = o[1:] + np.uint64(at)
        for key in truth:
            truth[key][r0:r1] = t[key]
        if r0 < keep_first:
            m = min(r1, keep_first) - r0
            head.append(b[:int(o[m])].copy())
        if keep_stride:
            for r in range(-(-r0 // keep_stride) * keep_stride, r1, keep_stride):
                if r >= stride_from:
                    strided.append(b[int(o[r - r0]):int(o[r - r0 + 1])].copy())
                    strided_idx.append(r)
        at += b.size
        if done[(k + 1) & 1] is not None:
            done[(k + 1) & 1].synchronize()  # the other work buffer (the next slice's) has left for the device
    copy_stream.synchronize()
    B = Batch()
    B.n, B.total_bases = n_reads, at
    B.d_buf = d_buf
    B.d_bases = d_buf[:at]
    B.offsets = offsets
    B.d_offs = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    B.truth = truth
    B.keep_first = min(keep_first, n_reads)
    B.head_bases = np.concatenate(head) if head else np.zeros(0, dtype=np.uint8)
    B.strided_idx = np.asarray(strided_idx, dtype=np.int64)
    B.strided_bases = np.concatenate(strided) if strided else np.zeros(0, dtype=np.uint8)
    B.strided_offsets = np.zeros(len(strided) + 1, dtype=np.uint64)
    if strided:
        B.strided_offsets[1:] = np.cumsum([x.size for x in strided])
    return B


def batch_from_host(torch, dev, reads, keep_first=None):
    """A batch whose reads exist in host memory (real reads from a file; a strong-scaling shard): uploaded whole."""
    B = Batch()
    offs = reads["offsets"]
    B.n, B.total_bases = offs.size - 1, int(offs[-1])
    B.d_buf = B.d_bases = torch.from_numpy(reads["bases"]).to(dev)
    B.offsets = offs
    B.d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    B.truth = {k: v for k, v in reads.items() if k not in ("bases", "offsets")}
    B.keep_first = B.n if keep_first is None else min(keep_first, B.n)
    B.head_bases = reads["bases"][:int(offs[B.keep_first])]
    B.strided_idx = np.zeros(0, dtype=np.int64)
    B.strided_bases = np.zeros(0, dtype=np.uint8)
    B.strided_offsets = np.zeros(1, dtype=np.uint64)
    return B


def host_reads(B, m):
    """The batch's first m reads as host arrays (m <= B.keep_first)."""
    assert m <= B.keep_first, "bench.py keeps only the first %d reads of a batch in host memory (asked for %d)" % (B.keep_first, m)
    d = {k: v[:m] for k, v in B.truth.items()}
    d["offsets"] = B.offsets[:m + 1]
    d["bases"] = B.head_bases[:int(B.offsets[m])]
    return d


def peak_rss_gb():
    import resource
    return round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0, 2)  # ru_maxrss is in KB on Linux


def shared_genome(make, dist, world, rank, tag):
    """The synthetic genome once per node: rank 0 synthesises it (with every granted thread) into /dev/shm, the other ranks map the
    file (copy-on-write pages: shared until written, and nobody writes) -- not 3.1 GB and a synthesis per rank.  Where no shared
    directory takes the file (a container with a 64-MB /dev/shm and no room in /tmp) every rank synthesises its own, as before round 6."""
    if world == 1:
        return make()
    import tempfile
    box = [None]
    g = off = None
    if rank == 0:
        g, off, names = make()
        for d in ("/dev/shm", tempfile.gettempdir()):
            path = os.path.join(d, "mq_bench_%s_%d_%s" % (os.environ.get("MASTER_PORT", "0"), os.getpid(), tag))
            try:
                st = os.statvfs(d)
                if st.f_bavail * st.f_frsize < g.nbytes + (64 << 20):
                    continue
                np.save(path + ".g.npy", g)
                np.save(path + ".off.npy", off)
                box[0] = path
                break
            except OSError:
                for sfx in (".g.npy", ".off.npy"):
                    try:
                        os.remove(path + sfx)
                    except OSError:
                        pass
    dist.broadcast_object_list(box, src=0)
    path = box[0]
    if path is None:  # no shared file: every rank its own copy
        if rank != 0:
            g, off, names = make()
        return g, off, ["chr%d" % (i + 1) for i in range(off.size - 1)]
    if rank != 0:
        off = np.load(path + ".off.npy")
        g = np.load(path + ".g.npy", mmap_mode="c")
    names = ["chr%d" % (i + 1) for i in range(off.size - 1)]
    dist.barrier()
    if rank == 0:  # every rank has the file mapped: the name can go (the pages live as long as the mappings)
        for sfx in (".g.npy", ".off.npy"):
            try:
                os.remove(path + sfx)
            except OSError:
                pass
    return g, off, names


def fake_ranks():
    """Test hook (tests/test_bench_ranks.py): MQ_BENCH_FAKE_RANKS=1 runs every rank on device 0 with the gloo backend and the two
    all-reduces on CPU tensors, so that the world > 1 branch executes on a one-GPU box."""
    return os.environ.get("MQ_BENCH_FAKE_RANKS", "") not in ("", "0")


def launch_ranks(n_gpus):
    """`python bench.py --gpus N` typed by itself (no WORLD_SIZE in the environment): start the N ranks as a FRESH child process --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <the same arguments>` -- relay its output and return
    its exit code.  Nothing here has imported torch or touched the GPU, and the child is started with subprocess, never os.exec*
    (a process that has initialised the GPU must not replace itself).  Rank 0's JSON line stays the only `{..."metric"...}` line."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL / cross-process device memory needs dmabuf IPC on this platform
    env.setdefault("OMP_NUM_THREADS", str(max(1, effective_cpus() // n_gpus)))
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: starting the ranks with %s\n" % (n_gpus, " ".join(cmd[1:10])))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    t_start = time.time()
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.reads_fastx and not args.reference_fasta:
        raise SystemExit("--reads-fastx needs --reference-fasta (reads of another genome would not map)")
    for pth in (args.reference_fasta, args.reads_fastx):
        if pth and not os.path.exists(pth):
            raise SystemExit("no such file: %s" % pth)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # the driver's command shape for N = 1, typed with N > 1
        sys.exit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    fake = fake_ranks()
    n_dev = torch.cuda.device_count()  # does not initialise the GPU
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU: the mapquik HIP path has no CPU fallback")
    if world > n_dev and not fake:
        raise SystemExit("--gpus %d but this node shows %d GPU(s): one rank per GPU (MQ_BENCH_FAKE_RANKS=1 runs every rank on device 0, "
                         "a test hook, not a measurement)" % (world, n_dev))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the mapquik HIP path has no CPU fallback")
    if fake:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if fake else dev  # where the tensors of the two all-reduces live
    # MQ_BENCH_FORCE_DIST=1 (test hook): the process group and every collective of the N > 1 path also at N = 1 -- RCCL ("nccl") with device
    # tensors on the one GPU a test box has: init, barrier, MAX / SUM all-reduce, all-gather (tests/test_bench_ranks.py)
    use_dist = world > 1 or os.environ.get("MQ_BENCH_FORCE_DIST", "") not in ("", "0")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if fake:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import mapquik_amd as mq
    from tools import sim

    ncpu = effective_cpus()
    threads = max(1, ncpu // world)
    P = mq.Params(k=args.k, l=args.l, density=args.density, seeding_variant=args.seeding_variant, fast_kh=args.fast_kh)  # defaults k=5 l=31 d=0.01, HPC on, c=4 s=11 g=2000 (src/main.rs:174-188)
    if args.seeding_variant or args.fast_kh:  # the checker follows (a process-wide switch of the oracle; legs with their own Params are skipped below)
        from oracle import oracle as _O
        _O.lib().mqo_set_variant(args.seeding_variant | (64 if args.fast_kh else 0))
        args.no_configs = True
        args.no_e2e = True

    # ---- a real reference (and real reads) when given: BASELINE configs 3 / 4 the day the files are on the box
    real_ref = args.reference_fasta is not None
    real = None
    if real_ref:
        from tools import realdata
        t0 = time.time()
        real = realdata.load_reference(args.reference_fasta)

    t0 = time.time() if not real_ref else t0
    # ---- Index::new with its capacity (src/index.rs:78-83): the table is allocated and cleared in the background from here on
    MAIZE_KW = dict(family_frac=1.9, n_families=400, family_div=(0.005, 0.025), tandem_frac=0.02, n_runs=300)
    lens = ([int(x) for x in (real[1][1:] - real[1][:-1])] if real_ref else
            [max(40, int(x * args.genome_scale)) for x in (sim.MAIZE_LIKE if args.genome_preset == "maize-like" else sim.CHM13_LIKE)])
    ix0 = mq.Index(P, device=local_rank)
    ix0.reserve_table(expected_kminmers(sum(lens), P))

    # ---- genome (same on every rank: the index is replicated; synthesised once per node)
    if real_ref:
        genome, ctg_off, ctg_names = real
        del real
    elif args.genome_preset == "maize-like":
        genome, ctg_off, ctg_names = shared_genome(lambda: sim.make_genome(lens, seed=args.seed, threads=ncpu, **MAIZE_KW), dist, world, rank, "g")
    elif args.genome_preset == "human-like":
        genome, ctg_off, ctg_names = shared_genome(lambda: sim.make_genome(lens, seed=args.seed, threads=ncpu, **sim.HUMAN_LIKE), dist, world, rank, "g")
    else:
        genome, ctg_off, ctg_names = shared_genome(lambda: sim.make_genome(lens, seed=args.seed, threads=ncpu, repeat_frac=args.repeat_frac,
                                                                          tandem_frac=args.tandem_frac, div=args.repeat_div), dist, world, rank, "g")
    t_genome = time.time() - t0

    # ---- index on this rank's GPU (Index::add_with_mer + into_read_only on device), from device-resident contigs
    ix, per_ref, n_unique, t_upload, t_index = build_index_device(mq, torch, dev, local_rank, P, genome, ctg_off, ctg_names, ix=ix0)
    st = ix.stats()
    # algorithmic bytes of the build = L_ref * b_in + n_kmm * S_slot * 2 (every reference k-min-mer's slot written and read back)
    ib_bytes = int(sum(lens)) * 1 + int(sum(per_ref)) * st["slot_bytes"] * 2
    index_build = dict(ms=round(t_index * 1e3, 2), algorithmic_bytes=ib_bytes, unit="GB/s", achieved=round(ib_bytes / max(t_index, 1e-9) / 1e9, 2),
                       peak=8000.0, frac=round(ib_bytes / max(t_index, 1e-9) / 1e9 / 8000.0, 5), reference_bases=int(sum(lens)),
                       reference_kminmers=int(sum(per_ref)), unique_kminmers=int(n_unique),
                       note="wall time of mq_index_add_ref_device x %d contigs + mq_index_finalize from device-resident contigs, including whatever "
                            "was left to wait for of the %.1f GB table's allocation (requested at Index creation like the reference's "
                            "DashMap::with_capacity, src/index.rs:83; fresh device memory costs ~30 ms per GB here); the contigs' upload (%.2f s) "
                            "is not in it" % (len(lens), st["table_bytes"] / 1e9, t_upload))

    # what the table's allocation + clear cost: timed by the library around its own hipMalloc + memset + synchronize (the reservation's
    # background thread, mq_index_reserve: this process's first large allocation, i.e. FRESH device memory -- a second allocation of
    # the same size, by torch or by hipMalloc, gets recycled memory in 3-8 ms or fresh memory in 0.5 s at the runtime's discretion and
    # measures nothing).  The build above overlaps it with genome synthesis, which a real run does not have.
    try:
        t_tbl = ix.table_alloc_ms()
        index_build["table_alloc_and_clear_ms"] = round(t_tbl, 2)
        index_build["ms_with_table_allocation"] = round(t_index * 1e3 + t_tbl, 2)
        index_build["table_note"] = ("%.1f GB: the library's own clock around hipMalloc + hipMemsetAsync + synchronize of THIS index's table on "
                                     "mq_index_reserve's thread (the process's first large allocation: fresh device memory); `ms` holds only what was left of "
                                     "it to wait for" % (st["table_bytes"] / 1e9))
    except Exception as ex:  # noqa: BLE001
        index_build["table_alloc_and_clear_ms"] = None
        index_build["table_note"] = repr(ex)[:200]

    # ---- this rank's batch of reads, resident in HBM
    t0 = time.time()
    strong = args.scaling == "strong"
    solo = rank == 0 and world == 1
    # reads kept in host memory as well: what the CPU baseline, the oracle checks and the file legs read (N = 1 only); every 32nd read of
    # the rest for the strided oracle check.  A rank of a multi-GPU run keeps nothing: its batch exists in HBM only.
    keep_first = 0
    if solo:
        keep_first = max(args.cpu_sample_reads or 49152, args.config_sample_reads if not args.no_configs else 0)
        if not args.no_e2e:
            keep_first = max(keep_first, 393216, args.e2e_file_reads, args.e2e_fastq_reads)
    read_names = None
    have_truth = True
    if args.reads_fastx:  # real reads: weak scaling = every rank its own slice of the file; strong = one slice dealt to the ranks
        from mapquik_amd.shard import shard_bounds
        skip = 0 if strong else rank * args.reads
        rr = realdata.load_reads(args.reads_fastx, args.reads, skip=skip)
        if rr["offsets"].size - 1 == 0:
            raise SystemExit("%s holds no read for rank %d (reads %d..)" % (args.reads_fastx, rank, skip))
        if strong:
            lo, hi = shard_bounds(rr["offsets"].size - 1, world, rank)
            fo = rr["offsets"]
            rr = dict(bases=rr["bases"][int(fo[lo]):int(fo[hi])], offsets=(fo[lo:hi + 1] - fo[lo]).astype(np.uint64), names=rr["names"][lo:hi])
        read_names = rr.pop("names")
        truth_ = realdata.truth_from_names(read_names, ctg_names)  # pbsim2fq names carry it; real reads do not
        have_truth = truth_ is not None
        B = batch_from_host(torch, dev, dict(rr, **(truth_ or {})))
        del rr
    elif strong:  # the same read set on every rank; this rank's contiguous shard of it, in host memory (it is mapped from there)
        from mapquik_amd.shard import shard_bounds
        lo, hi = shard_bounds(args.reads, world, rank)
        parts, po_, tr_ = [], [np.zeros(1, dtype=np.uint64)], {}
        for r0, r1, b, o, t in sim.read_slices(genome, ctg_off, hi - lo, seed=args.seed + 1000, first_read=lo, threads=threads):
            parts.append(b.copy())
            po_.append(o[1:] + po_[-1][-1])
            for k_, v_ in t.items():
                tr_.setdefault(k_, []).append(v_)
        B = batch_from_host(torch, dev, dict(bases=np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8), offsets=np.concatenate(po_),
                                             **{k_: np.concatenate(v_) for k_, v_ in tr_.items()}))
        del parts
    else:
        B = load_batch(torch, dev, sim, genome, ctg_off, args.reads, args.seed + 1000 + rank, threads, keep_first=keep_first,
                       keep_stride=32 if solo else 0, stride_from=min(args.cpu_sample_reads or 49152, args.config_sample_reads))
    offs = B.offsets
    n = B.n
    total_bases = B.total_bases
    d_bases, d_offs = B.d_bases, B.d_offs
    d_out = torch.empty(n * mq.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
    poison(d_out)  # a record no wave writes stays 0xFF..: counted below (`records_written`), never mistaken for an unmapped read
    t_reads = time.time() - t0
    ix.reserve(n, total_bases)
    stream = torch.cuda.current_stream(dev)

    pipe = HostPipeline(mq, ix, dict(bases=B.head_bases, offsets=offs)) if strong else None

    def step():
        if strong:
            pipe.run_pass()
        else:
            ix.map_batch_device(d_bases.data_ptr(), d_offs.data_ptr(), n, total_bases, d_out.data_ptr(), stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t_first_step = time.time() - t_start
    rss_setup = peak_rss_gb()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        step()
        ev[i][1].record(stream)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([float(total_bases), float(n)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        all_bases, all_reads = float(tb[0].item()), float(tb[1].item())
        mine = torch.tensor([elapsed_local, float(total_bases), t_first_step, rss_setup], dtype=torch.float64, device=red_dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = [round(float(g[1].item()) * args.steps / float(g[0].item()) / 1e9, 3) for g in gathered]
        per_rank_setup = [round(float(g[2].item()), 1) for g in gathered]
        per_rank_rss = [round(float(g[3].item()), 2) for g in gathered]
    else:
        all_bases, all_reads = float(total_bases), float(n)
        per_rank = None
        per_rank_setup, per_rank_rss = [round(t_first_step, 1)], [rss_setup]
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
    # every read of the batch has a record that a wave wrote (the buffer was poisoned before the warm-up)
    n_written = records_written(torch, d_out, n)
    if n_written != n:
        raise SystemExit("bench.py: %d of %d result records were never written: reads were lost by the launch" % (n - n_written, n))
    d_step = d_out.clone()  # the timed steps' results: compared below with the instrumented launch and the oracle

    # the same kernel on the first m reads of the batch: what a launch's fixed cost (start-up + tail) does to smaller batches
    smaller = {}
    if not strong and not args.no_smaller_batches:
        for m in (49152, 196608, 786432):
            if m < n:
                tb = int(offs[m])
                # warm-up worth ~55 ms of launches (2 n / m of them): the checks above leave the GPU idle for a moment (torch loads its
                # comparison kernels on first use) and an idle GPU clocks down -- the first dozen milliseconds after a pause run 20 % slow
                # (tools/small_probe.py: 49,152 reads 1.11 ms per launch from idle, 0.92 ms once the clocks are up)
                for _ in range(max(2, min(64, 2 * n // m))):
                    ix.map_batch_device(d_bases.data_ptr(), d_offs.data_ptr(), m, tb, d_out.data_ptr(), stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(10):
                    ix.map_batch_device(d_bases.data_ptr(), d_offs.data_ptr(), m, tb, d_out.data_ptr(), stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
                ms_m = e0.elapsed_time(e1) / 10
                smaller[str(m)] = {"gbases_s": round(tb / ms_m / 1e6, 1), "ms_per_launch": round(ms_m, 4)}
        if smaller:  # the whole batch once more, on a poisoned buffer again
            poison(d_out)
            ix.map_batch_device(d_bases.data_ptr(), d_offs.data_ptr(), n, total_bases, d_out.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(d_out, d_step), "a launch of the same batch gave other results than the timed steps"
    del d_out
    d_out = d_step
    hits = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=mq.hit_dtype)
    launch_order = dict(zip(("reads_flagged", "reads_first"), ix.last_map_order()))  # order_reads_kernel: reads taken up first
    if os.environ.get("MQ_BENCH_DUMP_HITS"):  # test hook: this rank's hits (strong scaling: of its shard of the one read set)
        np.save(os.path.join(os.environ["MQ_BENCH_DUMP_HITS"], "hits_rank%d_of_%d.npy" % (rank, world)), hits.view(np.uint8))
    n_kmm = int(hits["n_kminmers"].astype(np.int64).sum())
    n_mapped = int((hits["status"] == 1).sum())
    n_over = int((hits["status"] == 2).sum())
    # reads long enough for extract() (src/mers.rs:44) that listed k minimizers or more and still came back unmapped: what a lost read
    # would look like had the buffer been cleared instead of poisoned -- reported so that two runs can be compared
    rl = (offs[1:] - offs[:-1]).astype(np.int64)
    n_unmapped_seeded = int(((hits["status"] == 0) & (rl >= args.l + args.k - 1) & (hits["n_kminmers"] >= 1)).sum())

    # ---- roofline of the dominant (only) kernel of a step: map_kernel
    # algorithmic bytes per launch = SURVEY.md 8(d): L*b_in + n_kmm*S_slot*p_bar + S_out per read, with b_in = 1 B (ASCII in
    # HBM), S_slot = 32 B, p_bar = mean probes per lookup MEASURED on this batch by one instrumented launch outside the timed
    # region, S_out = 40 B (+ 8 B offset)
    d_out2 = torch.empty_like(d_out)
    poison(d_out2)
    lookups, extra = ix.probe_stats(d_bases.data_ptr(), d_offs.data_ptr(), n, total_bases, d_out2.data_ptr())
    torch.cuda.synchronize()
    assert records_written(torch, d_out2, n) == n, "the instrumented launch left result records unwritten"
    assert torch.equal(d_out, d_out2), "instrumented launch disagrees with the timed one"  # byte for byte: status, columns, n_kminmers of every read
    del d_out2
    p_bar = 1.0 + extra / max(lookups, 1)
    alg_bytes = total_bases * 1 + n_kmm * st["slot_bytes"] * p_bar + n * (8 + 40)  # SURVEY 8(d)'s S_out = 40 B (the result record is 48 B since ABI 3)
    achieved = alg_bytes / avg_kern_s / 1e9
    traffic = None
    traffic_commit = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if (tj.get("reads") == n and abs(tj.get("genome_scale", -1) - args.genome_scale) < 1e-9 and not real_ref and args.genome_preset == "planted-repeats"
                    and args.k == 5 and args.l == 31 and not args.seeding_variant and not args.fast_kh):  # (the counters were taken on the headline workload)
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_commit = tj.get("commit")
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=8000.0, unit="GB/s", frac=round(achieved / 8000.0, 4),
                    traffic=traffic, traffic_taken_at_commit=traffic_commit, kernel="map_kernel", avg_launch_ms=round(avg_kern_s * 1e3, 4),
                    algorithmic_bytes_per_launch=int(alg_bytes), mean_probes_per_lookup=round(p_bar, 4))

    # SURVEY.md 8(d)'s second bound: vector-instruction issue.  wave-instructions per launch and the shader clock come from the PMC
    # passes of the same workload (profiles/pmc_issue.json, tools/issue_json.py); the launch time is this run's; the ceiling is
    # what a SIMD issues at map_kernel's occupancy (four waves per SIMD as two 8-wave workgroups per CU) on the instruction mix
    # of stage B in the microbenchmark (profiles/r03_valu_occ.txt, profiles/r04_valu_enc.txt, profiles/r04_valu_enc_stepb.txt).
    ipath = os.path.join(ROOT, "profiles", "pmc_issue.json")
    if os.path.exists(ipath):
        try:
            ij = json.load(open(ipath))
            if (ij.get("reads") == n and abs(ij.get("genome_scale", -1) - args.genome_scale) < 1e-9 and ij.get("k") == args.k and not strong and not real_ref
                    and args.genome_preset == "planted-repeats" and args.l == 31 and not args.seeding_variant and not args.fast_kh):
                n_simd = int(ij["n_simd"])
                # cycles: the launch's busy cycles as the counters gave them (GRBM_GUI_ACTIVE / 8 XCDs) -- a property of the kernel on this
                # workload; this run's launch time only says what shader clock that implies here (the clock moves with the power state:
                # 1.95 GHz under the profiler, ~2.15 GHz in a plain run)
                cyc = float(ij["busy_cycles_per_launch"])
                cpi = cyc * n_simd / float(ij["wave_instructions_per_launch"])
                roofline["secondary"] = dict(bound="valu_issue", wave_instructions_per_launch=int(ij["wave_instructions_per_launch"]),
                                             valu=int(ij["valu"]), salu=int(ij["salu"]), lds=int(ij["lds"]), vmem=int(ij["vmem"]), n_simd=n_simd,
                                             busy_cycles_per_launch=int(cyc), cycles_per_instruction=round(cpi, 3),
                                             ceiling_cycles_per_instruction=ij["ceiling_cycles_per_instruction"],
                                             ceiling_scope="stage B's step only (10 VALU + 1 LDS as built, tools/valu_enc.hip, at this kernel's occupancy): a measured "
                                                           "floor of the kernel's densest loop, not of its whole instruction mix (80 % VALU / 14 % SALU / 6 % LDS over five stages)",
                                             frac=round(float(ij["ceiling_cycles_per_instruction"]) / cpi, 4),
                                             hardware_ceiling_cycles_per_instruction=2.0,
                                             hardware_ceiling_source="/opt/skills/guides/MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32 "
                                                                     "(one wave alone sustains one per 4)",
                                             frac_of_hardware_ceiling=round(2.0 / cpi, 4), waves_per_simd=ij.get("waves_per_simd"),
                                             implied_shader_clock_mhz_this_run=round(cyc / avg_kern_s / 1e6, 1),
                                             shader_clock_mhz_in_counter_pass=ij["shader_clock_mhz"],
                                             counters_taken_at_commit=ij.get("commit"), ceiling_source=ij.get("ceiling_source"),
                                             wait_share=ij.get("wait_share"), lds_bank_conflict_share=ij.get("lds_bank_conflict_share"),
                                             note="cycles per wave-instruction per SIMD = busy cycles x SIMDs / instructions, both counted by rocprofv3 --pmc "
                                                  "on this workload (separate passes); frac = ceiling / achieved: how close the kernel is to what a SIMD "
                                                  "can issue on stage B's instruction mix at this occupancy")
        except Exception as ex:  # noqa: BLE001
            roofline["secondary"] = {"error": repr(ex)[:200]}

    # ---- accuracy on the whole batch (BASELINE metric: "Q60 mapeval parity"): paftools-mapeval-style counts
    if have_truth:
        truth = B.truth
        pafs = {"mapped": (hits["status"] == 1).astype(np.uint32)}
        for a_ in ("ref_id", "rc", "mapq", "r_start", "r_end"):
            pafs[a_] = hits[a_]
        n_m, n_q60, n_q60_wrong = sim.mapeval(truth, pafs)
    else:  # real reads: no truth in the names -- truth-free counts only (the reference's Table 1 reports the same two for HG002)
        n_m, n_q60, n_q60_wrong = n_mapped, int(((hits["status"] == 1) & (hits["mapq"] == 60)).sum()), None

    # ---- CPU baseline: the C oracle ("port") on a bounded sample of the same reads, rank 0 at N=1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        po = O.params(k=args.k, l=args.l, density=args.density)
        t0 = time.time()
        ox = O.Index()
        ox.build_mt(genome, ctg_off, ctg_names, po, ncpu)
        t_cpu_index = time.time() - t0
        ns = min(args.cpu_sample_reads or 49152, n, B.keep_first)
        sb = B.head_bases[:int(offs[ns])]
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
        same = columns_equal(mq, hits[:ns], want)
        # and every 32nd read of the rest of the batch (a lost or misplaced result anywhere in the launch, not only among its first reads)
        n_strided = int(B.strided_idx.size)
        same_strided = columns_equal(mq, hits[B.strided_idx], ox.map_batch(B.strided_bases, B.strided_offsets, po, threads=ncpu)) if n_strided else None
        sb_bases = int(so[-1])
        cpu = dict(value=round(sb_bases / t_cpu / 1e9, 4), unit="Gbases/s", cores=ncpu, kind="port",
                   sample="first %d reads (%d bases) of the step batch, C oracle with %d pthreads (cgroup CPU quota of the box), median of %d "
                          "passes (%.1f s), index build (%.1f s) excluded" % (ns, sb_bases, ncpu, reps, t_cpu * reps, t_cpu_index),
                   passes=reps, spread=dict(best=round(sb_bases / t_lo / 1e9, 4), worst=round(sb_bases / t_hi / 1e9, 4)),
                   seconds=round(t_cpu * reps, 2), paf_columns_identical_to_gpu=same, strided_sample_reads=n_strided,
                   strided_sample_identical_to_gpu=same_strided, unique_kminmers_equal=bool(ox.count() == n_unique),
                   at_10_threads=dict(value=round(sb_bases / t10 / 1e9, 4), threads=10, passes=reps10,
                                      spread=dict(best=round(sb_bases / t10_lo / 1e9, 4), worst=round(sb_bases / t10_hi / 1e9, 4)),
                                      note="median; 10 pthreads on %d granted cores; mirrors --threads 10 of experiments/figure-k-l/get_mapstats.sh:6" % ncpu),
                   published=dict(value=1.56, unit="Gbases/s", threads=10, seconds=19.98,
                                  source="experiments/figure-k-l/k_perf.csv:5 (k=5: 19.98 s for CHM13 10X, ~31.2 Gbases; the reference's own "
                                         "run on its authors' machine, Rust path, not measured here)"))
        del ox

    # ---- kernel-only legs of the other configurations (rank 0, N=1): a human-like repeat landscape, a maize-like repetitive genome
    # (BASELINE config 5, experiments/simulate_maize.sh:9) and -k 7 (BASELINE config 4, experiments/table1.sh:50)
    configs = None
    if rank == 0 and world == 1 and not args.no_configs and not strong and not real_ref:  # (synthetic stand-ins of the other configurations)
        from oracle import oracle as O
        configs = {}
        nr = args.config_reads or args.reads
        lsteps, lwarm = max(3, min(args.steps, 10)), 1

        def run_leg(name, fn):
            t0 = time.time()
            try:
                configs[name] = fn()
                configs[name]["leg_wall_s"] = round(time.time() - t0, 1)
            except Exception as ex:  # noqa: BLE001
                configs[name] = {"error": repr(ex)[:300]}

        def leg_k7():
            P7, po7 = mq.Params(k=7, l=31, density=0.01), O.params(k=7, l=31, density=0.01)
            return kernel_leg(mq, sim, O, torch, dev, local_rank, P7, po7, genome, ctg_off, ctg_names, B, lsteps, lwarm, ncpu, args.config_sample_reads,
                              "the step batch's genome and reads at -k 7 -l 31 -d 0.01 (BASELINE config 4's parameters, experiments/table1.sh:50)", n_reads=nr)

        def leg_variant(v):
            # the third-party k-min-mer iterator's likeliest other readings (the crate's SIMD hash modes -- the
            # reference's default HashMode::HpcSimd, src/mers.rs:22-23 -- may hash on 32 bits with an f32 bound): the step batch at seeding
            # variant v, the oracle switched to the same variant for the column check
            Pv = mq.Params(k=args.k, l=args.l, density=args.density, seeding_variant=v)
            return kernel_leg(mq, sim, O, torch, dev, local_rank, Pv, O.params(k=args.k, l=args.l, density=args.density), genome, ctg_off, ctg_names, B, lsteps, lwarm,
                              ncpu, args.config_sample_reads, "the step batch's genome and reads at seeding variant %d (%s), k=%d l=%d d=%g" % (
                                  v, " + ".join(nm for bit, nm in ((1, "strict < on the bound"), (2, "f32 bound"), (4, "32-bit ntHash"), (8, "position = run end"),
                                                                   (16, "end from the compressed window"), (32, "rev on <=")) if v & bit), args.k, args.l, args.density),
                              n_reads=nr, variant=v)

        def leg_genome(lens_, kw, seed, workload):
            g, co, cn = sim.make_genome(lens_, seed=seed, threads=threads, **kw)
            Bl = load_batch(torch, dev, sim, g, co, nr, seed + 1, threads, keep_first=min(args.config_sample_reads, nr), keep_stride=0)
            try:
                return kernel_leg(mq, sim, O, torch, dev, local_rank, P, O.params(k=args.k, l=args.l, density=args.density), g, co, cn, Bl, lsteps, lwarm, ncpu,
                                  args.config_sample_reads, workload)
            finally:
                del Bl

        def leg_fast_kh():
            # the opt-in cheap tuple hash (MQ_FLAG_FAST_KH): never the headline; the PAF depends on the tuple hash through equality only
            # (src/index.rs:100-104,118-126), so the records must be the step's, byte for byte -- and the oracle's at its own bit 64
            Pf = mq.Params(k=args.k, l=args.l, density=args.density, fast_kh=True)
            return kernel_leg(mq, sim, O, torch, dev, local_rank, Pf, O.params(k=args.k, l=args.l, density=args.density), genome, ctg_off, ctg_names, B, lsteps, lwarm,
                              ncpu, args.config_sample_reads, "the step batch's genome and reads with MQ_FLAG_FAST_KH (an add-rotate-xor tuple hash in SipHash-1-3's place: "
                              "~80 instead of ~250 instructions per k-min-mer at k = 5); k=%d l=%d d=%g" % (args.k, args.l, args.density), n_reads=nr, variant=64, same_as=hits)

        if args.k != 7:
            run_leg("k7", leg_k7)
        for v in (4, 6):
            run_leg("variant%d" % v, lambda v=v: leg_variant(v))
        run_leg("fast_kh", leg_fast_kh)
        if args.genome_preset != "human-like":
            run_leg("human_like", lambda: leg_genome(lens, sim.HUMAN_LIKE, args.seed + 31,
                                                     "CHM13-sized genome (scale %.3g) with tools/sim.py HUMAN_LIKE repeats (6%% satellite arrays at 99.8%% identity, 5%% segmental "
                                                     "duplications, young interspersed copies) x pbsim-like HiFi reads; k=%d l=%d d=%g" % (args.genome_scale, args.k, args.l, args.density)))
        if args.genome_preset != "maize-like":
            run_leg("maize_like", lambda: leg_genome([max(40, int(x * args.genome_scale)) for x in sim.MAIZE_LIKE], MAIZE_KW, args.seed + 57,
                                                     "maize-B73-shaped genome (10 contigs, 2.13 Gbp x scale %.3g), ~85%% of the bases in 400 transposon-like families "
                                                     "(copies 1-5%% apart), 300 runs of N, x pbsim-like HiFi reads (BASELINE config 5, experiments/simulate_maize.sh:9); "
                                                     "k=%d l=%d d=%g" % (args.genome_scale, args.k, args.l, args.density)))

    # ---- end to end (rank 0, N=1): host buffers -> hits through three stream slots, and FASTA files -> PAF through the native driver
    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e:
        import tempfile
        e2e = {}
        n_hb = min(n, 393216, B.keep_first)  # (page-locking the whole step batch would take longer than the leg)
        reads = host_reads(B, B.keep_first)  # the batch's first reads in host memory: what the file legs write out
        gb, h_e2e = measure_host_buffers(mq, ix, dict(bases=reads["bases"][:int(offs[n_hb])], offsets=offs[:n_hb + 1]))
        e2e["host_buffers_gbases_s"] = round(gb, 2)
        e2e["host_buffers_reads"] = n_hb
        e2e["host_buffers_hits_identical"] = bool(np.array_equal(h_e2e.view(np.uint8), hits[:n_hb].view(np.uint8)))
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
            # (four host threads: the records are found on the device, the reader threads only cut the mapped file at record starts)
            leg("file_to_paf", reads=reads, n_reads=args.e2e_file_reads, threads=min(4, ncpu), workdir=wd, extra_args=kargs)
            e2e["file_to_paf_gbases_s"] = e2e["file_to_paf"].get("gbases_s")
            leg("file_to_paf_8_threads", reads=reads, n_reads=args.e2e_file_reads, threads=min(8, ncpu), workdir=wd, extra_args=kargs)
            # the same with every chunk read and parsed by the reader threads (earlier rounds' path), all granted threads
            leg("file_to_paf_host_parse", reads=reads, n_reads=args.e2e_file_reads, threads=ncpu, workdir=wd, extra_args=kargs, env={"MQ_DRIVER_HOST_PARSE": "1"})
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
                    big = reads if args.e2e_fastq_reads <= B.keep_first else sim.make_reads(genome, ctg_off, args.e2e_fastq_reads, seed=args.seed + 5000, threads=threads)
                    k7 = ["-k", "7", "-l", "31", "-d", "0.01"]
                    # (the driver's lean reader: header + sequence lines read with one pread per record, qualities never read; the other reader --
                    # records found on the device, the whole file on the link -- is profiles/r05_fastq_readers.txt)
                    leg("fastq_k7_to_paf", reads=big, n_reads=args.e2e_fastq_reads, threads=ncpu, workdir=wd, fastq=True, extra_args=k7, more_threads=(4, 8))
                    leg("fasta_k7_to_paf", reads=big, n_reads=args.e2e_fastq_reads, threads=ncpu, workdir=wd, fastq=False, extra_args=k7)
                    a_, b_ = e2e["fastq_k7_to_paf"].get("no_prefetch_gbases_s"), e2e["fasta_k7_to_paf"].get("no_prefetch_gbases_s")
                    e2e["fastq_over_fasta_rate"] = round(a_ / b_, 3) if a_ and b_ else None
                    del big
                except Exception as ex:  # noqa: BLE001
                    e2e["fastq_k7_to_paf"] = {"error": repr(ex)[:300]}

    if rank == 0:
        value = all_bases * args.steps / elapsed / 1e9
        line = {
            "metric": ("Gbases/s mapped (%s, k=%d l=%d d=%g)" % ("real reference" + (" + real reads" if args.reads_fastx else ", simulated HiFi reads"), args.k, args.l, args.density))
                      if real_ref else "Gbases/s mapped (sim %s HiFi, k=%d l=%d d=%g%s)" % ("maize-B73-like" if args.genome_preset == "maize-like" else "CHM13v2-like", args.k, args.l,
                                                                                           args.density, ", MQ_FLAG_FAST_KH: not the BASELINE tuple hash" if args.fast_kh else ""),
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
            "data": ("real" if args.reads_fastx else "real reference, simulated reads") if real_ref else "synthetic",
            "config": {
                "genome_preset": "real" if real_ref else args.genome_preset,
                "workload": ("reference %s (%d contigs, %.3f Gbp) x %s; k=%d l=%d d=%g HPC"
                             % (realdata.describe(args.reference_fasta), len(lens), sum(lens) / 1e9,
                                ("reads %s, %d per rank from read %d on%s" % (realdata.describe(args.reads_fastx), n, 0 if strong else rank * args.reads,
                                                                             "" if have_truth else " (no truth in the read names: q60_wrong is null)"))
                                if args.reads_fastx else "pbsim-like HiFi reads simulated from it (mean 24 kb, 1% error)", args.k, args.l, args.density))
                            if real_ref else
                            "%s synthetic genome (%d contigs, %.3f Gbp, scale %.3g, %s) "
                            "x pbsim-like HiFi reads (mean 24 kb, 1%% error); k=%d l=%d d=%g HPC"
                            % ("maize-B73v5-like" if args.genome_preset == "maize-like" else "CHM13v2.0-like", len(lens), sum(lens) / 1e9, args.genome_scale,
                               "human-like repeats: 6% satellite arrays, 5% segmental duplications, young interspersed copies" if args.genome_preset == "human-like"
                               else "maize-B73-shaped: 10 contigs, ~85% of the bases in 400 transposon-like families, 300 runs of N" if args.genome_preset == "maize-like"
                               else "%g%% planted repeats + %g%% tandem arrays" % (100 * args.repeat_frac, 100 * args.tandem_frac),
                               args.k, args.l, args.density),
                "seeding_variant": args.seeding_variant,
                "fast_kh": bool(args.fast_kh),
                "reads_per_step_per_gpu": n,
                "bases_per_step_per_gpu": total_bases,
                "index_unique_kminmers": int(n_unique),
                "index_table_bytes": int(st["table_bytes"]),
                "parallelism": ("one fixed read set in host memory dealt to %d GPU(s) (contiguous shards), index replicated, no data-path "
                                "collective; host buffers -> stream slots -> hits, PCIe included" if strong else
                                "reads sharded over %d GPU(s) (every rank its own HBM-resident batch), index replicated, no data-path collective") % world,
            },
            "mreads_per_s": round(all_reads * args.steps / elapsed / 1e6, 4),
            "per_rank_gbases_s": per_rank,
            "collectives": ("gloo, CPU tensors (MQ_BENCH_FAKE_RANKS)" if fake else "nccl (RCCL), device tensors") if use_dist else None,
            "mapped_frac": round(n_mapped / max(n, 1), 4),
            "overflow_reads": n_over,
            "records_written": n_written,
            "unmapped_reads_with_kminmers": n_unmapped_seeded,
            "kminmers_per_step": n_kmm,
            "launch_order": launch_order,
            "smaller_batches": smaller or None,
            "setup_s": {"genome": round(t_genome, 1), "genome_upload": round(t_upload, 2), "gpu_index": round(t_index, 3), "reads": round(t_reads, 1),
                        "process_start_to_first_timed_step": round(t_first_step, 1)},
            "per_rank_setup_s": per_rank_setup,
            "per_rank_peak_rss_gb": per_rank_rss,
            "peak_rss_gb_at_exit": peak_rss_gb(),
            "index_build": index_build,
            "q60": n_q60,
            "q60_wrong": n_q60_wrong,
            "mapped_reads": n_m,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "configs": configs,
            "end_to_end": e2e,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
