"""Thin host driver with mapquik's command-line surface (reference: src/main.rs:77-272, src/closures.rs:22-212).

    python -m mapquik_amd <reads.fa|fq[.gz]> --reference <ref.fa[.gz]> [-k -l -d -c -s -g -p --threads -b -q --nohpc ...]

Same flags, defaults, log lines and `<prefix>.paf` output as the reference binary; the hot path (ref_extract /
find_matches) runs on the GPU through the C ABI.  This is the SURVEY 8(f2/f3) row kept deliberately small: a
single-threaded FASTX reader that upper-cases (src/closures.rs:63,106), batches reads and writes PAF lines in input
order (src/closures.rs:117-123).  No CPU compute fallback exists.
"""
import argparse
import gzip
import resource
import sys
import time

import numpy as np

from . import api


def rust_duration(seconds):
    """`{:?}` of a std::time::Duration: s / ms / µs / ns with up to 9 significant fractional digits, zeros trimmed."""
    ns = int(round(seconds * 1e9))
    for unit, div in (("s", 10**9), ("ms", 10**6), ("µs", 10**3), ("ns", 1)):
        if ns >= div or unit == "ns":
            whole, frac = divmod(ns, div)
            if frac == 0 or div == 1:
                return "%d%s" % (whole, unit)
            digits = len(str(div)) - 1
            return "%d.%s%s" % (whole, ("%0*d" % (digits, frac)).rstrip("0"), unit)


def rust_float(x):
    """`{}` of an f64 as Rust prints it for the values that occur here (1.0 -> "1", 0.01 -> "0.01")."""
    if x == int(x) and abs(x) < 1e16:
        return str(int(x))
    return repr(float(x))


def is_fasta_name(name):
    """src/main.rs:196,202"""
    return (".fasta." in name or name.endswith(".fna") or ".fna." in name or ".fa." in name or name.endswith(".fa")
            or name.endswith(".fasta"))


def open_maybe_compressed(path):
    """get_reader (src/main.rs:60-75): raw, .gz; .lz4 is not available in this host driver."""
    if path.endswith(".lz4"):
        raise SystemExit("Error opening compressed file: lz4 input is not supported by this driver")
    if path.endswith(".gz"):
        return gzip.open(path, "rb")
    return open(path, "rb")


def read_fastx(path, fasta):
    """Yields (id, upper-cased sequence bytes).  FASTA may be multi-line; FASTQ is 4-line."""
    with open_maybe_compressed(path) as fh:
        if fasta:
            name, chunks = None, []
            for line in fh:
                line = line.rstrip(b"\r\n")
                if line.startswith(b">"):
                    if name is not None:
                        yield name, b"".join(chunks).upper()
                    name, chunks = line[1:].split(b" ", 1)[0].decode(), []  # seq_io's id(): up to the first SPACE (src/closures.rs:107)
                elif name is not None:
                    chunks.append(line)
            if name is not None:
                yield name, b"".join(chunks).upper()
        else:
            while True:
                h = fh.readline()
                if not h:
                    return
                s = fh.readline().rstrip(b"\r\n")
                fh.readline()
                fh.readline()
                yield h.rstrip(b"\r\n")[1:].split(b" ", 1)[0].decode(), s.upper()


def build_parser():
    ap = argparse.ArgumentParser(prog="mapquik", description="Original implementation of mapquik, a fast HiFi read mapper. (HIP backend)")
    ap.add_argument("reads", nargs="?")
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("-p", "--prefix")
    ap.add_argument("-k", type=int)
    ap.add_argument("-l", type=int)
    ap.add_argument("-d", "--density", type=float)
    ap.add_argument("-c", "--chain", type=int)
    ap.add_argument("-s", "--seed", type=int)
    ap.add_argument("-g", "--gap-diff", type=int, dest="gap_diff")
    ap.add_argument("--reference")
    ap.add_argument("--threads", type=int)
    ap.add_argument("--low-memory", action="store_true")
    ap.add_argument("--nosimd", action="store_true")
    ap.add_argument("--nohpc", action="store_true")
    ap.add_argument("--parallelfastx", action="store_true")
    ap.add_argument("-b", type=int)
    ap.add_argument("-q", type=int)
    ap.add_argument("--device", type=int, default=0, help="HIP device ordinal (extension)")
    ap.add_argument("--batch-bases", type=int, default=1 << 30, help="bases per GPU batch (extension)")
    ap.add_argument("--save-index", dest="save_index", help="write the finalized index (occupied slots only) for later runs (extension: the "
                    "reference re-indexes the FASTA on every run, src/closures.rs:24-94)")
    ap.add_argument("--index", dest="load_index", help="map against a saved index instead of indexing --reference; same -k -l -d --nohpc as it was "
                    "built with (extension)")
    ap.add_argument("--seeding-variant", type=int, default=0, dest="seeding_variant",
                    help="reading of the k-min-mer iterator's unpinned decisions, bits 1 2 4 8 16 32 (include/mapquik_hip.h); 0 = the frozen "
                         "reading (extension; tools/check_against_upstream.sh finds the value that reproduces the crate)")
    ap.add_argument("--fast-kh", action="store_true", dest="fast_kh",
                    help="cheap k-min-mer tuple hash instead of SipHash-1-3 (MQ_FLAG_FAST_KH): the same PAF -- the hash acts through equality only, "
                         "src/index.rs:118-126 -- with fewer instructions; KminmerHash.hash is then not the reference's value (extension)")
    ap.add_argument("--unmapped", action="store_true",
                    help="also write <prefix>.unmapped.out with the ids of reads that got no PAF line (extension; the reference "
                         "has this writer commented out, src/closures.rs:38-43; feeds a second pass with other parameters)")
    return ap


def banner_lines(opt):
    """The stdout lines main() prints before run_mers (src/main.rs:196-240); returns (lines, settings)."""
    out = []
    k, l, c, s, g, b, q, density, threads = 5, 31, 4, 11, 2000, 1, 200, 0.01, 8
    reads_fasta = is_fasta_name(opt.reads)
    ref_fasta = is_fasta_name(opt.reference) if opt.reference else False
    if reads_fasta:
        out += ["Input file: %s" % opt.reads, "Format: FASTA"]
    if ref_fasta and not getattr(opt, "load_index", None):
        out += ["Reference file: %s" % opt.reference, "Format: FASTA"]
    if opt.k is not None: k = opt.k
    else: out.append("Warning: Using default k value (%d)." % k)
    if opt.l is not None: l = opt.l
    else: out.append("Warning: Using default l value (%d)." % l)
    if opt.b is not None: b = opt.b
    else: out.append("Warning: Using default buffer size (%dX)." % b)
    if opt.q is not None: q = opt.q
    else: out.append("Warning: Using default queue length (%d)." % q)
    if opt.density is not None: density = opt.density
    else: out.append("Warning: Using default density value (%s%%)." % rust_float(density * 100.0))
    if opt.threads is not None: threads = opt.threads
    else: out.append("Warning: Using default number of threads (8).")
    if opt.chain is not None: c = opt.chain
    else: out.append("Warning: Using default minimum chain length (%d)." % c)
    if opt.seed is not None: s = opt.seed
    else: out.append("Warning: Using default minimum number of matching seeds (%d)." % s)
    if opt.gap_diff is not None: g = opt.gap_diff
    else: out.append("Warning: Using default maximum seed gap difference (%d)." % g)
    prefix = "mapquik-k%d-d%s-l%d" % (k, rust_float(density), l)
    if opt.prefix is not None: prefix = opt.prefix
    else: out.append("Warning: Using default output prefix (%s)." % prefix)
    use_hpc, use_simd = not opt.nohpc, not opt.nosimd
    if use_hpc:
        out.append("Using HPC ntHash, with SIMD" if use_simd else "Using HPC ntHash, scalar")
    else:
        out.append("Using regular ntHash (not HPC), with SIMD" if use_simd else "Using regular ntHash (not HPC), scalar")
    return out, dict(k=k, l=l, density=density, use_hpc=use_hpc, c=c, s=s, g=g, prefix=prefix, reads_fasta=reads_fasta,
                     ref_fasta=ref_fasta)


def main(argv=None):
    start = time.time()
    opt = build_parser().parse_args(argv)
    if not opt.reads:
        raise SystemExit("Please specify an input file.")
    if not opt.reference and not opt.load_index:
        raise SystemExit("Please specify a reference file.")
    if opt.load_index and opt.save_index:
        raise SystemExit("error: --index cannot be combined with --save-index")
    lines, st = banner_lines(opt)
    for ln in lines:
        print(ln)
    if opt.seeding_variant:
        print("Seeding variant %d (reading of rust-seq2kminmers other than the frozen one; include/mapquik_hip.h)." % opt.seeding_variant)
    if opt.fast_kh:
        print("Fast k-min-mer tuple hash (MQ_FLAG_FAST_KH): same PAF, KminmerHash.hash is not the reference's value.")
    params = api.Params(k=st["k"], l=st["l"], density=st["density"], use_hpc=st["use_hpc"], c=st["c"], s=st["s"], g=st["g"],
                        seeding_variant=opt.seeding_variant, fast_kh=opt.fast_kh)
    paf = open(st["prefix"] + ".paf", "w")  # src/closures.rs:32
    unm = open(st["prefix"] + ".unmapped.out", "w") if opt.unmapped else None

    t0 = time.time()
    if opt.load_index:
        index = api.Index.load(opt.load_index, device=opt.device)
        fp = index.get_params()
        if (fp.k, fp.l, fp.density, fp.use_hpc, fp.seeding_variant, fp.fast_kh) != (params.k, params.l, params.density, params.use_hpc, params.seeding_variant,
                                                                                     params.fast_kh):
            raise SystemExit("%s was built with -k %d -l %d -d %s%s --seeding-variant %d%s: run with the same seeding parameters"
                             % (opt.load_index, fp.k, fp.l, rust_float(fp.density), "" if fp.use_hpc else " --nohpc", fp.seeding_variant,
                                " --fast-kh" if fp.fast_kh else ""))
        index.set_map_params(params.c, params.s, params.g, bool(params.flags & api.MQ_FLAG_FOLD_CASE))
        ist = index.stats()
        print("Loaded index %s: %d references, %d k-min-mers." % (opt.load_index, ist["n_refs"], ist["n_kminmers"]))
        unique = ist["n_unique"]
    else:
        index = api.Index(params, device=opt.device)
        for ref_idx, (name, seq) in enumerate(read_fastx(opt.reference, st["ref_fasta"])):
            n = index.add_ref(ref_idx, name, np.frombuffer(seq, dtype=np.uint8))
            print("Indexed reference %s: %d k-min-mers." % (name, n))  # src/closures.rs:58
        unique = index.finalize()
        if opt.save_index:
            ts = time.time()
            index.save(opt.save_index)
            print("Saved index to %s in %s." % (opt.save_index, rust_duration(time.time() - ts)))
    print("Indexed %d unique k-min-mers in %s." % (unique, rust_duration(time.time() - t0)))  # src/closures.rs:92

    t0 = time.time()
    if opt.parallelfastx and not opt.reads.endswith((".gz", ".lz4")):
        print("Warning: using experimental rust-parallelfastx (exciting!)")  # src/closures.rs:192 (same reader here)

    def flush(names, seqs):
        if not names:
            return
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(x) for x in seqs])
        bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
        hits = index.map_batch(bases, offs)
        for ln in index.paf_lines(names, offs, hits):  # input order, unmapped reads write nothing (src/closures.rs:117-123)
            paf.write(ln + "\n")
        if unm is not None:
            for nm, h in zip(names, hits):
                if int(h["status"]) != api.MQ_HIT_MAPPED:
                    unm.write(nm + "\n")

    names, seqs, acc = [], [], 0
    for name, seq in read_fastx(opt.reads, st["reads_fasta"]):
        names.append(name)
        seqs.append(seq)
        acc += len(seq)
        if acc >= opt.batch_bases:
            flush(names, seqs)
            names, seqs, acc = [], [], 0
    flush(names, seqs)
    paf.close()
    if unm is not None:
        unm.close()
    print("Mapped query sequences in %s." % rust_duration(time.time() - t0))  # src/closures.rs:211
    print("Total execution time: %s" % rust_duration(time.time() - start))  # src/main.rs:270
    rss_gb = np.float32(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024) / np.float32(1024.0 ** 3)
    print("Maximum RSS: %sGB" % repr(float(rss_gb)))  # src/main.rs:271
    return 0


if __name__ == "__main__":
    sys.exit(main())
