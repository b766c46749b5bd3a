"""Seeded synthetic genomes and HiFi-like reads (inputs for tests and bench.py).

Wraps tools/mqsim.c.  Not part of the product path and not part of the oracle.
Stands in for pbsim + ecoli.genome.fa / CHM13v2.0, none of which are in this image
(reference recipes: example/simulate_pbsim.sh:7-14, experiments/simulate_chm13.sh).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmqsim.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "mqsim.c")
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= os.path.getmtime(src):
        return _LIB
    subprocess.check_call(["gcc", "-O3", "-std=c11", "-fPIC", "-shared", "-o", _LIB, src, "-lpthread", "-lm"], cwd=_HERE)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vp, u64, u32, dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_double
        L.mqsim_genome.argtypes = [vp, u64, u64, C.c_int]
        L.mqsim_plant_repeats.argtypes = [vp, u64, u64, u64, u64, u64, u64, dbl]
        L.mqsim_plant_families.argtypes = [vp, u64, u64, u64, u64, u64, u64, dbl, dbl]
        L.mqsim_plant_n.argtypes = [vp, u64, u64, u64, u64, u64]
        L.mqsim_plant_satellites.argtypes = [vp, u64, u64, u64, u64, u64, u64, u64, dbl]
        L.mqsim_read_caps.argtypes = [vp, u32, u32, dbl, dbl, u64, u64, u64, vp]
        L.mqsim_reads.argtypes = [vp, vp, u32, u32, dbl, dbl, u64, u64, dbl, dbl, dbl, u64, C.c_int, vp, vp, vp, vp, vp, vp, vp]
        L.mqsim_compact.argtypes = [vp, vp, vp, u32, vp, vp]
        L.mqsim_read_caps_range.argtypes = [vp, u32, u32, u32, dbl, dbl, u64, u64, u64, vp]
        L.mqsim_reads_range.argtypes = [vp, vp, u32, u32, u32, dbl, dbl, u64, u64, dbl, dbl, dbl, u64, C.c_int, vp, vp, vp, vp, vp, vp, vp]
        L.mqsim_write_fastx.restype = u64
        L.mqsim_write_fastx.argtypes = [C.c_char_p, vp, vp, u32, C.c_int, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# 25 contig lengths shaped like CHM13v2.0 (chr1..22, X, Y, M), rounded; sum ~3.117 Gbp
CHM13_LIKE = [248387328, 242696752, 201105948, 193574945, 182045439, 172126628, 160567428, 146259331, 150617247,
              134758134, 135127769, 133324548, 113566686, 101161492, 99753195, 96330374, 84276897, 80542538, 61707364,
              66210255, 45090682, 51324926, 154259566, 62460029, 16569]
ECOLI_LEN = [4641652]  # example/ecoli.genome.fa.fai:1


# 10 contigs shaped like maize B73 v5 chr1..10 (rounded; sum ~2.13 Gbp, longest 308 Mbp)
MAIZE_LIKE = [308452471, 243675191, 238017767, 250330460, 226353449, 181357234, 185808916, 182411202, 163004744, 152435371]


def make_genome(contig_lens, seed=913, threads=8, repeat_frac=0.0, tandem_frac=0.0, div=0.01, prefix="chr", family_frac=0.0,
                n_families=200, family_div=(0.01, 0.05), n_runs=0, satellite_frac=0.0, satellite_div=0.01, repeat_len=(1000, 20000)):
    """Uniform ACGT contigs; optionally overwrite ~repeat_frac of the bases with copied segments
    (1-20 kb, `div` divergence) and ~tandem_frac with tandem arrays; ~family_frac with copies of `n_families`
    transposon-like families (1-12 kb, per-copy divergence U[family_div]); `n_runs` runs of N (100-50,000 bases);
    ~satellite_frac with satellite arrays (0.2-3 Mbp of a 0.7-2.8 kb unit, copies `satellite_div` from the unit)."""
    lens = np.asarray(contig_lens, dtype=np.uint64)
    offsets = np.zeros(lens.size + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    total = int(offsets[-1])
    g = np.empty(total, dtype=np.uint8)
    lib().mqsim_genome(_p(g), total, seed, threads)
    if repeat_frac > 0 or tandem_frac > 0:
        mean_len = (repeat_len[0] + repeat_len[1]) // 2
        n_seg = int(total * repeat_frac / mean_len)
        n_tan = int(total * tandem_frac / mean_len)
        lib().mqsim_plant_repeats(_p(g), total, seed, n_seg, n_tan, repeat_len[0], repeat_len[1], div)
    if satellite_frac > 0:
        lo, hi = 200_000, 3_000_000
        if total >= 4 * hi:
            lib().mqsim_plant_satellites(_p(g), total, seed, max(1, int(total * satellite_frac / ((lo + hi) // 2))), lo, hi, 700, 2800, satellite_div)
    if family_frac > 0:
        lib().mqsim_plant_families(_p(g), total, seed, n_families, int(total * family_frac), 1000, 12000, family_div[0], family_div[1])
    if n_runs > 0:
        lib().mqsim_plant_n(_p(g), total, seed, n_runs, 100, 50000)
    names = ["%s%d" % (prefix, i + 1) for i in range(lens.size)]
    return g, offsets, names


# A human-like repeat landscape for the CHM13-sized stand-in (bench.py --genome-preset human-like): ~6 % satellite arrays (CHM13's
# share; copies 99.8 % identical), ~5 % segmental duplications of 10-200 kb at 1 % divergence, ~2 % young interspersed copies (0-3 % from their consensus),
# 1 % short tandem arrays.  Old interspersed repeats (the other ~45 % of a human genome) are > 10 % diverged: no k-min-mer of
# 5 x 31 bases survives that, so for this path they are unique sequence.
HUMAN_LIKE = dict(repeat_frac=0.05, repeat_len=(10000, 200000), div=0.01, tandem_frac=0.01, satellite_frac=0.06, satellite_div=0.001,
                  family_frac=0.02, n_families=300, family_div=(0.0, 0.03))


def make_reads(genome, ctg_off, n_reads, seed=1, len_mean=24000.0, len_sd=2300.0, len_min=100, len_max=25000,
               err=0.01, f_sub=0.10, f_ins=0.60, threads=8):
    """HiFi-like reads: N(len_mean, len_sd) clipped to [len_min, len_max], 50% reverse strand, per-base error
    `err` split sub:ins:del = f_sub : f_ins : rest.  Returns dict(bases, offsets, ctg, start, end, strand)."""
    ctg_off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
    n_ctg = ctg_off.size - 1
    caps = np.zeros(n_reads + 1, dtype=np.uint64)
    lib().mqsim_read_caps(_p(ctg_off), n_ctg, n_reads, len_mean, len_sd, len_min, len_max, seed, _p(caps))
    tmp = np.empty(int(caps[-1]), dtype=np.uint8)
    rl = np.zeros(n_reads, dtype=np.uint64)
    t_ctg = np.zeros(n_reads, dtype=np.uint32)
    t_start = np.zeros(n_reads, dtype=np.uint64)
    t_end = np.zeros(n_reads, dtype=np.uint64)
    t_strand = np.zeros(n_reads, dtype=np.uint8)
    lib().mqsim_reads(_p(genome), _p(ctg_off), n_ctg, n_reads, len_mean, len_sd, len_min, len_max, err, f_sub, f_ins,
                      seed, threads, _p(tmp), _p(caps), _p(rl), _p(t_ctg), _p(t_start), _p(t_end), _p(t_strand))
    total = int(rl.sum())
    bases = np.empty(total, dtype=np.uint8)
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    lib().mqsim_compact(_p(tmp), _p(caps), _p(rl), n_reads, _p(bases), _p(offsets))
    return dict(bases=bases, offsets=offsets, ctg=t_ctg, start=t_start, end=t_end, strand=t_strand)


def read_slices(genome, ctg_off, n_reads, seed=1, slice_reads=65536, len_mean=24000.0, len_sd=2300.0, len_min=100, len_max=25000,
                err=0.01, f_sub=0.10, f_ins=0.60, threads=8, buffers=None, first_read=0):
    """The reads of make_reads(genome, ctg_off, n_reads, seed, ...) slice by slice (a read is a pure function of the seed and its number):
    yields (r0, r1, bases, offsets, truth) per slice of <= slice_reads reads -- bases: the slice's reads back to back (a view of a work
    buffer that the slice after the next reuses), offsets: r1 - r0 + 1 slice-local uint64, truth: dict(ctg, start, end, strand).
    Host memory: two work buffers of one slice's capacity (65,536 HiFi reads: 1.7 GB each) instead of the batch twice over.
    buffers: a callable nbytes -> uint8 array (e.g. page-locked memory) for the two work buffers.
    first_read: the slices cover reads [first_read, first_read + n_reads) of the seeded set (r0, r1 count from first_read)."""
    ctg_off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
    n_ctg = ctg_off.size - 1
    alloc = buffers or (lambda nbytes: np.empty(nbytes, dtype=np.uint8))
    cap_max = int(slice_reads * (len_max + len_max // 16 + 64))
    work = [None, None]
    for k, r0 in enumerate(range(0, n_reads, slice_reads)):
        r1 = min(n_reads, r0 + slice_reads)
        m = r1 - r0
        caps = np.zeros(m + 1, dtype=np.uint64)
        lib().mqsim_read_caps_range(_p(ctg_off), n_ctg, first_read + r0, m, len_mean, len_sd, len_min, len_max, seed, _p(caps))
        if work[k & 1] is None:
            work[k & 1] = alloc(cap_max)
        tmp = work[k & 1]
        assert int(caps[-1]) <= tmp.size
        rl = np.zeros(m, dtype=np.uint64)
        t_ctg = np.zeros(m, dtype=np.uint32)
        t_start = np.zeros(m, dtype=np.uint64)
        t_end = np.zeros(m, dtype=np.uint64)
        t_strand = np.zeros(m, dtype=np.uint8)
        lib().mqsim_reads_range(_p(genome), _p(ctg_off), n_ctg, first_read + r0, m, len_mean, len_sd, len_min, len_max, err, f_sub, f_ins,
                                seed, threads, _p(tmp), _p(caps), _p(rl), _p(t_ctg), _p(t_start), _p(t_end), _p(t_strand))
        offsets = np.zeros(m + 1, dtype=np.uint64)
        lib().mqsim_compact(_p(tmp), _p(caps), _p(rl), m, _p(tmp), _p(offsets))  # in place: a read only ever moves towards the front (memmove)
        yield r0, r1, tmp[:int(offsets[-1])], offsets, dict(ctg=t_ctg, start=t_start, end=t_end, strand=t_strand)


def write_fastx(path, bases, offsets, n_reads, fastq=False, threads=8):
    """Reads 0..n_reads-1 as a FASTA / FASTQ file (ids r<i>, quality 'I'), written by `threads` threads.  Returns the file size."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    size = lib().mqsim_write_fastx(os.fsencode(path), _p(bases), _p(offsets), int(n_reads), 1 if fastq else 0, int(threads))
    if not size:
        raise OSError("could not write " + str(path))
    return int(size)


def read_names(reads, ctg_names):
    """pbsim2fq-style names: S1_<n>!<chr>!<start>!<end>!<strand> (example/nearperfect-ecoli.100.fa:1)."""
    out = []
    for i in range(reads["ctg"].size):
        out.append("S1_%d!%s!%d!%d!%s" % (i + 1, ctg_names[int(reads["ctg"][i])], int(reads["start"][i]),
                                           int(reads["end"][i]), "-" if reads["strand"][i] else "+"))
    return out


def mapeval(reads, pafs, min_overlap_frac=0.1):
    """Restatement of what `paftools.js mapeval` reports for our purposes: a mapped read is correct when it is on
    the true contig and strand and overlaps the true interval by >= 10% of the union.  Returns
    (n_mapped, n_q60, n_q60_wrong)."""
    m = pafs["mapped"] != 0
    same = (pafs["ref_id"] == reads["ctg"]) & ((pafs["rc"] != 0) == (reads["strand"] != 0))
    lo = np.maximum(pafs["r_start"].astype(np.int64), reads["start"].astype(np.int64))
    hi = np.minimum(pafs["r_end"].astype(np.int64) + 1, reads["end"].astype(np.int64))
    ulo = np.minimum(pafs["r_start"].astype(np.int64), reads["start"].astype(np.int64))
    uhi = np.maximum(pafs["r_end"].astype(np.int64) + 1, reads["end"].astype(np.int64))
    ov = np.clip(hi - lo, 0, None) / np.maximum(uhi - ulo, 1)
    ok = same & (ov >= min_overlap_frac)
    q60 = m & (pafs["mapq"] == 60)
    return int(m.sum()), int(q60.sum()), int((q60 & ~ok).sum())
