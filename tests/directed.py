"""Directed inputs for the reference's sharp edges, found by searching with the CPU oracle's branch counters
(oracle.Index.map_batch_diag).  Used by the CPU tests (the constructions really reach the branches) and by the GPU parity
tests (the HIP path agrees with the oracle on exactly those reads).

  quirk   Match::check parses as (A && B && C) || D (src/match.rs:39-43): a forward run is extended by any hit whose
          reference offset is +1, even on another reference.  With k = 1 a chimeric read A|B whose last A minimizer has
          offset o and whose first B minimizer has offset o + 1 puts both references into ONE Match keyed under A.
  tie     two candidate references with equal scores => None (src/mers.rs:104-108): a chimeric read with as many
          matching k-min-mers on A as on B.
  wrap    `as i32` casts in the gap tests (src/chain.rs:132-142): reads on a contig longer than 2^31 bases.
  usize   find_coords (src/mers.rs:131-183) computes in usize and wraps in a release build: a forward run that the quirk
          extended onto ANOTHER reference, at a position beyond the end of the reference the run is keyed under, makes
          `r_len - r_end - 1` wrap, and the reference prints q_end = 2^64 - something in column 4.
"""
import numpy as np


def _concat(seqs):
    bases = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([s.size for s in seqs])
    return bases, offs


QUIRK_PARAMS = dict(k=1, l=15, density=0.05)


def quirk_case(O, sim, n_want=24, seed=101):
    """Two random references + chimeric reads that extend a forward Match across the reference boundary.
    Returns (genome, ctg_off, names, bases, offsets, params_dict)."""
    g, off, names = sim.make_genome([150000, 150000], seed=seed)
    A, B = g[:150000], g[150000:]
    po = O.params(**QUIRK_PARAMS)
    ka, kb = O.kminmers(A, po), O.kminmers(B, po)  # k = 1: one k-min-mer per minimizer, offset = its ordinal
    l = QUIRK_PARAMS["l"]
    seqs = []
    rng = np.random.default_rng(seed)
    for o in rng.permutation(np.arange(40, min(len(ka), len(kb)) - 40))[:40 * n_want]:
        o = int(o)
        a_lo = int(ka[o - 12]["start"])                       # a dozen A minimizers, the last one has offset o
        a_hi = int(ka[o]["end"]) + 1                          # one raw base past start + l - 1
        a_hi = max(a_hi, int(ka[o]["start"]) + 2 * l)         # room for l compressed bases under HPC
        if a_hi > int(ka[o + 1]["start"]):
            continue
        b_lo = int(kb[o + 1]["start"])                        # the B part opens with B's minimizer of offset o + 1
        b_hi = int(kb[o + 14]["end"]) + 2 * l
        seqs.append(np.concatenate([A[a_lo:a_hi], B[b_lo:b_hi]]))
        if len(seqs) >= 6 * n_want:
            break
    bases, offs = _concat(seqs)
    return g, off, names, bases, offs, dict(QUIRK_PARAMS)


def usize_wrap_case(O, sim, n_want=24, seed=404):
    """quirk_case with the second reference B = 200 kb homopolymer run (one compressed base: no minimizer) + random sequence, so
    that B's minimizer of offset o + 1 lies beyond the END of reference A (150 kb): the run keyed under A ends with B's
    coordinates and find_coords' `r_len - r_end - 1` wraps."""
    ga, _, _ = sim.make_genome([150000], seed=seed)
    gb, _, _ = sim.make_genome([150000], seed=seed + 1)
    head = np.full(200000, ord("A"), dtype=np.uint8)
    if gb[0] == ord("A"):
        gb[0] = ord("C")
    A, B = ga, np.concatenate([head, gb])
    g = np.concatenate([A, B])
    off = np.array([0, A.size, A.size + B.size], dtype=np.uint64)
    names = ["short", "long"]
    po = O.params(**QUIRK_PARAMS)
    ka, kb = O.kminmers(A, po), O.kminmers(B, po)
    l = QUIRK_PARAMS["l"]
    seqs = []
    rng = np.random.default_rng(seed)
    for o in rng.permutation(np.arange(40, min(len(ka), len(kb)) - 40))[:40 * n_want]:
        o = int(o)
        a_lo = int(ka[o - 12]["start"])
        a_hi = max(int(ka[o]["end"]) + 1, int(ka[o]["start"]) + 2 * l)
        if a_hi > int(ka[o + 1]["start"]):
            continue
        b_lo = int(kb[o + 1]["start"])
        b_hi = int(kb[o + 14]["end"]) + 2 * l
        seqs.append(np.concatenate([A[a_lo:a_hi], B[b_lo:b_hi]]))
        if len(seqs) >= 6 * n_want:
            break
    bases, offs = _concat(seqs)
    return g, off, names, bases, offs, dict(QUIRK_PARAMS)


def tie_case(O, sim, n_want=24, seed=202):
    """Two random references + error-free chimeric reads A|B; the caller keeps those the oracle reports as ties."""
    g, off, names = sim.make_genome([300000, 300000], seed=seed)
    A, B = g[:300000], g[300000:]
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(60 * n_want):
        la, lb = int(rng.integers(1500, 4000)), int(rng.integers(1500, 4000))
        xa, xb = int(rng.integers(0, A.size - la)), int(rng.integers(0, B.size - lb))
        seqs.append(np.concatenate([A[xa:xa + la], B[xb:xb + lb]]))
    bases, offs = _concat(seqs)
    return g, off, names, bases, offs, dict()


def select(bases, offs, keep):
    """Sub-batch of the reads whose indices are in `keep`."""
    return _concat([bases[int(offs[i]):int(offs[i + 1])] for i in keep])


WRAP_HEAD = (1 << 31) - 30_000_000   # homopolymer prefix: the random part straddles 2^31
WRAP_TAIL = 60_000_000


def wrap_case(sim, n_reads=400, seed=303):
    """One contig of 2^31 + 30 Mbp: a homopolymer run (one base after compression; nothing to seed) followed by 60 Mbp of
    random sequence that straddles position 2^31.  Reads come from the random part, so their reference coordinates are
    around and above 2^31 and go through the `as i32` casts of the gap tests."""
    tail, toff, _ = sim.make_genome([WRAP_TAIL], seed=seed)
    g = np.empty(WRAP_HEAD + WRAP_TAIL, dtype=np.uint8)
    g[:WRAP_HEAD] = ord("A")
    g[WRAP_HEAD:] = tail
    if g[WRAP_HEAD] == ord("A"):
        g[WRAP_HEAD] = ord("C")
    reads = sim.make_reads(tail, toff, n_reads, seed=seed + 1, err=0.02)  # 2 % error: several Matches per read => gap tests run
    reads["start"] = reads["start"] + np.uint64(WRAP_HEAD)
    reads["end"] = reads["end"] + np.uint64(WRAP_HEAD)
    off = np.array([0, g.size], dtype=np.uint64)
    return g, off, ["chrBig"], reads
