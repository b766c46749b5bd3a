"""CPU: the directed constructions of tests/directed.py really reach the reference's sharp edges (oracle branch counters),
and the oracle's results on them match what src/match.rs / src/mers.rs prescribe.  The GPU counterpart is test_gpu_directed.py."""
import numpy as np

import directed as D


def test_quirk_reads_extend_a_forward_match_across_references(oracle, simlib):
    g, off, names, bases, offs, ps = D.quirk_case(oracle, simlib)
    po = oracle.params(**ps)
    ox = oracle.Index()
    ox.build_mt(g, off, names, po, 2)
    out, diag = ox.map_batch_diag(bases, offs, po, threads=4)
    hit = np.nonzero(diag["quirk_cross_ref"] > 0)[0]
    assert hit.size >= 10
    # such a read has ONE candidate reference (everything is keyed under the first entry's id, src/mers.rs:68) although
    # its second half comes from the other contig, and its Match count covers both halves
    i = int(hit[0])
    assert diag["n_candidates"][i] >= 1 and out["mapped"][i] == 1 and out["ref_id"][i] == 0
    assert diag["n_hits"][i] > 14  # hits of the A part (<= 13) plus the B part ended up in the scored Match(es)
    # counter-check with explicit entries: the rc form does test the reference id
    assert diag["rc_ext"].sum() == 0  # k = 1: every tuple is a palindrome => rev = false => forward runs only


def test_tie_reads_are_unmapped(oracle, simlib):
    g, off, names, bases, offs, ps = D.tie_case(oracle, simlib)
    po = oracle.params(**ps)
    ox = oracle.Index()
    ox.build_mt(g, off, names, po, 2)
    out, diag = ox.map_batch_diag(bases, offs, po, threads=4)
    ties = np.nonzero(diag["tie"] != 0)[0]
    assert ties.size >= 10
    assert (out["mapped"][ties] == 0).all() and (diag["n_candidates"][ties] == 2).all()
    assert (diag["n_candidates"] > 1).sum() > 1000  # every chimeric read offers two candidate references
    assert (out["mapped"][diag["tie"] == 0] != 0).mean() > 0.99


def test_usize_wrap_reads_print_a_wrapped_column(oracle, simlib):
    """find_coords in usize (src/mers.rs:131-183): a forward run keyed under the SHORT reference that ends on the long one has
    r_end beyond the short reference's end, `r_len - r_end - 1` wraps (release build) and column 4 becomes 2^64 - something."""
    g, off, names, bases, offs, ps = D.usize_wrap_case(oracle, simlib)
    po = oracle.params(**ps)
    ox = oracle.Index()
    ox.build_mt(g, off, names, po, 2)
    out, diag = ox.map_batch_diag(bases, offs, po, threads=4)
    wrapped = np.nonzero((out["mapped"] != 0) & (out["q_end"] >= (1 << 63)))[0]
    assert wrapped.size >= 10 and (diag["quirk_cross_ref"][wrapped] > 0).all() and (out["ref_id"][wrapped] == 0).all()
    i = int(wrapped[0])
    assert int(out["r_end"][i]) == int(off[1]) - 1          # clipped at the short reference's end
    line = oracle.paf_lines(ox, ["q%d" % j for j in range(offs.size - 1)], out)
    assert any(int(ln.split("\t")[3]) >= (1 << 63) for ln in line)
