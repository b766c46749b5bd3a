"""Evaluation tools around the path: the PAF concordance measure of the reference's experiments (experiments/intersect_pafs.py)."""
import subprocess
import sys
import os

from tools import paf_concordance as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(read, tgt, tlen, ts, te, qlen=1000):
    return "\t".join(map(str, [read, qlen, 0, qlen - 1, "+", tgt, tlen, ts, te, 50, tlen, 60])) + "\n"


def test_concordance_counts_and_cli(tmp_path):
    a, b = tmp_path / "a.paf", tmp_path / "b.paf"
    a.write_text(_line("r1", "chr1", 10_000_000, 1000, 2000) + _line("r2", "chr1", 10_000_000, 5000, 6000) +
                 _line("r3", "chr2", 5_000_000, 100, 1100) + _line("r4", "chr1", 10_000_000, 7000, 8000))
    b.write_text(_line("r1", "chr1", 10_000_000, 1100, 2100) +     # overlaps: 900 / 1100 > 0.1
                 _line("r2", "chr1", 10_000_000, 900_000, 901_000) +  # same target, far away
                 _line("r3", "chr1", 10_000_000, 100, 1100) +        # other target
                 _line("r5", "chr1", 10_000_000, 1, 2))
    c = pc.concordance(pc.parse_paf(a), pc.parse_paf(b))
    assert c == dict(concordant=1, discordant=2, different_target=1, only_in_1=1, only_in_2=1)
    # the reference's column reading (target length as "start") calls r2 concordant too
    cu = pc.concordance(pc.parse_paf(a, True), pc.parse_paf(b, True))
    assert cu["concordant"] == 2 and cu["different_target"] == 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "paf_concordance.py"), str(a), str(b)], capture_output=True, text=True)
    assert r.returncode == 0 and "Number of concordant mappings: 1 (25.0% of" in r.stdout
    assert "different chromosome: 1" in r.stdout


def test_overlap_ratio_edges():
    assert pc.overlap_ratio(0, 100, 100, 200) == 0.0
    assert pc.overlap_ratio(0, 100, 50, 150) == 50 / 150
    assert pc.overlap_ratio(200, 100, 150, 250) == 50 / 150  # reversed coordinates
    assert pc.overlap_ratio(5, 5, 5, 5) == 1.0
