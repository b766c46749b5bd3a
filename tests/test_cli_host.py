"""Host driver pieces that need no GPU: flag/log surface (src/main.rs:77-240) and the FASTX reader."""
import gzip
import os

import pytest

from mapquik_amd import cli


def test_default_banner_matches_reference_lines():
    opt = cli.build_parser().parse_args(["reads.fq", "--reference", "ref.fa"])
    lines, st = cli.banner_lines(opt)
    assert lines == [
        "Reference file: ref.fa", "Format: FASTA",
        "Warning: Using default k value (5).", "Warning: Using default l value (31).",
        "Warning: Using default buffer size (1X).", "Warning: Using default queue length (200).",
        "Warning: Using default density value (1%).", "Warning: Using default number of threads (8).",
        "Warning: Using default minimum chain length (4).", "Warning: Using default minimum number of matching seeds (11).",
        "Warning: Using default maximum seed gap difference (2000).",
        "Warning: Using default output prefix (mapquik-k5-d0.01-l31).", "Using HPC ntHash, with SIMD"]
    assert st["prefix"] == "mapquik-k5-d0.01-l31" and st["reads_fasta"] is False and st["ref_fasta"] is True


def test_example_command_line_of_the_reference():
    # example/run_ecoli.sh:26
    argv = "nearperfect-ecoli.100.fa --reference ecoli.genome.fa --debug -k 8 -d 0.01 -l 16 -p mapquik -g 100 --threads 11".split()
    lines, st = cli.banner_lines(cli.build_parser().parse_args(argv))
    assert (st["k"], st["l"], st["g"], st["prefix"], st["density"]) == (8, 16, 100, "mapquik", 0.01)
    assert lines[:4] == ["Input file: nearperfect-ecoli.100.fa", "Format: FASTA", "Reference file: ecoli.genome.fa", "Format: FASTA"]
    assert "Warning: Using default k value (5)." not in lines and "Warning: Using default buffer size (1X)." in lines
    lines2, _ = cli.banner_lines(cli.build_parser().parse_args(argv + ["--nohpc", "--nosimd"]))
    assert lines2[-1] == "Using regular ntHash (not HPC), scalar"


def test_fasta_detection_by_name():
    for n, want in (("a.fa", True), ("a.fasta", True), ("a.fna", True), ("a.fa.gz", True), ("a.fasta.gz", True), ("a.fq", False),
                    ("a.fastq.gz", False), ("reads.fa2", False)):
        assert cli.is_fasta_name(n) is want, n


def test_rust_formatting_helpers():
    assert cli.rust_duration(19.98) == "19.98s" and cli.rust_duration(0.5) == "500ms" and cli.rust_duration(229.5) == "229.5s"
    assert cli.rust_duration(0.0000015) == "1.5µs" and cli.rust_duration(2.0) == "2s" and cli.rust_duration(1.234567891) == "1.234567891s"
    assert cli.rust_float(1.0) == "1" and cli.rust_float(0.01) == "0.01" and cli.rust_float(2.5) == "2.5"


def test_fastx_reader(tmp_path):
    fa = tmp_path / "r.fa"
    fa.write_text(">s1 desc here\nacgt\nNNAC\n>s2\nTTTT\n\n>s3\n")
    assert list(cli.read_fastx(str(fa), True)) == [("s1", b"ACGTNNAC"), ("s2", b"TTTT"), ("s3", b"")]
    fq = tmp_path / "r.fastq.gz"
    with gzip.open(fq, "wb") as w:
        w.write(b"@q1 x\nacgtn\n+\nIIIII\n@q2\nGG\n+\nII\n")
    assert list(cli.read_fastx(str(fq), False)) == [("q1", b"ACGTN"), ("q2", b"GG")]
    with pytest.raises(SystemExit):
        list(cli.read_fastx(str(tmp_path / "x.lz4"), True))


def test_missing_arguments_exit_like_the_reference():
    with pytest.raises(SystemExit) as e:
        cli.main([])
    assert "Please specify an input file." in str(e.value)
    with pytest.raises(SystemExit) as e:
        cli.main(["reads.fa"])
    assert "Please specify a reference file." in str(e.value)


def _native():
    from mapquik_amd import build
    return build.build_cli()


def test_native_cli_builds_and_mirrors_banner(tmp_path):
    """The C++ driver (mapquik_amd/csrc/host) prints the same pre-run lines as the reference (and as the Python driver)."""
    import subprocess
    exe = _native()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 101 and "Please specify an input file." in r.stderr
    r = subprocess.run([exe, "reads.fq"], capture_output=True, text=True)
    assert r.returncode == 101 and "Please specify a reference file." in r.stderr
    argv = ["reads.fq", "--reference", "ref.fa", "-p", str(tmp_path / "o")]
    r = subprocess.run([exe] + argv, capture_output=True, text=True, cwd=tmp_path)
    want, _ = cli.banner_lines(cli.build_parser().parse_args(argv))
    got = r.stdout.splitlines()
    assert got[:len(want)] == want
    argv = "nearperfect-ecoli.100.fa --reference ecoli.genome.fa --debug -k 8 -d 0.01 -l 16 -p mapquik -g 100 --threads 11 --nohpc".split()
    r = subprocess.run([exe] + argv, capture_output=True, text=True, cwd=tmp_path)
    want, _ = cli.banner_lines(cli.build_parser().parse_args(argv))
    assert r.stdout.splitlines()[:len(want)] == want
