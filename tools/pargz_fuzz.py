#!/usr/bin/env python3
"""Randomized check of the many-thread gzip member inflater (par_gzip.hpp, through feeder_dump's `inflate` mode) against zlib:
tools/pargz_fuzz.py [cases] [seed].  Inputs: read-like text with varied repetition, compressors of all levels / strategies / memory
levels / flush patterns, segment sizes from 300 bytes up, 1-8 threads, small symbol buffers.  Prints a summary line; exit 1 on a difference."""
import os
import random
import subprocess
import sys
import tempfile
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mapquik_amd import build as B  # noqa: E402


def text(rng):
    kind = rng.randrange(5)
    n = rng.choice([2_000, 30_000, 200_000, 900_000])
    if kind == 0:  # random DNA lines
        return ("\n".join("".join(rng.choice("ACGT") for _ in range(rng.choice([60, 80, 5000]))) for _ in range(max(1, n // 200)))).encode()
    if kind == 1:  # reads off a tiny genome: long matches, far and near
        g = "".join(rng.choice("ACGT") for _ in range(rng.choice([500, 20_000, 100_000])))
        out = []
        while sum(map(len, out)) < n:
            L = rng.choice([50, 300, 3000])
            s = rng.randrange(0, max(1, len(g) - L))
            out.append(">r%d len=%d\n%s\n" % (len(out), L, g[s:s + L]))
        return "".join(out).encode()
    if kind == 2:  # FASTQ with skewed qualities
        out = []
        while sum(map(len, out)) < n:
            L = rng.choice([100, 150, 8000])
            out.append("@x%d\n%s\n+\n%s\n" % (len(out), "".join(rng.choice("ACGTN") for _ in range(L)), "".join(rng.choice("FFFFFFF:#,") for _ in range(L))))
        return "".join(out).encode()
    if kind == 3:  # very repetitive
        unit = "".join(rng.choice("ACGT\n") for _ in range(rng.choice([1, 3, 17, 300])))
        return (unit * (n // len(unit) + 1)).encode()
    return bytes(rng.choice(b"ACGTacgtNn\n>@+ \t0123456789") for _ in range(n))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    tool = B.build_feeder_dump()
    multi = 0
    with tempfile.TemporaryDirectory() as td:
        for case in range(cases):
            members, want = [], b""
            for _ in range(rng.choice([1, 1, 1, 2, 4])):
                d = text(rng)
                c = zlib.compressobj(rng.choice([0, 1, 2, 4, 6, 9]), zlib.DEFLATED, 31, rng.choice([1, 4, 8, 9]),
                                     rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
                step = rng.choice([0, 0, 5_000, 100_000])
                blob = b""
                if step:
                    for a in range(0, len(d), step):
                        blob += c.compress(d[a:a + step]) + c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))
                    blob += c.flush()
                else:
                    blob = c.compress(d) + c.flush()
                members.append(blob)
                want += d
            p = os.path.join(td, "c.gz")
            with open(p, "wb") as f:
                f.write(b"".join(members))
            seg = rng.choice([300, 2_000, 10_000, 60_000, 1 << 20])
            th = rng.randrange(1, 9)
            env = dict(os.environ, PARGZ_MINSEG=str(rng.choice([100, seg // 4 + 1])))
            if rng.random() < 0.3:
                env["PARGZ_RATIO"] = str(rng.choice([1, 2, 8]))
            if rng.random() < 0.3:
                env["PARGZ_ZLIB_CRC"] = "1"
            r = subprocess.run([tool, p, "inflate", str(seg), str(th)], capture_output=True, timeout=300, env=env)
            if r.returncode != 0 or r.stdout != want:
                keep = "/tmp/pargz_fuzz_fail_%d_%d.gz" % (seed, case)
                os.replace(p, keep)
                print("FAIL case %d seed %d: rc %d, %d vs %d bytes, seg %d threads %d env %s; input kept at %s\n%s" % (
                    case, seed, r.returncode, len(r.stdout), len(want), seg, th, {k: v for k, v in env.items() if k.startswith("PARGZ")}, keep, r.stderr.decode()[-500:]))
                sys.exit(1)
            multi += any(int(ln.split()[3]) > 1 for ln in r.stderr.decode().split("\n") if ln.startswith("rounds "))
    print("pargz_fuzz: %d cases (seed %d) identical to zlib; %d of them with segments that followed each other" % (cases, seed, multi))


if __name__ == "__main__":
    main()
