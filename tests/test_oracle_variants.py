"""The oracle's diagnostic variants of the UNPINNED seeding decisions (mqo_set_variant): every bit changes the
k-min-mer tuples on an input built for it, and variant 0 is what everything else in this repo uses.  tools/check_against_upstream.sh
diffs the real crate's tuples against twelve combinations of them; this test is what makes a match there meaningful."""
import random

import numpy as np
import pytest


@pytest.fixture()
def O():
    from oracle import oracle as O
    yield O
    O.lib().mqo_set_variant(0)


def _tuples(O, seq, p, v):
    O.lib().mqo_set_variant(v)
    km = O.kminmers(seq, p)
    O.lib().mqo_set_variant(0)
    return [(int(x["start"]), int(x["end"]), int(x["offset"]), int(x["rev"]), int(x["hash"])) for x in km]


def _rand_seq(n, seed, runs=False):
    rng = random.Random(seed)
    if not runs:
        return "".join(rng.choice("ACGT") for _ in range(n)).encode()
    out = []
    while len(out) < n:
        out.extend(rng.choice("ACGT") * rng.choice([1, 1, 2, 3, 5]))
    return "".join(out[:n]).encode()


def test_variant_zero_is_the_default(O):
    assert O.lib().mqo_get_variant() == 0
    seq = _rand_seq(20000, 1, runs=True)
    p = O.params()
    assert _tuples(O, seq, p, 0) == [(int(x["start"]), int(x["end"]), int(x["offset"]), int(x["rev"]), int(x["hash"])) for x in O.kminmers(seq, p)]


def _hash_equal_to_its_bound(O, want_f32_below):
    """A sequence of one l-mer (l = 12, k = 1, no HPC) whose canonical hash v has its low 11 bits clear, so that density = v / 2^64
    is exact in f64 and the 64-bit bound is v itself; want_f32_below: (float)density rounds DOWN, so the f32 bound is below v."""
    l = 12
    for seed in range(400000):
        s = _rand_seq(l, 1000 + seed)
        v = int(O.lib().mqo_ntc64(s, 0, l)) if hasattr(O.lib(), "mqo_ntc64") else None
        if v is None or v & 0x7FF or v == 0:
            continue
        d = v / 2.0 ** 64
        if int(O.lib().mqo_density_bound(d)) != v:
            continue
        if want_f32_below and float(np.float32(d)) >= d:
            continue
        return s, d, l
    pytest.skip("no suitable l-mer found")


def test_bit1_strict_less_than_drops_the_hash_equal_to_the_bound(O):
    import ctypes as C
    O.lib().mqo_ntc64.restype = C.c_uint64
    O.lib().mqo_ntc64.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t]
    O.lib().mqo_density_bound.restype = C.c_uint64
    O.lib().mqo_density_bound.argtypes = [C.c_double]
    s, d, l = _hash_equal_to_its_bound(O, False)
    p = O.params(k=1, l=l, density=d, use_hpc=False)
    assert len(_tuples(O, s, p, 0)) == 1 and len(_tuples(O, s, p, 1)) == 0


def test_bit2_f32_bound_moves_the_bound(O):
    import ctypes as C
    O.lib().mqo_ntc64.restype = C.c_uint64
    O.lib().mqo_ntc64.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t]
    O.lib().mqo_density_bound.restype = C.c_uint64
    O.lib().mqo_density_bound.argtypes = [C.c_double]
    s, d, l = _hash_equal_to_its_bound(O, True)
    p = O.params(k=1, l=l, density=d, use_hpc=False)
    assert len(_tuples(O, s, p, 0)) == 1 and len(_tuples(O, s, p, 2)) == 0


def test_bit4_32bit_hash_selects_other_minimizers(O):
    seq = _rand_seq(30000, 3, runs=True)
    p = O.params()
    a, b = _tuples(O, seq, p, 0), _tuples(O, seq, p, 4)
    assert len(a) > 100 and len(b) > 100 and a != b
    assert all(t[4] != 0 for t in b)
    # its bound variants move with it
    assert _tuples(O, seq, p, 4 | 2) != a


def test_bit8_position_is_the_run_end(O):
    seq = _rand_seq(30000, 4, runs=True)
    p = O.params()
    a, b = _tuples(O, seq, p, 0), _tuples(O, seq, p, 8)
    assert len(a) == len(b) > 100
    assert [t[2:] for t in a] == [t[2:] for t in b]          # same minimizers, offsets, orientation, hashes
    assert all(y[0] >= x[0] and y[1] >= x[1] for x, y in zip(a, b)) and any(y[0] > x[0] for x, y in zip(a, b))
    # without homopolymer runs the two conventions agree
    flat = b"ACGT" * 5000
    rng = random.Random(9)
    flat = bytes(rng.choice(b"ACGT") for _ in range(20000))
    flat = bytes(c for i, c in enumerate(flat) if i == 0 or c != flat[i - 1])
    assert _tuples(O, flat, p, 0) == _tuples(O, flat, p, 8)


def test_bit16_end_from_the_compressed_window(O):
    seq = _rand_seq(30000, 5, runs=True)
    p = O.params()
    a, b = _tuples(O, seq, p, 0), _tuples(O, seq, p, 16)
    assert len(a) == len(b) > 100
    assert [(t[0],) + t[2:] for t in a] == [(t[0],) + t[2:] for t in b]
    assert any(y[1] != x[1] for x, y in zip(a, b))           # raw l underestimates a window that holds runs
    assert all(y[1] >= x[1] for x, y in zip(a, b))


def test_bit32_palindromic_tuple_orientation(O):
    seq = _rand_seq(30000, 6, runs=True)
    p1 = O.params(k=1)
    a, b = _tuples(O, seq, p1, 0), _tuples(O, seq, p1, 32)
    assert len(a) > 100 and all(t[3] == 0 for t in a) and all(t[3] == 1 for t in b)   # a 1-tuple is its own reverse
    assert [t[:3] + t[4:] for t in a] == [t[:3] + t[4:] for t in b]
    p5 = O.params()
    assert _tuples(O, seq, p5, 0) == _tuples(O, seq, p5, 32)   # k = 5: no palindromic tuples in random sequence


def test_the_twelve_combinations_of_the_upstream_check_are_distinct_readings(O):
    seq = _rand_seq(40000, 7, runs=True)
    p = O.params()
    combos = [0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 24, 32]
    sigs = {}
    for v in combos:
        t = _tuples(O, seq, p, v)
        assert len(t) > 100
        sigs[v] = hash(tuple(t))
    # on ordinary input `<` vs `<=`, the f32 bound and the palindrome rule change nothing (they differ on measure-zero events);
    # the hash width and the two position conventions each give their own tuples
    assert sigs[0] == sigs[1] == sigs[32]
    assert len({sigs[0], sigs[4], sigs[8], sigs[16], sigs[24]}) == 5


def _kh_fast_py(m):
    """mqo_tuple_hash_fast restated: the product's MQ_FLAG_FAST_KH mixer (include/mapquik_hip.h)"""
    M = (1 << 64) - 1
    rotl = lambda x, r: ((x << r) | (x >> (64 - r))) & M
    x, y = 0x736f6d6570736575 ^ len(m), 0x646f72616e646f6d
    for w in m:
        x ^= w
        x = (x + y) & M
        y = rotl(y, 13) ^ x
        x = rotl(x, 32)
    x ^= 0xFF
    for r in (17, 21, 13, 16, 17, 21):
        x = (x + y) & M
        y = rotl(y, r) ^ x
        x = rotl(x, 32)
    return x ^ y


def test_bit64_fast_tuple_hash_changes_the_hash_and_nothing_else(O):
    """Variant bit 64 is the product's opt-in MQ_FLAG_FAST_KH, not a reading of the crate: the same tuples (positions, offsets, strands), another
    hash value, the same partition of the tuples by hash -- hence the same index hits and the same PAF (the reference uses the hash through
    equality only: src/index.rs:100-104,118-126)."""
    import ctypes as C
    rng = random.Random(3)
    for k in (1, 2, 5, 7, 8, 13, 32):
        m = [rng.getrandbits(58) for _ in range(k)]
        arr = (C.c_uint64 * k)(*m)
        assert int(O.lib().mqo_tuple_hash_fast(arr, k)) == _kh_fast_py(m)
        O.lib().mqo_set_variant(64)
        assert int(O.lib().mqo_tuple_hash(arr, k)) == _kh_fast_py(m)
        O.lib().mqo_set_variant(0)
        assert int(O.lib().mqo_tuple_hash(arr, k)) != _kh_fast_py(m)
    seq = _rand_seq(60000, 9, runs=True) + _rand_seq(3000, 10, runs=True) * 3  # repeats: equal tuples exist
    p = O.params()
    t0, t1 = _tuples(O, seq, p, 0), _tuples(O, seq, p, 64)
    assert len(t0) == len(t1) > 300 and [x[:4] for x in t0] == [x[:4] for x in t1]
    assert sum(1 for a, b in zip(t0, t1) if a[4] != b[4]) == len(t0)
    g0, g1 = {}, {}
    for i, (a, b) in enumerate(zip(t0, t1)):
        g0.setdefault(a[4], []).append(i)
        g1.setdefault(b[4], []).append(i)
    assert sorted(g0.values()) == sorted(g1.values()) and any(len(v) > 1 for v in g0.values())
    # and the mapped result: every PAF column equal
    ref = _rand_seq(200000, 21, runs=True)
    reads = [ref[a:a + 9000] for a in range(1000, 180000, 7000)]
    outs = []
    for v in (0, 64):
        O.lib().mqo_set_variant(v)
        ox = O.Index()
        ox.add_ref(0, "chr1", ref, p)
        bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
        offs = np.zeros(len(reads) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(r) for r in reads])
        outs.append(ox.map_batch(bases, offs, p, threads=2).tobytes())
        O.lib().mqo_set_variant(0)
    assert outs[0] == outs[1]
