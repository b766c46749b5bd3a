"""Oracle vs the committed known-answer vectors (tests/golden/kat_intree.json).

In-tree stages (src/match.rs, src/chain.rs, src/mers.rs, src/index.rs): hand-derived KATs.
Restated third-party arithmetic: published ntHash-1 and SipHash vectors.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_intree.json")


@pytest.fixture(scope="module")
def kat():
    return json.load(open(GOLD))


def _q(O, qs):
    a = np.zeros(len(qs), dtype=O.kminmer_dtype)
    for i, q in enumerate(qs):
        a[i]["start"], a[i]["end"], a[i]["rev"], a[i]["offset"], a[i]["hash"] = q["start"], q["end"], q["rev"], i, 1000 + i
    return a


def _r(O, rs):
    a = np.zeros(len(rs), dtype=O.entry_dtype)
    hit = np.zeros(len(rs), dtype=np.uint8)
    for i, r in enumerate(rs):
        if r is None:
            continue
        hit[i] = 1
        for f in ("id", "start", "end", "offset", "rc"):
            a[i][f] = r[f]
    return a, hit


def _matches(O, ms):
    a = np.zeros(len(ms), dtype=O.match_dtype)
    for i, m in enumerate(ms):
        for f in ("q_start", "q_end", "r_start", "r_end", "count", "rc"):
            a[i][f] = m[f]
    return a


def _coords(O, c):
    a = np.zeros(1, dtype=O.coords_dtype)
    for f in ("rc", "q_start", "q_end", "r_start", "r_end", "score", "mapq"):
        a[0][f] = c[f]
    return a


def _chain_matches(O, q, r):
    qa = _q(O, q)
    ra, hit = _r(O, r)
    out = np.zeros(len(q), dtype=O.match_dtype)
    ref = np.zeros(len(q), dtype=np.uint64)
    n = O.lib().mqo_chain_matches_explicit(O._ptr(qa), O._ptr(ra), O._ptr(hit), len(q), O._ptr(out), O._ptr(ref), len(q))
    return out[:n], ref[:n]


def _get_match(O, ms, c=4, s=11, g=2000):
    p = O.params(c=c, s=s, g=g)
    out = np.zeros(1, dtype=O.coords_dtype)
    ok = O.lib().mqo_chain_get_match(O._ptr(ms), len(ms), C.byref(p), O._ptr(out))
    return ok, out[0]


def test_nthash_published_vectors(oracle, kat):
    L = oracle.lib()
    for b, v in kat["nthash_seeds"].items():
        assert L.mqo_nt_seed(ord(b)) == int(v, 16)
    for v in kat["nthash"]:
        s = v["seq"].encode()
        if "fwd" in v:
            assert L.mqo_ntf64(s, 0, v["l"]) == int(v["fwd"], 16)
            assert L.mqo_ntr64(s, 0, v["l"]) == int(v["rev"], 16)
        assert L.mqo_ntc64(s, 0, v["l"]) == int(v["canon"], 16)


def test_nthash_reverse_complement_symmetry(oracle):
    L = oracle.lib()
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    rng = np.random.default_rng(7)
    for l in (1, 5, 16, 31, 32, 63, 64, 70):
        s = rng.choice([65, 67, 71, 84], size=l).astype(np.uint8)
        rc = bytes(comp[int(b)] for b in s[::-1])
        assert L.mqo_ntf64(s.tobytes(), 0, l) == L.mqo_ntr64(rc, 0, l)
        assert L.mqo_ntc64(s.tobytes(), 0, l) == L.mqo_ntc64(rc, 0, l)


def test_siphash_reference_vectors(oracle, kat):
    L = oracle.lib()
    k0 = int.from_bytes(bytes(range(8)), "little")
    k1 = int.from_bytes(bytes(range(8, 16)), "little")
    for v in kat["siphash24"]:
        n = v["msg_len"]
        assert L.mqo_siphash(bytes(range(n)), n, k0, k1, 2, 4) == int(v["out"], 16)


def test_density_bound(oracle):
    L = oracle.lib()
    assert L.mqo_density_bound(0.0) == 0
    assert L.mqo_density_bound(-1.0) == 0
    assert L.mqo_density_bound(1.0) == 2**64 - 1
    assert L.mqo_density_bound(2.0) == 2**64 - 1
    assert L.mqo_density_bound(0.5) == 2**63
    assert L.mqo_density_bound(0.01) == int(0.01 * float(2**64))


@pytest.mark.parametrize("name", ["C1", "C2", "C3a", "C3b"])
def test_chain_matches_kat(oracle, kat, name):
    v = kat[name]
    ms, ref = _chain_matches(oracle, v["q"], v["r"])
    assert len(ms) == len(v["matches"])
    for m, r, e in zip(ms, ref, v["matches"]):
        assert int(r) == e["ref"]
        for f in ("q_start", "q_end", "r_start", "r_end", "count", "rc"):
            assert int(m[f]) == e[f], (name, f)


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_get_match_and_paf_kat(oracle, kat, name):
    v = kat[name]
    ms, ref = _chain_matches(oracle, v["q"], v["r"])
    ok, c = _get_match(oracle, np.ascontiguousarray(ms))
    assert ok
    for f, e in v["coords"].items():
        assert int(c[f]) == e, (name, f)
    paf = np.zeros(1, dtype=oracle.paf_dtype)
    cc = _coords(oracle, v["coords"])
    oracle.lib().mqo_find_coords(v["q_len"], v["r_len"], int(ref[0]), oracle._ptr(cc), oracle._ptr(paf))
    assert oracle.format_paf(v["q_id"], v["r_name"], paf[0]) == v["paf"]


def test_filter_anchor_kat(oracle, kat):
    for v in kat["C4"]:
        ok, c = _get_match(oracle, _matches(oracle, v["matches"]), c=v["c"], s=v["s"], g=v["g"])
        assert ok
        for f, e in v["coords"].items():
            assert int(c[f]) == e, (v["g"], f)


def test_best_of_kat(oracle, kat):
    for v in kat["C5"]:
        a = np.asarray(v["scores"], dtype=np.uint64)
        assert oracle.lib().mqo_best_of(oracle._ptr(a) if a.size else None, a.size) == v["best"]


def test_find_coords_kat(oracle, kat):
    for v in kat["C6"]:
        paf = np.zeros(1, dtype=oracle.paf_dtype)
        cc = _coords(oracle, v["coords"])
        oracle.lib().mqo_find_coords(v["q_len"], v["r_len"], 0, oracle._ptr(cc), oracle._ptr(paf))
        for f, e in v["out"].items():
            assert int(paf[0][f]) == e, f
        assert int(paf[0]["score"]) == v["coords"]["score"] and int(paf[0]["mapq"]) == v["coords"]["mapq"]


def test_paf_line_shape_quoted_in_reference(oracle, kat):
    v = kat["paf_shape"]
    paf = np.zeros(1, dtype=oracle.paf_dtype)
    cc = _coords(oracle, v["coords"])
    oracle.lib().mqo_find_coords(v["q_len"], v["r_len"], 0, oracle._ptr(cc), oracle._ptr(paf))
    assert oracle.format_paf(v["q_id"], v["r_name"], paf[0]) == v["paf"]


def test_index_semantics_kat(oracle, kat):
    v = kat["index"]
    ix = oracle.Index()
    for h, id_, start, end, offset, rc in v["ops"]:
        ix.add(h, id_, start, end, offset, rc)
    for k, e in v["get"].items():
        got = ix.get(int(k))
        if e is None:
            assert got is None
        else:
            for f, x in e.items():
                assert int(got[f]) == x
    assert ix.count() == v["count"] and ix.keys() == v["keys"]


def test_check_compat_wrapping_i32(oracle):
    """`as i32` truncation in fwd/rc_gap_too_long (src/chain.rs:132-142): positions above 2^31 wrap."""
    a = _matches(oracle, [dict(q_start=100, q_end=600, r_start=2**31 + 1000, r_end=2**31 + 1500, count=2, rc=0),
                          dict(q_start=1000, q_end=1500, r_start=2**31 + 1900, r_end=2**31 + 2400, count=3, rc=0)])
    L = oracle.lib()
    assert L.mqo_check_match_compatible(oracle._ptr(a[0:1]), oracle._ptr(a[1:2]), 2000) == 1
    b = a.copy()
    b[1]["r_start"] = 2**31 + 9000
    assert L.mqo_check_match_compatible(oracle._ptr(b[0:1]), oracle._ptr(b[1:2]), 2000) == 0
