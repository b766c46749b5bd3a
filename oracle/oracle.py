"""ctypes binding of the CPU oracle (oracle/mapquik_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (mapquik_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmapquik_oracle.so")


def build(force=False):
    """Compile the C oracle with gcc (no reference sources involved)."""
    src = [os.path.join(_HERE, f) for f in ("mapquik_oracle.c", "mapquik_oracle.h")]
    if not force and os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src):
        return _LIB
    subprocess.check_call(
        ["gcc", "-O3", "-march=x86-64-v3", "-std=c11", "-fPIC", "-shared", "-o", _LIB, src[0], "-lpthread"], cwd=_HERE
    )
    return _LIB


class Params(C.Structure):
    _fields_ = [("k", C.c_uint64), ("l", C.c_uint64), ("density", C.c_double), ("use_hpc", C.c_int),
                ("c", C.c_uint64), ("s", C.c_uint64), ("g", C.c_uint64)]


kminmer_dtype = np.dtype([("hash", "<u8"), ("start", "<u8"), ("end", "<u8"), ("offset", "<u8"), ("rev", "<i4"), ("_pad", "<i4")])
minimizer_dtype = np.dtype([("pos", "<u8"), ("hash", "<u8")])
entry_dtype = np.dtype([("id", "<u8"), ("start", "<u8"), ("end", "<u8"), ("offset", "<u8"), ("rc", "<i4"), ("_pad", "<i4")])
match_dtype = np.dtype([("q_start", "<u8"), ("q_end", "<u8"), ("r_start", "<u8"), ("r_end", "<u8"), ("count", "<u8"),
                        ("rc", "<i4"), ("_pad", "<i4")])
coords_dtype = np.dtype([("rc", "<i4"), ("_pad", "<i4"), ("q_start", "<u8"), ("q_end", "<u8"), ("r_start", "<u8"),
                         ("r_end", "<u8"), ("score", "<u8"), ("mapq", "<u8")])
paf_dtype = np.dtype([("mapped", "<i4"), ("rc", "<i4"), ("ref_id", "<u8"), ("q_len", "<u8"), ("q_start", "<u8"),
                      ("q_end", "<u8"), ("r_len", "<u8"), ("r_start", "<u8"), ("r_end", "<u8"), ("score", "<u8"),
                      ("mapq", "<u8")])

diag_dtype = np.dtype([(n, "<u8") for n in ("n_kminmers", "n_hits", "n_matches", "n_candidates", "tie", "quirk_ext", "quirk_cross_ref",
                                            "rc_ext", "check_fail", "i32_wrap", "multi_match_refs", "filtered_out", "clip_start", "clip_end")])

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    vp, u64, sz = C.c_void_p, C.c_uint64, C.c_size_t
    PP = C.POINTER(Params)
    L.mqo_params_default.argtypes = [PP]
    L.mqo_nt_seed.restype = u64
    L.mqo_nt_seed.argtypes = [C.c_uint8]
    for f in (L.mqo_ntf64, L.mqo_ntr64, L.mqo_ntc64):
        f.restype = u64
        f.argtypes = [C.c_char_p, sz, sz]
    L.mqo_density_bound.restype = u64
    L.mqo_density_bound.argtypes = [C.c_double]
    L.mqo_siphash.restype = u64
    L.mqo_siphash.argtypes = [C.c_char_p, sz, u64, u64, C.c_int, C.c_int]
    L.mqo_tuple_hash.restype = u64
    L.mqo_tuple_hash.argtypes = [vp, sz]
    L.mqo_tuple_hash_fast.restype = u64
    L.mqo_tuple_hash_fast.argtypes = [vp, sz]
    for f in (L.mqo_minimizers, L.mqo_minimizers_naive, L.mqo_kminmers):
        f.restype = sz
        f.argtypes = [vp, sz, PP, vp, sz]
    L.mqo_index_new.restype = vp
    L.mqo_index_free.argtypes = [vp]
    L.mqo_index_add.argtypes = [vp, u64, u64, u64, u64, u64, C.c_int]
    L.mqo_index_get.restype = vp
    L.mqo_index_get.argtypes = [vp, u64]
    L.mqo_index_count.restype = u64
    L.mqo_index_count.argtypes = [vp]
    L.mqo_index_keys.restype = u64
    L.mqo_index_keys.argtypes = [vp]
    L.mqo_ref_extract.restype = u64
    L.mqo_ref_extract.argtypes = [vp, u64, vp, sz, PP]
    L.mqo_index_set_ref.argtypes = [vp, u64, C.c_char_p, u64]
    L.mqo_index_ref_len.restype = u64
    L.mqo_index_ref_len.argtypes = [vp, u64]
    L.mqo_index_ref_name.restype = C.c_char_p
    L.mqo_index_ref_name.argtypes = [vp, u64]
    L.mqo_index_n_refs.restype = u64
    L.mqo_index_n_refs.argtypes = [vp]
    L.mqo_index_build_mt.restype = u64
    L.mqo_index_build_mt.argtypes = [vp, vp, vp, C.c_uint32, PP, C.c_int, vp]
    L.mqo_match_check.restype = C.c_int
    L.mqo_match_check.argtypes = [vp, vp, vp, vp]
    L.mqo_check_match_compatible.restype = C.c_int
    L.mqo_check_match_compatible.argtypes = [vp, vp, u64]
    L.mqo_chain_matches_explicit.restype = sz
    L.mqo_chain_matches_explicit.argtypes = [vp, vp, vp, sz, vp, vp, sz]
    L.mqo_chain_get_match.restype = C.c_int
    L.mqo_chain_get_match.argtypes = [vp, sz, PP, vp]
    L.mqo_best_of.restype = C.c_int
    L.mqo_best_of.argtypes = [vp, sz]
    L.mqo_find_coords.argtypes = [u64, u64, u64, vp, vp]
    L.mqo_format_paf.restype = C.c_int
    L.mqo_format_paf.argtypes = [C.c_char_p, C.c_char_p, vp, C.c_char_p, sz]
    L.mqo_find_matches.argtypes = [vp, vp, sz, PP, vp]
    L.mqo_map_batch.argtypes = [vp, vp, vp, C.c_uint32, PP, C.c_int, vp]
    L.mqo_map_batch_diag.argtypes = [vp, vp, vp, C.c_uint32, PP, C.c_int, vp, vp]
    _lib = L
    return L


def params(k=5, l=31, density=0.01, use_hpc=True, c=4, s=11, g=2000):
    return Params(k, l, density, 1 if use_hpc else 0, c, s, g)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _seq(seq):
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def minimizers(seq, p, naive=False):
    s = _seq(seq)
    f = lib().mqo_minimizers_naive if naive else lib().mqo_minimizers
    n = f(_ptr(s), s.size, C.byref(p), None, 0)
    out = np.zeros(n, dtype=minimizer_dtype)
    if n:
        f(_ptr(s), s.size, C.byref(p), _ptr(out), n)
    return out


def kminmers(seq, p):
    s = _seq(seq)
    n = lib().mqo_kminmers(_ptr(s), s.size, C.byref(p), None, 0)
    out = np.zeros(n, dtype=kminmer_dtype)
    if n:
        lib().mqo_kminmers(_ptr(s), s.size, C.byref(p), _ptr(out), n)
    return out


class Index:
    """src/index.rs Index + ReadOnlyIndex + the ref_map of src/closures.rs:30."""

    def __init__(self):
        self.h = lib().mqo_index_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().mqo_index_free(self.h)
            self.h = None

    def add(self, h, id, start, end, offset, rc):
        lib().mqo_index_add(self.h, h, id, start, end, offset, 1 if rc else 0)

    def get(self, h):
        p = lib().mqo_index_get(self.h, h)
        if not p:
            return None
        return np.frombuffer((C.c_char * entry_dtype.itemsize).from_address(p), dtype=entry_dtype)[0].copy()

    def count(self):
        return lib().mqo_index_count(self.h)

    def keys(self):
        return lib().mqo_index_keys(self.h)

    def add_ref(self, ref_idx, name, seq, p):
        s = _seq(seq)
        n = lib().mqo_ref_extract(self.h, ref_idx, _ptr(s), s.size, C.byref(p))
        lib().mqo_index_set_ref(self.h, ref_idx, name.encode(), s.size)
        return n

    def set_ref(self, ref_idx, name, length):
        lib().mqo_index_set_ref(self.h, ref_idx, name.encode(), length)

    def build_mt(self, bases, offsets, names, p, threads):
        bases = _seq(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        cnt = np.zeros(n, dtype=np.uint64)
        lib().mqo_index_build_mt(self.h, _ptr(bases), _ptr(offsets), n, C.byref(p), threads, _ptr(cnt))
        for i in range(n):
            self.set_ref(i, names[i], int(offsets[i + 1] - offsets[i]))
        return cnt

    def ref_name(self, i):
        return lib().mqo_index_ref_name(self.h, i).decode()

    def ref_len(self, i):
        return lib().mqo_index_ref_len(self.h, i)

    def find_matches(self, seq, p):
        s = _seq(seq)
        out = np.zeros(1, dtype=paf_dtype)
        lib().mqo_find_matches(self.h, _ptr(s), s.size, C.byref(p), _ptr(out))
        return out[0]

    def map_batch(self, bases, offsets, p, threads=1):
        bases = _seq(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        out = np.zeros(n, dtype=paf_dtype)
        lib().mqo_map_batch(self.h, _ptr(bases), _ptr(offsets), n, C.byref(p), threads, _ptr(out))
        return out

    def map_batch_diag(self, bases, offsets, p, threads=1):
        """map_batch plus per-read branch counters (diag_dtype): which sharp edges of the reference each read reached."""
        bases = _seq(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        out = np.zeros(n, dtype=paf_dtype)
        diag = np.zeros(n, dtype=diag_dtype)
        lib().mqo_map_batch_diag(self.h, _ptr(bases), _ptr(offsets), n, C.byref(p), threads, _ptr(out), _ptr(diag))
        return out, diag


def format_paf(q_id, r_name, paf):
    rec = np.zeros(1, dtype=paf_dtype)
    rec[0] = paf
    buf = C.create_string_buffer(1024)
    lib().mqo_format_paf(q_id.encode(), r_name.encode(), _ptr(rec), buf, 1024)
    return buf.value.decode()


def paf_lines(index, names, pafs):
    """PAF text in input order, unmapped reads skipped (src/closures.rs:117-123)."""
    out = []
    for name, rec in zip(names, pafs):
        if rec["mapped"]:
            out.append(format_paf(name, index.ref_name(int(rec["ref_id"])), rec))
    return out
