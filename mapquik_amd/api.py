"""ctypes binding of libmapquik_hip.so + host mirror of the reference's interface for the hot path.

Reference seam (Rust): mers::ref_extract (src/mers.rs:15), mers::find_matches (src/mers.rs:77), Index/ReadOnlyIndex
(src/index.rs:73-128), Params (src/main.rs:33-47).  Same names and argument meaning here; errors raise MapquikError
(the reference panics).  There is NO CPU fallback: if the HIP library or a GPU is missing, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MQ_HIT_UNMAPPED, MQ_HIT_MAPPED, MQ_HIT_OVERFLOW = 0, 1, 2
MQ_FLAG_FOLD_CASE = 1
MQ_FLAG_FAST_KH = 2
MQ_FLAG_SEED_VARIANT_SHIFT = 8
MQ_ABI_VERSION = 4  # include/mapquik_hip.h

hit_dtype = np.dtype([("status", "<u4"), ("ref_id", "<u4"), ("rc", "<u4"), ("mapq", "<u4"), ("q_start", "<u4"), ("q_end", "<u4"),
                      ("r_start", "<u4"), ("r_end", "<u4"), ("score", "<u4"), ("n_kminmers", "<u4"), ("q_start_hi", "<u4"), ("q_end_hi", "<u4")])
kminmer_dtype = np.dtype([("hash", "<u8"), ("start", "<u4"), ("end", "<u4"), ("offset", "<u4"), ("rev", "<u4")])
assert hit_dtype.itemsize == 48 and kminmer_dtype.itemsize == 24


def hit_column(hits, name):
    """A PAF column of a hits array as uint64: columns 3 and 4 (q_start, q_end) are 64-bit in mq_hit (low word + *_hi)."""
    v = hits[name].astype(np.uint64)
    if name in ("q_start", "q_end"):
        v = v | (hits[name + "_hi"].astype(np.uint64) << np.uint64(32))
    return v

# every symbol include/mapquik_hip.h (the seam) and include/mapquik_hip_diag.h (measurement / diagnostics) declare
EXPORTS = ["mq_index_set_table_factor", "mq_ctx_submit_fastx", "mq_index_get_params", "mq_index_set_map_params", "mq_index_stage_begin", "mq_index_stage_piece", "mq_index_stage_done", "mq_index_add_ref_staged", "mq_ctx_submit_fasta", "mq_ctx_wait_fasta", "mq_index_reserve", "mq_host_register", "mq_host_unregister",
           "mq_last_error", "mq_abi_version", "mq_device_count", "mq_params_default", "mq_index_new", "mq_index_free",
           "mq_index_add_ref", "mq_index_add_ref_device", "mq_index_finalize", "mq_index_get_stats", "mq_index_ref_info",
           "mq_map_batch", "mq_map_batch_device", "mq_map_reserve", "mq_kminmers_batch", "mq_index_lookup", "mq_format_paf",
           "mq_last_map_ms", "mq_last_map_path_counts", "mq_last_map_order", "mq_host_alloc", "mq_host_free", "mq_index_save", "mq_index_load", "mq_index_clone", "mq_map_probe_stats",
           "mq_ctx_new", "mq_ctx_free", "mq_ctx_map_batch", "mq_ctx_submit", "mq_ctx_submit_spans", "mq_ctx_wait", "mq_ctx_reserve", "mq_ctx_map_batch_device", "mq_ctx_last_map_ms", "mq_probe_rate", "mq_last_stage_clocks", "mq_last_read_cycles", "mq_map_launch_waves", "mq_index_table_alloc_ms"]


class MapquikError(RuntimeError):
    pass


class Params(C.Structure):
    """src/main.rs:33-47; defaults src/main.rs:174-188."""
    _fields_ = [("k", C.c_uint32), ("l", C.c_uint32), ("density", C.c_double), ("use_hpc", C.c_uint32), ("c", C.c_uint32),
                ("s", C.c_uint32), ("g", C.c_uint32), ("flags", C.c_uint32)]

    def __init__(self, k=5, l=31, density=0.01, use_hpc=True, c=4, s=11, g=2000, fold_case=False, seeding_variant=0, fast_kh=False):
        """seeding_variant: MQ_SEEDVAR_* bits (include/mapquik_hip.h), 0 = the frozen reading of the third-party k-min-mer iterator.
        fast_kh: MQ_FLAG_FAST_KH, the opt-in cheap tuple hash (same PAF; mq_kminmer.hash is then not the reference's value)."""
        if not 0 <= int(seeding_variant) < 64:
            raise ValueError("seeding_variant must be 0..63")
        super().__init__(k, l, density, 1 if use_hpc else 0, c, s, g,
                         (MQ_FLAG_FOLD_CASE if fold_case else 0) | (MQ_FLAG_FAST_KH if fast_kh else 0) | (int(seeding_variant) << MQ_FLAG_SEED_VARIANT_SHIFT))

    @property
    def fast_kh(self):
        return bool(self.flags & MQ_FLAG_FAST_KH)

    @property
    def seeding_variant(self):
        return (self.flags >> MQ_FLAG_SEED_VARIANT_SHIFT) & 0x3F


class IndexStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_refs", "n_kminmers", "n_keys", "n_unique", "table_slots", "table_bytes", "slot_bytes")]


_lib = None


def load_library(path=None):
    """Load (building if stale) the HIP library.  Raises MapquikError when it cannot be built or loaded."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("MQ_LIB")  # A/B benchmarking of two builds in one session
    p = path or _build.LIB
    if path is None:
        try:
            _build.build()
        except Exception as e:  # no hipcc on this box: fine if a prebuilt .so travelled with the snapshot
            if not os.path.exists(p):
                raise MapquikError("libmapquik_hip.so is missing and could not be built: %s" % e)
    try:
        L = C.CDLL(p)
    except OSError as e:
        raise MapquikError("cannot load %s: %s" % (p, e))
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.mq_last_error.restype = C.c_char_p
    L.mq_abi_version.restype = C.c_int
    L.mq_device_count.restype = C.c_int
    L.mq_params_default.argtypes = [C.POINTER(Params)]
    L.mq_index_new.restype = vp
    L.mq_index_new.argtypes = [C.POINTER(Params), C.c_int]
    L.mq_index_free.argtypes = [vp]
    L.mq_index_add_ref.restype = C.c_int64
    L.mq_index_add_ref.argtypes = [vp, u32, C.c_char_p, vp, u64]
    L.mq_index_add_ref_device.restype = C.c_int64
    L.mq_index_add_ref_device.argtypes = [vp, u32, C.c_char_p, vp, u64]
    L.mq_index_finalize.restype = C.c_int64
    L.mq_index_finalize.argtypes = [vp]
    L.mq_index_get_stats.argtypes = [vp, C.POINTER(IndexStats)]
    L.mq_index_ref_info.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(u64)]
    L.mq_map_batch.argtypes = [vp, vp, vp, u32, vp]
    L.mq_map_batch_device.argtypes = [vp, vp, vp, u32, u64, vp, vp]
    L.mq_map_reserve.argtypes = [vp, u32, u64]
    L.mq_ctx_new.restype = vp
    L.mq_ctx_new.argtypes = [vp]
    L.mq_ctx_free.argtypes = [vp]
    L.mq_ctx_map_batch.argtypes = [vp, vp, vp, u32, vp]
    L.mq_ctx_submit.argtypes = [vp, vp, vp, u32, vp]
    L.mq_ctx_submit_spans.argtypes = [vp, vp, u64, vp, vp, u32, vp]
    L.mq_ctx_wait.argtypes = [vp]
    abi = L.mq_abi_version()
    if path is None and abi != MQ_ABI_VERSION:
        raise MapquikError("%s has ABI version %d, this binding is for %d: rebuild (python __graft_entry__.py)" % (p, abi, MQ_ABI_VERSION))
    # (an older build given by MQ_LIB / path for an A/B run may lack the newer symbols: each group under the guard of its own first symbol)
    if hasattr(L, "mq_index_reserve"):
        L.mq_index_reserve.argtypes = [vp, u64]
    if hasattr(L, "mq_ctx_submit_fasta"):  # ABI 3 on
        L.mq_host_register.argtypes = [vp, C.c_size_t]
        L.mq_host_unregister.argtypes = [vp]
        L.mq_ctx_submit_fasta.argtypes = [vp, vp, u64, u64]
        L.mq_ctx_wait_fasta.argtypes = [vp, C.POINTER(u32), C.POINTER(vp), C.POINTER(u32), C.POINTER(vp), C.POINTER(u32)]
    if hasattr(L, "mq_ctx_submit_fastx"):
        L.mq_ctx_submit_fastx.argtypes = [vp, vp, u64, u64, u32]
    if hasattr(L, "mq_index_stage_begin"):  # ABI 4
        L.mq_index_set_table_factor.argtypes = [vp, u32]
        L.mq_index_get_params.argtypes = [vp, C.POINTER(Params)]
        L.mq_index_set_map_params.argtypes = [vp, u32, u32, u32, C.c_int]
        L.mq_index_stage_begin.argtypes = [vp, u64]
        L.mq_index_stage_piece.argtypes = [vp, u64, vp, u64, C.POINTER(u64)]
        L.mq_index_stage_done.argtypes = [vp, u64, C.c_int]
        L.mq_index_add_ref_staged.restype = C.c_int64
        L.mq_index_add_ref_staged.argtypes = [vp, u32, C.c_char_p, u64, u64, u64]
    L.mq_ctx_reserve.argtypes = [vp, u32, u64]
    L.mq_ctx_map_batch_device.argtypes = [vp, vp, vp, u32, u64, vp, vp]
    L.mq_ctx_last_map_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.mq_probe_rate.argtypes = [vp, u32, u32, u32, u32, C.POINTER(C.c_float), C.POINTER(u64), C.POINTER(u64)]
    L.mq_kminmers_batch.argtypes = [vp, vp, vp, u32, vp, vp, vp]
    L.mq_index_lookup.argtypes = [vp, vp, u32, vp, vp, vp]
    L.mq_format_paf.argtypes = [vp, C.c_char_p, u64, vp, C.c_char_p, C.c_size_t]
    L.mq_last_map_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.mq_last_stage_clocks.argtypes = [vp, vp]
    if hasattr(L, "mq_last_read_cycles"):
        L.mq_last_read_cycles.argtypes = [vp, u32, vp, vp]
    L.mq_map_probe_stats.argtypes = [vp, vp, vp, u32, u64, vp, C.POINTER(u64), C.POINTER(u64)]
    if hasattr(L, "mq_map_launch_waves"):
        L.mq_map_launch_waves.argtypes = [vp, u32, C.POINTER(u32)]
        L.mq_index_table_alloc_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.mq_index_save.argtypes = [vp, C.c_char_p]
    L.mq_index_clone.restype = vp
    L.mq_index_clone.argtypes = [vp, C.c_int]
    L.mq_index_load.restype = vp
    L.mq_index_load.argtypes = [C.c_char_p, C.c_int]
    L.mq_host_alloc.restype = vp
    L.mq_host_alloc.argtypes = [C.c_size_t]
    L.mq_host_free.argtypes = [vp]
    L.mq_last_map_path_counts.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mq_last_map_order.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if path is None:
        _lib = L
    return L


def _err(L, what):
    return MapquikError("%s: %s" % (what, (L.mq_last_error() or b"").decode(errors="replace")))


class PinnedBuffer:
    """Page-locked host bytes (mq_host_alloc) exposed as a numpy uint8 array: fill it, pass `.array` to Index.map_batch."""

    def __init__(self, nbytes):
        self._L = load_library()
        self._p = self._L.mq_host_alloc(nbytes)
        if not self._p:
            raise _err(self._L, "mq_host_alloc")
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self._p))

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._L.mq_host_free(self._p)
            self._p = None

    __del__ = close


def device_count():
    return load_library().mq_device_count()


def _seq(seq):
    if isinstance(seq, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Index:
    """Index + ReadOnlyIndex (src/index.rs:73-128) + ref_map (src/closures.rs:30), resident in HBM."""

    def __init__(self, params=None, device=0, _handle=None):
        self._L = load_library()
        self.params = params or Params()
        self._h = _handle if _handle is not None else self._L.mq_index_new(C.byref(self.params), device)
        if not self._h:
            raise _err(self._L, "mq_index_new")
        self.device = device

    def save(self, path):
        """Write the finalized index to disk (parameters, reference table, slot table)."""
        if self._L.mq_index_save(self._h, os.fsencode(path)) != 0:
            raise _err(self._L, "mq_index_save")

    @classmethod
    def load(cls, path, device=0):
        """A finalized index read back from Index.save (its parameters: Index.get_params(); the mapping-time ones can be replaced with
        Index.set_map_params)."""
        L = load_library()
        h = L.mq_index_load(os.fsencode(path), device)
        if not h:
            raise _err(L, "mq_index_load")
        return cls(None, device, _handle=h)

    def clone(self, device):
        """A replica of this finalized index on another device (device-to-device copy, no re-indexing)."""
        h = self._L.mq_index_clone(self._h, device)
        if not h:
            raise _err(self._L, "mq_index_clone")
        return Index(self.params, device, _handle=h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mq_index_free(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    def add_ref(self, ref_idx, name, seq):
        """index_mers closure (src/closures.rs:46-51): returns the reference's k-min-mer count."""
        s = _seq(seq)
        n = self._L.mq_index_add_ref(self._h, ref_idx, name.encode(), _p(s), s.size)
        if n < 0:
            raise _err(self._L, "mq_index_add_ref")
        return n

    def add_ref_device(self, ref_idx, name, d_ptr, length):
        n = self._L.mq_index_add_ref_device(self._h, ref_idx, name.encode(), C.c_void_p(d_ptr), length)
        if n < 0:
            raise _err(self._L, "mq_index_add_ref_device")
        return n

    def last_read_cycles(self, n):
        """(cycles, start_ticks) of the n reads of the last probe_stats launch: what each read cost its wave, when it was taken up."""
        cyc = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint64)
        if self._L.mq_last_read_cycles(self._h, n, _p(cyc), _p(st)) != 0:
            raise _err(self._L, "mq_last_read_cycles")
        return cyc, st

    def set_table_factor(self, slots_per_kminmer):
        """Table slots per inserted k-min-mer (default 8; 2 for file-fed, host-bound runs).  Before reserve_table / finalize."""
        if self._L.mq_index_set_table_factor(self._h, int(slots_per_kminmer)) != 0:
            raise _err(self._L, "mq_index_set_table_factor")

    def get_params(self):
        """The parameters the index was built with (a loaded file's own)."""
        p = Params()
        if self._L.mq_index_get_params(self._h, C.byref(p)) != 0:
            raise _err(self._L, "mq_index_get_params")
        return p

    def set_map_params(self, c, s, g, fold_case=False):
        """Params.c / .s / .g (src/chain.rs:132-169) and the case folding act at mapping time only: a loaded index takes the caller's."""
        if self._L.mq_index_set_map_params(self._h, int(c), int(s), int(g), 1 if fold_case else 0) != 0:
            raise _err(self._L, "mq_index_set_map_params")

    def stage_begin(self, total_bytes):
        """The reference file in pieces (mq_index_stage_*): a device buffer of the file's size."""
        if self._L.mq_index_stage_begin(self._h, int(total_bytes)) != 0:
            raise _err(self._L, "mq_index_stage_begin")

    def stage_piece(self, at, piece):
        """Queues the copy of `piece` (uint8 array; page-locked for the full rate) to buffer offset `at`; returns the ticket."""
        s = _seq(piece)
        t = C.c_uint64()
        if self._L.mq_index_stage_piece(self._h, int(at), _p(s), s.size, C.byref(t)) != 0:
            raise _err(self._L, "mq_index_stage_piece")
        self._staged = getattr(self, "_staged", {})
        self._staged[t.value] = s  # the source stays alive until its copy is known to be done (stage_done) or the index is finalized
        return t.value

    def stage_done(self, ticket, wait=True):
        r = self._L.mq_index_stage_done(self._h, int(ticket), 1 if wait else 0)
        if r < 0:
            raise _err(self._L, "mq_index_stage_done")
        if r:  # copies complete in order: every source up to this ticket can go
            st = getattr(self, "_staged", {})
            for t in [t for t in st if t <= int(ticket)]:
                del st[t]
        return bool(r)

    def add_ref_staged(self, ref_idx, name, at, length, after_ticket=None):
        """ref_extract of the staging buffer's [at, at + length), behind piece `after_ticket` (None: every piece issued so far)."""
        t = 0xFFFFFFFFFFFFFFFF if after_ticket is None else int(after_ticket)
        n = self._L.mq_index_add_ref_staged(self._h, ref_idx, name.encode(), int(at), int(length), t)
        if n < 0:
            raise _err(self._L, "mq_index_add_ref_staged")
        return n

    def reserve_table(self, expected_kminmers):
        """DashMap::with_capacity (src/index.rs:83): the table for about this many k-min-mers is allocated and cleared in the background."""
        if self._L.mq_index_reserve(self._h, int(expected_kminmers)) != 0:
            raise _err(self._L, "mq_index_reserve")

    def finalize(self):
        """get_count + into_read_only (src/closures.rs:92-94): returns the unique k-min-mer count."""
        n = self._L.mq_index_finalize(self._h)
        if n < 0:
            raise _err(self._L, "mq_index_finalize")
        self._staged = {}  # every staged piece has been indexed: the sources can go
        return n

    def stats(self):
        st = IndexStats()
        if self._L.mq_index_get_stats(self._h, C.byref(st)) != 0:
            raise _err(self._L, "mq_index_get_stats")
        return {n: getattr(st, n) for n, _ in IndexStats._fields_}

    def ref_info(self, ref_id):
        name, ln = C.c_char_p(), C.c_uint64()
        if self._L.mq_index_ref_info(self._h, ref_id, C.byref(name), C.byref(ln)) != 0:
            raise _err(self._L, "mq_index_ref_info")
        return name.value.decode(), ln.value

    def map_batch(self, bases, offsets):
        """find_matches for every read (host buffers)."""
        bases = _seq(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        out = np.zeros(max(n, 0), dtype=hit_dtype)
        if n > 0 and self._L.mq_map_batch(self._h, _p(bases), _p(offsets), n, _p(out)) != 0:
            raise _err(self._L, "mq_map_batch")
        return out

    def map_batch_device(self, d_bases, d_offsets, n, total_bases, d_out, stream=0):
        """find_matches for n device-resident reads; total_bases = offsets[n] - offsets[0]."""
        rc = self._L.mq_map_batch_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n, int(total_bases), C.c_void_p(d_out),
                                         C.c_void_p(stream))
        if rc != 0:
            raise _err(self._L, "mq_map_batch_device")

    def reserve(self, n_reads, total_bases):
        if self._L.mq_map_reserve(self._h, n_reads, int(total_bases)) != 0:
            raise _err(self._L, "mq_map_reserve")

    def context(self):
        """A stream slot (mq_ctx): own stream, scratch and staging; contexts of one finalized index may map concurrently."""
        return Context(self)

    def last_map_ms(self):
        ms = C.c_float()
        if self._L.mq_last_map_ms(self._h, C.byref(ms)) != 0:
            raise _err(self._L, "mq_last_map_ms")
        return ms.value

    def last_stage_clocks(self):
        """Diagnostic (-DMQ_STAGE_CLOCKS builds): cycles per stage of the last launch, summed over waves (16 stages)."""
        out = (C.c_uint64 * 16)()
        if self._L.mq_last_stage_clocks(self._h, out) != 0:
            raise _err(self._L, "mq_last_stage_clocks")
        return [int(x) for x in out]

    def probe_stats(self, d_bases, d_offsets, n, total_bases, d_out):
        """(index lookups, slots visited beyond the home slot) of one instrumented launch: p-bar = 1 + extra / lookups."""
        a, b = C.c_uint64(), C.c_uint64()
        if self._L.mq_map_probe_stats(self._h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n, int(total_bases), C.c_void_p(d_out),
                                      C.byref(a), C.byref(b)) != 0:
            raise _err(self._L, "mq_map_probe_stats")
        return a.value, b.value

    def probe_rate(self, blocks, per_thread, bitmap_log2=0, table_too=1):
        """Diagnostic: (ms, lookups, extra steps) of blocks*256 threads probing per_thread random absent keys each."""
        ms, a, b = C.c_float(), C.c_uint64(), C.c_uint64()
        if self._L.mq_probe_rate(self._h, blocks, per_thread, bitmap_log2, table_too, C.byref(ms), C.byref(a), C.byref(b)) != 0:
            raise _err(self._L, "mq_probe_rate")
        return ms.value, a.value, b.value

    def last_map_path_counts(self):
        """(reads through the fast seeding path, reads through the general path) of the last launch."""
        a, b = C.c_uint32(), C.c_uint32()
        if self._L.mq_last_map_path_counts(self._h, C.byref(a), C.byref(b)) != 0:
            raise _err(self._L, "mq_last_map_path_counts")
        return a.value, b.value

    def table_alloc_ms(self):
        """What allocating + clearing this index's table took (on mq_index_reserve's thread or inside finalize), in ms."""
        ms = C.c_float()
        if self._L.mq_index_table_alloc_ms(self._h, C.byref(ms)) != 0:
            raise _err(self._L, "mq_index_table_alloc_ms")
        return ms.value

    def launch_waves(self, n_reads):
        """Persistent waves map_kernel employs for a launch of n_reads reads (wave w owns work items w and n_waves + w)."""
        a = C.c_uint32()
        if self._L.mq_map_launch_waves(self._h, int(n_reads), C.byref(a)) != 0:
            raise _err(self._L, "mq_map_launch_waves")
        return a.value

    def last_map_order(self):
        """(reads the ordering pass of the last launch took for short-period tandem arrays, reads it put first)."""
        a, b = C.c_uint32(), C.c_uint32()
        if self._L.mq_last_map_order(self._h, C.byref(a), C.byref(b)) != 0:
            raise _err(self._L, "mq_last_map_order")
        return a.value, b.value

    def kminmers_batch(self, bases, offsets, caps=None):
        """KminmersIterator output per sequence (src/mers.rs:41-54): list of structured arrays."""
        bases = _seq(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        if caps is None:  # one k-min-mer per base is an upper bound
            caps = (offsets[1:] - offsets[:-1]).astype(np.uint64)
        koff = np.zeros(n + 1, dtype=np.uint64)
        koff[1:] = np.cumsum(np.asarray(caps, dtype=np.uint64))
        out = np.zeros(int(koff[-1]), dtype=kminmer_dtype)
        counts = np.zeros(n, dtype=np.uint32)
        if n > 0 and self._L.mq_kminmers_batch(self._h, _p(bases), _p(offsets), n, _p(koff), _p(out), _p(counts)) != 0:
            raise _err(self._L, "mq_kminmers_batch")
        res = []
        for i in range(n):
            c = int(counts[i])
            if c > int(koff[i + 1] - koff[i]):
                raise MapquikError("k-min-mer window too small for sequence %d: %d > %d" % (i, c, int(koff[i + 1] - koff[i])))
            res.append(out[int(koff[i]):int(koff[i]) + c].copy())
        return res

    def lookup(self, hashes):
        """ReadOnlyIndex::get (src/index.rs:118-126) for many hashes."""
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        n = h.size
        found = np.zeros(n, dtype=np.uint8)
        ent = np.zeros(n, dtype=kminmer_dtype)
        ids = np.zeros(n, dtype=np.uint32)
        if n > 0 and self._L.mq_index_lookup(self._h, _p(h), n, _p(found), _p(ent), _p(ids)) != 0:
            raise _err(self._L, "mq_index_lookup")
        return found, ent, ids

    def format_paf(self, q_id, q_len, hit):
        rec = np.zeros(1, dtype=hit_dtype)
        rec[0] = hit
        cap = 4096
        while True:
            buf = C.create_string_buffer(cap)
            w = self._L.mq_format_paf(self._h, q_id.encode(), int(q_len), _p(rec), buf, cap)
            if w < 0:
                raise _err(self._L, "mq_format_paf")
            if w < cap:  # the return value is the full line length (snprintf): never hand back a cut line
                return buf.value.decode()
            cap = w + 1

    def paf_lines(self, names, offsets, hits):
        """PAF text in input order; unmapped reads produce no line (src/closures.rs:117-123)."""
        out = []
        for i, name in enumerate(names):
            st = int(hits[i]["status"])
            if st == MQ_HIT_OVERFLOW:
                raise MapquikError("read %s: Match-run scratch overflow (raise MQ_MATCH_CAP)" % name)
            if st == MQ_HIT_MAPPED:
                out.append(self.format_paf(name, int(offsets[i + 1] - offsets[i]), hits[i]))
        return out


class Context:
    """mq_ctx: one stream slot of a finalized index (the reference's per-thread worker, src/closures.rs:183,187)."""

    def __init__(self, index):
        self._L = index._L
        self._index = index  # keeps the index alive
        self._h = self._L.mq_ctx_new(index.handle)
        if not self._h:
            raise _err(self._L, "mq_ctx_new")
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.mq_ctx_free(self._h)
            self._h = None

    __del__ = close

    def map_batch(self, bases, offsets):
        bases = _seq(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        out = np.zeros(max(n, 0), dtype=hit_dtype)
        if n > 0 and self._L.mq_ctx_map_batch(self._h, _p(bases), _p(offsets), n, _p(out)) != 0:
            raise _err(self._L, "mq_ctx_map_batch")
        return out

    def submit(self, bases, offsets):
        """Queue a batch on the context's stream; wait() returns its hits."""
        bases = _seq(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        out = np.zeros(max(n, 0), dtype=hit_dtype)
        if n > 0 and self._L.mq_ctx_submit(self._h, _p(bases), _p(offsets), n, _p(out)) != 0:
            raise _err(self._L, "mq_ctx_submit")
        self._keep = (bases, offsets, out)

    def submit_spans(self, buf, starts, lens):
        """Queue reads given as spans of a raw buffer (FASTX bytes as they are): read i = buf[starts[i] : starts[i] + lens[i]]."""
        buf = _seq(buf)
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = starts.size
        out = np.zeros(max(n, 0), dtype=hit_dtype)
        if n > 0 and self._L.mq_ctx_submit_spans(self._h, _p(buf), buf.size, _p(starts), _p(lens), n, _p(out)) != 0:
            raise _err(self._L, "mq_ctx_submit_spans")
        self._keep = (buf, starts, out, lens)

    def submit_fasta(self, buf, begin=0, fastq=False):
        """Queue a piece of an uncompressed FASTA (or, fastq=True, four-line FASTQ) file that holds whole records (buf[begin:]): the
        records are found on the device."""
        buf = _seq(buf)
        if fastq:
            if self._L.mq_ctx_submit_fastx(self._h, _p(buf), int(begin), buf.size, 1) != 0:
                raise _err(self._L, "mq_ctx_submit_fastx")
        elif self._L.mq_ctx_submit_fasta(self._h, _p(buf), int(begin), buf.size) != 0:
            raise _err(self._L, "mq_ctx_submit_fasta")
        self._keep = (buf,)

    def wait_fasta(self):
        """(hits, line_ends, flags) of the piece submitted with submit_fasta; flags & 1 (MQ_FASTA_IRREGULAR): not two lines per record, nothing mapped."""
        n, nl, fl = C.c_uint32(), C.c_uint32(), C.c_uint32()
        le, hp = C.c_void_p(), C.c_void_p()
        if self._L.mq_ctx_wait_fasta(self._h, C.byref(n), C.byref(le), C.byref(nl), C.byref(hp), C.byref(fl)) != 0:
            raise _err(self._L, "mq_ctx_wait_fasta")
        self._keep = None
        if fl.value & 1 or n.value == 0:
            return np.zeros(0, dtype=hit_dtype), np.zeros(0, dtype=np.uint32), fl.value
        hits = np.frombuffer(C.string_at(hp.value, n.value * hit_dtype.itemsize), dtype=hit_dtype).copy()
        lines = np.frombuffer(C.string_at(le.value, nl.value * 4), dtype=np.uint32).copy()
        return hits, lines, fl.value

    def wait(self):
        if self._L.mq_ctx_wait(self._h) != 0:
            raise _err(self._L, "mq_ctx_wait")
        out = self._keep[2] if self._keep else np.zeros(0, dtype=hit_dtype)
        self._keep = None
        return out

    def map_batch_device(self, d_bases, d_offsets, n, total_bases, d_out, stream=0):
        if self._L.mq_ctx_map_batch_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n, int(total_bases), C.c_void_p(d_out),
                                           C.c_void_p(stream)) != 0:
            raise _err(self._L, "mq_ctx_map_batch_device")

    def last_map_ms(self):
        ms = C.c_float()
        if self._L.mq_ctx_last_map_ms(self._h, C.byref(ms)) != 0:
            raise _err(self._L, "mq_ctx_last_map_ms")
        return ms.value


def ref_extract(ref_idx, inp_seq_raw, params, mers_index, name=None):
    """mers::ref_extract (src/mers.rs:15-38) + ref_map.insert (src/closures.rs:49)."""
    assert params is mers_index.params or bytes(params) == bytes(mers_index.params)
    return mers_index.add_ref(ref_idx, name if name is not None else str(ref_idx), inp_seq_raw)


def find_matches(q_id, q_len, q_str, ref_map, mers_index, params):
    """mers::find_matches (src/mers.rs:77-102): the PAF line or None.  ref_map is carried by the index."""
    s = _seq(q_str)
    assert q_len == s.size
    hits = mers_index.map_batch(s, np.array([0, s.size], dtype=np.uint64))
    lines = mers_index.paf_lines([q_id], [0, s.size], hits)
    return lines[0] if lines else None
