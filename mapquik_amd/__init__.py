"""mapquik_amd -- MI355X (gfx950) implementation of mapquik's k-min-mer seeding + pseudo-chaining hot path.

The product is the C-ABI shared library (include/mapquik_hip.h, mapquik_amd/csrc/).  This package is the thin
Python host mirror of the reference's operator interface for that path (Params, Index, ref_extract, find_matches).
"""
from .api import (MQ_HIT_MAPPED, MQ_HIT_OVERFLOW, MQ_HIT_UNMAPPED, Index, MapquikError, Params, PinnedBuffer, device_count, find_matches,
                  hit_column, hit_dtype, kminmer_dtype, load_library, ref_extract)

__all__ = ["Params", "Index", "PinnedBuffer", "ref_extract", "find_matches", "MapquikError", "device_count", "load_library", "hit_dtype", "hit_column",
           "kminmer_dtype", "MQ_HIT_MAPPED", "MQ_HIT_UNMAPPED", "MQ_HIT_OVERFLOW"]
