/* mapquik_hip_diag.h -- measurement and diagnostic entry points of libmapquik_hip.so.  NOT part of the seam: nothing a maintainer binds
 * to replace mers::ref_extract / mers::find_matches (src/mers.rs:15-38, 77-102) lives here -- that is include/mapquik_hip.h, which
 * INTEGRATION.md mirrors one to one.  These serve bench.py (probes per lookup for the roofline), tools/ (probe rate, stage clocks) and
 * the tests (which seeding path a read took). */
#ifndef MAPQUIK_HIP_DIAG_H
#define MAPQUIK_HIP_DIAG_H

#include "mapquik_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: how many reads of the last map launch took the fast seeding path (ACGT-only) and how many
 * the general streaming path.  Both produce identical results.  Synchronises on the launch. */
int mq_last_map_path_counts(mq_index *idx, uint32_t *n_fast, uint32_t *n_general);

/* The launch order of the last map launch: how many reads looked like short-period tandem arrays to the ordering pass (n_flagged) and
 * how many of them were taken up first (n_first <= n_flagged: the front of the order holds 32,768). */
int mq_last_map_order(mq_index *idx, uint32_t *n_flagged, uint32_t *n_first);

/* What allocating and clearing the finalized index's table took (hipMalloc + memset + synchronize; milliseconds), wherever it ran: on
 * mq_index_reserve's background thread beside the reference phase, or inside mq_index_finalize.  0 for a loaded or cloned index. */
int mq_index_table_alloc_ms(mq_index *idx, float *ms);

/* How many persistent waves map_kernel employs for a launch of n_reads reads (its grid x 8): wave w's first two work items are items w
 * and n_waves + w of the launch order, the atomic counter hands out the rest -- the tests place reads on exactly those borders. */
int mq_map_launch_waves(mq_index *idx, uint32_t n_reads, uint32_t *n_waves);

/* Measurement aid: one instrumented (slower, never timed) launch of the same batch that counts index lookups and the slots
 * visited beyond each lookup's home slot: mean probes per lookup = 1 + extra_steps / lookups (SURVEY 8d's p-bar). */
int mq_map_probe_stats(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases, mq_hit *d_out,
                       uint64_t *lookups, uint64_t *extra_steps);

/* Diagnostic (tools/read_tail.py): after mq_map_probe_stats -- what each of the n reads of that launch cost its wave (shader-clock cycles) and
 * when the wave took it up (ticks of the 100-MHz constant clock): which reads make a launch's tail. */
int mq_last_read_cycles(mq_index *idx, uint32_t n, uint32_t *cycles, uint64_t *start_ticks);

/* Diagnostic (tools/probe_rate.py): blocks*256 threads each probe per_thread pseudo-random (absent) keys of the finalized table;
 * returns the kernel time, the lookups made and the slots visited beyond the home slots.  Measures the random-access rate the
 * memory system sustains on this table, detached from the map path.  bitmap_log2 != 0: test a stand-in bitmap of 2^bitmap_log2
 * bits (one in eight set) first, and probe the table only for keys whose bit is set (table_too) or not at all. */
int mq_probe_rate(mq_index *idx, uint32_t blocks, uint32_t per_thread, uint32_t bitmap_log2, uint32_t table_too, float *ms,
                  uint64_t *lookups, uint64_t *extra_steps);

/* Diagnostic: shader-clock cycles the waves of the last map launch of the index's default context spent in each of 16 stages
 * (list in mapquik_amd/csrc/mq_device.hpp, mq_clk), summed over waves.  Only a library built with -DMQ_STAGE_CLOCKS fills
 * them (tools/stage_clocks.py builds one beside the product library); the product build returns zeros. */
int mq_last_stage_clocks(mq_index *idx, uint64_t *out16);

/* Time of the last map launch sequence of the index's default context / of a context, from events on its stream. */
int mq_last_map_ms(mq_index *idx, float *ms);
int mq_ctx_last_map_ms(mq_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif

#endif /* MAPQUIK_HIP_DIAG_H */
