/*
 * mapquik_hip.h -- C ABI of the MI355X (gfx950) implementation of mapquik's hot path:
 * k-min-mer seeding -> unique-k-min-mer index lookup -> Match runs -> pseudo-chain -> PAF columns.
 *
 * This is the drop-in boundary.  The reference (ekimb/mapquik, Rust) has no FFI of its own; the
 * internal seam these entry points replace is the pair called from the seq_io worker closures
 * (src/closures.rs:46-51,100-104):
 *     mers::ref_extract (ref_idx, &[u8], &Params, &Index) -> usize            src/mers.rs:15
 *     mers::find_matches(q_id, q_len, &[u8], &ref_map, &ReadOnlyIndex, &Params) -> Option<String>   src/mers.rs:77
 * in batch form (many sequences per call), with plain pointers and sizes only.
 * INTEGRATION.md shows the `extern "C"` block a Rust maintainer would add.
 *
 * Conventions
 *   - Sequences are bytes exactly as the reference's seam receives them: upper-cased ASCII
 *     (src/closures.rs:63,106).  Any byte other than A,C,G,T hashes as ntHash seed 0.
 *   - All functions return 0 (or a non-negative count) on success and a negative MQ_E* code on error;
 *     mq_last_error() returns a thread-local message.  Nothing throws across the boundary.
 *   - The caller owns every buffer it passes; the library owns device memory behind mq_index.
 *   - The compute path is HIP only.  There is no CPU fallback: without a usable GPU every compute entry
 *     point fails with MQ_ENODEVICE.
 *   - Limits (checked, MQ_EINVAL otherwise): 1 <= l <= 64, 1 <= k <= 32, sequence length < 2^32, fewer than 2^30 reads per batch.
 */
#ifndef MAPQUIK_HIP_H
#define MAPQUIK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MQ_ABI_VERSION 4

#define MQ_OK 0
#define MQ_EINVAL (-1)
#define MQ_ENODEVICE (-2)
#define MQ_EHIP (-3)
#define MQ_ENOMEM (-4)
#define MQ_ESTATE (-5)
#define MQ_EOVERFLOW (-6)

/* Params (src/main.rs:33-47); defaults src/main.rs:174-188.  I/O-only members (b, q, threads, debug) are host-side. */
typedef struct mq_params {
    uint32_t k;       /* k-min-mer length, default 5 */
    uint32_t l;       /* minimizer length, default 31 */
    double   density; /* FH density, default 0.01 */
    uint32_t use_hpc; /* 1 unless --nohpc */
    uint32_t c;       /* minimum chain length, default 4 */
    uint32_t s;       /* minimum matching seeds, default 11 */
    uint32_t g;       /* maximum gap difference, default 2000 */
    uint32_t flags;   /* MQ_FLAG_*; default 0 */
} mq_params;
/* The reference upper-cases every sequence before the seam (to_ascii_uppercase, src/closures.rs:63,106).  With this flag the
 * kernels treat a-z as A-Z themselves, so a feeder can hand over raw FASTX bytes without touching them. */
#define MQ_FLAG_FOLD_CASE 1u
/* Opt-in: a cheap k-min-mer tuple hash in place of the reference's SipHash-1-3 (Rust DefaultHasher over the tuple, src/index.rs:100-104 via
 * the crate's KminmerHash).  The PAF depends on that hash only through EQUALITY (index hit / miss / duplicate, src/index.rs:118-126,
 * src/match.rs:39-58), so the lines are the same; mq_kminmer.hash is then NOT the reference's value.  An add-rotate-xor chain on two 64-bit
 * words (one step per minimizer, six to finish; the test suite's CPU checker has the same function).  The index and the reads must be hashed
 * alike: the flag belongs to the seeding parameters (a saved index carries it).  Default off; never used for a headline number. */
#define MQ_FLAG_FAST_KH 2u
/* Seeding variants (flags bits 8..13; default 0 = the frozen reading of DESIGN.md section 2).  The k-min-mer iterator is a third-party
 * crate (rust-seq2kminmers, Cargo.toml:30, no pinned revision; call sites src/mers.rs:22-27,53) that this image cannot build, so
 * six of its decisions are switchable: tools/check_against_upstream.sh finds, on a machine with cargo, which combination reproduces
 * the crate, and the product is then run with that value (`mapquik --seeding-variant v`).  Bits, any combination:
 *    1  D3      strict `<` on the density bound (frozen: `<=`)
 *    2  D2      FH = f32: the bound is computed in single precision
 *    4  D2/D12  H = u32: 32-bit ntHash (low halves of the seeds, rotations mod 32) and a 32-bit bound -- what a 16-lane AVX-512
 *               HashMode::HpcSimd (the reference's default mode, src/mers.rs:22) may use; hashes are zero-extended into the tuple
 *    8  D5      a minimizer's position = raw index of the LAST base of its first base's homopolymer run (frozen: the run head); needs l >= 2
 *   16  D6      end = raw position of the last compressed base of the last minimizer's l-mer (frozen: pos[k-1] + l - 1)
 *   32  D8      rev = reversed tuple <= forward tuple (frozen: strict <; differs on palindromic tuples only)
 * mq_index_new rejects undefined bits and unsupported combinations with MQ_EINVAL. */
#define MQ_FLAG_SEED_VARIANT_SHIFT 8
#define MQ_FLAG_SEED_VARIANT_MASK (0x3Fu << MQ_FLAG_SEED_VARIANT_SHIFT)
#define MQ_FLAG_SEED_VARIANT(v) (((uint32_t)(v) & 0x3Fu) << MQ_FLAG_SEED_VARIANT_SHIFT)
#define MQ_SEEDVAR_STRICT_BOUND 1u
#define MQ_SEEDVAR_F32_BOUND 2u
#define MQ_SEEDVAR_HASH32 4u
#define MQ_SEEDVAR_POS_RUN_END 8u
#define MQ_SEEDVAR_END_COMPRESSED 16u
#define MQ_SEEDVAR_REV_ON_EQUAL 32u

/* One k-min-mer as the reference's KminmerHash exposes it (fields used at src/index.rs:57-58,101). 24 bytes. */
typedef struct mq_kminmer {
    uint64_t hash;
    uint32_t start;
    uint32_t end;
    uint32_t offset;
    uint32_t rev;
} mq_kminmer;

/* Result of find_matches for one read: the numeric PAF columns of src/mers.rs:181.  48 bytes.
 * status 0 => the reference returns None (no line is written, src/closures.rs:119-121).
 * Columns 3 and 4 are 64-bit (low word, high word): find_coords computes in usize and a run that the Match::check precedence
 * quirk (src/match.rs:39-43) extended onto ANOTHER, longer reference makes `r_len - r_end - 1` wrap (src/mers.rs:131-183; release
 * builds wrap silently), so the reference prints values like 18446744073709547279 there -- and so does mq_format_paf. */
#define MQ_HIT_UNMAPPED 0u
#define MQ_HIT_MAPPED 1u
#define MQ_HIT_OVERFLOW 2u /* more Match runs than the per-read scratch holds: result NOT computed (loud, never silent) */
typedef struct mq_hit {
    uint32_t status;
    uint32_t ref_id;     /* index passed to mq_index_add_ref */
    uint32_t rc;         /* 1 => '-' */
    uint32_t mapq;       /* 0 or 60 */
    uint32_t q_start;    /* column 3, low 32 bits */
    uint32_t q_end;      /* column 4 (inclusive, as the reference prints it), low 32 bits */
    uint32_t r_start;    /* column 8 */
    uint32_t r_end;      /* column 9 (inclusive) */
    uint32_t score;      /* column 10: number of matching k-min-mers */
    uint32_t n_kminmers; /* k-min-mers extracted from the read (diagnostic) */
    uint32_t q_start_hi; /* column 3, high 32 bits (0 unless the usize arithmetic of find_coords wrapped) */
    uint32_t q_end_hi;   /* column 4, high 32 bits */
} mq_hit;

typedef struct mq_index mq_index; /* opaque: Index / ReadOnlyIndex (src/index.rs:73-128) + ref_map (src/closures.rs:30) */

typedef struct mq_index_stats {
    uint64_t n_refs;
    uint64_t n_kminmers;  /* total inserted, sum of mq_index_add_ref returns */
    uint64_t n_keys;      /* distinct hashes (incl. tombstones) */
    uint64_t n_unique;    /* Index::get_count(): non-tombstones (src/index.rs:90-92) */
    uint64_t table_slots; /* power of two */
    uint64_t table_bytes;
    uint64_t slot_bytes;
} mq_index_stats;

const char *mq_last_error(void);
int mq_abi_version(void);
/* number of HIP devices visible; 0 or negative => no compute possible */
int mq_device_count(void);
void mq_params_default(mq_params *p); /* src/main.rs:174-188 */

/* Index::new (src/index.rs:78-88) on HIP device `device`. */
mq_index *mq_index_new(const mq_params *params, int device);
void mq_index_free(mq_index *idx);

/* index_mers closure (src/closures.rs:46-51) = ref_extract (src/mers.rs:15-38) + ref_map.insert.
 * Returns the number of k-min-mers of this reference (the "Indexed reference {}: {} k-min-mers." count) or <0.
 * ref_id values must be distinct; seq is host memory (mq_index_add_ref) or device memory (.._device). */
int64_t mq_index_add_ref(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *seq, uint64_t len);
int64_t mq_index_add_ref_device(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *d_seq, uint64_t len);

/* The reference file handed over in pieces (the batch form of the reader loop of src/closures.rs:46-94 for a caller that never holds a
 * whole record in host memory): the file's bytes go to a device buffer of the index piece by piece, asynchronously, and a record is
 * indexed from there while the pieces behind it are still on their way.
 *   mq_index_stage_begin(idx, total_bytes)       the device buffer, total_bytes = the file's size; once per index, before the first piece
 *   mq_index_stage_piece(idx, at, src, n, &t)    queues the copy of src[0, n) to buffer offset `at` and returns; src (page-locked memory,
 *                                                mq_host_alloc, for the link's full rate) must stay untouched until
 *   mq_index_stage_done(idx, t, wait)            returns 1 (copied; src may be reused), 0 (not yet; only with wait == 0) or <0
 *   mq_index_add_ref_staged(idx, id, name, at, len, t)  = mq_index_add_ref_device on buffer[at, at + len), ordered (on the device) behind
 *                                                piece t and every piece issued before it -- the last piece that holds bytes of the
 *                                                record -- or, t = MQ_STAGE_ALL_ISSUED, behind every piece issued so far
 * Pieces may be issued from one thread while another asks for records.  The buffer is released by mq_index_finalize. */
#define MQ_STAGE_ALL_ISSUED (~(uint64_t)0)
int mq_index_stage_begin(mq_index *idx, uint64_t total_bytes);
int mq_index_stage_piece(mq_index *idx, uint64_t at, const uint8_t *src, uint64_t n, uint64_t *ticket);
int mq_index_stage_done(mq_index *idx, uint64_t ticket, int wait);
int64_t mq_index_add_ref_staged(mq_index *idx, uint32_t ref_id, const char *name, uint64_t at, uint64_t len, uint64_t after_ticket);

/* DashMap::with_capacity (src/index.rs:83 sizes its map for 39,821,990 k-min-mers at Index::new): a hint that about
 * expected_kminmers k-min-mers will be inserted.  The table is allocated and cleared in the background while the references are
 * added; mq_index_finalize adopts it when the size fits and allocates anew when it does not.  Fresh device memory costs ~30 ms per GB
 * here (0.5 s for a human genome's table): call this as early as the reference's size is known.  One reservation per index. */
int mq_index_reserve(mq_index *idx, uint64_t expected_kminmers);
/* Slots of the table per inserted k-min-mer (2..64, rounded up to a power of two of slots; default 8: load <= 1/8, the fastest lookups
 * for a kernel fed from HBM).  A caller bound by its host side (files -> PAF runs at a thirtieth of the kernel's rate) takes 2: a
 * quarter of the memory per replica and of its allocation, for 5 % of kernel speed it never sees.  Before mq_index_reserve / finalize. */
int mq_index_set_table_factor(mq_index *idx, uint32_t slots_per_kminmer);
/* get_count + into_read_only (src/closures.rs:92-94): dedup (a hash seen twice is a tombstone,
 * src/index.rs:94-104), build the HBM-resident table.  Returns the unique count or <0. */
int64_t mq_index_finalize(mq_index *idx);
int mq_index_get_stats(const mq_index *idx, mq_index_stats *out);
/* On-disk form of a finalized index (the reference has none: it re-indexes the FASTA on every run, src/closures.rs:24-94).
 * The file holds the parameters, the reference names/lengths and the slot table; mq_index_load returns a finalized index. */
int mq_index_save(const mq_index *idx, const char *path);
mq_index *mq_index_load(const char *path, int device);
/* A replica of a finalized index on HIP device `device` (device-to-device copy of the table; the multi-GPU drivers build the
 * index once and clone it instead of indexing the reference on every GPU).  Free it with mq_index_free. */
mq_index *mq_index_clone(const mq_index *src, int device);
int mq_index_ref_info(const mq_index *idx, uint32_t ref_id, const char **name, uint64_t *len);
/* The parameters an index was built with (a loaded file's k, l, density, use_hpc and seeding variant decide its keys), and the ones that
 * act at mapping time only -- Params.c / .s (Chain::get_match, src/chain.rs:147-169), .g (the gap tests, src/chain.rs:132-142) and the
 * case folding -- which a loaded index takes from its caller's command line. */
int mq_index_get_params(const mq_index *idx, mq_params *out);
int mq_index_set_map_params(mq_index *idx, uint32_t c, uint32_t s, uint32_t g, int fold_case);

/* find_matches (src/mers.rs:77-102) for n reads.  bases: concatenated reads; offsets: n+1 prefix offsets.
 * Host-buffer form: copies in, runs, copies out, synchronises.  Reads that overflow the per-wave Match scratch or whose
 * minimizer list outgrows its region are mapped again on the GPU with worst-case scratch, so no MQ_HIT_OVERFLOW is returned
 * from this entry point. */
int mq_map_batch(mq_index *idx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out);
/* Device-resident form: all pointers are device memory on the index's device; asynchronous on `stream`
 * (a hipStream_t, may be NULL).  total_bases = d_offsets[n] - d_offsets[0] (the host knows it: it built the offsets); it sizes
 * the per-read minimizer lists.  No allocation happens here once the scratch has grown to the batch shape (mq_map_reserve).
 * A read with more Match runs than the scratch holds (MQ_MATCH_CAP, default 2048) or with a minimizer list denser than
 * 4*density + 1/512 per base gets status MQ_HIT_OVERFLOW here (loud, never a wrong line). */
int mq_map_batch_device(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases,
                        mq_hit *d_out, void *stream);

/* THREADING CONTRACT.  The reference maps on --threads workers over one read-only index (src/closures.rs:183,187,
 * src/index.rs:108-116).  Here: every mq_index_* / mq_map_* entry point locks the index, so calls on ONE index from several
 * threads are safe and run one after another (they share the index's default context: one set of work counters, scratch and
 * staging buffers).  For launch sequences in flight TOGETHER on one finalized index, give every worker thread (or every
 * stream slot of a pipelined feeder) its own mq_ctx: a context owns a stream, work counters, Match scratch, minimizer
 * lists, staging buffers and events, and the finalized table is only read.  One context runs one launch sequence at a time
 * (calls on one context must not overlap; successive device-form calls must be ordered by their streams).
 * mq_index_add_ref / mq_index_finalize must have returned before any context maps; free every context before its index. */
typedef struct mq_ctx mq_ctx;
mq_ctx *mq_ctx_new(mq_index *idx);
void mq_ctx_free(mq_ctx *ctx);
/* mq_map_batch on this context. */
int mq_ctx_map_batch(mq_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out);
/* The same in two halves, for double buffering: submit queues H2D copy, kernels and D2H copy on the context's stream and
 * returns; `out` is filled (and overflow reads redone) by mq_ctx_wait.  bases/offsets/out must stay valid until then; bases
 * in page-locked memory (mq_host_alloc) overlaps the copy with other contexts' kernels. */
int mq_ctx_submit(mq_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out);
int mq_ctx_wait(mq_ctx *ctx);
/* Spans form of mq_ctx_submit for raw FASTX buffers: the whole buffer buf[0, buf_bytes) goes to the device as it is (headers,
 * line ends, quality lines and all) and read i is buf[starts[i], starts[i] + lens[i]); spans in order and disjoint.  With
 * MQ_FLAG_FOLD_CASE the host never has to touch a base.  starts/lens/buf/out must stay valid until mq_ctx_wait. */
int mq_ctx_submit_spans(mq_ctx *ctx, const uint8_t *buf, uint64_t buf_bytes, const uint64_t *starts, const uint32_t *lens, uint32_t n,
                        mq_hit *out);
/* FASTA records found on the device (the batch form of closures.rs:100-123 for a reader that does not parse): buf[begin, bytes) is a
 * piece of an uncompressed FASTA file that holds WHOLE records, begin at a record's '>'.  The bytes go to the device as they are,
 * kernels find the line ends, and read i is the second line of record i (a '\r' in front of the '\n' is cut; a last line without
 * '\n' ends at `bytes`).  mq_ctx_submit_fasta queues copy and scan and returns; mq_ctx_wait_fasta launches the map kernels once the
 * record count is known and returns pointers into the context's page-locked memory, valid until its next submit:
 *   line_ends[0 .. n_lines)  positions of the line ends in buf, ascending; record i: header = buf[(i ? line_ends[2i-1]+1 : begin), line_ends[2i]),
 *                            sequence = buf[line_ends[2i]+1, line_ends[2i+1])   (n_lines = 2 * n_reads)
 *   hits[0 .. n_reads)       as mq_ctx_wait fills them (overflow reads redone)
 * flags & MQ_FASTA_IRREGULAR: the piece is not "header line, sequence line" all through (sequences over several lines, blank lines,
 * more line ends than bytes / 16): nothing was mapped, n_reads = 0 -- parse it on the host and use mq_ctx_submit_spans.  buf must stay
 * valid until mq_ctx_wait_fasta has returned; page-locked memory (mq_host_alloc) gives the full PCIe rate. */
#define MQ_FASTA_IRREGULAR 1u
int mq_ctx_submit_fasta(mq_ctx *ctx, const uint8_t *buf, uint64_t begin, uint64_t bytes);
/* The same for either format (the reference reads FASTQ unless the file's name says FASTA, src/main.rs:196-205; its real-data headline
 * run is an uncompressed FASTQ, experiments/table1.sh:50).  MQ_FASTX_FASTQ: buf[begin, bytes) holds whole four-line records, begin at a
 * record's '@'; read i is the second line of record i, and mq_ctx_wait_fasta reports n_lines = 4 * n_reads line ends (record i: header
 * = buf[(i ? line_ends[4i-1]+1 : begin), line_ends[4i]), sequence = buf[line_ends[4i]+1, line_ends[4i+1])).  The device checks every
 * record the way the reference's reader would have to: '@' opens it, '+' opens its third line, one quality per base; anything else
 * (sequences or qualities over several lines, blank lines, a truncated record) comes back MQ_FASTA_IRREGULAR for the host's parser.
 * The quality bytes cross the link but no host thread reads them. */
#define MQ_FASTX_FASTA 0u
#define MQ_FASTX_FASTQ 1u
int mq_ctx_submit_fastx(mq_ctx *ctx, const uint8_t *buf, uint64_t begin, uint64_t bytes, uint32_t format);
int mq_ctx_wait_fasta(mq_ctx *ctx, uint32_t *n_reads, const uint32_t **line_ends, uint32_t *n_lines, const mq_hit **hits, uint32_t *flags);
/* Pre-size the context's device staging, minimizer lists and scratch for batches of up to n_reads reads / total_bytes buffer
 * bytes, so that the first submit does not pay for the allocations. */
int mq_ctx_reserve(mq_ctx *ctx, uint32_t n_reads, uint64_t total_bytes);
/* mq_map_batch_device on this context. */
int mq_ctx_map_batch_device(mq_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases,
                            mq_hit *d_out, void *stream);

/* Page-locked host memory for the buffers handed to mq_map_batch / mq_index_add_ref (a feeder that parses FASTX straight
 * into such buffers gets the full PCIe rate on the copy in; pageable buffers work too, at roughly a quarter of it). */
void *mq_host_alloc(size_t bytes);
void mq_host_free(void *p);
/* Page-lock / release a range the caller owns, whole pages (the feeder does this to slices of the mapped reads file, so that
 * mq_ctx_submit_fasta's copy is a DMA out of the page cache that no host thread waits for).  0 on success. */
int mq_host_register(void *ptr, size_t bytes);
int mq_host_unregister(void *ptr);

/* Pre-size the default context's scratch for batches of up to n_reads reads / total_bases bases, so that mq_map_batch_device
 * never allocates. */
int mq_map_reserve(mq_index *idx, uint32_t n_reads, uint64_t total_bases);

/* Parity/debug: the k-min-mers of each sequence as KminmersIterator yields them (src/mers.rs:41-54).
 * kmm_offsets (n+1, host) gives each sequence's capacity window in `out`; counts[i] receives the true count. */
int mq_kminmers_batch(mq_index *idx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, const uint64_t *kmm_offsets,
                      mq_kminmer *out, uint32_t *counts);

/* Parity/debug: probe the finalized table (ReadOnlyIndex::get, src/index.rs:118-126).  found[i]=1 on a live hit;
 * entries[i] = {hash, start, end, offset, rev=rc} with ref ids in ref_ids[i]. */
int mq_index_lookup(mq_index *idx, const uint64_t *hashes, uint32_t n, uint8_t *found, mq_kminmer *entries, uint32_t *ref_ids);

/* The format! of src/mers.rs:181 (no newline).  Returns the length or <0. */
int mq_format_paf(const mq_index *idx, const char *q_id, uint64_t q_len, const mq_hit *hit, char *buf, size_t cap);

/* Measurement and diagnostic entry points (probe statistics, stage clocks, launch timers): include/mapquik_hip_diag.h. */

#ifdef __cplusplus
}
#endif
#endif
