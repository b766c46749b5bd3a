/*
 * mapquik_oracle.h -- CPU restatement of mapquik's k-min-mer seeding + pseudo-chaining path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (mapquik_amd/, include/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker / CPU baseline.
 *
 * PARITY STATUS
 *   - Stages whose source is in /root/reference (src/index.rs, src/match.rs, src/chain.rs,
 *     src/mers.rs) are restated line-for-line in meaning and pinned by the hand-derived
 *     known-answer vectors of SURVEY.md Appendix C (tests/golden/kat_intree.json).
 *   - The seeding stage lives in the third-party crate `rust-seq2kminmers`
 *     (git = https://github.com/rchikhi/rust-seq2kminmers, NO rev/tag pin: Cargo.toml:30,
 *     Cargo.lock git-ignored).  Its source is not in /root/reference and there is no Rust
 *     toolchain or network here, so it is restated from its published algorithm
 *     (ntHash-1 64-bit; density selection; k consecutive minimizers; lexicographic
 *     canonicalisation; SipHash-1-3 tuple hash) as frozen in DESIGN.md "Seeding spec".
 *     ==> "parity unpinned" for the seeding stage: the ntHash arithmetic is checked against
 *     the published ntHash-1 known answers, but not against the crate itself.
 *
 * All coordinates are `usize` in the reference (64-bit, wrapping in release builds:
 * Cargo.toml:42-49); they are uint64_t here and every subtraction wraps the same way.
 */
#ifndef MAPQUIK_ORACLE_H
#define MAPQUIK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/main.rs:33-47 (Params); defaults src/main.rs:174-188 */
typedef struct {
    uint64_t k;       /* k-min-mer length (5) */
    uint64_t l;       /* minimizer length (31) */
    double   density; /* FH (0.01) */
    int      use_hpc; /* 1 unless --nohpc */
    uint64_t c;       /* minimum chain length (4) */
    uint64_t s;       /* minimum matching seeds (11) */
    uint64_t g;       /* max gap difference (2000) */
} mqo_params;

void mqo_params_default(mqo_params *p);

/* rust-seq2kminmers KminmerHash: fields used at src/index.rs:57-58,101; src/match.rs:22-27 */
typedef struct {
    uint64_t hash;
    uint64_t start;
    uint64_t end;
    uint64_t offset;
    int32_t  rev;
    int32_t  _pad;
} mqo_kminmer;

/* one selected l-mer: raw position of its first base (run head under HPC) + canonical ntHash */
typedef struct {
    uint64_t pos;
    uint64_t hash;
} mqo_minimizer;

/* src/index.rs:42-70 */
typedef struct {
    uint64_t id;
    uint64_t start;
    uint64_t end;
    uint64_t offset;
    int32_t  rc;
    int32_t  _pad;
} mqo_entry;

/* src/match.rs:10-18 */
typedef struct {
    uint64_t q_start;
    uint64_t q_end;
    uint64_t r_start;
    uint64_t r_end;
    uint64_t count;
    int32_t  rc;
    int32_t  _pad;
} mqo_match;

/* src/main.rs:31 PseudoChainCoords = (rc, q_start, q_end, r_start, r_end, score, mapq) */
typedef struct {
    int32_t  rc;
    int32_t  _pad;
    uint64_t q_start;
    uint64_t q_end;
    uint64_t r_start;
    uint64_t r_end;
    uint64_t score;
    uint64_t mapq;
} mqo_coords;

/* numeric columns of one PAF line (src/mers.rs:131-183) */
typedef struct {
    int32_t  mapped; /* 0 => find_matches returned None */
    int32_t  rc;
    uint64_t ref_id;
    uint64_t q_len;
    uint64_t q_start;
    uint64_t q_end;
    uint64_t r_len;
    uint64_t r_start;
    uint64_t r_end;
    uint64_t score;
    uint64_t mapq;
} mqo_paf;

/* ---- seeding (restated third-party algorithm; see header note) ---- */
uint64_t mqo_nt_seed(uint8_t c);                                   /* ntHash-1 seed table, non-ACGT -> 0 */
uint64_t mqo_ntf64(const uint8_t *s, size_t i, size_t l);          /* forward hash of s[i..i+l) */
uint64_t mqo_ntr64(const uint8_t *s, size_t i, size_t l);          /* reverse-complement hash */
uint64_t mqo_ntc64(const uint8_t *s, size_t i, size_t l);          /* min(fwd, rev) */
/* diagnostic variants of the unpinned seeding decisions (0 = the frozen reading; bit list in mapquik_oracle.c; they apply to
 * mqo_kminmers and everything built on it, bits 1 and 2 to mqo_minimizers as well) */
void mqo_set_variant(int v);
int mqo_get_variant(void);
uint64_t mqo_density_bound(double density);                        /* (density * u64::MAX as f64) as u64 */
uint64_t mqo_siphash(const uint8_t *msg, size_t len, uint64_t k0, uint64_t k1, int c_rounds, int d_rounds);
uint64_t mqo_tuple_hash(const uint64_t *mers, size_t k);           /* Rust DefaultHasher over a [u64] slice (variant bit 64: mqo_tuple_hash_fast) */
uint64_t mqo_tuple_hash_fast(const uint64_t *mers, size_t k);      /* the product's MQ_FLAG_FAST_KH mixer (not the reference's hash value) */

/* Both return the count; if out == NULL only count.  cap = capacity of out. */
size_t mqo_minimizers(const uint8_t *seq, size_t len, const mqo_params *p, mqo_minimizer *out, size_t cap);
size_t mqo_minimizers_naive(const uint8_t *seq, size_t len, const mqo_params *p, mqo_minimizer *out, size_t cap);
size_t mqo_kminmers(const uint8_t *seq, size_t len, const mqo_params *p, mqo_kminmer *out, size_t cap);

/* ---- index (src/index.rs) ---- */
typedef struct mqo_index mqo_index;
mqo_index *mqo_index_new(void);
void       mqo_index_free(mqo_index *ix);
void       mqo_index_add(mqo_index *ix, uint64_t h, uint64_t id, uint64_t start, uint64_t end, uint64_t offset, int rc);
const mqo_entry *mqo_index_get(const mqo_index *ix, uint64_t h);   /* tombstone => NULL */
uint64_t   mqo_index_count(const mqo_index *ix);                   /* non-tombstones */
uint64_t   mqo_index_keys(const mqo_index *ix);                    /* all keys incl. tombstones */
/* src/mers.rs:15-38; also registers (name,len) like closures.rs:46-51 */
uint64_t   mqo_ref_extract(mqo_index *ix, uint64_t ref_idx, const uint8_t *seq, size_t len, const mqo_params *p);
void       mqo_index_set_ref(mqo_index *ix, uint64_t ref_idx, const char *name, uint64_t len);
uint64_t   mqo_index_ref_len(const mqo_index *ix, uint64_t ref_idx);
const char *mqo_index_ref_name(const mqo_index *ix, uint64_t ref_idx);
uint64_t   mqo_index_n_refs(const mqo_index *ix);
/* multi-threaded build over many references (same final state: order independent, src/index.rs:94-104) */
uint64_t   mqo_index_build_mt(mqo_index *ix, const uint8_t *bases, const uint64_t *offsets, uint32_t n_refs,
                              const mqo_params *p, int threads, uint64_t *per_ref_counts);

/* ---- match / chain on explicit inputs (for KATs) ---- */
void mqo_match_new(mqo_match *m, const mqo_kminmer *q, const mqo_entry *r);
void mqo_match_update(mqo_match *m, const mqo_kminmer *q, const mqo_entry *r);
int  mqo_match_check(const mqo_match *m, const mqo_kminmer *q, const mqo_entry *r, const mqo_entry *p);
int  mqo_check_match_compatible(const mqo_match *h1, const mqo_match *h2, uint64_t g);
/* chain_matches on explicit (query k-min-mer, optional entry) pairs; entries[i].end==0 && hit[i]==0 => miss.
 * Writes matches grouped in emission order with the ref id of each; returns count. */
size_t mqo_chain_matches_explicit(const mqo_kminmer *q, const mqo_entry *r, const uint8_t *hit, size_t n,
                                  mqo_match *out, uint64_t *out_ref, size_t cap);
/* Chain::get_match over one reference's matches (src/chain.rs:147-169); returns 0 for None */
int  mqo_chain_get_match(const mqo_match *matches, size_t n, const mqo_params *p, mqo_coords *out);
/* find_largest_two_chains + determine_best_match (src/mers.rs:104-129): returns index or -1 */
int  mqo_best_of(const uint64_t *scores, size_t n);
/* find_coords (src/mers.rs:131-183) */
void mqo_find_coords(uint64_t q_len, uint64_t r_len, uint64_t ref_id, const mqo_coords *c, mqo_paf *out);
/* PAF text, no trailing newline; returns length written (snprintf semantics) */
int  mqo_format_paf(const char *q_id, const char *r_name, const mqo_paf *paf, char *buf, size_t cap);

/* Per-read branch counters (tests only): which sharp edges of the reference did this read reach? */
typedef struct {
    uint64_t n_kminmers;
    uint64_t n_hits;
    uint64_t n_matches;
    uint64_t n_candidates;     /* references with a Some(get_match) (src/mers.rs:81-86) */
    uint64_t tie;              /* 1 => top two scores equal => None (src/mers.rs:104-108) */
    uint64_t quirk_ext;        /* Match::check true through `|| D` although same-ref/same-strand is false (src/match.rs:39-43) */
    uint64_t quirk_cross_ref;  /* ... of which the hit is on another reference */
    uint64_t rc_ext;           /* extensions through (A && B && C) */
    uint64_t check_fail;       /* hits that fail check and start the next Match */
    uint64_t i32_wrap;         /* gap tests that saw a coordinate >= 2^31 (src/chain.rs:132-142) */
    uint64_t multi_match_refs; /* references with more than one Match (filter_matches_max ran, src/chain.rs:123-129) */
    uint64_t filtered_out;     /* Matches dropped by the co-linearity filter */
    uint64_t clip_start;       /* find_coords clipped at the reference start (src/mers.rs:131-183) */
    uint64_t clip_end;         /* ... at the reference end */
} mqo_diag;

/* ---- the hot path: src/mers.rs:77-102 ---- */
void mqo_find_matches(const mqo_index *ix, const uint8_t *seq, size_t len, const mqo_params *p, mqo_paf *out);
/* batch driver with a pthread pool (CPU baseline).  bases: concatenated reads, offsets: n+1 */
void mqo_map_batch(const mqo_index *ix, const uint8_t *bases, const uint64_t *offsets, uint32_t n,
                   const mqo_params *p, int threads, mqo_paf *out);
/* the same with per-read branch counters (diag: n records, or NULL) */
void mqo_find_matches_diag(const mqo_index *ix, const uint8_t *seq, size_t len, const mqo_params *p, mqo_paf *out, mqo_diag *diag);
void mqo_map_batch_diag(const mqo_index *ix, const uint8_t *bases, const uint64_t *offsets, uint32_t n,
                        const mqo_params *p, int threads, mqo_paf *out, mqo_diag *diag);

#ifdef __cplusplus
}
#endif
#endif
