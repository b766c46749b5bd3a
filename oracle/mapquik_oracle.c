/*
 * mapquik_oracle.c -- CPU restatement of mapquik's seeding + pseudo-chaining path.
 * TEST INFRASTRUCTURE ONLY (see mapquik_oracle.h).  Citations are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "mapquik_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ params */

/* src/main.rs:174-188 */
void mqo_params_default(mqo_params *p) {
    p->k = 5;
    p->l = 31;
    p->density = 0.01;
    p->use_hpc = 1;
    p->c = 4;
    p->s = 11;
    p->g = 2000;
}

/* ------------------------------------------------------------------ ntHash-1 (64-bit)
 * Third-party (rust-seq2kminmers, unpinned; ntHash-1 as published by Mohamadi et al. 2016 and
 * implemented by the `nthash` crate): seeds, fwd/rev definitions and the canonical min.
 * Rust's rotate_left/right take the amount modulo 64.
 */
static inline uint64_t rol64(uint64_t x, unsigned r) { r &= 63u; return r ? (x << r) | (x >> (64u - r)) : x; }
static inline uint64_t ror64(uint64_t x, unsigned r) { r &= 63u; return r ? (x >> r) | (x << (64u - r)) : x; }

#define NT_A 0x3c8bfbb395c60474ULL
#define NT_C 0x3193c18562a02b4cULL
#define NT_G 0x20323ed082572324ULL
#define NT_T 0x295549f54be24456ULL

uint64_t mqo_nt_seed(uint8_t c) {
    switch (c) {
    case 'A': return NT_A;
    case 'C': return NT_C;
    case 'G': return NT_G;
    case 'T': return NT_T;
    default: return 0; /* 'N' and (decision D10) every other byte */
    }
}
/* seed of the complementary base */
static inline uint64_t nt_seed_rc(uint8_t c) {
    switch (c) {
    case 'A': return NT_T;
    case 'C': return NT_G;
    case 'G': return NT_C;
    case 'T': return NT_A;
    default: return 0;
    }
}

/* fh = XOR_j rol(h(s[i+j]), l-1-j) */
uint64_t mqo_ntf64(const uint8_t *s, size_t i, size_t l) {
    uint64_t out = 0;
    for (size_t j = 0; j < l; j++) out ^= rol64(mqo_nt_seed(s[i + j]), (unsigned)((l - 1 - j) & 63));
    return out;
}
/* rh = XOR_j rol(h(comp(s[i+j])), j) */
uint64_t mqo_ntr64(const uint8_t *s, size_t i, size_t l) {
    uint64_t out = 0;
    for (size_t j = 0; j < l; j++) out ^= rol64(nt_seed_rc(s[i + j]), (unsigned)(j & 63));
    return out;
}
uint64_t mqo_ntc64(const uint8_t *s, size_t i, size_t l) {
    uint64_t f = mqo_ntf64(s, i, l), r = mqo_ntr64(s, i, l);
    return f < r ? f : r;
}

/* Hedge for the unpinned seeding decisions: the frozen reading is variant 0.  The other variants exist
 * so that tools/check_against_upstream.sh can tell, on a machine that can build the real crate, WHICH decision is wrong
 * if the k-min-mer dumps differ -- and the product can then be run with that reading.  Bits (any combination):
 *    1  D3   strict `<` on the density bound instead of `<=`
 *    2  D2   FH is f32: the bound is computed in single precision
 *    4  D2/D12  H is u32: ntHash in 32-bit words (seeds = low halves of the 64-bit seeds, rotations mod 32) and a 32-bit bound
 *            -- what a 16-lane AVX-512 implementation of the crate's SIMD modes may use
 *    8  D5   a minimizer's position = raw index of the LAST base of its first base's homopolymer run (frozen: the run head)
 *   16  D6   end = raw position of the last compressed base of the last minimizer's l-mer (frozen: pos[k-1] + l - 1, raw l)
 *   32  D8   rev = reversed tuple <= forward tuple (frozen: strict <; differs on palindromic tuples only)
 *   64  --   NOT a reading of the crate: the product's opt-in cheap tuple hash (MQ_FLAG_FAST_KH) in SipHash's place, see mqo_tuple_hash_fast
 * The product has the same six switches (mq_params.flags bits 8..13, include/mapquik_hip.h): tests/ set a variant here and the same
 * value there and compare the two, variant by variant; everything else in the repository runs the frozen reading. */
static int g_variant = 0;
void mqo_set_variant(int v) { g_variant = v; }
int mqo_get_variant(void) { return g_variant; }
static inline int keep_hash(uint64_t h, uint64_t bound) { return (g_variant & 1) ? h < bound : h <= bound; }

/* hash_bound = ((density as FH) * (H::MAX as FH)) as H ; Rust float->int casts saturate, NaN -> 0 */
uint64_t mqo_density_bound(double density) {
    if (g_variant & 4) { /* H = u32 */
        if (g_variant & 2) {
            float f = (float)density * 4294967295.0f;
            if (!(f > 0.0f)) return 0;
            if (f >= 4294967296.0f) return 0xFFFFFFFFull;
            return (uint64_t)(uint32_t)f;
        }
        double d = density * 4294967295.0;
        if (!(d > 0.0)) return 0;
        if (d >= 4294967296.0) return 0xFFFFFFFFull;
        return (uint64_t)(uint32_t)d;
    }
    if (g_variant & 2) {
        float f = (float)density * 18446744073709551615.0f;
        if (!(f > 0.0f)) return 0;
        if (f >= 18446744073709551616.0f) return UINT64_MAX;
        return (uint64_t)f;
    }
    double d = density * 18446744073709551615.0; /* u64::MAX as f64 == 2^64 */
    if (!(d > 0.0)) return 0;
    if (d >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)d;
}

/* variant 4: canonical ntHash of c[i, i+l) in 32-bit words, zero-extended */
static inline uint32_t rol32(uint32_t x, unsigned r) { r &= 31; return r ? (x << r) | (x >> (32 - r)) : x; }
static uint64_t ntc32(const uint8_t *c, size_t i, size_t l) {
    uint32_t f = 0, r = 0;
    for (size_t j = 0; j < l; j++) {
        f ^= rol32((uint32_t)mqo_nt_seed(c[i + j]), (unsigned)(l - 1 - j));
        r ^= rol32((uint32_t)nt_seed_rc(c[i + j]), (unsigned)j);
    }
    return f < r ? f : r;
}

/* ------------------------------------------------------------------ SipHash (Aumasson & Bernstein)
 * Rust's std DefaultHasher is SipHash-1-3 with a zero key.  Generic (c,d) rounds so the
 * published SipHash-2-4 reference vector can pin the round function in tests.
 */
#define SIPROUND(v0, v1, v2, v3) \
    do {                         \
        v0 += v1;                \
        v1 = rol64(v1, 13);      \
        v1 ^= v0;                \
        v0 = rol64(v0, 32);      \
        v2 += v3;                \
        v3 = rol64(v3, 16);      \
        v3 ^= v2;                \
        v0 += v3;                \
        v3 = rol64(v3, 21);      \
        v3 ^= v0;                \
        v2 += v1;                \
        v1 = rol64(v1, 17);      \
        v1 ^= v2;                \
        v2 = rol64(v2, 32);      \
    } while (0)

uint64_t mqo_siphash(const uint8_t *msg, size_t len, uint64_t k0, uint64_t k1, int c_rounds, int d_rounds) {
    uint64_t v0 = k0 ^ 0x736f6d6570736575ULL;
    uint64_t v1 = k1 ^ 0x646f72616e646f6dULL;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ULL;
    uint64_t v3 = k1 ^ 0x7465646279746573ULL;
    size_t nblk = len / 8;
    for (size_t b = 0; b < nblk; b++) {
        uint64_t m = 0;
        for (int j = 0; j < 8; j++) m |= (uint64_t)msg[b * 8 + j] << (8 * j);
        v3 ^= m;
        for (int r = 0; r < c_rounds; r++) SIPROUND(v0, v1, v2, v3);
        v0 ^= m;
    }
    uint64_t last = (uint64_t)(len & 0xff) << 56;
    for (size_t j = 0; j < (len & 7); j++) last |= (uint64_t)msg[nblk * 8 + j] << (8 * j);
    v3 ^= last;
    for (int r = 0; r < c_rounds; r++) SIPROUND(v0, v1, v2, v3);
    v0 ^= last;
    v2 ^= 0xff;
    for (int r = 0; r < d_rounds; r++) SIPROUND(v0, v1, v2, v3);
    return v0 ^ v1 ^ v2 ^ v3;
}

/* Decision D9: KH = DefaultHasher(SipHash-1-3, key 0) over `mers.hash(&mut h)` for a [u64] slice:
 * Rust writes the length prefix (usize, 8 bytes LE) and then the elements' bytes. */
/* variant bit 64 (the product's MQ_FLAG_FAST_KH, include/mapquik_hip.h): a cheap tuple hash in SipHash's place.  The PAF depends on KH
 * only through equality (index hit / miss / duplicate: src/index.rs:100-104,118-126), so any 64-bit tuple hash without collisions in
 * practice gives the same lines; this one is an add-rotate-xor chain on two 64-bit words, one step per element and six to finish
 * (avalanche 0.45-0.53 per output bit for every input bit, 32-bit halves collide at the birthday rate: profiles/r06_fast_kh.txt). */
static inline uint64_t rotl64_(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
uint64_t mqo_tuple_hash_fast(const uint64_t *mers, size_t k) {
    static const int R[6] = {17, 21, 13, 16, 17, 21};
    uint64_t x = 0x736f6d6570736575ULL ^ (uint64_t)k, y = 0x646f72616e646f6dULL;
    for (size_t i = 0; i < k; i++) {
        x ^= mers[i];
        x += y;
        y = rotl64_(y, 13) ^ x;
        x = rotl64_(x, 32);
    }
    x ^= 0xff;
    for (int j = 0; j < 6; j++) {
        x += y;
        y = rotl64_(y, R[j]) ^ x;
        x = rotl64_(x, 32);
    }
    return x ^ y;
}
uint64_t mqo_tuple_hash(const uint64_t *mers, size_t k) {
    uint8_t buf[8 * 65];
    if (k > 64) k = 64;
    if (g_variant & 64) return mqo_tuple_hash_fast(mers, k);
    uint64_t n = (uint64_t)k;
    for (int j = 0; j < 8; j++) buf[j] = (uint8_t)(n >> (8 * j));
    for (size_t i = 0; i < k; i++)
        for (int j = 0; j < 8; j++) buf[8 + 8 * i + j] = (uint8_t)(mers[i] >> (8 * j));
    return mqo_siphash(buf, 8 * (k + 1), 0, 0, 1, 3);
}

/* ------------------------------------------------------------------ minimizers
 * Decisions D3,D5: the hash runs over the homopolymer-compressed sequence when
 * use_hpc; the reported position is the raw index of the run head of the l-mer's first base;
 * an l-mer is kept iff min(fh,rh) <= bound; l-mers are visited left to right.
 */
size_t mqo_minimizers_naive(const uint8_t *seq, size_t len, const mqo_params *p, mqo_minimizer *out, size_t cap) {
    size_t l = (size_t)p->l;
    uint64_t bound = mqo_density_bound(p->density);
    uint8_t *c = (uint8_t *)malloc(len ? len : 1);
    uint64_t *pos = (uint64_t *)malloc((len ? len : 1) * sizeof(uint64_t));
    size_t n = 0;
    for (size_t i = 0; i < len; i++) {
        if (p->use_hpc && i > 0 && seq[i] == seq[i - 1]) continue;
        c[n] = seq[i];
        pos[n] = i;
        n++;
    }
    size_t cnt = 0;
    if (l >= 1 && n >= l) {
        for (size_t j = 0; j + l <= n; j++) {
            uint64_t h = mqo_ntc64(c, j, l);
            if (keep_hash(h, bound)) {
                if (out && cnt < cap) {
                    out[cnt].pos = pos[j];
                    out[cnt].hash = h;
                }
                cnt++;
            }
        }
    }
    free(c);
    free(pos);
    return cnt;
}

/* Rolling form (same results as the naive form; tested).  Roll step of the nthash crate:
 * fh' = rol(fh,1) ^ rol(h(out),l) ^ h(in);  rh' = ror(rh,1) ^ ror(hc(out),1) ^ rol(hc(in),l-1) */
/* growable single-pass core: *arr is malloc'ed/realloc'ed when grow != 0, else capped at cap */
static size_t minimizers_core(const uint8_t *seq, size_t len, const mqo_params *p, mqo_minimizer **arr, size_t cap, int grow) {
    size_t l = (size_t)p->l;
    if (l < 1) return 0;
    uint64_t bound = mqo_density_bound(p->density);
    /* ring of the last l compressed bases and their raw positions */
    uint8_t *rc = (uint8_t *)malloc(l);
    uint64_t *rp = (uint64_t *)malloc(l * sizeof(uint64_t));
    size_t n = 0; /* compressed bases seen */
    uint64_t fh = 0, rh = 0;
    size_t cnt = 0;
    int hpc = p->use_hpc;
    mqo_minimizer *out = *arr;
    for (size_t i = 0; i < len; i++) {
        uint8_t b = seq[i];
        if (hpc && i > 0 && b == seq[i - 1]) continue;
        size_t slot = n % l;
        if (n < l) {
            /* building the first window: fh = XOR rol(h(c_j), l-1-j) ; rh = XOR rol(hc(c_j), j) */
            fh ^= rol64(mqo_nt_seed(b), (unsigned)((l - 1 - n) & 63));
            rh ^= rol64(nt_seed_rc(b), (unsigned)(n & 63));
        } else {
            uint8_t o = rc[slot]; /* base leaving the window */
            fh = rol64(fh, 1) ^ rol64(mqo_nt_seed(o), (unsigned)(l & 63)) ^ mqo_nt_seed(b);
            rh = ror64(rh, 1) ^ ror64(nt_seed_rc(o), 1) ^ rol64(nt_seed_rc(b), (unsigned)((l - 1) & 63));
        }
        rc[slot] = b;
        rp[slot] = i;
        n++;
        if (n >= l) {
            uint64_t h = fh < rh ? fh : rh;
            if (keep_hash(h, bound)) {
                if (grow && cnt >= cap) {
                    cap = cap ? cap * 2 : 512;
                    out = (mqo_minimizer *)realloc(out, cap * sizeof(*out));
                }
                if (out && cnt < cap) {
                    out[cnt].pos = rp[n % l]; /* oldest element = first base of this window */
                    out[cnt].hash = h;
                }
                cnt++;
            }
        }
    }
    free(rc);
    free(rp);
    *arr = out;
    return cnt;
}

size_t mqo_minimizers(const uint8_t *seq, size_t len, const mqo_params *p, mqo_minimizer *out, size_t cap) {
    mqo_minimizer *a = out;
    return minimizers_core(seq, len, p, &a, out ? cap : 0, 0);
}

/* ------------------------------------------------------------------ k-min-mers
 * KminmersIterator (rust-seq2kminmers, call sites src/mers.rs:27,53): every k consecutive
 * minimizers (D11); start = pos[0]; end = pos[k-1] + l - 1 (D6); offset = running count (D7);
 * rev = reversed tuple < forward tuple, lexicographic on the hashes (D8); hash = D9.
 */
static void kminmer_from_window(const mqo_minimizer *w, size_t k, size_t l, uint64_t offset, mqo_kminmer *o) {
    uint64_t fwd[64], rev[64];
    for (size_t i = 0; i < k; i++) {
        fwd[i] = w[i].hash;
        rev[k - 1 - i] = w[i].hash;
    }
    int is_rev = (g_variant & 32) ? 1 : 0; /* all elements equal: `<` says forward, `<=` says reversed */
    for (size_t i = 0; i < k; i++) {
        if (rev[i] < fwd[i]) { is_rev = 1; break; }
        if (rev[i] > fwd[i]) { is_rev = 0; break; }
    }
    o->hash = mqo_tuple_hash(is_rev ? rev : fwd, k);
    o->start = w[0].pos;
    o->end = w[k - 1].pos + l - 1;
    o->offset = offset;
    o->rev = is_rev;
    o->_pad = 0;
}

/* Diagnostic variants 4 / 8 / 16 (see mqo_set_variant): the minimizers by the naive form over the compressed sequence, with the
 * raw position of each window's LAST compressed base beside them (last[]).  Returns the count; *m and *last are malloc'ed. */
static size_t minimizers_variant(const uint8_t *seq, size_t len, const mqo_params *p, mqo_minimizer **m, uint64_t **last) {
    const size_t l = (size_t)p->l;
    const uint64_t bound = mqo_density_bound(p->density);
    uint8_t *c = (uint8_t *)malloc(len ? len : 1);
    uint64_t *head = (uint64_t *)malloc((len ? len : 1) * sizeof(uint64_t));
    size_t n = 0;
    for (size_t i = 0; i < len; i++) {
        if (p->use_hpc && i > 0 && seq[i] == seq[i - 1]) continue;
        c[n] = seq[i];
        head[n] = i;
        n++;
    }
    size_t cnt = 0, cap = 0;
    *m = NULL;
    *last = NULL;
    if (l >= 1 && n >= l) {
        for (size_t j = 0; j + l <= n; j++) {
            const uint64_t h = (g_variant & 4) ? ntc32(c, j, l) : mqo_ntc64(c, j, l);
            if (!keep_hash(h, bound)) continue;
            if (cnt >= cap) {
                cap = cap ? cap * 2 : 512;
                *m = (mqo_minimizer *)realloc(*m, cap * sizeof(**m));
                *last = (uint64_t *)realloc(*last, cap * sizeof(**last));
            }
            /* position: run head, or (variant 8) the run's last base = the base before the next run head */
            (*m)[cnt].pos = (g_variant & 8) ? ((j + 1 < n ? head[j + 1] : (uint64_t)len) - 1) : head[j];
            (*m)[cnt].hash = h;
            (*last)[cnt] = head[j + l - 1];
            cnt++;
        }
    }
    free(c);
    free(head);
    return cnt;
}

/* returns a malloc'ed array (or NULL when empty) */
static size_t kminmers_dyn(const uint8_t *seq, size_t len, const mqo_params *p, mqo_kminmer **out) {
    size_t k = (size_t)p->k, l = (size_t)p->l;
    *out = NULL;
    if (k < 1 || k > 64 || l < 1) return 0;
    mqo_minimizer *m = NULL;
    uint64_t *last = NULL;
    const int var = g_variant & (4 | 8 | 16);
    size_t nmin = var ? minimizers_variant(seq, len, p, &m, &last) : minimizers_core(seq, len, p, &m, 0, 1);
    if (nmin < k) { free(m); free(last); return 0; }
    size_t cnt = nmin - k + 1;
    mqo_kminmer *km = (mqo_kminmer *)malloc(cnt * sizeof(*km));
    for (size_t i = 0; i < cnt; i++) {
        kminmer_from_window(m + i, k, l, (uint64_t)i, km + i);
        if (g_variant & 16) km[i].end = last[i + k - 1];
    }
    free(m);
    free(last);
    *out = km;
    return cnt;
}

size_t mqo_kminmers(const uint8_t *seq, size_t len, const mqo_params *p, mqo_kminmer *out, size_t cap) {
    mqo_kminmer *km = NULL;
    size_t cnt = kminmers_dyn(seq, len, p, &km);
    if (out)
        for (size_t i = 0; i < cnt && i < cap; i++) out[i] = km[i];
    free(km);
    return cnt;
}

/* ------------------------------------------------------------------ index (src/index.rs)
 * DashMap<KH, Entry> with the identity KnownHasher (src/index.rs:11-39) restated as an
 * open-addressed table; only the map semantics matter: insert; if a previous value existed
 * overwrite with Entry::empty() (src/index.rs:94-104); get() hides tombstones (118-126).
 */
typedef struct {
    char *name;
    uint64_t len;
} mqo_ref;

struct mqo_index {
    uint64_t *keys;
    mqo_entry *vals;
    uint8_t *used;
    uint64_t cap; /* power of two */
    uint64_t n;   /* keys stored */
    mqo_ref *refs;
    uint64_t n_refs, cap_refs;
};

static inline uint64_t slot_of(uint64_t h, uint64_t cap) {
    /* keys are already hashes in production, but KATs use small integers: mix */
    h ^= h >> 32;
    h *= 0x9E3779B97F4A7C15ULL;
    h ^= h >> 29;
    return h & (cap - 1);
}

mqo_index *mqo_index_new(void) {
    mqo_index *ix = (mqo_index *)calloc(1, sizeof(*ix));
    ix->cap = 1024;
    ix->keys = (uint64_t *)calloc(ix->cap, sizeof(uint64_t));
    ix->vals = (mqo_entry *)calloc(ix->cap, sizeof(mqo_entry));
    ix->used = (uint8_t *)calloc(ix->cap, 1);
    return ix;
}

void mqo_index_free(mqo_index *ix) {
    if (!ix) return;
    for (uint64_t i = 0; i < ix->n_refs; i++) free(ix->refs[i].name);
    free(ix->refs);
    free(ix->keys);
    free(ix->vals);
    free(ix->used);
    free(ix);
}

static void index_grow(mqo_index *ix, uint64_t newcap) {
    uint64_t *ok = ix->keys;
    mqo_entry *ov = ix->vals;
    uint8_t *ou = ix->used;
    uint64_t oc = ix->cap;
    ix->cap = newcap;
    ix->keys = (uint64_t *)calloc(newcap, sizeof(uint64_t));
    ix->vals = (mqo_entry *)calloc(newcap, sizeof(mqo_entry));
    ix->used = (uint8_t *)calloc(newcap, 1);
    for (uint64_t i = 0; i < oc; i++) {
        if (!ou[i]) continue;
        uint64_t s = slot_of(ok[i], newcap);
        while (ix->used[s]) s = (s + 1) & (newcap - 1);
        ix->used[s] = 1;
        ix->keys[s] = ok[i];
        ix->vals[s] = ov[i];
    }
    free(ok);
    free(ov);
    free(ou);
}

static void index_reserve(mqo_index *ix, uint64_t n_keys) {
    uint64_t need = 1024;
    while (need * 7 < n_keys * 10) need <<= 1; /* load <= 0.7 */
    if (need > ix->cap) index_grow(ix, need);
}

/* Index::add (src/index.rs:94-97) / add_with_mer (100-104) */
void mqo_index_add(mqo_index *ix, uint64_t h, uint64_t id, uint64_t start, uint64_t end, uint64_t offset, int rc) {
    if ((ix->n + 1) * 10 > ix->cap * 7) index_grow(ix, ix->cap * 2);
    uint64_t s = slot_of(h, ix->cap);
    while (ix->used[s]) {
        if (ix->keys[s] == h) {
            /* e.is_some() => insert(h, Entry::empty()) : id 0, start 0, end 0, offset 0, rc false */
            memset(&ix->vals[s], 0, sizeof(mqo_entry));
            return;
        }
        s = (s + 1) & (ix->cap - 1);
    }
    ix->used[s] = 1;
    ix->keys[s] = h;
    mqo_entry *e = &ix->vals[s];
    e->id = id;
    e->start = start;
    e->end = end;
    e->offset = offset;
    e->rc = rc ? 1 : 0;
    e->_pad = 0;
    ix->n++;
}

/* ReadOnlyIndex::get (src/index.rs:118-126): Entry::is_empty() <=> end == 0 (src/index.rs:67-69) */
const mqo_entry *mqo_index_get(const mqo_index *ix, uint64_t h) {
    uint64_t s = slot_of(h, ix->cap);
    while (ix->used[s]) {
        if (ix->keys[s] == h) return ix->vals[s].end != 0 ? &ix->vals[s] : NULL;
        s = (s + 1) & (ix->cap - 1);
    }
    return NULL;
}

/* Index::get_count (src/index.rs:90-92) */
uint64_t mqo_index_count(const mqo_index *ix) {
    uint64_t c = 0;
    for (uint64_t i = 0; i < ix->cap; i++)
        if (ix->used[i] && ix->vals[i].end != 0) c++;
    return c;
}
uint64_t mqo_index_keys(const mqo_index *ix) { return ix->n; }

void mqo_index_set_ref(mqo_index *ix, uint64_t ref_idx, const char *name, uint64_t len) {
    if (ref_idx >= ix->cap_refs) {
        uint64_t nc = ix->cap_refs ? ix->cap_refs : 16;
        while (nc <= ref_idx) nc *= 2;
        ix->refs = (mqo_ref *)realloc(ix->refs, nc * sizeof(mqo_ref));
        memset(ix->refs + ix->cap_refs, 0, (nc - ix->cap_refs) * sizeof(mqo_ref));
        ix->cap_refs = nc;
    }
    free(ix->refs[ref_idx].name);
    ix->refs[ref_idx].name = strdup(name ? name : "");
    ix->refs[ref_idx].len = len;
    if (ref_idx + 1 > ix->n_refs) ix->n_refs = ref_idx + 1;
}
uint64_t mqo_index_ref_len(const mqo_index *ix, uint64_t ref_idx) { return ref_idx < ix->n_refs ? ix->refs[ref_idx].len : 0; }
const char *mqo_index_ref_name(const mqo_index *ix, uint64_t ref_idx) {
    return (ref_idx < ix->n_refs && ix->refs[ref_idx].name) ? ix->refs[ref_idx].name : "";
}
uint64_t mqo_index_n_refs(const mqo_index *ix) { return ix->n_refs; }

/* mers::ref_extract (src/mers.rs:15-38) */
uint64_t mqo_ref_extract(mqo_index *ix, uint64_t ref_idx, const uint8_t *seq, size_t len, const mqo_params *p) {
    if (len < p->l + p->k - 1) return 0; /* src/mers.rs:18 */
    mqo_kminmer *km = NULL;
    size_t n = kminmers_dyn(seq, len, p, &km);
    if (!n) return 0;
    for (size_t i = 0; i < n; i++)
        mqo_index_add(ix, km[i].hash, ref_idx, km[i].start, km[i].end, km[i].offset, km[i].rev); /* src/index.rs:57-58 */
    free(km);
    return (uint64_t)n;
}

typedef struct {
    const uint8_t *bases;
    const uint64_t *offsets;
    uint32_t n_refs;
    const mqo_params *p;
    mqo_kminmer **km;
    uint64_t *cnt;
    volatile uint32_t *next;
} build_job;

static void *build_worker(void *arg) {
    build_job *j = (build_job *)arg;
    for (;;) {
        uint32_t r = __sync_fetch_and_add(j->next, 1);
        if (r >= j->n_refs) break;
        const uint8_t *seq = j->bases + j->offsets[r];
        size_t len = (size_t)(j->offsets[r + 1] - j->offsets[r]);
        j->cnt[r] = 0;
        j->km[r] = NULL;
        if (len < j->p->l + j->p->k - 1) continue;
        j->cnt[r] = kminmers_dyn(seq, len, j->p, &j->km[r]);
    }
    return NULL;
}

/* Same final map as calling mqo_ref_extract for r = 0..n_refs-1 (the map state is independent of
 * insertion order: src/index.rs:94-104), with the extraction spread over threads. */
uint64_t mqo_index_build_mt(mqo_index *ix, const uint8_t *bases, const uint64_t *offsets, uint32_t n_refs,
                            const mqo_params *p, int threads, uint64_t *per_ref_counts) {
    if (threads < 1) threads = 1;
    mqo_kminmer **km = (mqo_kminmer **)calloc(n_refs ? n_refs : 1, sizeof(*km));
    uint64_t *cnt = (uint64_t *)calloc(n_refs ? n_refs : 1, sizeof(*cnt));
    volatile uint32_t next = 0;
    build_job job = {bases, offsets, n_refs, p, km, cnt, &next};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, build_worker, &job);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    uint64_t total = 0;
    for (uint32_t r = 0; r < n_refs; r++) total += cnt[r];
    index_reserve(ix, ix->n + total);
    for (uint32_t r = 0; r < n_refs; r++) {
        for (uint64_t i = 0; i < cnt[r]; i++)
            mqo_index_add(ix, km[r][i].hash, r, km[r][i].start, km[r][i].end, km[r][i].offset, km[r][i].rev);
        if (per_ref_counts) per_ref_counts[r] = cnt[r];
        free(km[r]);
    }
    free(km);
    free(cnt);
    return total;
}

/* ------------------------------------------------------------------ branch counters (tests only)
 * Set per thread by mqo_find_matches_diag: which of the reference's sharp edges did this read reach?  They let the GPU
 * parity tests assert that their inputs really exercise the precedence quirk, ties, the `as i32` wrap and the clipping. */
static __thread mqo_diag *g_diag = NULL;

/* ------------------------------------------------------------------ Match (src/match.rs) */

/* Match::new src/match.rs:20-29 */
void mqo_match_new(mqo_match *m, const mqo_kminmer *q, const mqo_entry *r) {
    m->q_start = q->start;
    m->q_end = q->end;
    m->r_start = r->start;
    m->r_end = r->end;
    m->count = 1;
    m->rc = ((q->rev != 0) != (r->rc != 0));
    m->_pad = 0;
}
/* Match::update src/match.rs:31-37 */
void mqo_match_update(mqo_match *m, const mqo_kminmer *q, const mqo_entry *r) {
    if (m->rc) m->r_start = r->start;
    else m->r_end = r->end;
    m->q_end = q->end;
    m->count += 1;
}
/* Match::check src/match.rs:39-43.  Rust precedence: (A && B && C) || D with
 * C = self.rc && (p.offset as i32 - r.offset as i32 == 1), D = !self.rc && (r.offset as i32 - p.offset as i32 == 1).
 * `as i32` truncates; the i32 subtraction wraps in release builds. */
int mqo_match_check(const mqo_match *m, const mqo_kminmer *q, const mqo_entry *r, const mqo_entry *p) {
    int32_t po = (int32_t)(uint32_t)p->offset, ro = (int32_t)(uint32_t)r->offset;
    int32_t d_rc = (int32_t)((uint32_t)po - (uint32_t)ro);
    int32_t d_fw = (int32_t)((uint32_t)ro - (uint32_t)po);
    int A = (r->id == p->id);
    int B = ((((q->rev != 0) != (r->rc != 0)) ? 1 : 0) == (m->rc ? 1 : 0));
    int C = (m->rc && d_rc == 1);
    int D = (!m->rc && d_fw == 1);
    if (g_diag) {
        if (D && !(A && B)) g_diag->quirk_ext++;      /* forward run extended across references / strands (F8) */
        if (D && !A) g_diag->quirk_cross_ref++;
        if (A && B && C) g_diag->rc_ext++;
        if (!((A && B && C) || D)) g_diag->check_fail++; /* a hit that fails check starts the next Match */
    }
    return (A && B && C) || D;
}

/* chain_matches (src/mers.rs:57-73) + Match::extend (src/match.rs:45-58) on explicit lookups */
size_t mqo_chain_matches_explicit(const mqo_kminmer *q, const mqo_entry *r, const uint8_t *hit, size_t n,
                                  mqo_match *out, uint64_t *out_ref, size_t cap) {
    size_t i = 0, cnt = 0;
    while (i < n) {
        size_t cur = i++; /* query_it.next() */
        if (!hit[cur] || r[cur].end == 0) continue; /* index.get() == None */
        mqo_match h;
        mqo_match_new(&h, &q[cur], &r[cur]);
        const mqo_entry *p = &r[cur];
        uint64_t first_id = r[cur].id;
        /* extend */
        for (;;) {
            if (i >= n) break;                       /* peek() == None */
            if (!hit[i] || r[i].end == 0) { i++; break; } /* miss: consumed, stop */
            if (mqo_match_check(&h, &q[i], &r[i], p)) {
                mqo_match_update(&h, &q[i], &r[i]);
                p = &r[i];
                i++;
            } else break; /* hit that fails check: not consumed */
        }
        if (cnt < cap) {
            out[cnt] = h;
            out_ref[cnt] = first_id; /* matches_per_ref.entry(r.id) with the FIRST entry's id (src/mers.rs:68) */
        }
        cnt++;
    }
    return cnt;
}

/* ------------------------------------------------------------------ Chain (src/chain.rs) */

static int match_eq(const mqo_match *a, const mqo_match *b) { /* #[derive(PartialEq)] src/match.rs:10 */
    return a->q_start == b->q_start && a->q_end == b->q_end && a->r_start == b->r_start && a->r_end == b->r_end &&
           a->count == b->count && (a->rc != 0) == (b->rc != 0);
}

/* i32 helpers: `x as i32` truncation, wrapping sub, wrapping abs, `as usize` sign extension */
static inline int32_t as_i32(uint64_t x) { return (int32_t)(uint32_t)x; }
static inline int32_t wsub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline uint64_t abs_as_usize(int32_t x) {
    int32_t a = x < 0 ? (int32_t)(0u - (uint32_t)x) : x; /* i32::MIN.abs() wraps to MIN in release */
    return (uint64_t)(int64_t)a;
}
/* src/chain.rs:132-136 */
static int fwd_gap_too_long(uint64_t u_q_e, uint64_t u_r_e, uint64_t v_q_s, uint64_t v_r_s, uint64_t g) {
    if (g_diag && ((u_q_e | u_r_e | v_q_s | v_r_s) >> 31)) g_diag->i32_wrap++; /* a coordinate >= 2^31 went through `as i32` */
    int32_t g1 = wsub32(as_i32(v_q_s), as_i32(u_q_e));
    int32_t g2 = wsub32(as_i32(v_r_s), as_i32(u_r_e));
    return abs_as_usize(wsub32(g1, g2)) > g;
}
/* src/chain.rs:138-142 */
static int rc_gap_too_long(uint64_t u_r_s, uint64_t u_q_e, uint64_t v_q_s, uint64_t v_r_e, uint64_t g) {
    if (g_diag && ((u_r_s | u_q_e | v_q_s | v_r_e) >> 31)) g_diag->i32_wrap++;
    int32_t g1 = wsub32(as_i32(v_q_s), as_i32(u_q_e));
    int32_t g2 = wsub32(as_i32(u_r_s), as_i32(v_r_e));
    return abs_as_usize(wsub32(g1, g2)) > g;
}
/* src/chain.rs:43-63 */
int mqo_check_match_compatible(const mqo_match *h1, const mqo_match *h2, uint64_t g) {
    if (match_eq(h1, h2)) return 1;
    if ((h1->rc != 0) != (h2->rc != 0)) return 0;
    const mqo_match *u, *v;
    if (h1->q_start < h2->q_start) { u = h1; v = h2; } else { u = h2; v = h1; }
    if (u->rc) {
        if (u->r_start <= v->r_start || rc_gap_too_long(u->r_start, u->q_end, v->q_start, v->r_end, g)) return 0;
    } else if (v->r_start <= u->r_start || fwd_gap_too_long(u->q_end, u->r_end, v->q_start, v->r_start, g)) {
        return 0;
    }
    return 1;
}

/* get_match (src/chain.rs:147-169) with filter_matches_max (123-129), find_largest_match (93-104),
 * colinear_matches_per_match (65-75) */
int mqo_chain_get_match(const mqo_match *matches, size_t n, const mqo_params *p, mqo_coords *out) {
    if (n == 0) return 0; /* Chain::new of an empty Vec never happens; len_f == 0 => None */
    const mqo_match *first = NULL, *last = NULL;
    size_t len_f = 0;
    uint64_t score = 0;
    if (n > 1) {
        size_t mx = 0;
        uint64_t mx_count = 0;
        for (size_t i = 0; i < n; i++)
            if (matches[i].count > mx_count) { mx = i; mx_count = matches[i].count; }
        const mqo_match *anchor = &matches[mx];
        for (size_t i = 0; i < n; i++) {
            if (mqo_check_match_compatible(anchor, &matches[i], p->g)) {
                if (!first) first = &matches[i];
                last = &matches[i];
                len_f++;
                score += matches[i].count;
            } else if (g_diag) g_diag->filtered_out++;
        }
        if (g_diag) g_diag->multi_match_refs++;
    } else {
        first = last = &matches[0];
        len_f = 1;
        score = matches[0].count;
    }
    if (len_f == 0) return 0;
    uint64_t mapq = ((p->s != 0 && p->c != 0) && (len_f >= p->c || score >= p->s)) ? 60 : 0;
    int rc = first->rc != 0;
    out->rc = rc;
    out->_pad = 0;
    out->q_start = first->q_start;
    out->q_end = last->q_end - 1;
    if (rc && len_f > 1) {
        out->r_start = last->r_start;
        out->r_end = first->r_end - 1;
    } else {
        out->r_start = first->r_start;
        out->r_end = last->r_end - 1;
    }
    out->score = score;
    out->mapq = mapq;
    return 1;
}

/* find_largest_two_chains (src/mers.rs:110-129) + determine_best_match (104-108) */
int mqo_best_of(const uint64_t *scores, size_t n) {
    if (n == 0) return -1;
    if (n == 1) return 0; /* src/mers.rs:90 */
    size_t mx = 0, second = 0;
    uint64_t mx_c = 0, second_c = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t c = scores[i];
        if (c > mx_c) {
            second = mx;
            second_c = mx_c;
            mx = i;
            mx_c = c;
        } else if (c > second_c) {
            second = i;
            second_c = c;
        }
    }
    (void)second;
    if (mx_c == second_c) return -1;
    return (int)mx;
}

/* find_coords (src/mers.rs:131-183); usize arithmetic wraps */
void mqo_find_coords(uint64_t q_len, uint64_t r_len, uint64_t ref_id, const mqo_coords *c, mqo_paf *out) {
    uint64_t q_start = c->q_start, q_end = c->q_end, r_start = c->r_start, r_end = c->r_end;
    uint64_t final_r_start, final_r_end, exc_s, exc_e;
    uint64_t tail = q_len - q_end - 1;
    if (!c->rc) {
        if (r_start >= q_start) { final_r_start = r_start - q_start; exc_s = q_start; }
        else { final_r_start = 0; exc_s = r_start; if (g_diag) g_diag->clip_start++; }
        if (r_end + tail <= r_len - 1) { final_r_end = r_end + tail; exc_e = tail; }
        else { final_r_end = r_len - 1; exc_e = r_len - r_end - 1; if (g_diag) g_diag->clip_end++; }
    } else {
        if (r_end + q_start <= r_len - 1) { final_r_end = r_end + q_start; exc_s = q_start; }
        else { final_r_end = r_len - 1; exc_s = r_len - r_end - 1; if (g_diag) g_diag->clip_end++; }
        if (r_start >= tail) { final_r_start = r_start - tail; exc_e = tail; }
        else { final_r_start = 0; exc_e = r_start; if (g_diag) g_diag->clip_start++; }
    }
    out->mapped = 1;
    out->rc = c->rc;
    out->ref_id = ref_id;
    out->q_len = q_len;
    out->q_start = q_start - exc_s;
    out->q_end = q_end + exc_e;
    out->r_len = r_len;
    out->r_start = final_r_start;
    out->r_end = final_r_end;
    out->score = c->score;
    out->mapq = c->mapq;
}

/* format! at src/mers.rs:181: column 11 is r_len again, column 10 the score */
int mqo_format_paf(const char *q_id, const char *r_name, const mqo_paf *paf, char *buf, size_t cap) {
    return snprintf(buf, cap, "%s\t%llu\t%llu\t%llu\t%s\t%s\t%llu\t%llu\t%llu\t%llu\t%llu\t%llu", q_id,
                    (unsigned long long)paf->q_len, (unsigned long long)paf->q_start, (unsigned long long)paf->q_end,
                    paf->rc ? "-" : "+", r_name, (unsigned long long)paf->r_len, (unsigned long long)paf->r_start,
                    (unsigned long long)paf->r_end, (unsigned long long)paf->score, (unsigned long long)paf->r_len,
                    (unsigned long long)paf->mapq);
}

/* ------------------------------------------------------------------ find_matches (src/mers.rs:77-102) */
void mqo_find_matches(const mqo_index *ix, const uint8_t *seq, size_t len, const mqo_params *p, mqo_paf *out) {
    memset(out, 0, sizeof(*out));
    out->q_len = len;
    if (len < p->l + p->k - 1) return; /* extract(): None (src/mers.rs:44) */
    mqo_kminmer *km = NULL;
    size_t n = kminmers_dyn(seq, len, p, &km);
    if (!n) return;
    mqo_entry *ent = (mqo_entry *)calloc(n, sizeof(*ent));
    uint8_t *hit = (uint8_t *)calloc(n, 1);
    for (size_t i = 0; i < n; i++) {
        const mqo_entry *e = mqo_index_get(ix, km[i].hash); /* src/mers.rs:63 / src/match.rs:47 */
        if (e) { ent[i] = *e; hit[i] = 1; }
    }
    mqo_match *ms = (mqo_match *)malloc(n * sizeof(*ms));
    uint64_t *mref = (uint64_t *)malloc(n * sizeof(*mref));
    size_t nm = mqo_chain_matches_explicit(km, ent, hit, n, ms, mref, n);
    /* per reference (HashMap<usize, Vec<Match>>; per-ref order = query order): src/mers.rs:81-86 */
    uint8_t *done = (uint8_t *)calloc(nm ? nm : 1, 1);
    mqo_match *grp = (mqo_match *)malloc((nm ? nm : 1) * sizeof(*grp));
    uint64_t *scores = (uint64_t *)malloc((nm ? nm : 1) * sizeof(uint64_t));
    uint64_t *ids = (uint64_t *)malloc((nm ? nm : 1) * sizeof(uint64_t));
    mqo_coords *coords = (mqo_coords *)malloc((nm ? nm : 1) * sizeof(mqo_coords));
    size_t ncand = 0;
    for (size_t i = 0; i < nm; i++) {
        if (done[i]) continue;
        size_t ng = 0;
        for (size_t j = i; j < nm; j++)
            if (!done[j] && mref[j] == mref[i]) { grp[ng++] = ms[j]; done[j] = 1; }
        mqo_coords cc;
        if (mqo_chain_get_match(grp, ng, p, &cc)) {
            coords[ncand] = cc;
            ids[ncand] = mref[i];
            scores[ncand] = cc.score;
            ncand++;
        }
    }
    int best = mqo_best_of(scores, ncand);
    if (g_diag) {
        g_diag->n_kminmers = n;
        for (size_t i = 0; i < n; i++) g_diag->n_hits += hit[i];
        g_diag->n_matches = nm;
        g_diag->n_candidates = ncand;
        g_diag->tie = (ncand > 1 && best < 0);
    }
    if (best >= 0) {
        mqo_find_coords(len, mqo_index_ref_len(ix, ids[best]), ids[best], &coords[best], out);
    }
    free(km); free(ent); free(hit); free(ms); free(mref); free(done); free(grp); free(scores); free(ids); free(coords);
}

void mqo_find_matches_diag(const mqo_index *ix, const uint8_t *seq, size_t len, const mqo_params *p, mqo_paf *out, mqo_diag *diag) {
    memset(diag, 0, sizeof(*diag));
    g_diag = diag;
    mqo_find_matches(ix, seq, len, p, out);
    g_diag = NULL;
}

/* ------------------------------------------------------------------ batch driver (CPU baseline) */
typedef struct {
    const mqo_index *ix;
    const uint8_t *bases;
    const uint64_t *offsets;
    uint32_t n;
    const mqo_params *p;
    mqo_paf *out;
    volatile uint32_t *next;
    mqo_diag *diag; /* NULL in the timed baseline */
} map_job;

static void *map_worker(void *arg) {
    map_job *j = (map_job *)arg;
    for (;;) {
        uint32_t lo = __sync_fetch_and_add(j->next, 16);
        if (lo >= j->n) break;
        uint32_t hi = lo + 16 < j->n ? lo + 16 : j->n;
        for (uint32_t r = lo; r < hi; r++) {
            const uint8_t *sq = j->bases + j->offsets[r];
            const size_t ln = (size_t)(j->offsets[r + 1] - j->offsets[r]);
            if (j->diag) mqo_find_matches_diag(j->ix, sq, ln, j->p, &j->out[r], &j->diag[r]);
            else mqo_find_matches(j->ix, sq, ln, j->p, &j->out[r]);
        }
    }
    return NULL;
}

void mqo_map_batch(const mqo_index *ix, const uint8_t *bases, const uint64_t *offsets, uint32_t n,
                   const mqo_params *p, int threads, mqo_paf *out) {
    mqo_map_batch_diag(ix, bases, offsets, n, p, threads, out, NULL);
}

void mqo_map_batch_diag(const mqo_index *ix, const uint8_t *bases, const uint64_t *offsets, uint32_t n,
                        const mqo_params *p, int threads, mqo_paf *out, mqo_diag *diag) {
    if (threads < 1) threads = 1;
    volatile uint32_t next = 0;
    map_job job = {ix, bases, offsets, n, p, out, &next, diag};
    if (threads == 1) { map_worker(&job); return; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, map_worker, &job);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
}
