/*
 * mqsim.c -- seeded synthetic genomes and HiFi-like reads (test/bench input generator).
 *
 * Not part of the product path and not part of the oracle: it only makes inputs.  It stands in
 * for the pieces of the reference's example/experiment recipes that are not in this image
 * (pbsim, ecoli.genome.fa, CHM13v2.0): example/simulate_pbsim.sh:7-14, experiments/simulate_chm13.sh.
 * Read names follow paftools' pbsim2fq convention seen in example/nearperfect-ecoli.100.fa:
 *   S1_<n>!<chr>!<start>!<end>!<strand>
 * Everything is a pure function of the seeds (per-read RNG streams), independent of thread count.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

static const char ACGT[4] = {'A', 'C', 'G', 'T'};

typedef struct {
    uint8_t *out;
    uint64_t lo, hi, seed;
} gen_job;

static void *gen_worker(void *a) {
    gen_job *j = (gen_job *)a;
    /* block b of 1 Mi bases has its own stream: deterministic for any thread split on 1 Mi boundaries */
    for (uint64_t b = j->lo; b < j->hi; b += (1u << 20)) {
        uint64_t s = j->seed ^ (0xD1B54A32D192ED03ULL * ((b >> 20) + 1));
        uint64_t e = b + (1u << 20) < j->hi ? b + (1u << 20) : j->hi;
        uint64_t i = b;
        while (i < e) {
            uint64_t r = splitmix64(&s);
            for (int t = 0; t < 32 && i < e; t++, i++) {
                j->out[i] = (uint8_t)ACGT[r & 3];
                r >>= 2;
            }
        }
    }
    return NULL;
}

/* i.i.d. uniform ACGT */
void mqsim_genome(uint8_t *out, uint64_t len, uint64_t seed, int threads) {
    if (threads < 1) threads = 1;
    uint64_t nblk = (len + (1u << 20) - 1) >> 20;
    if ((uint64_t)threads > nblk) threads = nblk ? (int)nblk : 1;
    pthread_t th[256];
    gen_job jobs[256];
    if (threads > 256) threads = 256;
    for (int t = 0; t < threads; t++) {
        uint64_t b0 = nblk * (uint64_t)t / (uint64_t)threads, b1 = nblk * (uint64_t)(t + 1) / (uint64_t)threads;
        jobs[t].out = out;
        jobs[t].lo = b0 << 20;
        jobs[t].hi = (b1 << 20) < len ? (b1 << 20) : len;
        jobs[t].seed = seed;
        pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* Plant repeats: n_seg copies of random segments (length U[min_len,max_len]) pasted elsewhere with
 * per-base divergence `div` (substitutions), plus n_tandem tandem arrays (unit U[2,200], total U[min_len,max_len]).
 * Sequential and deterministic. */
void mqsim_plant_repeats(uint8_t *g, uint64_t len, uint64_t seed, uint64_t n_seg, uint64_t n_tandem, uint64_t min_len,
                         uint64_t max_len, double div) {
    uint64_t s = seed ^ 0xA5A5A5A55A5A5A5AULL;
    if (len < 4 * max_len) return;
    for (uint64_t i = 0; i < n_seg; i++) {
        uint64_t L = min_len + splitmix64(&s) % (max_len - min_len + 1);
        uint64_t src = splitmix64(&s) % (len - L);
        uint64_t dst = splitmix64(&s) % (len - L);
        if ((src < dst ? dst - src : src - dst) < L) continue;
        for (uint64_t t = 0; t < L; t++) {
            uint8_t b = g[src + t];
            if (div > 0 && u01(&s) < div) b = (uint8_t)ACGT[splitmix64(&s) & 3];
            g[dst + t] = b;
        }
    }
    for (uint64_t i = 0; i < n_tandem; i++) {
        uint64_t L = min_len + splitmix64(&s) % (max_len - min_len + 1);
        uint64_t unit = 2 + splitmix64(&s) % 199;
        uint64_t dst = splitmix64(&s) % (len - L);
        for (uint64_t t = unit; t < L; t++) g[dst + t] = g[dst + t - unit];
    }
}

/* Transposon-like repeat families (maize-shaped genomes, experiments/simulate_maize.sh context): n_fam consensus
 * sequences (length U[min_len,max_len], taken from the genome itself) are pasted until ~target_bases bases are covered;
 * every copy gets its own divergence U[div_lo,div_hi] (substitutions), so copies of one family differ by 2x that.
 * Sequential and deterministic. */
void mqsim_plant_families(uint8_t *g, uint64_t len, uint64_t seed, uint64_t n_fam, uint64_t target_bases, uint64_t min_len,
                          uint64_t max_len, double div_lo, double div_hi) {
    uint64_t s = seed ^ 0x5EEDFA1117ULL;
    if (len < 4 * max_len || n_fam == 0) return;
    uint64_t *fl = (uint64_t *)malloc(n_fam * sizeof(uint64_t));
    uint8_t **fs = (uint8_t **)malloc(n_fam * sizeof(uint8_t *));
    for (uint64_t f = 0; f < n_fam; f++) {
        fl[f] = min_len + splitmix64(&s) % (max_len - min_len + 1);
        const uint64_t src = splitmix64(&s) % (len - fl[f]);
        fs[f] = (uint8_t *)malloc(fl[f]);
        memcpy(fs[f], g + src, fl[f]);
    }
    uint64_t done = 0;
    while (done < target_bases) {
        const uint64_t f = splitmix64(&s) % n_fam;
        const uint64_t dst = splitmix64(&s) % (len - fl[f]);
        const double div = div_lo + (div_hi - div_lo) * u01(&s);
        /* geometric skipping: next substituted base */
        uint64_t t = 0;
        memcpy(g + dst, fs[f], fl[f]);
        if (div > 0) {
            const double lg = log(1.0 - div);
            for (;;) {
                double u = u01(&s);
                if (u < 1e-300) u = 1e-300;
                t += (uint64_t)(log(u) / lg);
                if (t >= fl[f]) break;
                g[dst + t] = (uint8_t)ACGT[splitmix64(&s) & 3];
                t++;
            }
        }
        done += fl[f];
    }
    for (uint64_t f = 0; f < n_fam; f++) free(fs[f]);
    free(fs);
    free(fl);
}

/* Satellite arrays (centromere-like): n_arrays arrays of length U[min_len,max_len]; each is a higher-order-repeat unit of
 * U[unit_lo,unit_hi] bases (taken from the genome at the array's start) repeated end to end, every copy with its own
 * substitutions at rate `div` -- so copies are (1 - 2 div) identical to each other, like alpha-satellite HOR arrays.  Reads from
 * inside an array have no unique k-min-mers.  Sequential and deterministic. */
void mqsim_plant_satellites(uint8_t *g, uint64_t len, uint64_t seed, uint64_t n_arrays, uint64_t min_len, uint64_t max_len,
                            uint64_t unit_lo, uint64_t unit_hi, double div) {
    uint64_t s = seed ^ 0x5A7E111735ULL;
    if (len < 4 * max_len || unit_hi < unit_lo || unit_lo < 2) return;
    uint8_t *unit = (uint8_t *)malloc(unit_hi);
    for (uint64_t i = 0; i < n_arrays; i++) {
        const uint64_t L = min_len + splitmix64(&s) % (max_len - min_len + 1);
        const uint64_t U = unit_lo + splitmix64(&s) % (unit_hi - unit_lo + 1);
        const uint64_t dst = splitmix64(&s) % (len - L);
        memcpy(unit, g + dst, U);
        const double lg = div > 0 ? log(1.0 - div) : 0;
        for (uint64_t at = 0; at < L; at += U) {
            const uint64_t n = at + U <= L ? U : L - at;
            memcpy(g + dst + at, unit, n);
            if (div > 0) {
                uint64_t t = 0;
                for (;;) {
                    double u = u01(&s);
                    if (u < 1e-300) u = 1e-300;
                    t += (uint64_t)(log(u) / lg);
                    if (t >= n) break;
                    g[dst + at + t] = (uint8_t)ACGT[splitmix64(&s) & 3];
                    t++;
                }
            }
        }
    }
    free(unit);
}

/* Runs of N (assembly gaps): n_runs runs of length U[min_len,max_len]. */
void mqsim_plant_n(uint8_t *g, uint64_t len, uint64_t seed, uint64_t n_runs, uint64_t min_len, uint64_t max_len) {
    uint64_t s = seed ^ 0x4E4E4E4EULL;
    if (len < 4 * max_len) return;
    for (uint64_t i = 0; i < n_runs; i++) {
        const uint64_t L = min_len + splitmix64(&s) % (max_len - min_len + 1);
        const uint64_t dst = splitmix64(&s) % (len - L);
        memset(g + dst, 'N', L);
    }
}

typedef struct {
    const uint8_t *genome;      /* concatenated contigs */
    const uint64_t *ctg_off;    /* n_ctg + 1 */
    uint32_t n_ctg;
    uint32_t n_reads;
    double len_mean, len_sd;
    uint64_t len_min, len_max;
    double err, f_sub, f_ins; /* f_del = 1 - f_sub - f_ins */
    uint64_t seed;
    /* outputs */
    uint8_t *bases;
    uint64_t *offsets;    /* n_reads + 1, filled by pass 1 (cumulative capacity) */
    uint64_t *read_len;   /* actual length per read (<= capacity) */
    uint32_t *t_ctg;
    uint64_t *t_start, *t_end; /* 0-based start, exclusive end on the contig */
    uint8_t *t_strand;    /* 0 '+', 1 '-' */
    volatile uint32_t *next;
    uint32_t r0;          /* the job's read i is read r0 + i of the seeded set (a slice of it: mqsim_reads_range) */
} read_job;

static inline double gauss(uint64_t *s) {
    double u1 = u01(s), u2 = u01(s);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static inline uint8_t comp(uint8_t b) {
    switch (b) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return b;
    }
}

/* template placement for read r (pure function of seed and r) */
static void place(const read_job *j, uint32_t r, uint32_t *ctg, uint64_t *start, uint64_t *tlen, uint8_t *strand, uint64_t *state) {
    uint64_t s = j->seed ^ (0x9E3779B97F4A7C15ULL * ((uint64_t)j->r0 + (uint64_t)r + 1));
    splitmix64(&s);
    double L = j->len_mean + j->len_sd * gauss(&s);
    if (L < (double)j->len_min) L = (double)j->len_min;
    if (L > (double)j->len_max) L = (double)j->len_max;
    uint64_t total = j->ctg_off[j->n_ctg];
    uint64_t tl = (uint64_t)L;
    for (;;) {
        uint64_t g = splitmix64(&s) % total;
        uint32_t c = 0;
        /* contigs are few: linear scan */
        while (c + 1 < j->n_ctg && j->ctg_off[c + 1] <= g) c++;
        uint64_t clen = j->ctg_off[c + 1] - j->ctg_off[c];
        if (clen == 0) continue;
        uint64_t t = tl < clen ? tl : clen;
        uint64_t st = g - j->ctg_off[c];
        if (st + t > clen) st = clen - t;
        *ctg = c;
        *start = st;
        *tlen = t;
        break;
    }
    *strand = (uint8_t)(splitmix64(&s) & 1);
    *state = s;
}

static void *read_worker(void *a) {
    read_job *j = (read_job *)a;
    for (;;) {
        uint32_t lo = __sync_fetch_and_add(j->next, 64);
        if (lo >= j->n_reads) break;
        uint32_t hi = lo + 64 < j->n_reads ? lo + 64 : j->n_reads;
        for (uint32_t r = lo; r < hi; r++) {
            uint32_t c;
            uint64_t st, tl, s;
            uint8_t strand;
            place(j, r, &c, &st, &tl, &strand, &s);
            const uint8_t *tpl = j->genome + j->ctg_off[c] + st;
            uint8_t *out = j->bases + j->offsets[r];
            uint64_t cap = j->offsets[r + 1] - j->offsets[r];
            uint64_t n = 0;
            for (uint64_t t = 0; t < tl && n < cap; t++) {
                uint8_t b = strand ? comp(tpl[tl - 1 - t]) : tpl[t];
                if (j->err > 0 && u01(&s) < j->err) {
                    double e = u01(&s);
                    if (e < j->f_sub) {
                        uint8_t nb;
                        do { nb = (uint8_t)ACGT[splitmix64(&s) & 3]; } while (nb == b);
                        out[n++] = nb;
                    } else if (e < j->f_sub + j->f_ins) {
                        out[n++] = (uint8_t)ACGT[splitmix64(&s) & 3];
                        if (n < cap) out[n++] = b;
                    } /* else deletion */
                } else {
                    out[n++] = b;
                }
            }
            j->read_len[r] = n;
            j->t_ctg[r] = c;
            j->t_start[r] = st;
            j->t_end[r] = st + tl;
            j->t_strand[r] = strand;
        }
    }
    return NULL;
}

/* Pass 1: capacities.  offsets[r+1]-offsets[r] = template length + slack for insertions. */
void mqsim_read_caps(const uint64_t *ctg_off, uint32_t n_ctg, uint32_t n_reads, double len_mean, double len_sd,
                     uint64_t len_min, uint64_t len_max, uint64_t seed, uint64_t *offsets) {
    read_job j;
    memset(&j, 0, sizeof(j));
    j.ctg_off = ctg_off;
    j.n_ctg = n_ctg;
    j.n_reads = n_reads;
    j.len_mean = len_mean;
    j.len_sd = len_sd;
    j.len_min = len_min;
    j.len_max = len_max;
    j.seed = seed;
    offsets[0] = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        uint32_t c;
        uint64_t st, tl, s;
        uint8_t strand;
        place(&j, r, &c, &st, &tl, &strand, &s);
        offsets[r + 1] = offsets[r] + tl + tl / 16 + 64;
    }
}

/* Pass 2: fill reads into `bases` at the capacities from pass 1; read_len gets actual lengths. */
void mqsim_reads(const uint8_t *genome, const uint64_t *ctg_off, uint32_t n_ctg, uint32_t n_reads, double len_mean,
                 double len_sd, uint64_t len_min, uint64_t len_max, double err, double f_sub, double f_ins, uint64_t seed,
                 int threads, uint8_t *bases, const uint64_t *offsets, uint64_t *read_len, uint32_t *t_ctg,
                 uint64_t *t_start, uint64_t *t_end, uint8_t *t_strand) {
    volatile uint32_t next = 0;
    read_job j = {genome, ctg_off, n_ctg, n_reads, len_mean, len_sd, len_min, len_max, err, f_sub, f_ins, seed,
                  bases, (uint64_t *)offsets, read_len, t_ctg, t_start, t_end, t_strand, &next, 0};
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, read_worker, &j);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* The same two passes for reads [r0, r0 + n_reads) of the seeded set (a read is a pure function of seed and its number): bench.py
 * synthesises a batch slice by slice straight into device memory instead of holding 39 GB of capacity layout + 37 GB of bases. */
void mqsim_read_caps_range(const uint64_t *ctg_off, uint32_t n_ctg, uint32_t r0, uint32_t n_reads, double len_mean, double len_sd,
                           uint64_t len_min, uint64_t len_max, uint64_t seed, uint64_t *offsets) {
    read_job j;
    memset(&j, 0, sizeof(j));
    j.ctg_off = ctg_off;
    j.n_ctg = n_ctg;
    j.n_reads = n_reads;
    j.len_mean = len_mean;
    j.len_sd = len_sd;
    j.len_min = len_min;
    j.len_max = len_max;
    j.seed = seed;
    j.r0 = r0;
    offsets[0] = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        uint32_t c;
        uint64_t st, tl, s;
        uint8_t strand;
        place(&j, r, &c, &st, &tl, &strand, &s);
        offsets[r + 1] = offsets[r] + tl + tl / 16 + 64;
    }
}
void mqsim_reads_range(const uint8_t *genome, const uint64_t *ctg_off, uint32_t n_ctg, uint32_t r0, uint32_t n_reads, double len_mean,
                       double len_sd, uint64_t len_min, uint64_t len_max, double err, double f_sub, double f_ins, uint64_t seed,
                       int threads, uint8_t *bases, const uint64_t *offsets, uint64_t *read_len, uint32_t *t_ctg,
                       uint64_t *t_start, uint64_t *t_end, uint8_t *t_strand) {
    volatile uint32_t next = 0;
    read_job j = {genome, ctg_off, n_ctg, n_reads, len_mean, len_sd, len_min, len_max, err, f_sub, f_ins, seed,
                  bases, (uint64_t *)offsets, read_len, t_ctg, t_start, t_end, t_strand, &next, r0};
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, read_worker, &j);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* compact reads from capacity layout to a dense layout: dst offsets = prefix sum of read_len */
void mqsim_compact(const uint8_t *src, const uint64_t *src_off, const uint64_t *read_len, uint32_t n_reads, uint8_t *dst,
                   uint64_t *dst_off) {
    dst_off[0] = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        memmove(dst + dst_off[r], src + src_off[r], read_len[r]);
        dst_off[r + 1] = dst_off[r] + read_len[r];
    }
}

/* ---- FASTA / FASTQ file writer (bench.py's end-to-end legs): read i becomes ">r<i>\n<bases>\n" or
 * "@r<i>\n<bases>\n+\n<quality 'I' x len>\n"; the file is laid out first (sizes are known), then written by `threads`
 * threads with pwrite, each its own range of reads.  Returns the file size, 0 on failure. */
#include <fcntl.h>
#include <stdio.h>
#include <unistd.h>
typedef struct {
    int fd;
    const uint8_t *bases;
    const uint64_t *off;
    const uint64_t *fpos;
    uint32_t lo, hi;
    int fastq;
    int ok;
} fx_job;
static int pwrite_all(int fd, const uint8_t *p, uint64_t n, uint64_t at) {
    while (n) {
        ssize_t w = pwrite(fd, p, n, (off_t)at);
        if (w <= 0) return 0;
        p += w;
        n -= (uint64_t)w;
        at += (uint64_t)w;
    }
    return 1;
}
static void *fx_worker(void *a) {
    fx_job *j = (fx_job *)a;
    const uint64_t cap = 8u << 20;
    uint8_t *buf = (uint8_t *)malloc(cap + 64);
    uint64_t fill = 0, at = j->fpos[j->lo];
    j->ok = buf != NULL;
    for (uint32_t r = j->lo; j->ok && r < j->hi; ++r) {
        const uint64_t L = j->off[r + 1] - j->off[r];
        char hdr[32];
        const int hl = snprintf(hdr, sizeof(hdr), "%cr%u\n", j->fastq ? '@' : '>', r);
        const uint64_t need = (uint64_t)hl + L + 1 + (j->fastq ? 2 + L + 1 : 0);
        if (fill + need > cap) {  /* flush; a record larger than the buffer is written piecewise below */
            j->ok = pwrite_all(j->fd, buf, fill, at);
            at += fill;
            fill = 0;
        }
        if (need > cap) {
            j->ok = j->ok && pwrite_all(j->fd, (const uint8_t *)hdr, (uint64_t)hl, at) && pwrite_all(j->fd, j->bases + j->off[r], L, at + hl) &&
                    pwrite_all(j->fd, (const uint8_t *)"\n", 1, at + hl + L);
            at += (uint64_t)hl + L + 1;
            if (j->fastq) {
                uint8_t *q = (uint8_t *)malloc(L + 3);
                j->ok = j->ok && q != NULL;
                if (q) {
                    q[0] = '+';
                    q[1] = '\n';
                    memset(q + 2, 'I', L);
                    q[2 + L] = '\n';
                    j->ok = j->ok && pwrite_all(j->fd, q, L + 3, at);
                    free(q);
                }
                at += L + 3;
            }
            continue;
        }
        memcpy(buf + fill, hdr, (size_t)hl);
        fill += (uint64_t)hl;
        memcpy(buf + fill, j->bases + j->off[r], L);
        fill += L;
        buf[fill++] = '\n';
        if (j->fastq) {
            buf[fill++] = '+';
            buf[fill++] = '\n';
            memset(buf + fill, 'I', L);
            fill += L;
            buf[fill++] = '\n';
        }
    }
    if (j->ok && fill) j->ok = pwrite_all(j->fd, buf, fill, at);
    free(buf);
    return NULL;
}
uint64_t mqsim_write_fastx(const char *path, const uint8_t *bases, const uint64_t *off, uint32_t n, int fastq, int threads) {
    uint64_t *fpos = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    if (!fpos) return 0;
    fpos[0] = 0;
    for (uint32_t r = 0; r < n; ++r) {
        char hdr[32];
        const int hl = snprintf(hdr, sizeof(hdr), "%cr%u\n", '>', r);
        const uint64_t L = off[r + 1] - off[r];
        fpos[r + 1] = fpos[r] + (uint64_t)hl + L + 1 + (fastq ? 2 + L + 1 : 0);
    }
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        free(fpos);
        return 0;
    }
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64];
    fx_job jobs[64];
    int ok = 1;
    for (int t = 0; t < threads; ++t) {
        jobs[t].fd = fd;
        jobs[t].bases = bases;
        jobs[t].off = off;
        jobs[t].fpos = fpos;
        jobs[t].lo = (uint32_t)((uint64_t)n * t / threads);
        jobs[t].hi = (uint32_t)((uint64_t)n * (t + 1) / threads);
        jobs[t].fastq = fastq;
        jobs[t].ok = 0;
        pthread_create(&th[t], NULL, fx_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) {
        pthread_join(th[t], NULL);
        ok = ok && jobs[t].ok;
    }
    const uint64_t size = fpos[n];
    free(fpos);
    ok = (close(fd) == 0) && ok;
    return ok ? size : 0;
}
