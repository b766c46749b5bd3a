// stub_mapquik_hip.cc -- a HOST-ONLY stand-in for libmapquik_hip.so with canned results, for the sanitizer builds of the native
// driver (make asan / make tsan): ASan, UBSan and TSan cannot run on the GPU build, and what they are wanted for is the driver's
// threads (feeder readers, inflate thread, per-GPU submitters, formatter pool, ordered writer), not the kernels.
// Every read "maps" to reference 0 with coordinates derived from its length and first bases; submit hands the batch to a worker
// thread per context so that mq_ctx_wait really waits.  Test infrastructure only: never linked into the product.
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mapquik_hip.h"

struct mq_index {
    mq_params p;
    int device;
    std::map<uint32_t, std::pair<std::string, uint64_t>> refs;
    bool finalized = false;
    uint64_t n_kmm = 0;
    // mq_index_stage_*: the reference file in pieces
    std::vector<uint8_t> stage;
    bool stage_begun = false;
    uint64_t stage_tickets = 0;
    std::mutex stage_mu;
};
struct mq_ctx {
    mq_index *idx;
    std::thread worker;
    bool pending = false;
    // mq_ctx_submit_fasta / mq_ctx_wait_fasta: line ends and hits of the chunk in flight
    std::vector<uint32_t> fx_lines;
    std::vector<mq_hit> fx_hits;
    uint32_t fx_flags = 0;
};
static thread_local std::string g_err;

static void canned(const mq_index *idx, const uint8_t *buf, uint64_t start, uint32_t len, mq_hit *h) {
    memset(h, 0, sizeof(*h));
    uint32_t x = len;
    for (uint32_t i = 0; i < len && i < 8; ++i) x = x * 131u + buf[start + i];
    if (len < 50 || idx->refs.empty()) {
        h->status = MQ_HIT_UNMAPPED;
        return;
    }
    const uint64_t rl = idx->refs.begin()->second.second;
    h->status = MQ_HIT_MAPPED;
    h->ref_id = idx->refs.begin()->first;
    h->rc = x & 1u;
    h->mapq = 60;
    h->q_start = 0;
    h->q_end = len - 1;
    h->r_start = (uint32_t)(rl > len ? x % (rl - len) : 0);
    h->r_end = h->r_start + len - 1;
    h->score = len / 100 + 1;
    h->n_kminmers = len / 80;
}

extern "C" {
const char *mq_last_error(void) { return g_err.c_str(); }
int mq_abi_version(void) { return MQ_ABI_VERSION; }
int mq_device_count(void) {
    const char *e = getenv("MQ_STUB_DEVICES");
    return e ? atoi(e) : 1;
}
void mq_params_default(mq_params *p) {
    p->k = 5; p->l = 31; p->density = 0.01; p->use_hpc = 1; p->c = 4; p->s = 11; p->g = 2000; p->flags = 0;
}
mq_index *mq_index_new(const mq_params *p, int device) {
    mq_index *i = new mq_index();
    i->p = *p;
    i->device = device;
    return i;
}
void mq_index_free(mq_index *i) { delete i; }
int64_t mq_index_add_ref(mq_index *i, uint32_t id, const char *name, const uint8_t *seq, uint64_t len) {
    if (!i || (!seq && len)) { g_err = "bad arguments"; return MQ_EINVAL; }
    uint64_t sum = 0;
    for (uint64_t j = 0; j < len; j += 4096) sum += seq[j];  // touch the caller's buffer (use-after-free shows under ASan)
    i->refs[id] = std::make_pair(std::string(name ? name : ""), len);
    i->n_kmm += len / 100 + (sum & 1);
    return (int64_t)(len / 100);
}
int mq_index_reserve(mq_index *, uint64_t) { return MQ_OK; }
int mq_index_set_table_factor(mq_index *, uint32_t f) { return f >= 2 && f <= 64 ? MQ_OK : MQ_EINVAL; }
// the reference file in pieces: a host buffer stands in for the device's; a piece is copied at once (its ticket is done when it returns)
int mq_index_stage_begin(mq_index *i, uint64_t total) {
    if (!i->stage.empty()) { g_err = "one staging buffer per index"; return MQ_ESTATE; }
    i->stage.assign((size_t)total + 1, 0);
    i->stage_begun = true;
    return MQ_OK;
}
int mq_index_stage_piece(mq_index *i, uint64_t at, const uint8_t *src, uint64_t n, uint64_t *ticket) {
    std::lock_guard<std::mutex> lk(i->stage_mu);
    if (!i->stage_begun || at + n + 1 > i->stage.size()) { g_err = "piece outside the staging buffer"; return MQ_EINVAL; }
    memcpy(i->stage.data() + at, src, (size_t)n);
    *ticket = i->stage_tickets++;
    return MQ_OK;
}
int mq_index_stage_done(mq_index *i, uint64_t ticket, int) {
    std::lock_guard<std::mutex> lk(i->stage_mu);
    if (ticket >= i->stage_tickets) { g_err = "unknown ticket"; return MQ_EINVAL; }
    return 1;
}
int64_t mq_index_add_ref_staged(mq_index *i, uint32_t id, const char *name, uint64_t at, uint64_t len, uint64_t after) {
    std::lock_guard<std::mutex> lk(i->stage_mu);
    if (after != MQ_STAGE_ALL_ISSUED && after >= i->stage_tickets) { g_err = "unknown ticket"; return MQ_EINVAL; }
    if (!i->stage_begun || at + len + 1 > i->stage.size()) { g_err = "record outside the staging buffer"; return MQ_EINVAL; }
    return mq_index_add_ref(i, id, name, i->stage.data() + at, len);
}
int64_t mq_index_finalize(mq_index *i) { i->finalized = true; return (int64_t)i->n_kmm; }
mq_index *mq_index_clone(const mq_index *s, int device) {
    mq_index *i = new mq_index();
    i->p = s->p;
    i->refs = s->refs;
    i->finalized = s->finalized;
    i->n_kmm = s->n_kmm;
    i->device = device;
    return i;
}
int mq_index_get_stats(const mq_index *i, mq_index_stats *o) {
    memset(o, 0, sizeof(*o));
    o->n_refs = i->refs.size();
    o->n_unique = i->n_kmm;
    return MQ_OK;
}
int mq_index_ref_info(const mq_index *i, uint32_t ref_id, const char **name, uint64_t *len) {
    auto it = i->refs.find(ref_id);
    if (it == i->refs.end()) { g_err = "unknown ref_id"; return MQ_EINVAL; }
    if (name) *name = it->second.first.c_str();
    if (len) *len = it->second.second;
    return MQ_OK;
}
int mq_index_get_params(const mq_index *i, mq_params *o) { *o = i->p; return MQ_OK; }
int mq_index_set_map_params(mq_index *i, uint32_t c, uint32_t s, uint32_t g, int fold) {
    i->p.c = c; i->p.s = s; i->p.g = g;
    i->p.flags = (i->p.flags & ~MQ_FLAG_FOLD_CASE) | (fold ? MQ_FLAG_FOLD_CASE : 0u);
    return MQ_OK;
}
int mq_index_save(const mq_index *, const char *) { g_err = "stub"; return MQ_EINVAL; }
mq_index *mq_index_load(const char *, int) { g_err = "stub"; return nullptr; }
mq_ctx *mq_ctx_new(mq_index *i) {
    mq_ctx *c = new mq_ctx();
    c->idx = i;
    return c;
}
void mq_ctx_free(mq_ctx *c) {
    if (!c) return;
    if (c->worker.joinable()) c->worker.join();
    delete c;
}
int mq_ctx_reserve(mq_ctx *, uint32_t, uint64_t) { return MQ_OK; }
int mq_ctx_submit_spans(mq_ctx *c, const uint8_t *buf, uint64_t bytes, const uint64_t *starts, const uint32_t *lens, uint32_t n, mq_hit *out) {
    if (c->pending) { g_err = "context has a submitted batch"; return MQ_ESTATE; }
    for (uint32_t i = 0; i < n; ++i)
        if (starts[i] + lens[i] > bytes) { g_err = "span outside the buffer"; return MQ_EINVAL; }
    c->pending = true;
    c->worker = std::thread([=]() {
        for (uint32_t i = 0; i < n; ++i) canned(c->idx, buf, starts[i], lens[i], &out[i]);
    });
    return MQ_OK;
}
// the device's record scan, on the host: line ends of buf[begin, bytes); FASTA: two lines per record, '>' in front of every header;
// FASTQ: four lines, '@' / '+' / one quality per base
int mq_ctx_submit_fastx(mq_ctx *c, const uint8_t *buf, uint64_t begin, uint64_t bytes, uint32_t format) {
    if (c->pending) { g_err = "context has a submitted batch"; return MQ_ESTATE; }
    if (format > MQ_FASTX_FASTQ) { g_err = "format"; return MQ_EINVAL; }
    c->pending = true;
    c->worker = std::thread([=]() {
        const size_t lpr = format == MQ_FASTX_FASTQ ? 4 : 2;
        c->fx_lines.clear();
        c->fx_hits.clear();
        c->fx_flags = 0;
        for (uint64_t p = begin; p < bytes; ++p)
            if (buf[p] == '\n') c->fx_lines.push_back((uint32_t)p);
        if (bytes > begin && buf[bytes - 1] != '\n') c->fx_lines.push_back((uint32_t)bytes);
        if (c->fx_lines.size() % lpr) c->fx_flags = MQ_FASTA_IRREGULAR;
        const size_t n = c->fx_lines.size() / lpr;
        auto cut = [&](uint64_t s, uint64_t e) { return (e > s && buf[e - 1] == '\r') ? e - 1 : e; };
        for (size_t i = 0; i < n && !c->fx_flags; ++i) {
            const uint64_t hs = i ? (uint64_t)c->fx_lines[lpr * i - 1] + 1 : begin, he = c->fx_lines[lpr * i], ss = he + 1, se = c->fx_lines[lpr * i + 1];
            if (format == MQ_FASTX_FASTQ) {
                const uint64_t ps = se + 1, pe = c->fx_lines[4 * i + 2], qs = pe + 1, qe = c->fx_lines[4 * i + 3];
                if (hs >= he || buf[hs] != '@' || ps >= pe || buf[ps] != '+' || cut(ss, se) - ss != cut(qs, qe) - qs) c->fx_flags = MQ_FASTA_IRREGULAR;
            } else if (hs >= he || buf[hs] != '>' || (ss < se && buf[ss] == '>')) {
                c->fx_flags = MQ_FASTA_IRREGULAR;
            }
        }
        if (c->fx_flags) return;
        c->fx_hits.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const uint64_t ss = (uint64_t)c->fx_lines[lpr * i] + 1;
            const uint64_t e = cut(ss, c->fx_lines[lpr * i + 1]);
            canned(c->idx, buf, ss, (uint32_t)(e - ss), &c->fx_hits[i]);
        }
    });
    return MQ_OK;
}
int mq_ctx_submit_fasta(mq_ctx *c, const uint8_t *buf, uint64_t begin, uint64_t bytes) { return mq_ctx_submit_fastx(c, buf, begin, bytes, MQ_FASTX_FASTA); }
int mq_ctx_wait_fasta(mq_ctx *c, uint32_t *n_reads, const uint32_t **line_ends, uint32_t *n_lines, const mq_hit **hits, uint32_t *flags) {
    if (c->worker.joinable()) c->worker.join();
    c->pending = false;
    *flags = c->fx_flags;
    *n_reads = c->fx_flags ? 0 : (uint32_t)c->fx_hits.size();
    *n_lines = c->fx_flags ? 0 : (uint32_t)c->fx_lines.size();
    *line_ends = c->fx_flags ? nullptr : c->fx_lines.data();
    *hits = c->fx_flags ? nullptr : c->fx_hits.data();
    return MQ_OK;
}
int mq_ctx_wait(mq_ctx *c) {
    if (c->worker.joinable()) c->worker.join();
    c->pending = false;
    return MQ_OK;
}
int mq_format_paf(const mq_index *i, const char *q_id, uint64_t q_len, const mq_hit *h, char *buf, size_t cap) {
    auto it = i->refs.find(h->ref_id);
    if (it == i->refs.end()) { g_err = "unknown ref"; return MQ_EINVAL; }
    const unsigned long long rl = it->second.second;
    return snprintf(buf, cap, "%s\t%llu\t%llu\t%llu\t%s\t%s\t%llu\t%u\t%u\t%u\t%llu\t%u", q_id, (unsigned long long)q_len,
                    ((unsigned long long)h->q_start_hi << 32) | h->q_start, ((unsigned long long)h->q_end_hi << 32) | h->q_end,
                    h->rc ? "-" : "+", it->second.first.c_str(), rl, h->r_start, h->r_end, h->score, rl, h->mapq);
}
void *mq_host_alloc(size_t n) { return malloc(n ? n : 1); }
void mq_host_free(void *p) { free(p); }
int mq_host_register(void *, size_t) { return MQ_OK; }
int mq_host_unregister(void *) { return MQ_OK; }
}
