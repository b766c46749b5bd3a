// mq_host_state.hpp -- host-side state behind the C ABI (part of the one translation unit mq_capi.hip): error text, mq_ctx (a stream slot), mq_index,
// launch geometry, scratch management.
#pragma once

// =================================================================== host side

constexpr uint32_t MQ_MAX_REF_ID = 1u << 24;
static thread_local std::string g_err;
static int set_err(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) {                                                                                   \
            char _b[512];                                                                                         \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return set_err(_e == hipErrorOutOfMemory ? MQ_ENOMEM : MQ_EHIP, _b);                                  \
        }                                                                                                         \
    } while (0)

struct mq_index;

// One stream slot: everything a map launch sequence writes (work counters, Match scratch, minimizer lists,
// events) plus the staging buffers of the host-buffer entry points.  Launch sequences of DIFFERENT contexts of one index
// may be in flight together (the index itself is read-only once finalized); one context runs one sequence at a time.
struct mq_ctx {
    mq_index *idx = nullptr;
    hipStream_t stream = nullptr;   // the context's own stream (host-buffer entry points)
    uint32_t *d_counter = nullptr;  // 64 words: SplitArgs::counters; [8..11] two 64-bit probe statistics of an instrumented launch
    MatchRec *scratch = nullptr;    // per mapping wave: cap_matches records
    size_t scratch_waves = 0;       // waves it has windows for
    unsigned long long *mz_hash = nullptr;
    uint32_t *mz_pos = nullptr;
    uint32_t *mz_last = nullptr;    // seeding variant 16 only: every minimizer's second position (allocated with mz_pos, else nullptr)
    uint64_t mz_cap = 0;            // list entries allocated
    uint32_t *mz_count = nullptr;
    uint64_t *mz_base = nullptr;
    uint32_t *queue = nullptr;
    uint4 *work = nullptr;          // map_kernel's work items (order_reads_kernel): WORK_FRONT_CAP + reads_cap descriptors
    uint64_t reads_cap = 0;
    uint64_t pool_base = 0, pool_cap = 0;  // of the last ctx_ensure: the pool behind the regular list regions
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // staging for the host-buffer entry points
    uint8_t *st_bases = nullptr;
    uint64_t st_bases_cap = 0;
    uint64_t *st_off = nullptr;
    uint64_t st_off_cap = 0;
    mq_hit *st_out = nullptr;
    uint64_t st_out_cap = 0;
    uint32_t *st_lens = nullptr;
    uint64_t st_lens_cap = 0;
    uint64_t *h_off = nullptr;      // page-locked: relative offsets on their way to the device
    uint64_t h_off_cap = 0;
    mq_hit *h_out = nullptr;        // page-locked: hits on their way back
    uint64_t h_out_cap = 0;
    // device-parsed FASTA chunks (mq_ctx_submit_fasta / mq_ctx_wait_fasta): tile counts / offsets, the line ends, the scan's result words
    uint32_t *fx_tile_counts = nullptr, *fx_tile_off = nullptr;
    uint64_t fx_tile_counts_cap = 0, fx_tile_off_cap = 0;
    uint32_t *fx_nl = nullptr, *h_fx_nl = nullptr;
    uint64_t fx_nl_cap = 0, h_fx_nl_cap = 0;
    uint32_t *fx_info = nullptr, *h_fx_info = nullptr;  // device / page-locked: lines, records, flags
    uint8_t *h_fx_tail = nullptr;                       // page-locked, one page: the piece's bytes behind its last page boundary
    bool fx_pending = false;
    uint32_t fx_lpr = 2;                                // lines per record of the piece in flight (FASTA 2, FASTQ 4)
    const uint8_t *fx_buf = nullptr;
    uint32_t fx_begin = 0, fx_bytes = 0;
    // a submitted, not yet waited-for batch
    bool pending = false;
    const uint8_t *p_bases = nullptr;
    const uint64_t *p_offsets = nullptr;
    const uint32_t *p_lens = nullptr;
    uint32_t p_n = 0;
    mq_hit *p_out = nullptr;
};

struct KmmChunk {
    RefKmm *d = nullptr;
    uint64_t n = 0, cap = 0;  // k-min-mers of several references share a chunk (assemblies with 10^5 small contigs)
};

struct mq_index {
    mq_params params;
    DevParams dp;
    int device = 0;
    int n_cu = 0;
    std::once_flag geometry_once;  // launch geometry is worked out once, by whichever context or entry point maps first
    int geometry_rc = MQ_OK;
    std::mutex mu;  // serialises the index-level entry points (add_ref, finalize, and everything that uses the default context)
    std::map<uint32_t, std::pair<std::string, uint64_t>> refs;
    std::vector<KmmChunk> chunks;
    uint64_t n_kmm_total = 0;
    bool finalized = false;
    uint32_t table_factor = 0;  // mq_index_set_table_factor: slots per inserted k-min-mer (0: the default, 8)
    Bucket *table = nullptr;  // nslots / 2 buckets + the extra bucket of the key 0
    uint64_t nslots = 0;
    uint64_t *d_ref_lens = nullptr;
    uint64_t n_unique = 0, n_keys = 0;
    // grow-only scratch of mq_index_add_ref (freed by finalize): no allocation per reference once it has grown
    uint8_t *bld_seq = nullptr;
    uint64_t bld_seq_cap = 0;
    unsigned long long *bld_seg_hash = nullptr, *bld_dense_hash = nullptr;  // per-segment minimizer lists, the reference's dense list
    uint32_t *bld_seg_pos = nullptr, *bld_dense_pos = nullptr;
    uint32_t *bld_seg_last = nullptr, *bld_dense_last = nullptr;  // seeding variant 16 only
    uint64_t bld_seg_hash_cap = 0, bld_seg_pos_cap = 0, bld_dense_hash_cap = 0, bld_dense_pos_cap = 0, bld_seg_last_cap = 0, bld_dense_last_cap = 0;
    uint32_t *bld_counts = nullptr, *bld_queue = nullptr;
    uint64_t bld_counts_cap = 0, bld_queue_cap = 0;
    unsigned long long *bld_seg_off = nullptr;
    uint64_t bld_seg_off_cap = 0;
    unsigned long long *bld_info = nullptr;  // device: [0] total minimizers, [1] overflow flag, [2..3] seed_ref_kernel's work counters
    int grid_ref = 0;                        // workgroups of seed_ref_kernel that stay resident
    // launch geometry (workgroups) and scratch sizes, fixed at the first map call
    uint32_t grid_fused = 0, grid_seed = 0, grid_map = 0;  // map_kernel; seed_reads_kernel, map_lists_kernel (split)
    uint32_t cap_matches = 0;
    bool split = false;             // diagnostic MQ_PIPELINE=split: the two phases as separate launches (a profiler then prices each)
    bool force_general = false;     // test hook MQ_FORCE_GENERAL=1: never take the fast seeding path
    bool heavy_first = true;        // order_reads_kernel puts reads that look like short-period tandem arrays first (MQ_HEAVY_FIRST=0: A/B hook, reads in their own order)
    int chain_chunk = 64;           // test hook: MQ_CHAIN_CHUNK=4 exercises the multi-chunk chain path
    mq_ctx *def_ctx = nullptr;      // the context behind the index-level map entry points
    double t_add_ms = 0;            // MQ_BUILD_TIMING: wall time spent in mq_index_add_ref[_device] so far
    // mq_index_stage_*: the reference file's bytes on their way to the device piece by piece (a buffer of the file's size, an upload
    // stream, one event per piece); a state of its own behind its own lock, so that pieces keep flowing while a record is being indexed
    std::mutex stg_mu;
    uint8_t *stg_buf = nullptr;
    uint64_t stg_bytes = 0;
    hipStream_t stg_stream = nullptr;
    std::vector<hipEvent_t> stg_events;  // ticket t = event t (tickets count from 0)
    uint64_t stg_issued = 0;             // pieces issued so far
    // mq_index_reserve: the table allocated and cleared ahead of time by a thread of its own (finalize adopts it when the size fits)
    std::thread rsv_thread;
    Bucket *rsv_table = nullptr;
    uint64_t rsv_nslots = 0;
    int rsv_err = 0;                // hipError_t of the background allocation
    double table_alloc_ms = 0;      // what allocating + clearing the table that is in use took (hipMalloc + memset + synchronize), wherever it ran
    double rsv_ms = 0;              // the same for the reservation (becomes table_alloc_ms when finalize adopts the reserved table)
};

// slots of the table for n inserted k-min-mers: MQ_TABLE_FACTOR (default 8: load <= 0.125) times n, rounded up to a power of two
static uint64_t table_slots_for(const mq_index *idx, uint64_t n_kmm) {
    const char *lf = getenv("MQ_TABLE_FACTOR");  // (test / experiment hook: overrides the caller's choice)
    const uint64_t factor = lf && atoi(lf) >= 2 ? (uint64_t)atoi(lf) : idx->table_factor ? (uint64_t)idx->table_factor : 8ull;  // >= 2: a full table would make a miss walk forever
    uint64_t nslots = 1024;
    while (nslots < factor * n_kmm) nslots <<= 1;
    return nslots;
}
static void rsv_join(mq_index *idx) {
    if (idx->rsv_thread.joinable()) idx->rsv_thread.join();
}

extern "C" {

const char *mq_last_error(void) { return g_err.c_str(); }
int mq_abi_version(void) { return MQ_ABI_VERSION; }

int mq_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_err(MQ_ENODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return 0;
    }
    return n;
}

void mq_params_default(mq_params *p) {
    p->k = 5;
    p->l = 31;
    p->density = 0.01;
    p->use_hpc = 1;
    p->c = 4;
    p->s = 11;
    p->g = 2000;
    p->flags = 0;
}

}  // extern "C"

// (density as FH * H::MAX as FH) as H with Rust's saturating float->int cast, for the seeding variant's FH (f64; f32 with bit 2) and H
// (u64; u32 with bit 4)
static uint64_t density_bound(double density, uint32_t variant) {
    if (variant & MQ_SEEDVAR_HASH32) {
        if (variant & MQ_SEEDVAR_F32_BOUND) {
            const float f = (float)density * 4294967295.0f;
            if (!(f > 0.0f)) return 0;
            if (f >= 4294967296.0f) return 0xFFFFFFFFull;
            return (uint64_t)(uint32_t)f;
        }
        const double d = density * 4294967295.0;
        if (!(d > 0.0)) return 0;
        if (d >= 4294967296.0) return 0xFFFFFFFFull;
        return (uint64_t)(uint32_t)d;
    }
    if (variant & MQ_SEEDVAR_F32_BOUND) {
        const float f = (float)density * 18446744073709551615.0f;
        if (!(f > 0.0f)) return 0;
        if (f >= 18446744073709551616.0f) return UINT64_MAX;
        return (uint64_t)f;
    }
    const double d = density * 18446744073709551615.0;
    if (!(d > 0.0)) return 0;
    if (d >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)d;
}
// DevParams of an index: the bound in the form the kernels compare with (`hash <= bound` on 64-bit words; DevParams::variant)
static void set_dev_bound(DevParams &dp, double density, uint32_t variant) {
    uint64_t b = density_bound(density, variant);
    dp.keep_none = 0;
    if (variant & MQ_SEEDVAR_STRICT_BOUND) {  // hash < b  <=>  hash <= b - 1; nothing is below 0
        if (b == 0) dp.keep_none = 1;
        else b -= 1;
    }
    if (variant & MQ_SEEDVAR_HASH32) b = (b & 0xFFFFFFFFull) | (b << 32);  // dup(bound32): compared with dup(hash32)
    dp.bound = b;
    dp.variant = variant;
}

static int use_device(const mq_index *idx) {
    HIPCHK(hipSetDevice(idx->device));
    return MQ_OK;
}

static size_t table_bytes_of(uint64_t nslots) { return (size_t)(nslots / 2 + 1) * sizeof(Bucket); }

static int alloc_table(mq_index *idx, uint64_t nslots) {
    if (idx->table) {
        HIPCHK(hipFree(idx->table));
        idx->table = nullptr;
    }
    if (nslots < 2) nslots = 2;  // whole buckets
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipMalloc((void **)&idx->table, table_bytes_of(nslots)));
    HIPCHK(hipMemset(idx->table, 0, table_bytes_of(nslots)));
    HIPCHK(hipDeviceSynchronize());
    idx->table_alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    idx->nslots = nslots;
    return MQ_OK;
}

template <class T>
static int grow(T *&p, uint64_t &cap, uint64_t need) {
    if (need <= cap) return MQ_OK;
    if (p) HIPCHK(hipFree(p));
    p = nullptr;
    cap = 0;
    uint64_t nc = need + need / 4 + 64;
    HIPCHK(hipMalloc((void **)&p, nc * sizeof(T)));
    cap = nc;
    return MQ_OK;
}
template <class T>
static int grow_pinned(T *&p, uint64_t &cap, uint64_t need) {
    if (need <= cap) return MQ_OK;
    if (p) HIPCHK(hipHostFree(p));
    p = nullptr;
    cap = 0;
    uint64_t nc = need + need / 4 + 64;
    HIPCHK(hipHostMalloc((void **)&p, nc * sizeof(T), hipHostMallocDefault));
    cap = nc;
    return MQ_OK;
}

// launch geometry: persistent waves, as many workgroups as stay resident
static int ensure_geometry_once(mq_index *idx) {
    auto occ_of = [&](const void *fn, int threads, int &occ) -> int {
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, 0));
        if (occ < 1) occ = 1;
        if (occ > 8) occ = 8;
        const char *oe = getenv("MQ_OCC");  // diagnostic: cap workgroups per CU
        if (oe && atoi(oe) >= 1 && atoi(oe) < occ) occ = atoi(oe);
        return MQ_OK;
    };
    int occ = 0, rc;
    // the grid of a launch sequence = the workgroups that are RESIDENT together, for every instantiation launched on it: wave w's first two
    // work items are its own (items w and n_waves + w, among them the heavy reads that go first), so a workgroup that has to wait for a
    // place would hold them back; the variants' instantiation and map_declined_kernel (more scratch) may fit fewer than map_kernel<64, false>
    {
        const void *fns[] = {(const void *)map_kernel<64, false, false>, (const void *)map_kernel<64, false, true>, (const void *)map_kernel<64, true, false>,
                             (const void *)map_kernel<64, true, true>, (const void *)map_kernel<4, false, false>, (const void *)map_kernel<4, false, true>,
                             (const void *)map_declined_kernel<64, false, false>, (const void *)map_declined_kernel<64, false, true>,
                             (const void *)map_declined_kernel<64, true, false>, (const void *)map_declined_kernel<64, true, true>,
                             (const void *)map_declined_kernel<4, false, false>, (const void *)map_declined_kernel<4, false, true>};
        int occ_min = 8;
        for (const void *fn : fns) {
            if ((rc = occ_of(fn, 64 * MAP_WAVES, occ))) return rc;
            occ_min = std::min(occ_min, occ);
        }
        occ = occ_min;
    }
    idx->grid_fused = (uint32_t)(occ * idx->n_cu);
    if ((rc = occ_of((const void *)seed_reads_kernel<0>, 64 * SEED_WAVES, occ))) return rc;
    idx->grid_seed = (uint32_t)(occ * idx->n_cu);
    if ((rc = occ_of((const void *)map_lists_kernel<64, false>, 64 * ML_WAVES, occ))) return rc;
    idx->grid_map = (uint32_t)(occ * idx->n_cu);
    // Match runs per read held in HBM scratch; a read with more runs is reported MQ_HIT_OVERFLOW (never silently wrong)
    const char *e = getenv("MQ_MATCH_CAP");
    idx->cap_matches = e ? (uint32_t)strtoul(e, nullptr, 10) : 2048u;
    if (idx->cap_matches < 1) idx->cap_matches = 1;
    return MQ_OK;
}

static int ensure_geometry(mq_index *idx) {
    std::call_once(idx->geometry_once, [idx] { idx->geometry_rc = ensure_geometry_once(idx); });  // contexts of one index start concurrently
    return idx->geometry_rc;
}

// list entries reserved per base, in 1/65536: 4 d + 1/512 -- canonical selection keeps 1-(1-d)^2 ~ 2 d of the l-mers, so this
// is at least twice the expected count (2.6 times under homopolymer compression); denser lists take the overflow redo
static uint32_t list_f16(const mq_index *idx) {
    double d = idx->params.density;
    if (!(d > 0)) d = 0;
    double f = 4.0 * d + 1.0 / 512.0;
    if (f > 1.0) f = 1.0;
    const char *e = getenv("MQ_LIST_F16");  // test hook: force list-region overflows
    if (e && atoi(e) >= 0) return (uint32_t)std::min(65536, atoi(e));
    return (uint32_t)std::ceil(f * 65536.0);
}
constexpr uint32_t LIST_SLACK = 64;

static int ctx_ensure(mq_ctx *c, uint32_t n, uint64_t total_bases, uint32_t f16) {
    mq_index *idx = c->idx;
    int rc = ensure_geometry(idx);
    if (rc) return rc;
    if (!c->d_counter) HIPCHK(hipMalloc((void **)&c->d_counter, 256));
    if (!c->ev0) {
        HIPCHK(hipEventCreate(&c->ev0));
        HIPCHK(hipEventCreate(&c->ev1));
    }
    {
        // Match scratch: one window per mapping wave THIS batch can employ (a launch never has more workgroups than reads / waves per
        // workgroup): a context that only ever sees chunks of a thousand reads does not pay for 4,096 waves' windows (268 MB of fresh
        // device memory, ~8 ms, per context)
        const size_t max_waves = std::max((size_t)idx->grid_fused * MAP_WAVES, (size_t)idx->grid_map * ML_WAVES);
        const size_t wpw = (size_t)std::max(MAP_WAVES, ML_WAVES);
        const size_t want = std::min(max_waves, ((size_t)n + wpw - 1) / wpw * wpw + wpw);
        if (want > c->scratch_waves) {
            if (c->scratch) HIPCHK(hipFree(c->scratch));
            c->scratch = nullptr;
            c->scratch_waves = 0;
            const size_t nw = std::min(max_waves, want + want / 4);
            HIPCHK(hipMalloc((void **)&c->scratch, nw * idx->cap_matches * sizeof(MatchRec)));
            c->scratch_waves = nw;
        }
    }
    if (n > c->reads_cap) {
        if (c->mz_count) HIPCHK(hipFree(c->mz_count));
        if (c->mz_base) HIPCHK(hipFree(c->mz_base));
        if (c->queue) HIPCHK(hipFree(c->queue));
        if (c->work) HIPCHK(hipFree(c->work));
        c->mz_count = c->queue = nullptr;
        c->mz_base = nullptr;
        c->work = nullptr;
        c->reads_cap = 0;
        const uint64_t nc = (uint64_t)n + n / 4 + 64;
        HIPCHK(hipMalloc((void **)&c->mz_count, nc * 4));
        HIPCHK(hipMalloc((void **)&c->mz_base, nc * 8));
        HIPCHK(hipMalloc((void **)&c->queue, nc * 4));
        HIPCHK(hipMalloc((void **)&c->work, (nc + WORK_FRONT_CAP) * sizeof(uint4)));
        c->reads_cap = nc;
    }
    // regular regions, then the pool for lists denser than their region (an eighth of the regular space, at least 1 M entries)
    const uint64_t regular = ((total_bases * f16) >> 16) + (uint64_t)LIST_SLACK * n + 64;
    const uint64_t pool = f16 >= 65536u ? 0 : std::max<uint64_t>(regular / 8, 1ull << 20);
    const uint64_t need = regular + pool;
    c->pool_base = regular;
    c->pool_cap = pool;
    if (need > c->mz_cap) {
        if (c->mz_hash) HIPCHK(hipFree(c->mz_hash));
        if (c->mz_pos) HIPCHK(hipFree(c->mz_pos));
        if (c->mz_last) HIPCHK(hipFree(c->mz_last));
        c->mz_hash = nullptr;
        c->mz_pos = nullptr;
        c->mz_last = nullptr;
        c->mz_cap = 0;
        const uint64_t nc = need + need / 8;
        HIPCHK(hipMalloc((void **)&c->mz_hash, nc * 8));
        HIPCHK(hipMalloc((void **)&c->mz_pos, nc * 4));
        if (idx->dp.variant & MQ_SEEDVAR_END_COMPRESSED) HIPCHK(hipMalloc((void **)&c->mz_last, nc * 4));
        c->mz_cap = nc;
    }
    return MQ_OK;
}

static void ctx_release(mq_ctx *c) {
    if (!c) return;
    if (c->stream) hipStreamSynchronize(c->stream);
    hipFree(c->d_counter);
    hipFree(c->scratch);
    hipFree(c->mz_hash);
    hipFree(c->mz_pos);
    hipFree(c->mz_last);
    hipFree(c->mz_count);
    hipFree(c->mz_base);
    hipFree(c->queue);
    hipFree(c->work);
    hipFree(c->st_bases);
    hipFree(c->st_off);
    hipFree(c->st_out);
    hipFree(c->st_lens);
    hipFree(c->fx_tile_counts);
    hipFree(c->fx_tile_off);
    hipFree(c->fx_nl);
    hipFree(c->fx_info);
    if (c->h_fx_nl) hipHostFree(c->h_fx_nl);
    if (c->h_fx_info) hipHostFree(c->h_fx_info);
    if (c->h_fx_tail) hipHostFree(c->h_fx_tail);
    if (c->h_off) hipHostFree(c->h_off);
    if (c->h_out) hipHostFree(c->h_out);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

static mq_ctx *ctx_create(mq_index *idx) {
    mq_ctx *c = new (std::nothrow) mq_ctx();
    if (!c) {
        set_err(MQ_ENOMEM, "out of host memory");
        return nullptr;
    }
    c->idx = idx;
    if (hipSetDevice(idx->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        set_err(MQ_EHIP, "hipStreamCreate failed");
        c->stream = nullptr;
        ctx_release(c);
        return nullptr;
    }
    return c;
}

static void free_stage(mq_index *idx) {
    std::lock_guard<std::mutex> lk(idx->stg_mu);
    if (idx->stg_stream) hipStreamSynchronize(idx->stg_stream);
    for (hipEvent_t e : idx->stg_events) hipEventDestroy(e);
    idx->stg_events.clear();
    if (idx->stg_stream) hipStreamDestroy(idx->stg_stream);
    idx->stg_stream = nullptr;
    hipFree(idx->stg_buf);
    idx->stg_buf = nullptr;
    idx->stg_bytes = 0;
    idx->stg_issued = 0;
}

static void free_build_scratch(mq_index *idx) {
    free_stage(idx);
    hipFree(idx->bld_seq);
    hipFree(idx->bld_seg_hash);
    hipFree(idx->bld_seg_pos);
    hipFree(idx->bld_dense_hash);
    hipFree(idx->bld_dense_pos);
    hipFree(idx->bld_seg_last);
    hipFree(idx->bld_dense_last);
    hipFree(idx->bld_counts);
    hipFree(idx->bld_queue);
    hipFree(idx->bld_seg_off);
    hipFree(idx->bld_info);
    idx->bld_seq = nullptr;
    idx->bld_seg_hash = idx->bld_dense_hash = nullptr;
    idx->bld_seg_pos = idx->bld_dense_pos = nullptr;
    idx->bld_seg_last = idx->bld_dense_last = nullptr;
    idx->bld_seg_last_cap = idx->bld_dense_last_cap = 0;
    idx->bld_counts = idx->bld_queue = nullptr;
    idx->bld_seg_off = nullptr;
    idx->bld_info = nullptr;
    idx->bld_seq_cap = idx->bld_seg_hash_cap = idx->bld_seg_pos_cap = idx->bld_dense_hash_cap = idx->bld_dense_pos_cap = 0;
    idx->bld_counts_cap = idx->bld_queue_cap = idx->bld_seg_off_cap = 0;
}
