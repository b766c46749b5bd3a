// mq_capi.hip -- kernels + the extern "C" boundary declared in include/mapquik_hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see mapquik_amd/build.py).  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mq_device.hpp"
#include "mq_fast.hpp"

using namespace mq;

// =================================================================== kernels

// Fused hot path: one wave per read; 4 waves per workgroup share the look-up tables; persistent waves pull read indices
// from an atomic counter.  CH: lanes per chunk in the chain stage (64 in production).  FAST=false forces the general path.
struct MapArgs {
    const uint8_t *bases;
    const uint64_t *offsets;
    uint32_t n;
    DevParams P;
    const Slot *table;
    uint64_t mask;
    const uint64_t *ref_lens;
    MatchRec *scratch_all;   // per wave: cap_matches records
    uint32_t cap_matches;
    uint8_t *fast_scratch;   // per wave: FAST_EM_BYTES + FAST_HM_WORDS*4
    uint32_t *work_counter;
    mq_hit *out;
    mq_kminmer *dump;
    const uint64_t *dump_off;
    uint32_t *dump_counts;
    uint32_t *stats;         // [0] reads through the fast path, [1] through the general path
    uint32_t stop_after;     // diagnostic (MQ_STOP_AFTER): 0 = run everything; 1/2/3 = stop a read after stage A / B / gather
};

constexpr int MAP_WAVES = 4;

#ifndef MQ_MIN_WAVES
#define MQ_MIN_WAVES 4
#endif
template <int CH, bool FAST, bool TIMING = false>
__global__ __launch_bounds__(64 * MAP_WAVES, MQ_MIN_WAVES) void map_kernel(const MapArgs A) {
    __shared__ WgTables T;
    __shared__ WaveLds SS[MAP_WAVES];
    build_tables(T, A.P.l);
    __syncthreads();  // the only workgroup-wide rendezvous; waves are independent from here on
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    WaveLds &S = SS[wv];
    const size_t wave_gid = (size_t)blockIdx.x * MAP_WAVES + wv;
    MatchRec *scratch = A.scratch_all + wave_gid * A.cap_matches;
    uint8_t *fs = A.fast_scratch + wave_gid * (size_t)(FAST_EM_BYTES + FAST_HM_WORDS * 4u);
    uint4 *em = reinterpret_cast<uint4 *>(fs);
    uint32_t *hm = reinterpret_cast<uint32_t *>(fs + FAST_EM_BYTES);
    const DevParams &P = A.P;
    uint32_t n_fast = 0, n_general = 0;
    unsigned long long tacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // diagnostic build only: cycles in A, B, C(+consume), finish, chain, total; extra probe steps; lookups
    const unsigned long long t_begin = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long r_begin = TIMING ? __builtin_amdgcn_s_memrealtime() : 0ull;  // 100 MHz
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(A.work_counter, 1u);
        r = rdfirst(r);
        if (r >= A.n) break;
        const uint64_t o0 = A.offsets[r], o1 = A.offsets[r + 1];
        const uint64_t len = o1 - o0;
        mq_hit h;
        h.status = MQ_HIT_UNMAPPED;
        h.ref_id = h.rc = h.mapq = h.q_start = h.q_end = h.r_start = h.r_end = h.score = h.n_kminmers = 0;
        uint32_t n_kmm = 0;
        // extract(): len < l + k - 1 => None (src/mers.rs:44)
        if (len >= (uint64_t)P.l + P.k - 1u) {
            mq_kminmer *d = nullptr;
            uint32_t dcap = 0;
            if (A.dump) {
                d = A.dump + A.dump_off[r];
                dcap = (uint32_t)(A.dump_off[r + 1] - A.dump_off[r]);
            }
            MapSink sink(A.table, A.mask, P, scratch, A.cap_matches, d, dcap);
            uint32_t mz_count = 0;
            bool done = false;
            if (FAST) done = fast_seed_sequence<MapSink, TIMING>(A.bases + o0, (uint32_t)len, P, T, S, sink, mz_count, em, hm, tacc, A.stop_after);
            if (done) n_fast++;
            else {
                n_general++;
                seed_segment(A.bases + o0, len, 0, len, P, S, sink, mz_count);
            }
            const unsigned long long t_f0 = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
            sink.finish(S, mz_count);
            n_kmm = sink.kmm_count;
            const unsigned long long t_f1 = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
            tacc[3] += t_f1 - t_f0;
            if (sink.n_matches > A.cap_matches) {
                h.status = MQ_HIT_OVERFLOW;
            } else if (sink.n_matches > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Match records written by this wave are in L2
                wave_sync();
                chain_stage<CH>(scratch, sink.n_matches, P, len, A.ref_lens, h);
            }
            if (TIMING) {
                tacc[4] += __builtin_amdgcn_s_memtime() - t_f1;
                tacc[6] += wave_sum_u32(sink.probe_steps);
                tacc[7] += n_kmm;
            }
        }
        h.n_kminmers = n_kmm;
        if (lane == 0) {
            A.out[r] = h;
            if (A.dump_counts) A.dump_counts[r] = n_kmm;
        }
        wave_sync();
    }
    if (A.stats && lane == 0) {
        if (n_fast) atomicAdd(&A.stats[0], n_fast);
        if (n_general) atomicAdd(&A.stats[1], n_general);
        if (TIMING) {
            tacc[5] = __builtin_amdgcn_s_memtime() - t_begin;
            tacc[8] = __builtin_amdgcn_s_memrealtime() - r_begin;
            unsigned long long *ts = reinterpret_cast<unsigned long long *>(A.stats + 2);
            for (int i = 0; i < 9; ++i) atomicAdd(&ts[i], tacc[i]);
        }
    }
}

// Reference path, stage 1: ordered minimizers of each fixed-size raw segment of one long sequence.
__global__ __launch_bounds__(64) void seed_segments_kernel(const uint8_t *__restrict__ seq, uint64_t len, uint64_t seg_len,
                                                           uint32_t n_seg, DevParams P, Minimizer *__restrict__ mz_out,
                                                           uint32_t cap, uint32_t *__restrict__ counts) {
    __shared__ WaveLds S;
    for (uint32_t s = blockIdx.x; s < n_seg; s += gridDim.x) {
        const uint64_t a = (uint64_t)s * seg_len;
        const uint64_t b = a + seg_len < len ? a + seg_len : len;
        ListSink sink(mz_out + (size_t)s * cap, cap);
        uint32_t mz_count = 0;
        seed_segment(seq, len, a, b, P, S, sink, mz_count);
        if (lane_id() == 0) counts[s] = sink.written;
        wave_sync();
    }
}

// stage 2: segment lists -> one dense ordered list
__global__ void compact_minimizers_kernel(const Minimizer *__restrict__ seg_lists, uint32_t cap, const uint32_t *__restrict__ counts,
                                          const uint64_t *__restrict__ seg_off, uint32_t n_seg, Minimizer *__restrict__ dense) {
    for (uint32_t s = blockIdx.x; s < n_seg; s += gridDim.x) {
        const uint32_t c = counts[s];
        const Minimizer *src = seg_lists + (size_t)s * cap;
        Minimizer *dst = dense + seg_off[s];
        for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) dst[i] = src[i];
    }
}

// stage 3: every k consecutive minimizers -> one reference k-min-mer (KminmersIterator; Entry::new_with_mer src/index.rs:57-58)
__global__ void ref_kminmers_kernel(const Minimizer *__restrict__ dense, uint64_t n_mz, DevParams P, uint32_t ref_id,
                                    RefKmm *__restrict__ out) {
    const uint64_t n_kmm = n_mz - P.k + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_kmm; i += (uint64_t)gridDim.x * blockDim.x) {
        bool rev;
        const uint64_t key = kminmer_hash(P.k, [&](uint32_t j) { return (uint64_t)dense[i + j].hash; }, rev);
        RefKmm r;
        r.hash = key;
        r.start = dense[i].pos;
        r.end = dense[i + P.k - 1].pos + P.l - 1u;
        r.offset = (uint32_t)i;
        r.id_rc = (ref_id << 1) | (rev ? 1u : 0u);
        out[i] = r;
    }
}

// Index::add_with_mer (src/index.rs:100-104) made order independent: the first claimant of a slot stores the entry,
// every insertion bumps the slot's count; a slot is live iff count == 1 (and end != 0, src/index.rs:67-69).
__global__ void insert_kernel(const RefKmm *__restrict__ kmm, uint64_t n, Slot *__restrict__ table, uint64_t mask) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const RefKmm r = kmm[i];
        uint64_t s;
        bool won = false;
        if (r.hash == 0) {
            s = mask + 1;
            won = atomicAdd(&table[s].pad, 1u) == 0;  // pad counts claims on the key-0 slot
        } else {
            s = r.hash & mask;
            for (;;) {
                unsigned long long prev = atomicCAS(&table[s].key, 0ull, r.hash);
                if (prev == 0ull) { won = true; break; }
                if (prev == r.hash) break;
                s = (s + 1) & mask;
            }
        }
        if (won) {
            table[s].start = r.start;
            table[s].end = r.end;
            table[s].offset = r.offset;
            table[s].id_rc = r.id_rc;
        }
        atomicAdd(&table[s].count, 1u);
    }
}

// Index::get_count (src/index.rs:90-92) + number of distinct keys
__global__ void count_kernel(const Slot *__restrict__ table, uint64_t nslots_plus1, unsigned long long *__restrict__ acc) {
    unsigned long long live = 0, keys = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots_plus1; i += (uint64_t)gridDim.x * blockDim.x) {
        const Slot v = table[i];
        if (v.count != 0) {
            keys++;
            if (v.count == 1 && v.end != 0) live++;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        live += __shfl_xor(live, d, 64);
        keys += __shfl_xor(keys, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], live);
        atomicAdd(&acc[1], keys);
    }
}

__global__ void lookup_kernel(const Slot *__restrict__ table, uint64_t mask, const uint64_t *__restrict__ keys, uint32_t n,
                              uint8_t *__restrict__ found, mq_kminmer *__restrict__ entries, uint32_t *__restrict__ ref_ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Slot e = {};
    const bool hit = probe_table(table, mask, keys[i], e);
    found[i] = hit ? 1 : 0;
    mq_kminmer k;
    k.hash = keys[i];
    k.start = hit ? e.start : 0;
    k.end = hit ? e.end : 0;
    k.offset = hit ? e.offset : 0;
    k.rev = hit ? (e.id_rc & 1u) : 0;
    entries[i] = k;
    ref_ids[i] = hit ? (e.id_rc >> 1) : 0;
}

// =================================================================== host side

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

struct KmmChunk {
    RefKmm *d = nullptr;
    uint64_t n = 0;
};

struct mq_index {
    mq_params params;
    DevParams dp;
    int device = 0;
    int n_cu = 0;
    std::map<uint32_t, std::pair<std::string, uint64_t>> refs;
    std::vector<KmmChunk> chunks;
    uint64_t n_kmm_total = 0;
    bool finalized = false;
    Slot *table = nullptr;  // nslots + 1
    uint64_t nslots = 0;
    uint64_t *d_ref_lens = nullptr;
    uint64_t n_unique = 0, n_keys = 0;
    // map scratch
    MatchRec *scratch = nullptr;
    uint32_t cap_matches = 0;
    uint32_t grid = 0;
    uint32_t *d_counter = nullptr;  // [0] work counter, [1] fast-path reads, [2] general-path reads
    uint8_t *fast_scratch = nullptr;
    bool force_general = false;     // test hook MQ_FORCE_GENERAL=1: never take the fast seeding path
    uint32_t stop_after = 0;        // diagnostic MQ_STOP_AFTER (instruction-count attribution; results are NOT valid)
    bool timing_once = false;       // set by mq_map_probe_stats for one instrumented launch
    bool stage_timing = false;      // diagnostic MQ_STAGE_TIMING=1: s_memtime stamps per stage (never for reported numbers)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // staging for the host-buffer entry points
    uint8_t *st_bases = nullptr;
    uint64_t st_bases_cap = 0;
    uint64_t *st_off = nullptr;
    uint64_t st_off_cap = 0;
    mq_hit *st_out = nullptr;
    uint64_t st_out_cap = 0;
    int chain_chunk = 64;  // test hook: MQ_CHAIN_CHUNK=4 exercises the multi-chunk chain path
};

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
    p->reserved = 0;
}

}  // extern "C"

// (density as FH * u64::MAX as FH) as u64 with Rust's saturating float->int cast
static uint64_t density_bound(double density) {
    double d = density * 18446744073709551615.0;
    if (!(d > 0.0)) return 0;
    if (d >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)d;
}

static int use_device(const mq_index *idx) {
    HIPCHK(hipSetDevice(idx->device));
    return MQ_OK;
}

static int alloc_table(mq_index *idx, uint64_t nslots) {
    if (idx->table) {
        HIPCHK(hipFree(idx->table));
        idx->table = nullptr;
    }
    HIPCHK(hipMalloc((void **)&idx->table, (nslots + 1) * sizeof(Slot)));
    HIPCHK(hipMemset(idx->table, 0, (nslots + 1) * sizeof(Slot)));
    idx->nslots = nslots;
    return MQ_OK;
}

static int ensure_scratch(mq_index *idx, uint32_t max_len) {
    (void)max_len;
    if (!idx->d_counter) HIPCHK(hipMalloc((void **)&idx->d_counter, 128));
    if (!idx->ev0) {
        HIPCHK(hipEventCreate(&idx->ev0));
        HIPCHK(hipEventCreate(&idx->ev1));
    }
    if (!idx->grid) {
        int occ = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, map_kernel<64, true>, 64 * MAP_WAVES, 0));
        if (occ < 1) occ = 1;
        if (occ > 8) occ = 8;
        const char *oe = getenv("MQ_OCC");  // diagnostic: cap workgroups per CU
        if (oe && atoi(oe) >= 1 && atoi(oe) < occ) occ = atoi(oe);
        idx->grid = (uint32_t)(occ * idx->n_cu);  // workgroups; MAP_WAVES persistent waves each
    }
    const size_t n_waves = (size_t)idx->grid * MAP_WAVES;
    if (!idx->scratch) {
        // Match runs per read held in HBM scratch; a read with more runs is reported MQ_HIT_OVERFLOW (never silently wrong)
        const char *e = getenv("MQ_MATCH_CAP");
        idx->cap_matches = e ? (uint32_t)strtoul(e, nullptr, 10) : 2048u;
        if (idx->cap_matches < 1) idx->cap_matches = 1;
        HIPCHK(hipMalloc((void **)&idx->scratch, n_waves * idx->cap_matches * sizeof(MatchRec)));
    }
    if (!idx->fast_scratch) HIPCHK(hipMalloc((void **)&idx->fast_scratch, n_waves * (size_t)(FAST_EM_BYTES + FAST_HM_WORDS * 4u)));
    return MQ_OK;
}

extern "C" {

mq_index *mq_index_new(const mq_params *params, int device) {
    if (!params) {
        set_err(MQ_EINVAL, "params is NULL");
        return nullptr;
    }
    if (params->l < 1 || params->l > MAX_L || params->k < 1 || params->k > MAX_K) {
        set_err(MQ_EINVAL, "unsupported k/l: need 1 <= l <= 64 and 1 <= k <= 32");
        return nullptr;
    }
    int n = mq_device_count();
    if (n <= 0) {
        set_err(MQ_ENODEVICE, "no HIP device: the mapquik HIP path has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        set_err(MQ_EINVAL, "device ordinal out of range");
        return nullptr;
    }
    mq_index *idx = new mq_index();
    idx->params = *params;
    idx->device = device;
    idx->dp.bound = density_bound(params->density);
    idx->dp.k = params->k;
    idx->dp.l = params->l;
    idx->dp.use_hpc = params->use_hpc ? 1 : 0;
    idx->dp.c = params->c;
    idx->dp.s = params->s;
    idx->dp.g = params->g;
    idx->dp.pad = 0;
    const char *cc = getenv("MQ_CHAIN_CHUNK");
    if (cc && atoi(cc) == 4) idx->chain_chunk = 4;
    const char *fg = getenv("MQ_FORCE_GENERAL");
    idx->force_general = fg && atoi(fg) != 0;
    const char *stt = getenv("MQ_STAGE_TIMING");
    idx->stage_timing = stt && atoi(stt) != 0;
    const char *sa = getenv("MQ_STOP_AFTER");
    idx->stop_after = sa ? (uint32_t)atoi(sa) : 0u;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        set_err(MQ_EHIP, "hipSetDevice/hipGetDeviceProperties failed");
        delete idx;
        return nullptr;
    }
    idx->n_cu = prop.multiProcessorCount;
    // a 1-slot empty table so that seeding-only calls work before finalize
    if (alloc_table(idx, 1) != MQ_OK) {
        delete idx;
        return nullptr;
    }
    return idx;
}

void mq_index_free(mq_index *idx) {
    if (!idx) return;
    hipSetDevice(idx->device);
    for (auto &c : idx->chunks)
        if (c.d) hipFree(c.d);
    if (idx->table) hipFree(idx->table);
    if (idx->d_ref_lens) hipFree(idx->d_ref_lens);
    if (idx->scratch) hipFree(idx->scratch);
    if (idx->d_counter) hipFree(idx->d_counter);
    if (idx->fast_scratch) hipFree(idx->fast_scratch);
    if (idx->st_bases) hipFree(idx->st_bases);
    if (idx->st_off) hipFree(idx->st_off);
    if (idx->st_out) hipFree(idx->st_out);
    if (idx->ev0) hipEventDestroy(idx->ev0);
    if (idx->ev1) hipEventDestroy(idx->ev1);
    delete idx;
}

int64_t mq_index_add_ref_device(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *d_seq, uint64_t len) {
    if (!idx || (!d_seq && len)) return set_err(MQ_EINVAL, "bad arguments");
    if (idx->finalized) return set_err(MQ_ESTATE, "index already finalized");
    if (len >= (1ull << 32)) return set_err(MQ_EINVAL, "sequence length must be < 2^32");
    if (ref_id >= (1u << 31)) return set_err(MQ_EINVAL, "ref_id must be < 2^31");
    if (idx->refs.count(ref_id)) return set_err(MQ_EINVAL, "duplicate ref_id");
    int rc = use_device(idx);
    if (rc) return rc;
    idx->refs[ref_id] = std::make_pair(std::string(name ? name : ""), len);
    const DevParams &P = idx->dp;
    if (len < (uint64_t)P.l + P.k - 1) return 0;  // src/mers.rs:18

    const uint64_t seg_len = 1ull << 16;
    const uint32_t n_seg = (uint32_t)((len + seg_len - 1) / seg_len);
    // expected minimizers per segment: 2 * density of the compressed l-mers; cap with slack, worst case on retry
    double dens = idx->params.density;
    if (!(dens > 0)) dens = 0;
    if (dens > 1) dens = 1;
    uint32_t cap = (uint32_t)std::min<double>((double)seg_len, 3.0 * 2.0 * dens * (double)seg_len + 1024.0);
    Minimizer *seg_lists = nullptr, *dense = nullptr;
    uint32_t *d_counts = nullptr;
    uint64_t *d_seg_off = nullptr;
    std::vector<uint32_t> counts(n_seg);
    std::vector<uint64_t> seg_off(n_seg + 1);
    auto cleanup = [&]() {
        if (seg_lists) hipFree(seg_lists);
        if (dense) hipFree(dense);
        if (d_counts) hipFree(d_counts);
        if (d_seg_off) hipFree(d_seg_off);
    };
#define HIPCHK_C(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            cleanup();                                                                                       \
            char _b[512];                                                                                    \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return set_err(_e == hipErrorOutOfMemory ? MQ_ENOMEM : MQ_EHIP, _b);                             \
        }                                                                                                    \
    } while (0)
    HIPCHK_C(hipMalloc((void **)&d_counts, (size_t)n_seg * sizeof(uint32_t)));
    const uint32_t grid = std::min<uint32_t>(n_seg, (uint32_t)idx->n_cu * 32u);
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK_C(hipMalloc((void **)&seg_lists, (size_t)n_seg * cap * sizeof(Minimizer)));
        hipLaunchKernelGGL(seed_segments_kernel, dim3(grid), dim3(64), 0, 0, d_seq, len, seg_len, n_seg, P, seg_lists, cap, d_counts);
        HIPCHK_C(hipGetLastError());
        HIPCHK_C(hipMemcpy(counts.data(), d_counts, (size_t)n_seg * sizeof(uint32_t), hipMemcpyDeviceToHost));
        bool overflow = false;
        for (uint32_t s = 0; s < n_seg; ++s) overflow |= counts[s] > cap;
        if (!overflow) break;
        if (attempt == 1) {
            cleanup();
            return set_err(MQ_EOVERFLOW, "minimizer list overflow at worst-case capacity (internal error)");
        }
        HIPCHK_C(hipFree(seg_lists));
        seg_lists = nullptr;
        cap = (uint32_t)seg_len;  // a segment cannot hold more run heads than bases
    }
    seg_off[0] = 0;
    for (uint32_t s = 0; s < n_seg; ++s) seg_off[s + 1] = seg_off[s] + counts[s];
    const uint64_t n_mz = seg_off[n_seg];
    int64_t n_kmm = 0;
    if (n_mz >= P.k) {
        n_kmm = (int64_t)(n_mz - P.k + 1);
        HIPCHK_C(hipMalloc((void **)&d_seg_off, (size_t)(n_seg + 1) * sizeof(uint64_t)));
        HIPCHK_C(hipMemcpy(d_seg_off, seg_off.data(), (size_t)(n_seg + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIPCHK_C(hipMalloc((void **)&dense, (size_t)n_mz * sizeof(Minimizer)));
        hipLaunchKernelGGL(compact_minimizers_kernel, dim3(std::min<uint32_t>(n_seg, 65535u)), dim3(64), 0, 0, seg_lists, cap, d_counts,
                           d_seg_off, n_seg, dense);
        HIPCHK_C(hipGetLastError());
        KmmChunk ch;
        ch.n = (uint64_t)n_kmm;
        HIPCHK_C(hipMalloc((void **)&ch.d, (size_t)n_kmm * sizeof(RefKmm)));
        const uint32_t kb = (uint32_t)std::min<uint64_t>(((uint64_t)n_kmm + 255) / 256, 65535ull);
        hipLaunchKernelGGL(ref_kminmers_kernel, dim3(kb), dim3(256), 0, 0, dense, n_mz, P, ref_id, ch.d);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            hipFree(ch.d);
            cleanup();
            return set_err(MQ_EHIP, std::string("ref_kminmers_kernel: ") + hipGetErrorString(e));
        }
        idx->chunks.push_back(ch);
        idx->n_kmm_total += (uint64_t)n_kmm;
    }
    cleanup();
#undef HIPCHK_C
    return n_kmm;
}

int64_t mq_index_add_ref(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *seq, uint64_t len) {
    if (!idx || (!seq && len)) return set_err(MQ_EINVAL, "bad arguments");
    int rc = use_device(idx);
    if (rc) return rc;
    uint8_t *d = nullptr;
    if (len) {
        HIPCHK(hipMalloc((void **)&d, len));
        hipError_t e = hipMemcpy(d, seq, len, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            hipFree(d);
            return set_err(MQ_EHIP, std::string("hipMemcpy H2D: ") + hipGetErrorString(e));
        }
    }
    int64_t r = mq_index_add_ref_device(idx, ref_id, name, d, len);
    if (d) hipFree(d);
    return r;
}

int64_t mq_index_finalize(mq_index *idx) {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    if (idx->finalized) return (int64_t)idx->n_unique;
    int rc = use_device(idx);
    if (rc) return rc;
    uint64_t nslots = 1024;
    const char *lf = getenv("MQ_TABLE_FACTOR");  // slots per inserted k-min-mer (power-of-two rounding on top); default 4 => load <= 0.25:
    // ~85 % of a read's lookups miss, a miss walks to the first empty slot, and every extra step is a dependent 128-B line fill
    const uint64_t factor = lf && atoi(lf) >= 1 ? (uint64_t)atoi(lf) : 4ull;
    while (nslots < factor * idx->n_kmm_total) nslots <<= 1;
    rc = alloc_table(idx, nslots);
    if (rc) return rc;
    for (auto &c : idx->chunks) {
        if (!c.n) continue;
        const uint32_t nb = (uint32_t)std::min<uint64_t>((c.n + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(insert_kernel, dim3(nb), dim3(256), 0, 0, c.d, c.n, idx->table, nslots - 1);
        HIPCHK(hipGetLastError());
    }
    unsigned long long *d_acc = nullptr;
    HIPCHK(hipMalloc((void **)&d_acc, 16));
    HIPCHK(hipMemset(d_acc, 0, 16));
    const uint32_t nb = (uint32_t)std::min<uint64_t>((nslots + 1 + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(count_kernel, dim3(nb), dim3(256), 0, 0, idx->table, nslots + 1, d_acc);
    HIPCHK(hipGetLastError());
    unsigned long long acc[2] = {0, 0};
    HIPCHK(hipMemcpy(acc, d_acc, 16, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d_acc));
    idx->n_unique = acc[0];
    idx->n_keys = acc[1];
    for (auto &c : idx->chunks)
        if (c.d) hipFree(c.d);
    idx->chunks.clear();
    // ref_map lengths (src/closures.rs:49), dense by ref id
    uint32_t max_id = 0;
    for (auto &kv : idx->refs) max_id = std::max(max_id, kv.first);
    std::vector<uint64_t> lens((size_t)max_id + 1, 0);
    for (auto &kv : idx->refs) lens[kv.first] = kv.second.second;
    HIPCHK(hipMalloc((void **)&idx->d_ref_lens, lens.size() * sizeof(uint64_t)));
    HIPCHK(hipMemcpy(idx->d_ref_lens, lens.data(), lens.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    idx->finalized = true;
    return (int64_t)idx->n_unique;
}

int mq_index_get_stats(const mq_index *idx, mq_index_stats *out) {
    if (!idx || !out) return set_err(MQ_EINVAL, "bad arguments");
    out->n_refs = idx->refs.size();
    out->n_kminmers = idx->n_kmm_total;
    out->n_keys = idx->n_keys;
    out->n_unique = idx->n_unique;
    out->table_slots = idx->nslots;
    out->table_bytes = (idx->nslots + 1) * sizeof(Slot);
    out->slot_bytes = sizeof(Slot);
    return MQ_OK;
}

// On-disk index (the reference has none and rebuilds on every run, src/closures.rs:24-94): header, parameters, reference
// table, then the finalized slot table verbatim.  Little-endian, this library's layout (MQ_INDEX_MAGIC names the version).
static const char MQ_INDEX_MAGIC[8] = {'M', 'Q', 'H', 'I', 'P', 'I', 'X', '1'};

int mq_index_save(const mq_index *idx, const char *path) {
    if (!idx || !path) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    int rc = use_device(idx);
    if (rc) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return set_err(MQ_EINVAL, std::string("cannot open for writing: ") + path);
    bool ok = fwrite(MQ_INDEX_MAGIC, 1, 8, f) == 8;
    const uint64_t hdr[6] = {sizeof(Slot), idx->nslots, idx->n_kmm_total, idx->n_keys, idx->n_unique, (uint64_t)idx->refs.size()};
    ok = ok && fwrite(&idx->params, sizeof(mq_params), 1, f) == 1 && fwrite(hdr, sizeof(hdr), 1, f) == 1;
    for (auto &kv : idx->refs) {
        const uint32_t id = kv.first, nl = (uint32_t)kv.second.first.size();
        ok = ok && fwrite(&id, 4, 1, f) == 1 && fwrite(&nl, 4, 1, f) == 1 && fwrite(&kv.second.second, 8, 1, f) == 1 &&
             (nl == 0 || fwrite(kv.second.first.data(), 1, nl, f) == nl);
    }
    const size_t total = (size_t)(idx->nslots + 1) * sizeof(Slot), chunk = 64u << 20;
    std::vector<uint8_t> buf(std::min(total, chunk));
    for (size_t o = 0; ok && o < total; o += chunk) {
        const size_t n = std::min(chunk, total - o);
        if (hipMemcpy(buf.data(), (const uint8_t *)idx->table + o, n, hipMemcpyDeviceToHost) != hipSuccess) {
            fclose(f);
            return set_err(MQ_EHIP, "hipMemcpy D2H failed while saving the index");
        }
        ok = fwrite(buf.data(), 1, n, f) == n;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? MQ_OK : set_err(MQ_EINVAL, std::string("short write: ") + path);
}

mq_index *mq_index_load(const char *path, int device) {
    if (!path) {
        set_err(MQ_EINVAL, "path is NULL");
        return nullptr;
    }
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_err(MQ_EINVAL, std::string("cannot open: ") + path);
        return nullptr;
    }
    char magic[8];
    mq_params p;
    uint64_t hdr[6];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, MQ_INDEX_MAGIC, 8) != 0 || fread(&p, sizeof(p), 1, f) != 1 ||
        fread(hdr, sizeof(hdr), 1, f) != 1 || hdr[0] != sizeof(Slot) || hdr[1] == 0 || (hdr[1] & (hdr[1] - 1)) != 0) {
        fclose(f);
        set_err(MQ_EINVAL, std::string("not a mapquik HIP index (or another layout version): ") + path);
        return nullptr;
    }
    mq_index *idx = mq_index_new(&p, device);
    if (!idx) {
        fclose(f);
        return nullptr;
    }
    bool ok = true;
    for (uint64_t i = 0; ok && i < hdr[5]; ++i) {
        uint32_t id = 0, nl = 0;
        uint64_t len = 0;
        ok = fread(&id, 4, 1, f) == 1 && fread(&nl, 4, 1, f) == 1 && fread(&len, 8, 1, f) == 1 && nl < (1u << 20);
        std::string name(nl, '\0');
        ok = ok && (nl == 0 || fread(&name[0], 1, nl, f) == nl);
        if (ok) idx->refs[id] = std::make_pair(name, len);
    }
    if (ok && alloc_table(idx, hdr[1]) != MQ_OK) ok = false;
    const size_t total = (size_t)(hdr[1] + 1) * sizeof(Slot), chunk = 64u << 20;
    std::vector<uint8_t> buf(std::min(total, chunk));
    for (size_t o = 0; ok && o < total; o += chunk) {
        const size_t n = std::min(chunk, total - o);
        ok = fread(buf.data(), 1, n, f) == n && hipMemcpy((uint8_t *)idx->table + o, buf.data(), n, hipMemcpyHostToDevice) == hipSuccess;
    }
    fclose(f);
    if (ok) {
        uint32_t max_id = 0;
        for (auto &kv : idx->refs) max_id = std::max(max_id, kv.first);
        std::vector<uint64_t> lens((size_t)max_id + 1, 0);
        for (auto &kv : idx->refs) lens[kv.first] = kv.second.second;
        ok = hipMalloc((void **)&idx->d_ref_lens, lens.size() * sizeof(uint64_t)) == hipSuccess &&
             hipMemcpy(idx->d_ref_lens, lens.data(), lens.size() * sizeof(uint64_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        mq_index_free(idx);
        set_err(MQ_EINVAL, std::string("truncated or unreadable index file: ") + path);
        return nullptr;
    }
    idx->n_kmm_total = hdr[2];
    idx->n_keys = hdr[3];
    idx->n_unique = hdr[4];
    idx->finalized = true;
    return idx;
}

int mq_index_ref_info(const mq_index *idx, uint32_t ref_id, const char **name, uint64_t *len) {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    auto it = idx->refs.find(ref_id);
    if (it == idx->refs.end()) return set_err(MQ_EINVAL, "unknown ref_id");
    if (name) *name = it->second.first.c_str();
    if (len) *len = it->second.second;
    return MQ_OK;
}

int mq_map_reserve(mq_index *idx, uint32_t max_len) {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    int rc = use_device(idx);
    if (rc) return rc;
    return ensure_scratch(idx, max_len);
}

}  // extern "C"

static int launch_map(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, mq_hit *d_out,
                      mq_kminmer *d_dump, const uint64_t *d_dump_off, uint32_t *d_dump_counts, hipStream_t st,
                      MatchRec *scratch_override = nullptr, uint32_t cap_override = 0, uint32_t grid_override = 0) {
    if (n == 0) return MQ_OK;
    HIPCHK(hipMemsetAsync(idx->d_counter, 0, 128, st));
    HIPCHK(hipEventRecord(idx->ev0, st));
    MapArgs A;
    A.bases = d_bases;
    A.offsets = d_offsets;
    A.n = n;
    A.P = idx->dp;
    A.table = idx->table;
    A.mask = idx->nslots - 1;
    A.ref_lens = idx->d_ref_lens;
    A.scratch_all = scratch_override ? scratch_override : idx->scratch;
    A.cap_matches = scratch_override ? cap_override : idx->cap_matches;
    A.fast_scratch = idx->fast_scratch;
    A.work_counter = idx->d_counter;
    A.out = d_out;
    A.dump = d_dump;
    A.dump_off = d_dump_off;
    A.dump_counts = d_dump_counts;
    A.stats = idx->d_counter + 2;
    A.stop_after = idx->stop_after;  // [2] fast reads, [3] general reads, [4..15] six 64-bit stage cycle sums
    uint32_t grid = std::min<uint32_t>(idx->grid, (n + MAP_WAVES - 1) / MAP_WAVES);
    if (grid_override) grid = std::min(grid, grid_override);
    const dim3 blk(64 * MAP_WAVES);
    if (idx->stage_timing || idx->timing_once) hipLaunchKernelGGL((map_kernel<64, true, true>), dim3(grid), blk, 0, st, A);
    else if (idx->chain_chunk == 4 && idx->force_general) hipLaunchKernelGGL((map_kernel<4, false>), dim3(grid), blk, 0, st, A);
    else if (idx->chain_chunk == 4) hipLaunchKernelGGL((map_kernel<4, true>), dim3(grid), blk, 0, st, A);
    else if (idx->force_general) hipLaunchKernelGGL((map_kernel<64, false>), dim3(grid), blk, 0, st, A);
    else hipLaunchKernelGGL((map_kernel<64, true>), dim3(grid), blk, 0, st, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(idx->ev1, st));
    idx->ev_valid = true;
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

extern "C" {

int mq_map_batch_device(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint32_t max_len,
                        mq_hit *d_out, void *stream) {
    if (!idx || (n && (!d_offsets || !d_out))) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    int rc = use_device(idx);
    if (rc) return rc;
    rc = ensure_scratch(idx, max_len);
    if (rc) return rc;
    return launch_map(idx, d_bases, d_offsets, n, d_out, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int mq_map_batch(mq_index *idx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out) {
    if (!idx || (n && (!offsets || !out))) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (n == 0) return MQ_OK;
    int rc = use_device(idx);
    if (rc) return rc;
    const uint64_t total = offsets[n] - offsets[0];
    uint64_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return set_err(MQ_EINVAL, "offsets must be non-decreasing");
        max_len = std::max(max_len, offsets[i + 1] - offsets[i]);
    }
    if (max_len >= (1ull << 32)) return set_err(MQ_EINVAL, "sequence length must be < 2^32");
    rc = ensure_scratch(idx, (uint32_t)max_len);
    if (rc) return rc;
    if ((rc = grow(idx->st_bases, idx->st_bases_cap, total + 1))) return rc;
    if ((rc = grow(idx->st_off, idx->st_off_cap, (uint64_t)n + 1))) return rc;
    if ((rc = grow(idx->st_out, idx->st_out_cap, (uint64_t)n))) return rc;
    std::vector<uint64_t> rel((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) rel[i] = offsets[i] - offsets[0];
    if (total) HIPCHK(hipMemcpy(idx->st_bases, bases + offsets[0], total, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(idx->st_off, rel.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    rc = launch_map(idx, idx->st_bases, idx->st_off, n, idx->st_out, nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    HIPCHK(hipMemcpy(out, idx->st_out, (size_t)n * sizeof(mq_hit), hipMemcpyDeviceToHost));
    // Reads whose Match runs did not fit the per-wave scratch come back as MQ_HIT_OVERFLOW: map those again on the GPU
    // with a worst-case scratch (a read cannot have more runs than bases) on a small grid.  Never a CPU path.
    std::vector<uint32_t> redo;
    for (uint32_t i = 0; i < n; ++i)
        if (out[i].status == MQ_HIT_OVERFLOW) redo.push_back(i);
    if (!redo.empty()) {
        uint64_t sub_total = 0, sub_max = 0;
        std::vector<uint64_t> so(redo.size() + 1, 0);
        for (size_t j = 0; j < redo.size(); ++j) {
            const uint64_t L = offsets[redo[j] + 1] - offsets[redo[j]];
            so[j + 1] = so[j] + L;
            sub_max = std::max(sub_max, L);
        }
        sub_total = so.back();
        std::vector<uint8_t> sb(sub_total ? sub_total : 1);
        for (size_t j = 0; j < redo.size(); ++j)
            memcpy(sb.data() + so[j], bases + offsets[redo[j]], (size_t)(so[j + 1] - so[j]));
        const uint32_t cap = (uint32_t)std::max<uint64_t>(sub_max, 1);
        const uint32_t rgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(idx->grid, (1ull << 30) / ((uint64_t)cap * sizeof(MatchRec) * MAP_WAVES)));
        MatchRec *big = nullptr;
        uint8_t *d_sb = nullptr;
        uint64_t *d_so = nullptr;
        mq_hit *d_sh = nullptr;
        hipError_t e = hipMalloc((void **)&big, (size_t)rgrid * MAP_WAVES * cap * sizeof(MatchRec));
        if (e == hipSuccess) e = hipMalloc((void **)&d_sb, sub_total + 1);
        if (e == hipSuccess) e = hipMalloc((void **)&d_so, so.size() * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&d_sh, redo.size() * sizeof(mq_hit));
        if (e == hipSuccess && sub_total) e = hipMemcpy(d_sb, sb.data(), sub_total, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_so, so.data(), so.size() * 8, hipMemcpyHostToDevice);
        int rrc = MQ_OK;
        std::vector<mq_hit> sh(redo.size());
        if (e == hipSuccess) rrc = launch_map(idx, d_sb, d_so, (uint32_t)redo.size(), d_sh, nullptr, nullptr, nullptr, 0, big, cap, rgrid);
        if (e == hipSuccess && rrc == MQ_OK) e = hipMemcpy(sh.data(), d_sh, redo.size() * sizeof(mq_hit), hipMemcpyDeviceToHost);
        hipFree(big); hipFree(d_sb); hipFree(d_so); hipFree(d_sh);
        if (e != hipSuccess) return set_err(e == hipErrorOutOfMemory ? MQ_ENOMEM : MQ_EHIP, std::string("overflow retry: ") + hipGetErrorString(e));
        if (rrc) return rrc;
        for (size_t j = 0; j < redo.size(); ++j) out[redo[j]] = sh[j];
    }
    return MQ_OK;
}

int mq_kminmers_batch(mq_index *idx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, const uint64_t *kmm_offsets,
                      mq_kminmer *out, uint32_t *counts) {
    if (!idx || (n && (!offsets || !kmm_offsets || !counts))) return set_err(MQ_EINVAL, "bad arguments");
    if (n == 0) return MQ_OK;
    int rc = use_device(idx);
    if (rc) return rc;
    rc = ensure_scratch(idx, 0);
    if (rc) return rc;
    const uint64_t total = offsets[n] - offsets[0];
    const uint64_t ktotal = kmm_offsets[n] - kmm_offsets[0];
    uint8_t *d_b = nullptr;
    uint64_t *d_o = nullptr, *d_ko = nullptr;
    mq_kminmer *d_k = nullptr;
    uint32_t *d_c = nullptr;
    mq_hit *d_h = nullptr;
    uint64_t *d_zero_lens = nullptr;
    auto cleanup = [&]() {
        hipFree(d_b); hipFree(d_o); hipFree(d_ko); hipFree(d_k); hipFree(d_c); hipFree(d_h); hipFree(d_zero_lens);
    };
    std::vector<uint64_t> rel((size_t)n + 1), krel((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) {
        rel[i] = offsets[i] - offsets[0];
        krel[i] = kmm_offsets[i] - kmm_offsets[0];
    }
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    ok(hipMalloc((void **)&d_b, total + 1));
    ok(hipMalloc((void **)&d_o, ((size_t)n + 1) * 8));
    ok(hipMalloc((void **)&d_ko, ((size_t)n + 1) * 8));
    ok(hipMalloc((void **)&d_k, (ktotal + 1) * sizeof(mq_kminmer)));
    ok(hipMalloc((void **)&d_c, (size_t)n * 4));
    ok(hipMalloc((void **)&d_h, (size_t)n * sizeof(mq_hit)));
    if (e == hipSuccess && total) ok(hipMemcpy(d_b, bases + offsets[0], total, hipMemcpyHostToDevice));
    if (e == hipSuccess) ok(hipMemcpy(d_o, rel.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    if (e == hipSuccess) ok(hipMemcpy(d_ko, krel.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    if (e != hipSuccess) {
        cleanup();
        return set_err(MQ_EHIP, std::string("mq_kminmers_batch setup: ") + hipGetErrorString(e));
    }
    // before finalize there is no ref table: the 1-slot empty table never hits, so ref_lens is never read
    rc = launch_map(idx, d_b, d_o, n, d_h, d_k, d_ko, d_c, 0);
    if (rc) {
        cleanup();
        return rc;
    }
    ok(hipMemcpy(counts, d_c, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (e == hipSuccess && ktotal && out) ok(hipMemcpy(out + kmm_offsets[0], d_k, ktotal * sizeof(mq_kminmer), hipMemcpyDeviceToHost));
    cleanup();
    if (e != hipSuccess) return set_err(MQ_EHIP, std::string("mq_kminmers_batch copy-out: ") + hipGetErrorString(e));
    return MQ_OK;
}

int mq_index_lookup(mq_index *idx, const uint64_t *hashes, uint32_t n, uint8_t *found, mq_kminmer *entries, uint32_t *ref_ids) {
    if (!idx || (n && (!hashes || !found || !entries || !ref_ids))) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (n == 0) return MQ_OK;
    int rc = use_device(idx);
    if (rc) return rc;
    uint64_t *d_k = nullptr;
    uint8_t *d_f = nullptr;
    mq_kminmer *d_e = nullptr;
    uint32_t *d_r = nullptr;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    ok(hipMalloc((void **)&d_k, (size_t)n * 8));
    ok(hipMalloc((void **)&d_f, (size_t)n));
    ok(hipMalloc((void **)&d_e, (size_t)n * sizeof(mq_kminmer)));
    ok(hipMalloc((void **)&d_r, (size_t)n * 4));
    if (e == hipSuccess) ok(hipMemcpy(d_k, hashes, (size_t)n * 8, hipMemcpyHostToDevice));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lookup_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, idx->table, idx->nslots - 1, d_k, n, d_f, d_e, d_r);
        ok(hipGetLastError());
    }
    if (e == hipSuccess) ok(hipMemcpy(found, d_f, (size_t)n, hipMemcpyDeviceToHost));
    if (e == hipSuccess) ok(hipMemcpy(entries, d_e, (size_t)n * sizeof(mq_kminmer), hipMemcpyDeviceToHost));
    if (e == hipSuccess) ok(hipMemcpy(ref_ids, d_r, (size_t)n * 4, hipMemcpyDeviceToHost));
    hipFree(d_k); hipFree(d_f); hipFree(d_e); hipFree(d_r);
    if (e != hipSuccess) return set_err(MQ_EHIP, std::string("mq_index_lookup: ") + hipGetErrorString(e));
    return MQ_OK;
}

int mq_format_paf(const mq_index *idx, const char *q_id, uint64_t q_len, const mq_hit *hit, char *buf, size_t cap) {
    if (!idx || !q_id || !hit || !buf) return set_err(MQ_EINVAL, "bad arguments");
    if (hit->status != MQ_HIT_MAPPED) return set_err(MQ_EINVAL, "hit is not mapped: the reference writes no line");
    auto it = idx->refs.find(hit->ref_id);
    if (it == idx->refs.end()) return set_err(MQ_EINVAL, "unknown ref_id in hit");
    const unsigned long long r_len = it->second.second;
    // src/mers.rs:181: column 11 repeats r_len, column 10 is the score
    int w = snprintf(buf, cap, "%s\t%llu\t%u\t%u\t%s\t%s\t%llu\t%u\t%u\t%u\t%llu\t%u", q_id, (unsigned long long)q_len, hit->q_start,
                     hit->q_end, hit->rc ? "-" : "+", it->second.first.c_str(), r_len, hit->r_start, hit->r_end, hit->score, r_len,
                     hit->mapq);
    return w;
}

void *mq_host_alloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_err(e == hipErrorOutOfMemory ? MQ_ENOMEM : MQ_EHIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

void mq_host_free(void *p) {
    if (p) hipHostFree(p);
}

int mq_last_map_path_counts(mq_index *idx, uint32_t *n_fast, uint32_t *n_general) {
    if (!idx || !n_fast || !n_general) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(idx->ev1));
    uint32_t v[2] = {0, 0};
    HIPCHK(hipMemcpy(v, idx->d_counter + 2, 8, hipMemcpyDeviceToHost));
    *n_fast = v[0];
    *n_general = v[1];
    return MQ_OK;
}

int mq_last_stage_cycles(mq_index *idx, uint64_t *cycles9) {
    if (!idx || !cycles9) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(idx->ev1));
    HIPCHK(hipMemcpy(cycles9, idx->d_counter + 4, 72, hipMemcpyDeviceToHost));
    return MQ_OK;
}

int mq_map_probe_stats(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint32_t max_len, mq_hit *d_out,
                       uint64_t *lookups, uint64_t *extra_steps) {
    if (!idx || !lookups || !extra_steps) return set_err(MQ_EINVAL, "bad arguments");
    idx->timing_once = true;
    int rc = mq_map_batch_device(idx, d_bases, d_offsets, n, max_len, d_out, nullptr);
    idx->timing_once = false;
    if (rc) return rc;
    uint64_t v[9];
    rc = mq_last_stage_cycles(idx, v);
    if (rc) return rc;
    *extra_steps = v[6];
    *lookups = v[7];
    return MQ_OK;
}

int mq_last_map_ms(mq_index *idx, float *ms) {
    if (!idx || !ms) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    HIPCHK(hipEventSynchronize(idx->ev1));
    HIPCHK(hipEventElapsedTime(ms, idx->ev0, idx->ev1));
    return MQ_OK;
}

}  // extern "C"
