// mq_capi.hip -- kernels + the extern "C" boundary declared in include/mapquik_hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see mapquik_amd/build.py).  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mq_device.hpp"
#include "mq_seed.hpp"

using namespace mq;

// =================================================================== kernels

// The map path.  One wave per read, persistent waves pulling read indices from an atomic counter.  Two phases per read:
//   seed   read -> ordered minimizer list {hash, raw position} in the read's HBM region     (mq_seed.hpp for ACGT-only reads,
//          the general streaming seeder seed_segment of mq_device.hpp for the rest)
//   map    list -> k-min-mers -> index probe -> Match runs -> chain -> mq_hit                (MapSink, chain_stage)
// map_kernel runs both phases back to back in the same wave (default): while one wave waits for its index probes (random
// 32-B slot reads: ~42 G lookups/s is all the memory system gives, tools/probe_rate.py) the other waves of the SIMD seed.
// MQ_PIPELINE=split runs the phases as three launches (seed_reads_kernel, seed_general_kernel, map_lists_kernel) so that a
// profiler prices each phase by itself; same device functions, same results.
// Read r's list lives at entries [base_r, base_r + cap_r) of mz_hash[] / mz_pos[]:
//   base_r = ((o0_r - o0_0) * f16 >> 16) + slack * r,   cap_r = (len_r * f16 >> 16) + slack
// (regions never overlap; f16/65536 = list entries reserved per base).  A list that outgrows its region (a read inside a
// short-period tandem array can be far denser than 2 d) is written again, at its now known size, into an exact-size region
// taken from a shared pool behind the regular regions.  Only when the pool is exhausted too does the read come back as
// MQ_HIT_OVERFLOW (the host-buffer entry points then redo it with f16 = 65536).
struct SplitArgs {
    const uint8_t *bases;
    const uint64_t *offsets;  // n + 1: read r starts at offsets[r]; offsets[n] = end of the buffer
    const uint32_t *lens;     // null: read r ends at offsets[r + 1]; else its length (raw FASTX buffers: headers and quality lines in between)
    uint32_t n;
    DevParams P;
    unsigned long long *mz_hash;
    uint32_t *mz_pos;
    uint32_t *mz_count;    // split pipeline only: list length of read r (or NOT_FAST / LIST_OVERFLOW)
    uint64_t *mz_base;     // split pipeline only: where read r's list starts (its regular region or a pool region)
    uint64_t pool_base, pool_cap;  // the pool: entries [pool_base, pool_base + pool_cap)
    uint32_t f16, slack;
    uint32_t *queue;       // split pipeline only: reads for the general seeder
    uint32_t *counters;    // [0] seed work, [1] map work, [2] queue length, [3] general work, [4] fast reads, [5] general reads,
                           // [6] lists moved to the pool, [12..13] 64-bit pool cursor
    uint32_t force_general;
    const Bucket *table;
    uint64_t mask;
    const uint64_t *ref_lens;
    MatchRec *scratch_all;  // per mapping wave: cap_matches records
    uint32_t cap_matches;
    mq_hit *out;
    mq_kminmer *dump;
    const uint64_t *dump_off;
    uint32_t *dump_counts;
    unsigned long long *stats64;  // instrumented launch only: [0] slots visited beyond the home slot, [1] lookups
};

__device__ __forceinline__ void list_region(const SplitArgs &A, uint64_t o0_rel, uint64_t len, uint32_t r, uint64_t &base, uint32_t &cap) {
    base = ((o0_rel * A.f16) >> 16) + (uint64_t)A.slack * r;
    const uint64_t c = ((len * A.f16) >> 16) + A.slack;
    cap = c > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)c;
}
constexpr uint32_t LIST_OVERFLOW = 0xFFFFFFFEu;  // list length value: the list fits neither its region nor the pool
// an exact-size pool region for a list of cnt entries (wave-uniform); false when the pool is exhausted
__device__ __forceinline__ bool pool_take(const SplitArgs &A, uint32_t cnt, uint64_t &base) {
    unsigned long long at = 0;
    if (lane_id() == 0) at = atomicAdd(reinterpret_cast<unsigned long long *>(A.counters + 12), (unsigned long long)cnt);
    at = rdlane64(at, 0);
    base = A.pool_base + at;
    return at + cnt <= A.pool_cap;
}

// seed phase, fast seeder: list length, SD_NOT_FAST (declined: non-ACGT byte, ...) or LIST_OVERFLOW; base moves with the list
template <int STOP = 0>
__device__ __forceinline__ uint32_t seed_read_fast(const SplitArgs &A, const SeedTables &T, SeedLds &S, const uint8_t *seq,
                                                   uint32_t len, uint64_t &base, uint32_t cap, uint32_t &n_moved, APre &pre, bool pre_valid) {
    uint32_t cnt = seed_sequence_fast<STOP>(seq, len, A.P, T, S, A.mz_hash + base, A.mz_pos + base, cap, pre, pre_valid);
    if (cnt != SD_NOT_FAST && cnt > cap) {  // denser than its region: once more, into an exact-size pool region
        if (pool_take(A, cnt, base)) {
            seed_sequence_fast(seq, len, A.P, T, S, A.mz_hash + base, A.mz_pos + base, cnt, pre, false);
            n_moved++;
        } else {
            cnt = LIST_OVERFLOW;
        }
    }
    return cnt;
}

// seed phase, general streaming seeder (any bytes, any length)
__device__ __forceinline__ uint32_t seed_read_general(const SplitArgs &A, WaveLds &S, const uint8_t *seq, uint64_t len, uint64_t &base,
                                                      uint32_t cap, uint32_t &n_moved) {
    uint32_t cnt;
    {
        SoaListSink sink(A.mz_hash + base, A.mz_pos + base, cap);
        uint32_t mz_count = 0;
        seed_segment(seq, len, 0, len, A.P, S, sink, mz_count);
        cnt = sink.written;
    }
    if (cnt > cap) {
        if (pool_take(A, cnt, base)) {
            SoaListSink sink(A.mz_hash + base, A.mz_pos + base, cnt);
            uint32_t mz_count = 0;
            seed_segment(seq, len, 0, len, A.P, S, sink, mz_count);
            n_moved++;
        } else {
            cnt = LIST_OVERFLOW;
        }
    }
    return cnt;
}

#ifndef MQ_ML_NB
#define MQ_ML_NB 7
#endif
constexpr int ML_NB = MQ_ML_NB;                              // lane-batches of 64 k-min-mers hashed and probed together
constexpr uint32_t ML_LIST_CAP = 64 * ML_NB + 64;      // minimizers staged in LDS at a time (64 * ML_NB + k - 1 used)
struct MapListLds {
    unsigned long long h[ML_LIST_CAP];
    uint32_t p[ML_LIST_CAP];
};

// map phase of read r: its list (cnt entries at base) -> mq_hit
// the read's result is left in h (all lanes hold it); store_hit() writes it: the fused kernel does that after it has taken the
// prefetched offsets of its next read out of their registers, so that this store's acknowledgement is nothing a wave waits for
__device__ __forceinline__ void store_hit(const SplitArgs &A, uint32_t r, const mq_hit &h) {
    if (lane_id() == 0) {
        A.out[r] = h;
        if (A.dump_counts) A.dump_counts[r] = h.n_kminmers;
    }
}

template <int CH, bool TIMING>
__device__ __forceinline__ void map_read(const SplitArgs &A, MapListLds &S, MatchRec *scratch, uint32_t r, uint64_t len, uint32_t cnt,
                                         uint64_t base, unsigned long long &t_steps, unsigned long long &t_lookups, mq_hit &h) {
    const uint32_t lane = lane_id();
    const DevParams &P = A.P;
    h.status = MQ_HIT_UNMAPPED;
    h.ref_id = h.rc = h.mapq = h.q_start = h.q_end = h.r_start = h.r_end = h.score = h.n_kminmers = h.q_start_hi = h.q_end_hi = 0;
    uint32_t n_kmm = 0;
    if (cnt == LIST_OVERFLOW) {
        h.status = MQ_HIT_OVERFLOW;  // the list fits neither its region nor the pool: nothing was computed for this read
    } else if (cnt >= P.k) {
        mq_kminmer *d = nullptr;
        uint32_t dcap = 0;
        if (A.dump) {
            d = A.dump + A.dump_off[r];
            dcap = (uint32_t)(A.dump_off[r + 1] - A.dump_off[r]);
        }
        MapSink sink(A.table, A.mask, P, scratch, A.cap_matches, d, dcap);
        const unsigned long long *lh = A.mz_hash + base;
        const uint32_t *lp = A.mz_pos + base;
        const uint32_t chunk = 64u * (uint32_t)ML_NB + P.k - 1u;
        for (uint32_t g = 0; g + P.k <= cnt;) {
            const uint32_t have = cnt - g < chunk ? cnt - g : chunk;
            {  // L2-served loads (the list may have been written by this very wave), ALL in flight before the first is stored: one L2
               // round trip per chunk (a loop that loads and stores 64 entries at a time exposes one per 64 entries)
                unsigned long long hv[ML_NB + 1];
                uint32_t pv[ML_NB + 1];
#pragma unroll
                for (int j = 0; j <= ML_NB; ++j) {
                    const uint32_t i = lane + 64u * (uint32_t)j;
                    hv[j] = 0;
                    pv[j] = 0;
                    if (i < have) {
                        hv[j] = ld_sc1_u64(lh + g + i);
                        pv[j] = ld_sc1_u32(lp + g + i);
                    }
                }
#pragma unroll
                for (int j = 0; j <= ML_NB; ++j) {
                    const uint32_t i = lane + 64u * (uint32_t)j;
                    if (i < have) {
                        S.h[i] = hv[j];
                        S.p[i] = pv[j];
                    }
                }
            }
            wave_sync();
            mq_clk(5);
            sink.template consume_list<ML_NB>(S.h, S.p, have);
            wave_sync();
            g += have - (P.k - 1u);
        }
        sink.finish_runs();
        n_kmm = sink.kmm_count;
        if (sink.n_matches > A.cap_matches) {
            h.status = MQ_HIT_OVERFLOW;
        } else if (sink.n_matches > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Match records written by this wave are in L2
            wave_sync();
            mq_clk(8);
            chain_stage<CH>(scratch, sink.n_matches, P, len, A.ref_lens, h);
        }
        if (TIMING) {
            t_steps += wave_sum_u32(sink.probe_steps);
            t_lookups += n_kmm;
        }
    }
    h.n_kminmers = n_kmm;
    mq_clk(9);
}

#ifndef MQ_MAP_WAVES
#define MQ_MAP_WAVES 8
#endif
#ifndef MQ_MAP_MIN_WAVES
#define MQ_MAP_MIN_WAVES 4
#endif
constexpr int MAP_WAVES = MQ_MAP_WAVES;

// per-wave LDS of the fused kernel: the phases of one read follow each other, so they share the memory
union MapWaveLds {
    SeedLds seed;
    WaveLds general;
    MapListLds map;
};

// CH: lanes per chunk in the chain stage (64 in production; 4 only in tests so that ordinary reads take the multi-chunk path)
template <int CH, bool TIMING = false>
__global__ __launch_bounds__(64 * MAP_WAVES, MQ_MAP_MIN_WAVES) void map_kernel(const SplitArgs A) {
    // one block of LDS with the tables FIRST: T.rot's entries are addressed through the 16-bit immediate offset of ds_read_b128
    __shared__ struct {
        SeedTables T;
        MapWaveLds SS[MAP_WAVES];
    } W;
    SeedTables &T = W.T;
    MapWaveLds(&SS)[MAP_WAVES] = W.SS;
    build_seed_tables(T, A.P.l);
    __syncthreads();  // the only workgroup-wide rendezvous; waves are independent from here on
    const uint32_t lane = lane_id();
    const uint32_t wv = rdfirst(threadIdx.x >> 6);  // wave-uniform: per-wave bases stay in SGPRs
    MapWaveLds &S = SS[wv];
    const size_t wave_gid = (size_t)blockIdx.x * MAP_WAVES + wv;
    MatchRec *scratch = A.scratch_all + wave_gid * A.cap_matches;
    const DevParams &P = A.P;
    const uint64_t o_base = A.offsets[0];
    uint32_t n_fast = 0, n_general = 0, n_moved = 0;
    unsigned long long t_steps = 0, t_lookups = 0;
#ifdef MQ_STAGE_CLOCKS
    if (lane == 0)
        for (int i = 0; i < MQ_N_CLK; ++i) mq_clk_lds().acc[wv][i] = 0;
    mq_clk(-1);
#endif
    // The work item after the current one is fetched while the current one is processed: its index (one atomic) during the seed
    // phase, its offsets during the map phase -- two dependent memory round trips per read that no wave waits for.  (Requesting
    // the next read's first super-row across the map phase as well was measured at -3 %: a wave's loads return in order, so the
    // map phase's first wait -- an L2 round trip for the list -- then sits behind an HBM one.)
    uint32_t r = 0;
    if (lane == 0) r = atomicAdd(&A.counters[0], 1u);
    r = rdfirst(r);
    uint64_t o0 = 0, len = 0;
    if (r < A.n) {
        o0 = A.offsets[r];
        len = A.lens ? (uint64_t)A.lens[r] : A.offsets[r + 1] - o0;
    }
    while (r < A.n) {
        uint32_t rn_v = 0;
        if (lane == 0) rn_v = atomicAdd(&A.counters[0], 1u);
        uint32_t cnt = 0;
        uint64_t base = 0;
        mq_clk(11);
        // extract(): len < l + k - 1 => None (src/mers.rs:44)
        if (len >> 32) {
            cnt = LIST_OVERFLOW;  // beyond the documented limit (checked on the host where the host sees the lengths): loud, not wrong
        } else if (len >= (uint64_t)P.l + P.k - 1u) {
            uint32_t cap;
            list_region(A, o0 - o_base, len, r, base, cap);
            APre pre;
            cnt = A.force_general ? SD_NOT_FAST : seed_read_fast(A, T, S.seed, A.bases + o0, (uint32_t)len, base, cap, n_moved, pre, false);
            if (cnt == SD_NOT_FAST) {
                n_general++;
                wave_sync();
                cnt = seed_read_general(A, S.general, A.bases + o0, len, base, cap, n_moved);
                mq_clk(10);
            } else {
                n_fast++;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's list stores have reached L2
            wave_sync();
            mq_clk(4);
        }
        const uint32_t rn = rdfirst(rn_v);
        unsigned long long n_o0 = 0, n_o1 = 0;
        uint32_t n_len = 0;
        if (lane == 0 && rn < A.n) {  // vector loads by one lane: in flight through the map phase (scalar loads would be waited for at its first LDS wait)
            n_o0 = A.offsets[rn];
            if (A.lens) n_len = A.lens[rn];
            else n_o1 = A.offsets[rn + 1];
        }
        mq_hit h;
        map_read<CH, TIMING>(A, S.map, scratch, r, len, cnt, base, t_steps, t_lookups, h);
        wave_sync();
        const uint32_t r_done = r;
        r = rn;
        o0 = rdlane64(n_o0, 0);
        len = A.lens ? (uint64_t)rdfirst(n_len) : rdlane64(n_o1, 0) - o0;
        asm volatile("" ::: "memory");  // the prefetched offsets are out of their registers before the result's store is issued
        store_hit(A, r_done, h);
    }
    if (lane == 0) {
        if (n_fast) atomicAdd(&A.counters[4], n_fast);
        if (n_general) atomicAdd(&A.counters[5], n_general);
        if (n_moved) atomicAdd(&A.counters[6], n_moved);
        if (TIMING) {
            atomicAdd(&A.stats64[0], t_steps);
            atomicAdd(&A.stats64[1], t_lookups);
        }
#ifdef MQ_STAGE_CLOCKS
        for (int i = 0; i < MQ_N_CLK; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(A.counters + 16) + i, mq_clk_lds().acc[wv][i]);
#endif
    }
}

// ------------------------------------------------------------------- the same phases as separate launches (MQ_PIPELINE=split)
#ifndef MQ_SEED_MIN_WAVES
#define MQ_SEED_MIN_WAVES 4
#endif
#ifndef MQ_SEED_WAVES
#define MQ_SEED_WAVES 8
#endif
constexpr int SEED_WAVES = MQ_SEED_WAVES;

template <int STOP = 0>
__global__ __launch_bounds__(64 * SEED_WAVES, MQ_SEED_MIN_WAVES) void seed_reads_kernel(const SplitArgs A) {
    __shared__ struct {
        SeedTables T;
        SeedLds SS[SEED_WAVES];
    } W;
    SeedTables &T = W.T;
    SeedLds(&SS)[SEED_WAVES] = W.SS;
    build_seed_tables(T, A.P.l);
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t wv = rdfirst(threadIdx.x >> 6);  // wave-uniform: per-wave bases stay in SGPRs
    SeedLds &S = SS[wv];
    const size_t wave_gid = (size_t)blockIdx.x * SEED_WAVES + wv;
    const DevParams &P = A.P;
    const uint64_t o_base = A.offsets[0];
    uint32_t n_fast = 0, n_general = 0, n_moved = 0;
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(&A.counters[0], 1u);
        r = rdfirst(r);
        if (r >= A.n) break;
        const uint64_t o0 = A.offsets[r];
        const uint64_t len = A.lens ? (uint64_t)A.lens[r] : A.offsets[r + 1] - o0;
        uint32_t cnt = 0;
        uint64_t base = 0;
        if (len >> 32) {
            cnt = LIST_OVERFLOW;
        } else if (len >= (uint64_t)P.l + P.k - 1u) {
            uint32_t cap;
            list_region(A, o0 - o_base, len, r, base, cap);
            APre pre;
            cnt = A.force_general ? SD_NOT_FAST : seed_read_fast<STOP>(A, T, S, A.bases + o0, (uint32_t)len, base, cap, n_moved, pre, false);
            if (cnt == SD_NOT_FAST) n_general++;
            else n_fast++;
        }
        if (lane == 0) {
            A.mz_count[r] = cnt;
            A.mz_base[r] = base;
            if (cnt == SD_NOT_FAST) A.queue[atomicAdd(&A.counters[2], 1u)] = r;
        }
        wave_sync();
    }
    if (lane == 0) {
        if (n_fast) atomicAdd(&A.counters[4], n_fast);
        if (n_general) atomicAdd(&A.counters[5], n_general);
        if (n_moved) atomicAdd(&A.counters[6], n_moved);
    }
}

// the reads queued by seed_reads_kernel, through the general streaming seeder
__global__ __launch_bounds__(64) void seed_general_kernel(const SplitArgs A) {
    __shared__ WaveLds S;
    const uint32_t lane = lane_id();
    const uint32_t nq = A.counters[2];
    const uint64_t o_base = A.offsets[0];
    uint32_t n_moved = 0;
    for (;;) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(&A.counters[3], 1u);
        i = rdfirst(i);
        if (i >= nq) break;
        const uint32_t r = A.queue[i];
        const uint64_t o0 = A.offsets[r];
        const uint64_t len = A.lens ? (uint64_t)A.lens[r] : A.offsets[r + 1] - o0;
        uint64_t base;
        uint32_t cap;
        list_region(A, o0 - o_base, len, r, base, cap);
        const uint32_t cnt = seed_read_general(A, S, A.bases + o0, len, base, cap, n_moved);
        if (lane == 0) {
            A.mz_count[r] = cnt;
            A.mz_base[r] = base;
        }
        wave_sync();
    }
    if (lane == 0 && n_moved) atomicAdd(&A.counters[6], n_moved);
}

#ifndef MQ_ML_MIN_WAVES
#define MQ_ML_MIN_WAVES 5
#endif
constexpr int ML_WAVES = 4;

template <int CH, bool TIMING = false>
__global__ __launch_bounds__(64 * ML_WAVES, MQ_ML_MIN_WAVES) void map_lists_kernel(const SplitArgs A) {
    __shared__ MapListLds SS[ML_WAVES];
    const uint32_t lane = lane_id();
    const uint32_t wv = rdfirst(threadIdx.x >> 6);  // wave-uniform: per-wave bases stay in SGPRs
    const size_t wave_gid = (size_t)blockIdx.x * ML_WAVES + wv;
    MatchRec *scratch = A.scratch_all + wave_gid * A.cap_matches;
    unsigned long long t_steps = 0, t_lookups = 0;
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(&A.counters[1], 1u);
        r = rdfirst(r);
        if (r >= A.n) break;
        const uint64_t len = A.lens ? (uint64_t)A.lens[r] : A.offsets[r + 1] - A.offsets[r];
        mq_hit h;
        map_read<CH, TIMING>(A, SS[wv], scratch, r, len, A.mz_count[r], A.mz_base[r], t_steps, t_lookups, h);
        store_hit(A, r, h);
        wave_sync();
    }
    if (TIMING && lane == 0) {
        atomicAdd(&A.stats64[0], t_steps);
        atomicAdd(&A.stats64[1], t_lookups);
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

// Index::add_with_mer (src/index.rs:100-104) made order independent: the first claimant of a slot stores the entry, every insertion
// bumps the slot's count; finalize (count_kernel) turns "inserted more than once" into the tombstone form end = 0.
// Walks the probe sequence of mq_device.hpp (home slot, other way of the home bucket, following buckets).
__device__ __forceinline__ void table_insert(Bucket *__restrict__ table, uint64_t mask, unsigned long long key, const Entry &e, uint32_t times) {
    const uint64_t nb = (mask + 1) >> 1;
    uint64_t b;
    uint32_t w = 0;
    bool won = false;
    if (key == 0) {
        b = nb;
        won = atomicAdd(&table[b].claims, 1u) == 0;
    } else {
        const uint64_t s0 = key & mask;
        b = s0 >> 1;
        w = (uint32_t)s0 & 1u;
        for (uint32_t step = 0;; ++step) {
            const unsigned long long prev = atomicCAS(&table[b].key[w], 0ull, key);
            if (prev == 0ull) { won = true; break; }
            if (prev == key) break;
            if (step == 0) {
                w ^= 1u;
            } else if (step == 1 || w == 1u) {
                b = b + 1 == nb ? 0 : b + 1;
                w = 0;
            } else {
                w = 1u;
            }
        }
    }
    if (won) table[b].pay[w] = e;
    atomicAdd(&table[b].count[w], times);
}

__global__ void insert_kernel(const RefKmm *__restrict__ kmm, uint64_t n, Bucket *__restrict__ table, uint64_t mask) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const RefKmm r = kmm[i];
        Entry e;
        e.start = r.start;
        e.end = r.end;
        e.offset = r.offset;
        e.id_rc = r.id_rc;
        table_insert(table, mask, r.hash, e, 1u);
    }
}

// One pass over the finished table: Index::get_count (src/index.rs:90-92) = live slots, the number of distinct keys, the largest
// reference id stored -- and the tombstone form: a key inserted more than once gets end = 0 (is_empty, src/index.rs:67-69), so
// that a lookup decides on the 16 payload bytes alone.  acc: [0] live, [1] keys, [2] max ref id + 1 over occupied slots.
__global__ void count_kernel(Bucket *__restrict__ table, uint64_t n_buckets_plus1, unsigned long long *__restrict__ acc) {
    unsigned long long live = 0, keys = 0, max_id1 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n_buckets_plus1; i += (uint64_t)gridDim.x * blockDim.x) {
        Bucket &B = table[i >> 1];
        const uint32_t w = (uint32_t)i & 1u;
        const uint32_t cnt = B.count[w];
        if (cnt != 0) {
            keys++;
            const Entry e = B.pay[w];
            const unsigned long long id1 = (unsigned long long)(e.id_rc >> 1) + 1ull;
            max_id1 = id1 > max_id1 ? id1 : max_id1;
            if (cnt == 1 && e.end != 0) live++;
            else if (e.end != 0) B.pay[w].end = 0;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        live += __shfl_xor(live, d, 64);
        keys += __shfl_xor(keys, d, 64);
        const unsigned long long o = __shfl_xor(max_id1, d, 64);
        max_id1 = o > max_id1 ? o : max_id1;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], live);
        atomicAdd(&acc[1], keys);
        atomicMax(&acc[2], max_id1);
    }
}

// On-disk form (mq_index_save / mq_index_load): the occupied slots only, 32 bytes each, in no particular order.
struct alignas(32) SavedSlot {
    Entry e;
    unsigned long long key;
    uint32_t count;
    uint32_t is_key0;  // 1: the entry of the key 0 (the extra bucket)
};
static_assert(sizeof(SavedSlot) == 32, "saved slot size");
__global__ void pack_slots_kernel(const Bucket *__restrict__ table, uint64_t n_buckets_plus1, SavedSlot *__restrict__ out,
                                  unsigned long long *__restrict__ cursor, uint64_t cap) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n_buckets_plus1; i += (uint64_t)gridDim.x * blockDim.x) {
        const Bucket &B = table[i >> 1];
        const uint32_t w = (uint32_t)i & 1u;
        const uint32_t cnt = B.count[w];
        if (cnt != 0) {
            const unsigned long long at = atomicAdd(cursor, 1ull);
            if (at < cap) {
                SavedSlot v;
                v.key = B.key[w];
                v.e = B.pay[w];
                v.count = cnt;
                v.is_key0 = (i >> 1) == n_buckets_plus1 - 1 ? 1u : 0u;
                out[at] = v;
            }
        }
    }
}
// mq_index_load: saved slots back into an empty table (keys are distinct, so every insertion claims its slot); flags[0] is set
// when an entry cannot be what mq_index_save wrote (a reference id beyond the file's reference table, a key 0 outside its slot).
__global__ void unpack_slots_kernel(const SavedSlot *__restrict__ in, uint64_t n, Bucket *__restrict__ table, uint64_t mask, uint32_t max_id,
                                    uint32_t *__restrict__ flags) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const SavedSlot v = in[i];
        if (v.count == 0 || (v.e.id_rc >> 1) > max_id || (v.is_key0 != 0) != (v.key == 0) || v.is_key0 > 1u) {
            atomicOr(flags, 1u);
            continue;
        }
        table_insert(table, mask, v.key, v.e, v.count);
    }
}

__global__ void lookup_kernel(const Bucket *__restrict__ table, uint64_t mask, const uint64_t *__restrict__ keys, uint32_t n,
                              uint8_t *__restrict__ found, mq_kminmer *__restrict__ entries, uint32_t *__restrict__ ref_ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Entry e = {};
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

// Diagnostic (tools/probe_rate.py): how many random index probes per second the memory system sustains, detached from
// everything else the map path does.  Every thread looks up `per_thread` pseudo-random keys (absent with probability ~1,
// like ~85 % of a read's k-min-mers), `ilp` home-slot loads in flight per thread.
__global__ void probe_rate_kernel(const Bucket *__restrict__ table, uint64_t mask, uint32_t per_thread, uint64_t seed,
                                  unsigned long long *__restrict__ acc, const uint32_t *__restrict__ bitmap, uint64_t bit_mask,
                                  uint32_t table_too) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nb = (mask + 1) >> 1;
    unsigned long long found = 0, steps = 0;
    auto mix = [](uint64_t z) {
        z += 0x9e3779b97f4a7c15ULL;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    };
    for (uint32_t j = 0; j < per_thread; j += 4) {
        uint64_t key[4];
        uint4 kk[4];
        uint32_t bw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            key[u] = mix(seed + tid * per_thread + j + u) | 1ull;
            if (bitmap) bw[u] = bitmap[(key[u] & bit_mask) >> 5];
            else kk[u] = ld_u4(&table[(key[u] & mask) >> 1].key[0]);
        }
        if (bitmap) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool maybe = (bw[u] >> (key[u] & 31u)) & 1u;
                found += maybe;
                kk[u] = (maybe && table_too) ? ld_u4(&table[(key[u] & mask) >> 1].key[0]) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // the probe sequence of mq_device.hpp: home way, other way, then the following buckets
            const uint32_t w0 = (uint32_t)(key[u] & mask) & 1u;
            uint64_t ka = u64_of(kk[u].x, kk[u].y), kb = u64_of(kk[u].z, kk[u].w);
            uint64_t kh = w0 ? kb : ka, kp = w0 ? ka : kb;
            uint64_t b = (key[u] & mask) >> 1;
            bool hit = kh == key[u], go = !hit && kh != 0;
            if (go) {
                steps++;
                hit = kp == key[u];
                go = !hit && kp != 0;
            }
            while (go) {
                b = b + 1 == nb ? 0 : b + 1;
                const uint4 v = ld_u4(&table[b].key[0]);
                ka = u64_of(v.x, v.y);
                kb = u64_of(v.z, v.w);
                steps++;
                hit = ka == key[u];
                go = !hit && ka != 0;
                if (go) {
                    steps++;
                    hit = kb == key[u];
                    go = !hit && kb != 0;
                }
            }
            if (hit) found += table[b].count[0];
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        found += __shfl_xor(found, d, 64);
        steps += __shfl_xor(steps, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], found);
        atomicAdd(&acc[1], steps);
    }
}

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
    unsigned long long *mz_hash = nullptr;
    uint32_t *mz_pos = nullptr;
    uint64_t mz_cap = 0;            // list entries allocated
    uint32_t *mz_count = nullptr;
    uint64_t *mz_base = nullptr;
    uint32_t *queue = nullptr;
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
    Bucket *table = nullptr;  // nslots / 2 buckets + the extra bucket of the key 0
    uint64_t nslots = 0;
    uint64_t *d_ref_lens = nullptr;
    uint64_t n_unique = 0, n_keys = 0;
    // grow-only scratch of mq_index_add_ref (freed by finalize): no allocation per reference once it has grown
    uint8_t *bld_seq = nullptr;
    uint64_t bld_seq_cap = 0;
    Minimizer *bld_seg_lists = nullptr, *bld_dense = nullptr;
    uint64_t bld_seg_lists_cap = 0, bld_dense_cap = 0;
    uint32_t *bld_counts = nullptr;
    uint64_t bld_counts_cap = 0;
    uint64_t *bld_seg_off = nullptr;
    uint64_t bld_seg_off_cap = 0;
    // launch geometry (workgroups) and scratch sizes, fixed at the first map call
    uint32_t grid_fused = 0, grid_seed = 0, grid_map = 0;  // map_kernel; seed_reads_kernel, map_lists_kernel (split)
    uint32_t cap_matches = 0;
    bool split = false;             // diagnostic MQ_PIPELINE=split: the two phases as separate launches (a profiler then prices each)
    bool force_general = false;     // test hook MQ_FORCE_GENERAL=1: never take the fast seeding path
    int chain_chunk = 64;           // test hook: MQ_CHAIN_CHUNK=4 exercises the multi-chunk chain path
    mq_ctx *def_ctx = nullptr;      // the context behind the index-level map entry points
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
    p->flags = 0;
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

static size_t table_bytes_of(uint64_t nslots) { return (size_t)(nslots / 2 + 1) * sizeof(Bucket); }

static int alloc_table(mq_index *idx, uint64_t nslots) {
    if (idx->table) {
        HIPCHK(hipFree(idx->table));
        idx->table = nullptr;
    }
    if (nslots < 2) nslots = 2;  // whole buckets
    HIPCHK(hipMalloc((void **)&idx->table, table_bytes_of(nslots)));
    HIPCHK(hipMemset(idx->table, 0, table_bytes_of(nslots)));
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
    if ((rc = occ_of((const void *)map_kernel<64, false>, 64 * MAP_WAVES, occ))) return rc;
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
    if (!c->scratch) {
        const size_t n_waves = std::max((size_t)idx->grid_fused * MAP_WAVES, (size_t)idx->grid_map * ML_WAVES);
        HIPCHK(hipMalloc((void **)&c->scratch, n_waves * idx->cap_matches * sizeof(MatchRec)));
    }
    if (n > c->reads_cap) {
        if (c->mz_count) HIPCHK(hipFree(c->mz_count));
        if (c->mz_base) HIPCHK(hipFree(c->mz_base));
        if (c->queue) HIPCHK(hipFree(c->queue));
        c->mz_count = c->queue = nullptr;
        c->mz_base = nullptr;
        c->reads_cap = 0;
        const uint64_t nc = (uint64_t)n + n / 4 + 64;
        HIPCHK(hipMalloc((void **)&c->mz_count, nc * 4));
        HIPCHK(hipMalloc((void **)&c->mz_base, nc * 8));
        HIPCHK(hipMalloc((void **)&c->queue, nc * 4));
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
        c->mz_hash = nullptr;
        c->mz_pos = nullptr;
        c->mz_cap = 0;
        const uint64_t nc = need + need / 8;
        HIPCHK(hipMalloc((void **)&c->mz_hash, nc * 8));
        HIPCHK(hipMalloc((void **)&c->mz_pos, nc * 4));
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
    hipFree(c->mz_count);
    hipFree(c->mz_base);
    hipFree(c->queue);
    hipFree(c->st_bases);
    hipFree(c->st_off);
    hipFree(c->st_out);
    hipFree(c->st_lens);
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

static void free_build_scratch(mq_index *idx) {
    hipFree(idx->bld_seq);
    hipFree(idx->bld_seg_lists);
    hipFree(idx->bld_dense);
    hipFree(idx->bld_counts);
    hipFree(idx->bld_seg_off);
    idx->bld_seq = nullptr;
    idx->bld_seg_lists = idx->bld_dense = nullptr;
    idx->bld_counts = nullptr;
    idx->bld_seg_off = nullptr;
    idx->bld_seq_cap = idx->bld_seg_lists_cap = idx->bld_dense_cap = idx->bld_counts_cap = idx->bld_seg_off_cap = 0;
}

extern "C" {

mq_index *mq_index_new(const mq_params *params, int device) try {
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
    idx->dp.fold = (params->flags & MQ_FLAG_FOLD_CASE) ? 1u : 0u;
    const char *cc = getenv("MQ_CHAIN_CHUNK");
    if (cc && atoi(cc) == 4) idx->chain_chunk = 4;
    const char *fg = getenv("MQ_FORCE_GENERAL");
    idx->force_general = fg && atoi(fg) != 0;
    const char *pl = getenv("MQ_PIPELINE");
    idx->split = pl && strcmp(pl, "split") == 0;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        set_err(MQ_EHIP, "hipSetDevice/hipGetDeviceProperties failed");
        delete idx;
        return nullptr;
    }
    idx->n_cu = prop.multiProcessorCount;
    // an empty one-bucket table so that seeding-only calls work before finalize
    if (alloc_table(idx, 2) != MQ_OK) {
        delete idx;
        return nullptr;
    }
    idx->def_ctx = ctx_create(idx);
    if (!idx->def_ctx) {
        hipFree(idx->table);
        delete idx;
        return nullptr;
    }
    return idx;
} catch (const std::bad_alloc &) {
    set_err(MQ_ENOMEM, "out of host memory");
    return nullptr;
} catch (const std::exception &e) {
    set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
    return nullptr;
}

void mq_index_free(mq_index *idx) {
    if (!idx) return;
    hipSetDevice(idx->device);
    for (auto &c : idx->chunks)
        if (c.d) hipFree(c.d);
    free_build_scratch(idx);
    if (idx->table) hipFree(idx->table);
    if (idx->d_ref_lens) hipFree(idx->d_ref_lens);
    ctx_release(idx->def_ctx);
    delete idx;
}

static int64_t add_ref_device_locked(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *d_seq, uint64_t len) {
    if (!idx || (!d_seq && len)) return set_err(MQ_EINVAL, "bad arguments");
    if (idx->finalized) return set_err(MQ_ESTATE, "index already finalized");
    if (len >= (1ull << 32)) return set_err(MQ_EINVAL, "sequence length must be < 2^32");
    if (ref_id >= MQ_MAX_REF_ID) return set_err(MQ_EINVAL, "ref_id must be < 2^24 (reference lengths are kept in a dense device array)");
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
    std::vector<uint32_t> counts(n_seg);
    std::vector<uint64_t> seg_off(n_seg + 1);
    if ((rc = grow(idx->bld_counts, idx->bld_counts_cap, n_seg))) return rc;
    const uint32_t grid = std::min<uint32_t>(n_seg, (uint32_t)idx->n_cu * 32u);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = grow(idx->bld_seg_lists, idx->bld_seg_lists_cap, (uint64_t)n_seg * cap))) return rc;
        hipLaunchKernelGGL(seed_segments_kernel, dim3(grid), dim3(64), 0, 0, d_seq, len, seg_len, n_seg, P, idx->bld_seg_lists, cap, idx->bld_counts);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(counts.data(), idx->bld_counts, (size_t)n_seg * sizeof(uint32_t), hipMemcpyDeviceToHost));
        bool overflow = false;
        for (uint32_t s = 0; s < n_seg; ++s) overflow |= counts[s] > cap;
        if (!overflow) break;
        if (attempt == 1) return set_err(MQ_EOVERFLOW, "minimizer list overflow at worst-case capacity (internal error)");
        cap = (uint32_t)seg_len;  // a segment cannot hold more run heads than bases
    }
    seg_off[0] = 0;
    for (uint32_t s = 0; s < n_seg; ++s) seg_off[s + 1] = seg_off[s] + counts[s];
    const uint64_t n_mz = seg_off[n_seg];
    int64_t n_kmm = 0;
    if (n_mz >= P.k) {
        n_kmm = (int64_t)(n_mz - P.k + 1);
        if ((rc = grow(idx->bld_seg_off, idx->bld_seg_off_cap, (uint64_t)n_seg + 1))) return rc;
        if ((rc = grow(idx->bld_dense, idx->bld_dense_cap, n_mz))) return rc;
        HIPCHK(hipMemcpy(idx->bld_seg_off, seg_off.data(), (size_t)(n_seg + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(compact_minimizers_kernel, dim3(std::min<uint32_t>(n_seg, 65535u)), dim3(64), 0, 0, idx->bld_seg_lists, cap,
                           idx->bld_counts, idx->bld_seg_off, n_seg, idx->bld_dense);
        HIPCHK(hipGetLastError());
        // the reference's k-min-mers go behind those of the previous references in the current chunk while it has room
        if (idx->chunks.empty() || idx->chunks.back().n + (uint64_t)n_kmm > idx->chunks.back().cap) {
            KmmChunk ch;
            ch.cap = std::max<uint64_t>((uint64_t)n_kmm, 4ull << 20);
            HIPCHK(hipMalloc((void **)&ch.d, (size_t)ch.cap * sizeof(RefKmm)));
            idx->chunks.push_back(ch);
        }
        KmmChunk &ch = idx->chunks.back();
        const uint32_t kb = (uint32_t)std::min<uint64_t>(((uint64_t)n_kmm + 255) / 256, 65535ull);
        hipLaunchKernelGGL(ref_kminmers_kernel, dim3(kb), dim3(256), 0, 0, idx->bld_dense, n_mz, P, ref_id, ch.d + ch.n);
        HIPCHK(hipGetLastError());
        ch.n += (uint64_t)n_kmm;
        idx->n_kmm_total += (uint64_t)n_kmm;
    }
    return n_kmm;  // everything above runs on the null stream: the next call's kernels (and finalize) are ordered behind it
}

int64_t mq_index_add_ref_device(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *d_seq, uint64_t len) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    return add_ref_device_locked(idx, ref_id, name, d_seq, len);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int64_t mq_index_add_ref(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *seq, uint64_t len) try {
    if (!idx || (!seq && len)) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    if (len >= (1ull << 32)) return set_err(MQ_EINVAL, "sequence length must be < 2^32");
    if ((rc = grow(idx->bld_seq, idx->bld_seq_cap, len + 64))) return rc;
    if (len) HIPCHK(hipMemcpy(idx->bld_seq, seq, len, hipMemcpyHostToDevice));
    return add_ref_device_locked(idx, ref_id, name, idx->bld_seq, len);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int64_t mq_index_finalize(mq_index *idx) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->finalized) return (int64_t)idx->n_unique;
    int rc = use_device(idx);
    if (rc) return rc;
    uint64_t nslots = 1024;
    // slots per inserted k-min-mer (power-of-two rounding on top); default 8 => load <= 0.125 (17 GB for a human genome, 6 % of
    // the HBM).  ~85 % of a read's lookups miss, a miss walks to the first empty slot, and every extra step is one more dependent
    // random access of a memory system that sustains ~52 G of them per second (tools/probe_rate.py).  Measured on the CHM13-like
    // bench: factor 2: 926, 4: 1000, 8: 1034, 16: 1044, 32: 1051 Gbases/s.
    const char *lf = getenv("MQ_TABLE_FACTOR");
    const uint64_t factor = lf && atoi(lf) >= 2 ? (uint64_t)atoi(lf) : 8ull;  // >= 2: a full table would make a miss walk forever
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
    HIPCHK(hipMalloc((void **)&d_acc, 24));
    HIPCHK(hipMemset(d_acc, 0, 24));
    const uint32_t nb = (uint32_t)std::min<uint64_t>((nslots + 2 + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(count_kernel, dim3(nb), dim3(256), 0, 0, idx->table, nslots / 2 + 1, d_acc);
    HIPCHK(hipGetLastError());
    unsigned long long acc[3] = {0, 0, 0};
    HIPCHK(hipMemcpy(acc, d_acc, 24, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d_acc));
    idx->n_unique = acc[0];
    idx->n_keys = acc[1];
    for (auto &c : idx->chunks)
        if (c.d) hipFree(c.d);
    idx->chunks.clear();
    free_build_scratch(idx);
    // ref_map lengths (src/closures.rs:49), dense by ref id
    uint32_t max_id = 0;
    for (auto &kv : idx->refs) max_id = std::max(max_id, kv.first);
    std::vector<uint64_t> lens((size_t)max_id + 1, 0);
    for (auto &kv : idx->refs) lens[kv.first] = kv.second.second;
    HIPCHK(hipMalloc((void **)&idx->d_ref_lens, lens.size() * sizeof(uint64_t)));
    HIPCHK(hipMemcpy(idx->d_ref_lens, lens.data(), lens.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    idx->finalized = true;
    return (int64_t)idx->n_unique;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_index_get_stats(const mq_index *idx, mq_index_stats *out) try {
    if (!idx || !out) return set_err(MQ_EINVAL, "bad arguments");
    out->n_refs = idx->refs.size();
    out->n_kminmers = idx->n_kmm_total;
    out->n_keys = idx->n_keys;
    out->n_unique = idx->n_unique;
    out->table_slots = idx->nslots;
    out->table_bytes = table_bytes_of(idx->nslots);
    out->slot_bytes = SLOT_BYTES;
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// On-disk index (the reference has none and rebuilds on every run, src/closures.rs:24-94): header, parameters, reference table,
// then the OCCUPIED slots only (32 bytes each: ~1.5 GB for a human genome instead of the 17 GB table at load 1/8); mq_index_load
// scatters them into a fresh table on the device.  Little-endian, this library's layout (MQ_INDEX_MAGIC names the version).
static const char MQ_INDEX_MAGIC[8] = {'M', 'Q', 'H', 'I', 'P', 'I', 'X', '2'};
constexpr size_t IX_IO_CHUNK = 64u << 20;  // bytes per page-locked transfer buffer (two of them: the copy overlaps the file I/O)

static bool write_all(int fd, const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    while (n) {
        const ssize_t w = ::write(fd, b, n);
        if (w <= 0) return false;
        b += w;
        n -= (size_t)w;
    }
    return true;
}
static bool read_all(int fd, void *p, size_t n) {
    uint8_t *b = (uint8_t *)p;
    while (n) {
        const ssize_t r = ::read(fd, b, n);
        if (r <= 0) return false;
        b += r;
        n -= (size_t)r;
    }
    return true;
}

int mq_index_save(const mq_index *idx, const char *path) try {
    if (!idx || !path) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    int rc = use_device(idx);
    if (rc) return rc;
    // occupied slots, packed on the device
    const uint64_t n_occ = idx->n_keys;
    SavedSlot *d_pack = nullptr;
    unsigned long long *d_cur = nullptr;
    uint8_t *h_buf[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int fd = -1;
    auto cleanup = [&]() {
        hipFree(d_pack);
        hipFree(d_cur);
        for (int i = 0; i < 2; ++i) {
            if (h_buf[i]) hipHostFree(h_buf[i]);
            if (ev[i]) hipEventDestroy(ev[i]);
        }
        if (st) hipStreamDestroy(st);
        if (fd >= 0) ::close(fd);
    };
    auto fail = [&](int code, const std::string &msg) {
        cleanup();
        return set_err(code, msg);
    };
    if (hipMalloc((void **)&d_pack, (size_t)(n_occ + 1) * sizeof(SavedSlot)) != hipSuccess || hipMalloc((void **)&d_cur, 8) != hipSuccess ||
        hipMemset(d_cur, 0, 8) != hipSuccess)
        return fail(MQ_ENOMEM, "mq_index_save: no device memory for the packed slots");
    const uint64_t nb1 = idx->nslots / 2 + 1;
    hipLaunchKernelGGL(pack_slots_kernel, dim3((uint32_t)std::min<uint64_t>((2 * nb1 + 255) / 256, 1u << 16)), dim3(256), 0, 0, idx->table, nb1, d_pack,
                       d_cur, n_occ);
    unsigned long long packed = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpy(&packed, d_cur, 8, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(MQ_EHIP, "mq_index_save: packing the table failed");
    if (packed != n_occ) return fail(MQ_ESTATE, "mq_index_save: the table holds another number of keys than the index records (internal error)");
    fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(MQ_EINVAL, std::string("cannot open for writing: ") + path);
    bool ok = write_all(fd, MQ_INDEX_MAGIC, 8);
    const uint64_t hdr[6] = {sizeof(SavedSlot), idx->nslots, idx->n_kmm_total, idx->n_keys, idx->n_unique, (uint64_t)idx->refs.size()};
    ok = ok && write_all(fd, &idx->params, sizeof(mq_params)) && write_all(fd, hdr, sizeof(hdr));
    for (auto &kv : idx->refs) {
        const uint32_t id = kv.first, nl = (uint32_t)kv.second.first.size();
        ok = ok && write_all(fd, &id, 4) && write_all(fd, &nl, 4) && write_all(fd, &kv.second.second, 8) && (nl == 0 || write_all(fd, kv.second.first.data(), nl));
    }
    const size_t total = (size_t)n_occ * sizeof(SavedSlot);
    if (ok && total) {
        bool hip_ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 2 && hip_ok; ++i)
            hip_ok = hipHostMalloc((void **)&h_buf[i], std::min(total, IX_IO_CHUNK), hipHostMallocDefault) == hipSuccess && hipEventCreate(&ev[i]) == hipSuccess;
        if (!hip_ok) return fail(MQ_EHIP, "mq_index_save: transfer buffers");
        // chunk i+1 crosses PCIe while chunk i goes to the file
        const size_t n_chunks = (total + IX_IO_CHUNK - 1) / IX_IO_CHUNK;
        auto issue = [&](size_t c) {
            const size_t o = c * IX_IO_CHUNK, n = std::min(IX_IO_CHUNK, total - o);
            return hipMemcpyAsync(h_buf[c & 1], (const uint8_t *)d_pack + o, n, hipMemcpyDeviceToHost, st) == hipSuccess &&
                   hipEventRecord(ev[c & 1], st) == hipSuccess;
        };
        hip_ok = issue(0);
        for (size_t c = 0; c < n_chunks && ok && hip_ok; ++c) {
            if (c + 1 < n_chunks) hip_ok = issue(c + 1);
            hip_ok = hip_ok && hipEventSynchronize(ev[c & 1]) == hipSuccess;
            const size_t o = c * IX_IO_CHUNK, n = std::min(IX_IO_CHUNK, total - o);
            ok = hip_ok && write_all(fd, h_buf[c & 1], n);
        }
        if (!hip_ok) return fail(MQ_EHIP, "mq_index_save: device-to-host copy failed");
    }
    const bool closed = ::close(fd) == 0;
    fd = -1;
    cleanup();
    return ok && closed ? MQ_OK : set_err(MQ_EINVAL, std::string("short write: ") + path);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

mq_index *mq_index_load(const char *path, int device) try {
    if (!path) {
        set_err(MQ_EINVAL, "path is NULL");
        return nullptr;
    }
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        set_err(MQ_EINVAL, std::string("cannot open: ") + path);
        return nullptr;
    }
    char magic[8];
    mq_params p;
    uint64_t hdr[6];
    if (!read_all(fd, magic, 8) || memcmp(magic, MQ_INDEX_MAGIC, 8) != 0 || !read_all(fd, &p, sizeof(p)) || !read_all(fd, hdr, sizeof(hdr)) ||
        hdr[0] != sizeof(SavedSlot) || hdr[1] < 2 || (hdr[1] & (hdr[1] - 1)) != 0 || hdr[1] > (1ull << 40) ||
        hdr[3] >= hdr[1] /* a table without an empty slot would make a miss walk forever */ || hdr[4] > hdr[3] || hdr[5] > MQ_MAX_REF_ID) {
        ::close(fd);
        set_err(MQ_EINVAL, std::string("not a mapquik HIP index (or another layout version): ") + path);
        return nullptr;
    }
    mq_index *idx = mq_index_new(&p, device);
    if (!idx) {
        ::close(fd);
        return nullptr;
    }
    bool ok = true;
    for (uint64_t i = 0; ok && i < hdr[5]; ++i) {
        uint32_t id = 0, nl = 0;
        uint64_t len = 0;
        ok = read_all(fd, &id, 4) && read_all(fd, &nl, 4) && read_all(fd, &len, 8) && nl < (1u << 20) && id < MQ_MAX_REF_ID;
        std::string name(nl, '\0');
        ok = ok && (nl == 0 || read_all(fd, &name[0], nl));
        if (ok) idx->refs[id] = std::make_pair(name, len);
    }
    uint32_t max_id = 0;
    for (auto &kv : idx->refs) max_id = std::max(max_id, kv.first);
    if (ok && alloc_table(idx, hdr[1]) != MQ_OK) ok = false;
    // file -> page-locked buffer -> device -> scatter kernel, by a few threads at once (each its own buffers and stream; the
    // kernels of different chunks insert into the same table with atomics): the file read, not the copy, is what takes time
    const size_t total = (size_t)hdr[3] * sizeof(SavedSlot);
    const off_t slots_at = ::lseek(fd, 0, SEEK_CUR);
    uint32_t *d_flags = nullptr;
    const char *why = "truncated or unreadable index file: ";
    if (ok && total) {
        ok = slots_at >= 0 && hipMalloc((void **)&d_flags, 4) == hipSuccess && hipMemset(d_flags, 0, 4) == hipSuccess &&
             hipDeviceSynchronize() == hipSuccess;  // the table's memset (null stream) is done before other streams write to it
        const size_t n_chunks = (total + IX_IO_CHUNK - 1) / IX_IO_CHUNK;
        const int n_thr = (int)std::min<size_t>(8, n_chunks);
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        auto work = [&]() {
            uint8_t *h = nullptr, *d = nullptr;
            hipStream_t st = nullptr;
            const size_t cb = std::min(total, IX_IO_CHUNK);
            bool good = hipSetDevice(device) == hipSuccess && (h = (uint8_t *)mq_host_alloc(cb)) != nullptr && hipMalloc((void **)&d, cb) == hipSuccess &&
                        hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
            while (good) {
                const size_t c = next.fetch_add(1);
                if (c >= n_chunks) break;
                const size_t o = c * IX_IO_CHUNK, n = std::min(IX_IO_CHUNK, total - o);
                size_t got = 0;
                while (got < n) {
                    const ssize_t r = ::pread(fd, h + got, n - got, slots_at + (off_t)(o + got));
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                if (got != n) { good = false; break; }
                good = hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st) == hipSuccess;
                const uint64_t ns = n / sizeof(SavedSlot);
                hipLaunchKernelGGL(unpack_slots_kernel, dim3((uint32_t)std::min<uint64_t>((ns + 255) / 256, 1u << 16)), dim3(256), 0, st,
                                   (const SavedSlot *)d, ns, idx->table, hdr[1] - 1, max_id, d_flags);
                good = good && hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            }
            if (!good) bad.store(1);
            if (st) hipStreamDestroy(st);
            hipFree(d);
            mq_host_free(h);
        };
        if (ok) {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; ++t) th.emplace_back(work);
            for (auto &t : th) t.join();
            ok = bad.load() == 0;
        }
        uint32_t flags = 1;
        ok = ok && hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok && flags) {
            ok = false;
            why = "corrupt index file (an entry names a reference the file does not have, or a malformed slot): ";
        }
        if (ok) ok = ::lseek(fd, slots_at + (off_t)total, SEEK_SET) >= 0;
    }
    uint8_t extra = 0;
    if (ok && ::read(fd, &extra, 1) != 0) {
        ok = false;
        why = "corrupt index file (bytes after the last slot): ";
    }
    ::close(fd);
    // what the file says about its table must be what the rebuilt table holds
    if (ok) {
        unsigned long long *d_acc = nullptr, acc[3] = {0, 0, 0};
        ok = hipMalloc((void **)&d_acc, 24) == hipSuccess && hipMemset(d_acc, 0, 24) == hipSuccess;
        if (ok) {
            const uint64_t nb1 = hdr[1] / 2 + 1;
            hipLaunchKernelGGL(count_kernel, dim3((uint32_t)std::min<uint64_t>((2 * nb1 + 255) / 256, 1u << 16)), dim3(256), 0, 0, idx->table, nb1, d_acc);
            ok = hipGetLastError() == hipSuccess && hipMemcpy(acc, d_acc, 24, hipMemcpyDeviceToHost) == hipSuccess;
        }
        hipFree(d_acc);
        if (ok && (acc[1] != hdr[3] || acc[0] != hdr[4] || (acc[2] != 0 && acc[2] - 1 > max_id))) {
            ok = false;
            why = "corrupt index file (key counts or reference ids disagree with its header): ";
        }
    }
    if (ok) {
        std::vector<uint64_t> lens((size_t)max_id + 1, 0);
        for (auto &kv : idx->refs) lens[kv.first] = kv.second.second;
        ok = hipMalloc((void **)&idx->d_ref_lens, lens.size() * sizeof(uint64_t)) == hipSuccess &&
             hipMemcpy(idx->d_ref_lens, lens.data(), lens.size() * sizeof(uint64_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    hipFree(d_flags);
    if (!ok) {
        mq_index_free(idx);
        set_err(MQ_EINVAL, std::string(why) + path);
        return nullptr;
    }
    idx->n_kmm_total = hdr[2];
    idx->n_keys = hdr[3];
    idx->n_unique = hdr[4];
    idx->finalized = true;
    return idx;
} catch (const std::bad_alloc &) {
    set_err(MQ_ENOMEM, "out of host memory");
    return nullptr;
} catch (const std::exception &e) {
    set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
    return nullptr;
}

// A replica of a finalized index on another device: the table travels device to device (xGMI between the GPUs of a node)
// instead of being rebuilt from the reference on every GPU.
mq_index *mq_index_clone(const mq_index *src, int device) try {
    if (!src) {
        set_err(MQ_EINVAL, "src is NULL");
        return nullptr;
    }
    if (!src->finalized) {
        set_err(MQ_ESTATE, "index not finalized");
        return nullptr;
    }
    mq_index *idx = mq_index_new(&src->params, device);
    if (!idx) return nullptr;
    idx->refs = src->refs;
    idx->n_kmm_total = src->n_kmm_total;
    idx->n_keys = src->n_keys;
    idx->n_unique = src->n_unique;
    bool ok = alloc_table(idx, src->nslots) == MQ_OK;
    if (ok) ok = hipMemcpyPeer(idx->table, device, src->table, src->device, table_bytes_of(src->nslots)) == hipSuccess;
    uint32_t max_id = 0;
    for (auto &kv : idx->refs) max_id = std::max(max_id, kv.first);
    const size_t nl = (size_t)max_id + 1;
    if (ok) ok = hipSetDevice(device) == hipSuccess && hipMalloc((void **)&idx->d_ref_lens, nl * sizeof(uint64_t)) == hipSuccess;
    if (ok) ok = hipMemcpyPeer(idx->d_ref_lens, device, src->d_ref_lens, src->device, nl * sizeof(uint64_t)) == hipSuccess;
    if (ok) ok = hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        mq_index_free(idx);
        set_err(MQ_EHIP, "mq_index_clone: device-to-device copy failed");
        return nullptr;
    }
    idx->finalized = true;
    return idx;
} catch (const std::bad_alloc &) {
    set_err(MQ_ENOMEM, "out of host memory");
    return nullptr;
} catch (const std::exception &e) {
    set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
    return nullptr;
}

int mq_index_ref_info(const mq_index *idx, uint32_t ref_id, const char **name, uint64_t *len) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    auto it = idx->refs.find(ref_id);
    if (it == idx->refs.end()) return set_err(MQ_EINVAL, "unknown ref_id");
    if (name) *name = it->second.first.c_str();
    if (len) *len = it->second.second;
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_reserve(mq_index *idx, uint32_t n_reads, uint64_t total_bases) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    return ctx_ensure(idx->def_ctx, n_reads, total_bases, list_f16(idx));
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

}  // extern "C"

struct LaunchOpt {
    mq_kminmer *d_dump = nullptr;
    const uint64_t *d_dump_off = nullptr;
    uint32_t *d_dump_counts = nullptr;
    MatchRec *scratch_override = nullptr;  // overflow redo: worst-case Match scratch on a small grid
    uint32_t cap_override = 0;
    uint32_t grid_override = 0;
    uint32_t f16 = 0;                      // 0 => list_f16(idx)
    const uint32_t *d_lens = nullptr;      // spans form: per-read lengths
    bool instrumented = false;             // mq_map_probe_stats: the launch that counts lookups and probe steps (slower, never timed)
};

// One launch sequence on stream `st` using the context's scratch.  ctx_ensure(c, n, total_bases, f16) must have succeeded.
static int launch_map(mq_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, mq_hit *d_out, hipStream_t st,
                      const LaunchOpt &o = LaunchOpt()) {
    mq_index *idx = c->idx;
    if (n == 0) return MQ_OK;
    HIPCHK(hipMemsetAsync(c->d_counter, 0, 256, st));
    HIPCHK(hipEventRecord(c->ev0, st));
    SplitArgs A;
    A.bases = d_bases;
    A.offsets = d_offsets;
    A.lens = o.d_lens;
    A.n = n;
    A.P = idx->dp;
    A.mz_hash = c->mz_hash;
    A.mz_pos = c->mz_pos;
    A.mz_count = c->mz_count;
    A.mz_base = c->mz_base;
    A.pool_base = c->pool_base;
    A.pool_cap = c->pool_cap;
    A.f16 = o.f16 ? o.f16 : list_f16(idx);
    A.slack = LIST_SLACK;
    A.queue = c->queue;
    A.counters = c->d_counter;
    A.force_general = idx->force_general ? 1u : 0u;
    A.table = idx->table;
    A.mask = idx->nslots - 1;
    A.ref_lens = idx->d_ref_lens;
    A.scratch_all = o.scratch_override ? o.scratch_override : c->scratch;
    A.cap_matches = o.scratch_override ? o.cap_override : idx->cap_matches;
    A.out = d_out;
    A.dump = o.d_dump;
    A.dump_off = o.d_dump_off;
    A.dump_counts = o.d_dump_counts;
    A.stats64 = reinterpret_cast<unsigned long long *>(c->d_counter + 8);
    if (!idx->split) {
        uint32_t grid = std::min<uint32_t>(idx->grid_fused, (n + MAP_WAVES - 1) / MAP_WAVES);
        if (o.grid_override) grid = std::min(grid, o.grid_override);
        const dim3 blk(64 * MAP_WAVES);
        if (o.instrumented) hipLaunchKernelGGL((map_kernel<64, true>), dim3(grid), blk, 0, st, A);
        else if (idx->chain_chunk == 4) hipLaunchKernelGGL((map_kernel<4, false>), dim3(grid), blk, 0, st, A);
        else hipLaunchKernelGGL((map_kernel<64, false>), dim3(grid), blk, 0, st, A);
        HIPCHK(hipGetLastError());
    } else {
        const uint32_t gs = std::min<uint32_t>(idx->grid_seed, (n + SEED_WAVES - 1) / SEED_WAVES);
        const char *ss = getenv("MQ_SEED_STOP");  // diagnostic: stage attribution by truncation (results are NOT valid)
        const int stop = ss ? atoi(ss) : 0;
        if (stop == 1) hipLaunchKernelGGL(seed_reads_kernel<1>, dim3(gs), dim3(64 * SEED_WAVES), 0, st, A);
        else if (stop == 2) hipLaunchKernelGGL(seed_reads_kernel<2>, dim3(gs), dim3(64 * SEED_WAVES), 0, st, A);
        else hipLaunchKernelGGL(seed_reads_kernel<0>, dim3(gs), dim3(64 * SEED_WAVES), 0, st, A);
        HIPCHK(hipGetLastError());
        // the reads the fast seeder declined: the queue length lives on the device, so the grid is fixed and waves that find
        // the queue empty leave at once
        const uint32_t gg = std::min<uint32_t>((uint32_t)idx->n_cu * 8u, n);
        hipLaunchKernelGGL(seed_general_kernel, dim3(gg), dim3(64), 0, st, A);
        HIPCHK(hipGetLastError());
        uint32_t gm = std::min<uint32_t>(idx->grid_map, (n + ML_WAVES - 1) / ML_WAVES);
        if (o.grid_override) gm = std::min(gm, o.grid_override);
        const dim3 blk(64 * ML_WAVES);
        if (o.instrumented) hipLaunchKernelGGL((map_lists_kernel<64, true>), dim3(gm), blk, 0, st, A);
        else if (idx->chain_chunk == 4) hipLaunchKernelGGL((map_lists_kernel<4, false>), dim3(gm), blk, 0, st, A);
        else hipLaunchKernelGGL((map_lists_kernel<64, false>), dim3(gm), blk, 0, st, A);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev1, st));
    c->ev_valid = true;
    return MQ_OK;
}

// Reads that came back MQ_HIT_OVERFLOW (more Match runs than the per-wave scratch holds, or a minimizer list denser than its
// region): map those again on the GPU with worst-case scratch and list regions on a small grid.  Never a CPU path.
static int redo_overflow(mq_ctx *c, const uint8_t *bases, const uint64_t *offsets, const uint32_t *lens, uint32_t n, mq_hit *out) {
    mq_index *idx = c->idx;
    std::vector<uint32_t> redo;
    for (uint32_t i = 0; i < n; ++i)
        if (out[i].status == MQ_HIT_OVERFLOW) redo.push_back(i);
    if (redo.empty()) return MQ_OK;
    uint64_t sub_max = 0;
    std::vector<uint64_t> so(redo.size() + 1, 0);
    for (size_t j = 0; j < redo.size(); ++j) {
        const uint64_t L = lens ? (uint64_t)lens[redo[j]] : offsets[redo[j] + 1] - offsets[redo[j]];
        so[j + 1] = so[j] + L;
        sub_max = std::max(sub_max, L);
    }
    const uint64_t sub_total = so.back();
    std::vector<uint8_t> sb(sub_total ? sub_total : 1);
    for (size_t j = 0; j < redo.size(); ++j) memcpy(sb.data() + so[j], bases + offsets[redo[j]], (size_t)(so[j + 1] - so[j]));
    const uint32_t cap = (uint32_t)std::max<uint64_t>(sub_max, 1);  // a read cannot have more runs than bases
    const uint32_t waves = std::max(MAP_WAVES, ML_WAVES);
    const uint32_t rgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(idx->grid_fused, idx->grid_map), (1ull << 30) / ((uint64_t)cap * sizeof(MatchRec) * waves)));
    int rc = ctx_ensure(c, (uint32_t)redo.size(), sub_total, 65536u);
    if (rc) return rc;
    MatchRec *big = nullptr;
    uint8_t *d_sb = nullptr;
    uint64_t *d_so = nullptr;
    mq_hit *d_sh = nullptr;
    hipError_t e = hipMalloc((void **)&big, (size_t)rgrid * waves * cap * sizeof(MatchRec));
    if (e == hipSuccess) e = hipMalloc((void **)&d_sb, sub_total + 1);
    if (e == hipSuccess) e = hipMalloc((void **)&d_so, so.size() * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_sh, redo.size() * sizeof(mq_hit));
    if (e == hipSuccess && sub_total) e = hipMemcpyAsync(d_sb, sb.data(), sub_total, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_so, so.data(), so.size() * 8, hipMemcpyHostToDevice, c->stream);
    int rrc = MQ_OK;
    std::vector<mq_hit> sh(redo.size());
    if (e == hipSuccess) {
        LaunchOpt o;
        o.scratch_override = big;
        o.cap_override = cap;
        o.grid_override = rgrid;
        o.f16 = 65536u;
        rrc = launch_map(c, d_sb, d_so, (uint32_t)redo.size(), d_sh, c->stream, o);
    }
    if (e == hipSuccess && rrc == MQ_OK) e = hipMemcpyAsync(sh.data(), d_sh, redo.size() * sizeof(mq_hit), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else hipStreamSynchronize(c->stream);
    hipFree(big);
    hipFree(d_sb);
    hipFree(d_so);
    hipFree(d_sh);
    if (e != hipSuccess) return set_err(e == hipErrorOutOfMemory ? MQ_ENOMEM : MQ_EHIP, std::string("overflow retry: ") + hipGetErrorString(e));
    if (rrc) return rrc;
    for (size_t j = 0; j < redo.size(); ++j) out[redo[j]] = sh[j];
    return MQ_OK;
}

// host buffers -> device staging -> launch sequence -> page-locked hits, all asynchronous on the context's stream.
// lens == nullptr: offsets has n + 1 entries and read i is bases[offsets[i], offsets[i+1]).  lens != nullptr (spans form): the
// whole buffer bases[0, buf_bytes) goes to the device and read i is bases[offsets[i], offsets[i] + lens[i]) (n offsets).
static int ctx_submit(mq_ctx *c, const uint8_t *bases, uint64_t buf_bytes, const uint64_t *offsets, const uint32_t *lens, uint32_t n,
                      mq_hit *out) {
    mq_index *idx = c->idx;
    if (c->pending) return set_err(MQ_ESTATE, "context has a submitted batch: call mq_ctx_wait first");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (n == 0) return MQ_OK;
    int rc = use_device(idx);
    if (rc) return rc;
    if ((rc = grow_pinned(c->h_off, c->h_off_cap, (uint64_t)n + 1))) return rc;
    if ((rc = grow_pinned(c->h_out, c->h_out_cap, (uint64_t)n))) return rc;
    uint64_t total, first;
    if (!lens) {
        first = offsets[0];
        total = offsets[n] - offsets[0];
        for (uint32_t i = 0; i < n; ++i) {
            if (offsets[i + 1] < offsets[i]) return set_err(MQ_EINVAL, "offsets must be non-decreasing");
            if (offsets[i + 1] - offsets[i] >= (1ull << 32)) return set_err(MQ_EINVAL, "sequence length must be < 2^32");
            c->h_off[i] = offsets[i] - first;
        }
    } else {
        first = 0;
        total = buf_bytes;
        uint64_t prev_end = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (offsets[i] < prev_end || offsets[i] + lens[i] > buf_bytes) return set_err(MQ_EINVAL, "spans must be in order, disjoint and inside the buffer");
            prev_end = offsets[i] + lens[i];
            c->h_off[i] = offsets[i];
        }
    }
    c->h_off[n] = total;
    if ((rc = ctx_ensure(c, n, total, list_f16(idx)))) return rc;
    if ((rc = grow(c->st_bases, c->st_bases_cap, total + 64))) return rc;
    if ((rc = grow(c->st_off, c->st_off_cap, (uint64_t)n + 1))) return rc;
    if ((rc = grow(c->st_out, c->st_out_cap, (uint64_t)n))) return rc;
    if (lens && (rc = grow(c->st_lens, c->st_lens_cap, (uint64_t)n))) return rc;
    if (total) HIPCHK(hipMemcpyAsync(c->st_bases, bases + first, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->st_off, c->h_off, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    LaunchOpt o;
    if (lens) {
        HIPCHK(hipMemcpyAsync(c->st_lens, lens, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        o.d_lens = c->st_lens;
    }
    rc = launch_map(c, c->st_bases, c->st_off, n, c->st_out, c->stream, o);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->h_out, c->st_out, (size_t)n * sizeof(mq_hit), hipMemcpyDeviceToHost, c->stream));
    c->pending = true;
    c->p_bases = bases;
    c->p_offsets = offsets;
    c->p_lens = lens;
    c->p_n = n;
    c->p_out = out;
    return MQ_OK;
}

static int ctx_wait(mq_ctx *c) {
    if (!c->pending) return MQ_OK;
    c->pending = false;
    int rc = use_device(c->idx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(c->p_out, c->h_out, (size_t)c->p_n * sizeof(mq_hit));
    return redo_overflow(c, c->p_bases, c->p_offsets, c->p_lens, c->p_n, c->p_out);
}

static int ctx_map_device(mq_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases, mq_hit *d_out,
                          hipStream_t st, bool instrumented = false) {
    mq_index *idx = c->idx;
    if (n && (!d_offsets || !d_out)) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (c->pending) return set_err(MQ_ESTATE, "context has a submitted batch: call mq_ctx_wait first");
    int rc = use_device(idx);
    if (rc) return rc;
    if ((rc = ctx_ensure(c, n, total_bases, list_f16(idx)))) return rc;
    LaunchOpt o;
    o.instrumented = instrumented;
    return launch_map(c, d_bases, d_offsets, n, d_out, st, o);
}

extern "C" {

mq_ctx *mq_ctx_new(mq_index *idx) try {
    if (!idx) {
        set_err(MQ_EINVAL, "idx is NULL");
        return nullptr;
    }
    return ctx_create(idx);
} catch (const std::bad_alloc &) {
    set_err(MQ_ENOMEM, "out of host memory");
    return nullptr;
} catch (const std::exception &e) {
    set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
    return nullptr;
}

void mq_ctx_free(mq_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->idx->device);
    ctx_release(ctx);
}

int mq_ctx_submit(mq_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out) try {
    if (!ctx || (n && (!offsets || !out))) return set_err(MQ_EINVAL, "bad arguments");
    return ctx_submit(ctx, bases, 0, offsets, nullptr, n, out);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_submit_spans(mq_ctx *ctx, const uint8_t *buf, uint64_t buf_bytes, const uint64_t *starts, const uint32_t *lens, uint32_t n,
                        mq_hit *out) try {
    if (!ctx || (n && (!buf || !starts || !lens || !out))) return set_err(MQ_EINVAL, "bad arguments");
    return ctx_submit(ctx, buf, buf_bytes, starts, lens, n, out);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_reserve(mq_ctx *ctx, uint32_t n_reads, uint64_t total_bytes) try {
    if (!ctx) return set_err(MQ_EINVAL, "ctx is NULL");
    mq_ctx *c = ctx;
    int rc = use_device(c->idx);
    if (rc) return rc;
    if ((rc = grow_pinned(c->h_off, c->h_off_cap, (uint64_t)n_reads + 1))) return rc;
    if ((rc = grow_pinned(c->h_out, c->h_out_cap, (uint64_t)n_reads))) return rc;
    if ((rc = ctx_ensure(c, n_reads, total_bytes, list_f16(c->idx)))) return rc;
    if ((rc = grow(c->st_bases, c->st_bases_cap, total_bytes + 64))) return rc;
    if ((rc = grow(c->st_off, c->st_off_cap, (uint64_t)n_reads + 1))) return rc;
    if ((rc = grow(c->st_out, c->st_out_cap, (uint64_t)n_reads))) return rc;
    return grow(c->st_lens, c->st_lens_cap, (uint64_t)n_reads);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_wait(mq_ctx *ctx) try {
    if (!ctx) return set_err(MQ_EINVAL, "ctx is NULL");
    return ctx_wait(ctx);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_map_batch(mq_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out) try {
    if (!ctx || (n && (!offsets || !out))) return set_err(MQ_EINVAL, "bad arguments");
    int rc = ctx_submit(ctx, bases, 0, offsets, nullptr, n, out);
    if (rc) return rc;
    return ctx_wait(ctx);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_map_batch_device(mq_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases,
                            mq_hit *d_out, void *stream) try {
    if (!ctx) return set_err(MQ_EINVAL, "ctx is NULL");
    return ctx_map_device(ctx, d_bases, d_offsets, n, total_bases, d_out, (hipStream_t)stream);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_batch_device(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases,
                        mq_hit *d_out, void *stream) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    return ctx_map_device(idx->def_ctx, d_bases, d_offsets, n, total_bases, d_out, (hipStream_t)stream);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_batch(mq_index *idx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out) try {
    if (!idx || (n && (!offsets || !out))) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = ctx_submit(idx->def_ctx, bases, 0, offsets, nullptr, n, out);
    if (rc) return rc;
    return ctx_wait(idx->def_ctx);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_kminmers_batch(mq_index *idx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, const uint64_t *kmm_offsets,
                      mq_kminmer *out, uint32_t *counts) try {
    if (!idx || (n && (!offsets || !kmm_offsets || !counts))) return set_err(MQ_EINVAL, "bad arguments");
    if (n == 0) return MQ_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    const uint64_t total = offsets[n] - offsets[0];
    const uint64_t ktotal = kmm_offsets[n] - kmm_offsets[0];
    for (uint32_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] >= (1ull << 32)) return set_err(MQ_EINVAL, "bad offsets / sequence length must be < 2^32");
    // parity/debug entry point: list regions sized for the worst case (one minimizer per base), so no sequence overflows
    rc = ctx_ensure(idx->def_ctx, n, total, 65536u);
    if (rc) return rc;
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
    {
        LaunchOpt o;
        o.d_dump = d_k;
        o.d_dump_off = d_ko;
        o.d_dump_counts = d_c;
        o.f16 = 65536u;
        rc = launch_map(idx->def_ctx, d_b, d_o, n, d_h, 0, o);
    }
    if (rc) {
        cleanup();
        return rc;
    }
    ok(hipMemcpy(counts, d_c, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (e == hipSuccess && ktotal && out) ok(hipMemcpy(out + kmm_offsets[0], d_k, ktotal * sizeof(mq_kminmer), hipMemcpyDeviceToHost));
    cleanup();
    if (e != hipSuccess) return set_err(MQ_EHIP, std::string("mq_kminmers_batch copy-out: ") + hipGetErrorString(e));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_index_lookup(mq_index *idx, const uint64_t *hashes, uint32_t n, uint8_t *found, mq_kminmer *entries, uint32_t *ref_ids) try {
    if (!idx || (n && (!hashes || !found || !entries || !ref_ids))) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (n == 0) return MQ_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
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
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_format_paf(const mq_index *idx, const char *q_id, uint64_t q_len, const mq_hit *hit, char *buf, size_t cap) try {
    if (!idx || !q_id || !hit || !buf) return set_err(MQ_EINVAL, "bad arguments");
    if (hit->status != MQ_HIT_MAPPED) return set_err(MQ_EINVAL, "hit is not mapped: the reference writes no line");
    auto it = idx->refs.find(hit->ref_id);
    if (it == idx->refs.end()) return set_err(MQ_EINVAL, "unknown ref_id in hit");
    const unsigned long long r_len = it->second.second;
    // src/mers.rs:181: column 11 repeats r_len, column 10 is the score
    const unsigned long long qs = ((unsigned long long)hit->q_start_hi << 32) | hit->q_start, qe = ((unsigned long long)hit->q_end_hi << 32) | hit->q_end;
    int w = snprintf(buf, cap, "%s\t%llu\t%llu\t%llu\t%s\t%s\t%llu\t%u\t%u\t%u\t%llu\t%u", q_id, (unsigned long long)q_len, qs, qe,
                     hit->rc ? "-" : "+", it->second.first.c_str(), r_len, hit->r_start, hit->r_end, hit->score, r_len,
                     hit->mapq);
    return w;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// Page-locked host memory.  hipHostMalloc pins at ~4 GB/s on this platform (and hipHostFree costs another 0.14 s per GB), which
// made the feeder's chunk pool the start-up cost of the read phase; an anonymous mapping backed by transparent huge pages,
// touched and then registered, is page-locked at ~15 GB/s and copies to the device at the full PCIe rate
// (tools/pin_rate.hip, profiles/r03_pin_rate.txt).  Falls back to hipHostMalloc when the mapping or the registration fails.
namespace {
std::mutex g_host_mu;
std::map<void *, std::pair<size_t, bool>> g_host_allocs;  // pointer -> (mapped bytes, true: mmap + hipHostRegister)
}  // namespace

void *mq_host_alloc(size_t bytes) {
    if (!bytes) bytes = 1;
    const size_t huge = 2u << 20;
    const size_t mapped = (bytes + huge - 1) / huge * huge;
    void *p = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p != MAP_FAILED) {
        madvise(p, mapped, MADV_HUGEPAGE);
        for (size_t o = 0; o < mapped; o += 4096) ((volatile uint8_t *)p)[o] = 0;  // fault the pages in (2 MB at a time under THP)
        if (hipHostRegister(p, mapped, hipHostRegisterDefault) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_host_mu);
            g_host_allocs[p] = std::make_pair(mapped, true);
            return p;
        }
        (void)hipGetLastError();
        munmap(p, mapped);
    }
    p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_err(e == hipErrorOutOfMemory ? MQ_ENOMEM : MQ_EHIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_allocs[p] = std::make_pair(bytes, false);
    return p;
}

void mq_host_free(void *p) {
    if (!p) return;
    std::pair<size_t, bool> info(0, false);
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = g_host_allocs.find(p);
        if (it == g_host_allocs.end()) return;  // not ours
        info = it->second;
        g_host_allocs.erase(it);
    }
    if (info.second) {
        hipHostUnregister(p);
        munmap(p, info.first);
    } else {
        hipHostFree(p);
    }
}

int mq_last_map_path_counts(mq_index *idx, uint32_t *n_fast, uint32_t *n_general) try {
    if (!idx || !n_fast || !n_general) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    if (!c->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(c->ev1));
    uint32_t v[2] = {0, 0};
    HIPCHK(hipMemcpy(v, c->d_counter + 4, 8, hipMemcpyDeviceToHost));
    *n_fast = v[0];
    *n_general = v[1];
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_probe_stats(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases, mq_hit *d_out,
                       uint64_t *lookups, uint64_t *extra_steps) try {
    if (!idx || !lookups || !extra_steps) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    int rc = ctx_map_device(c, d_bases, d_offsets, n, total_bases, d_out, nullptr, true);  // the choice travels with this launch: contexts never see it
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(c->ev1));
    uint64_t v[2];
    HIPCHK(hipMemcpy(v, c->d_counter + 8, 16, hipMemcpyDeviceToHost));
    *extra_steps = v[0];
    *lookups = v[1];
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// Diagnostic (-DMQ_STAGE_CLOCKS builds; zeros otherwise): shader-clock cycles the waves of the last map_kernel launch of the default
// context spent per stage, summed over waves (stage list: mq_device.hpp, mq_clk).
int mq_last_stage_clocks(mq_index *idx, uint64_t *out12) try {
    if (!idx || !out12) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    if (!c->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipMemcpy(out12, c->d_counter + 16, MQ_N_CLK * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_last_map_ms(mq_index *idx, float *ms) try {
    if (!idx || !ms) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    if (!c->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_probe_rate(mq_index *idx, uint32_t blocks, uint32_t per_thread, uint32_t bitmap_log2, uint32_t table_too, float *ms,
                  uint64_t *lookups, uint64_t *extra_steps) try {
    if (!idx || !ms || !lookups || !extra_steps) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    unsigned long long *d_acc = nullptr;
    uint32_t *bm = nullptr;
    uint64_t bit_mask = 0;
    HIPCHK(hipMalloc((void **)&d_acc, 16));
    HIPCHK(hipMemset(d_acc, 0, 16));
    if (bitmap_log2) {  // a stand-in bitmap with one bit in eight set
        bit_mask = (1ull << bitmap_log2) - 1;
        HIPCHK(hipMalloc((void **)&bm, (size_t)1 << (bitmap_log2 - 3)));
        HIPCHK(hipMemset(bm, 0x10, (size_t)1 << (bitmap_log2 - 3)));
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    per_thread = (per_thread + 3u) & ~3u;
    hipLaunchKernelGGL(probe_rate_kernel, dim3(blocks), dim3(256), 0, 0, idx->table, idx->nslots - 1, per_thread, 1ull, d_acc, bm, bit_mask, table_too);  // warm-up
    HIPCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe_rate_kernel, dim3(blocks), dim3(256), 0, 0, idx->table, idx->nslots - 1, per_thread, 0x1234567ull, d_acc, bm, bit_mask, table_too);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(ms, e0, e1));
    unsigned long long acc[2];
    HIPCHK(hipMemcpy(acc, d_acc, 16, hipMemcpyDeviceToHost));
    hipFree(d_acc);
    hipFree(bm);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *lookups = (uint64_t)blocks * 256ull * per_thread;
    *extra_steps = acc[1] / 2;  // two launches accumulated
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_last_map_ms(mq_ctx *ctx, float *ms) try {
    if (!ctx || !ms) return set_err(MQ_EINVAL, "bad arguments");
    if (!ctx->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

}  // extern "C"
