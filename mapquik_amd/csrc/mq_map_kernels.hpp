// mq_map_kernels.hpp -- the map path's kernels (part of the one translation unit mq_capi.hip): map_kernel (fused: seed + map phases of a read
// back to back in one wave) and the same device functions as separate launches (MQ_PIPELINE=split, diagnostic).
#pragma once

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
    uint32_t *mz_last;     // seeding variant 16 only (else nullptr): every minimizer's second position, same regions as mz_pos
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
    uint4 *work;                  // map_kernel's work items in launch order (order_reads_kernel)
    uint32_t heavy_first;         // order_reads_kernel: reads that look like short-period tandem arrays go first
};

// ------------------------------------------------------------------- launch order
// map_kernel takes work item i (an atomic counter) = descriptor work[WORK_FRONT_CAP - nf + i], i < nf + n:
//   {offset lo, offset hi, length (low 32 bits), read | WORK_SKIP | WORK_TOO_LONG}
// written per launch by order_reads_kernel (one lane per read): entry WORK_FRONT_CAP + r describes read r, and the nf = counters[7] reads
// that go FIRST sit in front of them (growing downwards; their natural entries carry WORK_SKIP).  One 16-byte load per read instead of
// two or three dependent on the layout of the caller's arrays -- and a place to decide the order.
// Which reads go first: those that look like a short-period tandem array.  A read inside an array of period p < l has p distinct l-mers;
// when one of them passes the density test the read lists a minimizer every p bases -- thousands instead of ~350 -- and costs its wave 10-40
// times the ordinary read (profiles/r05_read_tail.txt).  Taken up in the last third of a launch such a read IS the launch's tail (every
// other wave has left); taken up first it costs nothing but its own work.  The test (three windows of 48 bases near 1/6, 3/6, 5/6 of the read:
// some lag 1..16 matches in >= 27 of 32 positions) costs ~800 instructions per READ against map_kernel's ~13,000 per read per WAVE;
// it flags 0.1-0.3 % of the reads of a human-like batch, among them every read with > 10 x the median minimizer count and 21 of 23
// with > 5 x (tools/heavy_study.py).  The order changes nothing but the order: results are stored by read number.
constexpr uint32_t WORK_FRONT_CAP = 32768;  // reads that can go first (more are flagged: the rest stay where they are)
constexpr uint32_t WORK_SKIP = 0x80000000u, WORK_TOO_LONG = 0x40000000u, WORK_ID_MASK = 0x3FFFFFFFu;
constexpr uint32_t WORK_NF = 7;             // counters[WORK_NF]: reads flagged (may exceed WORK_FRONT_CAP)

// 48 bases at p: does some lag 1..16 match in >= 27 of the first 32 positions?
__device__ __forceinline__ bool window_periodic(const uint8_t *p) {
    uint32_t by[12];  // four 2-bit codes per byte: (c >> 1) & 3 of A C G T (either case), gathered by a multiplication without carries
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint4 v = *reinterpret_cast<const uint4_unaligned *>(p + 16 * j);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) by[4 * j + i] = ((((w[i] >> 1) & 0x03030303u) * 0x01041040u) >> 24);
    }
    const uint64_t s0 = (uint64_t)(by[0] | (by[1] << 8) | (by[2] << 16) | (by[3] << 24)) | ((uint64_t)(by[4] | (by[5] << 8) | (by[6] << 16) | (by[7] << 24)) << 32);
    const uint64_t s1 = (uint64_t)(by[8] | (by[9] << 8) | (by[10] << 16) | (by[11] << 24));
    bool hit = false;
#pragma unroll
    for (uint32_t lag = 1; lag <= 16; ++lag) {
        const uint64_t sh = (s0 >> (2u * lag)) | (s1 << (64u - 2u * lag));  // bases lag .. lag + 31
        const uint64_t d = s0 ^ sh;
        hit |= __popcll((d | (d >> 1)) & 0x5555555555555555ull) <= 5;
    }
    return hit;
}

__global__ __launch_bounds__(256) void order_reads_kernel(const SplitArgs A) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= A.n) return;
    const uint64_t o0 = A.offsets[r];
    const uint64_t len = A.lens ? (uint64_t)A.lens[r] : A.offsets[r + 1] - o0;
    uint32_t w = r;
    if (len >> 32) w |= WORK_TOO_LONG;
    else if (A.heavy_first && len >= 512u) {
        // (a window starts at a multiple of 64 bytes of the batch: its 48 bytes come from one 64-byte sector -- the pass is bound by the sectors it touches)
        const uint32_t ln = (uint32_t)len;
        auto win = [&](uint32_t at) { return A.bases + ((o0 + at) & ~(uint64_t)63); };  // ln >= 512: at >= 85, the window lies inside the read
        const bool p1 = window_periodic(win(ln / 6u)), p2 = window_periodic(win(ln / 2u)), p3 = window_periodic(win((ln / 6u) * 5u));  // (no short cut: the three windows' loads are in flight together)
        if (p1 | p2 | p3) {
            const uint32_t k = atomicAdd(&A.counters[WORK_NF], 1u);
            if (k < WORK_FRONT_CAP) {
                A.work[WORK_FRONT_CAP - 1u - k] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), ln, w);
                w |= WORK_SKIP;
            }
        }
    }
    A.work[WORK_FRONT_CAP + r] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), (uint32_t)len, w);
}

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
template <int STOP = 0, bool VAR = true>
__device__ __forceinline__ uint32_t seed_read_fast(const SplitArgs &A, const SeedTables &T, SeedLds &S, const uint8_t *seq,
                                                   uint32_t len, uint64_t &base, uint32_t cap, uint32_t &n_moved, APre &pre, bool pre_valid) {
    uint32_t cnt = seed_sequence_fast<STOP, false, VAR>(seq, len, A.P, T, S, A.mz_hash + base, A.mz_pos + base, cap, pre, pre_valid, SeedView(),
                                                        A.mz_last ? A.mz_last + base : nullptr);
    if (cnt != SD_NOT_FAST && cnt > cap) {  // denser than its region: once more, into an exact-size pool region
        if (pool_take(A, cnt, base)) {
            seed_sequence_fast<0, false, VAR>(seq, len, A.P, T, S, A.mz_hash + base, A.mz_pos + base, cnt, pre, false, SeedView(), A.mz_last ? A.mz_last + base : nullptr);
            n_moved++;
        } else {
            cnt = LIST_OVERFLOW;
        }
    }
    return cnt;
}

// seed phase, general streaming seeder (any bytes, any length)
template <bool VAR = true>
__device__ __forceinline__ uint32_t seed_read_general(const SplitArgs &A, WaveLds &S, const uint8_t *seq, uint64_t len, uint64_t &base,
                                                      uint32_t cap, uint32_t &n_moved) {
    uint32_t cnt;
    {
        SoaListSink sink(A.mz_hash + base, A.mz_pos + base, (VAR && A.mz_last) ? A.mz_last + base : nullptr, cap);
        uint32_t mz_count = 0;
        seed_segment<VAR>(seq, len, 0, len, A.P, S, sink, mz_count);
        cnt = sink.written;
    }
    if (cnt > cap) {
        if (pool_take(A, cnt, base)) {
            SoaListSink sink(A.mz_hash + base, A.mz_pos + base, (VAR && A.mz_last) ? A.mz_last + base : nullptr, cnt);
            uint32_t mz_count = 0;
            seed_segment<VAR>(seq, len, 0, len, A.P, S, sink, mz_count);
            n_moved++;
        } else {
            cnt = LIST_OVERFLOW;
        }
    }
    return cnt;
}

// ------------------------------------------------------------------- reads the fast seeder declined: stretch by stretch
// The fast seeder takes sequences of A C G T; a read with anything else (a run of N from a gap of the reference it was drawn from) went through
// the general streaming seeder whole, at 8-13 x the cost.  seed_read_hybrid cuts such a read: every stretch of >= HYB_MIN_CLEAN bytes (whole
// 64-byte blocks) of A C G T is seeded by the fast seeder as a sequence of its own (a VIEW: positions shifted, the byte in front of it known) -- that lists exactly the
// windows that lie wholly inside the stretch -- and the general seeder lists the rest: the windows that start between two stretches, and those
// that start in the last l - 1 run heads of a stretch and reach past its end (seed_segment's min_last filter drops the ones the fast seeder
// listed).  Same list as the general seeder's over the whole read (tests: MQ_FORCE_GENERAL=1 takes that one).
constexpr uint32_t HYB_MIN_CLEAN = 2048;
// bit i: the 64-byte block at p + 64 i (p a multiple of 64) lies inside the sequence and holds only A C G T (a c g t too when folding); 4 KB a call.
// The bytes come in as a CleanPiece (requested one call ahead: the scan's next 4 KB are on their way while these are looked at -- every call
// used to open with a memory round trip of its own, six of them for a 24-kb read).
struct CleanPiece {
    uint4 v[4];
};
__device__ __forceinline__ void clean_request(const uint8_t *__restrict__ seq, uint64_t len, uint64_t p, CleanPiece &c) {
    const uint64_t at = p + 64u * lane_id();
#pragma unroll
    for (int j = 0; j < 4; ++j) c.v[j] = make_uint4(0u, 0u, 0u, 0u);
    if (at + 64u <= len) {  // (a block that is not wholly inside the sequence is never looked at, and never read)
#pragma unroll
        for (int j = 0; j < 4; ++j) c.v[j] = *reinterpret_cast<const uint4_unaligned *>(seq + at + 16 * j);
    }
}
__device__ __forceinline__ uint64_t clean_blocks(const CleanPiece &c, uint64_t len, uint64_t p, bool fold) {
    const uint64_t at = p + 64u * lane_id();
    bool ok = false;
    if (at + 64u <= len) {
        const uint32_t fm = fold ? 0xDFDFDFDFu : 0xFFFFFFFFu;
        uint32_t all = 0x80808080u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w[4] = {c.v[j].x, c.v[j].y, c.v[j].z, c.v[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t x = w[i] & fm;
                uint32_t is = 0;  // 0x80 in every byte that is A, C, G or T: exact zero-byte masks of x ^ letter
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t y = x ^ (t == 0 ? 0x41414141u : t == 1 ? 0x43434343u : t == 2 ? 0x47474747u : 0x54545454u);
                    is |= ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
                }
                all &= is;
            }
        }
        ok = all == 0x80808080u;
    }
    return __ballot(ok);
}
// the first run of at least HYB_MIN_CLEAN / 64 consecutive clean blocks at or behind `from` (a multiple of 64): [s, d), both multiples of 64; false: none.
// One pass over the bytes between `from` and d, however the other bytes are spread.
__device__ __forceinline__ bool next_clean_stretch(const uint8_t *__restrict__ seq, uint64_t len, uint64_t from, bool fold, uint64_t &s, uint64_t &d) {
    constexpr uint64_t NONE = ~(uint64_t)0;
    uint64_t run_s = NONE;
    CleanPiece cur, nxt;
    if (from < len) clean_request(seq, len, from, cur);
    for (uint64_t p = from; p < len; p += 4096u) {
        if (p + 4096u < len) clean_request(seq, len, p + 4096u, nxt);
        const uint64_t m = clean_blocks(cur, len, p, fold);
        cur = nxt;
        uint32_t bit = 0;
        while (bit < 64u) {
            if (run_s == NONE) {
                const uint64_t rest = m >> bit;
                if (!rest) break;
                bit += (uint32_t)__ffsll((long long)rest) - 1u;
                run_s = p + 64u * bit;
            }
            const uint64_t rest0 = ~(m >> bit);  // (the bits shifted in from above read as "not clean": a run never passes bit 63 here)
            const uint32_t z = (uint32_t)__ffsll((long long)rest0) - 1u;
            if (bit + z >= 64u) break;  // the run goes on into the next 4 KB
            const uint64_t run_e = p + 64u * (bit + z);
            if (run_e - run_s >= HYB_MIN_CLEAN) {
                s = run_s;
                d = run_e;
                return true;
            }
            run_s = NONE;
            bit += z;
        }
    }
    if (run_s != NONE) {  // a run up to the sequence's last whole block
        const uint64_t run_e = len & ~(uint64_t)63;
        if (run_e > run_s && run_e - run_s >= HYB_MIN_CLEAN) {
            s = run_s;
            d = run_e;
            return true;
        }
    }
    return false;
}
// a0 in [lo, d] such that [a0, d) holds at least `need` run heads (or a0 = lo): where the general seeder has to start for the windows that reach d
__device__ __forceinline__ uint64_t heads_back(const uint8_t *__restrict__ seq, uint64_t lo, uint64_t d, uint32_t need, const DevParams &P) {
    const uint32_t lane = lane_id();
    uint64_t a0 = d;
    uint32_t heads = 0;
    while (a0 > lo && heads < need) {
        const uint32_t step = a0 - lo < 64u ? (uint32_t)(a0 - lo) : 64u;
        a0 -= step;
        bool head = false;
        if (lane < step) {
            const uint64_t i = a0 + lane;
            uint32_t b = seq[i], pb = i > 0 ? (uint32_t)seq[i - 1] : 0x100u;
            if (P.fold && b - 'a' < 26u) b -= 32u;
            if (P.fold && pb - 'a' < 26u) pb -= 32u;
            head = !P.use_hpc || b != pb;
        }
        heads += (uint32_t)__popcll(__ballot(head));
    }
    return a0;
}
// tacc (instrumented launch only, else nullptr): cycles spent [0] finding stretches, [1] in the fast seeder, [2] in the general seeder
template <bool VAR = true>
__device__ __forceinline__ uint32_t seed_read_hybrid(const SplitArgs &A, const SeedTables &T, SeedLds &SF, WaveLds &SG, const uint8_t *seq, uint64_t len,
                                                     uint64_t &base, uint32_t cap, uint32_t &n_moved, unsigned long long *tacc = nullptr) {
    const DevParams &P = A.P;
    auto pass = [&](uint64_t base_, uint32_t cap_) __attribute__((always_inline)) -> uint32_t {
        uint32_t n_out = 0;
        unsigned long long t_mark = tacc ? __builtin_amdgcn_s_memtime() : 0ull;
        auto charge = [&](int i) {
            if (tacc) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                tacc[i] += now - t_mark;
                t_mark = now;
            }
        };
        auto general = [&](uint64_t a, uint64_t b, uint64_t min_last) {
            charge(0);
            wave_sync();
            SoaListSink sink(A.mz_hash + base_ + n_out, A.mz_pos + base_ + n_out, (VAR && A.mz_last) ? A.mz_last + base_ + n_out : nullptr, cap_ > n_out ? cap_ - n_out : 0u);
            uint32_t mzc = 0;
            seed_segment<VAR>(seq, len, a, b, P, SG, sink, mzc, min_last);
            n_out += sink.written;
            wave_sync();
            charge(2);
        };
        if (A.force_general || len < 16u || (len >> 32)) {  // the test hook; the scanner's precondition; beyond the fast seeder's range
            general(0, len, 0);
            return n_out;
        }
        uint64_t seg_a = 0, seg_min_last = 0;  // the open general segment: windows that start at or behind seg_a (and end at or behind seg_min_last)
        uint64_t pos = 0, s = 0, d = 0;
        while (pos < len && next_clean_stretch(seq, len, pos, P.fold != 0, s, d)) {  // [s, d): whole 64-byte blocks of A C G T, at least HYB_MIN_CLEAN bytes
            pos = d;
            if (d - s > 0xFFFFFF00ull) continue;
            if (s > seg_a) general(seg_a, s, seg_min_last);  // the windows that start in front of the stretch
            APre pre;
            SeedView V;
            V.first_prev = 4u;
            if (s > 0) {  // the byte in front of the stretch as the seeder's 2-bit code (the stretch may begin inside a run: its first base is then no run head)
                uint32_t b = seq[s - 1];
                if (P.fold && b - 'a' < 26u) b -= 32u;
                V.first_prev = b == 'A' ? 0u : b == 'C' ? 1u : b == 'T' ? 2u : b == 'G' ? 3u : 4u;
            }
            V.elig_end = (uint32_t)(((d - s) + 127u) & ~(uint64_t)63);
            V.pos_add = (uint32_t)s;
            V.more_after = 0u;
            const uint32_t left = cap_ > n_out ? cap_ - n_out : 0u;
            charge(0);
            const uint32_t c = seed_sequence_fast<0, true, VAR>(seq + s, (uint32_t)(d - s), P, T, SF, A.mz_hash + base_ + n_out, A.mz_pos + base_ + n_out, left, pre, false, V,
                                                                A.mz_last ? A.mz_last + base_ + n_out : nullptr);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_sync();
            charge(1);
            if (c == SD_NOT_FAST) {  // (a candidate on the bound: the general seeder's test is exact) the stretch joins the general segment
                seg_a = s;
                seg_min_last = 0;
            } else {
                n_out += c;
                seg_a = heads_back(seq, s, d, P.l - 1u, P);
                seg_min_last = d;
            }
        }
        if (seg_a < len && seg_min_last < len) general(seg_a, len, seg_min_last);
        charge(0);
        return n_out;
    };
    // ONE call site, so that the pass is inlined here: as a function of its own (two call sites) it received T, SF and SG as generic pointers
    // and every LDS access of both seeders in it became a FLAT instruction (524 flat loads, 24 ds_ in round 5's code object) -- a declined read
    // cost 3-4 x what its instructions should
    uint32_t cnt = 0, cap_now = cap;
    for (int attempt = 0;; ++attempt) {
        cnt = pass(base, cap_now);
        if (cnt <= cap_now || attempt == 1) break;
        if (!pool_take(A, cnt, base)) {  // denser than its region: once more, into an exact-size pool region
            cnt = LIST_OVERFLOW;
            break;
        }
        cap_now = cnt;
        n_moved++;
    }
    return cnt;
}

// 6: a 24-kb HiFi read lists ~350 minimizers = 5.4 lane-batches, so six cover four reads in five in one chunk; 7 (rounds 3-4) covered
// nearly all at two more key + four more payload registers across probe_all: 1255-1257 against 1249-1250 Gbases/s (round 5, same box); 8
// does not fit 128 registers at all (169)
#ifndef MQ_ML_NB
#define MQ_ML_NB 6
#endif
constexpr int ML_NB = MQ_ML_NB;                              // lane-batches of 64 k-min-mers hashed and probed together
constexpr uint32_t ML_LIST_CAP = 64 * ML_NB + MAX_L;   // minimizers staged in LDS at a time (64 * ML_NB + k - 1 used, k <= 32)
static_assert(MAX_L >= 32, "k - 1 <= 31 entries of overlap between chunks");
struct MapListLds {
    unsigned long long h[ML_LIST_CAP];
    uint32_t p[ML_LIST_CAP];
    MatchRec rec[MAP_LDS_RECS];  // MapSink::lds_rec
};

// map phase of read r: its list (cnt entries at base) -> mq_hit
// the read's result is left in h (all lanes hold it); store_hit() writes it: the fused kernel does that after it has taken the
// prefetched descriptor of its next read out of its registers, so that this store's acknowledgement is nothing a wave waits for
__device__ __forceinline__ void store_hit(const SplitArgs &A, uint32_t r, const mq_hit &h) {
    if (lane_id() == 0) {
        A.out[r] = h;
        if (A.dump_counts) A.dump_counts[r] = h.n_kminmers;
    }
}

template <int CH, bool TIMING, bool VAR = true>
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
        MapSink sink(A.table, A.mask, P, scratch, A.cap_matches, S.rec, d, dcap, var_rev_eq<VAR>(P));
        const unsigned long long *lh = A.mz_hash + base;
        const uint32_t *lp = A.mz_pos + base;
        const uint32_t chunk = 64u * (uint32_t)ML_NB + P.k - 1u;
        for (uint32_t g = 0; g + P.k <= cnt;) {
            const uint32_t have = cnt - g < chunk ? cnt - g : chunk;
            {  // L2-served loads (the list was written by this very wave), ALL in flight before the first is stored: one L2 round trip
               // per chunk (a loop that loads and stores 64 entries at a time exposes one per 64 entries)
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
            // seeding variant 16: the chunk's second positions take the place of its hashes once those are used up (S.h as dwords)
            uint32_t *mzq = (VAR && A.mz_last) ? reinterpret_cast<uint32_t *>(S.h) : nullptr;
            sink.template consume_list<ML_NB>(S.h, S.p, have, [&]() {
                if (mzq) {
                    const uint32_t *lq = A.mz_last + base + g;
                    wave_sync();  // every lane's reads of the staged hashes are done
                    for (uint32_t i = lane; i < have; i += 64u) mzq[i] = ld_sc1_u32(lq + i);
                    wave_sync();
                }
            }, mzq);
            wave_sync();
            g += have - (P.k - 1u);
        }
        sink.finish_runs();
        n_kmm = sink.kmm_count;
        if (sink.n_matches > 0 && sink.n_matches <= MAP_LDS_RECS && sink.n_matches <= (uint32_t)CH) {
            // all of the read's records are in LDS: lane i takes record i, and the chain stage -- one chunk -- reads no Match record
            wave_sync();
            MatchRec m0 = {};
            if (lane_id() < sink.n_matches) m0 = S.rec[lane_id()];
            mq_clk(8);
            chain_stage<CH>(S.rec, sink.n_matches, P, len, A.ref_lens, h, &m0);
        } else {
            if (sink.n_matches > 0 && sink.n_matches <= A.cap_matches) {  // many records: those in LDS join the others in the scratch
                wave_sync();
                sink.records_to_scratch();
            }
            if (sink.n_matches > A.cap_matches) {
                h.status = MQ_HIT_OVERFLOW;
            } else if (sink.n_matches > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Match records written by this wave are in L2
                wave_sync();
                mq_clk(8);
                chain_stage<CH>(scratch, sink.n_matches, P, len, A.ref_lens, h);
            }
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

// per-wave LDS of the fused kernel: the phases of one read follow each other, so they share the memory.  (map_declined_kernel hands
// `seed` AND `general` to seed_read_hybrid, which runs the two seeders in turn: neither may carry LDS state from one call to the next --
// every seed_sequence_fast / seed_segment call starts from scratch and is fenced by wave_sync + vmcnt(0).)
union MapWaveLds {
    SeedLds seed;
    WaveLds general;
    MapListLds map;
};

// CH: lanes per chunk in the chain stage (64 in production; 4 only in tests so that ordinary reads take the multi-chunk path)
// VAR: built with the seeding variants (mq_params.flags bits 8..13); the launch for variant 0 -- the frozen reading, every timed launch --
// uses the instantiation without them
// (Experiments that lived here behind macros until round 5 -- the next read's first super-row prefetched into LDS, the read's minimizer
// list kept in LDS, the prefetch issued when the probes are done, staggered wave starts: all measured at or below the product path --
// are in git history, last at commit cc976e3; numbers in profiles/NOTES.md.)
template <int CH, bool TIMING = false, bool VAR = false>
__global__ __launch_bounds__(64 * MAP_WAVES, MQ_MAP_MIN_WAVES) void map_kernel(const SplitArgs A) {
    // one block of LDS with the tables FIRST: T.rot's entries are addressed through the 16-bit immediate offset of ds_read_b128
    __shared__ struct {
        SeedTables T;
        MapWaveLds SS[MAP_WAVES];
    } W;
    SeedTables &T = W.T;
    MapWaveLds(&SS)[MAP_WAVES] = W.SS;
    build_seed_tables(T, A.P.l, var_h32<VAR>(A.P));
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
    // phase, its descriptor during the map phase -- two dependent memory round trips per read that no wave waits for.  (Requesting
    // the next read's first super-row across the map phase as well was measured at -3 %: a wave's loads return in order, so the
    // map phase's first wait -- an L2 round trip for the list -- then sits behind an HBM one.)
    // work items = order_reads_kernel's descriptors: the reads that go first, then all reads in their own order (those that went first marked)
    const uint32_t nf = A.counters[WORK_NF] < WORK_FRONT_CAP ? A.counters[WORK_NF] : WORK_FRONT_CAP;
    const uint4 *work = A.work + (WORK_FRONT_CAP - nf);
    const uint32_t n_items = A.n + nf;
    uint32_t r = 0xFFFFFFFFu;  // the read being worked on; none: the wave is done
    uint64_t o0 = 0, len = 0;
    // A wave's first two work items are its own (item w and item n_waves + w of wave w): taken from the counter, the launch opened with two atomics
    // per wave on ONE address at the same instant -- 8,192 of them are served in ~65 us, and the wave served last starts that much later
    // (tools/launch_fixed_cost.py, tools/stage_clocks.py --reads 4096).  The counter hands out the items from 2 n_waves on.
    const uint32_t n_waves = gridDim.x * (uint32_t)MAP_WAVES;
    uint32_t own1 = (uint32_t)wave_gid, own2 = n_waves + (uint32_t)wave_gid;  // 0xFFFFFFFF once taken
    // a work item taken with nothing to do meanwhile: a wave's first, and the one after an entry whose read went first (one in hundreds)
    auto take_now = [&]() {
        for (;;) {
            uint32_t i;
            if (own1 != 0xFFFFFFFFu) {
                i = own1;
                own1 = 0xFFFFFFFFu;
            } else if (own2 != 0xFFFFFFFFu) {  // (the first was an entry whose read went first: the wave's second item is still its own)
                i = own2;
                own2 = 0xFFFFFFFFu;
            } else {
                i = 0;
                if (lane == 0) i = atomicAdd(&A.counters[0], 1u);
                i = rdfirst(i) + 2u * n_waves;
            }
            r = 0xFFFFFFFFu;
            if (i >= n_items) return;
            const uint4 d = work[i];
            if (d.w & WORK_SKIP) continue;
            r = d.w & WORK_ID_MASK;
            o0 = ((uint64_t)d.y << 32) | d.x;
            len = (uint64_t)d.z | ((uint64_t)((d.w >> 30) & 1u) << 32);
            return;
        }
    };
    take_now();
    APre pre;  // the current read's first super-row (requested by seed_sequence_fast itself: nothing requests it ahead of the read)
    while (r < A.n) {
        // the instrumented launch (TIMING, never timed) also notes what every read cost its wave: mq_last_read_cycles
        const unsigned long long t_read0 = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
        const unsigned long long t_real0 = TIMING ? __builtin_amdgcn_s_memrealtime() : 0ull;
        // THE hand-over of work items (one place): the item after this read is the wave's own second item or comes from the counter
        uint32_t rn_v = own2;
        if (own2 == 0xFFFFFFFFu) {
            if (lane == 0) rn_v = atomicAdd(&A.counters[0], 1u);
            rn_v += 2u * n_waves;
        }
        own2 = 0xFFFFFFFFu;
        uint32_t cnt = 0;
        uint64_t base = 0;
        mq_clk(11);
        // extract(): len < l + k - 1 => None (src/mers.rs:44)
        if (len >> 32) {
            cnt = LIST_OVERFLOW;  // beyond the documented limit (checked on the host where the host sees the lengths): loud, not wrong
        } else if (len >= (uint64_t)P.l + P.k - 1u && !var_keep_none<VAR>(P)) {
            uint32_t cap;
            list_region(A, o0 - o_base, len, r, base, cap);
            cnt = A.force_general ? SD_NOT_FAST : seed_read_fast<0, VAR>(A, T, S.seed, A.bases + o0, (uint32_t)len, base, cap, n_moved, pre, false);
            if (cnt == SD_NOT_FAST) {
                n_general++;
                // a byte other than A C G T (or a candidate on the bound): the read is queued for map_declined_kernel, which seeds it stretch
                // by stretch and maps it (what is stored for it here -- an unmapped record -- is overwritten there)
                if (lane == 0) A.queue[atomicAdd(&A.counters[2], 1u)] = r;
                cnt = 0;
                mq_clk(10);
            } else {
                n_fast++;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's list stores have reached L2
            wave_sync();
            mq_clk(4);
        }
        const uint32_t rn = rdfirst(rn_v);
        uint4 nd = make_uint4(0u, 0u, 0u, 0u);
        if (lane == 0 && rn < n_items) nd = work[rn];  // a vector load by one lane: in flight through the map phase (a scalar load would be waited for at its first LDS wait)
        mq_hit h;
        map_read<CH, TIMING, VAR>(A, S.map, scratch, r, len, cnt, base, t_steps, t_lookups, h);
        wave_sync();
        const uint32_t r_done = r;
        const uint32_t nw = rdfirst(nd.w);
        o0 = ((uint64_t)rdfirst(nd.y) << 32) | rdfirst(nd.x);
        len = (uint64_t)rdfirst(nd.z) | ((uint64_t)((nw >> 30) & 1u) << 32);
        r = rn < n_items ? (nw & WORK_ID_MASK) : 0xFFFFFFFFu;
        asm volatile("" ::: "memory");  // the prefetched descriptor is out of its registers before the result's store is issued
        store_hit(A, r_done, h);
        if (rn < n_items && (nw & WORK_SKIP)) take_now();  // that read went first
        if (TIMING && lane == 0) {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_read0;
            A.mz_count[r_done] = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dt;
            A.mz_base[r_done] = t_real0;
        }
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

// The reads map_kernel queued (the fast seeder declined them): seeded stretch by stretch (seed_read_hybrid), mapped, stored.  Launched behind
// map_kernel in every launch sequence; its waves leave at once when the queue is empty.
template <int CH, bool TIMING = false, bool VAR = false>
__global__ __launch_bounds__(64 * MAP_WAVES, MQ_MAP_MIN_WAVES) void map_declined_kernel(const SplitArgs A) {
    const uint32_t nq = A.counters[2];
    if (nq == 0) return;
    __shared__ struct {
        SeedTables T;
        MapWaveLds SS[MAP_WAVES];
    } W;
    build_seed_tables(W.T, A.P.l, var_h32<VAR>(A.P));
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t wv = rdfirst(threadIdx.x >> 6);
    MapWaveLds &S = W.SS[wv];
    const size_t wave_gid = (size_t)blockIdx.x * MAP_WAVES + wv;
    MatchRec *scratch = A.scratch_all + wave_gid * A.cap_matches;
    const uint64_t o_base = A.offsets[0];
    uint32_t n_moved = 0;
    unsigned long long t_steps = 0, t_lookups = 0;
    for (;;) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(&A.counters[3], 1u);
        i = rdfirst(i);
        if (i >= nq) break;
        const uint32_t r = A.queue[i];
        const unsigned long long t_read0 = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
        const unsigned long long t_real0 = TIMING ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const uint64_t o0 = A.offsets[r];
        const uint64_t len = A.lens ? (uint64_t)A.lens[r] : A.offsets[r + 1] - o0;
        uint64_t base;
        uint32_t cap;
        list_region(A, o0 - o_base, len, r, base, cap);
        unsigned long long tacc[3] = {0ull, 0ull, 0ull};
        const uint32_t cnt = seed_read_hybrid<VAR>(A, W.T, S.seed, S.general, A.bases + o0, len, base, cap, n_moved, TIMING ? tacc : nullptr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's list stores have reached L2
        wave_sync();
        mq_hit h;
        map_read<CH, TIMING, VAR>(A, S.map, scratch, r, len, cnt, base, t_steps, t_lookups, h);
        wave_sync();
        store_hit(A, r, h);
        if (TIMING && lane == 0) {  // mq_last_read_cycles: this read's cycles here; in the start word's place bit 63 and the seeding split in
            // units of 256 cycles: stretch finder | fast seeder << 21 | general seeder << 42 (tools/read_tail.py)
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_read0;
            A.mz_count[r] = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dt;
            auto u21 = [](unsigned long long c) { return (c >> 8) > 0x1FFFFFull ? 0x1FFFFFull : (c >> 8); };
            A.mz_base[r] = (1ull << 63) | u21(tacc[0]) | (u21(tacc[1]) << 21) | (u21(tacc[2]) << 42);
            (void)t_real0;
        }
    }
    if (lane == 0) {
        if (n_moved) atomicAdd(&A.counters[6], n_moved);
        if (TIMING) {
            atomicAdd(&A.stats64[0], t_steps);
            atomicAdd(&A.stats64[1], t_lookups);
        }
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
    build_seed_tables(T, A.P.l, var_h32<true>(A.P));
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
        } else if (len >= (uint64_t)P.l + P.k - 1u && !var_keep_none<true>(P)) {
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
