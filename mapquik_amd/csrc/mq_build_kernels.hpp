// mq_build_kernels.hpp -- kernels of the index build, the on-disk form and the lookup entry point (part of the one translation unit
// mq_capi.hip).
#pragma once

// ------------------------------------------------------------------- index build (mers::ref_extract + Index::add_with_mer, src/mers.rs:15-38, src/index.rs:94-104)
// Stage 1: the ordered minimizers of a reference, segment by segment.  A segment = REF_SEG raw bases; its VIEW = the segment plus a
// halo of REF_HALO bases (two whole tiles of the fast seeder) seeded like a read (seed_sequence_fast<.., VIEW>): only the minimizers
// whose l-mer starts inside the segment are listed, the halo is there so that those l-mers end inside the view.  A view with a byte
// other than A C G T, an inexact candidate, or fewer than l - 1 run heads in its halo goes to the general streaming seeder
// (seed_segment over [a, b) of the whole sequence) through a queue; both give the same list (tests: MQ_FORCE_GENERAL=1, N runs and
// homopolymer runs across segment borders).
constexpr uint32_t REF_HALO = 2048;
constexpr uint32_t REF_SEG = 2u * SD_TILE_RAW - REF_HALO;
static_assert(REF_SEG % 64u == 0 && REF_SEG > REF_HALO, "segment geometry");
struct RefSeedArgs {
    const uint8_t *seq;
    uint64_t len;
    uint32_t n_seg;
    DevParams P;
    unsigned long long *seg_hash;  // segment s: entries [s * cap, s * cap + counts[s])
    uint32_t *seg_pos;
    uint32_t *seg_last;  // seeding variant 16 only (else nullptr): every minimizer's second position
    uint32_t cap;
    uint32_t *counts;    // minimizers of segment s (may exceed cap: the list is then incomplete and the host retries with room)
    uint32_t *queue;     // segments the fast seeder declined
    uint32_t *counters;  // [0] fast work, [1] queue length, [2] general work
    uint32_t force_general;
};

__global__ __launch_bounds__(64 * SEED_WAVES, MQ_SEED_MIN_WAVES) void seed_ref_kernel(const RefSeedArgs A) {
    __shared__ struct {
        SeedTables T;
        SeedLds SS[SEED_WAVES];
    } W;
    build_seed_tables(W.T, A.P.l, var_h32<true>(A.P));
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t wv = rdfirst(threadIdx.x >> 6);
    SeedLds &S = W.SS[wv];
    for (;;) {
        uint32_t s = 0;
        if (lane == 0) s = atomicAdd(&A.counters[0], 1u);
        s = rdfirst(s);
        if (s >= A.n_seg) break;
        const uint64_t a = (uint64_t)s * REF_SEG;
        const uint64_t rest = A.len - a;
        const uint32_t vlen = rest < (uint64_t)(REF_SEG + REF_HALO) ? (uint32_t)rest : REF_SEG + REF_HALO;
        SeedView V;
        V.first_prev = 4u;
        if (a > 0) {  // the base in front of the view, as the seeder's 2-bit code (A 0, C 1, T 2, G 3)
            uint32_t b = A.seq[a - 1];
            if (A.P.fold && b - 'a' < 26u) b -= 32u;
            V.first_prev = b == 'A' ? 0u : b == 'C' ? 1u : b == 'T' ? 2u : b == 'G' ? 3u : 4u;
        }
        V.elig_end = REF_SEG;
        V.pos_add = (uint32_t)a;
        V.more_after = a + vlen < A.len ? 1u : 0u;
        APre pre;
        const size_t at = (size_t)s * A.cap;
        const uint32_t cnt = A.force_general ? SD_NOT_FAST
                                             : seed_sequence_fast<0, true>(A.seq + a, vlen, A.P, W.T, S, A.seg_hash + at, A.seg_pos + at, A.cap, pre, false, V,
                                                                           A.seg_last ? A.seg_last + at : nullptr);
        if (lane == 0) {
            A.counts[s] = cnt;
            if (cnt == SD_NOT_FAST) A.queue[atomicAdd(&A.counters[1], 1u)] = s;
        }
        wave_sync();
    }
}

// the segments queued by seed_ref_kernel, through the general streaming seeder
__global__ __launch_bounds__(64) void seed_ref_general_kernel(const RefSeedArgs A) {
    __shared__ WaveLds S;
    const uint32_t lane = lane_id();
    const uint32_t nq = A.counters[1];
    for (;;) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(&A.counters[2], 1u);
        i = rdfirst(i);
        if (i >= nq) break;
        const uint32_t s = A.queue[i];
        const uint64_t a = (uint64_t)s * REF_SEG;
        const uint64_t b = a + REF_SEG < A.len ? a + REF_SEG : A.len;
        const size_t at = (size_t)s * A.cap;
        SoaListSink sink(A.seg_hash + at, A.seg_pos + at, A.seg_last ? A.seg_last + at : nullptr, A.cap);
        uint32_t mz_count = 0;
        seed_segment(A.seq, A.len, a, b, A.P, S, sink, mz_count);
        if (lane == 0) A.counts[s] = sink.written;
        wave_sync();
    }
}

// Stage 2: where every segment's list goes in the reference's dense list (exclusive scan of the counts, one workgroup), the total,
// and the segments whose list outgrew its region (their counts are the true ones: they are seeded again, straight into their place in
// the dense list, by seed_ref_redo_kernel).  info: [0] total, [1] number of such segments (zeroed by the host); over_queue: their numbers.
__global__ __launch_bounds__(1024) void scan_counts_kernel(const uint32_t *__restrict__ counts, uint32_t n_seg, uint32_t cap,
                                                           unsigned long long *__restrict__ seg_off, unsigned long long *__restrict__ info,
                                                           uint32_t *__restrict__ over_queue) {
    __shared__ unsigned long long part[1024];
    const uint32_t t = threadIdx.x, per = (n_seg + 1023u) / 1024u;
    const uint32_t lo = t * per < n_seg ? t * per : n_seg, hi = lo + per < n_seg ? lo + per : n_seg;
    unsigned long long sum = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = counts[i];
        if (c > cap) over_queue[atomicAdd(&info[1], 1ull)] = i;
        sum += c;
    }
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const unsigned long long v = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = part[t] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        seg_off[i] = run;
        run += counts[i];
    }
    if (t == 1023u) {
        seg_off[n_seg] = part[1023];
        info[0] = part[1023];
    }
}

// a segment whose list did not fit its region (a short-period tandem array can be far denser than 2 d): once more through the general
// seeder, into its exact-size place in the dense list
__global__ __launch_bounds__(64) void seed_ref_redo_kernel(const RefSeedArgs A, const uint32_t *__restrict__ over_queue, uint32_t n_over,
                                                           const unsigned long long *__restrict__ seg_off, unsigned long long *__restrict__ dense_hash,
                                                           uint32_t *__restrict__ dense_pos, uint32_t *__restrict__ dense_last) {
    __shared__ WaveLds S;
    for (uint32_t i = blockIdx.x; i < n_over; i += gridDim.x) {
        const uint32_t s = over_queue[i];
        const uint64_t a = (uint64_t)s * REF_SEG;
        const uint64_t b = a + REF_SEG < A.len ? a + REF_SEG : A.len;
        SoaListSink sink(dense_hash + seg_off[s], dense_pos + seg_off[s], dense_last ? dense_last + seg_off[s] : nullptr, A.counts[s]);
        uint32_t mz_count = 0;
        seed_segment(A.seq, A.len, a, b, A.P, S, sink, mz_count);
        wave_sync();
    }
}

// Stage 3: segment lists -> one dense ordered list
__global__ void compact_lists_kernel(const unsigned long long *__restrict__ seg_hash, const uint32_t *__restrict__ seg_pos, uint32_t cap,
                                     const uint32_t *__restrict__ counts, const unsigned long long *__restrict__ seg_off, uint32_t n_seg,
                                     unsigned long long *__restrict__ dense_hash, uint32_t *__restrict__ dense_pos,
                                     const uint32_t *__restrict__ seg_last, uint32_t *__restrict__ dense_last) {
    for (uint32_t s = blockIdx.x; s < n_seg; s += gridDim.x) {
        const uint32_t c = counts[s];
        if (c > cap) continue;  // seeded again into its place (seed_ref_redo_kernel)
        const size_t src = (size_t)s * cap;
        const unsigned long long dst = seg_off[s];
        for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
            dense_hash[dst + i] = seg_hash[src + i];
            dense_pos[dst + i] = seg_pos[src + i];
            if (dense_last) dense_last[dst + i] = seg_last[src + i];
        }
    }
}

// Stage 4: every k consecutive minimizers -> one reference k-min-mer (KminmersIterator; Entry::new_with_mer src/index.rs:57-58)
__global__ void ref_kminmers_kernel(const unsigned long long *__restrict__ dense_hash, const uint32_t *__restrict__ dense_pos, uint64_t n_mz, DevParams P,
                                    uint32_t ref_id, RefKmm *__restrict__ out, const uint32_t *__restrict__ dense_last) {
    const uint64_t n_kmm = n_mz - P.k + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_kmm; i += (uint64_t)gridDim.x * blockDim.x) {
        bool rev;
        auto get = [&](uint32_t j) { return (uint64_t)dense_hash[i + j]; };
        const bool re = var_rev_eq<true>(P);
        const bool fk = P.fast_kh != 0;
        const uint64_t key = P.k == 5u ? kminmer_hash_fixed<5>(get, rev, re, fk) : P.k == 7u ? kminmer_hash_fixed<7>(get, rev, re, fk)
                             : P.k == 8u ? kminmer_hash_fixed<8>(get, rev, re, fk) : kminmer_hash(P.k, get, rev, re, fk);
        RefKmm r;
        r.hash = key;
        r.start = dense_pos[i];
        r.end = dense_last ? dense_last[i + P.k - 1] : dense_pos[i + P.k - 1] + P.l - 1u;  // (seeding variant 16: the second position)
        r.offset = (uint32_t)i;
        r.id_rc = (ref_id << 1) | (rev ? 1u : 0u);
        out[i] = r;
    }
}

// Index::add_with_mer (src/index.rs:100-104) made order independent: the first claimant of a slot stores the entry; a key that is
// inserted again gets ENTRY_DUP set in its entry (the tombstone: "second insert => empty entry", is_empty <=> end == 0,
// src/index.rs:67-69, 94-104 -- an entry whose end is 0 to begin with is born with the bit, the reference cannot tell it from
// an empty one either).  Every field of the entry is written so that the order of the claimant's stores and a duplicate's flag does
// not matter: the table starts zeroed and id_rc is only ever OR-ed.  acc (device counters): [0] keys claimed, [1] keys that turned dead.
// Walks the probe sequence of mq_device.hpp (home slot, other way of the home bucket, following buckets).
__device__ __forceinline__ void table_insert(Bucket *__restrict__ table, uint64_t mask, unsigned long long key, const Entry &e, uint32_t times,
                                             uint32_t &n_claimed, uint32_t &n_dead) {
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
    // the claimant: one 8-byte store (start, end) and one 64-bit OR into the zeroed (offset, id_rc) pair; a duplicate: the flag alone
    unsigned long long bits = (unsigned long long)ENTRY_DUP << 32;
    if (won) {
        n_claimed++;
        *reinterpret_cast<unsigned long long *>(&table[b].pay[w].start) = (unsigned long long)e.start | ((unsigned long long)e.end << 32);
        const uint32_t idb = (e.id_rc & ~ENTRY_DUP) | ((times > 1u || e.end == 0u) ? ENTRY_DUP : 0u);
        bits = (unsigned long long)e.offset | ((unsigned long long)idb << 32);
    }
    const unsigned long long old = atomicOr(reinterpret_cast<unsigned long long *>(&table[b].pay[w].offset), bits);
    if (((bits >> 32) & ENTRY_DUP) && !((old >> 32) & ENTRY_DUP)) n_dead++;
}
__device__ __forceinline__ void flush_insert_counts(uint32_t n_claimed, uint32_t n_dead, unsigned long long *__restrict__ acc) {
    const uint32_t c = wave_sum_u32(n_claimed), d = wave_sum_u32(n_dead);
    if (lane_id() == 0) {
        if (c) atomicAdd(&acc[0], (unsigned long long)c);
        if (d) atomicAdd(&acc[1], (unsigned long long)d);
    }
}

__global__ void insert_kernel(const RefKmm *__restrict__ kmm, uint64_t n, Bucket *__restrict__ table, uint64_t mask, unsigned long long *__restrict__ acc) {
    uint32_t n_claimed = 0, n_dead = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const RefKmm r = kmm[i];
        Entry e;
        e.start = r.start;
        e.end = r.end;
        e.offset = r.offset;
        e.id_rc = r.id_rc;
        table_insert(table, mask, r.hash, e, 1u, n_claimed, n_dead);
    }
    flush_insert_counts(n_claimed, n_dead, acc);
}

// One pass over a finished table (mq_index_load's check of a file against its header; the build counts while it inserts):
// Index::get_count (src/index.rs:90-92) = live slots, the number of distinct keys, the largest reference id stored.
// acc: [0] live, [1] keys, [2] max ref id + 1 over occupied slots.
__global__ void count_kernel(const Bucket *__restrict__ table, uint64_t n_buckets_plus1, unsigned long long *__restrict__ acc) {
    unsigned long long live = 0, keys = 0, max_id1 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n_buckets_plus1; i += (uint64_t)gridDim.x * blockDim.x) {
        const Bucket &B = table[i >> 1];
        const uint32_t w = (uint32_t)i & 1u;
        const bool extra = (i >> 1) == n_buckets_plus1 - 1;
        const bool occupied = extra ? (w == 0 && B.claims != 0) : B.key[w] != 0;
        if (occupied) {
            keys++;
            const Entry e = B.pay[w];
            const unsigned long long id1 = (unsigned long long)((e.id_rc & ~ENTRY_DUP) >> 1) + 1ull;
            max_id1 = id1 > max_id1 ? id1 : max_id1;
            if (entry_live(e)) live++;
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
        const bool extra = (i >> 1) == n_buckets_plus1 - 1;
        const bool occupied = extra ? (w == 0 && B.claims != 0) : B.key[w] != 0;
        if (occupied) {
            const unsigned long long at = atomicAdd(cursor, 1ull);
            if (at < cap) {
                SavedSlot v;
                v.key = extra ? 0ull : B.key[w];
                v.e = B.pay[w];
                v.count = (v.e.id_rc & ENTRY_DUP) ? 2u : 1u;  // the file says "once" or "more than once" (a dead entry), as it always could have
                v.e.id_rc &= ~ENTRY_DUP;
                v.is_key0 = extra ? 1u : 0u;
                out[at] = v;
            }
        }
    }
}
// mq_index_load: saved slots back into an empty table (keys are distinct, so every insertion claims its slot); flags[0] is set
// when an entry cannot be what mq_index_save wrote (a reference id beyond the file's reference table, a key 0 outside its slot).
__global__ void unpack_slots_kernel(const SavedSlot *__restrict__ in, uint64_t n, Bucket *__restrict__ table, uint64_t mask, uint32_t max_id,
                                    uint32_t *__restrict__ flags) {
    uint32_t n_claimed = 0, n_dead = 0;  // not used here: mq_index_load counts the finished table against the file's header
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const SavedSlot v = in[i];
        if (v.count == 0 || (v.e.id_rc >> 1) > max_id || (v.is_key0 != 0) != (v.key == 0) || v.is_key0 > 1u) {
            atomicOr(flags, 1u);
            continue;
        }
        table_insert(table, mask, v.key, v.e, v.count, n_claimed, n_dead);
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
