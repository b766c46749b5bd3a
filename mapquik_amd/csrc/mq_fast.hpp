// mq_fast.hpp -- the fast seeding path (ACGT-only sequences that fit one LDS tile).
//
//   stage A  lanes own 16 consecutive raw bases (one 16-B load each, 1 KiB per wave row):
//            SWAR decode ASCII -> 2-bit codes, validity check with v_perm_b32, homopolymer compression through a
//            1024-entry LDS look-up (index = previous code + 4 codes), wave prefix sum, ds_or of the packed bits into
//            the tile's code stream in LDS.  By-products: HPC count at every 64-base block (LDS) and the run-head
//            bit mask (HBM scratch, read back only for the ~2 % selected positions).
//   stage B  lanes own contiguous chunks of HPC positions and ROLL ntHash over them:
//            fh' = rol(fh,1) ^ rol(h(out),l) ^ h(in),  rh' = ror(rh,1) ^ ror(hc(out),1) ^ rol(hc(in),l-1)
//            with one 16-entry LDS table indexed by (out,in) -> one ds_read_b128 per step.  Selected l-mers go to a
//            per-lane list in HBM scratch (worst-case sized: no overflow path).
//   stage C  64 minimizers at a time, in order: locate (lane, slot) by binary search over the lane prefix sums, fetch,
//            map the HPC index back to the raw position (block search + select on the head mask), hand to the sink.
// Anything else (non-ACGT bytes, sequences longer than the tile) takes the general streaming path in mq_device.hpp.
#pragma once
#include "mq_device.hpp"

namespace mq {

constexpr uint32_t FAST_CODES_CAP = 20480;                        // HPC codes per tile
constexpr uint32_t FAST_MAX_ROWS = 32;                            // raw rows of 1024 bases per tile
constexpr uint32_t FAST_MAX_D = 21;                               // odd, >= ceil(FAST_CODES_CAP / 1024)
constexpr uint32_t FAST_EM_PER_LANE = 16 * FAST_MAX_D;            // worst-case emissions of one lane
constexpr uint32_t FAST_EM_BYTES = 64 * FAST_EM_PER_LANE * 16;    // per wave
constexpr uint32_t FAST_HM_WORDS = FAST_MAX_ROWS * 64;            // per wave (uint32 each)
static_assert(FAST_CODES_DW * 16 >= FAST_CODES_CAP + FAST_EM_PER_LANE + 64 + 48, "code stream padding");
static_assert(FAST_CNT_N >= FAST_MAX_ROWS * 16 + 1, "cnt64 size");

// workgroup-shared look-up tables (built once per workgroup)
struct WgTables {
    uint4 roll[16];      // index out | in<<2 : {rol(h(out),l)^h(in) lo,hi ; ror(hc(out),1)^rol(hc(in),l-1) lo,hi}
    uint4 warm[4];       // index code        : {h(c) lo,hi ; rol(hc(c),l-1) lo,hi}
    uint16_t lut[1024];  // index prev | c0<<2 | c1<<4 | c2<<6 | c3<<8 : compacted codes (8 bits) | 2*count << 8
};

// 2-bit code = (ASCII >> 1) & 3 : A=0 C=1 T=2 G=3 ; complement = code ^ 2
__device__ __forceinline__ uint64_t fast_seed(uint32_t code) {
    return code == 0 ? 0x3c8bfbb395c60474ULL : code == 1 ? 0x3193c18562a02b4cULL : code == 2 ? 0x295549f54be24456ULL : 0x20323ed082572324ULL;
}

__device__ __forceinline__ void build_tables(WgTables &T, uint32_t l) {
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
        uint32_t prev = i & 3u, out = 0, n = 0;
        for (uint32_t m = 0; m < 4; ++m) {
            const uint32_t c = (i >> (2 + 2 * m)) & 3u;
            if (c != prev) {
                out |= c << (2 * n);
                n++;
            }
            prev = c;
        }
        T.lut[i] = (uint16_t)(out | ((2 * n) << 8));
    }
    if (threadIdx.x < 16) {
        const uint32_t o = threadIdx.x & 3u, in = threadIdx.x >> 2;
        const uint64_t f = rotl64(fast_seed(o), l) ^ fast_seed(in);
        const uint64_t r = rotr64(fast_seed(o ^ 2u), 1) ^ rotl64(fast_seed(in ^ 2u), l - 1u);
        T.roll[threadIdx.x] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
    }
    if (threadIdx.x < 4) {
        const uint64_t f = fast_seed(threadIdx.x);
        const uint64_t r = rotl64(fast_seed(threadIdx.x ^ 2u), l - 1u);
        T.warm[threadIdx.x] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
    }
}

typedef uint4 __attribute__((aligned(1))) uint4_unaligned;

__device__ __forceinline__ uint32_t ld_sc1_u32(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2-served: never a stale L1 line
}
__device__ __forceinline__ uint64_t ld_sc1_u64(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// pack 4 ASCII bases (one dword) into 8 bits of 2-bit codes; t = w & 0x06060606
__device__ __forceinline__ uint32_t pack4(uint32_t t) {
    const uint32_t x = t | (t << 6);
    const uint32_t y = x | (x << 12);
    return (y >> 19) & 0xFFu;
}

// ------------------------------------------------------------------ stage A
// Returns false when the sequence needs the general path (non-ACGT byte, too long for the tile).
__device__ __forceinline__ bool fast_stage_a(const uint8_t *__restrict__ seq, uint32_t len, bool use_hpc, const WgTables &T,
                                             WaveLds &S, uint32_t *__restrict__ hm_scratch, uint32_t &n_codes, uint32_t &n_blocks) {
    const uint32_t lane = lane_id();
    const uint32_t n_rows = (len + 1023u) >> 10;
    if (n_rows > FAST_MAX_ROWS) return false;
    for (uint32_t i = lane * 4u; i < FAST_CODES_DW; i += 256u) *reinterpret_cast<uint4 *>(&S.f.codes[i]) = make_uint4(0, 0, 0, 0);
    wave_sync();
    const uint32_t fill = (uint32_t)seq[len - 1] * 0x01010101u;
    uint32_t b2 = 0;  // bits written so far = 2 * codes
    uint32_t bad = 0;
    uint32_t carry_prev = 0;
    for (uint32_t r = 0; r < n_rows; ++r) {
        if (b2 > 2u * (FAST_CODES_CAP - 1024u)) return false;
        const uint32_t pos = (r << 10) + lane * 16u;
        uint32_t w0, w1, w2, w3;
        if (pos + 16u <= len) {
            const uint4 v = *reinterpret_cast<const uint4_unaligned *>(seq + pos);
            w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
        } else {
            // tail: bytes past the end repeat the last base (never a run head under HPC; masked without HPC)
            uint32_t ww[4] = {fill, fill, fill, fill};
            const uint32_t nv = pos < len ? len - pos : 0u;
#pragma unroll
            for (uint32_t j = 0; j < 15; ++j)
                if (j < nv) ww[j >> 2] = (ww[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | ((uint32_t)seq[pos + j] << (8 * (j & 3)));
            w0 = ww[0]; w1 = ww[1]; w2 = ww[2]; w3 = ww[3];
        }
        const uint32_t t0 = w0 & 0x06060606u, t1 = w1 & 0x06060606u, t2 = w2 & 0x06060606u, t3 = w3 & 0x06060606u;
        // selector bytes 0,2 pick from S1 ('A','C'), 4,6 from S0 ('T','G'): reconstructs the byte iff it was A/C/G/T
        constexpr uint32_t S1 = 0x00430041u, S0 = 0x00470054u;
        bad |= (__builtin_amdgcn_perm(S0, S1, t0) ^ w0) | (__builtin_amdgcn_perm(S0, S1, t1) ^ w1) |
               (__builtin_amdgcn_perm(S0, S1, t2) ^ w2) | (__builtin_amdgcn_perm(S0, S1, t3) ^ w3);
        const uint32_t p = pack4(t0) | (pack4(t1) << 8) | (pack4(t2) << 16) | (pack4(t3) << 24);
        uint32_t out, n2, hm;
        if (use_hpc) {
            uint32_t pc = (uint32_t)__shfl_up((int)(p >> 30), 1, 64);
            if (lane == 0) pc = r == 0 ? ((p & 3u) ^ 1u) : carry_prev;  // first base of the sequence is always a head
            const uint32_t q = (p << 2) | pc;
            const uint32_t e0 = T.lut[q & 0x3FFu], e1 = T.lut[(q >> 8) & 0x3FFu], e2 = T.lut[(q >> 16) & 0x3FFu], e3 = T.lut[p >> 22];
            out = e0 & 0xFFu;
            uint32_t sh = e0 >> 8;
            out |= (e1 & 0xFFu) << sh;
            sh += e1 >> 8;
            out |= (e2 & 0xFFu) << sh;
            sh += e2 >> 8;
            out |= (e3 & 0xFFu) << sh;
            n2 = sh + (e3 >> 8);
            const uint32_t d = p ^ q;
            hm = (d | (d >> 1)) & 0x55555555u;
        } else {
            const uint32_t nv = pos < len ? (len - pos < 16u ? len - pos : 16u) : 0u;
            const uint32_t m = nv >= 16u ? 0xFFFFFFFFu : ((1u << (2u * nv)) - 1u);
            out = p & m;
            n2 = 2u * nv;
            hm = 0x55555555u & m;
        }
        uint32_t incl = n2;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, dd, 64);
            if (lane >= (uint32_t)dd) incl += o;
        }
        const uint32_t bo = b2 + incl - n2;
        if (n2) {
            const uint32_t sh = bo & 31u;
            atomicOr(&S.f.codes[bo >> 5], out << sh);
            const uint32_t hi = sh ? out >> (32u - sh) : 0u;
            if (hi) atomicOr(&S.f.codes[(bo >> 5) + 1u], hi);
        }
        if ((lane & 3u) == 0) S.f.cnt64[r * 16u + (lane >> 2)] = (uint16_t)(bo >> 1);
        hm_scratch[r * 64u + lane] = hm;
        b2 += rdlane(incl, 63);
        carry_prev = rdlane(p, 63) >> 30;
    }
    n_blocks = n_rows * 16u;
    n_codes = b2 >> 1;
    if (lane == 0) S.f.cnt64[n_blocks] = (uint16_t)n_codes;
    wave_sync();
    return __ballot(bad != 0) == 0;
}

// ------------------------------------------------------------------ stage B
struct Hash2 {
    uint32_t flo, fhi, rlo, rhi;
    __device__ __forceinline__ void roll(const uint4 tv) {
        const uint32_t nfhi = __builtin_amdgcn_alignbit(fhi, flo, 31), nflo = __builtin_amdgcn_alignbit(flo, fhi, 31);  // rol 1
        const uint32_t nrlo = __builtin_amdgcn_alignbit(rhi, rlo, 1), nrhi = __builtin_amdgcn_alignbit(rlo, rhi, 1);    // ror 1
        flo = nflo ^ tv.x;
        fhi = nfhi ^ tv.y;
        rlo = nrlo ^ tv.z;
        rhi = nrhi ^ tv.w;
    }
};

// Rolls ntHash over windows [0, w_eff) of the tile's code stream.  Lane L owns windows [L*16d, (L+1)*16d).
// Selected windows are appended to the lane's list in em (entry e of lane L at em[e*64+L]): {hash lo, hash hi, j, 0}.
__device__ __forceinline__ uint32_t fast_stage_b(const WgTables &T, const WaveLds &S, const DevParams &P, uint32_t w_eff,
                                                 uint4 *__restrict__ em) {
    const uint32_t lane = lane_id();
    const uint32_t l = P.l;
    uint32_t d = (w_eff + 1023u) >> 10;
    d |= 1u;  // odd dword stride between lanes: conflict-free ds_read_b32 of the per-lane streams
    const uint32_t lc = 16u * d;
    const uint32_t start = lane * lc;
    const uint32_t nvalid = start < w_eff ? (w_eff - start < lc ? w_eff - start : lc) : 0u;
    const uint32_t base_dw = nvalid ? lane * d : 0u;
    const uint32_t bhi = (uint32_t)(P.bound >> 32);
    Hash2 h = {0, 0, 0, 0};
    // warm-up: the lane's first window, Horner form (fh = rol(fh,1)^h(c), rh = ror(rh,1)^rol(hc(c),l-1))
    for (uint32_t m0 = 0; m0 < l; m0 += 16u) {
        const uint32_t dw = S.f.codes[base_dw + (m0 >> 4)];
        const uint32_t cnt = l - m0 < 16u ? l - m0 : 16u;
        for (uint32_t m = 0; m < cnt; ++m) h.roll(T.warm[(dw >> (2u * m)) & 3u]);
    }
    const uint32_t in_dw = l >> 4, in_sh = 2u * (l & 15u);
    uint32_t e = 0;
    uint32_t prev_in = S.f.codes[base_dw + in_dw];
    for (uint32_t blk = 0; blk < d; ++blk) {
        const uint32_t ow = S.f.codes[base_dw + blk];
        const uint32_t nxt = S.f.codes[base_dw + blk + in_dw + 1u];
        const uint32_t iw = __builtin_amdgcn_alignbit(nxt, prev_in, in_sh);
        prev_in = nxt;
        // nibble m of xe / xo = out | in<<2 for step 2m / 2m+1
        const uint32_t xe = (ow & 0x33333333u) | ((iw & 0x33333333u) << 2);
        const uint32_t xo = ((ow >> 2) & 0x33333333u) | (iw & 0xCCCCCCCCu);
#pragma unroll
        for (uint32_t t = 0; t < 16; ++t) {
            const bool cand = h.fhi <= bhi || h.rhi <= bhi;
            if (__ballot(cand)) {
                const uint64_t F = ((uint64_t)h.fhi << 32) | h.flo, R = ((uint64_t)h.rhi << 32) | h.rlo;
                const uint32_t tt = 16u * blk + t;
                if (tt < nvalid && (F <= P.bound || R <= P.bound)) {
                    const uint64_t hv = F < R ? F : R;
                    em[e * 64u + lane] = make_uint4((uint32_t)hv, (uint32_t)(hv >> 32), start + tt, 0u);
                    e++;
                }
            }
            const uint32_t x = (t & 1u) ? xo : xe;
            const uint32_t nib = (x >> (4u * (t >> 1))) & 0xFu;
            h.roll(T.roll[nib]);
        }
    }
    return e;
}

// ------------------------------------------------------------------ stage C
// raw position (tile-relative) of the run head with HPC index j
__device__ __forceinline__ uint32_t fast_rawpos(const WaveLds &S, uint32_t n_blocks, const uint32_t *__restrict__ hm_scratch, uint32_t j) {
    uint32_t lo = 0, hi = n_blocks;  // largest block b with cnt64[b] <= j
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)S.f.cnt64[mid] <= j) lo = mid;
        else hi = mid;
    }
    uint32_t r = j - (uint32_t)S.f.cnt64[lo];
    const uint64_t *hp = reinterpret_cast<const uint64_t *>(hm_scratch + lo * 4u);
    const uint64_t m01 = ld_sc1_u64(hp), m23 = ld_sc1_u64(hp + 1);
    uint32_t mw = (uint32_t)m01, w = 0;
    uint32_t c = (uint32_t)__popc(mw);
    if (r >= c) {
        r -= c; mw = (uint32_t)(m01 >> 32); w = 1; c = (uint32_t)__popc(mw);
        if (r >= c) {
            r -= c; mw = (uint32_t)m23; w = 2; c = (uint32_t)__popc(mw);
            if (r >= c) { r -= c; mw = (uint32_t)(m23 >> 32); w = 3; }
        }
    }
    // r-th set bit of mw (bits sit at even positions)
    uint32_t bit = 0;
#pragma unroll
    for (uint32_t width = 16; width >= 2; width >>= 1) {
        const uint32_t half = (uint32_t)__popc(mw & ((1u << width) - 1u));
        if (r >= half) {
            r -= half;
            mw >>= width;
            bit += width;
        }
    }
    return lo * 64u + w * 16u + (bit >> 1);
}

template <class Sink>
__device__ __forceinline__ void fast_stage_c(WaveLds &S, Sink &sink, uint32_t &mz_count, uint32_t my_count, const uint4 *__restrict__ em,
                                             const uint32_t *__restrict__ hm_scratch, uint32_t n_blocks, uint32_t raw_base) {
    const uint32_t lane = lane_id();
    uint32_t incl = my_count;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, dd, 64);
        if (lane >= (uint32_t)dd) incl += o;
    }
    S.f.lane_prefix[lane] = incl - my_count;
    const uint32_t total = rdlane(incl, 63);
    if (lane == 0) S.f.lane_prefix[64] = total;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's scratch stores have reached L2
    wave_sync();
    for (uint32_t g0 = 0; g0 < total; g0 += 64u) {
        const uint32_t g = g0 + lane;
        if (g < total) {
            uint32_t lo = 0, hi = 64;  // largest L with prefix[L] <= g
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (S.f.lane_prefix[mid] <= g) lo = mid;
                else hi = mid;
            }
            const uint32_t ei = g - S.f.lane_prefix[lo];
            const uint64_t *rec = reinterpret_cast<const uint64_t *>(em + (ei * 64u + lo));
            const uint64_t hv = ld_sc1_u64(rec);
            const uint32_t j = (uint32_t)ld_sc1_u64(rec + 1);
            S.mz_hash[mz_count + lane] = hv;
            S.mz_pos[mz_count + lane] = raw_base + fast_rawpos(S, n_blocks, hm_scratch, j);
        }
        mz_count += total - g0 < 64u ? total - g0 : 64u;
        wave_sync();
        sink.on_minimizers(S, mz_count);
    }
}

// Whole sequence through the fast path.  Returns false (nothing emitted to the sink) if it does not qualify.
template <class Sink>
__device__ __forceinline__ bool fast_seed_sequence(const uint8_t *__restrict__ seq, uint32_t len, const DevParams &P, const WgTables &T,
                                                   WaveLds &S, Sink &sink, uint32_t &mz_count, uint4 *__restrict__ em,
                                                   uint32_t *__restrict__ hm_scratch) {
    uint32_t n_codes = 0, n_blocks = 0;
    if (!fast_stage_a(seq, len, P.use_hpc != 0, T, S, hm_scratch, n_codes, n_blocks)) return false;
    if (n_codes < P.l) return true;  // fewer compressed bases than one l-mer: no minimizers
    const uint32_t my = fast_stage_b(T, S, P, n_codes - P.l + 1u, em);
    fast_stage_c(S, sink, mz_count, my, em, hm_scratch, n_blocks, 0u);
    return true;
}

}  // namespace mq
