// mq_fast.hpp -- the fast seeding path (ACGT-only sequences; long ones are cut into LDS tiles).
//
//   stage A  super-rows of 4096 raw bases, every lane owns one 64-base block (four 16-B loads, the next super-row's four
//            already in flight): SWAR decode ASCII -> 2-bit codes (OR-merge + 4x4 transpose of 2-bit elements), validity
//            check with v_perm_b32, homopolymer compression through a 1024-entry LDS look-up (index = previous code +
//            4 codes), one wave prefix sum per super-row, ds_or of the packed bits into the tile's code stream in LDS.
//            By-products: compressed count at every 64-base block (LDS) and the run-head bit masks (HBM scratch, read
//            back only for the ~2 % selected positions).
//   stage B  lanes own contiguous chunks of compressed positions and ROLL ntHash over them:
//            fh' = rol(fh,1) ^ rol(h(out),l) ^ h(in),  rh' = ror(rh,1) ^ ror(hc(out),1) ^ rol(hc(in),l-1)
//            with one 16-entry LDS table indexed by (out,in) -> one ds_read_b128 per step, four look-ups in flight.
//            Selected l-mers are appended (ballot + mbcnt) to one dense per-wave list in HBM scratch, tagged (slot, lane).
//   stage C  the code stream is dead, its LDS becomes the ordered minimizer list: a record of lane L, slot e belongs at
//            lane_prefix[L] + e; raw position = block search in the per-block counts + select on the head mask; then the
//            sink hashes every k-min-mer, issues all home-slot probes together and runs the Match-run logic per 64.
// A sequence longer than one tile (32,768 bases / 20,480 compressed bases) is processed tile by tile: the last l-1
// compressed bases (codes + raw positions) and the last k-1 minimizers carry over.  Sequences with a non-ACGT byte take the
// general streaming path in mq_device.hpp.
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

// 16 ASCII bases (four dwords masked with 0x06060606: code<<1 in every byte) -> 32 bits, code j at bits 2j..2j+1.
// Merge the dwords so that byte b holds bases b, 4+b, 8+b, 12+b, then transpose the 4x4 matrix of 2-bit elements.
__device__ __forceinline__ uint32_t pack16(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {
    uint32_t v = (t0 >> 1) | (t1 << 1) | (t2 << 3) | (t3 << 5);
    uint32_t x = ((v >> 6) ^ v) & 0x00CC00CCu;
    v ^= x ^ (x << 6);
    x = ((v >> 12) ^ v) & 0x0000F0F0u;
    v ^= x ^ (x << 12);
    return v;
}

// ------------------------------------------------------------------ stage A
// A super-row is 4096 raw bases: every lane owns 64 consecutive bases = one 64-base block (four 16-B loads in flight,
// plus the next super-row's four: 8 KiB per wave outstanding), one wave prefix sum per super-row.
// One tile: raw bases [raw0, raw_end) where raw_end is the end of the sequence or the last super-row boundary the code
// stream (which starts with carry_n carried codes) and the block tables can hold.  Returns false on a non-ACGT byte.
__device__ __forceinline__ bool fast_stage_a(const uint8_t *__restrict__ seq, uint32_t len, uint32_t raw0, uint32_t carry_n,
                                             uint32_t &carry_prev, bool use_hpc, const WgTables &T, WaveLds &S,
                                             uint32_t *__restrict__ hm_scratch, uint32_t &n_codes, uint32_t &n_blocks, uint32_t &raw_end) {
    const uint32_t lane = lane_id();
    uint32_t n_sr = (len - raw0 + 4095u) >> 12;
    if (n_sr * 4u > FAST_MAX_ROWS) n_sr = FAST_MAX_ROWS / 4u;
    const uint32_t fill = (uint32_t)seq[len - 1] * 0x01010101u;
    // 16 bases at pos; bytes past the end repeat the last base (never a run head under HPC; masked without HPC)
    auto load_piece = [&](uint32_t pos) -> uint4 {
        if (pos + 16u <= len) return *reinterpret_cast<const uint4_unaligned *>(seq + pos);
        unsigned long long lo = ((unsigned long long)fill << 32) | fill, hi = lo;
        const uint32_t nv = pos < len ? len - pos : 0u;
        for (uint32_t j = 0; j < nv; ++j) {
            const unsigned long long b = seq[pos + j];
            if (j < 8u) lo = (lo & ~(0xFFull << (8u * j))) | (b << (8u * j));
            else hi = (hi & ~(0xFFull << (8u * (j - 8u)))) | (b << (8u * (j - 8u)));
        }
        return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    };
    uint4 nx0, nx1, nx2, nx3;
    {
        const uint32_t pos = raw0 + lane * 64u;
        nx0 = load_piece(pos);
        nx1 = load_piece(pos + 16u);
        nx2 = load_piece(pos + 32u);
        nx3 = load_piece(pos + 48u);
    }
    for (uint32_t i = lane * 4u; i < FAST_CODES_DW; i += 256u) *reinterpret_cast<uint4 *>(&S.f.codes[i]) = make_uint4(0, 0, 0, 0);
    wave_sync();
    if (lane < 4u && carry_n) S.f.codes[lane] = S.f.carry_codes[lane];  // the carried codes open the stream
    uint32_t b2 = 2u * carry_n;  // bits written so far = 2 * codes
    uint32_t bad = 0;
    uint32_t sr_done = n_sr;
    constexpr uint32_t S1 = 0x00430041u, S0 = 0x00470054u;  // v_perm pool: selector 0,2 -> 'A','C' ; 4,6 -> 'T','G'
    for (uint32_t sr = 0; sr < n_sr; ++sr) {
        const uint32_t pos = raw0 + (sr << 12) + lane * 64u;
        const uint4 c0 = nx0, c1 = nx1, c2 = nx2, c3 = nx3;
        if (sr + 1u < n_sr) {
            const uint32_t np = pos + 4096u;
            nx0 = load_piece(np);
            nx1 = load_piece(np + 16u);
            nx2 = load_piece(np + 32u);
            nx3 = load_piece(np + 48u);
        }
        uint32_t p[4];
        {
            const uint4 cc[4] = {c0, c1, c2, c3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t t0 = cc[j].x & 0x06060606u, t1 = cc[j].y & 0x06060606u, t2 = cc[j].z & 0x06060606u, t3 = cc[j].w & 0x06060606u;
                // reconstructs each byte iff it was A/C/G/T
                bad |= (__builtin_amdgcn_perm(S0, S1, t0) ^ cc[j].x) | (__builtin_amdgcn_perm(S0, S1, t1) ^ cc[j].y) |
                       (__builtin_amdgcn_perm(S0, S1, t2) ^ cc[j].z) | (__builtin_amdgcn_perm(S0, S1, t3) ^ cc[j].w);
                p[j] = pack16(t0, t1, t2, t3);
            }
        }
        uint32_t out[4], n2[4], hm[4];
        if (use_hpc) {
            uint32_t pc = (uint32_t)__shfl_up((int)(p[3] >> 30), 1, 64);
            if (lane == 0) pc = (sr == 0 && raw0 == 0) ? ((p[0] & 3u) ^ 1u) : carry_prev;  // first base of the sequence is always a head
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = (p[j] << 2) | pc;
                pc = p[j] >> 30;
                const uint32_t e0 = T.lut[q & 0x3FFu], e1 = T.lut[(q >> 8) & 0x3FFu], e2 = T.lut[(q >> 16) & 0x3FFu], e3 = T.lut[p[j] >> 22];
                uint32_t o = e0 & 0xFFu, sh = e0 >> 8;
                o |= (e1 & 0xFFu) << sh;
                sh += e1 >> 8;
                o |= (e2 & 0xFFu) << sh;
                sh += e2 >> 8;
                o |= (e3 & 0xFFu) << sh;
                out[j] = o;
                n2[j] = sh + (e3 >> 8);
                const uint32_t d = p[j] ^ q;
                hm[j] = (d | (d >> 1)) & 0x55555555u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t pp = pos + 16u * (uint32_t)j;
                const uint32_t nv = pp < len ? (len - pp < 16u ? len - pp : 16u) : 0u;
                const uint32_t m = nv >= 16u ? 0xFFFFFFFFu : ((1u << (2u * nv)) - 1u);
                out[j] = p[j] & m;
                n2[j] = 2u * nv;
                hm[j] = 0x55555555u & m;
            }
        }
        const uint32_t mine = n2[0] + n2[1] + n2[2] + n2[3];
        uint32_t incl = mine;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, dd, 64);
            if (lane >= (uint32_t)dd) incl += o;
        }
        const uint32_t total = rdlane(incl, 63);
        if (b2 + total > 2u * FAST_CODES_CAP) {  // does not fit: the tile ends before this super-row (sr > 0 always)
            sr_done = sr;
            break;
        }
        uint32_t bo = b2 + incl - mine;
        S.f.cnt64[sr * 64u + lane] = (uint16_t)(bo >> 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n2[j]) {
                const uint32_t sh = bo & 31u;
                atomicOr(&S.f.codes[bo >> 5], out[j] << sh);
                const uint32_t hi = sh ? out[j] >> (32u - sh) : 0u;
                if (hi) atomicOr(&S.f.codes[(bo >> 5) + 1u], hi);
            }
            bo += n2[j];
        }
        *reinterpret_cast<uint4 *>(hm_scratch + (sr * 64u + lane) * 4u) = make_uint4(hm[0], hm[1], hm[2], hm[3]);
        b2 += total;
        carry_prev = rdlane(p[3], 63) >> 30;
    }
    n_blocks = sr_done * 64u;
    raw_end = raw0 + (sr_done << 12) < len ? raw0 + (sr_done << 12) : len;
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
// Selected windows are appended to ONE dense per-wave list in HBM scratch (L2-resident in practice) in emission order:
// {hash lo, hash hi, j, slot<<6 | lane}; lane L's records, in slot order, are its windows in position order.
// Every dynamic instruction counts here (the kernel is issue-bound): the per-step test is min(fh.hi, rh.hi) <= hi(bound)
// (two VALU ops + one scalar branch); the exact 64-bit comparison runs only when some lane is a candidate.
__device__ __forceinline__ uint32_t fast_stage_b(const WgTables &T, const WaveLds &S, const DevParams &P, uint32_t w_eff,
                                                 uint4 *__restrict__ em) {
    const uint32_t lane = lane_id();
    const uint32_t l = P.l;
    const uint32_t d = (w_eff + 1023u) >> 10;  // 16-step blocks per lane
    const uint32_t lc = 16u * d;
    const uint32_t start = lane * lc;
    const uint32_t nvalid = start < w_eff ? (w_eff - start < lc ? w_eff - start : lc) : 0u;
    const uint32_t base_dw = nvalid ? lane * d : 0u;
    const uint32_t bhi = (uint32_t)(P.bound >> 32);
    uint32_t etag = lane;     // (slot in this lane's ordered list) << 6 | lane : where the record belongs in window order
    if (nvalid) {  // lanes beyond the last window sit out (exec-masked): they would only burn power
        Hash2 h = {0, 0, 0, 0};
        // warm-up: the lane's first window, Horner form (fh = rol(fh,1)^h(c), rh = ror(rh,1)^rol(hc(c),l-1))
        for (uint32_t m0 = 0; m0 < l; m0 += 16u) {
            const uint32_t dw = S.f.codes[base_dw + (m0 >> 4)];
            const uint32_t cnt = l - m0 < 16u ? l - m0 : 16u;
            for (uint32_t m = 0; m < cnt; ++m) h.roll(T.warm[(dw >> (2u * m)) & 3u]);
        }
        const uint32_t in_dw = l >> 4, in_sh = 2u * (l & 15u);
        // nibble m of xe / xo = out | in<<2 for step 2m / 2m+1 of a 16-step block
        auto mk_xe = [](uint32_t ow, uint32_t iw) { return (ow & 0x33333333u) | ((iw & 0x33333333u) << 2); };
        auto mk_xo = [](uint32_t ow, uint32_t iw) { return ((ow >> 2) & 0x33333333u) | (iw & 0xCCCCCCCCu); };
        auto nib = [](uint32_t xe, uint32_t xo, uint32_t s) { return (((s & 1u) ? xo : xe) >> (4u * (s >> 1))) & 0xFu; };
        uint32_t prev_in = S.f.codes[base_dw + in_dw];
        uint32_t xe, xo;
        {
            const uint32_t ow = S.f.codes[base_dw];
            const uint32_t nxt = S.f.codes[base_dw + in_dw + 1u];
            const uint32_t iw = __builtin_amdgcn_alignbit(nxt, prev_in, in_sh);
            prev_in = nxt;
            xe = mk_xe(ow, iw);
            xo = mk_xo(ow, iw);
        }
        // ring of table values for the next four steps: their LDS reads are in flight while a step tests / emits
        uint4 tv[4];
    #pragma unroll
        for (uint32_t s = 0; s < 4; ++s) tv[s] = T.roll[nib(xe, xo, s)];
        uint32_t wcount = 0;      // records written by the wave so far (wave-uniform): the list is dense, in emission order
        for (uint32_t blk = 0; blk < d; ++blk) {
            uint32_t xe_n, xo_n;
            {
                const uint32_t ow = S.f.codes[base_dw + blk + 1u];
                const uint32_t nxt = S.f.codes[base_dw + blk + in_dw + 2u];
                const uint32_t iw = __builtin_amdgcn_alignbit(nxt, prev_in, in_sh);
                prev_in = nxt;
                xe_n = mk_xe(ow, iw);
                xo_n = mk_xo(ow, iw);
            }
            const uint32_t t_lo = 16u * blk;
    #pragma unroll
            for (uint32_t t = 0; t < 16; ++t) {
                const bool cand = (h.fhi < h.rhi ? h.fhi : h.rhi) <= bhi;  // high words only: 2 VALU ops per step
                if (__ballot(cand)) {
                    const uint64_t F = ((uint64_t)h.fhi << 32) | h.flo, R = ((uint64_t)h.rhi << 32) | h.rlo;
                    const uint64_t hv = F < R ? F : R;
                    const bool ok = hv <= P.bound && t_lo + t < nvalid;
                    const uint64_t okm = __ballot(ok);
                    if (ok) {
                        em[wcount + mbcnt64(okm)] = make_uint4((uint32_t)hv, (uint32_t)(hv >> 32), start + t_lo + t, etag);
                        etag += 64u;
                    }
                    wcount += (uint32_t)__popcll(okm);
                }
                h.roll(tv[t & 3u]);
                tv[t & 3u] = (t + 4u < 16u) ? T.roll[nib(xe, xo, t + 4u)] : T.roll[nib(xe_n, xo_n, t + 4u - 16u)];
            }
            xe = xe_n;
            xo = xo_n;
        }
    }
    return etag >> 6;
}

// ------------------------------------------------------------------ stage C
constexpr uint32_t FAST_LIST_CAP = 432;  // ordered minimizers held at once, in the (now dead) code-stream LDS: 432 * 12 B
constexpr int FAST_NB = 7;               // ceil(FAST_LIST_CAP / 64)
static_assert(FAST_LIST_CAP * 3 <= FAST_CODES_DW, "minimizer list must fit the code stream region");
static_assert(FAST_NB * 64 >= FAST_LIST_CAP, "batches");

// r-th run head (0-based) of the 64-base block whose four 16-base head masks are m[0..3] (bits at even positions)
__device__ __forceinline__ uint32_t select_head(uint64_t m01, uint64_t m23, uint32_t r) {
    uint32_t mw = (uint32_t)m01, w = 0;
    uint32_t c = (uint32_t)__popc(mw);
    if (r >= c) {
        r -= c; mw = (uint32_t)(m01 >> 32); w = 1; c = (uint32_t)__popc(mw);
        if (r >= c) {
            r -= c; mw = (uint32_t)m23; w = 2; c = (uint32_t)__popc(mw);
            if (r >= c) { r -= c; mw = (uint32_t)(m23 >> 32); w = 3; }
        }
    }
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
    return w * 16u + (bit >> 1);
}

// Records [i0, i0 + 64*FAST_NB) of the wave's dense emission list -> their places in the ordered LDS list: a record of
// lane L, slot e belongs at lane_prefix[L] + e.  Only places in [g_lo, g_lo + n_new) are taken (one pass when the tile's
// minimizers fit the list, which is the normal case).  Every scratch load of the block is issued before its first use.
__device__ __forceinline__ void fast_gather(WaveLds &S, const uint4 *__restrict__ em, const uint32_t *__restrict__ hm_scratch,
                                            uint32_t n_blocks, uint32_t n_codes, uint32_t raw_base, uint32_t carry_n, uint32_t total,
                                            uint32_t i0, uint32_t g_lo, uint32_t n_new, unsigned long long *bh, uint32_t *bp, uint32_t at) {
    const uint32_t lane = lane_id();
    const float scale = (float)n_blocks / (float)n_codes;
    uint32_t jj[FAST_NB], dst[FAST_NB];
    {
        uint64_t hv[FAST_NB], jt[FAST_NB];
#pragma unroll
        for (int i = 0; i < FAST_NB; ++i) {
            hv[i] = 0;
            jt[i] = 0;
            const uint32_t ri = i0 + (uint32_t)i * 64u + lane;
            if (ri < total) {
                const uint64_t *rec = reinterpret_cast<const uint64_t *>(em + ri);
                hv[i] = ld_sc1_u64(rec);
                jt[i] = ld_sc1_u64(rec + 1);
            }
        }
#pragma unroll
        for (int i = 0; i < FAST_NB; ++i) {
            const uint32_t ri = i0 + (uint32_t)i * 64u + lane;
            jj[i] = (uint32_t)jt[i];
            dst[i] = 0xFFFFFFFFu;
            if (ri < total) {
                const uint32_t tag = (uint32_t)(jt[i] >> 32);
                const uint32_t g = S.f.lane_prefix[tag & 63u] + (tag >> 6);
                if (g - g_lo < n_new) {
                    dst[i] = at + (g - g_lo);
                    bh[dst[i]] = hv[i];
                }
            }
        }
    }
    uint32_t br[FAST_NB];  // block << 7 | rank of the head inside the block
    uint64_t m01[FAST_NB], m23[FAST_NB];
#pragma unroll
    for (int i = 0; i < FAST_NB; ++i) {
        br[i] = 0;
        m01[i] = m23[i] = 0;
        if (dst[i] != 0xFFFFFFFFu && jj[i] >= carry_n) {
            const uint32_t j = jj[i];
            uint32_t b = (uint32_t)((float)j * scale);  // interpolate, then walk to the block with cnt64[b] <= j < cnt64[b+1]
            if (b >= n_blocks) b = n_blocks - 1u;
            while ((uint32_t)S.f.cnt64[b] > j) --b;
            while ((uint32_t)S.f.cnt64[b + 1u] <= j) ++b;
            br[i] = (b << 7) | (j - (uint32_t)S.f.cnt64[b]);
            const uint64_t *hp = reinterpret_cast<const uint64_t *>(hm_scratch + b * 4u);
            m01[i] = ld_sc1_u64(hp);
            m23[i] = ld_sc1_u64(hp + 1);
        }
    }
#pragma unroll
    for (int i = 0; i < FAST_NB; ++i) {
        if (dst[i] != 0xFFFFFFFFu)
            bp[dst[i]] = jj[i] < carry_n ? S.f.carry_pos[jj[i]]  // a window that starts in the previous tile's last l-1 bases
                                         : raw_base + (br[i] >> 7) * 64u + select_head(m01[i], m23[i], br[i] & 127u);
    }
}

// Raw position of compressed base j of the current tile (used for the bases carried into the next tile)
__device__ __forceinline__ uint32_t fast_rawpos_one(const WaveLds &S, const uint32_t *__restrict__ hm_scratch, uint32_t n_blocks,
                                                    uint32_t n_codes, uint32_t raw_base, uint32_t carry_n, uint32_t j) {
    if (j < carry_n) return S.f.carry_pos[j];
    uint32_t b = (uint32_t)((float)j * ((float)n_blocks / (float)n_codes));
    if (b >= n_blocks) b = n_blocks - 1u;
    while ((uint32_t)S.f.cnt64[b] > j) --b;
    while ((uint32_t)S.f.cnt64[b + 1u] <= j) ++b;
    const uint64_t *hp = reinterpret_cast<const uint64_t *>(hm_scratch + b * 4u);
    return raw_base + b * 64u + select_head(ld_sc1_u64(hp), ld_sc1_u64(hp + 1), j - (uint32_t)S.f.cnt64[b]);
}

// mz_carry (in/out): minimizers carried from the previous tile, stashed in S.f.stash_hash / stash_pos.
template <class Sink>
__device__ __forceinline__ void fast_stage_c(WaveLds &S, const DevParams &P, Sink &sink, uint32_t my_count, const uint4 *__restrict__ em,
                                             const uint32_t *__restrict__ hm_scratch, uint32_t n_blocks, uint32_t n_codes, uint32_t raw_base,
                                             uint32_t carry_n, uint32_t &mz_carry, bool more_tiles, uint32_t stop_after = 0) {
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
    // the code stream is dead now: its LDS holds the ordered minimizer list
    unsigned long long *bh = reinterpret_cast<unsigned long long *>(&S.f.codes[0]);
    uint32_t *bp = &S.f.codes[2u * FAST_LIST_CAP];
    uint32_t carry = mz_carry;
    if (lane < carry) {
        bh[lane] = S.f.stash_hash[lane];
        bp[lane] = S.f.stash_pos[lane];
    }
    uint32_t have = carry;
    for (uint32_t g_lo = 0; g_lo < total;) {
        const uint32_t room = FAST_LIST_CAP - carry;
        const uint32_t n_new = total - g_lo < room ? total - g_lo : room;
        for (uint32_t i0 = 0; i0 < total; i0 += 64u * (uint32_t)FAST_NB)
            fast_gather(S, em, hm_scratch, n_blocks, n_codes, raw_base, carry_n, total, i0, g_lo, n_new, bh, bp, carry);
        wave_sync();
        have = carry + n_new;
        if (stop_after != 3u) sink.template consume_list<FAST_NB>(bh, bp, have);
        g_lo += n_new;
        if (g_lo < total) {  // more chunks: the last k-1 minimizers open the next chunk's windows
            const uint32_t c = P.k - 1u < have ? P.k - 1u : have;
            unsigned long long th = 0;
            uint32_t tp = 0;
            if (lane < c) {
                th = bh[have - c + lane];
                tp = bp[have - c + lane];
            }
            wave_sync();
            if (lane < c) {
                bh[lane] = th;
                bp[lane] = tp;
            }
            wave_sync();
            carry = c;
            have = c;
        }
    }
    if (more_tiles) {  // the last k-1 minimizers seen so far open the next tile's k-min-mers
        const uint32_t c = P.k - 1u < have ? P.k - 1u : have;
        if (lane < c) {
            S.f.stash_hash[lane] = bh[have - c + lane];
            S.f.stash_pos[lane] = bp[have - c + lane];
        }
        mz_carry = c;
        wave_sync();
    }
}

// true iff every byte of seq[from, len) is one of A C G T
__device__ __forceinline__ bool fast_all_acgt(const uint8_t *__restrict__ seq, uint32_t from, uint32_t len) {
    const uint32_t lane = lane_id();
    constexpr uint32_t S1 = 0x00430041u, S0 = 0x00470054u;
    uint32_t bad = 0;
    for (uint32_t pos = from + lane * 16u; pos < len; pos += 1024u) {
        if (pos + 16u <= len) {
            const uint4 v = *reinterpret_cast<const uint4_unaligned *>(seq + pos);
            bad |= (__builtin_amdgcn_perm(S0, S1, v.x & 0x06060606u) ^ v.x) | (__builtin_amdgcn_perm(S0, S1, v.y & 0x06060606u) ^ v.y) |
                   (__builtin_amdgcn_perm(S0, S1, v.z & 0x06060606u) ^ v.z) | (__builtin_amdgcn_perm(S0, S1, v.w & 0x06060606u) ^ v.w);
        } else {
            for (uint32_t q = pos; q < len; ++q) {
                const uint32_t b = seq[q];
                bad |= (b != 'A' && b != 'C' && b != 'G' && b != 'T') ? 1u : 0u;
            }
        }
    }
    return __ballot(bad != 0) == 0;
}

// Whole sequence through the fast path, tile by tile.  Returns false (nothing emitted to the sink) if it does not qualify.
// TIMING (diagnostic builds only): tacc[0..2] += cycles spent in stages A, B, C.
template <class Sink, bool TIMING = false>
__device__ __forceinline__ bool fast_seed_sequence(const uint8_t *__restrict__ seq, uint32_t len, const DevParams &P, const WgTables &T,
                                                   WaveLds &S, Sink &sink, uint32_t &mz_count, uint4 *__restrict__ em,
                                                   uint32_t *__restrict__ hm_scratch, unsigned long long *tacc = nullptr,
                                                   uint32_t stop_after = 0) {
    (void)mz_count;
    const uint32_t lane = lane_id();
    uint32_t raw0 = 0, carry_n = 0, carry_prev = 0, mz_carry = 0;
    bool rest_checked = false;
    while (raw0 < len) {
        uint32_t n_codes = 0, n_blocks = 0, raw_end = 0;
        const unsigned long long t0 = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
        const bool ok = fast_stage_a(seq, len, raw0, carry_n, carry_prev, P.use_hpc != 0, T, S, hm_scratch, n_codes, n_blocks, raw_end);
        const unsigned long long t1 = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
        if (TIMING) tacc[0] += t1 - t0;
        if (!ok) return false;  // only ever in the first tile: later tiles were pre-checked below before anything was emitted
        const bool more = raw_end < len;
        if (more && !rest_checked) {
            if (!fast_all_acgt(seq, raw_end, len)) return false;
            rest_checked = true;
        }
        if (stop_after == 1u) return true;
        if (n_codes >= P.l) {
            const uint32_t my = fast_stage_b(T, S, P, n_codes - P.l + 1u, em);
            const unsigned long long t2 = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
            if (TIMING) tacc[1] += t2 - t1;
            if (stop_after == 2u) return true;
            // the bases carried into the next tile: read their codes and raw positions before stage C reuses the code stream
            uint32_t new_cn = 0, cpos = 0, ccode = 0;
            if (more) {
                new_cn = P.l - 1u;  // n_codes >= l here
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < new_cn) cpos = fast_rawpos_one(S, hm_scratch, n_blocks, n_codes, raw0, carry_n, n_codes - new_cn + lane);
                if (lane < 4u) {
                    const uint32_t sb = 2u * (n_codes - new_cn) + 32u * lane;
                    ccode = __builtin_amdgcn_alignbit(S.f.codes[(sb >> 5) + 1u], S.f.codes[sb >> 5], sb & 31u);
                }
            }
            fast_stage_c(S, P, sink, my, em, hm_scratch, n_blocks, n_codes, raw0, carry_n, mz_carry, more, stop_after);
            if (more) {
                if (lane < new_cn) S.f.carry_pos[lane] = cpos;
                if (lane < 4u) S.f.carry_codes[lane] = ccode;  // bits beyond 2*new_cn are cleared by the mask below
                wave_sync();
                if (lane < 4u) {
                    const uint32_t keep = 2u * new_cn > 32u * lane ? 2u * new_cn - 32u * lane : 0u;
                    S.f.carry_codes[lane] = keep >= 32u ? ccode : (ccode & ((1u << keep) - 1u));
                }
                carry_n = new_cn;
            }
            if (TIMING) tacc[2] += __builtin_amdgcn_s_memtime() - t2;
        } else if (more) {
            // fewer compressed bases than one l-mer so far (a very long homopolymer run): carry every one of them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t cpos = 0, ccode = 0;
            if (lane < n_codes) cpos = fast_rawpos_one(S, hm_scratch, n_blocks, n_codes, raw0, carry_n, lane);
            if (lane < 4u) ccode = S.f.codes[lane];
            wave_sync();
            if (lane < n_codes) S.f.carry_pos[lane] = cpos;
            if (lane < 4u) S.f.carry_codes[lane] = ccode;
            carry_n = n_codes;
        }
        wave_sync();
        raw0 = raw_end;
    }
    return true;
}

}  // namespace mq
