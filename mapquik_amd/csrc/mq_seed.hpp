// mq_seed.hpp -- the fast seeder: an ACGT-only sequence -> its ordered minimizer list (hash, raw position) in HBM.
//
// One wave per sequence, tile by tile (12,288 raw bases = three super-rows of 64 bases per lane); per-wave LDS 8.1 KB, so that
// workgroups of 8 waves (78.1 KB with the shared tables) run 2 to a CU.  Everything lives in LDS between the stages; nothing but
// the final list goes to HBM.  (Measured on MI355X: 2 super-rows 920, 3 super-rows 950 Gbases/s; 5 or 6 waves per SIMD at
// 2 super-rows bought nothing, the fused kernel needs 128 VGPRs for its map phase.)
//   stage A  decode + homopolymer compression: SWAR ASCII -> 2-bit codes (OR-merge + 4x4 transpose of 2-bit elements),
//            validity by v_perm_b32 reconstruction, a 1024-entry LDS look-up (previous code + 4 codes -> compacted codes,
//            count, run-head bits), one DPP prefix sum per super-row, ds_or of the packed codes into the tile's code stream.
//            By-products kept in LDS: run-head bit mask (1 bit per raw base) and the compressed count at every 64-base block.
//   stage B  lanes own contiguous chunks of ceil(windows / 64) compressed positions and ROLL ntHash over them in a rotating
//            frame (the hashes are kept un-rotated: four XORs per step with pre-rotated roll terms from an 8 KB table, the
//            rotation sits in the ds_read_b128's immediate offset), four look-ups in flight.
//            The per-step test is min(fh.hi, rh.hi) <= hi(bound) (one v_alignbit per strand to un-rotate the high word); its
//            outcome is shifted into a per-lane flag word (v_cmp + v_addc: no branch, no scalar work, nothing stored); the
//            flag words go to LDS once per 32 steps, word-major (conflict-free).
//   stage R  every lane lists the windows of its own candidates at their places (prefix sum of the lanes' counts); then lane =
//            candidate, in position order: the window's two hashes computed again from the code stream, four bases per look-up
//            (256-entry table; l = 31: all eight look-ups in flight at once); exact 64-bit test; raw position = the block
//            (per-block counts of three candidate blocks compared at once) + select on its 64-bit head mask; then {hash, pos}
//            go to the sequence's region of the minimizer buffer.
// A sequence with a byte other than A C G T (or a candidate that passes the high-word test but not the exact one) is
// handed to the general streaming seeder (mq_device.hpp) through a queue; both produce the same list.
#pragma once
#include "mq_device.hpp"

namespace mq {

constexpr uint32_t SD_SR_RAW = 4096;                       // raw bases per super-row: 64 per lane
#ifndef MQ_SD_MAX_SR
#define MQ_SD_MAX_SR 3
#endif
constexpr uint32_t SD_MAX_SR = MQ_SD_MAX_SR;               // super-rows per tile
constexpr uint32_t SD_TILE_RAW = SD_SR_RAW * SD_MAX_SR;    // 12288
constexpr uint32_t SD_BLOCKS = SD_TILE_RAW / 64;           // 64-base blocks per tile
constexpr uint32_t SD_CODES_MAX = SD_TILE_RAW + MAX_L - 1; // codes of one tile incl. the carried l-1 (no compression at all)
constexpr uint32_t SD_LC_MAX = (SD_CODES_MAX + 63) / 64;   // windows per lane
constexpr uint32_t SD_FLAG_WORDS = (SD_LC_MAX + 31) / 32;  // 32-step flag words per lane
constexpr uint32_t SD_CODES_DW = 4 * ((63 * SD_LC_MAX + 16 * ((SD_LC_MAX + 15) / 16) + MAX_L + 32) / 64 + 1);  // packed 2-bit codes + zero read-ahead padding (whole uint4s)
static_assert((63 * SD_LC_MAX + 16 * ((SD_LC_MAX + 15) / 16) + MAX_L + 32) / 16 < SD_CODES_DW, "code stream read-ahead padding");
static_assert(2 * SD_CODES_MAX / 32 + 2 < SD_CODES_DW, "stage A writes three dwords from a piece pair's first");
#ifndef MQ_SD_OWNER_CAP
#define MQ_SD_OWNER_CAP 256
#endif
// 1: a tile's first super-row is requested by the previous tile's stage A and rides through its stages B and R in 16 registers
// (no tile opens with an exposed HBM round trip); 0: every tile requests its own (builds that must stay under 64 registers
// and have eight waves per SIMD to cover the wait)
#ifndef MQ_SD_CROSS_PREFETCH
#define MQ_SD_CROSS_PREFETCH 1
#endif
constexpr bool SD_CROSS_PREFETCH = MQ_SD_CROSS_PREFETCH != 0;
constexpr uint32_t SD_OWNER_CAP = MQ_SD_OWNER_CAP;                     // candidates listed per round of stage R (four lane-batches)
static_assert(SD_BLOCKS <= 256, "block_of holds block numbers in a byte");

// workgroup-shared look-up tables (built once per workgroup)
struct SeedTables {
    uint4 rot[32 * 16];  // index s*16 + (out | in<<2), s = 0..31 : {ror(A,s) lo,hi ; rol(B,s) lo,hi} with the roll terms
                         //   A = rol(h(out),l)^h(in), B = ror(hc(out),1)^rol(hc(in),l-1); rotation by s+32 = the same entry, halves swapped (32-bit hashes: {ror32(A,s), rol32(B,s)} twice)
    uint4 rem[64];       // index c0 | c1<<2 | c2<<4 : the last l mod 4 Horner steps at once (same form as quad, l mod 4 codes; entry 0 unused when l mod 4 = 0)
    uint4 quad[256];     // index c0 | c1<<2 | c2<<4 | c3<<6 : four Horner steps at once, {F4 lo,hi ; R4 lo,hi} with
                         //   F4 = rol(h(c0),3)^rol(h(c1),2)^rol(h(c2),1)^h(c3),  R4 = ror(X0,3)^ror(X1,2)^ror(X2,1)^X3,  X = rol(hc(c),l-1)
    uint16_t lut[1024];  // index prev | c0<<2 | c1<<4 | c2<<6 | c3<<8 : compacted codes (8 bits) | 2*count << 8 | head bits << 12
};

// per-wave LDS of the seed kernel's fast path
constexpr uint32_t SD_BLOCK_OF_BYTES = (SD_CODES_MAX / 64 + 3 + 15) / 16 * 16;
struct SeedLds {
    uint32_t codes[SD_CODES_DW];                 // the tile's 2-bit code stream (carried l-1 codes first)
    uint8_t block_of[SD_BLOCK_OF_BYTES];         // block_of[c]: the 64-base block that holds code 64 c (the walk to a code's block starts there);
                                                 // directly behind codes: stage A zeroes both with one loop (block_of[0] must be 0)
    unsigned long long heads[SD_BLOCKS];         // bit b of heads[k]: raw base 64k + b of the tile is a run head
    uint16_t cnt[SD_BLOCKS + 4];                 // index in the code stream of block k's first run head; cnt[n_blocks] = n_codes
    uint32_t flagw[SD_FLAG_WORDS * 64];          // stage B -> stage R: bit t of flagw[w * 64 + L] <=> step 32 w + t of lane L is a candidate (word-major:
                                                 // a wave's write or read of one word index touches 64 consecutive dwords, conflict-free)
    uint32_t carry_codes[4];                     // codes carried into the next tile (<= 63)
    uint32_t carry_pos[64];                      // their raw positions
    uint16_t cand[SD_OWNER_CAP];                 // stage R: candidate (in position order, one round of them) -> its window (code index in the tile)
};

// 2-bit code = (ASCII >> 1) & 3 : A=0 C=1 T=2 G=3 ; complement = code ^ 2
// h32 (seeding variant 4): the low half of the seed in both halves of the word -- every table entry, every hash and the bound then
// have that form, and the 64-bit code below computes the 32-bit ntHash (rotations mod 32) in each half (DevParams::variant)
__device__ __forceinline__ uint64_t seed_of(uint32_t code, bool h32) {
    const uint64_t s = code == 0 ? 0x3c8bfbb395c60474ULL : code == 1 ? 0x3193c18562a02b4cULL : code == 2 ? 0x295549f54be24456ULL : 0x20323ed082572324ULL;
    return h32 ? dup_low(s) : s;
}

__device__ __forceinline__ void build_seed_tables(SeedTables &T, uint32_t l, bool h32) {
    auto seed_of = [h32](uint32_t code) { return mq::seed_of(code, h32); };
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
        uint32_t prev = i & 3u, out = 0, n = 0, hb = 0;
        for (uint32_t m = 0; m < 4; ++m) {
            const uint32_t c = (i >> (2 + 2 * m)) & 3u;
            if (c != prev) {
                out |= c << (2 * n);
                n++;
                hb |= 1u << m;
            }
            prev = c;
        }
        T.lut[i] = (uint16_t)(out | ((2 * n) << 8) | (hb << 12));
    }
    for (uint32_t i = threadIdx.x; i < 32u * 16u; i += blockDim.x) {
        const uint32_t sft = i >> 4, o = i & 3u, in = (i >> 2) & 3u;
        const uint64_t f = rotr64(rotl64(seed_of(o), l) ^ seed_of(in), sft);
        const uint64_t r = rotl64(rotr64(seed_of(o ^ 2u), 1) ^ rotl64(seed_of(in ^ 2u), l - 1u), sft);
        // (32-bit hashes, seeding variant 4: both halves of f and of r are the 32-bit term; stage_b_block32 reads {f, r} as ONE 8-byte entry)
        T.rot[i] = h32 ? make_uint4((uint32_t)f, (uint32_t)r, (uint32_t)f, (uint32_t)r) : make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
    }
    for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x) {
        uint64_t f = 0, r = 0;
        for (uint32_t m = 0; m < (l & 3u); ++m) {
            const uint32_t c = (i >> (2 * m)) & 3u;
            f = rotl64(f, 1) ^ seed_of(c);
            r = rotr64(r, 1) ^ rotl64(seed_of(c ^ 2u), l - 1u);
        }
        T.rem[i] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
    }
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        uint64_t f = 0, r = 0;
        for (uint32_t m = 0; m < 4; ++m) {
            const uint32_t c = (i >> (2 * m)) & 3u;
            f = rotl64(f, 1) ^ seed_of(c);
            r = rotr64(r, 1) ^ rotl64(seed_of(c ^ 2u), l - 1u);
        }
        T.quad[i] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
    }
}


typedef uint4 __attribute__((aligned(1))) uint4_unaligned;

__device__ __forceinline__ uint64_t ld_sc1_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2-served: never a stale L1 line
}
__device__ __forceinline__ uint32_t ld_sc1_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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

// both strands' rolling ntHash of one window, as four dwords; roll() = one step with the table value of (out, in)
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

// ------------------------------------------------------------------ stage A
// 16 bases at pos.  The load itself never branches: its address is clamped to len - 16 (len >= 16 is the caller's precondition), so a
// piece that reaches past the end comes back as the 16 bytes that END at len -- never a byte beyond the sequence -- and fix_piece()
// shifts it into place (bytes past the end repeat the last base: never a run head under HPC; masked without HPC).  Only a sequence's
// last super-row has such pieces, so the fix-up sits behind a wave-uniform test; everywhere else a piece costs a v_min and a load
// (it used to cost a divergent if / else per piece: ~14 instructions and two register copies, a ninth of stage A).
__device__ __forceinline__ uint4 load_piece(const uint8_t *__restrict__ seq, uint32_t len, uint32_t pos) {
    const uint32_t p = pos < len - 16u ? pos : len - 16u;
    return *reinterpret_cast<const uint4_unaligned *>(seq + p);
}
// v = load_piece(seq, len, pos) of a piece with pos + 16 > len: the piece as stage A wants it
__device__ __forceinline__ uint4 fix_piece(const uint4 v, uint32_t len, uint32_t pos) {
    if (pos + 16u <= len) return v;
    const uint32_t fill = (v.w >> 24) * 0x01010101u;  // seq[len - 1] is the last of the 16 bytes that end at len
    const uint32_t nv = pos < len ? len - pos : 0u;   // valid bytes: 0..15
    const uint32_t s = 16u - nv;                      // shift the 32-byte value [v, fill...] right by s bytes
    uint32_t w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;
    if (s & 4u) { w0 = w1; w1 = w2; w2 = w3; w3 = fill; }
    if (s & 8u) { w0 = w2; w1 = w3; w2 = fill; w3 = fill; }
    if (s & 16u) { w0 = fill; w1 = fill; w2 = fill; w3 = fill; }
    const uint32_t sb = s & 3u;
    return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sb), __builtin_amdgcn_alignbyte(w2, w1, sb),
                      __builtin_amdgcn_alignbyte(w3, w2, sb), __builtin_amdgcn_alignbyte(fill, w3, sb));
}

// The first super-row of a tile, on its way from HBM: requested by whoever ran before the tile's stage A (the previous tile's
// stage A for the stages B and R in between, the previous read's seed phase for its map phase), so that no tile opens with an
// exposed HBM round trip.  64 bases per lane = 16 registers.
struct APre {
    uint4 nx0, nx1, nx2, nx3;
};
// A window of a longer sequence seeded as a sequence of its own (the reference segments of the index build): what differs from a read.
struct SeedView {
    uint32_t first_prev;  // 2-bit code of the base in front of the view; 4: none, or not A C G T (the view's first base is a run head)
    uint32_t elig_end;    // only minimizers whose l-mer starts at a view-relative raw position < elig_end are listed (a multiple of 64)
    uint32_t pos_add;     // added to every listed position: the view's offset in the sequence
    uint32_t more_after;  // the sequence goes on behind the view: the bases from elig_end on must hold l - 1 run heads (else: declined)
};
__device__ __forceinline__ void stage_a_request(const uint8_t *__restrict__ seq, uint32_t len, uint32_t raw0, APre &pre) {
    const uint32_t pos = raw0 + lane_id() * 64u;
    pre.nx0 = load_piece(seq, len, pos);
    pre.nx1 = load_piece(seq, len, pos + 16u);
    pre.nx2 = load_piece(seq, len, pos + 32u);
    pre.nx3 = load_piece(seq, len, pos + 48u);
}

// One tile: raw bases [raw0, raw_end), raw_end = min(raw0 + SD_TILE_RAW, len); the code stream opens with carry_n carried codes.
// pre holds the tile's first super-row (stage_a_request) and leaves with the next tile's when the sequence goes on.
// Returns false on a byte other than A C G T.
__device__ __forceinline__ bool seed_stage_a(const uint8_t *__restrict__ seq, uint32_t len, uint32_t raw0, uint32_t carry_n,
                                             uint32_t &carry_prev, bool use_hpc, bool fold, const SeedTables &T, SeedLds &S, uint32_t &n_codes,
                                             uint32_t &n_blocks, uint32_t &raw_end, APre &pre, bool first_is_head = true) {
    const uint32_t lane = lane_id();
    uint32_t n_sr = (len - raw0 + SD_SR_RAW - 1u) / SD_SR_RAW;
    if (n_sr > SD_MAX_SR) n_sr = SD_MAX_SR;
    const uint32_t tile_end = raw0 + n_sr * SD_SR_RAW;  // >= len in the sequence's last tile
    uint4 &nx0 = pre.nx0, &nx1 = pre.nx1, &nx2 = pre.nx2, &nx3 = pre.nx3;
    if (!SD_CROSS_PREFETCH && raw0 != 0) stage_a_request(seq, len, raw0, pre);
    static_assert(offsetof(SeedLds, block_of) == sizeof(uint32_t) * SD_CODES_DW && SD_CODES_DW % 4 == 0, "block_of is zeroed with the code stream");
    // block_of[0] = 0 with the rest: codes [carry_n, 64): the walk starts at block 0 (whose range may begin after code 0)
    for (uint32_t i = lane * 4u; i < SD_CODES_DW + SD_BLOCK_OF_BYTES / 4u; i += 256u)
        *reinterpret_cast<uint4 *>(reinterpret_cast<uint32_t *>(&S) + i) = make_uint4(0, 0, 0, 0);
    wave_sync();
    if (lane < 4u && carry_n) S.codes[lane] = S.carry_codes[lane];  // the carried codes open the stream
    uint32_t b2 = 2u * carry_n;  // bits written so far = 2 * codes
    uint32_t bad = 0;
    constexpr uint32_t S1 = 0x00430041u, S0 = 0x00470054u;  // v_perm pool: selector 0,2 -> 'A','C' ; 4,6 -> 'T','G'
    for (uint32_t sr = 0; sr < n_sr; ++sr) {
        const uint32_t pos = raw0 + sr * SD_SR_RAW + lane * 64u;
        const bool more_sr = sr + 1u < n_sr || (SD_CROSS_PREFETCH && tile_end < len);  // the tile's last super-row requests the next tile's first
        if (raw0 + (sr + 1u) * SD_SR_RAW > len) {  // the sequence ends inside this super-row (wave-uniform, once per sequence): pieces past the end
            nx0 = fix_piece(nx0, len, pos);
            nx1 = fix_piece(nx1, len, pos + 16u);
            nx2 = fix_piece(nx2, len, pos + 32u);
            nx3 = fix_piece(nx3, len, pos + 48u);
        }
#ifdef MQ_STAGE_A_SPLIT
        if (sr == 0) mq_clk(raw0 == 0 ? 10 : 13);  // what precedes the loop (code stream zeroed, carried codes)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        mq_clk(sr == 0 ? (raw0 == 0 ? 0 : 14) : 12);  // the wait for the super-row's bases
#endif
        uint32_t p[4];
        // decode piece j, then send the load of the NEXT super-row's piece j into the registers just freed: 16 registers of
        // bases in flight plus the piece being decoded, instead of two whole super-rows.  (When nothing follows the load is made all
        // the same, from a clamped address: a handful of wasted loads per sequence instead of a branch and register copies per piece.)
        (void)more_sr;
        auto decode = [&](uint4 &nx, uint32_t j) {
            const uint32_t t0 = nx.x & 0x06060606u, t1 = nx.y & 0x06060606u, t2 = nx.z & 0x06060606u, t3 = nx.w & 0x06060606u;
            // reconstructs each byte iff it was A/C/G/T
            bad |= (__builtin_amdgcn_perm(S0, S1, t0) ^ nx.x) | (__builtin_amdgcn_perm(S0, S1, t1) ^ nx.y) |
                   (__builtin_amdgcn_perm(S0, S1, t2) ^ nx.z) | (__builtin_amdgcn_perm(S0, S1, t3) ^ nx.w);
            p[j] = pack16(t0, t1, t2, t3);
            nx = load_piece(seq, len, pos + SD_SR_RAW + 16u * j);
        };
        decode(nx0, 0);
        decode(nx1, 1);
        decode(nx2, 2);
        decode(nx3, 3);
        uint32_t out[4], n2[4], hb[4];
        if (use_hpc) {
            // code of the base before this lane's block: the previous lane's last base (DPP wave_shr:1), lane 0 takes the carry
            uint32_t pc = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(p[3] >> 30), 0x138, 0xf, 0xf, false);
            // first base of the sequence is always a head (a view's: unless the base in front of it is the same, carried in carry_prev)
            if (lane == 0) pc = (sr == 0 && raw0 == 0 && first_is_head) ? ((p[0] & 3u) ^ 1u) : carry_prev;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = (p[j] << 2) | pc;
                pc = p[j] >> 30;
                const uint32_t e0 = T.lut[q & 0x3FFu], e1 = T.lut[(q >> 8) & 0x3FFu], e2 = T.lut[(q >> 16) & 0x3FFu], e3 = T.lut[p[j] >> 22];
                uint32_t o = e0 & 0xFFu, sh = (e0 >> 8) & 15u;
                o |= (e1 & 0xFFu) << sh;
                sh += (e1 >> 8) & 15u;
                o |= (e2 & 0xFFu) << sh;
                sh += (e2 >> 8) & 15u;
                o |= (e3 & 0xFFu) << sh;
                out[j] = o;
                n2[j] = sh + ((e3 >> 8) & 15u);
                hb[j] = (e0 >> 12) | ((e1 >> 12) << 4) | ((e2 >> 12) << 8) | ((e3 >> 12) << 12);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t pp = pos + 16u * (uint32_t)j;
                const uint32_t nv = pp < len ? (len - pp < 16u ? len - pp : 16u) : 0u;
                const uint32_t m = nv >= 16u ? 0xFFFFFFFFu : ((1u << (2u * nv)) - 1u);
                out[j] = p[j] & m;
                n2[j] = 2u * nv;
                hb[j] = nv >= 16u ? 0xFFFFu : ((1u << nv) - 1u);
            }
        }
        const uint32_t mine = n2[0] + n2[1] + n2[2] + n2[3];
        const uint32_t incl = wave_incl_scan_u32(mine);
        const uint32_t total = rdlane(incl, 63);
        uint32_t bo = b2 + incl - mine;
        S.cnt[sr * 64u + lane] = (uint16_t)(bo >> 1);
        {  // a block holds at most 64 codes, so at most one multiple of 64 falls into its code range [c0, c1)
            const uint32_t c0 = bo >> 1, c1 = (bo + mine) >> 1, m = (c0 + 63u) & ~63u;
            if (m < c1) S.block_of[m >> 6] = (uint8_t)(sr * 64u + lane);
        }
        S.heads[sr * 64u + lane] = (unsigned long long)(hb[0] | (hb[1] << 16)) | ((unsigned long long)(hb[2] | (hb[3] << 16)) << 32);
        // The lane's codes go into the stream two pieces at a time: the pair joined in registers (<= 64 bits), shifted to its bit offset
        // (<= 96 bits) and OR-ed into three dwords, whatever their content -- a zero costs an LDS operation, a test costs a branch.
        // (Piece by piece, each with its two conditional ORs, this was 76 instructions a super-row; now 30.)  The dwords behind the
        // stream's end that this may touch are padding (SD_CODES_DW) and receive zeros only.
        auto put_pair = [&](uint32_t o_a, uint32_t n_a, uint32_t o_b, uint32_t bit) {
            const uint64_t v = (uint64_t)o_a | ((uint64_t)o_b << n_a);  // n_a <= 32
            const uint32_t sh = bit & 31u;
            const uint64_t x = (uint64_t)(uint32_t)v << sh, y = (v >> 32) << sh;
            uint32_t *dst = &S.codes[bit >> 5];
            atomicOr(dst, (uint32_t)x);
            atomicOr(dst + 1, (uint32_t)(x >> 32) | (uint32_t)y);
            atomicOr(dst + 2, (uint32_t)(y >> 32));
        };
        put_pair(out[0], n2[0], out[1], bo);
        put_pair(out[2], n2[2], out[3], bo + n2[0] + n2[1]);
        b2 += total;
        carry_prev = rdlane(p[3], 63) >> 30;
#ifdef MQ_STAGE_A_SPLIT
        mq_clk(15);
#endif
    }
    n_blocks = n_sr * 64u;
    raw_end = raw0 + n_sr * SD_SR_RAW < len ? raw0 + n_sr * SD_SR_RAW : len;
    n_codes = b2 >> 1;
    if (lane == 0) S.cnt[n_blocks] = (uint16_t)n_codes;
    wave_sync();
    // bad holds byte ^ its reconstruction from the 2-bit code: zero for A C G T; exactly 0x20 for a c g t (accepted when folding)
    return __ballot((bad & (fold ? 0xDFDFDFDFu : 0xFFFFFFFFu)) != 0) == 0;
}

// ------------------------------------------------------------------ stage B
// Both strands' ntHash of the l-mer that starts at code index a of the tile's code stream, from scratch: Horner form
// (fh = rol(fh,1)^h(c), rh = ror(rh,1)^rol(hc(c),l-1)), four bases per table look-up, the last l mod 4 in one look-up of their own.
// L known at compile time: every look-up of the window is in flight before the first is used, and nothing loops.
template <uint32_t L>
__device__ __forceinline__ Hash2 window_hash_fixed(const SeedTables &T, const SeedLds &S, uint32_t a) {
    constexpr uint32_t NDW = (L + 15u) / 16u, NQ = L / 4u, REM = L & 3u;
    uint32_t c[NDW + 1];
#pragma unroll
    for (uint32_t m = 0; m <= NDW; ++m) c[m] = S.codes[(a >> 4) + m];
    uint32_t dw[NDW];
#pragma unroll
    for (uint32_t m = 0; m < NDW; ++m) dw[m] = __builtin_amdgcn_alignbit(c[m + 1], c[m], 2u * (a & 15u));  // 16 codes from a + 16 m
    uint4 tv[NQ + 1];
#pragma unroll
    for (uint32_t q = 0; q < NQ; ++q) tv[q] = T.quad[(dw[q >> 2] >> (8u * (q & 3u))) & 0xFFu];
    if (REM) tv[NQ] = T.rem[(dw[NQ >> 2] >> (8u * (NQ & 3u))) & ((1u << (2u * REM)) - 1u)];
    Hash2 h = {tv[0].x, tv[0].y, tv[0].z, tv[0].w};
#pragma unroll
    for (uint32_t q = 1; q < NQ; ++q) {
        const uint32_t nfhi = __builtin_amdgcn_alignbit(h.fhi, h.flo, 28), nflo = __builtin_amdgcn_alignbit(h.flo, h.fhi, 28);  // rol 4
        const uint32_t nrlo = __builtin_amdgcn_alignbit(h.rhi, h.rlo, 4), nrhi = __builtin_amdgcn_alignbit(h.rlo, h.rhi, 4);    // ror 4
        h.flo = nflo ^ tv[q].x;
        h.fhi = nfhi ^ tv[q].y;
        h.rlo = nrlo ^ tv[q].z;
        h.rhi = nrhi ^ tv[q].w;
    }
    if (REM) {
        const uint32_t nfhi = __builtin_amdgcn_alignbit(h.fhi, h.flo, 32u - REM), nflo = __builtin_amdgcn_alignbit(h.flo, h.fhi, 32u - REM);
        const uint32_t nrlo = __builtin_amdgcn_alignbit(h.rhi, h.rlo, REM), nrhi = __builtin_amdgcn_alignbit(h.rlo, h.rhi, REM);
        h.flo = nflo ^ tv[NQ].x;
        h.fhi = nfhi ^ tv[NQ].y;
        h.rlo = nrlo ^ tv[NQ].z;
        h.rhi = nrhi ^ tv[NQ].w;
    }
    return h;
}

// The same for l = 4 NQ + r, NQ known at compile time, r = l mod 4 at run time (wave-uniform): the NQ quad look-ups and the
// remainder's are in flight together; nothing loops.
template <uint32_t NQ>
__device__ __forceinline__ Hash2 window_hash_q(const SeedTables &T, const SeedLds &S, uint32_t l, uint32_t a) {
    constexpr uint32_t NDW = (4u * NQ + 3u + 15u) / 16u;
    const uint32_t rem = l & 3u;
    uint32_t c[NDW + 1];
#pragma unroll
    for (uint32_t m = 0; m <= NDW; ++m) c[m] = S.codes[(a >> 4) + m];
    uint32_t dw[NDW];
#pragma unroll
    for (uint32_t m = 0; m < NDW; ++m) dw[m] = __builtin_amdgcn_alignbit(c[m + 1], c[m], 2u * (a & 15u));
    uint4 tv[NQ + 1];
#pragma unroll
    for (uint32_t q = 0; q < NQ; ++q) tv[q] = T.quad[(dw[q >> 2] >> (8u * (q & 3u))) & 0xFFu];
    tv[NQ] = T.rem[(dw[NQ >> 2] >> (8u * (NQ & 3u))) & ((1u << (2u * rem)) - 1u)];  // entry 0 (unused) when rem == 0
    Hash2 h = {tv[0].x, tv[0].y, tv[0].z, tv[0].w};
#pragma unroll
    for (uint32_t q = 1; q < NQ; ++q) {
        const uint32_t nfhi = __builtin_amdgcn_alignbit(h.fhi, h.flo, 28), nflo = __builtin_amdgcn_alignbit(h.flo, h.fhi, 28);  // rol 4
        const uint32_t nrlo = __builtin_amdgcn_alignbit(h.rhi, h.rlo, 4), nrhi = __builtin_amdgcn_alignbit(h.rlo, h.rhi, 4);    // ror 4
        h.flo = nflo ^ tv[q].x;
        h.fhi = nfhi ^ tv[q].y;
        h.rlo = nrlo ^ tv[q].z;
        h.rhi = nrhi ^ tv[q].w;
    }
    if (rem) {
        const uint32_t nfhi = __builtin_amdgcn_alignbit(h.fhi, h.flo, 32u - rem), nflo = __builtin_amdgcn_alignbit(h.flo, h.fhi, 32u - rem);
        const uint32_t nrlo = __builtin_amdgcn_alignbit(h.rhi, h.rlo, rem), nrhi = __builtin_amdgcn_alignbit(h.rlo, h.rhi, rem);
        h.flo = nflo ^ tv[NQ].x;
        h.fhi = nfhi ^ tv[NQ].y;
        h.rlo = nrlo ^ tv[NQ].z;
        h.rhi = nrhi ^ tv[NQ].w;
    }
    return h;
}

__device__ __forceinline__ Hash2 window_hash(const SeedTables &T, const SeedLds &S, uint32_t l, uint32_t a) {
    if (l == 31u) return window_hash_fixed<31>(T, S, a);  // the reference's default l (src/main.rs: -l 31) and every BASELINE configuration
    // the other window lengths the reference's scripts use (example/run_ecoli.sh:26 -l 16, the second pass's -l 14, the l sweep of
    // experiments/figure-k-l): 12 <= l < 32 by the number of whole quads; anything else through the loop below
    switch (l >> 2) {
        case 3: return window_hash_q<3>(T, S, l, a);
        case 4: return window_hash_q<4>(T, S, l, a);
        case 5: return window_hash_q<5>(T, S, l, a);
        case 6: return window_hash_q<6>(T, S, l, a);
        case 7: return window_hash_q<7>(T, S, l, a);
        default: break;
    }
    Hash2 h = {0, 0, 0, 0};
    for (uint32_t m0 = 0; m0 < l; m0 += 16u) {
        const uint32_t d = (a + m0) >> 4;
        const uint32_t dw = __builtin_amdgcn_alignbit(S.codes[d + 1u], S.codes[d], 2u * ((a + m0) & 15u));  // 16 codes from a + m0
        const uint32_t cnt = l - m0 < 16u ? l - m0 : 16u;
        const uint32_t nq = cnt >> 2;
        for (uint32_t q = 0; q < nq; ++q) {
            const uint4 tv = T.quad[(dw >> (8u * q)) & 0xFFu];
            const uint32_t nfhi = __builtin_amdgcn_alignbit(h.fhi, h.flo, 28), nflo = __builtin_amdgcn_alignbit(h.flo, h.fhi, 28);  // rol 4
            const uint32_t nrlo = __builtin_amdgcn_alignbit(h.rhi, h.rlo, 4), nrhi = __builtin_amdgcn_alignbit(h.rlo, h.rhi, 4);    // ror 4
            h.flo = nflo ^ tv.x;
            h.fhi = nfhi ^ tv.y;
            h.rlo = nrlo ^ tv.z;
            h.rhi = nrhi ^ tv.w;
        }
        const uint32_t r = cnt & 3u;  // non-zero in the last chunk only (= l mod 4)
        if (r) {
            const uint4 tv = T.rem[(dw >> (8u * nq)) & ((1u << (2u * r)) - 1u)];
            const uint32_t nfhi = __builtin_amdgcn_alignbit(h.fhi, h.flo, 32u - r), nflo = __builtin_amdgcn_alignbit(h.flo, h.fhi, 32u - r);
            const uint32_t nrlo = __builtin_amdgcn_alignbit(h.rhi, h.rlo, r), nrhi = __builtin_amdgcn_alignbit(h.rlo, h.rhi, r);
            h.flo = nflo ^ tv.x;
            h.fhi = nfhi ^ tv.y;
            h.rlo = nrlo ^ tv.z;
            h.rhi = nrhi ^ tv.w;
        }
    }
    return h;
}

// One 16-step block of stage B in the ROTATING FRAME.  With F_t, R_t the two hashes of the lane's window t, the lane keeps
//   G_t = ror(F_t, t),  H_t = rol(R_t, t):   G_{t+1} = G_t ^ ror(A_t, t+1),  H_{t+1} = H_t ^ rol(B_t, t+1)
// (A_t, B_t = the roll terms of (out, in) at step t), so a step updates the hashes with four XORs and no rotate; the rotated
// terms come from T.rot with the rotation in the read's immediate offset (t mod 64 is a compile-time constant: PH = block
// number mod 4).  Only the density test needs un-rotated bits, and only the high words: one v_alignbit per strand with a
// constant amount.  LIM_CHECK: the lane's last, partial block (steps >= lim are not taken).
template <int PH, bool LIM_CHECK>
__device__ __forceinline__ uint32_t stage_b_block(const SeedTables &T, uint32_t &glo, uint32_t &ghi, uint32_t &hlo, uint32_t &hhi, uint4 (&tv)[4],
                                                  uint32_t xe, uint32_t xo, uint32_t xe_n, uint32_t xo_n, uint32_t bhi, uint32_t lim) {
    // byte offset of entry (out | in<<2) of step s within a rotation's 16 entries = nibble (s >> 1) of xe / xo, times 16: ONE
    // SDWA instruction per step (byte select + "& 0xF0" for a high nibble, byte select + "<< 4" cut to a byte for a low one)
    // where shift + and take two
    auto off16 = [](uint32_t xe_, uint32_t xo_, uint32_t s) -> uint32_t {
        const uint32_t x = (s & 1u) ? xo_ : xe_, m = s >> 1;
        uint32_t r;
        if (m & 1u) {
            switch (m >> 1) {
                case 0: r = x & 0xF0u; break;
                case 1: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(x), "s"(0xF0u)); break;
                case 2: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(x), "s"(0xF0u)); break;
                default: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(x), "s"(0xF0u)); break;
            }
        } else {
            switch (m >> 1) {
                case 0: asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(x), "s"(4u)); break;
                case 1: asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(x), "s"(4u)); break;
                case 2: asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(x), "s"(4u)); break;
                default: asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(x), "s"(4u)); break;
            }
        }
        return r;
    };
    auto rot_at = [&T](uint32_t s4, uint32_t off) { return *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(&T.rot[s4 * 16u]) + off); };
    uint32_t fbits = 0;  // step t of the block ends up at bit 15 - t
#pragma unroll
    for (uint32_t t = 0; t < 16; ++t) {
        if (!LIM_CHECK || t < lim) {
            constexpr uint32_t dummy = 0;
            (void)dummy;
            const uint32_t TT = 16u * (uint32_t)PH + t;  // step number mod 64 (compile-time after unrolling)
            // high words of F = rol(G, TT) and R = ror(H, TT)
            const uint32_t fh = TT == 0 ? ghi : TT < 32 ? __builtin_amdgcn_alignbit(ghi, glo, 32u - TT) : TT == 32 ? glo : __builtin_amdgcn_alignbit(glo, ghi, 64u - TT);
            const uint32_t rh = TT == 0 ? hhi : TT < 32 ? __builtin_amdgcn_alignbit(hlo, hhi, TT) : TT == 32 ? hlo : __builtin_amdgcn_alignbit(hhi, hlo, TT - 32u);
            const uint32_t mhi = fh < rh ? fh : rh;
            const uint4 e = tv[t & 3u];
            // fbits = 2 * fbits + (mhi <= bhi): v_cmp into vcc, v_addc with vcc as carry-in (high words only; the exact test runs in
            // stage R) -- with the step's four xors BETWEEN the two: back to back the pair costs a third of the step (an instruction that
            // reads vcc right behind the one that wrote it; tools/valu_enc.hip, profiles/r04_valu_enc_stepb.txt: 30 -> 22 cycles per step).
            // The xors are plain v_xor_b32 in asm anyway: left to itself the SLP vectoriser pairs the words and, for the swapped halves,
            // first materialises the swap with two v_pk_mov_b32 per step.
            if (((TT + 1u) & 63u) < 32u) {
                asm("v_cmp_ge_u32_e32 vcc, %10, %9\n\tv_xor_b32 %0, %0, %5\n\tv_xor_b32 %1, %1, %6\n\tv_xor_b32 %2, %2, %7\n\tv_xor_b32 %3, %3, %8\n\t"
                    "v_addc_co_u32_e32 %4, vcc, %4, %4, vcc"
                    : "+v"(glo), "+v"(ghi), "+v"(hlo), "+v"(hhi), "+v"(fbits) : "v"(e.x), "v"(e.y), "v"(e.z), "v"(e.w), "v"(mhi), "s"(bhi) : "vcc");
            } else {  // rotation by s + 32: the stored entry with its halves swapped
                asm("v_cmp_ge_u32_e32 vcc, %10, %9\n\tv_xor_b32 %0, %0, %5\n\tv_xor_b32 %1, %1, %6\n\tv_xor_b32 %2, %2, %7\n\tv_xor_b32 %3, %3, %8\n\t"
                    "v_addc_co_u32_e32 %4, vcc, %4, %4, vcc"
                    : "+v"(glo), "+v"(ghi), "+v"(hlo), "+v"(hhi), "+v"(fbits) : "v"(e.y), "v"(e.x), "v"(e.w), "v"(e.z), "v"(mhi), "s"(bhi) : "vcc");
            }
            // the look-up of step t + 4 (rotation (TT + 5) mod 64), in flight while the next steps run
            const uint32_t s4 = (TT + 5u) & 31u;
            tv[t & 3u] = (t + 4u < 16u) ? rot_at(s4, off16(xe, xo, t + 4u)) : rot_at(s4, off16(xe_n, xo_n, t + 4u - 16u));
        }
    }
    if (LIM_CHECK) fbits <<= 16u - lim;
    return fbits;
}

// Rolls ntHash over windows [0, w_eff) of the tile's code stream; lane L owns windows [L*lc, (L+1)*lc), lc = ceil(w_eff/64).
// The outcome of every step's high-word test goes into the lane's flag words (S.flagw, one 32-step word per store); nothing
// else is kept (stage R computes a candidate's hashes again).
// Steps past the last window (only the last active lane has them) may set bits too: stage R masks them.
__device__ __forceinline__ void seed_stage_b(const SeedTables &T, SeedLds &S, const DevParams &P, uint32_t w_eff) {
    const uint32_t lane = lane_id();
    const uint32_t l = P.l;
    const uint32_t lc = (w_eff + 63u) >> 6;
    const uint32_t s0 = lane * lc;
    const uint32_t bhi = (uint32_t)(P.bound >> 32);
    if (s0 < w_eff) {  // lanes beyond the last window sit out (exec-masked)
        const Hash2 h0 = window_hash(T, S, l, s0);  // the lane's first window: G_0 = F_0, H_0 = R_0
        uint32_t glo = h0.flo, ghi = h0.fhi, hlo = h0.rlo, hhi = h0.rhi;
        // nibble m of xe / xo = out | in<<2 for step 2m / 2m+1 of a 16-step block
        auto mk_xe = [](uint32_t ow, uint32_t iw) { return (ow & 0x33333333u) | ((iw & 0x33333333u) << 2); };
        auto mk_xo = [](uint32_t ow, uint32_t iw) { return ((ow >> 2) & 0x33333333u) | (iw & 0xCCCCCCCCu); };
        auto nib = [](uint32_t xe, uint32_t xo, uint32_t s) { return (((s & 1u) ? xo : xe) >> (4u * (s >> 1))) & 0xFu; };
        const uint32_t o_dw = s0 >> 4, o_sh = 2u * (s0 & 15u);
        const uint32_t i_dw = (s0 + l) >> 4, i_sh = 2u * ((s0 + l) & 15u);
        uint32_t prev_o = S.codes[o_dw], prev_i = S.codes[i_dw];
        uint32_t xe, xo;
        {
            const uint32_t no = S.codes[o_dw + 1u], ni = S.codes[i_dw + 1u];
            const uint32_t ow = __builtin_amdgcn_alignbit(no, prev_o, o_sh), iw = __builtin_amdgcn_alignbit(ni, prev_i, i_sh);
            prev_o = no;
            prev_i = ni;
            xe = mk_xe(ow, iw);
            xo = mk_xo(ow, iw);
        }
        // ring of table values for the next four steps (rotations 1..4): their LDS reads are in flight while a step tests and updates
        uint4 tv[4];
#pragma unroll
        for (uint32_t s = 0; s < 4; ++s) tv[s] = T.rot[(s + 1u) * 16u + nib(xe, xo, s)];
        const uint32_t nb = (lc + 15u) >> 4;
        // the code words of the block after block b, as nibbles (out | in<<2) of its even and odd steps
        auto next_x = [&](uint32_t b, uint32_t &xe_n, uint32_t &xo_n) {
            const uint32_t no = S.codes[o_dw + b + 2u], ni = S.codes[i_dw + b + 2u];
            const uint32_t ow = __builtin_amdgcn_alignbit(no, prev_o, o_sh), iw = __builtin_amdgcn_alignbit(ni, prev_i, i_sh);
            prev_o = no;
            prev_i = ni;
            xe_n = mk_xe(ow, iw);
            xo_n = mk_xo(ow, iw);
        };
        // flag word w of the lane (words a lane never writes are never looked at: stage R masks by the lane's window count)
        auto put = [&](uint32_t w_at, uint32_t v) { S.flagw[w_at * 64u + lane] = v; };
        // Whole 64-step groups: the four rotation phases of the frame back to back in straight-line code, so that the hash state, the
        // ring of table values and the code words stay where they are (a loop over single blocks with a switch over the phase
        // costs ~14 register moves per block at the merge points).
        uint32_t blk = 0;
        for (const uint32_t full = (lc >> 6) << 2; blk < full; blk += 4u) {
            uint32_t xe1, xo1, xe2, xo2, xe3, xo3, xe4, xo4;
            next_x(blk, xe1, xo1);
            const uint32_t b0 = stage_b_block<0, false>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe1, xo1, bhi, 16u);
            next_x(blk + 1u, xe2, xo2);
            const uint32_t b1 = stage_b_block<1, false>(T, glo, ghi, hlo, hhi, tv, xe1, xo1, xe2, xo2, bhi, 16u);
            put(blk >> 1, (__brev(b0) >> 16) | (__brev(b1) & 0xFFFF0000u));  // bit t <=> step t
            next_x(blk + 2u, xe3, xo3);
            const uint32_t b2 = stage_b_block<2, false>(T, glo, ghi, hlo, hhi, tv, xe2, xo2, xe3, xo3, bhi, 16u);
            next_x(blk + 3u, xe4, xo4);
            const uint32_t b3 = stage_b_block<3, false>(T, glo, ghi, hlo, hhi, tv, xe3, xo3, xe4, xo4, bhi, 16u);
            put((blk >> 1) + 1u, (__brev(b2) >> 16) | (__brev(b3) & 0xFFFF0000u));
            xe = xe4;
            xo = xo4;
        }
        // the last, incomplete group: up to four blocks, the last of them possibly partial
        uint32_t word = 0;  // the 32-step flag word being filled (two 16-step blocks)
        for (; blk < nb; ++blk) {
            uint32_t xe_n, xo_n;
            next_x(blk, xe_n, xo_n);
            const uint32_t lim = lc - 16u * blk;  // steps left (wave-uniform): the last block may be partial
            uint32_t fbits;
            if (lim >= 16u) {
                switch (blk & 3u) {
                    case 0: fbits = stage_b_block<0, false>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe_n, xo_n, bhi, 16u); break;
                    case 1: fbits = stage_b_block<1, false>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe_n, xo_n, bhi, 16u); break;
                    case 2: fbits = stage_b_block<2, false>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe_n, xo_n, bhi, 16u); break;
                    default: fbits = stage_b_block<3, false>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe_n, xo_n, bhi, 16u); break;
                }
            } else {
                switch (blk & 3u) {
                    case 0: fbits = stage_b_block<0, true>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe_n, xo_n, bhi, lim); break;
                    case 1: fbits = stage_b_block<1, true>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe_n, xo_n, bhi, lim); break;
                    case 2: fbits = stage_b_block<2, true>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe_n, xo_n, bhi, lim); break;
                    default: fbits = stage_b_block<3, true>(T, glo, ghi, hlo, hhi, tv, xe, xo, xe_n, xo_n, bhi, lim); break;
                }
            }
            if (blk & 1u) {
                put(blk >> 1, word | (__brev(fbits) & 0xFFFF0000u));
            } else {
                word = __brev(fbits) >> 16;
            }
            xe = xe_n;
            xo = xo_n;
        }
        if (nb & 1u) put(nb >> 1, word);
    }
}

// Stage B of seeding variant 4 (MQ_SEEDVAR_HASH32): ntHash on 32-bit words.  The tables hold the 64-bit form of the duplicated low seed
// halves (dup(x) = x | x << 32: rol64 of dup = dup of rol32), whose halves ARE the 32-bit roll terms: build_seed_tables stores an entry as
// {ror32(A32, s), rol32(B32, s)} twice; the rotating frame has period 32 (PH = block number mod 2), a step keeps one word per strand: 2 v_alignbit (F = rol32(G, t),
// R = ror32(H, t)), v_min, v_cmp ... 2 v_xor ... v_addc and the SDWA table offset: 8 VALU + 1 LDS where the 64-bit frame takes 10 + 1.
// Same flags as seed_stage_b on the duplicated words (dup(a) <= dup(b) <=> a <= b), which the variant ran on before.
template <int PH, bool LIM_CHECK>
__device__ __forceinline__ uint32_t stage_b_block32(const SeedTables &T, uint32_t &g, uint32_t &h, uint2 (&tv)[4], uint32_t xe, uint32_t xo, uint32_t xe_n,
                                                    uint32_t xo_n, uint32_t bhi, uint32_t lim) {
    auto off16 = [](uint32_t xe_, uint32_t xo_, uint32_t s) -> uint32_t {  // as in stage_b_block: byte offset of entry (out | in<<2) of step s
        const uint32_t x = (s & 1u) ? xo_ : xe_, m = s >> 1;
        uint32_t r;
        if (m & 1u) {
            switch (m >> 1) {
                case 0: r = x & 0xF0u; break;
                case 1: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(x), "s"(0xF0u)); break;
                case 2: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(x), "s"(0xF0u)); break;
                default: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(x), "s"(0xF0u)); break;
            }
        } else {
            switch (m >> 1) {
                case 0: asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(x), "s"(4u)); break;
                case 1: asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(x), "s"(4u)); break;
                case 2: asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(x), "s"(4u)); break;
                default: asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(x), "s"(4u)); break;
            }
        }
        return r;
    };
    auto rot_at = [&T](uint32_t s4, uint32_t off) {  // the entry's first half: one ds_read_b64, the rotation in its immediate offset
        return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(&T.rot[s4 * 16u]) + off);
    };
    uint32_t fbits = 0;  // step t of the block ends up at bit 15 - t
#pragma unroll
    for (uint32_t t = 0; t < 16; ++t) {
        if (!LIM_CHECK || t < lim) {
            const uint32_t TT = 16u * (uint32_t)PH + t;  // step number mod 32 (compile-time after unrolling)
            const uint32_t fh = TT == 0 ? g : __builtin_amdgcn_alignbit(g, g, 32u - TT);  // F = rol32(G, TT)
            const uint32_t rh = TT == 0 ? h : __builtin_amdgcn_alignbit(h, h, TT);        // R = ror32(H, TT)
            const uint32_t mhi = fh < rh ? fh : rh;
            const uint2 e = tv[t & 3u];
            asm("v_cmp_ge_u32_e32 vcc, %6, %5\n\tv_xor_b32 %0, %0, %3\n\tv_xor_b32 %1, %1, %4\n\tv_addc_co_u32_e32 %2, vcc, %2, %2, vcc"
                : "+v"(g), "+v"(h), "+v"(fbits) : "v"(e.x), "v"(e.y), "v"(mhi), "s"(bhi) : "vcc");
            const uint32_t s4 = (TT + 5u) & 31u;  // the look-up of step t + 4 (rotation (TT + 5) mod 32), in flight while the next steps run
            tv[t & 3u] = (t + 4u < 16u) ? rot_at(s4, off16(xe, xo, t + 4u)) : rot_at(s4, off16(xe_n, xo_n, t + 4u - 16u));
        }
    }
    if (LIM_CHECK) fbits <<= 16u - lim;
    return fbits;
}

__device__ __forceinline__ void seed_stage_b32(const SeedTables &T, SeedLds &S, const DevParams &P, uint32_t w_eff) {
    const uint32_t lane = lane_id();
    const uint32_t l = P.l;
    const uint32_t lc = (w_eff + 63u) >> 6;
    const uint32_t s0 = lane * lc;
    const uint32_t bhi = (uint32_t)(P.bound >> 32);  // = the 32-bit bound (P.bound is its duplicate)
    if (s0 < w_eff) {
        const Hash2 h0 = window_hash(T, S, l, s0);  // duplicated words: either half is the 32-bit hash
        uint32_t g = h0.flo, h = h0.rlo;
        auto mk_xe = [](uint32_t ow, uint32_t iw) { return (ow & 0x33333333u) | ((iw & 0x33333333u) << 2); };
        auto mk_xo = [](uint32_t ow, uint32_t iw) { return ((ow >> 2) & 0x33333333u) | (iw & 0xCCCCCCCCu); };
        auto nib = [](uint32_t xe, uint32_t xo, uint32_t s) { return (((s & 1u) ? xo : xe) >> (4u * (s >> 1))) & 0xFu; };
        const uint32_t o_dw = s0 >> 4, o_sh = 2u * (s0 & 15u);
        const uint32_t i_dw = (s0 + l) >> 4, i_sh = 2u * ((s0 + l) & 15u);
        uint32_t prev_o = S.codes[o_dw], prev_i = S.codes[i_dw];
        uint32_t xe, xo;
        {
            const uint32_t no = S.codes[o_dw + 1u], ni = S.codes[i_dw + 1u];
            const uint32_t ow = __builtin_amdgcn_alignbit(no, prev_o, o_sh), iw = __builtin_amdgcn_alignbit(ni, prev_i, i_sh);
            prev_o = no;
            prev_i = ni;
            xe = mk_xe(ow, iw);
            xo = mk_xo(ow, iw);
        }
        uint2 tv[4];
#pragma unroll
        for (uint32_t s = 0; s < 4; ++s) {
            const uint4 e = T.rot[(s + 1u) * 16u + nib(xe, xo, s)];
            tv[s] = make_uint2(e.x, e.y);
        }
        const uint32_t nb = (lc + 15u) >> 4;
        auto next_x = [&](uint32_t b, uint32_t &xe_n, uint32_t &xo_n) {
            const uint32_t no = S.codes[o_dw + b + 2u], ni = S.codes[i_dw + b + 2u];
            const uint32_t ow = __builtin_amdgcn_alignbit(no, prev_o, o_sh), iw = __builtin_amdgcn_alignbit(ni, prev_i, i_sh);
            prev_o = no;
            prev_i = ni;
            xe_n = mk_xe(ow, iw);
            xo_n = mk_xo(ow, iw);
        };
        auto put = [&](uint32_t w_at, uint32_t v) { S.flagw[w_at * 64u + lane] = v; };
        uint32_t blk = 0;
        for (const uint32_t full = (lc >> 5) << 1; blk < full; blk += 2u) {  // whole 32-step groups: the frame's two phases back to back, one flag word
            uint32_t xe1, xo1, xe2, xo2;
            next_x(blk, xe1, xo1);
            const uint32_t b0 = stage_b_block32<0, false>(T, g, h, tv, xe, xo, xe1, xo1, bhi, 16u);
            next_x(blk + 1u, xe2, xo2);
            const uint32_t b1 = stage_b_block32<1, false>(T, g, h, tv, xe1, xo1, xe2, xo2, bhi, 16u);
            put(blk >> 1, (__brev(b0) >> 16) | (__brev(b1) & 0xFFFF0000u));  // bit t <=> step t
            xe = xe2;
            xo = xo2;
        }
        uint32_t word = 0;  // the last, incomplete group: up to two blocks, the last of them possibly partial
        for (; blk < nb; ++blk) {
            uint32_t xe_n, xo_n;
            next_x(blk, xe_n, xo_n);
            const uint32_t lim = lc - 16u * blk;
            uint32_t fbits;
            if (lim >= 16u) {
                fbits = (blk & 1u) ? stage_b_block32<1, false>(T, g, h, tv, xe, xo, xe_n, xo_n, bhi, 16u) : stage_b_block32<0, false>(T, g, h, tv, xe, xo, xe_n, xo_n, bhi, 16u);
            } else {
                fbits = (blk & 1u) ? stage_b_block32<1, true>(T, g, h, tv, xe, xo, xe_n, xo_n, bhi, lim) : stage_b_block32<0, true>(T, g, h, tv, xe, xo, xe_n, xo_n, bhi, lim);
            }
            if (blk & 1u) {
                put(blk >> 1, word | (__brev(fbits) & 0xFFFF0000u));
            } else {
                word = __brev(fbits) >> 16;
            }
            xe = xe_n;
            xo = xo_n;
        }
        if (nb & 1u) put(nb >> 1, word);
    }
}

// ------------------------------------------------------------------ stage R
// r-th run head (0-based) of a 64-base block with head mask m
__device__ __forceinline__ uint32_t select_bit64(unsigned long long m, uint32_t r) {
    uint32_t mw = (uint32_t)m, bit = 0;
    const uint32_t c = (uint32_t)__popc(mw);
    if (r >= c) {
        r -= c;
        mw = (uint32_t)(m >> 32);
        bit = 32;
    }
#pragma unroll
    for (uint32_t width = 16; width >= 1; width >>= 1) {
        const uint32_t half = (uint32_t)__popc(mw & ((1u << width) - 1u));
        if (r >= half) {
            r -= half;
            mw >>= width;
            bit += width;
        }
    }
    return bit;
}

// Raw position of compressed base j of the current tile (a base of the previous tile's last l-1: from carry_pos), without a
// dependent walk: block_of gives the first candidate block, the counts and head masks of it and of the next two are read together
// (one LDS round trip after block_of's), the block is picked by comparison.  A lane whose base lies further on (blocks of very
// few run heads: long homopolymer runs) walks on from there, block by block.
__device__ __forceinline__ uint32_t seed_rawpos_batch(const SeedLds &S, uint32_t raw_base, uint32_t carry_n, uint32_t j) {
    const uint32_t cpos = S.carry_pos[j < 63u ? j : 63u];
    const uint32_t b0 = S.block_of[j >> 6];
    const uint32_t b1 = b0 + 1u < SD_BLOCKS ? b0 + 1u : SD_BLOCKS - 1u, b2 = b0 + 2u < SD_BLOCKS ? b0 + 2u : SD_BLOCKS - 1u;
    const uint32_t c0 = S.cnt[b0], c1 = S.cnt[b0 + 1u], c2 = S.cnt[b0 + 2u], c3 = S.cnt[b0 + 3u];
    const unsigned long long h0 = S.heads[b0], h1 = S.heads[b1], h2 = S.heads[b2];
    // cnt[n_blocks] = n_codes > j ends the walk before any stale count is looked at
    const bool s1 = c1 <= j, s2 = s1 && c2 <= j, s3 = s2 && c3 <= j;
    uint32_t b = b0 + (s1 ? 1u : 0u) + (s2 ? 1u : 0u);
    unsigned long long hd = s2 ? h2 : s1 ? h1 : h0;
    uint32_t cb = s2 ? c2 : s1 ? c1 : c0;
    if (__ballot(s3 && j >= carry_n)) {
        if (s3 && j >= carry_n) {
            b = b0 + 3u;
            while ((uint32_t)S.cnt[b + 1u] <= j) ++b;
            hd = S.heads[b];
            cb = S.cnt[b];
        }
    }
    const uint32_t pos = raw_base + b * 64u + select_bit64(hd, j - cb);
    return j < carry_n ? cpos : pos;
}

// Where a sequence's minimizers go: entry `dest` of its list.
struct GlobalList {  // a region of the minimizer buffers in device memory (entries beyond cap are dropped: the caller sees the count)
    unsigned long long *__restrict__ hash;
    uint32_t *__restrict__ pos;
    uint32_t cap;
    uint32_t *__restrict__ last;  // seeding variant 16 only (else nullptr): every minimizer's second position
    __device__ __forceinline__ void put(uint32_t dest, uint64_t hv, uint32_t p, uint32_t lp) const {
        if (dest < cap) {
            hash[dest] = hv;
            pos[dest] = p;
            if (last) last[dest] = lp;
        }
    }
};
// Lists the tile's candidates in position order (lane = candidate), resolves their raw positions and appends them to the
// sequence's minimizer list.  Returns the number of minimizers appended; sets inexact when a candidate fails the exact
// 64-bit test (the sequence then goes to the general path, whose test is exact by construction).
// Every lane first writes the windows of its own candidates to their places in the list (its flags are in registers: lowest
// set bit, clear, next), so that a candidate's lane afterwards reads ONE value and starts its look-ups -- no search for the owning
// lane, no bit select in another lane's flags.
template <bool VIEW = false, class Out = GlobalList, bool VAR = true>
__device__ __forceinline__ uint32_t seed_stage_r(const SeedTables &T, SeedLds &S, const DevParams &P, uint32_t w_eff,
                                                 uint32_t n_blocks, uint32_t n_codes, uint32_t raw_base, uint32_t carry_n,
                                                 const Out &out, uint32_t out_base, bool &inexact, const SeedView &V = SeedView()) {
    const uint32_t lane = lane_id();
    const uint32_t lc = (w_eff + 63u) >> 6;
    uint32_t n_listed = 0;  // VIEW: candidates that start before V.elig_end (positions ascend: they come first)
    // this lane's flags as 32-step words, masked to its real windows (steps [0, nv)): words and lanes stage B did not write hold
    // stale bits, all of them beyond nv
    constexpr uint32_t NW = SD_FLAG_WORDS;
    const uint32_t s0 = lane * lc;
    const uint32_t nv = s0 < w_eff ? (w_eff - s0 < lc ? w_eff - s0 : lc) : 0u;
    uint32_t f[NW];
    uint32_t my_count = 0;
#pragma unroll
    for (uint32_t w = 0; w < NW; ++w) f[w] = S.flagw[w * 64u + lane];
#pragma unroll
    for (uint32_t w = 0; w < NW; ++w) {
        const uint32_t k = nv > 32u * w ? nv - 32u * w : 0u;
        f[w] &= (k >= 32u ? ~0u : ((1u << k) - 1u));
        my_count += (uint32_t)__popc(f[w]);
    }
    const uint32_t incl = wave_incl_scan_u32(my_count);
    const uint32_t total = rdlane(incl, 63);
    const uint32_t my_prefix = incl - my_count;
    const uint32_t nw_used = (lc + 31u) >> 5;  // words that can hold a flag at all (wave-uniform)
#ifdef MQ_STAGE_R_SPLIT  // diagnostic: stage R's time by part -- 12 flags read, masked, counted, scanned; 13 listing; 14 window hashes; 15 raw positions + stores; 2 the rest
    mq_clk(12);
#endif
    for (uint32_t r0 = 0; r0 < total; r0 += SD_OWNER_CAP) {  // rounds of SD_OWNER_CAP candidates (one round unless the density is high)
        uint32_t at = my_prefix - r0;  // place of this lane's next candidate in the round's list (wraps below zero before the round)
#pragma unroll
        for (uint32_t w = 0; w < NW; ++w) {
            if (w < nw_used) {
                uint32_t g = f[w];
                while (__ballot(g != 0u)) {  // one pass lists one candidate of every lane that still has one in this word; straight-line body
                    const bool has = g != 0u;
                    const uint32_t t = s0 + 32u * w + (uint32_t)__ffs((int)g) - 1u;
                    if (has && at < SD_OWNER_CAP) S.cand[at] = (uint16_t)t;
                    at += has ? 1u : 0u;
                    g &= g - 1u;  // 0 stays 0
                }
            }
        }
        wave_sync();
#ifdef MQ_STAGE_R_SPLIT
        mq_clk(13);
#endif
        const uint32_t r1 = total - r0 < SD_OWNER_CAP ? total : r0 + SD_OWNER_CAP;
        for (uint32_t i0 = r0; i0 < r1; i0 += 64u) {
            const uint32_t i = i0 + lane;
            if (i < r1) {
                const uint32_t j = S.cand[i - r0];
                const Hash2 wh = window_hash(T, S, P.l, j);
#ifdef MQ_STAGE_R_SPLIT
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" ::"v"(wh.flo), "v"(wh.fhi), "v"(wh.rlo), "v"(wh.rhi));
                mq_clk(14);
#endif
                const uint32_t pos = seed_rawpos_batch(S, raw_base, carry_n, j);
                const uint64_t F = ((uint64_t)wh.fhi << 32) | wh.flo, R = ((uint64_t)wh.rhi << 32) | wh.rlo;
                const uint64_t hv = F < R ? F : R;
                if (hv > P.bound) inexact = true;
                const uint32_t dest = out_base + i;
                const bool listed = !VIEW || pos < V.elig_end;
                // seeding variants (wave-uniform, off in the frozen reading): 8 lists the last base of the first base's run = the base in
                // front of the next run head (l >= 2: code j + 1 is inside the window); 16 adds the window's last compressed base
                uint32_t rep = pos, lp = 0;
                if (var_pos_end<VAR>(P)) rep = seed_rawpos_batch(S, raw_base, carry_n, j + 1u) - 1u;
                if (var_end_compressed<VAR>(P)) lp = seed_rawpos_batch(S, raw_base, carry_n, j + P.l - 1u) + (VIEW ? V.pos_add : 0u);
                if (listed) out.put(dest, list_hash<VAR>(P, hv), VIEW ? rep + V.pos_add : rep, lp);
                if (VIEW) n_listed += (uint32_t)__popcll(__ballot(listed));
            }
#ifdef MQ_STAGE_R_SPLIT
            mq_clk(15);
#endif
        }
        wave_sync();
    }
    inexact = __ballot(inexact) != 0;
    return VIEW ? rdfirst(n_listed) : total;
}

// Whole sequence through the fast path, tile by tile.  Returns the number of minimizers (may exceed out_cap: overflow, the
// list is then incomplete) or 0xFFFFFFFF when the sequence does not qualify (non-ACGT byte / inexact candidate).
// STOP (diagnostic builds of the split pipeline only, never a product path): 1 = stage A only, 2 = stages A and B; the lists are
// then incomplete on purpose: a profiler attributes instructions and time to the stages by difference.
constexpr uint32_t SD_NOT_FAST = 0xFFFFFFFFu;
// a sequence the fast path takes at all (stage A reads whole 16-byte pieces; its tail piece is the 16 bytes that end at len)
__device__ __forceinline__ bool seed_fast_eligible(uint64_t len) { return len >= 16u && (len >> 32) == 0; }
// pre / pre_valid: the sequence's first super-row already requested by the caller (stage_a_request); else it is requested here.
// VIEW: seq[0, len) is a window of a longer sequence (SeedView): only the minimizers that start before V.elig_end are listed and
// counted, positions are shifted by V.pos_add, the base in front of the view decides whether its first base is a run head.
template <int STOP = 0, bool VIEW = false, class Out = GlobalList, bool VAR = true>
__device__ __forceinline__ uint32_t seed_sequence_fast_to(const uint8_t *__restrict__ seq, uint32_t len, const DevParams &P, const SeedTables &T,
                                                          SeedLds &S, const Out &out, APre &pre, bool pre_valid, const SeedView &V = SeedView()) {
    const uint32_t lane = lane_id();
    uint32_t raw0 = 0, carry_n = 0, carry_prev = VIEW ? (V.first_prev & 3u) : 0u, n_out = 0;
    uint32_t halo_heads = 0, seg_heads = 0;  // VIEW: run heads at or behind V.elig_end / in front of it
    if (!seed_fast_eligible(len)) return SD_NOT_FAST;
    if (!pre_valid) stage_a_request(seq, len, 0, pre);
    while (raw0 < len) {
        uint32_t n_codes = 0, n_blocks = 0, raw_end = 0;
        const bool ok = seed_stage_a(seq, len, raw0, carry_n, carry_prev, P.use_hpc != 0, P.fold != 0, T, S, n_codes, n_blocks, raw_end, pre,
                                     !VIEW || V.first_prev >= 4u);
        mq_clk(0);
        if (!ok) return SD_NOT_FAST;
        if (VIEW && V.more_after) {  // every l-mer that starts before elig_end must END inside the view: l - 1 run heads behind elig_end do it
            uint32_t c = 0, c0 = 0;
            for (uint32_t k = lane; k < n_blocks; k += 64u) {
                const uint32_t n = (uint32_t)__popcll(S.heads[k]);
                if (raw0 + 64u * k >= V.elig_end) c += n;
                else c0 += n;
            }
            halo_heads += wave_sum_u32(c);
            seg_heads += wave_sum_u32(c0);
        }
        const bool more = raw_end < len;
        if (STOP != 1 && n_codes >= P.l) {
            const uint32_t w_eff = n_codes - P.l + 1u;
            if (var_h32<VAR>(P)) seed_stage_b32(T, S, P, w_eff);  // seeding variant 4: one word per strand
            else seed_stage_b(T, S, P, w_eff);
            mq_clk(1);
            if (STOP != 2) {
                bool inexact = false;
                n_out += seed_stage_r<VIEW, Out, VAR>(T, S, P, w_eff, n_blocks, n_codes, raw0, carry_n, out, n_out, inexact, V);
                mq_clk(2);
                if (inexact) return SD_NOT_FAST;
            }
        }
        if (more) {
            // the last l-1 compressed bases (all of them when the tile has fewer: a very long homopolymer run) open the next
            // tile's code stream; their raw positions stay available for windows that start in them
            const uint32_t new_cn = n_codes < P.l - 1u ? n_codes : P.l - 1u;
            uint32_t cpos = 0, ccode = 0;
            if (lane < new_cn) cpos = seed_rawpos_batch(S, raw0, carry_n, n_codes - new_cn + lane);
            if (lane < 4u) {
                const uint32_t sb = 2u * (n_codes - new_cn) + 32u * lane;
                ccode = __builtin_amdgcn_alignbit(S.codes[(sb >> 5) + 1u], S.codes[sb >> 5], sb & 31u);
                const uint32_t keep = 2u * new_cn > 32u * lane ? 2u * new_cn - 32u * lane : 0u;
                ccode = keep >= 32u ? ccode : (ccode & ((1u << keep) - 1u));
            }
            wave_sync();
            if (lane < new_cn) S.carry_pos[lane] = cpos;
            if (lane < 4u) S.carry_codes[lane] = ccode;
            carry_n = new_cn;
        }
        wave_sync();
        raw0 = raw_end;
        mq_clk(3);
    }
    // too few run heads behind elig_end (long homopolymer runs): the general seeder takes the segment -- unless no l-mer starts in it at all
    if (VIEW && V.more_after && halo_heads + 1u < P.l && seg_heads != 0u) return SD_NOT_FAST;
    return n_out;
}
// the list in device memory: mz_hash[0, out_cap), mz_pos[0, out_cap)
template <int STOP = 0, bool VIEW = false, bool VAR = true>
__device__ __forceinline__ uint32_t seed_sequence_fast(const uint8_t *__restrict__ seq, uint32_t len, const DevParams &P, const SeedTables &T,
                                                       SeedLds &S, unsigned long long *__restrict__ mz_hash,
                                                       uint32_t *__restrict__ mz_pos, uint32_t out_cap, APre &pre, bool pre_valid,
                                                       const SeedView &V = SeedView(), uint32_t *__restrict__ mz_last = nullptr) {
    const GlobalList out = {mz_hash, mz_pos, out_cap, VAR ? mz_last : nullptr};
    return seed_sequence_fast_to<STOP, VIEW, GlobalList, VAR>(seq, len, P, T, S, out, pre, pre_valid, V);
}

}  // namespace mq
