// mq_device.hpp -- gfx950 device code for mapquik's hot path (seeding, index probe, Match runs, pseudo-chain).
//
// One wavefront (64 lanes) owns one sequence (or one segment of a long reference).  Everything is
// integer / byte work: no MFMA.  This header holds the GENERAL streaming seeder (any length, any bytes; the fast seeder for
// ACGT-only sequences is mq_seed.hpp), the index probe, the Match-run builder and the chain stage.
// Structure of the streaming seeder (lanes = consecutive positions):
//   raw bytes --(head flags, ballot/mbcnt compaction)--> HPC ring in LDS --(64-wide XOR prefix scan of
//   rotated ntHash seeds)--> canonical l-mer hashes --(density predicate, ballot compaction)--> ordered
//   minimizer list in LDS --> sink (k-min-mers -> probe -> runs, or a global minimizer list).
//
// ntHash without a serial roll: with X(t) = XOR_{u<=t} ror(h(c_u), u) and Y(t) = XOR_{u<=t} rol(hc(c_u), u),
//   fh(window ending at e) = rol(X(e) ^ X(e-l), e)        rh = ror(Y(e) ^ Y(e-l), e-l+1)
// (all rotation amounts mod 64; blocks of 64 HPC positions are 64-aligned so amounts are lane constants).
//
// Reference semantics restated here (citations relative to the reference tree):
//   KminmersIterator (rust-seq2kminmers, call sites src/mers.rs:27,53)   -> seed_segment / seed_sequence_fast + kminmer_hash
//   ReadOnlyIndex::get (src/index.rs:118-126)                           -> probe_table
//   Match::new/update/check/extend (src/match.rs:20-58), chain_matches (src/mers.rs:57-73) -> MapSink::batch_runs
//   Chain::get_match (src/chain.rs:147-169) and helpers                  -> chain_stage
//   find_largest_two_chains/determine_best_match/find_coords (src/mers.rs:104-183) -> chain_stage
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mapquik_hip.h"
#include "../../include/mapquik_hip_diag.h"

namespace mq {

constexpr int WAVE = 64;
constexpr int RING = 512;    // HPC ring entries kept per wave: l - 1 <= 63 behind the block being hashed + 63 waiting + one write group of the walk (<= 256)
constexpr int MZ_CAP = 160;  // < (64 + k - 1) carried + 64 appended, k <= 32
constexpr int MAX_K = 32;
constexpr int MAX_L = 64;

struct DevParams {
    uint64_t bound;  // density bound: keep l-mer iff min(fh,rh) <= bound
    uint32_t k, l, use_hpc, c, s, g;
    uint32_t fold;  // 1: a-z count as A-Z (to_ascii_uppercase of src/closures.rs:63,106 done here instead of by the caller)
    // Seeding variant (MQ_SEEDVAR_* bits, include/mapquik_hip.h; 0 = the frozen reading).  What the kernels see of each bit:
    //   1, 2   nothing: `bound` already is what `hash <= bound` must be compared with (bound - 1 for the strict test; keep_none when
    //          even that cannot say it: strict test against a bound of 0)
    //   4      the seeds are the low halves of the 64-bit seeds, DUPLICATED into both halves of a 64-bit word: rol64 of such a word
    //          is rol32 of its half in both halves, XOR keeps the form, and dup(a) <= dup(b) <=> a <= b -- so every kernel runs
    //          unchanged on dup(h32) against bound = dup(bound32), and only a hash on its way into a list is cut to its low word
    //   8      a minimizer's listed position is (raw position of the NEXT run head) - 1
    //   16     a minimizer carries a second position (raw position of its window's last compressed base) in a list of its own
    //   32     kminmer_hash: an undecided (palindromic) tuple counts as reversed
    uint32_t variant;
    uint32_t keep_none;  // 1: no l-mer passes the density test at all
    uint32_t fast_kh;    // 1: MQ_FLAG_FAST_KH -- the tuple hash is kh_fast() instead of SipHash-1-3 (index and reads alike)
};
// VAR = false: code built for the frozen reading only -- the variant bits are not even looked at, so map_kernel's instantiation for
// variant 0 (every timed launch) carries none of the variants' code or registers; every other kernel is built with VAR = true and
// decides at run time (wave-uniform branches), with identical results for variant 0
template <bool VAR> __device__ __forceinline__ bool var_h32(const DevParams &P) { return VAR && (P.variant & MQ_SEEDVAR_HASH32) != 0; }
template <bool VAR> __device__ __forceinline__ bool var_pos_end(const DevParams &P) { return VAR && (P.variant & MQ_SEEDVAR_POS_RUN_END) != 0; }
template <bool VAR> __device__ __forceinline__ bool var_end_compressed(const DevParams &P) { return VAR && (P.variant & MQ_SEEDVAR_END_COMPRESSED) != 0; }
template <bool VAR> __device__ __forceinline__ bool var_rev_eq(const DevParams &P) { return VAR && (P.variant & MQ_SEEDVAR_REV_ON_EQUAL) != 0; }
template <bool VAR> __device__ __forceinline__ bool var_keep_none(const DevParams &P) { return VAR && P.keep_none != 0; }
// a hash as it goes into a minimizer list: the 32-bit variant's value zero-extended
template <bool VAR> __device__ __forceinline__ uint64_t list_hash(const DevParams &P, uint64_t h) { return var_h32<VAR>(P) ? (h & 0xFFFFFFFFull) : h; }

// Index table: 64-byte buckets of two 32-byte slots, both keys first so that ONE 16-byte load decides most lookups.
//   slot s = bucket s >> 1, way s & 1;  key == 0 <=> empty (a real key 0 lives in way 0 of one extra bucket behind the table).
// Probe sequence of a key (insert and lookup walk the same one; src/index.rs:11-39's identity hasher gives the home slot):
//   home slot (key & mask), the other way of the home bucket, then the following buckets way 0, way 1, ...
// A slot is live iff its key matches, pay.end != 0 and ENTRY_DUP is not set in pay.id_rc: a key inserted more than once gets the
// bit while the table is built (the order-independent form of "second insert => tombstone", src/index.rs:94-104; is_empty <=>
// end == 0, src/index.rs:67-69), so a lookup needs the 16 payload bytes only, and only for a key that matched.
struct alignas(16) Entry {
    uint32_t start, end, offset, id_rc;
};
struct alignas(64) Bucket {
    unsigned long long key[2];
    Entry pay[2];
    uint32_t claims;    // extra bucket only: insertions of the key 0 (its key field cannot tell "present" from "empty")
    uint32_t pad[3];
};
static_assert(sizeof(Bucket) == 64, "bucket size");
// Entry::id_rc bit 31: the key was inserted more than once (or its entry's end is 0): the reference's tombstone, an entry that
// compares as empty (src/index.rs:67-69, 94-104).  Set while the table is built (table_insert); reference ids stay below 2^24.
constexpr uint32_t ENTRY_DUP = 0x80000000u;
__device__ __forceinline__ bool entry_live(const Entry &e) { return e.end != 0u && !(e.id_rc & ENTRY_DUP); }
constexpr uint32_t SLOT_BYTES = 32;  // table bytes per slot

// reference k-min-mer waiting for insertion (same layout as mq_kminmer, rev field = id<<1|rc)
struct alignas(8) RefKmm {
    unsigned long long hash;
    uint32_t start, end, offset, id_rc;
};
static_assert(sizeof(RefKmm) == 24, "RefKmm size");

// src/match.rs:10-18 plus the ref id of the run's first entry (src/mers.rs:68) and a "grouped" flag
struct alignas(16) MatchRec {
    uint32_t q_start, q_end, r_start, r_end, count, ref, rc, done;
};

// Per-wave LDS of the general streaming seeder
struct WaveLds {
    unsigned long long mz_hash[MZ_CAP];
    uint32_t mz_pos[MZ_CAP];
    uint32_t mz_last[MZ_CAP];  // variant 16 only: raw position of the window's last compressed base
    uint32_t ring_pos[RING];
    uint8_t ring_code[RING];
};

// ------------------------------------------------------------------ wave helpers
// LDS hand-off between lanes of ONE wave (waves of a workgroup work on different reads and never rendezvous):
// LDS operations of a wave execute in order; this only stops the compiler from moving them across the point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The empty asm makes every call site compute its own copy: with a pure lane id the compiler hoists every lane-dependent address
// and mask of EVERY phase out of the persistent per-read loop and keeps them all live (161 VGPRs for the fused kernel, 112 with this).
__device__ __forceinline__ uint32_t lane_id() {
    uint32_t x = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(x));
    return x;
}
// Diagnostic builds only (-DMQ_STAGE_CLOCKS): wave time per stage.  mq_clk(i) charges the cycles since the wave's previous stamp
// to stage i; map_kernel adds every wave's totals to counters[16..] at its end (read back by mq_last_stage_clocks).
//   0 stage A  1 stage B  2 stage R  3 tile carry  4 list stores acknowledged  5 list -> LDS  6 tuple hashes + probe issue
//   7 probe resolve + runs  8 runs finished, Match records in L2  9 chain + result  10 general seeder  11 next read (atomic, offsets)
//   with -DMQ_STAGE_A_SPLIT stage A's time is charged to: 10 / 13 what precedes the super-row loop in a sequence's first / later
//   tiles; 0 / 14 the wait for the bases of the first super-row of a first / later tile; 12 that wait for the other super-rows;
//   15 the work (decode, look-ups, scan, stream) -- three more stamps per super-row, so only for looking inside stage A
//   with -DMQ_STAGE_MAP_SPLIT stage 7 is split: 12 home buckets' keys arrived and compared, payloads requested; 13 lookups that walk on;
//   14 payloads arrived; 7 the runs
constexpr int MQ_N_CLK = 16;
#ifdef MQ_STAGE_CLOCKS
struct StageClkLds {
    unsigned long long acc[16][MQ_N_CLK];  // a workgroup has at most 16 waves (1,024 threads)
    unsigned long long last[16];
};
__device__ __forceinline__ StageClkLds &mq_clk_lds() {
    __shared__ StageClkLds C;
    return C;
}
__device__ __forceinline__ void mq_clk(int i) {
    StageClkLds &C = mq_clk_lds();
    const uint32_t w = threadIdx.x >> 6;
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63u) == 0) {
        if (i >= 0) C.acc[w][i] += now - C.last[w];
        C.last[w] = now;
    }
}
#elif defined(MQ_CODE_MARKS)
// Diagnostic build (tools/code_sizes.sh, never loaded): every stage stamp becomes a symbol in the code object -- mq_mark_<stage>_<n> -- so that
// llvm-readelf can say how many bytes of map_kernel lie between two stamps (which stage owns how much of the instruction cache's 64 KB).
__device__ __forceinline__ void mq_clk(int i) { asm volatile("mq_mark_%c0_%=:" ::"n"(i + 1)); }
#else
__device__ __forceinline__ void mq_clk(int) {}
#endif
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, int l) {
    return ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32) | rdlane((uint32_t)v, l);
}
__device__ __forceinline__ uint32_t rdfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d) {
    uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64);
    uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rotl64(uint64_t x, uint32_t r) { return __builtin_rotateleft64(x, (uint64_t)(r & 63u)); }
__device__ __forceinline__ uint64_t rotr64(uint64_t x, uint32_t r) { return __builtin_rotateright64(x, (uint64_t)(r & 63u)); }
// Wave reductions on the DPP network (row_shr inside each row of 16, then row_bcast:15 / row_bcast:31 across rows): VALU-rate
// moves, where __shfl_xor would go through ds_bpermute (~24 cycles of issue each, profiles/r02_valu_issue.txt).  Lanes that
// a shift leaves without a source read `old` = the operation's identity.  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return rdlane(wave_incl_scan_u32(v), 63); }
// the same scan with XOR (the general seeder's prefix of rotated seeds), on the two halves of a 64-bit word
__device__ __forceinline__ uint32_t wave_incl_xor_scan_u32(uint32_t x) {
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ uint64_t wave_incl_xor_scan_u64(uint64_t v) {
    return ((uint64_t)wave_incl_xor_scan_u32((uint32_t)(v >> 32)) << 32) | wave_incl_xor_scan_u32((uint32_t)v);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return rdlane(x, 63);
}

// ------------------------------------------------------------------ ntHash-1 seeds (64-bit), non-ACGT -> 0
__device__ __forceinline__ uint32_t base_code(uint32_t b) {
    return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u;
}
// h32 (variant 4): the low half of the seed in both halves of the word
__device__ __forceinline__ uint64_t dup_low(uint64_t s) { return (s & 0xFFFFFFFFull) | (s << 32); }
__device__ __forceinline__ uint64_t nt_seed(uint32_t code, bool h32 = false) {
    const uint64_t s = code == 0 ? 0x3c8bfbb395c60474ULL
                     : code == 1 ? 0x3193c18562a02b4cULL
                     : code == 2 ? 0x20323ed082572324ULL
                     : code == 3 ? 0x295549f54be24456ULL
                                 : 0ULL;
    return h32 ? dup_low(s) : s;
}
__device__ __forceinline__ uint32_t comp_code(uint32_t code) { return code < 4 ? 3u - code : 4u; }

// ------------------------------------------------------------------ SipHash-1-3, key 0 (Rust DefaultHasher) over [len, m_0..m_{k-1}]
// The state is kept as 32-bit halves: a rotation by 32 is then a renaming, the others are two v_alignbit each, and a 64-bit add is
// v_add_co + v_addc.  Written on uint64_t the compiler emits v_lshl_add_u64 / v_lshlrev_b64 on aligned register pairs plus the
// moves that re-pair the halves after every rotation by 32: 35 instructions a round against 28 (tools/sip_enc.hip,
// profiles/r04_sip_enc.txt: 88 against 79 cycles a round per SIMD at this kernel's occupancy).
struct U2 {
    uint32_t lo, hi;
};
__device__ __forceinline__ U2 u2_of(uint64_t v) { return {(uint32_t)v, (uint32_t)(v >> 32)}; }
__device__ __forceinline__ uint64_t u64_of(U2 v) { return ((uint64_t)v.hi << 32) | v.lo; }
// a += b; b = rotl(b, R), 0 < R < 32 -- every add of a SipHash round is followed by a rotation of its addend.  One asm block: the
// two v_alignbit sit between v_add_co and v_addc, which covers the two wait states gfx950 wants between a VALU write of vcc and a
// VALU read of it (the compiler fills them with s_nop), and plain C for the add is fused back into v_lshl_add_u64.
template <uint32_t R>
__device__ __forceinline__ void add_rotl_u2(U2 &a, U2 &b) {
    U2 s, r;
    asm("v_add_co_u32_e32 %0, vcc, %4, %6\n\tv_alignbit_b32 %2, %6, %7, %8\n\tv_alignbit_b32 %3, %7, %6, %8\n\tv_addc_co_u32_e32 %1, vcc, %5, %7, vcc"
        : "=&v"(s.lo), "=&v"(s.hi), "=&v"(r.lo), "=&v"(r.hi)
        : "v"(a.lo), "v"(a.hi), "v"(b.lo), "v"(b.hi), "n"(32u - R)
        : "vcc");
    a = s;
    b = r;
}
__device__ __forceinline__ U2 xor_u2(U2 a, U2 b) { return {a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ U2 swap_u2(U2 a) { return {a.hi, a.lo}; }

// the state after init() and the first word (the tuple's length k, the same for every k-min-mer): folded at compile time
struct SipState {
    uint64_t v0, v1, v2, v3;
};
constexpr uint64_t sip_rotl(uint64_t x, unsigned r) { return (x << r) | (x >> (64u - r)); }
constexpr SipState sip_round(SipState s) {
    s.v0 += s.v1; s.v1 = sip_rotl(s.v1, 13); s.v1 ^= s.v0; s.v0 = sip_rotl(s.v0, 32);
    s.v2 += s.v3; s.v3 = sip_rotl(s.v3, 16); s.v3 ^= s.v2;
    s.v0 += s.v3; s.v3 = sip_rotl(s.v3, 21); s.v3 ^= s.v0;
    s.v2 += s.v1; s.v1 = sip_rotl(s.v1, 17); s.v1 ^= s.v2; s.v2 = sip_rotl(s.v2, 32);
    return s;
}
constexpr SipState sip_initial() { return {0x736f6d6570736575ULL, 0x646f72616e646f6dULL, 0x6c7967656e657261ULL, 0x7465646279746573ULL}; }
constexpr SipState sip_after_word(SipState s, uint64_t m) {
    s.v3 ^= m;
    s = sip_round(s);
    s.v0 ^= m;
    return s;
}

struct Sip13 {
    U2 v0, v1, v2, v3;
    __device__ __forceinline__ void set(const SipState &s) {
        v0 = u2_of(s.v0);
        v1 = u2_of(s.v1);
        v2 = u2_of(s.v2);
        v3 = u2_of(s.v3);
    }
    __device__ __forceinline__ void init() { set(sip_initial()); }
    template <uint64_t FIRST>
    __device__ __forceinline__ void init_after() {  // init(); word(FIRST);
        constexpr SipState s = sip_after_word(sip_initial(), FIRST);
        set(s);
    }
    __device__ __forceinline__ void round() {
        add_rotl_u2<13>(v0, v1); v1 = xor_u2(v1, v0); v0 = swap_u2(v0);
        add_rotl_u2<16>(v2, v3); v3 = xor_u2(v3, v2);
        add_rotl_u2<21>(v0, v3); v3 = xor_u2(v3, v0);
        add_rotl_u2<17>(v2, v1); v1 = xor_u2(v1, v2); v2 = swap_u2(v2);
    }
    __device__ __forceinline__ void word(U2 m) {
        v3 = xor_u2(v3, m);
        round();
        v0 = xor_u2(v0, m);
    }
    __device__ __forceinline__ void word(uint64_t m) { word(u2_of(m)); }
    __device__ __forceinline__ uint64_t finish(uint32_t nbytes) {
        const uint32_t b = (nbytes & 0xffu) << 24;  // the length byte, bits 56..63
        v3.hi ^= b;
        round();
        v0.hi ^= b;
        v2.lo ^= 0xffu;
        round();
        round();
        round();
        return u64_of(xor_u2(xor_u2(v0, v1), xor_u2(v2, v3)));
    }
};

// MQ_FLAG_FAST_KH: the opt-in cheap tuple hash (include/mapquik_hip.h).  Two 64-bit words on 32-bit halves:
//   x ^= m; x += y; y = rotl(y, 13) ^ x; x = rotl(x, 32)      per minimizer (8 instructions: the rotation by 32 is a renaming)
//   x ^= 0xff; six more steps with rotations 17 21 13 16 17 21; result x ^ y       -- ~80 instructions for k = 5 against SipHash-1-3's ~250
struct KhFast {
    U2 x, y;
    __device__ __forceinline__ void init(uint32_t k) {
        x = u2_of(0x736f6d6570736575ULL ^ (uint64_t)k);
        y = u2_of(0x646f72616e646f6dULL);
    }
    template <uint32_t R>
    __device__ __forceinline__ void step() {
        add_rotl_u2<R>(x, y);  // x += y; y = rotl(y, R)
        y = xor_u2(y, x);
        x = swap_u2(x);
    }
    __device__ __forceinline__ void word(U2 m) {
        x = xor_u2(x, m);
        step<13>();
    }
    __device__ __forceinline__ uint64_t finish() {
        x.lo ^= 0xffu;
        step<17>();
        step<21>();
        step<13>();
        step<16>();
        step<17>();
        step<21>();
        return u64_of(xor_u2(x, y));
    }
};

// canonical orientation + tuple hash of k minimizer hashes read through `get(i)`, i = 0..k-1 (forward order)
// rev_eq (variant 32): a tuple equal to its reverse counts as reversed (`<=` instead of `<`)
template <class Get>
__device__ __forceinline__ uint64_t kminmer_hash(uint32_t k, Get get, bool &rev, bool rev_eq = false, bool fast_kh = false) {
    rev = rev_eq;
    for (uint32_t i = 0; i < k; ++i) {
        uint64_t a = get(i), b = get(k - 1 - i);
        if (b < a) { rev = true; break; }
        if (b > a) { rev = false; break; }
    }
    if (fast_kh) {
        KhFast f;
        f.init(k);
        for (uint32_t i = 0; i < k; ++i) f.word(u2_of(rev ? get(k - 1 - i) : get(i)));
        return f.finish();
    }
    Sip13 h;
    h.init();
    h.word((uint64_t)k);
    for (uint32_t i = 0; i < k; ++i) h.word(rev ? get(k - 1 - i) : get(i));
    return h.finish(8u * (k + 1u));
}

// The same for a k known at compile time: the k hashes are read once (all reads in flight together), the orientation comes out of
// k/2 comparisons without a branch (pairs past the middle repeat the earlier ones, which were equal if the loop got that far), and
// a reverse tuple is the forward one with its pairs (i, k-1-i) exchanged under a lane mask -- three bit operations a half-word.
// (`r ? w[k-1-i] : w[i]` compiled to a select of the INDEX and a chain of k-1 compare/select pairs per word.)
template <uint32_t K, class Get>
__device__ __forceinline__ uint64_t kminmer_hash_fixed(Get get, bool &rev, bool rev_eq = false, bool fast_kh = false) {
    uint64_t w[K];
#pragma unroll
    for (uint32_t i = 0; i < K; ++i) w[i] = get(i);
    bool r = false, decided = false;
#pragma unroll
    for (uint32_t i = 0; i < K / 2u; ++i) {
        r = decided ? r : (w[K - 1u - i] < w[i]);
        decided = decided || (w[K - 1u - i] != w[i]);
    }
    r = decided ? r : rev_eq;  // a palindromic tuple: forward under `<`, reversed under `<=` (the exchange below leaves it as it is)
    rev = r;
    U2 m[K];
#pragma unroll
    for (uint32_t i = 0; i < K; ++i) m[i] = u2_of(w[i]);
    const uint32_t swap = r ? ~0u : 0u;
#pragma unroll
    for (uint32_t i = 0; i < K / 2u; ++i) {
        const uint32_t dlo = (m[i].lo ^ m[K - 1u - i].lo) & swap, dhi = (m[i].hi ^ m[K - 1u - i].hi) & swap;
        m[i].lo ^= dlo;
        m[i].hi ^= dhi;
        m[K - 1u - i].lo ^= dlo;
        m[K - 1u - i].hi ^= dhi;
    }
    if (fast_kh) {  // (wave-uniform: mq_params.flags)
        KhFast f;
        f.init(K);
#pragma unroll
        for (uint32_t i = 0; i < K; ++i) f.word(m[i]);
        return f.finish();
    }
    Sip13 h;
    h.template init_after<(uint64_t)K>();
#pragma unroll
    for (uint32_t i = 0; i < K; ++i) h.word(m[i]);
    return h.finish(8u * (K + 1u));
}

// ------------------------------------------------------------------ index probe (ReadOnlyIndex::get, src/index.rs:118-126)
// table has nslots = mask + 1 slots (nslots / 2 buckets) plus one extra bucket [nslots / 2] for the key 0.
__device__ __forceinline__ uint4 ld_u4(const void *p) { return *reinterpret_cast<const uint4 *>(p); }
__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
__device__ __forceinline__ bool probe_table(const Bucket *__restrict__ table, uint64_t mask, uint64_t key, Entry &out) {
    const uint64_t nb = (mask + 1) >> 1;
    if (key == 0) {
        out = table[nb].pay[0];
        return entry_live(out);
    }
    const uint64_t s0 = key & mask;
    uint64_t b = s0 >> 1;
    uint32_t w = (uint32_t)s0 & 1u;
    for (uint32_t step = 0;; ++step) {
        const unsigned long long k = table[b].key[w];
        if (k == key) {
            out = table[b].pay[w];
            return entry_live(out);
        }
        if (k == 0) return false;
        if (step == 0) {
            w ^= 1u;  // the other way of the home bucket
        } else if (step == 1 || w == 1u) {
            b = b + 1 == nb ? 0 : b + 1;
            w = 0;
        } else {
            w = 1u;
        }
    }
}

// ------------------------------------------------------------------ streaming seeder
// Emits, in order, every minimizer whose l-mer starts at a homopolymer-run head with raw index in [a, b).
// Sink interface: void on_minimizers(WaveLds&, uint32_t &mz_count)  (called after each block, wave-uniform)
// 16 bytes at `at` of a sequence of len bytes (those at or behind len read as 0); n = how many of them are inside
__device__ __forceinline__ uint4 load16_tail(const uint8_t *__restrict__ seq, uint64_t len, uint64_t at, uint32_t &n) {
    typedef uint4 __attribute__((aligned(1))) uint4_u;
    if (at + 16u <= len) {
        n = 16u;
        return *reinterpret_cast<const uint4_u *>(seq + at);
    }
    n = at < len ? (uint32_t)(len - at) : 0u;
    uint64_t lo = 0, hi = 0;  // (the sequence's last piece only; no array: a dynamically indexed one lives in scratch memory)
    for (uint32_t j = 0; j < n; ++j) {
        const uint64_t b = seq[at + j];
        if (j < 8u) lo |= b << (8u * j);
        else hi |= b << (8u * (j - 8u));
    }
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
// first position >= from whose byte differs from v (compared as the general seeder compares: a-z as A-Z when folding), or len; 4 KB a step
__device__ __forceinline__ uint64_t next_byte_differing(const uint8_t *__restrict__ seq, uint64_t len, uint64_t from, uint32_t v, bool fold) {
    const uint32_t lane = lane_id();
    const uint32_t m8 = (fold && v - 'A' < 26u) ? 0xDFu : 0xFFu;  // v is what a byte folds TO: a letter matches its lower case too
    const uint32_t mm = m8 * 0x01010101u, vv = (v & 0xFFu) * 0x01010101u;
    if (v > 0xFFu) return from;  // no byte at all (the sequence's start)
    for (uint64_t p = from; p < len; p += 4096u) {
        const uint64_t at = p + 64u * lane;
        uint32_t first = 64u;  // index of the lane's first differing byte
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            uint32_t n;
            const uint4 q = load16_tail(seq, len, at + 16u * (uint32_t)j, n);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                const uint32_t d = (w[i] & mm) ^ vv;
                if (d) {
                    const uint32_t k = 16u * (uint32_t)j + 4u * (uint32_t)i + ((uint32_t)__ffs((int)d) - 1u) / 8u;
                    if (k < 16u * (uint32_t)j + n) first = k;  // (bytes behind len read as 0: not a difference)
                }
            }
        }
        const uint64_t m = __ballot(first < 64u);
        if (m) {
            const int l0 = __ffsll((long long)m) - 1;
            return p + 64u * (uint32_t)l0 + rdlane(first, l0);
        }
    }
    return len;
}
// min_last > 0: only windows whose LAST compressed base lies at or behind raw position min_last (seed_read_hybrid: the windows that lie
// wholly in front of it were listed by the fast seeder)
template <bool VAR = true, class Sink>
__device__ __forceinline__ void seed_segment(const uint8_t *__restrict__ seq, uint64_t len, uint64_t a, uint64_t b,
                                             const DevParams &P, WaveLds &S, Sink &sink, uint32_t &mz_count, uint64_t min_last = 0) {
    const uint32_t lane = lane_id();
    const uint32_t l = P.l;
    if (a >= len || a >= b) return;
    uint32_t hbase = 0, hproc = 0;  // HPC positions seen / folded into the scan (segment-local, wave-uniform)
    uint32_t s_elig = 0;            // heads with raw index < b
    uint64_t prevF = 0, prevR = 0;  // per-lane inclusive prefixes of the previous block
    uint64_t carryF = 0, carryR = 0;
    uint32_t prev_byte = a > 0 ? (uint32_t)seq[a - 1] : 0x100u;
    if (P.fold && prev_byte - 'a' < 26u) prev_byte -= 32u;
    // lane constants
    const uint32_t from = (lane - l) & 63u;
    const bool src_cur = lane + l <= 63u;
    const uint32_t rot_r = (lane - l + 1u) & 63u;
    const bool h32 = var_h32<VAR>(P), pos_end = var_pos_end<VAR>(P), with_last = var_end_compressed<VAR>(P);  // seeding variants (wave-uniform)

    auto process_block = [&](uint32_t nvalid) {
        const uint32_t idx = hproc + lane;
        const bool valid = lane < nvalid;
        const uint32_t code = valid ? (uint32_t)S.ring_code[idx & (RING - 1)] : 4u;
        uint64_t tf = rotr64(nt_seed(code, h32), lane);
        uint64_t tr = rotl64(nt_seed(comp_code(code), h32), lane);
        tf = wave_incl_xor_scan_u64(tf);  // DPP row shifts and broadcasts (as ds_bpermute steps each of the six was an LDS round trip)
        tr = wave_incl_xor_scan_u64(tr);
        tf ^= carryF;
        tr ^= carryR;
        const uint64_t of = shfl64(src_cur ? tf : prevF, (int)from);
        const uint64_t orr = shfl64(src_cur ? tr : prevR, (int)from);
        const uint64_t fh = rotl64(tf ^ of, lane);
        const uint64_t rh = rotr64(tr ^ orr, rot_r);
        const uint64_t h = fh < rh ? fh : rh;
        const uint32_t j = idx - (l - 1u);  // HPC index of the window's first base
        const bool sel = valid && idx >= l - 1u && j < s_elig && h <= P.bound && !var_keep_none<VAR>(P) &&
                         (min_last == 0 || (uint64_t)S.ring_pos[idx & (RING - 1)] >= min_last);
        const uint64_t sm = __ballot(sel);
        if (sel) {
            const uint32_t o = mz_count + mbcnt64(sm);
            S.mz_hash[o] = list_hash<VAR>(P, h);
            // variant 8: the last base of the first base's run = the base in front of the next run head (l >= 2: inside the window)
            S.mz_pos[o] = pos_end ? S.ring_pos[(j + 1u) & (RING - 1)] - 1u : S.ring_pos[j & (RING - 1)];
            if (with_last) S.mz_last[o] = S.ring_pos[idx & (RING - 1)];  // variant 16: the window's last compressed base
        }
        mz_count += (uint32_t)__popcll(sm);
        prevF = tf;
        prevR = tr;
        carryF = rdlane64(tf, 63);
        carryR = rdlane64(tr, 63);
        wave_sync();
        sink.on_minimizers(S, mz_count);
    };

    // The walk: 1 KB per step, 16 bytes per lane.  What it looks for are run heads (a byte that differs from the byte in front of it; every byte
    // without HPC) -- the hashing above is per 64 HEADS, not per 64 bytes.  Sequences that come here are what the fast seeder declined: gaps of
    // the reference (N by the ten thousand), reads drawn from inside one (N with a sequencing error every ~140 bytes: two heads per error), the
    // neighbourhood of a stray byte in otherwise clean sequence.  A lane compares its 16 bytes with the same bytes shifted by one (SWAR,
    // exact non-zero-byte mask), the heads' places in the ring come from one wave scan, and each lane writes its own heads, lowest first.
    // (Round 5 walked 64 bytes per step, one per lane: 375 steps for a 24-kb read from inside a gap against 24, each with its ballots, its
    // scalar bookkeeping and a memory round trip: such a read cost 4-5 x an ordinary read, 0.7 % of the maize-like batch.)
    constexpr int AHEAD = 2;  // steps in flight behind the current one
    auto load16 = [&](uint64_t at) -> uint4 {
        uint32_t n;
        return load16_tail(seq, len, at, n);
    };
    uint4 nb[AHEAD + 1];
#pragma unroll
    for (int j = 0; j <= AHEAD; ++j) nb[j] = load16(a + 1024u * (uint32_t)j + 16u * lane);
    for (uint64_t pos = a;; pos += 1024) {
        const uint64_t at = pos + 16u * lane;
        uint4 v = nb[0];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) nb[j] = nb[j + 1];
        nb[AHEAD] = load16(at + 1024u * (uint32_t)(AHEAD + 1));
        const uint32_t nin = at < len ? (len - at < 16u ? (uint32_t)(len - at) : 16u) : 0u;  // bytes of this lane inside the sequence
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (P.fold) {  // a-z -> A-Z, byte-wise: 0x20 in every byte of 0x61..0x7A (no carry leaves a byte: the tests run on the low seven bits)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t t = w[i] & 0x7F7F7F7Fu;
                w[i] ^= (((t + 0x1F1F1F1Fu) & ~(t + 0x05050505u) & ~w[i]) & 0x80808080u) >> 2;
            }
        }
        // the byte in front of the lane's first: the previous lane's last (DPP wave_shr:1; lane 0 keeps `old` = the byte in front of the step)
        const uint32_t pl = (uint32_t)__builtin_amdgcn_update_dpp((int)(prev_byte & 0xFFu), (int)(w[3] >> 24), 0x138, 0xf, 0xf, false);
        uint32_t hm = 0xFFFFu;  // bit j: byte j is a run head
        if (P.use_hpc) {
            const uint32_t sh[4] = {(w[0] << 8) | pl, __builtin_amdgcn_alignbyte(w[1], w[0], 3), __builtin_amdgcn_alignbyte(w[2], w[1], 3),
                                    __builtin_amdgcn_alignbyte(w[3], w[2], 3)};
            hm = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t d = w[i] ^ sh[i];
                const uint32_t nz = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 7;  // 1 in every byte that differs from the byte in front of it
                hm |= ((nz * 0x01020408u) >> 24 & 0xFu) << (4 * i);                             // those four bits side by side (no two partial products meet)
            }
            if (prev_byte > 0xFFu && at == 0) hm |= 1u;  // the sequence's first byte has nothing in front of it: a head whatever it is
        }
        hm &= nin >= 16u ? 0xFFFFu : ((1u << nin) - 1u);
        const uint32_t cnt = (uint32_t)__popc(hm);
        // heads with raw index < b (the windows that START there are this segment's)
        const uint32_t eb = at >= b ? 0u : (b - at >= 16u ? 0xFFFFu : ((1u << (uint32_t)(b - at)) - 1u));
        const uint32_t incl2 = wave_incl_scan_u32(cnt | ((uint32_t)__popc(hm & eb) << 16));  // both counts in one scan (<= 1024 each)
        const uint32_t tot2 = rdlane(incl2, 63);
        const uint32_t total = tot2 & 0xFFFFu;
        s_elig += tot2 >> 16;
        const uint32_t excl = (incl2 & 0xFFFFu) - cnt;
        const bool full = pos + 1024u <= len;
        const uint32_t last_byte = rdlane(w[3] >> 24, 63);  // (used only when the step lies inside the sequence)
        // the heads go into the ring in groups the ring has room for: all at once unless the step is dense (then 16 lanes = 256 bytes at a time)
        const uint32_t ngrp = total <= 256u ? 1u : 4u, glanes = 64u / ngrp;
        const uint32_t hb0 = hbase;  // the step's first head goes here
        for (uint32_t gi = 0; gi < ngrp; ++gi) {
            const bool mine = lane / glanes == gi;
            uint32_t m = mine ? hm : 0u;
            uint32_t o = hb0 + excl;
            while (__ballot(m != 0u)) {
                if (m) {
                    const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
                    const uint32_t wd = bit < 8u ? (bit < 4u ? w[0] : w[1]) : (bit < 12u ? w[2] : w[3]);
                    const uint32_t bt = (wd >> (8u * (bit & 3u))) & 0xFFu;
                    S.ring_code[o & (RING - 1)] = (uint8_t)base_code(bt);
                    S.ring_pos[o & (RING - 1)] = (uint32_t)(at + bit);
                    ++o;
                    m &= m - 1u;
                }
            }
            hbase += ngrp == 1u ? total : (rdlane(incl2, (int)(glanes * (gi + 1u) - 1u)) & 0xFFFFu) - (gi ? (rdlane(incl2, (int)(glanes * gi - 1u)) & 0xFFFFu) : 0u);
            wave_sync();
            while (hbase - hproc >= 64u) {
                process_block(64u);
                hproc += 64u;
            }
        }
        prev_byte = last_byte;
        if (P.use_hpc && total == 0u && full && pos + 2048u <= len) {
            // 1 KB that repeats the byte in front of it (a gap of the reference: tens of thousands of N; a very long homopolymer run): jump to the
            // KB that holds the next different byte -- nothing in between is a run head (bytes are compared as the walk compares them: folded)
            const uint64_t q = next_byte_differing(seq, len, pos + 1024u, prev_byte, P.fold != 0);
            const uint64_t skip = ((q - (pos + 1024u)) >> 10) << 10;
            if (skip) {
                pos += skip;
#pragma unroll
                for (int j = 0; j <= AHEAD; ++j) nb[j] = load16(pos + 1024u * (uint32_t)(j + 1) + 16u * lane);
            }
        }
        const bool end_of_seq = pos + 1024 >= len;
        // behind b: done once the last eligible window is complete -- or at once when no run head lies in [a, b) at all (a segment inside
        // a long homopolymer run has nothing to seed; without this it would read on to the end of the run, however far that is)
        const bool past = (pos + 1024 >= b) && (s_elig == 0u || hbase >= s_elig + l - 1u);
        if (end_of_seq || past) break;
    }
    if (hbase > hproc) process_block(hbase - hproc);
}

// ------------------------------------------------------------------ chain helpers (src/chain.rs)
__device__ __forceinline__ uint64_t abs_as_usize(int32_t x) {
    int32_t a = x < 0 ? (int32_t)(0u - (uint32_t)x) : x;  // i32::MIN.abs() wraps in release
    return (uint64_t)(int64_t)a;
}
// fwd_gap_too_long / rc_gap_too_long (src/chain.rs:132-142): `as i32` casts, wrapping subtraction
__device__ __forceinline__ bool gap_too_long(uint32_t v_q_s, uint32_t u_q_e, uint32_t x, uint32_t y, uint32_t g) {
    int32_t g1 = (int32_t)(v_q_s - u_q_e);
    int32_t g2 = (int32_t)(x - y);
    return abs_as_usize((int32_t)((uint32_t)g1 - (uint32_t)g2)) > (uint64_t)g;
}
// check_match_compatible (src/chain.rs:43-63), h1 = anchor
__device__ __forceinline__ bool match_compatible(const MatchRec &h1, const MatchRec &h2, uint32_t g) {
    if (h1.q_start == h2.q_start && h1.q_end == h2.q_end && h1.r_start == h2.r_start && h1.r_end == h2.r_end &&
        h1.count == h2.count && h1.rc == h2.rc)
        return true;
    if (h1.rc != h2.rc) return false;
    const bool h1_first = h1.q_start < h2.q_start;
    const MatchRec &u = h1_first ? h1 : h2;
    const MatchRec &v = h1_first ? h2 : h1;
    if (u.rc) {
        if (u.r_start <= v.r_start || gap_too_long(v.q_start, u.q_end, u.r_start, v.r_end, g)) return false;
    } else if (v.r_start <= u.r_start || gap_too_long(v.q_start, u.q_end, v.r_start, u.r_end, g)) {
        return false;
    }
    return true;
}

__device__ __forceinline__ MatchRec rd_match(const MatchRec &m, int l) {
    MatchRec r;
    r.q_start = rdlane(m.q_start, l);
    r.q_end = rdlane(m.q_end, l);
    r.r_start = rdlane(m.r_start, l);
    r.r_end = rdlane(m.r_end, l);
    r.count = rdlane(m.count, l);
    r.ref = rdlane(m.ref, l);
    r.rc = rdlane(m.rc, l);
    r.done = 0;
    return r;
}

// Per-reference Chain::get_match + best-of + find_coords.  CH = lanes used per chunk (64; smaller only in tests
// so that ordinary inputs exercise the multi-chunk path).
// first: lane i's record i already in a register (the caller read the first chunk itself); with nm <= CH nothing is read here then
template <int CH>
__device__ __forceinline__ void chain_stage(MatchRec *__restrict__ scratch, uint32_t nm, const DevParams &P, uint64_t q_len,
                                            const uint64_t *__restrict__ ref_lens, mq_hit &out, const MatchRec *first = nullptr) {
    const uint32_t lane = lane_id();
    // find_largest_two_chains state (src/mers.rs:110-129)
    uint32_t max_count = 0, second_count = 0, n_cand = 0;
    MatchRec bfirst = {}, blast = {};
    uint32_t b_ref = 0, b_score = 0, b_mapq = 0, b_len = 0;
    for (uint32_t c0 = 0; c0 < nm; c0 += CH) {
        const bool v0 = lane < (uint32_t)CH && c0 + lane < nm;
        MatchRec m = {};
        if (first && c0 == 0) m = *first;
        else if (v0) m = scratch[c0 + lane];
        uint64_t pending = __ballot(v0 && !m.done);
        while (pending) {
            const int lead = __ffsll((long long)pending) - 1;
            const uint32_t r = rdlane(m.ref, lead);
            // pass A: Chain::len and find_largest_match (src/chain.rs:93-104): first index with the largest count
            uint32_t n_in = 0, best_cnt = 0;
            MatchRec anchor = {};
            for (uint32_t c = c0; c < nm; c += CH) {
                const bool v = lane < (uint32_t)CH && c + lane < nm;
                MatchRec x = {};
                if (c == c0) x = m;
                else if (v) x = scratch[c + lane];
                const bool in = v && x.ref == r && !x.done;
                const uint64_t im = __ballot(in);
                if (!im) continue;
                n_in += (uint32_t)__popcll(im);
                const uint32_t cnt = in ? x.count : 0u;
                const uint32_t mx = wave_max_u32(cnt);
                if (mx > best_cnt) {
                    const int fl = __ffsll((long long)__ballot(in && cnt == mx)) - 1;
                    best_cnt = mx;
                    anchor = rd_match(x, fl);
                }
            }
            // pass B: filter_matches_max (src/chain.rs:123-129) when len > 1; first/last kept, score
            uint32_t nkept = 0, score = 0;
            MatchRec first = {}, last = {};
            if (n_in == 1u) {
                // a reference with ONE Match (the stray hit of a repeat copy: most of the extra candidates of a repetitive genome): Chain::len
                // == 1, nothing to filter -- that Match (the anchor: this chunk's lead lane) is the first and the last, its count the score
                nkept = 1u;
                score = best_cnt;
                first = last = anchor;
            } else
            for (uint32_t c = c0; c < nm; c += CH) {
                const bool v = lane < (uint32_t)CH && c + lane < nm;
                MatchRec x = {};
                if (c == c0) x = m;
                else if (v) x = scratch[c + lane];
                const bool in = v && x.ref == r && !x.done;
                const uint64_t im = __ballot(in);
                if (!im) continue;
                if (c != c0 && in) scratch[c + lane].done = 1u;
                const bool keep = in && match_compatible(anchor, x, P.g);
                const uint64_t km = __ballot(keep);
                if (!km) continue;
                if (nkept == 0) first = rd_match(x, __ffsll((long long)km) - 1);
                last = rd_match(x, 63 - __clzll((long long)km));
                nkept += (uint32_t)__popcll(km);
                score += wave_sum_u32(keep ? x.count : 0u);
            }
            pending &= ~__ballot(v0 && m.ref == r);
            if (nkept == 0) continue;  // len_f == 0 => None (cannot happen: the anchor is compatible with itself)
            // find_largest_two_chains update (strict >)
            n_cand++;
            if (score > max_count) {
                second_count = max_count;
                max_count = score;
                bfirst = first;
                blast = last;
                b_ref = r;
                b_score = score;
                b_len = nkept;
                b_mapq = ((P.s != 0 && P.c != 0) && (nkept >= P.c || score >= P.s)) ? 60u : 0u;
            } else if (score > second_count) {
                second_count = score;
            }
        }
    }
    out.status = MQ_HIT_UNMAPPED;
    out.ref_id = 0;
    out.rc = 0;
    out.mapq = 0;
    out.q_start = out.q_end = out.r_start = out.r_end = out.score = out.q_start_hi = out.q_end_hi = 0;
    if (n_cand == 0) return;
    if (n_cand > 1 && max_count == second_count) return;  // determine_best_match: tie => None (src/mers.rs:106)
    // get_match coordinates (src/chain.rs:162-168), usize arithmetic
    const bool rc = bfirst.rc != 0;
    const uint64_t q_start = bfirst.q_start;
    const uint64_t q_end = (uint64_t)blast.q_end - 1;
    uint64_t r_start, r_end;
    if (rc && b_len > 1) {
        r_start = blast.r_start;
        r_end = (uint64_t)bfirst.r_end - 1;
    } else {
        r_start = bfirst.r_start;
        r_end = (uint64_t)blast.r_end - 1;
    }
    // find_coords (src/mers.rs:131-183)
    const uint64_t r_len = ref_lens[b_ref];
    const uint64_t tail = q_len - q_end - 1;
    uint64_t frs, fre, exc_s, exc_e;
    if (!rc) {
        if (r_start >= q_start) { frs = r_start - q_start; exc_s = q_start; }
        else { frs = 0; exc_s = r_start; }
        if (r_end + tail <= r_len - 1) { fre = r_end + tail; exc_e = tail; }
        else { fre = r_len - 1; exc_e = r_len - r_end - 1; }
    } else {
        if (r_end + q_start <= r_len - 1) { fre = r_end + q_start; exc_s = q_start; }
        else { fre = r_len - 1; exc_s = r_len - r_end - 1; }
        if (r_start >= tail) { frs = r_start - tail; exc_e = tail; }
        else { frs = 0; exc_e = r_start; }
    }
    out.status = MQ_HIT_MAPPED;
    out.ref_id = b_ref;
    out.rc = rc ? 1u : 0u;
    out.mapq = b_mapq;
    // usize arithmetic, wrapping like a release build: exc_s / exc_e wrap when the run's r_end lies beyond THIS reference's end
    // (a run the precedence quirk extended onto another, longer reference); all 64 bits go out
    const uint64_t qs64 = q_start - exc_s, qe64 = q_end + exc_e;
    out.q_start = (uint32_t)qs64;
    out.q_end = (uint32_t)qe64;
    out.q_start_hi = (uint32_t)(qs64 >> 32);
    out.q_end_hi = (uint32_t)(qe64 >> 32);
    out.r_start = (uint32_t)frs;
    out.r_end = (uint32_t)fre;
    out.score = b_score;
}

// ------------------------------------------------------------------ the map stage's consumer of an ordered minimizer list
// Match records a wave keeps in LDS next to the staged list (the space the seed phase's larger LDS leaves unused during the map phase):
// an ordinary read has a handful, and chaining them from LDS saves the wait for their stores and the loads' round trip through L2
constexpr uint32_t MAP_LDS_RECS = 48;
static_assert(MAP_LDS_RECS <= 64, "records_to_scratch: one record per lane");
struct MapSink {
    const Bucket *__restrict__ table;
    uint64_t mask;
    const DevParams &P;
    MatchRec *__restrict__ scratch;
    uint32_t cap_matches;
    MatchRec *lds_rec;  // the first MAP_LDS_RECS Match records also go to LDS: a read with no more than that is chained from there
    mq_kminmer *__restrict__ dump;  // optional k-min-mer dump window for this read
    uint32_t dump_cap;
    bool rev_eq;  // seeding variant 32: a palindromic tuple counts as reversed (a constant false in code built without the variants)
    // wave-uniform state
    uint32_t kmm_count = 0;
    uint32_t n_matches = 0;
    bool c_open = false;  // a run is open across the batch boundary; M holds it so far
    MatchRec M = {};
    uint32_t c_hit = 0, c_id = 0, c_off = 0, c_sigma = 0;  // last element of the previous batch
    uint32_t probe_steps = 0;  // per lane: slots visited beyond the home slot (diagnostic: mean probes per lookup)

    __device__ MapSink(const Bucket *t, uint64_t m, const DevParams &p, MatchRec *s, uint32_t cap, MatchRec *lr, mq_kminmer *d, uint32_t dc, bool re)
        : table(t), mask(m), P(p), scratch(s), cap_matches(cap), lds_rec(lr), dump(d), dump_cap(dc), rev_eq(re) {}
    // record i of the read: the first MAP_LDS_RECS stay in LDS, later ones go to their place in the wave's scratch in device memory
    // (records_to_scratch() then moves the LDS ones there too).  An ordinary read's records never leave LDS: no store's
    // acknowledgement is outstanding when the map phase ends.
    __device__ __forceinline__ void put_rec(uint32_t i, const MatchRec &m) const {
        // a typed LDS store: left generic, the compiler folds the two branches into ONE flat store through a selected pointer (26 flat stores
        // in round 5's map_kernel) -- which goes the memory path's way even when it lands in LDS and counts on both wait counters
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) u32x4 lds_u4;
        if (i < MAP_LDS_RECS) {
            lds_u4 *d = (lds_u4 *)(&lds_rec[i]);
            d[0] = u32x4{m.q_start, m.q_end, m.r_start, m.r_end};
            d[1] = u32x4{m.count, m.ref, m.rc, m.done};
        } else if (i < cap_matches) {
            scratch[i] = m;
        }
    }
    __device__ __forceinline__ void records_to_scratch() const {
        const uint32_t n = n_matches < MAP_LDS_RECS ? n_matches : MAP_LDS_RECS, i = lane_id();
        if (i < n && i < cap_matches) scratch[i] = lds_rec[i];
    }

    // k-min-mer `base + lane` of a minimizer list: canonical orientation, tuple hash, query coordinates
    __device__ __forceinline__ void batch_keys(const unsigned long long *mzh, const uint32_t *mzp, uint32_t base, bool act,
                                               uint64_t &key, bool &rev, uint32_t &q_start, uint32_t &q_end) const {
        const uint32_t i0 = base + lane_id();
        key = 0;
        rev = false;
        q_start = q_end = 0;
        if (act) {
            auto get = [&](uint32_t i) { return (uint64_t)mzh[i0 + i]; };
            // 5: the reference's default k (src/main.rs: -k 5); 7: experiments/table1.sh:50; 8: example/run_ecoli.sh:26
            const bool re = rev_eq, fk = P.fast_kh != 0;
            key = P.k == 5u ? kminmer_hash_fixed<5>(get, rev, re, fk) : P.k == 7u ? kminmer_hash_fixed<7>(get, rev, re, fk)
                  : P.k == 8u ? kminmer_hash_fixed<8>(get, rev, re, fk) : kminmer_hash(P.k, get, rev, re, fk);
            q_start = mzp[i0];
            q_end = mzp[i0 + P.k - 1] + P.l - 1u;
        }
    }

    // ---- index probes of a whole chunk of lane-batches, resolved together (ReadOnlyIndex::get, src/index.rs:118-126).
    // Per (lane, batch) one 16-byte load fetches both keys of the home bucket; every batch's load is in flight before the first is
    // looked at.  A key that matches gets its 16 payload bytes fetched at once (same 64-byte line: the L2 has it); the few lookups
    // whose home bucket is full of other keys walk on bucket by bucket in rounds shared by all batches.  Memory latencies a chunk
    // exposes: one for the keys, one for the payloads, one per extra round (rare) -- not one or more per batch.
    // state of a lookup, two bits per batch in `st`: 0 = miss, 1 = key found (payload load issued into kk[c]), 3 = walking on
    __device__ __forceinline__ const Bucket *home_bucket(uint64_t key) const {
        return table + (key == 0 ? (mask + 1) >> 1 : (key & mask) >> 1);
    }
    template <int NB>
    __device__ __forceinline__ void probe_all(const uint64_t (&key)[NB], uint4 (&kk)[NB], uint32_t actbits, uint32_t &st) {
        st = 0;
        const uint64_t nb = (mask + 1) >> 1;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if ((actbits >> c) & 1u) {
                const uint64_t k = key[c];
                const Bucket *B = home_bucket(k);
                uint32_t way, s2;
                if (k == 0) {  // the extra bucket's way 0: its payload says whether the key 0 is present
                    s2 = 1u;
                    way = 0;
                } else {
                    const uint32_t w0 = (uint32_t)k & (uint32_t)mask & 1u;
                    const uint64_t ka = u64_of(kk[c].x, kk[c].y), kb = u64_of(kk[c].z, kk[c].w);
                    const uint64_t kh = w0 ? kb : ka, kp = w0 ? ka : kb;
                    if (kh == k) { s2 = 1u; way = w0; }
                    else if (kh == 0) { s2 = 0u; way = 0; }
                    else if (kp == k) { s2 = 1u; way = w0 ^ 1u; probe_steps++; }
                    else if (kp == 0) { s2 = 0u; way = 0; probe_steps++; }
                    else { s2 = 3u; way = 0; probe_steps++; }
                }
                if (s2 == 1u) kk[c] = ld_u4(&B->pay[way]);
                st |= s2 << (2 * c);
            }
        }
#ifdef MQ_STAGE_MAP_SPLIT
        mq_clk(12);
#endif
        // walking on (about one lookup in a hundred: both ways of the home bucket hold other keys).  Every lane takes ONE of its
        // walking lookups at a time -- its key by a select over the batches, a loop over the following buckets shared by all lanes,
        // the outcome written back by a select -- so the rare path holds six registers instead of looping over all NB batches.
        while (__ballot((st & 0xAAAAAAAAu) != 0)) {
            const bool walking = (st & 0xAAAAAAAAu) != 0;
            const uint32_t wc = walking ? ((uint32_t)__builtin_ctz(st & 0xAAAAAAAAu) >> 1) : 0xFFu;  // the lane's first batch in state 3
            uint64_t k = 0;
#pragma unroll
            for (int c = 0; c < NB; ++c) k = wc == (uint32_t)c ? key[c] : k;
            uint64_t b = (k & mask) >> 1;
            uint4 res = make_uint4(0, 0, 0, 0);
            uint32_t s2 = walking ? 3u : 0u;
            while (__ballot(s2 == 3u)) {
                if (s2 == 3u) {
                    b = (b + 1) & (nb - 1);  // a power of two of buckets
                    const uint4 v = ld_u4(&table[b].key[0]);
                    const uint64_t ka = u64_of(v.x, v.y), kb = u64_of(v.z, v.w);
                    uint32_t way = 0;
                    probe_steps++;
                    if (ka == k) { s2 = 1u; }
                    else if (ka == 0) { s2 = 0u; }
                    else {
                        probe_steps++;
                        if (kb == k) { s2 = 1u; way = 1u; }
                        else if (kb == 0) { s2 = 0u; }
                    }
                    if (s2 == 1u) res = ld_u4(&table[b].pay[way]);
                }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (wc == (uint32_t)c) {
                    kk[c] = res;
                    st = (st & ~(3u << (2 * c))) | (s2 << (2 * c));
                }
            }
        }
    }

    // one batch of n <= 64 consecutive k-min-mers (lane = k-min-mer) with their index entries: dump + Match runs
    __device__ __forceinline__ void batch_runs(uint32_t n, uint64_t key, bool rev, uint32_t q_start, uint32_t q_end, bool hit, const Entry &e) {
        const uint32_t lane = lane_id();
        if (dump && lane < n && kmm_count + lane < dump_cap) {
            mq_kminmer d;
            d.hash = key;
            d.start = q_start;
            d.end = q_end;
            d.offset = kmm_count + lane;
            d.rev = rev ? 1u : 0u;
            dump[kmm_count + lane] = d;
        }
        kmm_count += n;
        // chain_matches + Match::extend (src/mers.rs:57-73, src/match.rs:45-58), lane = k-min-mer.
        // Run state after element i: 0 = no open run (miss), 1 = forward run, 2 = reverse run (Match.rc of the run's
        // first element).  Element i maps the previous state to the next one; the maps compose associatively, so a wave
        // scan resolves every state.  Match::check (src/match.rs:39-43) parses as (A && B && C) || D:
        //   reverse run continues iff same ref id && (q.rev != r.rc) && p.offset - r.offset == 1   (cr)
        //   forward run continues iff r.offset - p.offset == 1                                      (cf)
        const uint32_t r_id = e.id_rc >> 1;
        const uint32_t srel = (hit && (rev != ((e.id_rc & 1u) != 0))) ? 1u : 0u;  // q.rev != r.rc
        // previous element's entry: DPP wave shift right by one lane; lane 0 has no source lane and keeps `old` = the carried element
        const uint32_t hit_p = (uint32_t)__builtin_amdgcn_update_dpp((int)c_hit, (int)(hit ? 1u : 0u), 0x138, 0xf, 0xf, false);
        const uint32_t id_p = (uint32_t)__builtin_amdgcn_update_dpp((int)c_id, (int)r_id, 0x138, 0xf, 0xf, false);
        const uint32_t off_p = (uint32_t)__builtin_amdgcn_update_dpp((int)c_off, (int)e.offset, 0x138, 0xf, 0xf, false);
        const bool both = hit && hit_p;
        const bool cf = both && ((int32_t)(e.offset - off_p) == 1);
        const bool cr = both && r_id == id_p && srel == 1u && ((int32_t)(off_p - e.offset) == 1);
        // Element i maps the previous state to: miss -> 0; hit -> sv = (srel ? 2 : 1) unless it continues a run
        // (state 1 && cf -> 1, state 2 && cr -> 2).  cr implies sv == 2 and cf with sv == 1 gives 1 either way, so every
        // element is a CONSTANT map except "cf && srel" (1 -> 1, anything else -> 2), which is idempotent under
        // composition: the state after it is 1 iff the nearest preceding constant element (or the carry) left state 1.
        const uint32_t sv = srel ? 2u : 1u;
        const bool nonconst = cf && srel == 1u;
        uint32_t sigma = hit ? sv : 0u;
        if (__ballot(nonconst)) {  // rare (a forward step between entries whose strands disagree with the read's): most batches skip this
            const uint64_t constmask = __ballot(!nonconst);
            const uint64_t constF = __ballot(!nonconst && hit && sv == 1u);
            const uint64_t lt_mask = lane ? (~0ull >> (64u - lane)) : 0ull;  // lanes below this one
            const uint64_t below = constmask & lt_mask;
            const bool baseF = below ? ((constF >> (63 - __clzll((long long)below))) & 1ull) != 0 : (c_sigma == 1u);
            if (nonconst) sigma = baseF ? 1u : 2u;
        }
        const uint32_t sigma_p = (uint32_t)__builtin_amdgcn_update_dpp((int)c_sigma, (int)sigma, 0x138, 0xf, 0xf, false);  // lane 0: the carried state
        const bool prevF = sigma_p == 1u, prevR = sigma_p == 2u;
        const bool isnew = hit && !((prevF && cf) || (prevR && cr));
        const uint64_t hitmask = __ballot(hit);
        const uint64_t newmask = __ballot(isnew);
        // the run carried in from the previous batch ends unless element 0 continues it
        const bool cont0 = (hitmask & 1ull) && !(newmask & 1ull);
        if (c_open && n > 0 && !cont0) {
            if (lane == 0) put_rec(n_matches, M);
            n_matches++;
            c_open = false;
        }
        if (n > 0) {
            // element i ends a run iff it is a hit and the next element is a miss or starts a new run;
            // the last element of the batch never ends one here (its run stays open for the next batch / finish()).
            const uint64_t next_breaks = ((~hitmask | newmask) >> 1);
            const bool is_end = hit && lane + 1u < n && ((next_breaks >> lane) & 1ull);
            const uint64_t endmask = __ballot(is_end);
            const uint64_t below = newmask & ((lane >= 63u) ? ~0ull : ((2ull << lane) - 1ull));
            const int fl = below ? 63 - __clzll((long long)below) : -1;  // first element of this lane's run
            const int src = fl < 0 ? 0 : fl;
            const uint32_t f_qs = (uint32_t)__shfl((int)q_start, src, 64);
            const uint32_t f_rs = (uint32_t)__shfl((int)e.start, src, 64);
            const uint32_t f_re = (uint32_t)__shfl((int)e.end, src, 64);
            const uint32_t f_id = (uint32_t)__shfl((int)r_id, src, 64);
            const uint32_t f_rc = (uint32_t)__shfl((int)srel, src, 64);
            MatchRec m;
            if (fl >= 0) {
                m.q_start = f_qs;
                m.rc = f_rc;
                m.ref = f_id;
                m.count = lane - (uint32_t)fl + 1u;
                m.r_start = f_rc ? e.start : f_rs;  // Match::update: rc moves r_start, forward moves r_end
                m.r_end = f_rc ? f_re : e.end;
            } else {  // the run began in an earlier batch: extend the carried Match
                m.q_start = M.q_start;
                m.rc = M.rc;
                m.ref = M.ref;
                m.count = M.count + lane + 1u;
                m.r_start = M.rc ? e.start : M.r_start;
                m.r_end = M.rc ? M.r_end : e.end;
            }
            m.q_end = q_end;
            m.done = 0;
            if (is_end) {
                const uint32_t mi = n_matches + mbcnt64(endmask);
                put_rec(mi, m);
            }
            n_matches += (uint32_t)__popcll(endmask);
            // carry out: state of the last element
            const int last = (int)n - 1;
            c_hit = (uint32_t)((hitmask >> last) & 1ull);
            c_id = rdlane(r_id, last);
            c_off = rdlane(e.offset, last);
            c_sigma = rdlane(sigma, last);
            if (c_hit) {
                M.q_start = rdlane(m.q_start, last);
                M.q_end = rdlane(m.q_end, last);
                M.r_start = rdlane(m.r_start, last);
                M.r_end = rdlane(m.r_end, last);
                M.count = rdlane(m.count, last);
                M.ref = rdlane(m.ref, last);
                M.rc = rdlane(m.rc, last);
                M.done = 0;
                c_open = true;
            } else {
                c_open = false;
            }
        }
    }

    // All k-min-mers of an ordered minimizer list (have <= 64*NB + k - 1 entries) in one go: every tuple hash first, every home
    // bucket's keys in flight together, the probes resolved together (probe_all), then the runs batch by batch.
    // hashes_done(): called once the minimizers' hashes (mzh) have all been read -- the positions (mzp) are still needed
    // mzq (variant 16 only, else nullptr): the minimizers' second positions (the window's last compressed base), valid once
    // hashes_done() has returned (the caller stages them where the hashes were)
    template <int NB, class F>
    __device__ __forceinline__ void consume_list(const unsigned long long *mzh, const uint32_t *mzp, uint32_t have, const F &hashes_done,
                                                 const uint32_t *mzq = nullptr) {
        if (have < P.k) return;
        const uint32_t lane = lane_id();
        const uint32_t K = have - P.k + 1u;
        uint64_t key[NB];
        uint4 kk[NB];
        uint32_t revbits = 0, actbits = 0;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            key[c] = 0;
            kk[c] = make_uint4(0, 0, 0, 0);
            if ((uint32_t)c * 64u < K) {
                const bool act = (uint32_t)c * 64u + lane < K;
                bool rev;
                uint32_t qs, qe;
                batch_keys(mzh, mzp, (uint32_t)c * 64u, act, key[c], rev, qs, qe);
                revbits |= (rev ? 1u : 0u) << c;
                if (act) {
                    actbits |= 1u << c;
                    kk[c] = ld_u4(&home_bucket(key[c])->key[0]);
                }
                __builtin_amdgcn_sched_barrier(0);  // one tuple hash at a time: interleaving NB of them costs ~10 registers each
            }
        }
        hashes_done();
        mq_clk(6);
        uint32_t st;
        probe_all<NB>(key, kk, actbits, st);
#ifdef MQ_STAGE_MAP_SPLIT
        mq_clk(13);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mq_clk(14);
#endif
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if ((uint32_t)c * 64u < K) {
                const uint32_t n = K - (uint32_t)c * 64u < 64u ? K - (uint32_t)c * 64u : 64u;
                const bool act = lane < n;
                const uint32_t i0 = (uint32_t)c * 64u + lane;
                uint32_t qs = 0, qe = 0;
                if (act) {
                    qs = mzp[i0];
                    qe = mzq ? mzq[i0 + P.k - 1] : mzp[i0 + P.k - 1] + P.l - 1u;
                }
                Entry e;
                e.start = kk[c].x;
                e.end = kk[c].y;
                e.offset = kk[c].z;
                e.id_rc = kk[c].w;
                const bool hit = ((st >> (2 * c)) & 3u) == 1u && entry_live(e);
                batch_runs(n, key[c], ((revbits >> c) & 1u) != 0, qs, qe, hit, e);
            }
        }
        mq_clk(7);
    }

    // the run still open after the last k-min-mer of the read ends there
    __device__ __forceinline__ void finish_runs() {
        if (c_open) {
            if (lane_id() == 0) put_rec(n_matches, M);
            n_matches++;
            c_open = false;
        }
    }

};

// sink of the general seeder when it writes lists: ordered minimizers to an SoA list region (hash[], pos[]) -- the split
// pipeline's reads and the reference segments the fast seeder declined
struct SoaListSink {
    unsigned long long *__restrict__ out_hash;
    uint32_t *__restrict__ out_pos;
    uint32_t *__restrict__ out_last;  // variant 16 only (else nullptr): the second position of every minimizer
    uint32_t cap;
    uint32_t written = 0;  // may exceed cap (the map stage then reports the read as overflowed)
    __device__ SoaListSink(unsigned long long *h, uint32_t *p, uint32_t *q, uint32_t c) : out_hash(h), out_pos(p), out_last(q), cap(c) {}
    __device__ __forceinline__ void on_minimizers(WaveLds &S, uint32_t &mz_count) {
        const uint32_t lane = lane_id();
        for (uint32_t base = 0; base < mz_count; base += 64u) {
            const uint32_t i = base + lane;
            if (i < mz_count && written + i < cap) {
                out_hash[written + i] = S.mz_hash[i];
                out_pos[written + i] = S.mz_pos[i];
                if (out_last) out_last[written + i] = S.mz_last[i];
            }
        }
        written += mz_count;
        mz_count = 0;
        wave_sync();
    }
};

}  // namespace mq
