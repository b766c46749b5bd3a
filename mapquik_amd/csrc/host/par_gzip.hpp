// par_gzip.hpp -- ONE gzip member inflated by many threads.
//
// A deflate stream is sequential twice over: a block can only be found by decoding the one before it, and a match may copy
// from the 32 KB before the block.  Both are worked around the way rapidgzip / pugz do it (published technique; nothing of the
// kind is in the reference, whose get_reader hands a .gz to a single-threaded flate2 decoder, src/main.rs:60-75):
//   * the member is cut into segments of compressed bytes; the thread of segment j > 0 SEARCHES a block start at or after its
//     cut: a bit position where a dynamic-Huffman block header parses under strict rules (complete code sets, an end-of-block
//     code), the whole block decodes, every literal of it is text, and the next block header parses too;
//   * from there it decodes into 16-BIT SYMBOLS: a literal is its byte, a copy out of the 32 KB it cannot know is the marker
//     0x8000 + position in that window.  Copies of copies carry the markers along;
//   * thread j stops at the block boundary where thread j+1 started -- it has to land on that bit EXACTLY, otherwise the start
//     of j+1 was a false one, its output is dropped and j decodes on through it (so a stream in which no start can be found --
//     stored blocks, binary data, giant blocks -- is decoded by segment 0 alone: slow, never wrong);
//   * when the segments of a round are done, their windows are resolved in order (the last 32 KB of segment j, translated, is
//     what segment j+1's markers point into: 32 KB of work per segment) and then all segments are translated to bytes in
//     parallel through a 64 K-entry table, CRC-32 computed per segment and combined; the member's trailer (CRC-32, ISIZE)
//     is checked at its end exactly as a sequential decoder would.
// Rounds bound the memory (symbols are twice the output) and let the caller parse a round's bytes while the next one inflates.
#pragma once
#include <zlib.h>  // crc32, crc32_combine

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

namespace mapquik {
namespace pargz {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

constexpr uint32_t WIN = 32768;
constexpr int LIT_PRIM = 11, DIST_PRIM = 9, CL_PRIM = 7;
constexpr uint32_t K_LIT = 0, K_BASE = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4;
// table entry: bits 0-7 code length (sub-table pointer: index bits), 8-10 kind, 11-15 extra bits, 16-31 value
static inline constexpr uint32_t mk_entry(uint32_t val, uint32_t kind, uint32_t extra, uint32_t nbits) { return (val << 16) | (extra << 11) | (kind << 8) | nbits; }
static inline uint32_t e_kind(uint32_t e) { return (e >> 8) & 7u; }

struct Tables {
    uint32_t lit[(1u << LIT_PRIM) + 288 * 16];
    uint32_t dist[(1u << DIST_PRIM) + 32 * 64];
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t bitrev(uint32_t c, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) r |= ((c >> i) & 1u) << (len - 1 - i);
    return r;
}

// Canonical Huffman code (RFC 1951 3.2.2) -> look-up table indexed by the next `prim` bits of the stream, longer codes through
// sub-tables.  kind: 0 = literal/length alphabet, 1 = distance alphabet, 2 = code-length alphabet.  Which sets are acceptable follows
// zlib's inflate (an over-subscribed set never; an incomplete one only as a single one-bit code; no code at all is a table whose use is an error).
static bool build_table(const uint8_t *lens, int n, int prim, uint32_t *tab, int kind) {
    uint16_t count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    int maxl = 15;
    while (maxl > 0 && !count[maxl]) --maxl;
    const uint32_t bad = mk_entry(0, K_BAD, 0, 1);
    for (uint32_t i = 0; i < (1u << prim); ++i) tab[i] = bad;
    if (maxl == 0) return true;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    if (left > 0 && maxl != 1) return false;
    uint16_t next[16];
    {
        uint32_t code = 0;
        count[0] = 0;
        for (int l = 1; l <= 15; ++l) {
            code = (code + count[l - 1]) << 1;
            next[l] = (uint16_t)code;
        }
    }
    auto entry_of = [&](int sym, int len) -> uint32_t {
        if (kind == 0) {
            if (sym < 256) return mk_entry((uint32_t)sym, K_LIT, 0, (uint32_t)len);
            if (sym == 256) return mk_entry(0, K_EOB, 0, (uint32_t)len);
            if (sym < 286) return mk_entry(LEN_BASE[sym - 257], K_BASE, LEN_EXTRA[sym - 257], (uint32_t)len);
            return mk_entry(0, K_BAD, 0, (uint32_t)len);
        }
        if (kind == 1) {
            if (sym < 30) return mk_entry(DIST_BASE[sym], K_BASE, DIST_EXTRA[sym], (uint32_t)len);
            return mk_entry(0, K_BAD, 0, (uint32_t)len);
        }
        return mk_entry((uint32_t)sym, K_LIT, 0, (uint32_t)len);
    };
    // codes longer than prim: the sub-table of a prefix is as wide as its longest code needs
    uint8_t sub_bits[1u << LIT_PRIM];
    bool any_long = maxl > prim;
    if (any_long) {
        memset(sub_bits, 0, (size_t)1 << prim);
        uint16_t nx[16];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < n; ++s) {
            const int len = lens[s];
            if (!len) continue;
            const uint32_t c = nx[len]++;
            if (len > prim) {
                const uint32_t pre = bitrev(c, len) & ((1u << prim) - 1u);
                if (sub_bits[pre] < len - prim) sub_bits[pre] = (uint8_t)(len - prim);
            }
        }
        uint32_t used = 1u << prim;
        for (uint32_t pre = 0; pre < (1u << prim); ++pre)
            if (sub_bits[pre]) {
                tab[pre] = mk_entry(used, K_SUB, 0, sub_bits[pre]);
                for (uint32_t i = 0; i < (1u << sub_bits[pre]); ++i) tab[used + i] = bad;
                used += 1u << sub_bits[pre];
            }
    }
    for (int s = 0; s < n; ++s) {
        const int len = lens[s];
        if (!len) continue;
        const uint32_t c = next[len]++;
        const uint32_t r = bitrev(c, len);
        const uint32_t e = entry_of(s, len);
        if (len <= prim) {
            for (uint32_t i = r; i < (1u << prim); i += 1u << len) tab[i] = e;
        } else {
            const uint32_t pre = r & ((1u << prim) - 1u);
            const uint32_t off = tab[pre] >> 16, sb = tab[pre] & 0xFFu;
            for (uint32_t i = r >> prim; i < (1u << sb); i += 1u << (len - prim)) tab[off + i] = e;
        }
    }
    return true;
}

struct Bits {
    const uint8_t *base = nullptr, *p = nullptr, *end = nullptr;
    uint64_t buf = 0;
    int n = 0;  // valid bits in buf; negative after reading past the end of the input
    void init(const uint8_t *b, const uint8_t *e, uint64_t bitpos) {
        base = b;
        end = e;
        p = b + (bitpos >> 3);
        if (p > end) p = end;
        buf = 0;
        n = 0;
        refill();
        const int k = (int)(bitpos & 7u);
        buf >>= k;
        n -= k;
    }
    inline void refill() {
        if (p + 8 <= end) {
            uint64_t v;
            memcpy(&v, p, 8);
            buf |= v << n;
            p += (63 - n) >> 3;
            n |= 56;
        } else {
            while (n <= 56 && p < end) {
                buf |= (uint64_t)*p++ << n;
                n += 8;
            }
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1ull)); }
    inline void drop(int k) {
        buf >>= k;
        n -= k;
    }
    inline uint32_t take(int k) {
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
    uint64_t bitpos() const { return (uint64_t)(p - base) * 8u - (uint64_t)(int64_t)n; }
};

struct TextSet {
    bool ok[256];
    TextSet() {
        for (int i = 0; i < 256; ++i) ok[i] = (i >= 32 && i < 127) || i == '\n' || i == '\r' || i == '\t';
    }
};
static const TextSet TEXT;

static const Tables &fixed_tables() {
    static const Tables *T = [] {
        Tables *t = new Tables();
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        build_table(l, 288, LIT_PRIM, t->lit, 0);
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(d, 32, DIST_PRIM, t->dist, 1);
        return t;
    }();
    return *T;
}

enum Rc { RC_OK = 0, RC_CORRUPT = 1, RC_TRUNCATED = 2, RC_NOSPACE = 3, RC_NOT_TEXT = 4, RC_NOMEM = 5 };

// dynamic block header (after the 3 header bits) -> t.  strict: what a block START SEARCH demands on top of validity.
static Rc read_dynamic_header(Bits &b, Tables &t, bool strict) {
    b.refill();
    const uint32_t hlit = b.take(5) + 257u, hdist = b.take(5) + 1u, hclen = b.take(4) + 4u;
    if (hlit > 286u || hdist > 30u) return RC_CORRUPT;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    b.refill();
    for (uint32_t i = 0; i < hclen; ++i) {
        if (i == 12) b.refill();
        cl[order[i]] = (uint8_t)b.take(3);
    }
    if (b.n < 0) return RC_TRUNCATED;
    uint32_t clt[1u << CL_PRIM];
    if (!build_table(cl, 19, CL_PRIM, clt, 2)) return RC_CORRUPT;
    uint8_t lens[286 + 30 + 8];
    uint32_t i = 0;
    const uint32_t total = hlit + hdist;
    while (i < total) {
        b.refill();
        if (b.n <= 0) return RC_TRUNCATED;
        const uint32_t e = clt[b.peek(CL_PRIM)];
        if (e_kind(e) != K_LIT) return RC_CORRUPT;
        b.drop((int)(e & 0xFFu));
        const uint32_t sym = e >> 16;
        if (sym < 16) {
            lens[i++] = (uint8_t)sym;
        } else {
            uint32_t rep, val = 0;
            if (sym == 16) {
                if (i == 0) return RC_CORRUPT;
                val = lens[i - 1];
                rep = 3u + b.take(2);
            } else if (sym == 17) {
                rep = 3u + b.take(3);
            } else {
                rep = 11u + b.take(7);
            }
            if (i + rep > total) return RC_CORRUPT;
            while (rep--) lens[i++] = (uint8_t)val;
        }
    }
    if (b.n < 0) return RC_TRUNCATED;
    if (lens[256] == 0) return RC_CORRUPT;  // no end-of-block code
    if (!build_table(lens, (int)hlit, LIT_PRIM, t.lit, 0)) return RC_CORRUPT;
    if (!build_table(lens + hlit, (int)hdist, DIST_PRIM, t.dist, 1)) return RC_CORRUPT;
    if (strict) {
        // a block start found by search: gzip/zlib/pigz/libdeflate emit COMPLETE literal/length sets, and either a complete distance set or at most one code
        uint32_t nl = 0, nd = 0, left = 1u << 15, dleft = 1u << 15;
        for (uint32_t s = 0; s < hlit; ++s)
            if (lens[s]) {
                nl++;
                left -= 1u << (15 - lens[s]);
            }
        for (uint32_t s = 0; s < hdist; ++s)
            if (lens[hlit + s]) {
                nd++;
                dleft -= 1u << (15 - lens[hlit + s]);
            }
        if (nl < 2 || left != 0) return RC_CORRUPT;
        if (nd > 1 && dleft != 0) return RC_CORRUPT;
    }
    return RC_OK;
}

// The symbols of one Huffman block up to its end-of-block code.  out: 16-bit symbols; `lowest`: the earliest symbol a copy may
// reach (data start - what is known / assumed to precede it).  CHECK_TEXT: any literal outside TEXT ends the attempt.
template <bool CHECK_TEXT>
static Rc inflate_huffman_block(Bits &b, const Tables &t, uint16_t *&outp, uint16_t *out_end, const uint16_t *lowest) {
    uint16_t *out = outp;
    const uint32_t *lit = t.lit, *dt = t.dist;
    Rc rc = RC_OK;
    for (;;) {
        if (b.n < 48) b.refill();
        uint32_t e = lit[b.buf & ((1u << LIT_PRIM) - 1u)];
        if (__builtin_expect((e & 0x700u) == (K_SUB << 8), 0)) e = lit[(e >> 16) + ((uint32_t)(b.buf >> LIT_PRIM) & ((1u << (e & 0xFFu)) - 1u))];
        b.buf >>= (e & 0xFFu);
        b.n -= (int)(e & 0xFFu);
        if ((e & 0x700u) == 0) {  // literal
            if (CHECK_TEXT && !TEXT.ok[e >> 16]) {
                rc = RC_NOT_TEXT;
                break;
            }
            if (__builtin_expect(out >= out_end, 0)) {
                rc = RC_NOSPACE;
                break;
            }
            *out++ = (uint16_t)(e >> 16);
            // a second and third literal out of the bits already there (a literal code is at most 15 bits: 48 cover three)
            e = lit[b.buf & ((1u << LIT_PRIM) - 1u)];
            if ((e & 0x700u) == 0 && (!CHECK_TEXT || TEXT.ok[e >> 16]) && out < out_end) {
                b.buf >>= (e & 0xFFu);
                b.n -= (int)(e & 0xFFu);
                *out++ = (uint16_t)(e >> 16);
                e = lit[b.buf & ((1u << LIT_PRIM) - 1u)];
                if ((e & 0x700u) == 0 && (!CHECK_TEXT || TEXT.ok[e >> 16]) && out < out_end) {
                    b.buf >>= (e & 0xFFu);
                    b.n -= (int)(e & 0xFFu);
                    *out++ = (uint16_t)(e >> 16);
                }
            }
            if (__builtin_expect(b.n < 0, 0)) {
                rc = RC_TRUNCATED;
                break;
            }
            continue;
        }
        const uint32_t kind = (e >> 8) & 7u;
        if (kind == K_EOB) {
            if (b.n < 0) rc = RC_TRUNCATED;
            break;
        }
        if (__builtin_expect(kind != K_BASE, 0)) {
            rc = b.n < 0 ? RC_TRUNCATED : RC_CORRUPT;
            break;
        }
        const uint32_t lx = (e >> 11) & 31u;
        const uint32_t len = (e >> 16) + (uint32_t)(b.buf & ((1u << lx) - 1u));
        b.buf >>= lx;
        b.n -= (int)lx;
        uint32_t d = dt[b.buf & ((1u << DIST_PRIM) - 1u)];
        if (__builtin_expect((d & 0x700u) == (K_SUB << 8), 0)) d = dt[(d >> 16) + ((uint32_t)(b.buf >> DIST_PRIM) & ((1u << (d & 0xFFu)) - 1u))];
        b.buf >>= (d & 0xFFu);
        b.n -= (int)(d & 0xFFu);
        if (__builtin_expect(((d >> 8) & 7u) != K_BASE, 0)) {
            rc = b.n < 0 ? RC_TRUNCATED : RC_CORRUPT;
            break;
        }
        const uint32_t dx = (d >> 11) & 31u;
        const uint32_t dist = (d >> 16) + (uint32_t)(b.buf & ((1u << dx) - 1u));
        b.buf >>= dx;
        b.n -= (int)dx;
        if (__builtin_expect(b.n < 0, 0)) {
            rc = RC_TRUNCATED;
            break;
        }
        if (__builtin_expect((size_t)(out - lowest) < dist, 0)) {
            rc = RC_CORRUPT;  // a copy from before the start of the data
            break;
        }
        if (__builtin_expect((size_t)(out_end - out) < len + 4u, 0)) {
            rc = RC_NOSPACE;
            break;
        }
        const uint16_t *src = out - dist;
        if (dist >= 4) {
            for (uint32_t i = 0; i < len; i += 4) memcpy(out + i, src + i, 8);  // up to 3 symbols of slop, overwritten by what follows
        } else {
            for (uint32_t i = 0; i < len; ++i) out[i] = src[i];
        }
        out += len;
    }
    outp = out;
    return rc;
}

// One block at b (its 3 header bits first).  final: the block's BFINAL bit.
template <bool CHECK_TEXT>
static Rc inflate_block(Bits &b, Tables &scratch, uint16_t *&out, uint16_t *out_end, const uint16_t *lowest, bool &final, bool strict = false) {
    b.refill();
    if (b.n < 3) return RC_TRUNCATED;
    final = b.take(1) != 0;
    const uint32_t type = b.take(2);
    if (type == 0) {
        b.drop(b.n & 7);  // to the byte boundary
        b.refill();
        if (b.n < 32) return RC_TRUNCATED;
        const uint32_t len = b.take(16), nlen = b.take(16);
        if ((len ^ 0xFFFFu) != nlen) return RC_CORRUPT;
        // the bytes follow at a byte boundary: go back to plain byte addressing
        const uint8_t *q = b.p - (b.n >> 3);
        if ((size_t)(b.end - q) < len) return RC_TRUNCATED;
        if ((size_t)(out_end - out) < len) return RC_NOSPACE;
        for (uint32_t i = 0; i < len; ++i) {
            if (CHECK_TEXT && !TEXT.ok[q[i]]) return RC_NOT_TEXT;
            out[i] = q[i];
        }
        out += len;
        b.p = q + len;
        b.buf = 0;
        b.n = 0;
        return RC_OK;
    }
    if (type == 1) return inflate_huffman_block<CHECK_TEXT>(b, fixed_tables(), out, out_end, lowest);
    if (type == 3) return RC_CORRUPT;
    const Rc rc = read_dynamic_header(b, scratch, strict);
    if (rc != RC_OK) return rc;
    return inflate_huffman_block<CHECK_TEXT>(b, scratch, out, out_end, lowest);
}

// Is bit position `at` the start of a dynamic block (see the file header for what is demanded)?
static bool block_starts_at(const uint8_t *in, const uint8_t *in_end, uint64_t at, Tables &scratch, uint16_t *sym, size_t sym_cap) {
    Bits b;
    b.init(in, in_end, at);
    if (b.n < 17) return false;
    const uint32_t h = b.peek(17);
    // BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29 (the cheap part of the test, for 7 of 8 positions the only one)
    if ((h & 7u) != 4u || ((h >> 3) & 31u) > 29u || ((h >> 8) & 31u) > 29u) return false;
    uint16_t *out = sym + WIN;
    bool final = false;
    if (inflate_block<true>(b, scratch, out, sym + sym_cap, sym, final, true) != RC_OK) return false;
    if (out == sym + WIN) return false;  // an empty block proves nothing
    // what follows has to look like a block too
    b.refill();
    if (b.n < 3) return false;
    const uint32_t nt = (b.peek(3) >> 1) & 3u;
    if (nt == 3) return false;
    if (nt == 2) {
        b.drop(3);
        if (read_dynamic_header(b, scratch, true) != RC_OK) return false;
    }
    return true;
}

// an anonymous mapping whose pages exist once written (symbols of a segment: sized for the worst case, touched as far as needed)
struct Region {
    void *p = nullptr;
    size_t cap = 0;
    Region() = default;
    explicit Region(size_t n) { reset(n); }
    bool huge = false;
    void reset(size_t n) {
        release();
        cap = ((n + (2u << 20) - 1) / (2u << 20)) * (2u << 20);
        p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            p = nullptr;
            throw Error("cannot map memory for a gzip segment");
        }
        if (huge) madvise(p, cap, MADV_HUGEPAGE);
    }
    void release() {
        if (p) munmap(p, cap);
        p = nullptr;
    }
    ~Region() { release(); }
    Region(const Region &) = delete;
    Region &operator=(const Region &) = delete;
};

struct Options {
    int threads = 8;
    uint64_t seg_bytes = 6u << 20;     // compressed bytes per segment (upper bound; a round = threads segments)
    uint64_t min_seg_bytes = 1u << 20;  // below this a member's rest is decoded by segment 0 alone
    uint32_t max_ratio = 48;            // symbols reserved per compressed byte of a segment (beyond: RC_NOSPACE -> the round is redone by segment 0 with room)
    bool timing = false;
    uint32_t (*crc_fn)(uint32_t, const void *, size_t) = nullptr;  // a faster CRC-32 than zlib's when there is one (libdeflate_crc32: same convention)
};

// symbols -> bytes: blocks without a marker (nearly all of them: markers die out as the text they stand for stops being copied)
// are narrowed, the others go through the table
static inline void translate(const uint16_t *sy, uint8_t *o, uint64_t n, const uint8_t *lut) {
    uint64_t i = 0;
    for (; i + 64 <= n; i += 64) {
        uint16_t any = 0;
        for (int k = 0; k < 64; ++k) any |= sy[i + k];
        if (!(any & 0x8000u)) {
            for (int k = 0; k < 64; ++k) o[i + k] = (uint8_t)sy[i + k];
        } else {
            for (int k = 0; k < 64; ++k) o[i + k] = lut[sy[i + k]];
        }
    }
    for (; i < n; ++i) o[i] = lut[sy[i]];
}

// Inflates the gzip member at in[0, in_len) into out[out_pos, out_cap).  on_round(total bytes of the member so far, finished):
// called after every round (the bytes before `total` are final).  Returns the compressed size of the member (header + deflate
// data + trailer); *produced = its inflated size.  Throws Error on a corrupt / truncated member or when out_cap is too small
// (what() says which; "space" asks the caller for a larger buffer).
class MemberInflater {
  public:
    MemberInflater(const uint8_t *in, uint64_t in_len, const Options &opt) : in_(in), in_len_(in_len), opt_(opt) {
        if (opt_.threads < 1) opt_.threads = 1;
    }

    uint64_t run(uint8_t *out, uint64_t out_cap, uint64_t *produced, const std::function<void(uint64_t, bool)> &on_round) {
        const uint64_t hdr = header_size();
        const uint8_t *din = in_ + hdr;
        const uint8_t *dend = in_ + in_len_;
        uint64_t bit = 0;       // where the next round starts (exact)
        uint64_t total = 0;     // bytes of the member produced so far
        uint32_t crc = 0;  // CRC-32 of nothing
        bool final = false;
        const int T = opt_.threads;
        uint64_t max_ratio = opt_.max_ratio;
        rounds_ = max_chain_ = 0;
        win_.assign(WIN, 0);
        segs_.clear();
        for (int j = 0; j < T; ++j) segs_.emplace_back(new Seg());
        while (!final) {
            const uint64_t byte0 = bit >> 3, bit0 = bit;
            const uint64_t rest = (uint64_t)(dend - din) - byte0;
            // segments of this round: whole rounds of threads x seg_bytes, the member's rest (up to 1.5 rounds' worth) in equal parts
            int ns = T;
            uint64_t S = opt_.seg_bytes;
            bool last_round = false;
            if (rest < (uint64_t)T * S * 3 / 2) {
                last_round = true;
                S = rest / (uint64_t)T;
                if (S < opt_.min_seg_bytes) {
                    S = opt_.min_seg_bytes;
                    ns = (int)(rest / S);
                    if (ns < 1) ns = 1;
                }
            }
            const uint64_t round_end_bit = last_round ? ~0ull : (byte0 + (uint64_t)ns * S) * 8u;
            for (int j = 0; j < ns; ++j) {
                Seg &s = *segs_[(size_t)j];
                s.nominal_bit = j == 0 ? bit : (byte0 + (uint64_t)j * S) * 8u;
                s.sync.store(j == 0 ? (int64_t)bit : PENDING);
                s.n_sym = 0;
                s.end_bit = 0;
                s.final = false;
                s.next_seg = -1;
                s.rc = RC_OK;
                s.nospace = false;
                const uint64_t span = (j + 1 < ns || !last_round) ? S : rest - (uint64_t)j * S;
                const size_t cap_sym = (size_t)WIN + (size_t)std::max<uint64_t>(span * max_ratio, 4u << 20) + 64;
                if (!s.sym.p || s.sym.cap < cap_sym * 2) s.sym.reset(cap_sym * 2);
                s.cap_sym = s.sym.cap / 2;
            }
            abort_.store(false);
            // segment 0 knows what precedes it
            const uint64_t known = total < WIN ? total : WIN;
            {
                uint16_t *sy = (uint16_t *)segs_[0]->sym.p;
                for (uint32_t i = 0; i < WIN; ++i) sy[i] = i >= WIN - known ? win_[i] : 0;  // (kept here, not read back from `out`: the caller may have given those pages back)
            }
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            try {
                th.reserve((size_t)ns);
                for (int j = 1; j < ns; ++j) th.emplace_back([&, j] { seg_worker(j, ns, din, dend, round_end_bit, WIN); });
            } catch (const std::exception &) {
                // a thread could not be started (EAGAIN under a pid / thread limit): the segments nobody will decode publish "no start" so
                // that no predecessor waits for them, the started ones are told to stop and joined -- an Error, not std::terminate
                abort_.store(true);
                for (int j = (int)th.size() + 1; j < ns; ++j) {
                    int64_t pend = PENDING;
                    segs_[(size_t)j]->sync.compare_exchange_strong(pend, NONE);
                }
                for (auto &t : th) t.join();
                throw Error("cannot start a thread for a gzip segment");
            }
            seg_worker(0, ns, din, dend, round_end_bit, (uint32_t)known);
            for (auto &t : th) t.join();
            const auto t1 = std::chrono::steady_clock::now();
            // the chain of segments that really follow each other
            std::vector<int> chain;
            {
                int k = 0;
                for (;;) {
                    Seg &s = *segs_[(size_t)k];
                    if (s.rc == RC_TRUNCATED) throw Error("gzip stream truncated");
                    if (s.rc == RC_NOMEM) throw Error("out of memory inflating a gzip member");
                    if (s.rc != RC_OK) throw Error("gzip stream corrupt");
                    chain.push_back(k);
                    if (s.final || s.next_seg < 0 || s.next_seg >= ns) break;
                    k = s.next_seg;
                }
            }
            uint64_t round_bytes = 0;
            for (int k : chain) {
                segs_[(size_t)k]->out_off = total + round_bytes;
                round_bytes += segs_[(size_t)k]->n_sym;
            }
            if (total + round_bytes > out_cap) throw Error("space (the member's output)");
            // windows in order: segment k's markers point into the last 32 KB before it
            std::vector<std::vector<uint8_t>> lut(chain.size());
            {
                std::vector<uint8_t> &win = win_;  // enters as the window before this round, leaves as the one after it
                for (size_t c = 0; c < chain.size(); ++c) {
                    Seg &s = *segs_[(size_t)chain[c]];
                    lut[c].assign(65536, 0);
                    for (uint32_t v = 0; v < 256; ++v) lut[c][v] = (uint8_t)v;
                    if (c > 0) memcpy(&lut[c][0x8000], win.data(), WIN);
                    // this segment's last 32 KB (its prefix included when it produced less), resolved: the next one's window
                    const uint16_t *sy = (const uint16_t *)s.sym.p + s.n_sym;  // = end - WIN of [prefix | data]
                    std::vector<uint8_t> nw(WIN);
                    for (uint32_t i = 0; i < WIN; ++i) nw[i] = lut[c][sy[i]];
                    win.swap(nw);
                }
            }
            // translate + CRC, all segments at once
            {
                std::vector<std::thread> tt;
                std::atomic<size_t> nextc{0};
                auto work = [&] {
                    for (;;) {
                        const size_t c = nextc.fetch_add(1);
                        if (c >= chain.size()) break;
                        Seg &s = *segs_[(size_t)chain[c]];
                        const uint16_t *sy = (const uint16_t *)s.sym.p + WIN;
                        uint8_t *o = out + s.out_off;
                        const uint8_t *L = lut[c].data();
                        const uint64_t n = s.n_sym;
                        translate(sy, o, n, L);
                        uint32_t cr = 0;
                        for (uint64_t q = 0; q < n;) {
                            const uint64_t m = std::min<uint64_t>(n - q, 1u << 30);
                            cr = opt_.crc_fn ? opt_.crc_fn(cr, o + q, (size_t)m) : (uint32_t)crc32(cr, o + q, (uInt)m);
                            q += m;
                        }
                        s.crc = cr;
                    }
                };
                const int nt = (int)std::min<size_t>(chain.size(), (size_t)T);
                try {
                    tt.reserve((size_t)nt);
                    for (int i = 1; i < nt; ++i) tt.emplace_back(work);
                } catch (const std::exception &) {
                    // fewer helpers than wanted: the started ones and this thread share the segments (work() takes them from one counter)
                }
                work();
                for (auto &t : tt) t.join();
            }
            for (int k : chain) crc = (uint32_t)crc32_combine(crc, segs_[(size_t)k]->crc, (z_off_t)segs_[(size_t)k]->n_sym);
            total += round_bytes;
            rounds_++;
            if ((int)chain.size() > max_chain_) max_chain_ = (int)chain.size();
            const Seg &lastseg = *segs_[(size_t)chain.back()];
            final = lastseg.final;
            bit = lastseg.end_bit;
            if (opt_.timing) {
                const auto t2 = std::chrono::steady_clock::now();
                fprintf(stderr, "pargz round: %d segments of %.1f MB, %zu in the chain, %.1f MB out, decode %.3f s, resolve %.3f s\n", ns, S / 1e6, chain.size(),
                        round_bytes / 1e6, std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
            }
            if (!final && chain.size() == 1 && lastseg.end_bit == bit0) {  // no progress: a block that does not fit the symbol buffer, or nonsense
                if (!lastseg.nospace || max_ratio > 4096) throw Error("gzip stream corrupt");
                max_ratio *= 8;
            }
            if (on_round) on_round(total, final);
        }
        // trailer: CRC-32 and ISIZE at the next byte boundary
        const uint64_t tpos = (bit + 7u) >> 3;
        if ((uint64_t)(dend - din) < tpos + 8u) throw Error("gzip stream truncated");
        uint32_t c32, isz;
        memcpy(&c32, din + tpos, 4);
        memcpy(&isz, din + tpos + 4, 4);
        if (c32 != crc || isz != (uint32_t)total) throw Error("gzip stream corrupt");
        *produced = total;
        return hdr + tpos + 8u;
    }

    int rounds() const { return rounds_; }        // of the last run
    int max_chain() const { return max_chain_; }  // segments that followed each other in its best round (1: no block start was found)

  private:
    static constexpr int64_t PENDING = -1, NONE = -2;
    struct Seg {
        uint64_t nominal_bit = 0;
        std::atomic<int64_t> sync{PENDING};  // bit position of the block this segment starts at; NONE: found none
        Region sym;                          // [WIN symbols of what precedes | the segment's symbols]
        size_t cap_sym = 0;
        uint64_t n_sym = 0, end_bit = 0, out_off = 0;
        bool final = false;
        int next_seg = -1;  // the segment whose start this one landed on (>= the number of segments: the round's end)
        Rc rc = RC_OK;
        uint32_t crc = 0;
        bool nospace = false;  // stopped at its last whole block because the symbol buffer was full
    };

    uint64_t header_size() const {
        if (in_len_ < 18 || in_[0] != 0x1f || in_[1] != 0x8b || in_[2] != 8) throw Error("gzip stream truncated or corrupt");
        const uint8_t flg = in_[3];
        if (flg & 0xE0) throw Error("gzip stream corrupt");
        uint64_t p = 10;
        if (flg & 4) {
            if (p + 2 > in_len_) throw Error("gzip stream truncated");
            p += 2u + (in_[p] | (in_[p + 1] << 8));
        }
        for (int f = 0; f < 2; ++f)
            if (flg & (f == 0 ? 8 : 16)) {
                while (p < in_len_ && in_[p]) ++p;
                ++p;
            }
        if (flg & 2) p += 2;
        if (p >= in_len_) throw Error("gzip stream truncated");
        return p;
    }

    // a worker never leaves its segment's start unpublished (its predecessor waits for it) and never lets an exception escape its thread
    void seg_worker(int j, int ns, const uint8_t *din, const uint8_t *dend, uint64_t round_end_bit, uint32_t known) {
        Seg &s = *segs_[(size_t)j];
        try {
            seg_work(j, ns, din, dend, round_end_bit, known);
        } catch (...) {  // out of memory for the tables: the round fails as a whole
            s.rc = RC_NOMEM;
            s.n_sym = 0;
        }
        int64_t pend = PENDING;
        s.sync.compare_exchange_strong(pend, NONE);
    }

    void seg_work(int j, int ns, const uint8_t *din, const uint8_t *dend, uint64_t round_end_bit, uint32_t known) {
        Seg &s = *segs_[(size_t)j];
        uint16_t *sym = (uint16_t *)s.sym.p;
        std::unique_ptr<Tables> tab(new Tables());
        uint64_t start;
        if (j == 0) {
            start = (uint64_t)s.sync.load();
        } else {
            // a block start in [nominal, next segment's nominal), looked for in the segment's first 2 MB (blocks are far smaller)
            for (uint32_t i = 0; i < WIN; ++i) sym[i] = (uint16_t)(0x8000u + i);
            uint64_t lim = j + 1 < ns ? segs_[(size_t)j + 1]->nominal_bit : (uint64_t)(dend - din) * 8u;
            if (lim > s.nominal_bit + (16u << 20)) lim = s.nominal_bit + (16u << 20);
            int64_t found = NONE;
            for (uint64_t at = s.nominal_bit; at < lim; ++at) {
                if ((at & 0xFFFFu) == 0 && abort_.load(std::memory_order_relaxed)) break;
                if (block_starts_at(din, dend, at, *tab, sym, s.cap_sym)) {
                    found = (int64_t)at;
                    break;
                }
            }
            s.sync.store(found, std::memory_order_release);
            if (found == NONE) return;
            start = (uint64_t)found;
        }
        Bits b;
        b.init(din, dend, start);
        uint16_t *out = sym + WIN;
        uint16_t *out_end = sym + s.cap_sym - 8;
        const uint16_t *lowest = sym + WIN - known;
        uint16_t *good_out = out;  // the end of the last whole block
        uint64_t good_bit = start;
        int target = j + 1;  // the segment whose start ends this one
        for (;;) {
            bool fin = false;
            const Rc rc = inflate_block<false>(b, *tab, out, out_end, lowest, fin);
            if (rc == RC_NOSPACE) {  // stop at the last whole block; the next round goes on from there (with more room if that is no progress)
                out = good_out;
                s.end_bit = good_bit;
                s.nospace = true;
                break;
            }
            if (rc != RC_OK) {
                s.rc = rc;
                break;
            }
            const uint64_t at = b.bitpos();
            good_out = out;
            good_bit = at;
            s.end_bit = at;
            if (fin) {
                s.final = true;
                if (j == 0) abort_.store(true);  // the member ends inside the first segment: the others decode for nothing
                break;
            }
            if (j != 0 && abort_.load(std::memory_order_relaxed)) break;
            bool stop = false;
            while (target < ns && at >= segs_[(size_t)target]->nominal_bit) {
                int64_t t;
                while ((t = segs_[(size_t)target]->sync.load(std::memory_order_acquire)) == PENDING) std::this_thread::yield();
                if (t == NONE || (uint64_t)t < at) {
                    ++target;  // no start there, or one this decoder walked over: not a block boundary after all
                    continue;
                }
                if ((uint64_t)t == at) {
                    s.next_seg = target;
                    stop = true;
                }
                break;  // t > at: decode on towards it
            }
            if (!stop && target >= ns && at >= round_end_bit) {
                s.next_seg = ns;
                stop = true;
            }
            if (stop) break;
        }
        s.n_sym = (uint64_t)(out - (sym + WIN));
    }

    const uint8_t *in_;
    uint64_t in_len_;
    Options opt_;
    std::vector<std::unique_ptr<Seg>> segs_;
    std::atomic<bool> abort_{false};
    std::vector<uint8_t> win_;  // the last 32 KB of the member's output so far (right-aligned)
    int rounds_ = 0, max_chain_ = 0;
};

}  // namespace pargz
}  // namespace mapquik
