// fastx_records.hpp -- what a FASTX chunk is and how records are found in it on the host: Chunk, record boundaries (FASTA '>', FASTQ
// '@' / '+' / equal lengths), the in-place parser (parse_chunk), spans from line ends found on the device (spans_from_line_ends), and the
// run-time bindings of liblz4 / libdeflate.  Included by fastx_feeder.hpp.
#pragma once
#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mapquik_hip.h"

namespace mapquik {
namespace feeder {

struct IdSpan {
    uint64_t off;
    uint32_t len;
};

// one unit of work through the pipeline: raw bytes + the reads found in them
struct Chunk {
    size_t seq_no = 0;
    uint8_t *buf = nullptr;  // page-locked (mq_host_alloc)
    uint64_t cap = 0, begin = 0, bytes = 0;  // whole records occupy buf[begin, bytes)
    std::vector<uint64_t> starts;
    std::vector<uint32_t> lens;
    std::vector<IdSpan> ids;
    std::vector<mq_hit> hits;
    std::string paf, unmapped, unmapped_fa;  // formatted output of this chunk
    // set by the whole-member gzip reader: the chunk's bytes still sit in a member's inflate buffer (kept alive by ext_hold);
    // the parser thread that takes the chunk copies them into buf first
    const uint8_t *ext_src = nullptr;
    std::shared_ptr<void> ext_hold;
    // set by the raw FASTA reader when the consumer finds the records itself (on the device: mq_ctx_submit_fasta): buf[begin, bytes)
    // holds whole records, starts / lens / ids are empty until somebody fills them (parse_chunk on the host, or spans_from_line_ends)
    bool unparsed = false;
    // the mapped-FASTA reader: buf points INTO the read-only mapping of the file (the chunk's records as they lie there); own is the pool's
    // allocation (recycle() puts it back).  materialize() gives such a chunk bytes of its own (the irregular-chunk fallback compacts in place).
    uint8_t *own = nullptr;
    uint8_t *locked_at = nullptr;  // the whole pages of the view that the reader page-locked (mq_host_register); Feeder::recycle releases them
    uint64_t locked_len = 0;
    std::vector<uint8_t> priv;
    void materialize() {
        if (!own || buf == own) return;
        priv.assign(buf, buf + bytes);
        priv.resize(bytes + 64);
        buf = priv.data();
    }
    void clear() {
        if (own) buf = own;
        priv.clear();
        unparsed = false;
        ext_src = nullptr;
        ext_hold.reset();
        begin = bytes = 0;
        starts.clear();
        lens.clear();
        ids.clear();
        hits.clear();
        paf.clear();
        unmapped.clear();
        unmapped_fa.clear();
    }
};

struct FeederError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------- record boundaries
constexpr uint64_t NEED_MORE = ~0ull;

// 1: a FASTQ record starts at p; 0: it does not; 2: cannot tell without bytes beyond `end` (never when at_eof)
inline int fastq_record_at(const uint8_t *b, uint64_t p, uint64_t end, bool at_eof) {
    // '@' at a line start whose line after next starts with '+' and whose quality line is as long as its sequence line:
    // tells a header from a quality line that happens to begin with '@'
    if (p >= end) return at_eof ? 0 : 2;
    if (b[p] != '@') return 0;
    const int unknown = at_eof ? 0 : 2;  // at the end of the input an "@" line with fewer than three lines behind it starts no record (a quality line that begins with "@")
    const uint8_t *e1 = (const uint8_t *)memchr(b + p, '\n', end - p);
    if (!e1) return unknown;
    const uint8_t *e2 = (const uint8_t *)memchr(e1 + 1, '\n', b + end - (e1 + 1));
    if (!e2 || e2 + 1 >= b + end) return unknown;
    if (e2[1] != '+') return 0;
    const uint8_t *e3 = (const uint8_t *)memchr(e2 + 1, '\n', b + end - (e2 + 1));
    if (!e3) return unknown;
    const uint8_t *e4 = (const uint8_t *)memchr(e3 + 1, '\n', b + end - (e3 + 1));
    if (!e4) return at_eof ? ((b + end - e3) == (e2 - e1) ? 1 : 0) : 2;  // last record of a file without a final newline
    return (e4 - e3) == (e2 - e1) ? 1 : 0;
}

// first record start at or after `from`; `end` if there is none; NEED_MORE if that cannot be decided inside [.., end)
inline uint64_t next_record_start(const uint8_t *b, uint64_t from, uint64_t end, bool fastq, bool at_eof) {
    uint64_t p = from;
    if (p > 0 && b[p - 1] != '\n') {  // move to the next line start
        const uint8_t *e = (const uint8_t *)memchr(b + p, '\n', end - p);
        if (!e) return at_eof ? end : NEED_MORE;
        p = (uint64_t)(e - b) + 1;
    }
    while (p < end) {
        if (fastq) {
            const int r = fastq_record_at(b, p, end, at_eof);
            if (r == 1) return p;
            if (r == 2) return NEED_MORE;
        } else if (b[p] == '>') {
            return p;
        }
        const uint8_t *e = (const uint8_t *)memchr(b + p, '\n', end - p);
        if (!e) return at_eof ? end : NEED_MORE;
        p = (uint64_t)(e - b) + 1;
    }
    return at_eof ? end : NEED_MORE;
}

// ---------------------------------------------------------------- spans of a chunk whose line ends were found elsewhere
// line_ends: ascending positions of the '\n's of c.buf[c.begin, c.bytes) (a last line without one ends at c.bytes), lpr lines per
// record (mq_ctx_wait_fasta: 2 for FASTA, 4 for FASTQ; header and sequence are a record's first two): the same starts / lens / ids
// parse_chunk gives for such a chunk.
inline void spans_from_line_ends(Chunk &c, const uint32_t *line_ends, uint32_t n_lines, uint32_t lpr = 2) {
    const uint8_t *b = c.buf;
    const uint32_t n = n_lines / lpr;
    c.starts.resize(n);
    c.lens.resize(n);
    c.ids.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t hs = i ? (uint64_t)line_ends[lpr * i - 1] + 1 : c.begin;
        uint64_t he = line_ends[lpr * i];
        const uint64_t ss = he + 1;
        uint64_t se = line_ends[lpr * i + 1];
        if (se > ss && b[se - 1] == '\r') --se;
        if (he > hs + 1 && b[he - 1] == '\r') --he;
        uint64_t s = hs + 1, e = s;  // seq_io's id(): the header line up to its first SPACE
        while (e < he && b[e] != ' ') ++e;
        c.starts[i] = ss;
        c.lens[i] = (uint32_t)(se - ss);
        c.ids[i] = {s, (uint32_t)(e - s)};
    }
}

// ---------------------------------------------------------------- parser: whole records in c.buf[0, c.bytes) -> spans
inline void parse_chunk(Chunk &c, bool fastq) {
    uint8_t *b = c.buf;
    const uint64_t end = c.bytes;
    uint64_t p = c.begin;
    auto line_end = [&](uint64_t from) -> uint64_t {
        const uint8_t *e = (const uint8_t *)memchr(b + from, '\n', end - from);
        return e ? (uint64_t)(e - b) : end;
    };
    auto add_id = [&](uint64_t h0, uint64_t h1) {  // seq_io's id(): the header line up to its first SPACE (a TAB is part of the id)
        if (h1 > h0 + 1 && b[h1 - 1] == '\r') --h1;  // CR-LF files: the CR is not part of the line
        uint64_t s = h0 + 1, e = s;
        while (e < h1 && b[e] != ' ') ++e;
        c.ids.push_back({s, (uint32_t)(e - s)});
    };
    while (p < end) {
        if (b[p] == '\n' || b[p] == '\r') { ++p; continue; }
        if (fastq) {
            if (b[p] != '@') throw FeederError("malformed FASTQ record");
            const uint64_t e1 = line_end(p);
            add_id(p, e1);
            const uint64_t s = e1 + 1 < end ? e1 + 1 : end;
            const uint64_t e2 = line_end(s);
            uint64_t sl = e2 - s;
            if (sl && b[s + sl - 1] == '\r') --sl;
            if (sl >= (1ull << 32)) throw FeederError("sequence length must be < 2^32");
            c.starts.push_back(s);
            c.lens.push_back((uint32_t)sl);
            const uint64_t e3 = e2 < end ? line_end(e2 + 1) : end;                 // '+' line
            uint64_t e4 = e3 < end ? e3 + 1 + (e2 - s) : end;                       // quality: as long as the sequence line
            if (e4 > end || (e4 < end && b[e4] != '\n')) e4 = e3 < end ? line_end(e3 + 1) : end;
            p = e4 < end ? e4 + 1 : end;
        } else {
            if (b[p] != '>') throw FeederError("malformed FASTA record");
            const uint64_t e1 = line_end(p);
            add_id(p, e1);
            uint64_t s = e1 + 1 < end ? e1 + 1 : end;
            uint64_t e2 = line_end(s);
            uint64_t dst = e2;
            if (dst > s && b[dst - 1] == '\r') --dst;
            uint64_t q = e2 < end ? e2 + 1 : end;
            while (q < end && b[q] != '>') {  // further sequence lines: compact them onto the first one
                const uint64_t e = line_end(q);
                uint64_t n = e - q;
                if (n && b[q + n - 1] == '\r') --n;
                if (n) memmove(b + dst, b + q, n);
                dst += n;
                q = e < end ? e + 1 : end;
            }
            if (dst - s >= (1ull << 32)) throw FeederError("sequence length must be < 2^32");
            c.starts.push_back(s);
            c.lens.push_back((uint32_t)(dst - s));
            p = q;
        }
    }
}

// ---------------------------------------------------------------- lz4 frame decoder through liblz4.so.1 (no headers in the image)
struct Lz4 {
    void *lib = nullptr;
    void *ctx = nullptr;
    size_t (*create)(void **, unsigned) = nullptr;
    size_t (*free_)(void *) = nullptr;
    size_t (*decompress)(void *, void *, size_t *, const void *, size_t *, const void *) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    Lz4() {
        lib = dlopen("liblz4.so.1", RTLD_NOW);
        if (!lib) throw FeederError("Error opening compressed file: liblz4.so.1 not found");
        create = (size_t(*)(void **, unsigned))dlsym(lib, "LZ4F_createDecompressionContext");
        free_ = (size_t(*)(void *))dlsym(lib, "LZ4F_freeDecompressionContext");
        decompress = (size_t(*)(void *, void *, size_t *, const void *, size_t *, const void *))dlsym(lib, "LZ4F_decompress");
        is_error = (unsigned (*)(size_t))dlsym(lib, "LZ4F_isError");
        if (!create || !free_ || !decompress || !is_error || is_error(create(&ctx, 100))) throw FeederError("liblz4: LZ4F API not usable");
    }
    ~Lz4() {
        if (ctx) free_(ctx);
        if (lib) dlclose(lib);
    }
};

// ---------------------------------------------------------------- libdeflate through libdeflate.so.0 (no headers in the image)
// Whole-buffer inflate, ~3x zlib's rate on FASTX text.  Optional: without the library everything goes through zlib.
struct Deflate {
    void *lib = nullptr;
    void *(*alloc)(void) = nullptr;
    void (*free_)(void *) = nullptr;
    // enum libdeflate_result: 0 success, 1 bad data, 2 short output, 3 insufficient space
    int (*raw)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;                  // libdeflate_deflate_decompress
    int (*gzip_ex)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;    // libdeflate_gzip_decompress_ex
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;                                      // libdeflate_crc32 (optional)
    Deflate() {
        if (getenv("MQ_FEEDER_NO_LIBDEFLATE")) return;  // test hook: the zlib paths
        lib = dlopen("libdeflate.so.0", RTLD_NOW);
        if (!lib) return;
        alloc = (void *(*)(void))dlsym(lib, "libdeflate_alloc_decompressor");
        free_ = (void (*)(void *))dlsym(lib, "libdeflate_free_decompressor");
        raw = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(lib, "libdeflate_deflate_decompress");
        gzip_ex = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(lib, "libdeflate_gzip_decompress_ex");
        crc = (uint32_t(*)(uint32_t, const void *, size_t))dlsym(lib, "libdeflate_crc32");
        if (!alloc || !free_ || !raw || !gzip_ex) {
            dlclose(lib);
            lib = nullptr;
        }
    }
    ~Deflate() { if (lib) dlclose(lib); }
    bool ok() const { return lib != nullptr; }
};

// an anonymous, huge-page-backed buffer (a gzip member's inflated bytes)
struct BigBuf {
    uint8_t *p = nullptr;
    uint64_t cap = 0;
    explicit BigBuf(uint64_t n) {
        cap = ((n + (2u << 20) - 1) / (2u << 20)) * (2u << 20);
        // address space only: pages exist once written (and go back once parsed), so the mapping is not to be charged in full
        p = (uint8_t *)mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            p = nullptr;
            throw FeederError("cannot map memory for a gzip member");
        }
        madvise(p, cap, MADV_HUGEPAGE);
    }
    ~BigBuf() { if (p) munmap(p, cap); }
    BigBuf(const BigBuf &) = delete;
    BigBuf &operator=(const BigBuf &) = delete;
};

}  // namespace feeder
}  // namespace mapquik
