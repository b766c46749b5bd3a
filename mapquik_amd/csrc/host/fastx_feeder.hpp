// fastx_feeder.hpp -- the read feeder of the native driver: FASTA/FASTQ (raw, gzip, lz4) -> page-locked chunks of raw file
// bytes + per-read spans, ready for mq_ctx_submit_spans.  Replaces what the reference gets from seq_io's worker pool and
// get_reader (src/main.rs:60-75, src/closures.rs:177-187).
//
// Design: a base is copied ONCE on the host (file / inflate output -> page-locked chunk) and never touched again: the
// chunk goes to the GPU as it is (headers, line ends, quality lines), reads are (start, length) spans of it, and the
// kernels fold a-z to A-Z themselves (MQ_FLAG_FOLD_CASE).  Raw files are cut into chunks at record boundaries and read
// with pread by N threads in parallel; compressed input is inflated by one thread straight into chunks (gzip and lz4
// streams are sequential by nature) and parsed by the others; BGZF (bgzip) files are the exception: their blocks are
// independent deflate streams with their sizes in the headers, so the file is indexed once and then read like a raw file,
// every reader thread inflating the blocks of its own chunk.  Multi-line FASTA records are compacted in place.
// Uncompressed FASTQ is the exception to "raw bytes as they are": half of such a file is quality values nobody reads, so the
// file is mapped and every reader copies the header and sequence lines of its records only -- the quality lines are never
// touched (not read from the page cache, not copied, not sent over PCIe).
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/mapquik_hip.h"
#include "par_gzip.hpp"

#include "fastx_records.hpp"

namespace mapquik {
namespace feeder {

// ---------------------------------------------------------------- the feeder
class Feeder {
  public:
    // chunk_bytes: target raw bytes per chunk; n_threads: reader/parser threads
    // alloc/release: where chunk buffers come from (page-locked memory through mq_host_alloc unless a test says otherwise)
    Feeder(const std::string &path, bool fastq, uint64_t chunk_bytes, int n_threads, int max_chunks,
           std::function<void *(size_t)> alloc = mq_host_alloc, std::function<void(void *)> release = mq_host_free,
           std::function<int(void *, size_t)> lock = mq_host_register, std::function<int(void *)> unlock = mq_host_unregister)
        : path_(path), fastq_(fastq), chunk_bytes_(chunk_bytes < 64 ? 64 : chunk_bytes), n_threads_(n_threads < 1 ? 1 : n_threads),
          max_chunks_(max_chunks), alloc_(alloc), release_(release), lock_(lock), unlock_(unlock) {
        auto ends = [&](const char *t) {
            const size_t n = strlen(t);
            return path.size() >= n && path.compare(path.size() - n, n, t) == 0;
        };
        kind_ = ends(".gz") ? 1 : ends(".lz4") ? 2 : 0;
        fd_ = open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw FeederError("Error opening compressed file: " + path);  // get_reader's message (src/main.rs:62)
        struct stat st;
        fstat(fd_, &st);
        file_size_ = (uint64_t)st.st_size;
        if (kind_ == 1 && index_bgzf()) kind_ = 3;  // logical (inflated) size from here on; chunked and read like a raw file
        if (kind_ == 1 && deflate_.ok() && file_size_ > 0) {
            // a plain gzip file: members inflated whole into a buffer of their own -- large ones by all threads, round by round, the
            // pages of a round given back as soon as the parsers have copied its records out (memory stays bounded whatever the
            // file's size); with a single thread a member is one libdeflate call whose whole output has to be resident, so files
            // beyond MQ_GZ_WHOLE_LIMIT then stream through zlib
            const char *lim = getenv("MQ_GZ_WHOLE_LIMIT");
            const uint64_t limit = lim ? strtoull(lim, nullptr, 10) : (4ull << 30);
            const char *pz = getenv("MQ_PARGZ");
            const bool par_possible = !lim && n_threads_ >= 2 && !(pz && atoi(pz) == 0);
            if (file_size_ <= limit || par_possible) {
                const uint8_t *m = (const uint8_t *)mmap(nullptr, file_size_, PROT_READ, MAP_SHARED, fd_, 0);
                if (m != MAP_FAILED) {
                    map_ = m;
                    map_size_ = file_size_;
                    gz_whole_ = true;
                }
            }
        }
        if (kind_ == 0 && fastq_ && file_size_ > 0 && !getenv("MQ_FEEDER_NO_LEAN_FASTQ")) lean_fastq_ = true;
        if (kind_ == 0 || kind_ == 3) {
            if (chunk_bytes_ > file_size_ + 1) chunk_bytes_ = file_size_ + 1;
            n_raw_chunks_ = (size_t)((file_size_ + chunk_bytes_ - 1) / chunk_bytes_);
        }
    }
    ~Feeder() {
        release_buffers();
        if (map_) munmap((void *)map_, map_size_);
        if (fd_ >= 0) close(fd_);
    }

    // The pool's page-locked buffers back to the system, by as many threads as there are buffers (un-pinning and unmapping 33 MB takes
    // ~3 ms and the driver has a dozen): for a consumer that is done with every chunk and wants its teardown short.  The destructor
    // does the same.
    void release_buffers() {
        stop();
        populate_stop_ = true;
        if (populate_thread_.joinable()) populate_thread_.join();
        std::vector<std::thread> th;
        for (auto &c : all_) {
            void *b = c->own ? c->own : c->buf;
            c->own = c->buf = nullptr;
            if (b) th.emplace_back([this, b] { release_(b); });
        }
        for (auto &t : th) t.join();
        all_.clear();
        free_.clear();
        ready_.clear();
        to_parse_.clear();
    }

    // EXPERIMENT (MQ_FEEDER_MAPPED_FASTA=1; off by default).  Raw FASTA whose records the consumer finds (leave_unparsed): map the file
    // now and fill the mapping's page tables in the background -- no byte of the file is read, the kernel only enters the page-cache
    // pages into this process's address space.  A chunk is then a view of the mapping and its copy to the device a DMA out of the page
    // cache.  Measured (profiles/r04_file_h2d.txt, profiles/r04_feeder_scaling.txt): a probe copies from a mapping with full page
    // tables at 46-50 GB/s with two threads and from a fresh one at 12-17; inside the driver, mapped while the reference is indexed,
    // this path reaches 26-31 Gbases/s against 34-35 for pread into page-locked chunks, which therefore stays the default.
    void premap() {
        const char *e = getenv("MQ_FEEDER_MAPPED_FASTA");
        if (!(leave_unparsed_ && kind_ == 0 && !fastq_ && file_size_ > 0) || !e || atoi(e) == 0 || map_) return;
        const uint8_t *m = (const uint8_t *)mmap(nullptr, file_size_, PROT_READ, MAP_SHARED, fd_, 0);
        if (m == MAP_FAILED) return;
        map_ = m;
        map_size_ = file_size_;
        mapped_fasta_ = true;
        page_ = (uint64_t)sysconf(_SC_PAGESIZE);
        lock_pages_ = page_ > 0 && getenv("MQ_FEEDER_PAGE_LOCK") != nullptr;  // experiment: the reader threads page-lock each chunk's pages
        populate_thread_ = std::thread([this] {
            const uint64_t step = 64ull << 20;
            for (uint64_t o = 0; o < map_size_ && !populate_stop_.load(std::memory_order_relaxed); o += step) {
                const uint64_t n = std::min<uint64_t>(step, map_size_ - o);
#ifdef MADV_POPULATE_READ
                if (madvise((void *)(map_ + o), n, MADV_POPULATE_READ) != 0) break;  // an older kernel: pages are entered as they are touched
#else
                if (madvise((void *)(map_ + o), n, 22) != 0) break;
#endif
                populated_.store(o + n, std::memory_order_release);
            }
        });
    }

    void start() {
        if (mapped_fasta_) {
            for (int t = 0; t < n_threads_; ++t) threads_.emplace_back([this] { mapped_fasta_worker(); });
        } else if (lean_fastq_ && !leave_unparsed_) {  // (records found by the consumer: the chunked reader below hands the file's bytes over as they are)
            for (int t = 0; t < n_threads_; ++t) threads_.emplace_back([this] { lean_fastq_worker(); });
        } else if (kind_ == 0 || kind_ == 3) {
            for (int t = 0; t < n_threads_; ++t) threads_.emplace_back([this] { raw_worker(); });
        } else {
            if (gz_whole_) threads_.emplace_back([this] { gzip_member_worker(); });
            else threads_.emplace_back([this] { inflate_worker(); });
            for (int t = 0; t < n_threads_; ++t) threads_.emplace_back([this] { parse_worker(); });
        }
    }

    // the consumer gives up (an error elsewhere in its pipeline): next() returns nullptr from now on, workers waiting for a buffer
    // leave; chunks still held by the consumer need not be recycled
    void abort() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            aborted_ = true;
            stopping_ = true;
        }
        cv_.notify_all();
    }

    // next parsed chunk (any order; seq_no says where it belongs) or nullptr at the end of the input
    Chunk *next() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return aborted_ || !ready_.empty() || finished_locked() || !error_.empty(); });
        if (aborted_) return nullptr;
        if (!error_.empty()) throw FeederError(error_);
        if (ready_.empty()) return nullptr;
        Chunk *c = ready_.front();
        ready_.pop_front();
        return c;
    }
    // next(), without waiting: a parsed chunk if one is ready, else nullptr -- `end` says whether the input is exhausted (or the
    // consumer aborted).  For consumers that hold chunks of their own and must not sit on them while nothing new arrives.
    Chunk *poll(bool &end) {
        std::lock_guard<std::mutex> lk(mu_);
        end = false;
        if (aborted_) {
            end = true;
            return nullptr;
        }
        if (!error_.empty()) throw FeederError(error_);
        if (ready_.empty()) {
            end = finished_locked();
            return nullptr;
        }
        Chunk *c = ready_.front();
        ready_.pop_front();
        return c;
    }
    // hand a chunk back for re-use
    void recycle(Chunk *c) {
        if (c->locked_len) {  // a view of the mapped file whose pages were locked for the copy to the device
            unlock_(c->locked_at);
            c->locked_at = nullptr;
            c->locked_len = 0;
        }
        c->clear();
        {
            std::lock_guard<std::mutex> lk(mu_);
            free_.push_back(c);
        }
        cv_.notify_all();
    }
    size_t chunks_total() const { return produced_.load(); }
    uint64_t bytes_in() const { return file_size_; }
    // Before start(): n buffers of the pool allocated (page-locked) now, by as many threads, so that the first chunks do not wait for
    // them -- the pool's buffers do not depend on the input's bytes
    void preallocate(int n) {
        if (mapped_views()) return;
        std::vector<std::thread> th;
        std::vector<Chunk *> got((size_t)std::max(0, std::min(n, max_chunks_)), nullptr);
        const uint64_t need = std::min<uint64_t>(chunk_bytes_ + (1u << 20) + 2, file_size_ + 2);
        for (size_t i = 0; i < got.size(); ++i)
            th.emplace_back([&, i] {
                try { got[i] = get_buffer(need, true); } catch (const std::exception &) {}
            });
        for (auto &t : th) t.join();
        for (Chunk *c : got)
            if (c) recycle(c);
    }
    // Before start(): chunks of an uncompressed FASTA or FASTQ file are handed over as they were read, Chunk::unparsed set, no host thread
    // having looked at a base (the consumer submits them with mq_ctx_submit_fastx; a chunk that comes back MQ_FASTA_IRREGULAR is
    // parsed with parse_chunk after all).  Compressed input is parsed here as always.
    void leave_unparsed(bool on) { leave_unparsed_ = on; }
    bool fastq() const { return fastq_; }
    // chunks will be (after start(): are) views of the mapped file (see start())
    bool mapped_views() const { return mapped_fasta_; }
    const char *kind_name() const { return kind_ == 0 ? "raw" : kind_ == 1 ? (gz_whole_ ? "gzip (libdeflate, whole members)" : "gzip") : kind_ == 2 ? "lz4" : "bgzf"; }

  private:
    bool finished_locked() const { return done_workers_ == (int)threads_.size(); }

    Chunk *get_buffer(uint64_t need, bool force = false) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            for (auto it = free_.begin(); it != free_.end(); ++it)
                if ((*it)->cap >= need) {
                    Chunk *c = *it;
                    free_.erase(it);
                    return c;
                }
            if (force || (int)all_.size() < max_chunks_ || free_.size() == all_.size()) {  // grow the pool (or replace a too-small buffer when nothing is in flight)
                lk.unlock();
                std::unique_ptr<Chunk> c(new Chunk());
                const uint64_t cap = mapped_fasta_ ? std::max<uint64_t>(need, 64) : std::max<uint64_t>(need, std::min<uint64_t>(chunk_bytes_ + chunk_bytes_ / 8 + (1u << 20), file_size_ + 64));
                c->buf = (uint8_t *)alloc_(cap);
                if (!c->buf) throw FeederError("cannot allocate a chunk buffer");
                c->own = c->buf;
                c->cap = cap;
                lk.lock();
                all_.push_back(std::move(c));
                return all_.back().get();
            }
            if (stopping_) throw FeederError("stopped");
            cv_.wait(lk);
        }
    }

    void publish(Chunk *c) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            ready_.push_back(c);
        }
        produced_++;
        cv_.notify_all();
    }
    void worker_done(const std::string &err) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!err.empty() && error_.empty()) error_ = err;
            done_workers_++;
        }
        cv_.notify_all();
    }

    // mapped FASTA: chunk i = a view of the records whose first byte lies in [i*CH, (i+1)*CH) of the mapping; only the lines around the
    // two cuts are looked at
    void mapped_fasta_worker() {
        std::string err;
        try {
            for (;;) {
                Chunk *c = get_buffer(64);
                const size_t i = next_raw_.fetch_add(1);
                if (i >= n_raw_chunks_) {
                    recycle(c);
                    break;
                }
                const uint64_t lo = (uint64_t)i * chunk_bytes_, hi = std::min<uint64_t>(lo + chunk_bytes_, file_size_);
                const uint64_t first = lo ? next_record_start(map_, lo, file_size_, false, true) : 0;
                const uint64_t last = hi < file_size_ ? next_record_start(map_, hi, file_size_, false, true) : file_size_;
                c->seq_no = i;
                if (first >= last || first >= hi) {
                    c->begin = c->bytes = 0;  // no record starts in this chunk (inside a long record)
                } else {
                    if (last - first >= (1ull << 32)) throw FeederError("sequence length must be < 2^32");
                    c->buf = const_cast<uint8_t *>(map_) + first;
                    c->begin = 0;
                    c->bytes = last - first;
                    c->unparsed = true;
                    // Page-lock the chunk's whole pages [floor(first), floor(last)): the copy to the device is then a DMA out of the page cache
                    // that no thread waits for.  (Experimental, off by default: profiles/r04_file_h2d.txt -- locking the pages of a fresh
                    // mapping runs at 12-17 GB/s whatever the thread count.)  The ranges of consecutive chunks tile the file, so no page is locked twice; the chunk's last partial page belongs to the next
                    // chunk's range, and mq_ctx_submit_fasta moves those < 4 KB through a buffer of its own.  Released by recycle().
                    if (lock_pages_) {
                        const uint64_t a = first / page_ * page_, b = last / page_ * page_;
                        if (b > a && lock_(const_cast<uint8_t *>(map_) + a, (size_t)(b - a)) == 0) {
                            c->locked_at = const_cast<uint8_t *>(map_) + a;
                            c->locked_len = b - a;
                        }
                    }
                }
                publish(c);
            }
        } catch (const std::exception &e) { err = e.what(); }
        worker_done(err);
    }

    // raw file: chunk i owns the records whose first byte lies in [i*CH, (i+1)*CH)
    void raw_worker() {
        std::string err;
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        bool z_ok = false;
        try {
            if (kind_ == 3) {
                if (inflateInit2(&zs, -15) != Z_OK) throw FeederError("inflateInit2 failed");  // raw deflate: BGZF payloads
                z_ok = true;
            }
            for (;;) {
                // buffer first, chunk number second: every numbered chunk then owns a buffer, so the consumer (which may hold
                // later chunks while it waits for an earlier one) can never starve the earliest chunk of memory
                Chunk *c = get_buffer(std::min<uint64_t>(chunk_bytes_ + (1u << 20) + 2, file_size_ + 2));
                const size_t i = next_raw_.fetch_add(1);
                if (i >= n_raw_chunks_) {
                    recycle(c);
                    break;
                }
                const uint64_t lo = (uint64_t)i * chunk_bytes_, hi = std::min<uint64_t>(lo + chunk_bytes_, file_size_);
                // read [lo - 1, hi + tail): one byte before to know whether lo is a line start; the tail until the owning
                // record of hi's successor is complete (grown as needed)
                uint64_t tail = std::min<uint64_t>(1u << 20, file_size_ - hi);
                const uint64_t from = lo ? lo - 1 : 0;
                uint64_t got = 0;  // bytes of [from, ...) already in the buffer: a longer tail only reads what is missing
                for (;;) {
                    const uint64_t want = hi + tail - from;
                    if (c->cap < want) {  // a record longer than the tail: a private, larger buffer (beyond the pool limit if need be)
                        Chunk *big = get_buffer(want, true);
                        if (got) memcpy(big->buf, c->buf, got);
                        recycle(c);
                        c = big;
                    }
                    fetch(c->buf + got, from + got, want - got, zs);
                    got = want;
                    const uint64_t skip = lo ? 1 : 0;  // index of byte `lo` in the buffer
                    const bool at_eof = hi + tail >= file_size_;
                    const uint64_t first = lo ? next_record_start(c->buf, skip, got, fastq_, at_eof) : 0;
                    uint64_t last = got;
                    if (first != NEED_MORE && hi < file_size_) last = next_record_start(c->buf, skip + (hi - lo), got, fastq_, at_eof);
                    if (first == NEED_MORE || last == NEED_MORE) {  // the record that straddles hi is longer than the tail
                        tail = std::min<uint64_t>(std::max<uint64_t>(tail * 4, chunk_bytes_), file_size_ - hi);
                        continue;
                    }
                    if (first >= last || first >= skip + (hi - lo)) {
                        c->begin = c->bytes = 0;  // no record starts in this chunk (inside a long record)
                    } else {
                        c->begin = first;
                        c->bytes = last;
                    }
                    break;
                }
                c->seq_no = i;
                if (leave_unparsed_ && kind_ == 0 && c->bytes > c->begin) c->unparsed = true;  // the consumer finds the records (on the device), FASTA or FASTQ
                else parse_chunk(*c, fastq_);
                publish(c);
            }
        } catch (const std::exception &e) { err = e.what(); }
        if (z_ok) inflateEnd(&zs);
        worker_done(err);
    }

    // uncompressed FASTQ, lean: chunk i owns the records whose first byte lies in [i*CH, (i+1)*CH) of the file and reads their header and
    // sequence lines -- and nothing else -- straight into its page-locked buffer: ONE pread per record of about the record's header +
    // sequence length (the longest of the eight records before it and a margin; what it reads too much, the start of the '+' and
    // quality lines, is overwritten by the next record), the '+' line found in that surplus.  The byte at the place where the quality
    // line must end if it is as long as the sequence line (the validator's test, fastq_record_at) is the byte in FRONT of the next
    // record: it comes with the next record's read (into the place of this record's own line end, which is put back) -- round 5 read
    // it with a pread of its own, a second system call per record, and asked for 1.125 x the record before, which one record in six
    // outgrew (a second, doubled read).  Half the file's bytes never leave the page
    // cache: 1 byte per base from the file and over the link instead of 2 (a reader that maps the file pays for the page tables of
    // all of it: 12-17 GB/s at any thread count, profiles/r04_file_h2d.txt; this one runs at pread's rate).
    void lean_fastq_worker() {
        std::string err;
        try {
            const uint64_t end = file_size_;
            std::vector<uint8_t> win;  // scratch of the boundary searches
            auto rd = [&](uint8_t *dst, uint64_t off, uint64_t n) {
                uint64_t got = 0;
                while (got < n) {
                    const ssize_t r = pread(fd_, dst + got, n - got, (off_t)(off + got));
                    if (r <= 0) throw FeederError("read error: " + path_);
                    got += (uint64_t)r;
                }
            };
            // first record start at or after `from`, decided like the chunked reader's cut (next_record_start over a window that grows until
            // the validator can tell)
            auto record_start_from = [&](uint64_t from) -> uint64_t {
                uint64_t W = 1u << 18;
                for (;;) {
                    const uint64_t a = from - 1, b = std::min<uint64_t>(end, from + W);
                    win.resize((size_t)(b - a));
                    rd(win.data(), a, b - a);
                    const uint64_t r = next_record_start(win.data(), 1, b - a, true, b >= end);
                    if (r != NEED_MORE) return a + r;
                    W *= 4;
                }
            };
            auto byte_at = [&](uint64_t off) -> uint8_t {
                uint8_t x = 0;
                rd(&x, off, 1);
                return x;
            };
            // the line end at or after `from` (file offsets), read in small steps: only for what the surplus of a record's read did not hold
            auto line_end_from = [&](uint64_t from) -> uint64_t {
                uint8_t tmp[4096];
                for (uint64_t q = from; q < end;) {
                    const uint64_t n = std::min<uint64_t>(sizeof(tmp), end - q);
                    rd(tmp, q, n);
                    const uint8_t *e = (const uint8_t *)memchr(tmp, '\n', (size_t)n);
                    if (e) return q + (uint64_t)(e - tmp);
                    q += n;
                }
                return end;
            };
            uint64_t est = 32768;  // bytes to ask for per record: header + sequence line of the records before it, and a margin
            uint64_t hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // header + sequence bytes of the last eight records
            unsigned hist_at = 0;
            for (;;) {
                Chunk *c = get_buffer(std::min<uint64_t>(chunk_bytes_ / 2 + (1u << 20) + 2, file_size_ + 2));
                const size_t i = next_raw_.fetch_add(1);
                if (i >= n_raw_chunks_) {
                    recycle(c);
                    break;
                }
                const uint64_t lo = (uint64_t)i * chunk_bytes_, hi = std::min<uint64_t>(lo + chunk_bytes_, file_size_);
                // [first, last): from the first record start at or after lo to the first one at or after hi -- the same cut as the
                // chunked reader's, so that a last record the validator cannot vouch for (CR-LF file without a final newline)
                // stays with its predecessor
                uint64_t p = lo ? record_start_from(lo) : 0;
                const uint64_t last = hi < end ? record_start_from(hi) : end;
                if (p >= hi) p = last;  // no record starts in this chunk
                uint64_t w = 0;  // bytes of the chunk in use
                bool pending = false;     // the byte at p - 1 (where the record before must end) is still to be looked at: it comes with this record's read
                uint64_t E3_prev = 0;     // end of the '+' line of the record before (where the search for its real end starts when that byte is no '\n')
                while (p < last) {
                    // the record's header and sequence lines into the buffer at w: `est` bytes, more while a line end is missing
                    uint64_t got = 0, e1 = NEED_MORE, e2 = NEED_MORE;  // e1, e2: indices in c->buf of the two line ends (or of the data's end at EOF)
                    bool redo = false;
                    for (;;) {
                        const uint64_t want = std::min<uint64_t>(got ? got * 2 : est, end - p);
                        if (w + want + 64 > c->cap) {  // records longer than the buffer: a private, larger one
                            Chunk *big = get_buffer(std::max<uint64_t>(w + want + 64, 2 * c->cap), true);
                            if (w + got) memcpy(big->buf, c->buf, w + got);
                            big->starts.swap(c->starts);
                            big->lens.swap(c->lens);
                            big->ids.swap(c->ids);
                            recycle(c);
                            c = big;
                        }
                        if (pending) {  // (got == 0, w >= 1: buf[w - 1] is the line end of the record before)
                            rd(c->buf + w - 1, p - 1, want + 1);
                            const uint8_t chk = c->buf[w - 1];
                            c->buf[w - 1] = '\n';
                            pending = false;
                            if (chk != '\n') {  // the quality line of the record before is not as long as its sequence line: to its real end
                                const uint64_t E4r = E3_prev < end ? line_end_from(E3_prev + 1) : end;
                                p = E4r < end ? E4r + 1 : end;
                                redo = true;
                                break;
                            }
                        } else {
                            rd(c->buf + w + got, p + got, want - got);
                        }
                        const uint64_t from = e1 == NEED_MORE ? w : e1 + 1;  // (what was searched already holds no line end)
                        got = want;
                        const bool at_eof = p + got >= end;
                        if (e1 == NEED_MORE) {
                            const uint8_t *e = (const uint8_t *)memchr(c->buf + from, '\n', (size_t)(w + got - from));
                            if (e) e1 = (uint64_t)(e - c->buf);
                            else if (at_eof) e1 = e2 = w + got;
                        }
                        if (e1 != NEED_MORE && e2 == NEED_MORE) {
                            const uint64_t s = e1 + 1 < w + got ? e1 + 1 : w + got;
                            const uint8_t *e = (const uint8_t *)memchr(c->buf + s, '\n', (size_t)(w + got - s));
                            if (e) e2 = (uint64_t)(e - c->buf);
                            else if (at_eof) e2 = w + got;
                        }
                        if (e2 != NEED_MORE) break;
                    }
                    if (redo) continue;
                    if (c->buf[w] == '\n' || c->buf[w] == '\r') {  // blank bytes between records (rare): step over them
                        ++p;
                        continue;
                    }
                    if (c->buf[w] != '@') throw FeederError("malformed FASTQ record");
                    const uint64_t E2 = p + (e2 - w);                            // file offset of the sequence line's end
                    const uint64_t s = e1 + 1 < e2 ? e1 + 1 : e2;                // the sequence line in the buffer: [s, e2)
                    const uint64_t S = p + (s - w);
                    uint64_t sl = e2 - s;
                    if (sl && c->buf[s + sl - 1] == '\r') --sl;
                    if (sl >= (1ull << 32)) throw FeederError("sequence length must be < 2^32");
                    uint64_t h1 = e1;
                    if (h1 > w + 1 && c->buf[h1 - 1] == '\r') --h1;
                    uint64_t ie = w + 1;
                    while (ie < h1 && c->buf[ie] != ' ') ++ie;  // seq_io's id(): up to the first space
                    c->ids.push_back({w + 1, (uint32_t)(ie - (w + 1))});
                    c->starts.push_back(s);
                    c->lens.push_back((uint32_t)sl);
                    // '+' line: its end is in the surplus of the read more often than not; then a quality line as long as the sequence
                    // line (else: to the next line end, like parse_chunk)
                    uint64_t E3 = end;
                    if (E2 < end) {
                        const uint64_t ps = e2 + 1;
                        const uint8_t *e = ps < w + got ? (const uint8_t *)memchr(c->buf + ps, '\n', (size_t)(w + got - ps)) : nullptr;
                        E3 = e ? p + ((uint64_t)(e - c->buf) - w) : line_end_from(p + got);
                    }
                    uint64_t E4 = E3 < end ? E3 + 1 + (E2 - S) : end;
                    if (E4 > end) {
                        E4 = E3 < end ? line_end_from(E3 + 1) : end;
                    } else if (E4 < end) {
                        if (E4 + 1 < last && e2 < w + got) pending = true;  // looked at with the next record's read
                        else if (byte_at(E4) != '\n') E4 = line_end_from(E3 + 1);
                    }
                    E3_prev = E3;
                    hist[hist_at++ & 7u] = E2 - p;
                    uint64_t longest = 0;
                    for (uint64_t hlen : hist) longest = std::max(longest, hlen);
                    est = std::max<uint64_t>(4096, longest + longest / 32 + 256);
                    w = e2 < w + got ? e2 + 1 : e2;  // the next record overwrites what was read beyond the sequence line
                    p = E4 < end ? E4 + 1 : end;
                }
                c->begin = 0;
                c->bytes = w;
                c->seq_no = i;
                publish(c);
            }
        } catch (const std::exception &e) { err = e.what(); }
        worker_done(err);
    }

    // bytes [off, off + n) of the (logical) file into dst
    void fetch(uint8_t *dst, uint64_t off, uint64_t n, z_stream &zs) {
        if (kind_ == 0) {
            uint64_t got = 0;
            while (got < n) {
                const ssize_t r = pread(fd_, dst + got, n - got, (off_t)(off + got));
                if (r <= 0) throw FeederError("read error: " + path_);
                got += (uint64_t)r;
            }
            return;
        }
        // BGZF: the blocks that overlap [off, off + n); a block wholly inside inflates straight into dst
        size_t b = (size_t)(std::upper_bound(bg_uoff_.begin(), bg_uoff_.end(), off) - bg_uoff_.begin()) - 1;
        uint8_t tmp[65536];
        const uint64_t end = off + n;
        for (; b + 1 < bg_uoff_.size() && bg_uoff_[b] < end; ++b) {
            const uint64_t u0 = bg_uoff_[b], u1 = bg_uoff_[b + 1];
            if (u1 == u0) continue;
            const bool whole = u0 >= off && u1 <= end;
            uint8_t *out = whole ? dst + (u0 - off) : tmp;
            const uint8_t *cin = map_ + bg_coff_[b] + bg_hdr_[b];
            const size_t cin_n = (size_t)(bg_coff_[b + 1] - bg_coff_[b] - bg_hdr_[b] - 8);
            if (deflate_.ok()) {
                thread_local struct TlsD {
                    void *d = nullptr;
                    void (*fr)(void *) = nullptr;
                    ~TlsD() { if (d && fr) fr(d); }
                } tls;
                if (!tls.d) {
                    tls.d = deflate_.alloc();
                    tls.fr = deflate_.free_;
                    if (!tls.d) throw FeederError("libdeflate: no decompressor");
                }
                size_t got = 0;
                if (deflate_.raw(tls.d, cin, cin_n, out, (size_t)(u1 - u0), &got) != 0 || got != (size_t)(u1 - u0))
                    throw FeederError("BGZF block corrupt: " + path_);
            } else {
                if (inflateReset(&zs) != Z_OK) throw FeederError("inflateReset failed");
                zs.next_in = const_cast<Bytef *>(cin);
                zs.avail_in = (uInt)cin_n;
                zs.next_out = out;
                zs.avail_out = (uInt)(u1 - u0);
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0) throw FeederError("BGZF block corrupt: " + path_);
            }
            {  // the block's CRC-32 (the four bytes before ISIZE): a damaged block of the right length is an error, as for flate2
                const uint8_t *t = map_ + bg_coff_[b + 1] - 8;
                const uint32_t want = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
                const uint32_t have = deflate_.crc ? deflate_.crc(0, out, (size_t)(u1 - u0)) : (uint32_t)crc32(0L, out, (uInt)(u1 - u0));
                if (have != want) throw FeederError("BGZF block corrupt: " + path_);
            }
            if (!whole) {
                const uint64_t a = std::max(u0, off), e = std::min(u1, end);
                memcpy(dst + (a - off), tmp + (a - u0), e - a);
            }
        }
    }

    // BGZF (bgzip): every block is a gzip member whose extra field 'BC' holds the block size; the last four bytes of a block
    // hold its inflated size.  Returns false (plain gzip) unless the WHOLE file parses as BGZF blocks.
    bool index_bgzf() {
        if (file_size_ < 28) return false;
        const uint8_t *m = (const uint8_t *)mmap(nullptr, file_size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) return false;
        std::vector<uint64_t> coff, uoff;
        std::vector<uint16_t> hdr;
        uint64_t p = 0, u = 0;
        bool ok = true;
        while (p < file_size_) {
            if (p + 18 > file_size_ || m[p] != 0x1f || m[p + 1] != 0x8b || m[p + 2] != 8 || !(m[p + 3] & 4)) { ok = false; break; }
            const uint32_t xlen = m[p + 10] | (m[p + 11] << 8);
            uint32_t bsize = 0;
            for (uint32_t q = 0; q + 4 <= xlen;) {  // subfields: SI1 SI2 SLEN(2) data
                const uint8_t *f = m + p + 12 + q;
                if (p + 12 + q + 4 > file_size_) break;
                const uint32_t sl = f[2] | (f[3] << 8);
                if (f[0] == 'B' && f[1] == 'C' && sl == 2 && p + 12 + q + 6 <= file_size_) bsize = (f[4] | (f[5] << 8)) + 1u;
                q += 4 + sl;
            }
            if (!bsize || bsize < 12 + xlen + 8 || p + bsize > file_size_) { ok = false; break; }
            const uint8_t *t = m + p + bsize - 4;
            const uint32_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
            if (isize > 65536) { ok = false; break; }
            coff.push_back(p);
            uoff.push_back(u);
            hdr.push_back((uint16_t)(12 + xlen));
            p += bsize;
            u += isize;
        }
        if (!ok || coff.empty()) {
            munmap((void *)m, file_size_);
            return false;
        }
        coff.push_back(p);
        uoff.push_back(u);
        map_ = m;
        map_size_ = file_size_;
        bg_coff_.swap(coff);
        bg_uoff_.swap(uoff);
        bg_hdr_.swap(hdr);
        file_size_ = u;  // the logical file
        return true;
    }

    // compressed input: one thread inflates into chunks cut at record boundaries; parse_worker threads parse them
    void inflate_worker() {
        std::string err;
        try {
            std::vector<uint8_t> in(4u << 20);
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            std::unique_ptr<Lz4> lz;
            struct ZEnd {  // inflateEnd on every way out (an exception on a corrupt or truncated stream included)
                z_stream *z = nullptr;
                ~ZEnd() { if (z) inflateEnd(z); }
            } zend;
            if (kind_ == 1) {
                if (inflateInit2(&zs, 15 + 32) != Z_OK) throw FeederError("inflateInit2 failed");
                zend.z = &zs;
            } else {
                lz.reset(new Lz4());
            }
            const uint64_t cap_need = chunk_bytes_ + chunk_bytes_ / 8 + (1u << 20);
            Chunk *c = get_buffer(cap_need);
            size_t seq = 0;
            uint64_t file_pos = 0;
            size_t in_have = 0, in_pos = 0;
            bool eof = false;
            bool mid_stream = false;  // inside a gzip member / an lz4 frame: the input may not end here (flate2's UnexpectedEof)
            auto cut_and_publish = [&](bool final) {
                // keep whole records in c, carry the incomplete last one to a fresh chunk
                Chunk *nxt = nullptr;
                uint64_t keep = c->bytes;
                if (!final) {
                    // last record start in the second half of the buffer
                    uint64_t p = c->bytes / 2, lastrec = 0;
                    for (;;) {  // candidates too close to the end to be validated are not taken: the cut lands on a sure record start
                        const uint64_t q = next_record_start(c->buf, p, c->bytes, fastq_, false);
                        if (q == NEED_MORE || q >= c->bytes) break;
                        lastrec = q;
                        p = q + 1;
                    }
                    if (lastrec == 0) throw FeederError("a single record does not fit a chunk: raise --batch-bases");
                    keep = lastrec;
                    nxt = get_buffer(cap_need);
                    memcpy(nxt->buf, c->buf + keep, c->bytes - keep);
                    nxt->bytes = c->bytes - keep;
                }
                c->bytes = keep;
                c->seq_no = seq++;
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    to_parse_.push_back(c);
                }
                cv_.notify_all();
                c = nxt;
            };
            while (!eof) {
                if (in_pos == in_have) {
                    const ssize_t r = pread(fd_, in.data(), in.size(), (off_t)file_pos);
                    if (r < 0) throw FeederError("read error: " + path_);
                    if (r == 0) {
                        if (mid_stream) throw FeederError(std::string(kind_ == 1 ? "gzip" : "lz4") + " stream truncated: " + path_);
                        break;
                    }
                    file_pos += (uint64_t)r;
                    in_have = (size_t)r;
                    in_pos = 0;
                }
                while (in_pos < in_have) {
                    if (c->bytes + (1u << 16) > c->cap - 64 || c->bytes >= chunk_bytes_) cut_and_publish(false);
                    size_t produced = 0, consumed = 0;
                    if (kind_ == 1) {
                        zs.next_in = in.data() + in_pos;
                        zs.avail_in = (uInt)(in_have - in_pos);
                        zs.next_out = c->buf + c->bytes;
                        zs.avail_out = (uInt)std::min<uint64_t>(c->cap - 64 - c->bytes, 1u << 30);
                        const uInt out0 = zs.avail_out;
                        const int rc = inflate(&zs, Z_NO_FLUSH);
                        consumed = (in_have - in_pos) - zs.avail_in;
                        produced = out0 - zs.avail_out;
                        if (rc == Z_STREAM_END) {
                            mid_stream = false;
                            if (zs.avail_in > 0 || file_pos < file_size_) inflateReset(&zs);  // concatenated gzip members
                        } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
                            throw FeederError("gzip stream corrupt: " + path_);
                        } else if (consumed || produced) {
                            mid_stream = true;
                        }
                    } else {
                        size_t dst = (size_t)(c->cap - 64 - c->bytes), src = in_have - in_pos;
                        const size_t rc = lz->decompress(lz->ctx, c->buf + c->bytes, &dst, in.data() + in_pos, &src, nullptr);
                        if (lz->is_error(rc)) throw FeederError("lz4 stream corrupt: " + path_);
                        consumed = src;
                        produced = dst;
                        mid_stream = rc != 0;  // LZ4F_decompress returns 0 exactly when a frame is complete
                    }
                    in_pos += consumed;
                    c->bytes += produced;
                    if (!consumed && !produced) break;
                }
            }
            cut_and_publish(true);
        } catch (const std::exception &e) { err = e.what(); }
        {
            std::lock_guard<std::mutex> lk(mu_);
            inflate_done_ = true;
        }
        worker_done(err);
    }

    void parse_worker() {
        std::string err;
        try {
            for (;;) {
                Chunk *c = nullptr;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return !to_parse_.empty() || inflate_done_ || stopping_; });
                    if (to_parse_.empty()) break;
                    c = to_parse_.front();
                    to_parse_.pop_front();
                }
                if (c->ext_src) {  // whole-member gzip reader: the bytes come out of the member's inflate buffer here, in parallel
                    memcpy(c->buf, c->ext_src, c->bytes);
                    // the whole pages of this range are not needed again (the inflater keeps its own copy of the last 32 KB): back to
                    // the system, so that a member of any size costs the memory of the rounds in flight
                    const uintptr_t pa = ((uintptr_t)c->ext_src + 4095u) & ~(uintptr_t)4095u, pb = ((uintptr_t)c->ext_src + c->bytes) & ~(uintptr_t)4095u;
                    if (pb > pa) madvise((void *)pa, pb - pa, MADV_DONTNEED);
                    c->ext_src = nullptr;
                    c->ext_hold.reset();
                }
                parse_chunk(*c, fastq_);
                publish(c);
            }
        } catch (const std::exception &e) { err = e.what(); }
        worker_done(err);
    }

    // plain gzip, member by member: a member is inflated whole into a huge-page buffer (behind the unfinished record the previous
    // member may have ended with), cut into chunk-sized ranges at record boundaries, and the ranges are handed to the parser
    // threads, which copy them into page-locked chunk buffers and parse them.  A large member is inflated by all threads
    // (par_gzip.hpp: block starts found by search, 16-bit symbols, windows resolved afterwards) and its ranges go to the parsers
    // round by round while the next round inflates; a small one by one libdeflate call.
    void gzip_member_worker() {
        std::string err;
        void *d = nullptr;
        try {
            d = deflate_.alloc();
            if (!d) throw FeederError("libdeflate: no decompressor");
            size_t seq = 0;
            uint64_t p = 0;
            std::vector<uint8_t> carry;  // the previous member's unfinished last record
            auto envu = [](const char *k, uint64_t dflt) {
                const char *v = getenv(k);
                return v ? strtoull(v, nullptr, 10) : dflt;
            };
            const bool par_on = envu("MQ_PARGZ", 1) != 0 && n_threads_ >= 2;
            const uint64_t par_min = envu("MQ_PARGZ_MIN", 16u << 20);  // compressed bytes from which a member is worth many threads
            bool prev_small = false;                                   // a file of many small members: do not start a round of threads for each
            while (p < file_size_) {
                if (file_size_ - p < 18 || map_[p] != 0x1f || map_[p + 1] != 0x8b) throw FeederError("gzip stream truncated or corrupt: " + path_);
                const uint64_t rest = file_size_ - p;
                const bool par_ok = par_on && rest >= par_min;
                bool par = par_ok && !prev_small;
                // address space; pages exist once written.  All threads: what deflate can expand to at most (1032 : 1), within 16 TB (but
                // 64 : 1 at least), since pages behind the parsers go back to the system; one call: 12 : 1, doubled when it was not enough.
                // A one-call attempt made only because the PREVIOUS member was small is bounded (a small member fits 16 x par_min): a
                // member that does not fit is a large one after all and goes to all threads, so that a tiny first member in front of a
                // multi-GB one does not make the large one inflate fully resident.
                auto cap_for = [&](bool all_threads) -> uint64_t {
                    if (all_threads) return std::max<uint64_t>(64 * rest, std::min<uint64_t>(1100 * rest, 16ull << 40)) + carry.size();
                    const uint64_t one = std::max<uint64_t>(64u << 20, 12 * rest);
                    return (par_ok ? std::min<uint64_t>(one, std::max<uint64_t>(64u << 20, 16 * par_min)) : one) + carry.size();
                };
                uint64_t cap = cap_for(par);
                std::shared_ptr<BigBuf> big;
                size_t ain = 0, aout = 0;
                uint64_t a = 0;  // start of the bytes not yet handed to a parser
                auto hand_over = [&](uint64_t from, uint64_t to) {
                    Chunk *c = get_buffer(to - from + 64, (to - from + 64) > chunk_bytes_ + chunk_bytes_ / 8 + (1u << 20));
                    c->begin = 0;
                    c->bytes = to - from;
                    c->ext_src = big->p + from;
                    c->ext_hold = big;
                    c->seq_no = seq++;
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        to_parse_.push_back(c);
                    }
                    cv_.notify_all();
                };
                for (;;) {
                    for (;;) {  // a refused reservation (strict overcommit accounting) is asked for again at a quarter, down to 2 x the rest of the file
                        try {
                            big = std::make_shared<BigBuf>(cap + 64);
                            break;
                        } catch (const FeederError &) {
                            if (cap / 4 < 2 * (file_size_ - p) + carry.size() + (64u << 20)) throw;
                            cap /= 4;
                        }
                    }
                    if (!carry.empty()) memcpy(big->p, carry.data(), carry.size());
                    const auto tt0 = std::chrono::steady_clock::now();
                    int rc = 0;
                    if (par) {
                        pargz::Options o;
                        o.threads = n_threads_;
                        o.seg_bytes = envu("MQ_PARGZ_SEG", o.seg_bytes);
                        o.min_seg_bytes = envu("MQ_PARGZ_MINSEG", o.min_seg_bytes);
                        o.timing = getenv("MQ_FEEDER_TIMING") != nullptr;
                        o.crc_fn = deflate_.crc;
                        try {
                            pargz::MemberInflater inf(map_ + p, file_size_ - p, o);
                            uint64_t produced = 0;
                            ain = (size_t)inf.run(big->p + carry.size(), cap - carry.size(), &produced, [&](uint64_t so_far, bool finished) {
                                if (finished) return;  // the member's tail is cut below, where it is known whether more members follow
                                const uint64_t avail = carry.size() + so_far;
                                while (a + chunk_bytes_ < avail) {  // ranges that end at a record start found with bytes to spare
                                    const uint64_t q = next_record_start(big->p, a + chunk_bytes_, avail, fastq_, false);
                                    if (q == NEED_MORE || q >= avail) break;
                                    hand_over(a, q);
                                    a = q;
                                }
                            });
                            aout = (size_t)produced;
                        } catch (const pargz::Error &e) {
                            if (strncmp(e.what(), "space", 5) == 0 && a == 0) rc = 3;
                            else throw FeederError(std::string(strncmp(e.what(), "space", 5) == 0 ? "gzip member expands beyond the buffer" : e.what()) + ": " + path_);
                        }
                    } else {
                        rc = deflate_.gzip_ex(d, map_ + p, (size_t)(file_size_ - p), big->p + carry.size(), (size_t)(cap - carry.size()), &ain, &aout);
                    }
                    if (getenv("MQ_FEEDER_TIMING")) fprintf(stderr, "gzip member (%s): rc %d, %zu -> %zu bytes in %.3f s\n", par ? "all threads" : "libdeflate", rc, ain, aout, std::chrono::duration<double>(std::chrono::steady_clock::now() - tt0).count());
                    if (rc == 0) break;
                    if (rc != 3 || cap > (1ull << 37)) throw FeederError("gzip stream truncated or corrupt: " + path_);
                    if (!par && par_ok) {  // not a small member after all
                        par = true;
                        cap = cap_for(true);
                        continue;
                    }
                    cap *= 2;  // insufficient space: a member compressed better than expected
                }
                prev_small = ain < par_min;
                const uint64_t total = carry.size() + aout;
                carry.clear();
                p += ain;
                const bool last_member = p >= file_size_;
                // ranges [a, b): b = the first record start at or after a + chunk_bytes_ (the end of the data in the last member)
                while (a < total) {
                    uint64_t b = total;
                    if (a + chunk_bytes_ < total) {
                        const uint64_t q = next_record_start(big->p, a + chunk_bytes_, total, fastq_, last_member);
                        b = (q == NEED_MORE) ? total : q;
                    }
                    if (b >= total && !last_member) {
                        // the tail may hold an unfinished record: keep everything from the last sure record start for the next member
                        uint64_t lastrec = a, from = a;
                        for (;;) {
                            const uint64_t q = next_record_start(big->p, from, total, fastq_, false);
                            if (q == NEED_MORE || q >= total) break;
                            lastrec = q;
                            from = q + 1;
                        }
                        if (lastrec == a && a != 0) {  // no further record start inside [a, total): all of it is carry
                            carry.assign(big->p + a, big->p + total);
                            break;
                        }
                        if (lastrec > a) {
                            carry.assign(big->p + lastrec, big->p + total);
                            b = lastrec;
                        } else {  // a == 0 and no second record start: the whole member is (part of) one record
                            carry.assign(big->p, big->p + total);
                            break;
                        }
                    }
                    hand_over(a, b);
                    a = b;
                }
            }
            if (!carry.empty()) {  // (cannot happen: the last member's tail is cut with at_eof) -- never drop bytes silently
                Chunk *c = get_buffer(carry.size() + 64, true);
                memcpy(c->buf, carry.data(), carry.size());
                c->begin = 0;
                c->bytes = carry.size();
                c->seq_no = seq++;
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    to_parse_.push_back(c);
                }
                cv_.notify_all();
            }
        } catch (const std::exception &e) { err = e.what(); }
        if (d) deflate_.free_(d);
        {
            std::lock_guard<std::mutex> lk(mu_);
            inflate_done_ = true;
        }
        worker_done(err);
    }

    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stopping_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
    }

    std::string path_;
    bool fastq_;
    uint64_t chunk_bytes_;
    int n_threads_, max_chunks_;
    std::function<void *(size_t)> alloc_;
    std::function<void(void *)> release_;
    std::function<int(void *, size_t)> lock_;  // page-lock / release whole pages of the mapped file (mq_host_register / mq_host_unregister)
    std::function<int(void *)> unlock_;
    int kind_ = 0;  // 0 raw, 1 gzip, 2 lz4, 3 BGZF (indexed, read like raw)
    bool lean_fastq_ = false;  // raw FASTQ read lean: header and sequence lines only (lean_fastq_worker)
    bool leave_unparsed_ = false;
    bool mapped_fasta_ = false;  // raw FASTA, records found by the consumer: chunks are views of the mapped file
    bool lock_pages_ = false;    // ... and their pages are locked for the copy to the device
    uint64_t page_ = 4096;
    bool gz_whole_ = false;    // plain gzip, members inflated whole by libdeflate
    Deflate deflate_;
    const uint8_t *map_ = nullptr;  // BGZF: the compressed file, mapped
    uint64_t map_size_ = 0;
    std::vector<uint64_t> bg_coff_, bg_uoff_;  // per block (+ end): compressed / inflated offsets
    std::vector<uint16_t> bg_hdr_;             // per block: header bytes before the deflate data
    int fd_ = -1;
    uint64_t file_size_ = 0;
    size_t n_raw_chunks_ = 0;
    std::atomic<size_t> next_raw_{0}, produced_{0};
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::unique_ptr<Chunk>> all_;
    std::deque<Chunk *> free_, ready_, to_parse_;
    std::vector<std::thread> threads_;
    std::thread populate_thread_;  // premap(): fills the mapping's page tables ahead of the readers
    std::atomic<bool> populate_stop_{false};
    std::atomic<uint64_t> populated_{0};
    int done_workers_ = 0;
    bool inflate_done_ = false, stopping_ = false, aborted_ = false;
    std::string error_;
};

}  // namespace feeder
}  // namespace mapquik
