// ref_loader.hpp -- the reference FASTA of the native driver, read for ONE purpose: every record, whole, in file order, as fast
// as the host can deliver it to mq_index_add_ref (src/closures.rs:46-94 reads it through seq_io and indexes record by record).
//
// A reference has few, very long records (a human chromosome is one 50-250 MB record), which is the worst case of the chunked
// read feeder (fastx_feeder.hpp): the record that straddles a chunk is read by one thread.  Here the file is read ONCE by all
// threads in parallel (pread of 16-MB blocks into one anonymous, huge-page-backed mapping of the file's size), record starts
// are found by the same threads, multi-line records are compacted in place by a pool (one record per task) while the caller
// already indexes the first ones.  Uncompressed FASTA only: compressed or FASTQ references go through the feeder.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace mapquik {
namespace feeder {

class RefLoader {
  public:
    struct Record {
        std::string id;        // seq_io's id(): the header up to its first space
        uint64_t seq = 0;      // offset of the (compacted) sequence in the buffer
        uint64_t len = 0;
        uint64_t region_end = 0;  // first byte after the record in the file
    };

    // eager: the file is read (by n_threads threads of its own) from the constructor on, while the caller does something else -- the
    // native driver constructs the loader BEFORE its first HIP call: reading 3.1 GB takes 0.08-0.1 s, bringing the HIP runtime up 0.2 s,
    // on different threads.  wait_read() (or for_each) joins.
    RefLoader(const std::string &path, int n_threads, bool eager = false) : path_(path), n_threads_(n_threads < 1 ? 1 : n_threads) {
        fd_ = open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("Error opening compressed file: " + path);  // get_reader's message (src/main.rs:62)
        struct stat st;
        fstat(fd_, &st);
        size_ = (uint64_t)st.st_size;
        mapped_ = ((size_ + 64 + (2u << 20) - 1) / (2u << 20)) * (2u << 20);
        buf_ = (uint8_t *)mmap(nullptr, mapped_, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (buf_ == MAP_FAILED) {
            buf_ = nullptr;
            close(fd_);
            throw std::runtime_error("cannot map memory for the reference: " + path);
        }
        madvise(buf_, mapped_, MADV_HUGEPAGE);
        if (eager)
            reader_ = std::thread([this] {
                try {
                    read_all();
                } catch (const std::exception &e) { read_err_ = e.what(); }
            });
    }
    ~RefLoader() {
        if (reader_.joinable()) reader_.join();
        stop_pool();
        if (buf_) munmap(buf_, mapped_);
        if (fd_ >= 0) close(fd_);
    }
    RefLoader(const RefLoader &) = delete;
    RefLoader &operator=(const RefLoader &) = delete;

    // fn(const Record &, const uint8_t *sequence) for every record, in file order.  The sequence stays valid until the loader dies.
    // the whole file is in the buffer (eager loaders: joins the reading; others: reads now)
    void wait_read() {
        if (read_done_) return;
        if (reader_.joinable()) {
            reader_.join();
            if (!read_err_.empty()) throw std::runtime_error(read_err_);
        } else {
            read_all();
        }
        read_done_ = true;
    }
    // the buffer the file was read into (whole 2-MB pages, huge-page backed: page-locking it costs milliseconds, tools/pin_rate.hip)
    uint8_t *data() const { return buf_; }
    uint64_t mapped_bytes() const { return mapped_; }
    uint64_t file_bytes() const { return size_; }
    template <class F>
    void for_each(F fn) {
        wait_read();
        find_records();
        start_pool();
        for (size_t i = 0; i < recs_.size(); ++i) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return ready_[i] || !error_.empty(); });
                if (!error_.empty()) throw std::runtime_error(error_);
            }
            fn(recs_[i], buf_ + recs_[i].seq);
        }
    }
    size_t n_records() const { return recs_.size(); }

  private:
    static constexpr uint64_t BLOCK = 16u << 20;

    void read_all() {
        const size_t n_blocks = (size_t)((size_ + BLOCK - 1) / BLOCK);
        std::atomic<size_t> next{0};
        std::vector<std::vector<uint64_t>> found(n_blocks);  // per block: offsets of '>' at a line start ('>' at a block's first byte: checked later)
        std::string err;
        std::mutex emu;
        auto work = [&]() {
            try {
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= n_blocks) break;
                    const uint64_t lo = (uint64_t)b * BLOCK, hi = std::min<uint64_t>(lo + BLOCK, size_);
                    uint64_t got = 0;
                    while (lo + got < hi) {
                        const ssize_t r = pread(fd_, buf_ + lo + got, hi - lo - got, (off_t)(lo + got));
                        if (r <= 0) throw std::runtime_error("read error: " + path_);
                        got += (uint64_t)r;
                    }
                    for (uint64_t p = lo; p < hi;) {
                        const uint8_t *q = (const uint8_t *)memchr(buf_ + p, '>', hi - p);
                        if (!q) break;
                        const uint64_t at = (uint64_t)(q - buf_);
                        if (at == lo || buf_[at - 1] == '\n') found[b].push_back(at);
                        p = at + 1;
                    }
                }
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> lk(emu);
                if (err.empty()) err = e.what();
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads_; ++t) th.emplace_back(work);
        for (auto &t : th) t.join();
        if (!err.empty()) throw std::runtime_error(err);
        for (size_t b = 0; b < n_blocks; ++b)
            for (uint64_t at : found[b])
                if (at == 0 || buf_[at - 1] == '\n') starts_.push_back(at);  // block-first candidates: the byte before is there now
    }

    void find_records() {
        // anything before the first record must be blank (seq_io would reject it; so does the feeder's parser)
        const uint64_t first = starts_.empty() ? size_ : starts_[0];
        for (uint64_t p = 0; p < first; ++p)
            if (buf_[p] != '\n' && buf_[p] != '\r') throw std::runtime_error("malformed FASTA record");
        recs_.resize(starts_.size());
        ready_.assign(starts_.size(), 0);
        for (size_t i = 0; i < starts_.size(); ++i) recs_[i].region_end = i + 1 < starts_.size() ? starts_[i + 1] : size_;
    }

    // header -> id; sequence lines compacted in place
    void prepare(size_t i) {
        Record &r = recs_[i];
        const uint64_t h0 = starts_[i], end = r.region_end;
        const uint8_t *e1 = (const uint8_t *)memchr(buf_ + h0, '\n', end - h0);
        uint64_t h1 = e1 ? (uint64_t)(e1 - buf_) : end;
        const uint64_t s = h1 < end ? h1 + 1 : end;
        if (h1 > h0 + 1 && buf_[h1 - 1] == '\r') --h1;
        uint64_t ie = h0 + 1;
        while (ie < h1 && buf_[ie] != ' ') ++ie;
        r.id.assign((const char *)buf_ + h0 + 1, ie - (h0 + 1));
        uint64_t dst = s, q = s;
        while (q < end) {
            const uint8_t *e = (const uint8_t *)memchr(buf_ + q, '\n', end - q);
            const uint64_t le = e ? (uint64_t)(e - buf_) : end;
            uint64_t n = le - q;
            if (n && buf_[q + n - 1] == '\r') --n;
            if (n && dst != q) memmove(buf_ + dst, buf_ + q, n);
            dst += n;
            q = le < end ? le + 1 : end;
        }
        if (dst - s >= (1ull << 32)) throw std::runtime_error("sequence length must be < 2^32");
        r.seq = s;
        r.len = dst - s;
    }

    void start_pool() {
        next_rec_ = 0;
        const int n = (int)std::min<size_t>((size_t)n_threads_, std::max<size_t>(recs_.size(), 1));
        for (int t = 0; t < n; ++t)
            pool_.emplace_back([this] {
                for (;;) {
                    const size_t i = next_rec_.fetch_add(1);
                    if (i >= recs_.size()) return;
                    std::string err;
                    try {
                        prepare(i);
                    } catch (const std::exception &e) { err = e.what(); }
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        if (!err.empty() && error_.empty()) error_ = err;
                        ready_[i] = 1;
                    }
                    cv_.notify_all();
                }
            });
    }
    void stop_pool() {
        next_rec_ = (size_t)-1 / 2;
        for (auto &t : pool_)
            if (t.joinable()) t.join();
        pool_.clear();
    }

    std::string path_;
    int n_threads_;
    int fd_ = -1;
    uint64_t size_ = 0, mapped_ = 0;
    uint8_t *buf_ = nullptr;
    std::thread reader_;
    std::string read_err_;
    bool read_done_ = false;
    std::vector<uint64_t> starts_;
    std::vector<Record> recs_;
    std::vector<char> ready_;
    std::atomic<size_t> next_rec_{0};
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::string error_;
};

// RefStreamer -- the same file, never held in host memory: the common shape of a reference FASTA written by a tool (one header line, one
// sequence line per record) goes to the device as it is read.  Reader threads pread 16-MB blocks into a small pool of page-locked
// chunks and note where the line ends and the '>' at line starts are; the calling thread queues every block's copy into the index's
// staging buffer the moment it is read (mq_index_stage_piece: the PCIe link runs beside the reads, 3.1 GB in ~0.08 s) and recycles a
// chunk when its copy is done; a third thread hands every record to the index as soon as its last block is on its way
// (mq_index_add_ref_staged: the build's kernels wait on the device for the pieces, not on the host).  The host looks at a
// header line only (pread of its few bytes) and never at a base: lower case and CR-LF are the kernels' / the spans' business.
// Anything else -- sequences over several lines, blank lines, text before the first '>' -- ends the run as `irregular` and the caller
// falls back to RefLoader (a line-wrapped FASTA shows in its first block, before anything was indexed).
class RefStreamer {
  public:
    struct Hooks {
        std::function<void *(size_t)> alloc;                                          // page-locked memory (mq_host_alloc)
        std::function<void(void *)> release;
        std::function<uint64_t(uint64_t at, const uint8_t *src, uint64_t n)> piece;   // mq_index_stage_piece: returns the ticket, throws on error
        std::function<bool(uint64_t ticket, bool wait)> done;                         // mq_index_stage_done
    };
    struct Result {
        bool irregular = false;  // not "header line, sequence line" all through
        size_t handed = 0;       // records handed to the callback before that was noticed
        size_t records = 0;
    };
    static constexpr uint64_t BLOCK = 16u << 20;

    RefStreamer(const std::string &path, int n_threads, Hooks hooks) : path_(path), n_threads_(n_threads < 1 ? 1 : n_threads), hooks_(std::move(hooks)) {
        fd_ = open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("Error opening compressed file: " + path);  // get_reader's message (src/main.rs:62)
        struct stat st;
        fstat(fd_, &st);
        size_ = (uint64_t)st.st_size;
    }
    ~RefStreamer() {
        for (void *c : all_chunks_) hooks_.release(c);
        if (fd_ >= 0) close(fd_);
    }
    RefStreamer(const RefStreamer &) = delete;
    RefStreamer &operator=(const RefStreamer &) = delete;
    uint64_t file_bytes() const { return size_; }

    // fn(record number, id, offset of the sequence in the file = in the staging buffer, length), in file order, from a thread of its own
    template <class F>
    Result run(F fn) {
        Result res;
        const size_t n_blocks = (size_t)((size_ + BLOCK - 1) / BLOCK);
        const size_t pool_size = (size_t)n_threads_ + 4;
        std::mutex mu;
        std::condition_variable cv;
        std::vector<void *> pool;           // free chunks
        size_t allocated = 0;               // chunks made so far (each reader makes its own first one: the pinning runs in parallel)
        std::deque<Scan> scanned;           // blocks read and scanned, any order
        std::atomic<size_t> next_block{0};
        bool stop = false;
        std::string err;
        auto reader = [&]() {
            try {
                for (;;) {
                    const size_t b = next_block.fetch_add(1);
                    if (b >= n_blocks) return;
                    void *chunk = nullptr;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || !pool.empty() || allocated < pool_size; });
                        if (stop) return;
                        if (!pool.empty()) {
                            chunk = pool.back();
                            pool.pop_back();
                        } else {
                            ++allocated;
                        }
                    }
                    if (!chunk) {
                        chunk = hooks_.alloc((size_t)BLOCK);
                        if (!chunk) throw std::runtime_error("cannot allocate a page-locked block for the reference");
                        std::lock_guard<std::mutex> lk(mu);
                        all_chunks_.push_back(chunk);
                    }
                    Scan sc;
                    sc.block = b;
                    sc.chunk = (uint8_t *)chunk;
                    const uint64_t lo = (uint64_t)b * BLOCK, hi = std::min<uint64_t>(lo + BLOCK, size_);
                    sc.n = hi - lo;
                    uint64_t got = 0;
                    while (got < sc.n) {
                        const ssize_t r = pread(fd_, sc.chunk + got, sc.n - got, (off_t)(lo + got));
                        if (r <= 0) throw std::runtime_error("read error: " + path_);
                        got += (uint64_t)r;
                    }
                    scan_block(sc, lo);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        scanned.push_back(std::move(sc));
                    }
                    cv.notify_all();
                }
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> lk(mu);
                if (err.empty()) err = e.what();
                stop = true;
                cv.notify_all();
            }
        };
        // the indexer: records in file order, as soon as the calling thread has queued their last block's copy
        struct Rec {
            std::string id;
            uint64_t at, len;
        };
        std::deque<Rec> to_index;
        bool no_more_records = false;
        std::thread indexer([&]() {
            size_t k = 0;
            for (;;) {
                Rec r;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !to_index.empty() || no_more_records; });
                    if (to_index.empty()) return;
                    r = std::move(to_index.front());
                    to_index.pop_front();
                }
                try {
                    fn(k, r.id, r.at, r.len);
                    ++k;
                    std::lock_guard<std::mutex> lk(mu);
                    res.handed = k;
                } catch (const std::exception &e) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (err.empty()) err = e.what();
                    stop = true;
                    to_index.clear();
                    no_more_records = true;
                    cv.notify_all();
                    return;
                }
            }
        });
        std::vector<std::thread> readers;
        for (int t = 0; t < n_threads_; ++t) readers.emplace_back(reader);

        std::deque<std::pair<uint64_t, void *>> inflight;  // (ticket, chunk) in issue order
        std::map<size_t, Scan> held;                       // scanned blocks waiting for their turn in the line bookkeeping
        size_t next_in_order = 0, issued = 0;
        // line bookkeeping over the whole file: the start of the current line, and for a record in the making its header line
        uint64_t line_start = 0;
        bool have_header = false;
        uint64_t hdr_start = 0, hdr_end = 0;  // [hdr_start, hdr_end): the header line without its line end
        int cur_first = -1;                   // first byte of the line being read; -1: not seen yet (it opens the next block)
        uint8_t prev_last = 0;                // the last byte of the block consumed before
        auto give_back = [&](void *chunk) {
            {
                std::lock_guard<std::mutex> lk(mu);
                pool.push_back(chunk);
            }
            cv.notify_all();
        };
        auto recycle = [&](bool must) {  // chunks whose copy is done go back to the pool; must: wait for the oldest one
            while (!inflight.empty()) {
                if (!hooks_.done(inflight.front().first, must)) break;
                give_back(inflight.front().second);
                inflight.pop_front();
                must = false;
            }
        };
        // a line [line_start, nl) ends (at its '\n', or at the file's end); cr: a '\r' in front of the line end.  Lines alternate: header
        // ('>' first), sequence (anything else first, or empty).  false: the file is not of that shape.
        auto end_line = [&](uint64_t nl, bool cr) -> bool {
            const uint64_t ls = line_start, le = (cr && nl > ls) ? nl - 1 : nl;
            const bool starts_gt = ls < nl && cur_first == '>';
            line_start = nl + 1;
            if (!have_header) {
                if (ls == le) return true;     // a blank line where a header may start (before the first record, between records, at the end): skipped, as seq_io does
                if (!starts_gt) return false;  // text before the first '>', a sequence that goes on over several lines
                have_header = true;
                hdr_start = ls;
                hdr_end = le;
                return true;
            }
            if (starts_gt) return false;  // a header without its sequence line
            have_header = false;
            Rec r;
            r.at = ls;
            r.len = le - ls;
            if (r.len >= (1ull << 32)) throw std::runtime_error("sequence length must be < 2^32");
            // seq_io's id(): the header up to its first space -- the only bytes of the file the host reads
            const uint64_t hl = hdr_end - hdr_start;
            std::string h((size_t)hl, '\0');
            uint64_t got = 0;
            while (got < hl) {
                const ssize_t q = pread(fd_, &h[(size_t)got], (size_t)(hl - got), (off_t)(hdr_start + got));
                if (q <= 0) throw std::runtime_error("read error: " + path_);
                got += (uint64_t)q;
            }
            size_t ie = 1;
            while (ie < h.size() && h[ie] != ' ') ++ie;
            r.id = h.substr(1, ie - 1);
            {
                std::lock_guard<std::mutex> lk(mu);
                to_index.push_back(std::move(r));
            }
            ++res.records;
            cv.notify_all();
            return true;
        };
        try {
            while (issued < n_blocks && !res.irregular) {
                Scan sc;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    if (scanned.empty() && !stop) {
                        lk.unlock();
                        recycle(false);
                        lk.lock();
                        // nothing scanned yet: with copies in flight wait for the oldest (a reader may be waiting for its chunk), else for a scan
                        if (scanned.empty() && !stop) {
                            if (!inflight.empty()) {
                                lk.unlock();
                                recycle(true);
                                continue;
                            }
                            cv.wait(lk, [&] { return !scanned.empty() || stop; });
                        }
                    }
                    if (stop) break;
                    if (scanned.empty()) continue;
                    sc = std::move(scanned.front());
                    scanned.pop_front();
                }
                inflight.emplace_back(hooks_.piece((uint64_t)sc.block * BLOCK, sc.chunk, sc.n), sc.chunk);
                ++issued;
                const size_t blk = sc.block;
                sc.chunk = nullptr;  // (the bytes are the link's now; the bookkeeping below uses what the scan noted)
                held.emplace(blk, std::move(sc));
                for (auto it = held.find(next_in_order); it != held.end() && !res.irregular; it = held.find(next_in_order)) {
                    const Scan &s = it->second;
                    const uint64_t lo = (uint64_t)s.block * BLOCK;
                    if (s.too_many) res.irregular = true;  // a block full of line ends: a line-wrapped FASTA (or very many tiny records): the host parser's case
                    if (cur_first < 0 && s.n) cur_first = s.first_byte;  // the line that opens this block
                    for (size_t k = 0; k < s.nl.size() && !res.irregular; ++k) {
                        const NL &e = s.nl[k];
                        const bool cr = e.pos == lo ? (lo > 0 && prev_last == '\r') : e.cr != 0;
                        if (!end_line(e.pos, cr)) res.irregular = true;
                        cur_first = e.next_known ? (int)e.next : -1;
                    }
                    if (s.n) prev_last = s.last_byte;
                    held.erase(it);
                    ++next_in_order;
                }
            }
            if (!res.irregular && !stop) {
                if (line_start < size_ && !end_line(size_, false)) res.irregular = true;  // the file's last line has no '\n'
                if (have_header) res.irregular = true;                                    // a header without its sequence line
                if (res.records == 0) res.irregular = true;                               // (an empty file: the host parser says what it is)
            }
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(mu);
            if (err.empty()) err = e.what();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            no_more_records = true;
            if (res.irregular) to_index.clear();
        }
        cv.notify_all();
        for (auto &t : readers) t.join();
        indexer.join();
        try {
            while (!inflight.empty()) recycle(true);
        } catch (const std::exception &e) {
            if (err.empty()) err = e.what();
        }
        if (!err.empty()) throw std::runtime_error(err);
        return res;
    }

  private:
    struct NL {
        uint64_t pos;         // file offset of a '\n'
        uint8_t cr;           // a '\r' in front of it (inside the block)
        uint8_t next_known;   // the byte behind it lies in this block ...
        uint8_t next;         // ... and is this one: the first byte of the next line
    };
    struct Scan {
        size_t block = 0;
        uint8_t *chunk = nullptr;
        uint64_t n = 0;
        std::vector<NL> nl;
        bool too_many = false;
        uint8_t last_byte = 0, first_byte = 0;
    };
    static constexpr size_t MAX_LINES_PER_BLOCK = 65536;  // single-line records of >= 512 bytes on average; beyond: the host parser's case

    void scan_block(Scan &sc, uint64_t lo) const {
        const uint8_t *p = sc.chunk, *end = sc.chunk + sc.n;
        if (sc.n) {
            sc.first_byte = sc.chunk[0];
            sc.last_byte = sc.chunk[sc.n - 1];
        }
        while (p < end) {
            const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
            if (!q) break;
            if (sc.nl.size() >= MAX_LINES_PER_BLOCK) {
                sc.too_many = true;
                return;
            }
            NL e;
            e.pos = lo + (uint64_t)(q - sc.chunk);
            e.cr = (q > sc.chunk && q[-1] == '\r') ? 1 : 0;
            e.next_known = q + 1 < end ? 1 : 0;
            e.next = q + 1 < end ? q[1] : 0;
            sc.nl.push_back(e);
            p = q + 1;
        }
    }

    std::string path_;
    int n_threads_;
    Hooks hooks_;
    int fd_ = -1;
    uint64_t size_ = 0;
    std::vector<void *> all_chunks_;
};

}  // namespace feeder
}  // namespace mapquik
