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

    RefLoader(const std::string &path, int n_threads) : path_(path), n_threads_(n_threads < 1 ? 1 : n_threads) {
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
    }
    ~RefLoader() {
        stop_pool();
        if (buf_) munmap(buf_, mapped_);
        if (fd_ >= 0) close(fd_);
    }
    RefLoader(const RefLoader &) = delete;
    RefLoader &operator=(const RefLoader &) = delete;

    // fn(const Record &, const uint8_t *sequence) for every record, in file order.  The sequence stays valid until the loader dies.
    template <class F>
    void for_each(F fn) {
        read_all();
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
    std::vector<uint64_t> starts_;
    std::vector<Record> recs_;
    std::vector<char> ready_;
    std::atomic<size_t> next_rec_{0};
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::string error_;
};

}  // namespace feeder
}  // namespace mapquik
