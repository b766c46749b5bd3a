// mapquik_host.hpp -- C++ host-side mirror of the reference's interface for the hot path, on top of the C ABI
// (include/mapquik_hip.h).  The reference is Rust; there is no Rust toolchain in this image, so the layer a Rust
// maintainer would write over the extern "C" block is written in C++ with the same names and argument
// meaning:   Params (src/main.rs:33-47) . Index / ReadOnlyIndex (src/index.rs:73-128) . mers::ref_extract (src/mers.rs:15)
//            . mers::find_matches (src/mers.rs:77) -> std::optional<std::string> for Option<String>.
// Errors: the reference panics; this layer throws mapquik::Error (never across the C ABI).  No CPU compute path exists.
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mapquik_hip.h"

namespace mapquik {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// src/main.rs:33-47, defaults src/main.rs:174-188
struct Params {
    size_t k = 5;
    size_t l = 31;
    double density = 0.01;
    bool use_hpc = true;
    bool use_simd = true;   // a speed switch in the reference (src/main.rs:150-155); no effect here
    bool use_pfx = false;
    bool debug = false;
    bool a = false;
    bool fold_case = false;  // extension: the kernels treat a-z as A-Z (the reference upper-cases on the host, src/closures.rs:63,106)
    unsigned seeding_variant = 0;  // extension: MQ_SEEDVAR_* bits (include/mapquik_hip.h); 0 = the frozen reading of rust-seq2kminmers
    bool fast_kh = false;    // extension: MQ_FLAG_FAST_KH, the cheap tuple hash (same PAF; KminmerHash.hash is then not the reference's value)
    size_t c = 4;
    size_t s = 11;
    size_t g = 2000;
    size_t b = 1;
    size_t q = 200;
    mq_params to_abi() const {
        mq_params p;
        p.k = (uint32_t)k;
        p.l = (uint32_t)l;
        p.density = density;
        p.use_hpc = use_hpc ? 1u : 0u;
        p.c = (uint32_t)c;
        p.s = (uint32_t)s;
        p.g = (uint32_t)g;
        p.flags = (fold_case ? MQ_FLAG_FOLD_CASE : 0u) | (fast_kh ? MQ_FLAG_FAST_KH : 0u) | MQ_FLAG_SEED_VARIANT(seeding_variant);
        return p;
    }
};

inline std::string last_error() { return mq_last_error() ? mq_last_error() : ""; }

class ReadOnlyIndex;

// Index (src/index.rs:73-105) + the ref_map of src/closures.rs:30
class Index {
  public:
    explicit Index(const Params &params, int device = 0) : params_(params) {
        const mq_params p = params.to_abi();
        h_ = mq_index_new(&p, device);
        if (!h_) throw Error("Index::new: " + last_error());
    }
    Index(const Index &) = delete;
    Index &operator=(const Index &) = delete;
    Index(Index &&o) noexcept : h_(o.h_), params_(o.params_) { o.h_ = nullptr; }
    ~Index() { mq_index_free(h_); }
    mq_index *handle() const { return h_; }
    const Params &params() const { return params_; }
    // table slots per inserted k-min-mer (mq_index_set_table_factor): 8 by default, 2 for host-bound file-fed runs
    void table_factor(uint32_t slots_per_kminmer) {
        if (mq_index_set_table_factor(h_, slots_per_kminmer) != MQ_OK) throw Error("Index::table_factor: " + last_error());
    }
    // DashMap::with_capacity (src/index.rs:83): the table for about n k-min-mers is allocated and cleared while the references load
    void with_capacity(uint64_t n_kminmers) {
        if (mq_index_reserve(h_, n_kminmers) != MQ_OK) throw Error("Index::with_capacity: " + last_error());
    }
    // get_count (src/index.rs:90-92) is reported by into_read_only()
    ReadOnlyIndex into_read_only() &&;

  private:
    friend class ReadOnlyIndex;
    mq_index *h_ = nullptr;
    Params params_;
};

// ReadOnlyIndex (src/index.rs:108-128): finalized, resident in HBM
class ReadOnlyIndex {
  public:
    ReadOnlyIndex(const ReadOnlyIndex &) = delete;
    ReadOnlyIndex(ReadOnlyIndex &&o) noexcept : h_(o.h_), unique_(o.unique_) { o.h_ = nullptr; }
    ~ReadOnlyIndex() { mq_index_free(h_); }
    mq_index *handle() const { return h_; }
    uint64_t unique_count() const { return unique_; }
    static ReadOnlyIndex load(const std::string &path, int device = 0) {
        mq_index *h = mq_index_load(path.c_str(), device);
        if (!h) throw Error("ReadOnlyIndex::load: " + last_error());
        mq_index_stats st;
        mq_index_get_stats(h, &st);
        return ReadOnlyIndex(h, st.n_unique);
    }
    // a replica on another device (device-to-device copy of the table)
    ReadOnlyIndex clone_to(int device) const {
        mq_index *h = mq_index_clone(h_, device);
        if (!h) throw Error("ReadOnlyIndex::clone_to: " + last_error());
        return ReadOnlyIndex(h, unique_);
    }
    void save(const std::string &path) const {
        if (mq_index_save(h_, path.c_str()) != MQ_OK) throw Error("ReadOnlyIndex::save: " + last_error());
    }

  private:
    friend class Index;
    ReadOnlyIndex(mq_index *h, uint64_t u) : h_(h), unique_(u) {}
    mq_index *h_ = nullptr;
    uint64_t unique_ = 0;
};

inline ReadOnlyIndex Index::into_read_only() && {
    const int64_t u = mq_index_finalize(h_);
    if (u < 0) throw Error("Index::into_read_only: " + last_error());
    mq_index *h = h_;
    h_ = nullptr;
    return ReadOnlyIndex(h, (uint64_t)u);
}

// The format! of src/mers.rs:181 for many reads: the same bytes as mq_format_paf, appended to a string, with the reference names and
// lengths looked up once per reference (a formatter thread of the native driver writes ~200,000 lines per batch; snprintf and a map
// lookup per line were 3-4 us of it).
class PafWriter {
  public:
    explicit PafWriter(const ReadOnlyIndex &index) : idx_(index.handle()) {}
    // appends "q_id\tq_len\t...\tmapq\n" for a mapped hit
    void append(std::string &out, const char *q_id, size_t q_id_len, uint64_t q_len, const mq_hit &h) {
        const Ref &r = ref(h.ref_id);
        char num[8 * 21 + 16];
        char *p = num;
        auto u = [&](uint64_t x) {
            char t[20];
            int n = 0;
            do { t[n++] = (char)('0' + x % 10); x /= 10; } while (x);
            while (n) *p++ = t[--n];
        };
        out.append(q_id, q_id_len);
        *p++ = '\t'; u(q_len);
        *p++ = '\t'; u(((uint64_t)h.q_start_hi << 32) | h.q_start);
        *p++ = '\t'; u(((uint64_t)h.q_end_hi << 32) | h.q_end);
        *p++ = '\t'; *p++ = h.rc ? '-' : '+'; *p++ = '\t';
        out.append(num, (size_t)(p - num));
        out += r.name;
        p = num;
        *p++ = '\t'; u(r.len);
        *p++ = '\t'; u(h.r_start);
        *p++ = '\t'; u(h.r_end);
        *p++ = '\t'; u(h.score);
        *p++ = '\t'; u(r.len);
        *p++ = '\t'; u(h.mapq);
        *p++ = '\n';
        out.append(num, (size_t)(p - num));
    }

  private:
    struct Ref {
        std::string name;
        uint64_t len = 0;
        bool known = false;
    };
    const Ref &ref(uint32_t id) {
        if (id >= refs_.size()) refs_.resize((size_t)id + 1);
        Ref &r = refs_[id];
        if (!r.known) {
            const char *name = nullptr;
            uint64_t len = 0;
            if (mq_index_ref_info(idx_, id, &name, &len) != MQ_OK) throw Error("find_coords: " + last_error());
            r.name = name ? name : "";
            r.len = len;
            r.known = true;
        }
        return r;
    }
    mq_index *idx_;
    std::vector<Ref> refs_;
};

namespace mers {

// mers::ref_extract (src/mers.rs:15-38) + ref_map.insert (src/closures.rs:49).  Returns the reference's k-min-mer count.
inline size_t ref_extract(size_t ref_idx, const std::string &ref_id, const uint8_t *inp_seq_raw, size_t len, const Params &, Index &mers_index) {
    const int64_t n = mq_index_add_ref(mers_index.handle(), (uint32_t)ref_idx, ref_id.c_str(), inp_seq_raw, (uint64_t)len);
    if (n < 0) throw Error("ref_extract: " + last_error());
    return (size_t)n;
}

// Batch form of find_matches: one Option<String> per read, in input order.
inline std::vector<std::optional<std::string>> find_matches_batch(const std::vector<std::string> &q_ids, const uint8_t *bases,
                                                                  const std::vector<uint64_t> &offsets, const ReadOnlyIndex &mers_index,
                                                                  const Params &) {
    const uint32_t n = (uint32_t)q_ids.size();
    std::vector<std::optional<std::string>> out(n);
    if (!n) return out;
    std::vector<mq_hit> hits(n);
    if (mq_map_batch(mers_index.handle(), bases, offsets.data(), n, hits.data()) != MQ_OK) throw Error("find_matches: " + last_error());
    std::vector<char> buf(4096);
    for (uint32_t i = 0; i < n; ++i) {
        if (hits[i].status == MQ_HIT_MAPPED) {
            int w = mq_format_paf(mers_index.handle(), q_ids[i].c_str(), offsets[i + 1] - offsets[i], &hits[i], buf.data(), buf.size());
            if (w >= (int)buf.size()) {  // long read / contig names: the return value is the full length, format again
                buf.resize((size_t)w + 1);
                w = mq_format_paf(mers_index.handle(), q_ids[i].c_str(), offsets[i + 1] - offsets[i], &hits[i], buf.data(), buf.size());
            }
            if (w < 0) throw Error("find_coords: " + last_error());
            out[i] = std::string(buf.data(), (size_t)w);
        } else if (hits[i].status != MQ_HIT_UNMAPPED) {
            throw Error("find_matches: read " + q_ids[i] + " could not be processed");
        }
    }
    return out;
}

// mers::find_matches (src/mers.rs:77-102): the PAF line, or nothing when the reference returns None.
inline std::optional<std::string> find_matches(const std::string &q_id, size_t q_len, const uint8_t *q_str, const ReadOnlyIndex &mers_index,
                                               const Params &params) {
    return find_matches_batch({q_id}, q_str, {0, (uint64_t)q_len}, mers_index, params)[0];
}

}  // namespace mers
}  // namespace mapquik
