// mapquik (HIP backend) -- command-line driver with the reference's surface: src/main.rs:77-272 (flags, defaults, log
// lines) and src/closures.rs:22-212 (index the reference, map the reads, write <prefix>.paf in input order).
// The hot path runs on the GPU through mapquik_host.hpp / the C ABI.  Reads come through fastx_feeder.hpp (parallel chunk
// reader for raw files, one inflate thread + parser threads for .gz / .lz4), go to the GPU as raw FASTX bytes + spans
// (mq_ctx_submit_spans, three stream slots per GPU so that copy-in, kernels and copy-out of consecutive chunks overlap),
// and the PAF is formatted by a small thread pool and written in input order.  The reference FASTA comes through the same
// feeder (its records go to mq_index_add_ref straight from the page-locked chunks).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/resource.h>
#include <sys/stat.h>

#include "fastx_feeder.hpp"
#include "mapquik_host.hpp"
#include "ref_loader.hpp"

using namespace mapquik;
using Clock = std::chrono::steady_clock;

static double secs(Clock::time_point a) { return std::chrono::duration<double>(Clock::now() - a).count(); }

// MQ_DRIVER_TIMING=1 (diagnostic, stderr): where the wall time of a whole job goes -- seconds since main() started at every step of the
// reference phase, the map phase and the teardown (on a 0.1-s map phase the fixed costs around it are most of the job)
static const Clock::time_point g_t_main = Clock::now();
static const bool g_timeline = getenv("MQ_DRIVER_TIMING") != nullptr;
static void tl(const char *what) {
    if (g_timeline) fprintf(stderr, "[+%.3f s] %s\n", secs(g_t_main), what);
}

// `{:?}` of a std::time::Duration
static std::string rust_duration(double seconds) {
    unsigned long long ns = (unsigned long long)(seconds * 1e9 + 0.5);
    const char *unit[4] = {"s", "ms", "\xC2\xB5s", "ns"};
    const unsigned long long div[4] = {1000000000ull, 1000000ull, 1000ull, 1ull};
    for (int u = 0; u < 4; ++u) {
        if (ns >= div[u] || u == 3) {
            unsigned long long whole = ns / div[u], frac = ns % div[u];
            char buf[64];
            if (frac == 0 || div[u] == 1) {
                snprintf(buf, sizeof(buf), "%llu%s", whole, unit[u]);
                return buf;
            }
            int digits = u == 0 ? 9 : u == 1 ? 6 : 3;
            char fr[16];
            snprintf(fr, sizeof(fr), "%0*llu", digits, frac);
            std::string f(fr);
            while (!f.empty() && f.back() == '0') f.pop_back();
            snprintf(buf, sizeof(buf), "%llu.%s%s", whole, f.c_str(), unit[u]);
            return buf;
        }
    }
    return "0ns";
}

// `{}` of an f64 for the values that occur here
static std::string rust_float(double x) {
    char buf[64];
    if (x == (double)(long long)x && std::abs(x) < 1e15) {
        snprintf(buf, sizeof(buf), "%lld", (long long)x);
        return buf;
    }
    for (int prec = 1; prec < 18; ++prec) {
        snprintf(buf, sizeof(buf), "%.*g", prec, x);
        if (strtod(buf, nullptr) == x) break;
    }
    return buf;
}

static bool contains(const std::string &s, const char *t) { return s.find(t) != std::string::npos; }
static bool ends_with(const std::string &s, const char *t) {
    const size_t n = strlen(t);
    return s.size() >= n && s.compare(s.size() - n, n, t) == 0;
}
// src/main.rs:196,202
static bool is_fasta_name(const std::string &n) {
    return contains(n, ".fasta.") || ends_with(n, ".fna") || contains(n, ".fna.") || contains(n, ".fa.") || ends_with(n, ".fa") ||
           ends_with(n, ".fasta");
}

struct Opt {
    std::string reads, reference, prefix;
    bool has_prefix = false, debug = false, low_memory = false, nosimd = false, nohpc = false, parallelfastx = false, unmapped = false;
    long k = -1, l = -1, c = -1, s = -1, g = -1, threads = -1, b = -1, q = -1;
    double density = -1;
    int device = 0;
    int gpus = 1;
    int seeding_variant = 0;  // MQ_SEEDVAR_* bits (include/mapquik_hip.h)
    bool fast_kh = false;     // MQ_FLAG_FAST_KH
    bool last_pass = true;  // no second pass follows this one
    int table_factor = 2;  // table slots per inserted k-min-mer: this driver is bound by its host side (mq_index_set_table_factor)
    std::string save_index, load_index;  // --save-index / --index: the on-disk index (the reference has none and re-indexes on every run)
    std::string second;  // "k2,l2,d2"
    long k2 = 0, l2 = 0;
    double d2 = 0;
    unsigned long long batch_bases = 0;  // raw input bytes per chunk (page-locked buffers this size); 0 = 32 MB, and 64 MB for a reads file of 2 GB or
                                         // more read by at most four threads (28.5 -> 30.8 Gbases/s on the 4.6-GB FASTA of the bench on one box, no
                                         // change on another; with eight or more readers the rate does not move and the job's wall time grows by
                                         // 0.06-0.13 s: twice the page-locked memory to set up; tools/chunk_probe.py, profiles/NOTES.md)
};

static void usage() {
    puts("mapquik 0.1.0 (HIP backend)\nOriginal implementation of mapquik, a fast HiFi read mapper.\n\n"
         "USAGE:\n    mapquik [FLAGS] [OPTIONS] [reads]\n\nFLAGS:\n        --debug\n        --low-memory\n        --nohpc\n        --nosimd\n"
         "        --parallelfastx\n        --unmapped      (extension) also write <prefix>.unmapped.out\n\nOPTIONS:\n"
         "    -b <b>\n    -c, --chain <chain>\n    -d, --density <density>\n    -g, --gap-diff <gap-diff>\n    -k <k>\n    -l <l>\n"
         "    -p, --prefix <prefix>\n    -q <q>\n        --reference <reference>\n    -s, --seed <seed>\n        --threads <threads>\n"
         "        --device <n>    (extension) first HIP device ordinal\n        --gpus <n>      (extension) shard read batches over n GPUs, index replicated\n        --batch-bases <n> (extension) raw input bytes per chunk\n        --table-factor <n> (extension) index table slots per k-min-mer (default 2 here: a file-fed run is host-bound; the library's default for HBM-resident batches is 8)\n        --save-index <file> (extension) write the finalized index (occupied slots only) for later runs\n        --index <file>  (extension) map against a saved index instead of indexing --reference (same -k -l -d --nohpc as it was built with)\n        --seeding-variant <v> (extension) reading of the k-min-mer iterator's unpinned decisions, bits 1 2 4 8 16 32 (include/mapquik_hip.h); 0 = frozen\n        --fast-kh       (extension) cheap k-min-mer tuple hash instead of SipHash-1-3: same PAF (the hash acts through equality only), fewer instructions\n        --second-pass <k2,l2,d2> (extension) map the unmapped reads again with these parameters: <prefix>-k2-l2-d2.{fa,paf,unmapped.out}\n\nARGS:\n    <reads>");
}

// the last two lines of a run (src/main.rs:270-271)
static void print_totals(Clock::time_point start) {
    printf("Total execution time: %s\n", rust_duration(secs(start)).c_str());  // src/main.rs:270
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    const float gb = (float)((double)ru.ru_maxrss * 1024.0) / 1024.0f / 1024.0f / 1024.0f;
    char fb[64];
    for (int prec = 1; prec < 12; ++prec) {
        snprintf(fb, sizeof(fb), "%.*g", prec, (double)gb);
        if ((float)strtod(fb, nullptr) == gb) break;
    }
    printf("Maximum RSS: %sGB\n", strchr(fb, '.') || strchr(fb, 'e') ? fb : (std::string(fb) + ".0").c_str());  // src/main.rs:271
}

// One run of the reference's flow (src/closures.rs:22-212): index the reference, map the reads, write <prefix>.paf in input
// order.  second_fa != "": the reads left unmapped also go to that FASTA file (for the second pass).
static int run_pass(const Opt &o, const Params &P, const std::string &reads_path, bool reads_fasta, bool ref_fasta, const std::string &prefix,
                    size_t threads, const std::string &second_fa) {
        FILE *paf = fopen((prefix + ".paf").c_str(), "w");  // src/closures.rs:32
        if (!paf) { fprintf(stderr, "Couldn't create %s.paf\n", prefix.c_str()); return 101; }
        FILE *unm = (o.unmapped || !o.second.empty()) ? fopen((prefix + ".unmapped.out").c_str(), "w") : nullptr;
        FILE *ufa = second_fa.empty() ? nullptr : fopen(second_fa.c_str(), "w");  // the unmapped reads as FASTA (seqtk subseq in the reference's script)

        // An uncompressed reference FASTA goes to the device through a small pool of page-locked blocks as it is read (RefStreamer, below);
        // a file that is not one sequence line per record is read into host memory whole, its records joined there (RefLoader).
        // MQ_DRIVER_REF_PRELOAD=1 (experiment): the whole-file read starts HERE, before the first HIP call -- bringing the HIP runtime up
        // takes 0.15-0.3 s of one thread, reading 3.1 GB 0.08-0.1 s of the others -- and the records go to the device from that buffer,
        // page-locked in one call.  Measured slower than streaming on the bench's job (profiles/r05_driver_medians.txt): 3 GB of host
        // memory cost 0.14 s to hand back on this platform (pages are cleared when freed: tools/thp_probe.c, 45 ms per GB), whoever does it.
        const int n_parse = (int)std::max<size_t>(1, threads);
        const bool ref_plain = ref_fasta && !ends_with(o.reference, ".gz") && !ends_with(o.reference, ".lz4");
        const bool ref_host = getenv("MQ_DRIVER_REF_HOST") != nullptr;  // diagnostic: earlier rounds' path (records copied from pageable memory one by one)
        const bool ref_preload = getenv("MQ_DRIVER_REF_PRELOAD") != nullptr && !o.low_memory;
        std::unique_ptr<feeder::RefLoader> preload;
        if (ref_plain && o.load_index.empty() && (ref_preload || ref_host)) preload.reset(new feeder::RefLoader(o.reference, n_parse, !ref_host));

        // --gpus N: the index is replicated (every GPU indexes the same reference), read batches are dealt round-robin,
        // PAF lines are written in batch order = input order.  No collective: reads are independent.
        const int n_dev = mq_device_count();
        tl("HIP runtime up (first HIP call returned)");
        const bool fake = getenv("MQ_FAKE_MULTI") != nullptr;  // test hook: several workers on one device
        if (!fake && o.device + o.gpus > n_dev && n_dev > 0) {
            fprintf(stderr, "mapquik: --gpus %d from device %d needs %d devices, %d visible\n", o.gpus, o.device, o.device + o.gpus, n_dev);
            return 101;
        }
        auto dev_of = [&](int g) { return fake && n_dev > 0 ? (o.device + g) % n_dev : o.device + g; };

        // The read feeder (constructed here, started below): parsing reads does not depend on the index.
        using feeder::Chunk;
        const int n_slots = 3;  // stream slots per GPU: copy-in, kernels and copy-out of consecutive chunks overlap
        const int n_format = std::max(2, std::min(8, n_parse));  // PAF formatters (the reader threads of a mapped FASTA file have next to nothing to do)
        unsigned long long batch_bases = o.batch_bases;
        if (batch_bases == 0) {
            struct stat sb;
            batch_bases = (n_parse <= 4 && stat(reads_path.c_str(), &sb) == 0 && (unsigned long long)sb.st_size >= (2ull << 30)) ? (1ull << 26) : (1ull << 25);
        }
        feeder::Feeder feed(reads_path, !reads_fasta, batch_bases, n_parse, n_parse + o.gpus * (n_slots + 1) + n_format + 2);
        // An uncompressed FASTA file goes to the GPU as it lies in the file: the reader threads only copy file bytes into page-locked
        // chunks (pread, cut at record starts), the records are found on the device (mq_ctx_submit_fasta) and the host reads a header
        // only to print it.  MQ_DRIVER_HOST_PARSE=1: every chunk is parsed by the reader threads as in earlier rounds (same PAF; tests compare).
        // FASTQ: the lean reader by default -- header and sequence lines only, one pread per record, qualities never read: 1 byte per base
        // from the file and on the link: 12 / 20 / 31 / 35 Gbases/s at 2 / 4 / 8 / 16 reader threads against 9 / 17 / 22 / 21 with the
        // records found on the device, where the whole file (2 bytes per base) is read and crosses the link
        // (profiles/r05_fastq_readers.txt).  MQ_DRIVER_FASTQ=device selects that path (mq_ctx_submit_fastx).
        bool on_device = getenv("MQ_DRIVER_HOST_PARSE") == nullptr;
        if (on_device && !reads_fasta) {
            const char *fq = getenv("MQ_DRIVER_FASTQ");
            on_device = fq && strcmp(fq, "device") == 0;
        }
        feed.leave_unparsed(on_device);  // (acts on uncompressed input only, FASTA or FASTQ)
        const uint32_t fx_format = reads_fasta ? MQ_FASTX_FASTA : MQ_FASTX_FASTQ, fx_lpr = reads_fasta ? 2u : 4u;
        feed.premap();  // MQ_FEEDER_MAPPED_FASTA=1 only (experiment): the file is mapped, not read, while the reference is indexed
        // The read feeder starts when the index is ready.  MQ_DRIVER_PREFETCH=1 starts it while the reference is still being indexed
        // (it then allocates its page-locked chunk buffers and parses the first chunks early): that was the default while pinning
        // the pool was the read phase's start-up cost; with the huge-page pool it makes the map phase 15 % shorter and the index
        // phase twice as long (the pool's hipHostRegister calls and the index calls share the driver) -- 1.47 s against 1.0 s for
        // the whole job on the bench's input.
        const bool prefetch = getenv("MQ_DRIVER_PREFETCH") != nullptr && getenv("MQ_DRIVER_NO_PREFETCH") == nullptr;
        bool feed_started = false;
        auto start_feed = [&]() {
            if (!feed_started) {
                feed.start();
                feed_started = true;
            }
        };

        auto t0 = Clock::now();
        // index_mers (src/closures.rs:46-51) per reference record, in file order, on the first GPU; the finalized table is then
        // copied device to device to the other GPUs (mq_index_clone).  The kernels fold soft-masked lower case.
        std::vector<std::unique_ptr<Index>> building(1);
        if (o.load_index.empty()) {
            building[0].reset(new Index(P, dev_of(0)));
            building[0]->table_factor((uint32_t)o.table_factor);
            tl("Index::new returned (HIP runtime up, device chosen)");
        }
        const bool stream_ref = ref_plain && o.load_index.empty() && !ref_preload && !ref_host;  // RefStreamer (below)
        // The stream slots of the map phase (device staging, minimizer lists, Match scratch: a few hundred MB of device memory per
        // submitting thread) and the feeder's first page-locked chunk buffers depend on neither the reference nor the reads: the first
        // GPU's are set up by a thread of its own BESIDE the reference phase (0.03-0.04 s of a 0.1-s phase when they came after it).
        const int n_sub = feed.mapped_views() ? 2 : 1;  // submitting threads per GPU (see below)
        std::vector<std::vector<mq_ctx *>> slots((size_t)(o.gpus * n_sub), std::vector<mq_ctx *>((size_t)n_slots, nullptr));
        auto make_slots = [&](size_t gw, mq_index *h) -> std::string {
            for (int sl = 0; sl < n_slots; ++sl) {
                if (slots[gw][sl]) continue;
                slots[gw][sl] = mq_ctx_new(h);
                if (!slots[gw][sl]) return std::string("mq_ctx_new: ") + last_error();
                const uint64_t cb = std::min<uint64_t>(batch_bases + batch_bases / 8 + (1u << 20), feed.bytes_in() + 64);
                if (mq_ctx_reserve(slots[gw][sl], (uint32_t)std::min<uint64_t>(cb / 16000 + 512, 1u << 24), cb) != MQ_OK)  // (sized for long reads; a chunk of short reads makes its slot grow once)
                    return std::string("mq_ctx_reserve: ") + last_error();
            }
            return std::string();
        };
        std::thread early;
        std::string early_err;
        bool pool_ready = false;
        auto early_join = [&]() {
            if (early.joinable()) early.join();
        };
        auto early_start = [&]() {
            if (getenv("MQ_DRIVER_LATE_SLOTS") != nullptr) return;  // (diagnostic: everything after the reference phase, as in earlier rounds)
            mq_index *h = building[0]->handle();
            early = std::thread([&, h]() {
                for (int w = 0; w < n_sub && early_err.empty(); ++w) early_err = make_slots((size_t)w, h);
                if (!prefetch && early_err.empty()) {
                    try {
                        feed.preallocate(n_parse + n_slots);
                        pool_ready = true;
                    } catch (const std::exception &e) { early_err = e.what(); }
                }
            });
        };
        struct EarlyGuard {  // an exception on the way: the thread is joined before its captures go away
            std::thread &t;
            ~EarlyGuard() { if (t.joinable()) t.join(); }
        } early_guard{early};
        auto reserve_table = [&]() {
            if (!ref_plain || getenv("MQ_DRIVER_NO_RESERVE") != nullptr) return;
            // Index::new sizes its map before the first insert (src/index.rs:83: with_capacity(39,821,990), CHM13 at the defaults); here the
            // expected count follows from the reference's size: canonical selection keeps 1 - (1 - d)^2 of the l-mers, homopolymer
            // compression about three quarters of the bases.  The table is allocated in the background while the reference is read and seeded.
            struct stat rst;
            if (stat(o.reference.c_str(), &rst) == 0 && rst.st_size > 0) {
                const double d = std::min(1.0, std::max(0.0, P.density));
                building[0]->with_capacity((uint64_t)((double)rst.st_size * (1.0 - (1.0 - d) * (1.0 - d)) * (P.use_hpc ? 0.75 : 1.0)) + 1);
            }
        };
        std::thread ref_reaper;
        struct ReaperGuard {
            std::thread &t;
            ~ReaperGuard() { if (t.joinable()) t.join(); }
        } reaper_guard{ref_reaper};
        bool ref_done = false;
        bool res_streamer_used = false;  // the streamer ran (and gave the file back): the index's staging buffer holds its pieces
        bool streamer_index_dropped = false;  // ... after records had been indexed: that index was replaced by a new one
        std::unique_ptr<ReadOnlyIndex> loaded;
        if (!o.load_index.empty()) {
            // --index: the finalized table from a file written by --save-index (occupied slots only; validated against its header on load)
            building.clear();
            loaded.reset(new ReadOnlyIndex(ReadOnlyIndex::load(o.load_index, dev_of(0))));
            mq_params fp;
            if (mq_index_get_params(loaded->handle(), &fp) != MQ_OK) throw Error("mq_index_get_params: " + last_error());
            const mq_params want = P.to_abi();
            if (fp.k != want.k || fp.l != want.l || fp.density != want.density || fp.use_hpc != want.use_hpc ||
                (fp.flags & (MQ_FLAG_SEED_VARIANT_MASK | MQ_FLAG_FAST_KH)) != (want.flags & (MQ_FLAG_SEED_VARIANT_MASK | MQ_FLAG_FAST_KH))) {
                char msg[640];
                snprintf(msg, sizeof(msg), "%s was built with -k %u -l %u -d %s%s --seeding-variant %u%s: run with the same seeding parameters (this run: -k %u -l %u -d %s%s --seeding-variant %u%s)",
                         o.load_index.c_str(), fp.k, fp.l, rust_float(fp.density).c_str(), fp.use_hpc ? "" : " --nohpc", (fp.flags & MQ_FLAG_SEED_VARIANT_MASK) >> MQ_FLAG_SEED_VARIANT_SHIFT,
                         (fp.flags & MQ_FLAG_FAST_KH) ? " --fast-kh" : "",
                         want.k, want.l, rust_float(want.density).c_str(), want.use_hpc ? "" : " --nohpc", (want.flags & MQ_FLAG_SEED_VARIANT_MASK) >> MQ_FLAG_SEED_VARIANT_SHIFT,
                         (want.flags & MQ_FLAG_FAST_KH) ? " --fast-kh" : "");
                throw Error(msg);
            }
            // the chaining thresholds and the case folding are this run's (they act at mapping time only)
            if (mq_index_set_map_params(loaded->handle(), want.c, want.s, want.g, (want.flags & MQ_FLAG_FOLD_CASE) ? 1 : 0) != MQ_OK) throw Error("mq_index_set_map_params: " + last_error());
            mq_index_stats st;
            mq_index_get_stats(loaded->handle(), &st);
            printf("Loaded index %s: %llu references, %llu k-min-mers.\n", o.load_index.c_str(), (unsigned long long)st.n_refs, (unsigned long long)st.n_kminmers);
            ref_done = true;
            tl("index file loaded");
        } else {
            if (stream_ref || (preload && !ref_host)) {
                // the device's staging buffer of the reference FIRST: device allocations queue behind each other, and this one (the file's
                // size) must not wait behind the table's, which is larger and not needed before the last record is indexed
                struct stat rst;
                if (stat(o.reference.c_str(), &rst) != 0) throw Error("Error opening compressed file: " + o.reference);  // get_reader's message (src/main.rs:62)
                if (mq_index_stage_begin(building[0]->handle(), (uint64_t)rst.st_size) != MQ_OK) throw Error("mq_index_stage_begin: " + last_error());
                tl("staging buffer for the reference allocated");
            }
            reserve_table();
            early_start();
        }
        if (!ref_done && stream_ref) {
            // an uncompressed FASTA of one sequence line per record (what assemblers and this repository's tools write): streamed to the
            // device block by block as it is read, records indexed while the blocks behind them are still on the link (RefStreamer);
            // the host reads the header lines only.  Any other shape: the loader below.
            feeder::RefStreamer::Hooks hooks;
            hooks.alloc = [](size_t n) { return mq_host_alloc(n); };
            hooks.release = [](void *q) { mq_host_free(q); };
            mq_index *h = building[0]->handle();
            hooks.piece = [h](uint64_t at, const uint8_t *src, uint64_t n) {
                uint64_t t = 0;
                if (mq_index_stage_piece(h, at, src, n, &t) != MQ_OK) throw Error("mq_index_stage_piece: " + last_error());
                return t;
            };
            hooks.done = [h](uint64_t t, bool wait) {
                const int r = mq_index_stage_done(h, t, wait ? 1 : 0);
                if (r < 0) throw Error("mq_index_stage_done: " + last_error());
                return r == 1;
            };
            feeder::RefStreamer rs(o.reference, n_parse, hooks);
            std::vector<std::string> lines;  // printed once the file's shape is known to be regular (else the loader below prints its own)
            res_streamer_used = true;
            const feeder::RefStreamer::Result res = rs.run([&](size_t k, const std::string &id, uint64_t at, uint64_t len) {
                const int64_t cnt = mq_index_add_ref_staged(h, (uint32_t)k, id.c_str(), at, len, MQ_STAGE_ALL_ISSUED);  // index_mers, src/closures.rs:46-51
                if (cnt < 0) throw Error("ref_extract: " + last_error());
                lines.push_back("Indexed reference " + id + ": " + std::to_string(cnt) + " k-min-mers.");  // src/closures.rs:58
            });
            if (!res.irregular) {
                for (const std::string &ln : lines) puts(ln.c_str());
                ref_done = true;
                tl("reference streamed: every record handed to ref_extract");
            } else {
                // not one sequence line per record (a line-wrapped FASTA shows in its first block, before anything was indexed): an index
                // that took records already is dropped, and the file goes through the loader below
                if (res.handed > 0) {  // (an index that has seen nothing stays: its table is being allocated in the background already)
                    early_join();      // the slots set up so far belong to the index that goes away
                    for (auto &v : slots) for (auto &c : v) { mq_ctx_free(c); c = nullptr; }
                    building[0].reset();
                    building[0].reset(new Index(P, dev_of(0)));
                    building[0]->table_factor((uint32_t)o.table_factor);
                    streamer_index_dropped = true;
                    reserve_table();
                }
                tl("reference is not one line per record: host loader");
            }
        }
        if (ref_done) {
        } else if (ref_plain && !(o.low_memory && res_streamer_used)) {  // (--low-memory: a file the streamer gave back goes through the chunked reader below, record by record)
            // an uncompressed FASTA: the whole file read once by all threads (since before the HIP runtime came up), multi-line records
            // compacted in place by a pool, records handed over whole and in order (ref_loader.hpp).  The buffer is page-locked in one
            // call (huge pages: milliseconds), every record's bytes are queued for the device the moment the record is ready
            // (mq_index_stage_piece: the link runs at its full rate, 3.1 GB in 0.06 s) and a second thread indexes record after record
            // behind its piece (mq_index_add_ref_staged) -- copied record by record from pageable memory this was 0.24 s.
            if (!preload) preload.reset(new feeder::RefLoader(o.reference, n_parse, false));  // (the streamer above sent the file here)
            feeder::RefLoader &rl = *preload;
            rl.wait_read();
            tl("reference file in host memory");
            mq_index *h = building[0]->handle();
            struct stat rst;
            bool staged = !ref_host && stat(o.reference.c_str(), &rst) == 0 && (uint64_t)rst.st_size == rl.file_bytes();
            if (staged && streamer_index_dropped) {  // the streamer's index went away with its staging buffer: the new index gets one
                if (mq_index_stage_begin(h, rl.file_bytes()) != MQ_OK) staged = false;
            }
            bool locked = false;
            if (staged) {
                locked = mq_host_register(rl.data(), (size_t)rl.mapped_bytes()) == MQ_OK;  // (not locked: the copies still work, at the pageable rate)
                tl(locked ? "reference buffer page-locked" : "reference buffer could not be page-locked: pageable copies");
            }
            struct Job {
                size_t idx;
                std::string id;
                uint64_t at, len, ticket;
            };
            std::mutex jmu;
            std::condition_variable jcv;
            std::deque<Job> jobs;
            bool jobs_done = false;
            std::string jerr;
            std::vector<std::string> lines;
            std::thread indexer;
            if (staged)
                indexer = std::thread([&]() {
                    for (;;) {
                        Job j;
                        {
                            std::unique_lock<std::mutex> lk(jmu);
                            jcv.wait(lk, [&] { return !jobs.empty() || jobs_done; });
                            if (jobs.empty()) return;
                            j = std::move(jobs.front());
                            jobs.pop_front();
                        }
                        const int64_t cnt = mq_index_add_ref_staged(h, (uint32_t)j.idx, j.id.c_str(), j.at, j.len, j.ticket);  // index_mers, src/closures.rs:46-51
                        if (cnt < 0) {
                            std::lock_guard<std::mutex> lk(jmu);
                            if (jerr.empty()) jerr = "ref_extract: " + last_error();
                            return;
                        }
                        printf("Indexed reference %s: %lld k-min-mers.\n", j.id.c_str(), (long long)cnt);  // src/closures.rs:58
                    }
                });
            size_t ref_idx = 0;
            std::string ferr;
            uint64_t last_ticket = 0;
            bool any_ticket = false;
            try {
                rl.for_each([&](const feeder::RefLoader::Record &r, const uint8_t *seq) {
                    if (ref_idx == 0) tl("first reference record ready");
                    if (prefetch) start_feed();  // the whole file has been read by now: the host threads are free
                    if (staged) {
                        uint64_t t = 0;
                        if (mq_index_stage_piece(h, r.seq, seq, r.len, &t) != MQ_OK) throw Error("mq_index_stage_piece: " + last_error());
                        last_ticket = t;
                        any_ticket = true;
                        {
                            std::lock_guard<std::mutex> lk(jmu);
                            if (!jerr.empty()) throw Error(jerr);
                            jobs.push_back(Job{ref_idx, r.id, r.seq, r.len, t});
                        }
                        jcv.notify_all();
                    } else {
                        const size_t cnt = mers::ref_extract(ref_idx, r.id, seq, r.len, P, *building[0]);
                        printf("Indexed reference %s: %zu k-min-mers.\n", r.id.c_str(), cnt);  // src/closures.rs:58
                    }
                    ++ref_idx;
                });
            } catch (const std::exception &e) { ferr = e.what(); }
            {
                std::lock_guard<std::mutex> lk(jmu);
                jobs_done = true;
                if (!ferr.empty()) jobs.clear();
            }
            jcv.notify_all();
            if (indexer.joinable()) indexer.join();
            // the copies read the buffer until the last piece is done (a record too short to be seeded is "indexed" without waiting for its piece)
            if (any_ticket) mq_index_stage_done(h, last_ticket, 1);
            if (locked) mq_host_unregister(rl.data());
            if (!ferr.empty()) throw Error(ferr);
            if (!jerr.empty()) throw Error(jerr);
            tl("every reference record indexed");
            // (the buffer goes back to the system at the end of the run, beside the rest of the teardown: handing 3 GB back takes 0.14 s
            // here and stalls whoever maps or allocates memory meanwhile -- index finalisation, stream-slot set-up)
        } else {
            // compressed (or FASTQ) reference: through the chunked feeder, pageable chunk buffers (every reference byte is copied to
            // the device exactly once)
            if (prefetch) start_feed();
            feeder::Feeder rfeed(o.reference, !ref_fasta, 1ull << 28, n_parse, n_parse + 4, [](size_t n) { return malloc(n); },
                                 [](void *q) { free(q); }, [](void *, size_t) { return 0; }, [](void *) { return 0; });
            rfeed.start();
            std::map<size_t, Chunk *> held;
            size_t next = 0, ref_idx = 0;
            std::string name;
            auto flush = [&]() {
                for (auto it = held.find(next); it != held.end(); it = held.find(next)) {
                    Chunk *c = it->second;
                    for (size_t i = 0; i < c->starts.size(); ++i) {
                        name.assign((const char *)c->buf + c->ids[i].off, c->ids[i].len);
                        const size_t cnt = mers::ref_extract(ref_idx, name, c->buf + c->starts[i], c->lens[i], P, *building[0]);
                        printf("Indexed reference %s: %zu k-min-mers.\n", name.c_str(), cnt);  // src/closures.rs:58
                        ++ref_idx;
                    }
                    held.erase(it);
                    rfeed.recycle(c);
                    ++next;
                }
            };
            while (Chunk *c = rfeed.next()) {
                held[c->seq_no] = c;
                flush();
            }
            flush();
        }
        std::vector<std::unique_ptr<ReadOnlyIndex>> ro((size_t)o.gpus);
        tl("every reference record handed to ref_extract");
        {
            if (loaded) {
                ro[0] = std::move(loaded);
            } else {
                ro[0].reset(new ReadOnlyIndex(std::move(*building[0]).into_read_only()));
                tl("into_read_only returned (table allocated, k-min-mers inserted)");
                building.clear();
            }
            if (!o.save_index.empty()) {
                const auto ts = Clock::now();
                ro[0]->save(o.save_index);
                printf("Saved index to %s in %s.\n", o.save_index.c_str(), rust_duration(secs(ts)).c_str());
            }
            std::vector<std::string> errs((size_t)o.gpus);
            std::vector<std::thread> th;
            for (int g = 1; g < o.gpus; ++g)
                th.emplace_back([&, g]() {
                    try {
                        ro[g].reset(new ReadOnlyIndex(ro[0]->clone_to(dev_of(g))));
                    } catch (const Error &e) { errs[g] = e.what(); }
                });
            for (auto &t : th) t.join();
            for (auto &e : errs) if (!e.empty()) throw Error(e);
        }
        // The stream slots of the other GPUs (the first GPU's were set up beside the reference phase) and whatever of the first GPU's
        // is still missing.
        early_join();
        if (!early_err.empty()) {
            for (auto &v : slots) for (auto c : v) mq_ctx_free(c);
            throw Error(early_err);
        }
        {
            std::vector<std::string> errs(slots.size());
            std::vector<std::thread> th;
            for (size_t gw = 0; gw < slots.size(); ++gw)
                if (!slots[gw][(size_t)n_slots - 1]) th.emplace_back([&, gw]() { errs[gw] = make_slots(gw, ro[gw / (size_t)n_sub]->handle()); });
            if (!prefetch && !pool_ready) th.emplace_back([&]() { feed.preallocate(n_parse + n_slots); });  // and the first page-locked chunk buffers
            for (auto &t : th) t.join();
            for (auto &e : errs)
                if (!e.empty()) {
                    for (auto &v : slots) for (auto c : v) mq_ctx_free(c);
                    throw Error(e);
                }
        }
        tl("replicas cloned, stream slots and first chunk buffers set up");
        printf("Indexed %llu unique k-min-mers in %s.\n", (unsigned long long)ro[0]->unique_count(), rust_duration(secs(t0)).c_str());

        t0 = Clock::now();
        if (P.use_pfx && !ends_with(reads_path, ".gz") && !ends_with(reads_path, ".lz4")) puts("Warning: using experimental rust-parallelfastx (exciting!)");
        start_feed();
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Chunk *> to_format;               // mapped, waiting for a formatter
        std::map<size_t, Chunk *> done;              // formatted, waiting for their turn in the output
        // Submitting threads per GPU (n_sub).  Chunks that are views of the mapped file are copied to the device from pageable memory: that
        // copy occupies the thread that asks for it, hence two (experimental path, MQ_FEEDER_MAPPED_FASTA=1; see Feeder::premap).
        int gpu_workers_left = o.gpus * n_sub;
        int formatting = 0;                          // chunks a formatter is working on right now
        std::string werr;
        auto fail = [&](const std::string &m) {
            {
                std::lock_guard<std::mutex> lk(mu);
                if (werr.empty()) werr = m;
            }
            cv.notify_all();
        };
        auto failed = [&]() {
            std::lock_guard<std::mutex> lk(mu);
            return !werr.empty();
        };
        const bool drv_timing = getenv("MQ_DRIVER_TIMING") != nullptr;  // diagnostic: where the map phase's threads spend their time (stderr)
        std::atomic<long long> t_submit_us{0}, t_finish_us{0}, t_fetch_us{0}, t_format_us{0}, t_write_us{0};
        auto us_since = [](Clock::time_point a) { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - a).count(); };
        const char *fail_at_env = getenv("MQ_DRIVER_FAIL_AT");  // test hook: the worker that takes this chunk number reports a failure
        const long fail_at = fail_at_env ? atol(fail_at_env) : -1;
        std::vector<std::thread> workers;
        for (int gw = 0; gw < o.gpus * n_sub; ++gw)
            workers.emplace_back([&, gw]() {
                std::vector<mq_ctx *> &ctx = slots[(size_t)gw];
                std::vector<Chunk *> inflight((size_t)n_slots, nullptr);
                std::vector<size_t> age((size_t)n_slots, 0);  // submit order of the chunk in the slot
                size_t submitted = 0;
                auto finish_slot = [&](int sl) {
                    if (!inflight[sl]) return;
                    const auto tf0 = Clock::now();
                    Chunk *fc = inflight[sl];
                    if (fc->unparsed) {  // records found on the device: hits and line ends come back together
                        uint32_t n = 0, n_lines = 0, flags = 0;
                        const uint32_t *line_ends = nullptr;
                        const mq_hit *hits = nullptr;
                        if (mq_ctx_wait_fasta(ctx[sl], &n, &line_ends, &n_lines, &hits, &flags) != MQ_OK) {
                            fail(std::string("mq_ctx_wait_fasta: ") + last_error());
                        } else if (flags & MQ_FASTA_IRREGULAR) {
                            // sequences over several lines, blank lines, ...: this chunk the old way (parsed here, spans to the device)
                            try {
                                fc->materialize();  // a view of the mapped file: the parser compacts sequence lines in place
                                feeder::parse_chunk(*fc, !reads_fasta);
                                fc->hits.resize(fc->starts.size());
                                if (!fc->starts.empty() &&
                                    (mq_ctx_submit_spans(ctx[sl], fc->buf, fc->bytes, fc->starts.data(), fc->lens.data(), (uint32_t)fc->starts.size(), fc->hits.data()) != MQ_OK ||
                                     mq_ctx_wait(ctx[sl]) != MQ_OK))
                                    fail(std::string("mq_ctx_submit_spans: ") + last_error());
                            } catch (const std::exception &e) { fail(e.what()); }
                        } else {
                            feeder::spans_from_line_ends(*fc, line_ends, n_lines, fx_lpr);
                            fc->hits.assign(hits, hits + n);
                        }
                        fc->unparsed = false;
                    } else if (mq_ctx_wait(ctx[sl]) != MQ_OK) fail(std::string("mq_ctx_wait: ") + last_error());
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        to_format.push_back(inflight[sl]);
                    }
                    inflight[sl] = nullptr;
                    cv.notify_all();
                    t_finish_us += us_since(tf0);
                };
                // the slot to use next: a free one, else the one submitted longest ago (-1 with free_only when none is free)
                auto oldest_busy = [&]() {
                    int best = -1;
                    for (int sl = 0; sl < n_slots; ++sl)
                        if (inflight[sl] && (best < 0 || age[sl] < age[best])) best = sl;
                    return best;
                };
                try {
                    for (;;) {
                        if (failed()) break;  // somebody failed: stop pulling chunks
                        // Never wait for a new chunk while holding submitted ones: the writer may be waiting for exactly one of
                        // them while every other buffer of the pool sits behind the writer (formatted, out of turn) -- then no
                        // new chunk can ever be parsed.  With nothing ready, the oldest submitted chunk is passed on first.
                        bool end = false;
                        Chunk *c = feed.poll(end);
                        if (!c) {
                            if (end) break;
                            const int busy = oldest_busy();
                            if (busy >= 0) {
                                finish_slot(busy);
                                continue;
                            }
                            const auto tq0 = Clock::now();
                            c = feed.next();
                            t_fetch_us += us_since(tq0);
                            if (!c) break;
                        }
                        if (fail_at >= 0 && (long)c->seq_no == fail_at) throw Error("injected failure (MQ_DRIVER_FAIL_AT)");
                        int sl = -1;
                        for (int q = 0; q < n_slots; ++q)
                            if (!inflight[q]) { sl = q; break; }
                        if (sl < 0) {
                            sl = oldest_busy();
                            finish_slot(sl);
                        }
                        if (c->unparsed) {
                            const auto ts0 = Clock::now();
                            if (mq_ctx_submit_fastx(ctx[sl], c->buf, c->begin, c->bytes, fx_format) != MQ_OK) throw Error(std::string("mq_ctx_submit_fastx: ") + last_error());
                            t_submit_us += us_since(ts0);
                            inflight[sl] = c;
                            age[sl] = submitted++;
                            continue;
                        }
                        c->hits.resize(c->starts.size());
                        if (c->starts.empty()) {  // nothing to map in this chunk (the middle of a very long record)
                            std::lock_guard<std::mutex> lk(mu);
                            to_format.push_back(c);
                            cv.notify_all();
                            continue;
                        }
                        const auto ts0 = Clock::now();
                        if (mq_ctx_submit_spans(ctx[sl], c->buf, c->bytes, c->starts.data(), c->lens.data(), (uint32_t)c->starts.size(),
                                                c->hits.data()) != MQ_OK)
                            throw Error(std::string("mq_ctx_submit_spans: ") + last_error());
                        t_submit_us += us_since(ts0);
                        inflight[sl] = c;
                        age[sl] = submitted++;
                    }
                    for (int q = oldest_busy(); q >= 0; q = oldest_busy()) finish_slot(q);
                } catch (const std::exception &e) { fail(e.what()); }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    gpu_workers_left--;
                }
                cv.notify_all();
            });
        std::vector<std::thread> formatters;
        for (int f = 0; f < n_format; ++f)
            formatters.emplace_back([&]() {
                std::string id;
                PafWriter pw(*ro[0]);
                for (;;) {
                    Chunk *c = nullptr;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return !to_format.empty() || gpu_workers_left == 0; });
                        if (to_format.empty()) return;
                        c = to_format.front();
                        to_format.pop_front();
                        formatting++;
                    }
                    const auto tm0 = Clock::now();
                    try {
                    c->paf.reserve(c->starts.size() * 96);
                    for (size_t i = 0; i < c->starts.size(); ++i) {
                        const mq_hit &h = c->hits[i];
                        if (h.status == MQ_HIT_MAPPED) {
                            pw.append(c->paf, (const char *)c->buf + c->ids[i].off, c->ids[i].len, c->lens[i], h);  // src/mers.rs:181
                            continue;
                        }
                        id.assign((const char *)c->buf + c->ids[i].off, c->ids[i].len);
                        if (h.status == MQ_HIT_UNMAPPED) {
                            if (unm) { c->unmapped += id; c->unmapped.push_back('\n'); }
                            if (ufa) {
                                c->unmapped_fa.push_back('>');
                                c->unmapped_fa += id;
                                c->unmapped_fa.push_back('\n');
                                c->unmapped_fa.append((const char *)c->buf + c->starts[i], c->lens[i]);
                                c->unmapped_fa.push_back('\n');
                            }
                        } else {
                            fail("find_matches: read " + id + " could not be processed");
                            break;
                        }
                    }
                    } catch (const std::exception &e) { fail(e.what()); }
                    t_format_us += us_since(tm0);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        done[c->seq_no] = c;
                        formatting--;
                    }
                    cv.notify_all();
                }
            });
        // main thread: chunks in input order (main_thread_mer, src/closures.rs:117-123)
        for (size_t next_out = 0;;) {
            Chunk *c = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] {
                    return done.count(next_out) != 0 || !werr.empty() || (gpu_workers_left == 0 && to_format.empty() && formatting == 0);
                });
                auto it = done.find(next_out);
                if (it == done.end()) break;  // everything written, or a worker failed
                c = it->second;
                done.erase(it);
            }
            const auto tw0 = Clock::now();
            if (!c->paf.empty()) fwrite(c->paf.data(), 1, c->paf.size(), paf);
            if (unm && !c->unmapped.empty()) fwrite(c->unmapped.data(), 1, c->unmapped.size(), unm);
            if (ufa && !c->unmapped_fa.empty()) fwrite(c->unmapped_fa.data(), 1, c->unmapped_fa.size(), ufa);
            feed.recycle(c);
            t_write_us += us_since(tw0);
            ++next_out;
        }
        // On a failure the chunks in flight are never recycled, so the feeder's workers (waiting for a buffer) and the GPU workers
        // (waiting for a chunk) would wait forever: the feeder is told to give up, which wakes both.
        if (failed()) feed.abort();
        for (auto &t : workers) t.join();
        cv.notify_all();
        for (auto &t : formatters) if (t.joinable()) t.join();
        if (!werr.empty()) {
            for (auto &v : slots) for (auto c : v) mq_ctx_free(c);
            fclose(paf);
            if (unm) fclose(unm);
            if (ufa) fclose(ufa);
            remove((prefix + ".paf").c_str());  // never leave a partial PAF behind a failure
            throw Error(werr);
        }
        fclose(paf);
        if (unm) fclose(unm);
        if (ufa) fclose(ufa);
        if (drv_timing)
            fprintf(stderr, "map phase %.3f s; summed over threads: submit %.3f s, finish (wait + spans) %.3f s, waiting for a chunk %.3f s (%d submitters), "
                            "format %.3f s (%d formatters), write + recycle %.3f s\n", secs(t0), t_submit_us / 1e6, t_finish_us / 1e6, t_fetch_us / 1e6, o.gpus * n_sub,
                    t_format_us / 1e6, n_format, t_write_us / 1e6);
        printf("Mapped query sequences in %s.\n", rust_duration(secs(t0)).c_str());  // src/closures.rs:211
        tl("PAF written");
        if (o.last_pass && getenv("MQ_DRIVER_FAST_EXIT") != nullptr) {
            // EXPERIMENT: the run's output is complete and closed -- leave without unwinding (stream slots, the table, the page-locked
            // pool: the operating system takes a process's device and host memory back in one go when it ends)
            print_totals(g_t_main);
            fflush(stdout);
            fflush(stderr);
            _exit(0);
        }
        // teardown, side by side: the stream slots and then the indexes (contexts go before their index) on one thread, the feeder's
        // page-locked pool on this one -- 0.08 s one after the other on a 0.6-s job
        {
            std::thread dev_side([&]() {
                for (auto &v : slots) for (auto c : v) mq_ctx_free(c);
                ro.clear();
            });
            if (preload) ref_reaper = std::thread([pl = preload.release()]() { delete pl; });
            feed.release_buffers();
            dev_side.join();
            if (ref_reaper.joinable()) ref_reaper.join();
        }
        tl("stream slots, indexes and the page-locked pool released");
    return 0;
}

int main(int argc, char **argv) {
    const auto start = Clock::now();
    Opt o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "error: %s needs a value\n", a.c_str()); exit(2); }
            return argv[++i];
        };
        if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a == "--debug") o.debug = true;
        else if (a == "--low-memory") o.low_memory = true;
        else if (a == "--nosimd") o.nosimd = true;
        else if (a == "--nohpc") o.nohpc = true;
        else if (a == "--parallelfastx") o.parallelfastx = true;
        else if (a == "--unmapped") o.unmapped = true;
        else if (a == "-p" || a == "--prefix") { o.prefix = val(); o.has_prefix = true; }
        else if (a == "-k") o.k = atol(val());
        else if (a == "-l") o.l = atol(val());
        else if (a == "-d" || a == "--density") o.density = atof(val());
        else if (a == "-c" || a == "--chain") o.c = atol(val());
        else if (a == "-s" || a == "--seed") o.s = atol(val());
        else if (a == "-g" || a == "--gap-diff") o.g = atol(val());
        else if (a == "--reference") o.reference = val();
        else if (a == "--threads") o.threads = atol(val());
        else if (a == "-b") o.b = atol(val());
        else if (a == "-q") o.q = atol(val());
        else if (a == "--device") o.device = atoi(val());
        else if (a == "--gpus") o.gpus = std::max(1, atoi(val()));
        else if (a == "--batch-bases") o.batch_bases = strtoull(val(), nullptr, 10);
        else if (a == "--table-factor") {
            o.table_factor = atoi(val());
            if (o.table_factor < 2 || o.table_factor > 64) { fprintf(stderr, "error: --table-factor wants 2..64\n"); return 2; }
        }
        else if (a == "--save-index") o.save_index = val();
        else if (a == "--index") o.load_index = val();
        else if (a == "--seeding-variant") {
            o.seeding_variant = atoi(val());
            if (o.seeding_variant < 0 || o.seeding_variant > 63) { fprintf(stderr, "error: --seeding-variant wants 0..63 (bits 1 2 4 8 16 32)\n"); return 2; }
        }
        else if (a == "--fast-kh") o.fast_kh = true;
        else if (a == "--second-pass") {
            o.second = val();
            char *e1 = nullptr, *e2 = nullptr;
            o.k2 = strtol(o.second.c_str(), &e1, 10);
            o.l2 = (e1 && *e1 == ',') ? strtol(e1 + 1, &e2, 10) : 0;
            o.d2 = (e2 && *e2 == ',') ? atof(e2 + 1) : -1;
            if (o.k2 < 1 || o.l2 < 1 || !(o.d2 > 0)) { fprintf(stderr, "error: --second-pass wants k2,l2,d2\n"); return 2; }
        }
        else if (!a.empty() && a[0] == '-') { fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", a.c_str()); return 2; }
        else o.reads = a;
    }
    if (o.reads.empty()) { fprintf(stderr, "Please specify an input file.\n"); return 101; }          // panic!, src/main.rs:191
    if (o.reference.empty() && o.load_index.empty()) { fprintf(stderr, "Please specify a reference file.\n"); return 101; }   // src/main.rs:192
    if (!o.load_index.empty() && (!o.save_index.empty() || !o.second.empty())) {
        fprintf(stderr, "error: --index cannot be combined with --save-index or --second-pass (a second pass builds its own index)\n");
        return 2;
    }

    Params P;
    size_t threads = 8;
    const bool reads_fasta = is_fasta_name(o.reads), ref_fasta = is_fasta_name(o.reference);
    if (reads_fasta) printf("Input file: %s\nFormat: FASTA\n", o.reads.c_str());
    if (ref_fasta && o.load_index.empty()) printf("Reference file: %s\nFormat: FASTA\n", o.reference.c_str());
    if (o.k >= 0) P.k = (size_t)o.k; else printf("Warning: Using default k value (%zu).\n", P.k);
    if (o.l >= 0) P.l = (size_t)o.l; else printf("Warning: Using default l value (%zu).\n", P.l);
    if (o.b >= 0) P.b = (size_t)o.b; else printf("Warning: Using default buffer size (%zuX).\n", P.b);
    if (o.q >= 0) P.q = (size_t)o.q; else printf("Warning: Using default queue length (%zu).\n", P.q);
    if (o.density >= 0) P.density = o.density; else printf("Warning: Using default density value (%s%%).\n", rust_float(P.density * 100.0).c_str());
    if (o.threads >= 0) threads = (size_t)o.threads; else printf("Warning: Using default number of threads (8).\n");
    if (o.c >= 0) P.c = (size_t)o.c; else printf("Warning: Using default minimum chain length (%zu).\n", P.c);
    if (o.s >= 0) P.s = (size_t)o.s; else printf("Warning: Using default minimum number of matching seeds (%zu).\n", P.s);
    if (o.g >= 0) P.g = (size_t)o.g; else printf("Warning: Using default maximum seed gap difference (%zu).\n", P.g);
    std::string prefix = "mapquik-k" + std::to_string(P.k) + "-d" + rust_float(P.density) + "-l" + std::to_string(P.l);
    if (o.has_prefix) prefix = o.prefix; else printf("Warning: Using default output prefix (%s).\n", prefix.c_str());
    P.debug = o.debug;
    P.use_hpc = !o.nohpc;
    P.use_simd = !o.nosimd;
    P.use_pfx = o.parallelfastx;
    P.fold_case = true;  // raw FASTX bytes go to the GPU: the kernels do the reference's to_ascii_uppercase
    P.seeding_variant = (unsigned)o.seeding_variant;
    if (o.seeding_variant) printf("Seeding variant %d (reading of rust-seq2kminmers other than the frozen one; include/mapquik_hip.h).\n", o.seeding_variant);
    P.fast_kh = o.fast_kh;
    if (o.fast_kh) puts("Fast k-min-mer tuple hash (MQ_FLAG_FAST_KH): same PAF, KminmerHash.hash is not the reference's value.");
    if (P.use_hpc) puts(P.use_simd ? "Using HPC ntHash, with SIMD" : "Using HPC ntHash, scalar");
    else puts(P.use_simd ? "Using regular ntHash (not HPC), with SIMD" : "Using regular ntHash (not HPC), scalar");

    const std::string second_prefix = prefix + "-" + std::to_string(o.k2) + "-" + std::to_string(o.l2) + "-" + rust_float(o.d2);
    try {
        tl("arguments parsed");
        o.last_pass = o.second.empty();
        int rc = run_pass(o, P, o.reads, reads_fasta, ref_fasta, prefix, threads, o.second.empty() ? std::string() : second_prefix + ".fa");
        tl("run_pass returned (index, feeder and its page-locked pool released)");
        if (rc) return rc;
        if (!o.second.empty()) {
            // the second pass of experiments/chm13/run_chm13_mapquik_unmapped.sh:8-24: the reads the first pass left unmapped,
            // mapped again with (k2, l2, d2) against a second index of the same reference
            P.k = (size_t)o.k2;
            P.l = (size_t)o.l2;
            P.density = o.d2;
            printf("Second pass: %s with k=%zu l=%zu density=%s\n", (second_prefix + ".fa").c_str(), P.k, P.l, rust_float(P.density).c_str());
            o.last_pass = true;
            rc = run_pass(o, P, second_prefix + ".fa", true, ref_fasta, second_prefix, threads, std::string());
            if (rc) return rc;
        }
    } catch (const std::exception &e) {  // mapquik::Error, feeder::FeederError: the reference panics (exit code 101)
        fflush(stdout);
        fprintf(stderr, "mapquik: %s\n", e.what());
        return 101;
    }
    tl("all passes done");
    print_totals(start);
    return 0;
}
