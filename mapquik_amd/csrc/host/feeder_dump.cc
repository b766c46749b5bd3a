// feeder_dump -- test tool for fastx_feeder.hpp (no GPU needed: chunk buffers come from malloc).
// usage: feeder_dump <file> <fasta|fastq|ref> <chunk_bytes> <threads>   -> one line per read, in input order: id TAB length TAB sequence
//        (ref: through the reference loader, ref_loader.hpp; FEEDER_DUMP_QUIET=1 prints only "records bases" -- for timing)
//        feeder_dump <file.gz> inflate <segment_bytes> <threads>        -> the inflated bytes of all members (par_gzip.hpp alone);
//        stderr: "rounds R max_chain C" (FEEDER_DUMP_QUIET=1: no bytes, "bytes seconds" on stdout)
#include <cstdio>
#include <cstdlib>
#include <map>

#include "fastx_feeder.hpp"
#include "ref_loader.hpp"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    using namespace mapquik::feeder;
    try {
        if (std::string(argv[2]) == "inflate") {
            namespace pz = mapquik::pargz;
            const int fd = open(argv[1], O_RDONLY);
            if (fd < 0) throw std::runtime_error("cannot open");
            struct stat st;
            fstat(fd, &st);
            if (st.st_size == 0) return 0;
            const uint8_t *m = (const uint8_t *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
            pz::Options o;
            o.threads = atoi(argv[4]);
            o.seg_bytes = strtoull(argv[3], nullptr, 10);
            o.min_seg_bytes = getenv("PARGZ_MINSEG") ? strtoull(getenv("PARGZ_MINSEG"), nullptr, 10) : o.seg_bytes / 4 + 1;
            if (getenv("PARGZ_RATIO")) o.max_ratio = (uint32_t)atoi(getenv("PARGZ_RATIO"));
            o.timing = getenv("MQ_FEEDER_TIMING") != nullptr;
            Deflate dl;
            if (!getenv("PARGZ_ZLIB_CRC")) o.crc_fn = dl.crc;
            uint64_t cap = (uint64_t)st.st_size * (getenv("PARGZ_OUT_RATIO") ? strtoull(getenv("PARGZ_OUT_RATIO"), nullptr, 10) : 1100) + (1u << 20);
            pz::Region out(cap);
            uint64_t pos = 0, total = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (pos < (uint64_t)st.st_size) {
                pz::MemberInflater inf(m + pos, (uint64_t)st.st_size - pos, o);
                uint64_t prod = 0;
                pos += inf.run((uint8_t *)out.p + total, cap - total, &prod, nullptr);
                total += prod;
                fprintf(stderr, "rounds %d max_chain %d\n", inf.rounds(), inf.max_chain());
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (getenv("FEEDER_DUMP_QUIET")) printf("%llu %.4f\n", (unsigned long long)total, dt);
            else fwrite(out.p, 1, total, stdout);
            return 0;
        }
        if (std::string(argv[2]) == "ref") {
            RefLoader rl(argv[1], atoi(argv[4]));
            const bool quiet = getenv("FEEDER_DUMP_QUIET") != nullptr;
            unsigned long long n = 0, bases = 0;
            rl.for_each([&](const RefLoader::Record &r, const uint8_t *seq) {
                ++n;
                bases += r.len;
                if (quiet) return;
                fwrite(r.id.data(), 1, r.id.size(), stdout);
                printf("\t%llu\t", (unsigned long long)r.len);
                fwrite(seq, 1, r.len, stdout);
                putchar('\n');
            });
            if (quiet) printf("%llu %llu\n", n, bases);
            return 0;
        }
        Feeder f(argv[1], std::string(argv[2]) == "fastq", strtoull(argv[3], nullptr, 10), atoi(argv[4]), atoi(argv[4]) + 4,
                 [](size_t n) { return malloc(n); }, [](void *p) { free(p); }, [](void *, size_t) { return 0; }, [](void *) { return 0; });
        // FEEDER_DUMP_UNPARSED=1: what the native driver does for an uncompressed FASTA file, with the device's part done here: chunks
        // come unparsed (FEEDER_DUMP_MAPPED=1: as views of the mapped file), the line ends are found by a plain scan, a chunk that is not
        // "header line, sequence line" all through is parsed by parse_chunk after all (materialized first when it is a view)
        const bool unparsed_mode = getenv("FEEDER_DUMP_UNPARSED") != nullptr;
        if (unparsed_mode) {
            f.leave_unparsed(true);
            f.premap();
        }
        if (getenv("FEEDER_DUMP_KIND")) fprintf(stderr, "kind=%s\n", f.kind_name());
        f.start();
        unsigned long long n_unparsed = 0, n_irregular = 0;
        auto host_scan = [&](Chunk *c) {
            ++n_unparsed;
            std::vector<uint32_t> le;
            for (uint64_t p = c->begin; p < c->bytes; ++p)
                if (c->buf[p] == '\n') le.push_back((uint32_t)p);
            if (c->bytes > c->begin && c->buf[c->bytes - 1] != '\n') le.push_back((uint32_t)c->bytes);
            bool irregular = (le.size() & 1) != 0;
            for (size_t i = 0; !irregular && i < le.size() / 2; ++i) {
                const uint64_t hs = i ? (uint64_t)le[2 * i - 1] + 1 : c->begin, he = le[2 * i], ss = he + 1;
                irregular = hs >= he || c->buf[hs] != '>' || (ss < le[2 * i + 1] && c->buf[ss] == '>');
            }
            if (irregular) {
                ++n_irregular;
                c->materialize();
                parse_chunk(*c, false);
            } else {
                spans_from_line_ends(*c, le.data(), (uint32_t)le.size());
            }
            c->unparsed = false;
        };
        std::map<size_t, Chunk *> held;
        size_t next = 0;
        const bool quiet = getenv("FEEDER_DUMP_QUIET") != nullptr;
        unsigned long long n_rec = 0, n_bases = 0;
        auto flush = [&]() {
            for (auto it = held.find(next); it != held.end(); it = held.find(next)) {
                Chunk *c = it->second;
                for (size_t i = 0; i < c->starts.size(); ++i) {
                    if (quiet) {
                        ++n_rec;
                        n_bases += c->lens[i];
                        continue;
                    }
                    fwrite(c->buf + c->ids[i].off, 1, c->ids[i].len, stdout);
                    printf("\t%u\t", c->lens[i]);
                    fwrite(c->buf + c->starts[i], 1, c->lens[i], stdout);
                    putchar('\n');
                }
                held.erase(it);
                f.recycle(c);
                ++next;
            }
        };
        while (Chunk *c = f.next()) {
            if (c->unparsed) host_scan(c);
            held[c->seq_no] = c;
            flush();
        }
        flush();
        if (unparsed_mode) fprintf(stderr, "unparsed chunks %llu irregular %llu mapped %d\n", n_unparsed, n_irregular, f.mapped_views() ? 1 : 0);
        if (!held.empty()) {
            fprintf(stderr, "missing chunk %zu\n", next);
            return 1;
        }
        if (quiet) printf("%llu %llu\n", n_rec, n_bases);
    } catch (const std::exception &e) {
        fprintf(stderr, "feeder_dump: %s\n", e.what());
        return 1;
    }
    return 0;
}
