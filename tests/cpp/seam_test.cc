// seam_test -- drives the C++ mirror of the reference's seam (mapquik_host.hpp: Index, ReadOnlyIndex, mers::ref_extract,
// mers::find_matches, mers::find_matches_batch; src/mers.rs:15,77 and src/index.rs:73-128) the way src/closures.rs does, and
// prints what the reference would write to <prefix>.paf.  tests/test_gpu_parity.py compares the output with the oracle's PAF.
//   usage: seam_test <ref.fa> <reads.fa> <single|batch> [k l density]      (single-line FASTA, '>' id lines)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../mapquik_amd/csrc/host/mapquik_host.hpp"

using namespace mapquik;

static std::vector<std::pair<std::string, std::string>> read_fasta(const char *path) {
    std::vector<std::pair<std::string, std::string>> out;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line[0] == '>') out.emplace_back(line.substr(1, line.find(' ') == std::string::npos ? std::string::npos : line.find(' ') - 1), std::string());
        else if (!out.empty()) out.back().second += line;
    }
    return out;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    try {
        Params P;
        if (argc >= 7) {
            P.k = (size_t)atol(argv[4]);
            P.l = (size_t)atol(argv[5]);
            P.density = atof(argv[6]);
        }
        const auto refs = read_fasta(argv[1]);
        const auto reads = read_fasta(argv[2]);
        Index index(P);
        for (size_t i = 0; i < refs.size(); ++i)
            mers::ref_extract(i, refs[i].first, (const uint8_t *)refs[i].second.data(), refs[i].second.size(), P, index);
        const ReadOnlyIndex ro = std::move(index).into_read_only();
        fprintf(stderr, "unique %llu\n", (unsigned long long)ro.unique_count());
        if (std::string(argv[3]) == "single") {
            for (const auto &r : reads) {
                const auto line = mers::find_matches(r.first, r.second.size(), (const uint8_t *)r.second.data(), ro, P);
                if (line) puts(line->c_str());
            }
        } else {
            std::vector<std::string> ids;
            std::string bases;
            std::vector<uint64_t> offsets(1, 0);
            for (const auto &r : reads) {
                ids.push_back(r.first);
                bases += r.second;
                offsets.push_back(bases.size());
            }
            for (const auto &line : mers::find_matches_batch(ids, (const uint8_t *)bases.data(), offsets, ro, P))
                if (line) puts(line->c_str());
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "seam_test: %s\n", e.what());
        return 101;
    }
    return 0;
}
