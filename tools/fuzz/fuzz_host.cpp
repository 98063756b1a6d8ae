// Sanitizer fuzz driver for the host-side parsers of libakaze_hip.so (CPU build only: AddressSanitizer and
// UBSan are not available for the GPU code on this pool).  Built and run by tests/test_fuzz_host.py:
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -I../../akaze-rust_amd/csrc fuzz_host.cpp \
//       ../../akaze-rust_amd/csrc/akz_image.cpp ../../akaze-rust_amd/csrc/akz_io.cpp -lz
// usage: fuzz_host <iterations> <workdir> <seed files...>; every seed is corrupted <iterations> times
// (truncation, byte flips, 0xff runs, deletions) and fed to the decoder that matches its extension.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/akaze_hip.h"

#include "../../akaze-rust_amd/csrc/akz_internal.hpp"

namespace akz {  // the symbols the parsers need from the rest of the library
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
DefaultSource& default_source() {
    static thread_local DefaultSource src;
    return src;
}
}  // namespace akz
extern "C" const char* akz_last_error(void) { return akz::g_err.c_str(); }

static std::vector<unsigned char> slurp(const char* p) {
    std::vector<unsigned char> v;
    FILE* f = fopen(p, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n > 0 ? (size_t)n : 0);
    if (n > 0 && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}
static bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int iters = atoi(argv[1]);
    const std::string work = argv[2];
    std::mt19937 rng(20261003);
    int ok = 0, err = 0;
    for (int a = 3; a < argc; ++a) {
        const std::string name = argv[a];
        const std::vector<unsigned char> base = slurp(argv[a]);
        if (base.empty()) { fprintf(stderr, "cannot read seed %s\n", argv[a]); return 2; }
        const bool features = name.find("features") != std::string::npos, matches = name.find("matches") != std::string::npos;
        const std::string cur = work + "/cur" + (ends_with(name, ".json") ? ".json" : ends_with(name, ".cbor") ? ".cbor" : ".img");
        for (int it = 0; it < iters; ++it) {
            std::vector<unsigned char> v = base;
            switch (it % 4) {
                case 0: v.resize(rng() % v.size()); break;
                case 1: for (int k = 0; k < 1 + (int)(rng() % 8); ++k) v[rng() % v.size()] = (unsigned char)rng(); break;
                case 2: { const size_t p = rng() % v.size(), n = std::min<size_t>(v.size() - p, 1 + rng() % 64); std::fill(v.begin() + p, v.begin() + p + n, 0xff); } break;
                default: { const size_t p = rng() % v.size(); v.erase(v.begin() + p, v.begin() + std::min(v.size(), p + 1 + rng() % 200)); } break;
            }
            FILE* f = fopen(cur.c_str(), "wb");
            if (!f) return 2;
            fwrite(v.data(), 1, v.size(), f);
            fclose(f);
            int st;
            if (features) {
                uint64_t nk = 0, nd = 0, nb = 0;
                st = akz_read_features(cur.c_str(), nullptr, nullptr, 0, 0, &nk, &nd, &nb);
                if (st == AKZ_OK && nk < 100000 && nd * nb < (1u << 26)) {
                    std::vector<akz_keypoint> k(nk ? nk : 1);
                    std::vector<uint8_t> d(nd * nb ? nd * nb : 1);
                    st = akz_read_features(cur.c_str(), k.data(), d.data(), nk, nd * nb, &nk, &nd, &nb);
                }
            } else if (matches) {
                uint64_t n = 0;
                st = akz_read_matches(cur.c_str(), nullptr, 0, &n);
                if (st == AKZ_OK && n < 1000000) {
                    std::vector<akz_match> m(n ? n : 1);
                    st = akz_read_matches(cur.c_str(), m.data(), n, &n);
                }
            } else {
                uint32_t w = 0, h = 0, c = 0;
                uint8_t* px = nullptr;
                st = akz_image_load(cur.c_str(), &w, &h, &c, &px);
                if (st == AKZ_OK) akz_image_free(px);
            }
            st == AKZ_OK ? ++ok : ++err;
        }
    }
    printf("decoded %d, rejected %d\n", ok, err);
    return 0;
}
