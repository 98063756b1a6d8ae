// Shared pieces of the three command-line tools that mirror akaze-util/src/bin/*.rs on top of the C ABI
// (include/akaze_hip.h).  Host C++ only; the work happens in libakaze_hip.so.
#pragma once
#include <sys/stat.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/akaze_hip.h"

namespace cli {

// env_logger with filter_or("AKAZE_LOG", "info") (extract_features.rs:57-58): info lines always, debug
// lines when AKAZE_LOG=debug / trace.
inline bool debug_enabled() {
    const char* e = getenv("AKAZE_LOG");
    return e && (!strcmp(e, "debug") || !strcmp(e, "trace"));
}
#define CLI_INFO(...)  do { fprintf(stderr, "INFO  "); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } while (0)
#define CLI_DEBUG(...) do { if (cli::debug_enabled()) { fprintf(stderr, "DEBUG "); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

// The reference's binaries panic on any failure (unwrap / expect); these print the reason and exit 1.
[[noreturn]] inline void die(const char* what) {
    fprintf(stderr, "error: %s: %s\n", what, akz_last_error());
    exit(1);
}
#define CLI_TRY(expr) do { if ((expr) != AKZ_OK) cli::die(#expr); } while (0)

struct Opt {
    char short_name;
    const char* long_name;
    const char* value_name;
    const char* help;
};
struct Spec {
    const char* name;
    const char* about;
    std::vector<std::pair<const char*, const char*>> positionals;  // name, help
    std::vector<Opt> options;
};
struct Args {
    std::vector<std::string> pos;
    std::map<std::string, std::string> opt;  // by long name
    const char* get(const char* long_name) const {
        auto it = opt.find(long_name);
        return it == opt.end() ? nullptr : it->second.c_str();
    }
};

inline void usage(const Spec& s, FILE* f) {
    fprintf(f, "%s 0.1\nJohn Stalbaum\n%s\n\nUSAGE:\n    %s [OPTIONS]", s.name, s.about, s.name);
    for (auto& p : s.positionals) fprintf(f, " <%s>", p.first);
    fprintf(f, "\n\nFLAGS:\n    -h, --help       Prints help information\n    -V, --version    Prints version information\n");
    if (!s.options.empty()) fprintf(f, "\nOPTIONS:\n");
    for (auto& o : s.options) fprintf(f, "    -%c, --%s <%s>    %s\n", o.short_name, o.long_name, o.value_name, o.help);
    fprintf(f, "\nARGS:\n");
    for (auto& p : s.positionals) fprintf(f, "    <%s>    %s\n", p.first, p.second);
}

// clap 2.33 surface of the reference tools: positionals by index, `-x VALUE` / `--long VALUE` / `--long=VALUE`.
inline Args parse(const Spec& s, int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string t = argv[i];
        if (t == "-h" || t == "--help") { usage(s, stdout); exit(0); }
        if (t == "-V" || t == "--version") { printf("%s 0.1\n", s.name); exit(0); }
        const Opt* hit = nullptr;
        std::string inline_val;
        bool has_inline = false;
        if (t.size() >= 2 && t[0] == '-' && t[1] != '-') {
            for (auto& o : s.options)
                if (o.short_name == t[1]) hit = &o;
            if (hit && t.size() > 2) { inline_val = t.substr(t[2] == '=' ? 3 : 2); has_inline = true; }
        } else if (t.size() > 2 && t[0] == '-' && t[1] == '-') {
            const size_t eq = t.find('=');
            const std::string name = t.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            for (auto& o : s.options)
                if (name == o.long_name) hit = &o;
            if (hit && eq != std::string::npos) { inline_val = t.substr(eq + 1); has_inline = true; }
        } else {
            a.pos.push_back(t);
            continue;
        }
        if (!hit) {
            fprintf(stderr, "error: Found argument '%s' which wasn't expected, or isn't valid in this context\n\n", t.c_str());
            usage(s, stderr);
            exit(1);
        }
        if (has_inline) a.opt[hit->long_name] = inline_val;
        else if (i + 1 < argc) a.opt[hit->long_name] = argv[++i];
        else {
            fprintf(stderr, "error: The argument '--%s <%s>' requires a value but none was supplied\n", hit->long_name, hit->value_name);
            exit(1);
        }
    }
    if (a.pos.size() != s.positionals.size()) {
        fprintf(stderr, "error: The following required arguments were not provided (or too many were given):\n");
        for (size_t i = a.pos.size(); i < s.positionals.size(); ++i) fprintf(stderr, "    <%s>\n", s.positionals[i].first);
        fprintf(stderr, "\n");
        usage(s, stderr);
        exit(1);
    }
    return a;
}

struct Features {
    std::vector<akz_keypoint> keypoints;
    std::vector<uint8_t> descriptors;
    uint64_t desc_bytes = 0;
};

// akaze::extract_features(path, options) -> Features (+ the result handle when the caller wants the planes)
inline Features extract(akz_ctx* ctx, const char* path, const akz_config& cfg, bool keep_planes, akz_result** keep = nullptr) {
    akz_result* res = nullptr;
    CLI_TRY(akz_extract_features_file(ctx, path, &cfg, keep_planes ? AKZ_KEEP_ALL_PLANES : 0, &res));
    Features f;
    uint64_t nl = 0, nk = 0;
    CLI_TRY(akz_result_counts(res, 0, &nl, &nk, &f.desc_bytes));
    f.keypoints.resize(nk);
    f.descriptors.resize(nk * f.desc_bytes);
    if (nk) {
        CLI_TRY(akz_result_keypoints(res, 0, f.keypoints.data()));
        CLI_TRY(akz_result_descriptors(res, 0, f.descriptors.data()));
    }
    if (keep) *keep = res;
    else akz_result_free(res);
    return f;
}

inline Features read_features(const char* path) {
    Features f;
    uint64_t nk = 0, nd = 0;
    CLI_TRY(akz_read_features(path, nullptr, nullptr, 0, 0, &nk, &nd, &f.desc_bytes));
    f.keypoints.resize(nk);
    f.descriptors.resize(nd * f.desc_bytes);
    CLI_TRY(akz_read_features(path, f.keypoints.data(), f.descriptors.data(), nk, nd * f.desc_bytes, &nk, &nd, &f.desc_bytes));
    if (nk != nd) {  // the reference would index keypoints[match.index] past the end and panic (lib.rs:267-274)
        fprintf(stderr, "error: %s holds %llu keypoints but %llu descriptors\n", path, (unsigned long long)nk, (unsigned long long)nd);
        exit(1);
    }
    return f;
}

inline std::vector<akz_match> match(akz_ctx* ctx, const Features& a, const Features& b) {
    // match_features(.., 0.86, 1000, 3.0): the constants of match_features.rs:70-78 / extract_and_match.rs:101-109
    std::vector<akz_match> m(a.keypoints.size() ? a.keypoints.size() : 1);
    uint64_t n = 0;
    if (a.desc_bytes && b.desc_bytes && a.desc_bytes != b.desc_bytes) {
        fprintf(stderr, "error: the two feature sets have descriptors of different lengths (%llu and %llu bytes)\n",
                (unsigned long long)a.desc_bytes, (unsigned long long)b.desc_bytes);
        exit(1);
    }
    const uint64_t db = a.desc_bytes ? a.desc_bytes : b.desc_bytes;
    const uint64_t nd0 = db ? a.descriptors.size() / db : 0, nd1 = db ? b.descriptors.size() / db : 0;
    CLI_TRY(akz_match_features(ctx, a.keypoints.data(), a.keypoints.size(), a.descriptors.data(), nd0, b.keypoints.data(),
                               b.keypoints.size(), b.descriptors.data(), nd1, db, 0.86, 1000, 3.0f, m.data(), &n));
    m.resize(n);
    return m;
}

// the debug pictures are PNG streams (the reference's RgbImage::save picks the encoder from the extension; this
// library has a PNG encoder only): other extensions are refused rather than written with the wrong content
inline bool is_png_path(const char* p) {
    const size_t n = strlen(p);
    return n >= 4 && (strcmp(p + n - 4, ".png") == 0 || strcmp(p + n - 4, ".PNG") == 0);
}
inline void require_png_path(const char* p, const char* option) {
    if (!is_png_path(p)) {
        fprintf(stderr, "error: %s %s: only .png output is supported\n", option, p);
        exit(1);
    }
}

inline akz_ctx* open_context() {
    akz_ctx* ctx = nullptr;
    CLI_TRY(akz_ctx_create(0, nullptr, &ctx));
    return ctx;
}

inline bool file_exists(const char* p) {
    struct stat st;
    return stat(p, &st) == 0;
}
inline bool mkdir_p(const std::string& dir) {
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && !file_exists(cur.c_str()) && mkdir(cur.c_str(), 0777) != 0) return false;
        }
        if (i < dir.size()) cur += dir[i];
    }
    return true;
}

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

}  // namespace cli
