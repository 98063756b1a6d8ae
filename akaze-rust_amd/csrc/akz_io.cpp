// On-disk formats of akaze-util (akaze-util/src/lib.rs:10-67), SURVEY.md 8(f) rank 2, so that files
// written by the reference's CLIs can be consumed here and vice versa.  Host code.
//
//   Features { keypoints: Vec<Keypoint>, descriptors: Vec<Descriptor> }   (lib.rs:10-14)
//   matches: [Match]                                                       (lib.rs:44-53)
//
// The file extension selects the encoding exactly as the reference does (lib.rs:24-28): ".json" ->
// serde_json, anything else (the CLIs use ".cbor"!) -> bincode 1.x default options: little-endian,
// fixed-width integers, u64 sequence lengths, usize as u64, tuples/structs as the plain concatenation of
// their fields.  Keypoint = point.0 f32, point.1 f32, response f32, size f32, octave u64, class_id u64,
// angle f32 (36 bytes, akaze/src/types/keypoint.rs:8-30); Descriptor = u64 length + bytes; Match =
// index_0 u64, index_1 u64, distance f64 (feature_match.rs:9-16).
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "akz_internal.hpp"

namespace akz {
namespace {

bool is_json(const char* path) {
    const char* dot = std::strrchr(path, '.');
    const char* slash = std::strrchr(path, '/');
    return dot && (!slash || dot > slash) && std::strcmp(dot, ".json") == 0;
}
int read_file(const char* path, std::string& out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) {
        set_error(std::string("cannot open ") + path);
        return AKZ_ERR_INVALID_ARG;
    }
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
    std::fclose(f);
    return AKZ_OK;
}
int write_file(const char* path, const std::string& data) {
    FILE* f = std::fopen(path, "wb");
    if (!f || std::fwrite(data.data(), 1, data.size(), f) != data.size()) {
        if (f) std::fclose(f);
        set_error(std::string("cannot write ") + path);
        return AKZ_ERR_INVALID_ARG;
    }
    std::fclose(f);
    return AKZ_OK;
}
template <typename T>
void put(std::string& s, T v) { s.append(reinterpret_cast<const char*>(&v), sizeof(T)); }  // host is little-endian
template <typename T>
bool get(const std::string& s, size_t& pos, T& v) {
    if (pos + sizeof(T) > s.size()) return false;
    std::memcpy(&v, s.data() + pos, sizeof(T));
    pos += sizeof(T);
    return true;
}
template <typename T>
void num(std::string& s, T v) {  // shortest text that round-trips, like serde_json's ryu output
    if (!std::isfinite((double)v)) {
        s += "null";  // serde_json writes non-finite floats as null
        return;
    }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);
    std::string t(buf, r.ptr);
    if (t.find_first_of(".eEn") == std::string::npos) t += ".0";  // keep it a float for serde
    s += t;
}

// ---- a minimal JSON reader (objects, arrays, numbers, null; strings only as keys) ----
struct Json {
    const std::string& s;
    size_t p = 0;
    bool ok = true;
    explicit Json(const std::string& str) : s(str) {}
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) ++p; }
    bool eat(char c) {
        ws();
        if (p < s.size() && s[p] == c) { ++p; return true; }
        return false;
    }
    bool expect(char c) { if (!eat(c)) ok = false; return ok; }
    std::string key() {
        std::string k;
        if (!expect('"')) return k;
        while (p < s.size() && s[p] != '"') k += s[p++];
        expect('"');
        expect(':');
        return k;
    }
    double number() {
        ws();
        if (s.compare(p, 4, "null") == 0) { p += 4; return std::nan(""); }
        const char* b = s.c_str() + p;
        char* e = nullptr;
        const double v = std::strtod(b, &e);
        if (e == b) ok = false;
        p += (size_t)(e - b);
        return v;
    }
    // an unsigned integer field (serde's u8 / u32 / u64 / usize): digits only, exact, at most `max`; anything else —
    // null, a sign, a fraction, an exponent, a value out of range — is an error, as it is for serde
    uint64_t uint(uint64_t max) {
        ws();
        const size_t b = p;
        uint64_t v = 0;
        bool over = false;
        while (p < s.size() && s[p] >= '0' && s[p] <= '9') {
            const uint64_t d = (uint64_t)(s[p] - '0');
            if (v > (UINT64_MAX - d) / 10) over = true;
            else v = v * 10 + d;
            ++p;
        }
        if (p == b || over || v > max || (p < s.size() && (s[p] == '.' || s[p] == 'e' || s[p] == 'E'))) ok = false;
        return ok ? v : 0;
    }
    void skip() {  // any value
        ws();
        if (eat('{')) { if (!eat('}')) { do { key(); skip(); } while (ok && eat(',')); expect('}'); } }
        else if (eat('[')) { if (!eat(']')) { do skip(); while (ok && eat(',')); expect(']'); } }
        else if (p < s.size() && s[p] == '"') { ++p; while (p < s.size() && s[p] != '"') { if (s[p] == '\\') ++p; ++p; } expect('"'); }
        else if (s.compare(p, 4, "true") == 0) p += 4;
        else if (s.compare(p, 5, "false") == 0) p += 5;
        else number();
    }
};

struct FeaturesData {
    std::vector<akz_keypoint> kps;
    std::vector<std::vector<uint8_t>> desc;
};

int parse_features_bincode(const std::string& s, FeaturesData& f) {
    size_t pos = 0;
    uint64_t n = 0;
    if (!get(s, pos, n) || n > s.size()) goto bad;
    f.kps.resize((size_t)n);
    for (auto& k : f.kps) {
        std::memset(&k, 0, sizeof(k));
        if (!get(s, pos, k.x) || !get(s, pos, k.y) || !get(s, pos, k.response) || !get(s, pos, k.size) ||
            !get(s, pos, k.octave) || !get(s, pos, k.class_id) || !get(s, pos, k.angle))
            goto bad;
    }
    if (!get(s, pos, n) || n > s.size()) goto bad;
    f.desc.resize((size_t)n);
    for (auto& d : f.desc) {
        uint64_t len = 0;
        if (!get(s, pos, len) || len > s.size() - pos) goto bad;  // (pos + len could wrap)
        d.assign(s.begin() + (long)pos, s.begin() + (long)(pos + len));
        pos += len;
    }
    if (pos != s.size()) goto bad;
    return AKZ_OK;
bad:
    set_error("malformed bincode Features file");
    return AKZ_ERR_INVALID_ARG;
}
int parse_features_json(const std::string& s, FeaturesData& f) {
    Json j(s);
    j.expect('{');
    if (!j.eat('}')) {
        do {
            const std::string k = j.key();
            if (k == "keypoints") {
                j.expect('[');
                if (!j.eat(']')) {
                    do {
                        akz_keypoint kp;
                        std::memset(&kp, 0, sizeof(kp));
                        j.expect('{');
                        do {
                            const std::string fk = j.key();
                            if (fk == "point") { j.expect('['); kp.x = (float)j.number(); j.expect(','); kp.y = (float)j.number(); j.expect(']'); }
                            else if (fk == "response") kp.response = (float)j.number();
                            else if (fk == "size") kp.size = (float)j.number();
                            else if (fk == "octave") kp.octave = j.uint(UINT64_MAX);
                            else if (fk == "class_id") kp.class_id = j.uint(UINT64_MAX);
                            else if (fk == "angle") kp.angle = (float)j.number();
                            else j.skip();
                        } while (j.ok && j.eat(','));
                        j.expect('}');
                        f.kps.push_back(kp);
                    } while (j.ok && j.eat(','));
                    j.expect(']');
                }
            } else if (k == "descriptors") {
                j.expect('[');
                if (!j.eat(']')) {
                    do {
                        std::vector<uint8_t> d;
                        j.expect('{');
                        do {
                            const std::string fk = j.key();
                            if (fk == "vector") {
                                j.expect('[');
                                if (!j.eat(']')) {
                                    do d.push_back((uint8_t)j.uint(255)); while (j.ok && j.eat(','));
                                    j.expect(']');
                                }
                            } else j.skip();
                        } while (j.ok && j.eat(','));
                        j.expect('}');
                        f.desc.push_back(std::move(d));
                    } while (j.ok && j.eat(','));
                    j.expect(']');
                }
            } else {
                j.skip();
            }
        } while (j.ok && j.eat(','));
        j.expect('}');
    }
    if (!j.ok) {
        set_error("malformed JSON Features file");
        return AKZ_ERR_INVALID_ARG;
    }
    return AKZ_OK;
}

}  // namespace
}  // namespace akz

using namespace akz;

extern "C" {

// akaze_util::serialize_features_to_file — akaze-util/src/lib.rs:17-30
int akz_write_features(const char* path, const akz_keypoint* kps, uint64_t n, const uint8_t* desc,
                       uint64_t desc_bytes) {
    if (!path || (n && (!kps || !desc))) return AKZ_ERR_INVALID_ARG;
    std::string s;
    if (is_json(path)) {
        s += "{\"keypoints\":[";
        for (uint64_t i = 0; i < n; ++i) {
            if (i) s += ',';
            s += "{\"point\":[";
            num(s, kps[i].x); s += ','; num(s, kps[i].y);
            s += "],\"response\":"; num(s, kps[i].response);
            s += ",\"size\":"; num(s, kps[i].size);
            s += ",\"octave\":" + std::to_string(kps[i].octave) + ",\"class_id\":" + std::to_string(kps[i].class_id);
            s += ",\"angle\":"; num(s, kps[i].angle);
            s += '}';
        }
        s += "],\"descriptors\":[";
        for (uint64_t i = 0; i < n; ++i) {
            if (i) s += ',';
            s += "{\"vector\":[";
            for (uint64_t b = 0; b < desc_bytes; ++b) {
                if (b) s += ',';
                s += std::to_string((unsigned)desc[i * desc_bytes + b]);
            }
            s += "]}";
        }
        s += "]}";
    } else {
        put<uint64_t>(s, n);
        for (uint64_t i = 0; i < n; ++i) {
            put(s, kps[i].x); put(s, kps[i].y); put(s, kps[i].response); put(s, kps[i].size);
            put<uint64_t>(s, kps[i].octave); put<uint64_t>(s, kps[i].class_id); put(s, kps[i].angle);
        }
        put<uint64_t>(s, n);
        for (uint64_t i = 0; i < n; ++i) {
            put<uint64_t>(s, desc_bytes);
            s.append(reinterpret_cast<const char*>(desc + i * desc_bytes), desc_bytes);
        }
    }
    return write_file(path, s);
}

// akaze_util::deserialize_features_from_file — lib.rs:33-42.  Call with kps == NULL to obtain the counts;
// descriptors must all have the same length (they do when written by either implementation).
int akz_read_features(const char* path, akz_keypoint* kps, uint8_t* desc, uint64_t cap_keypoints,
                      uint64_t cap_desc_bytes, uint64_t* n_keypoints, uint64_t* n_descriptors, uint64_t* desc_bytes) {
    if (!path) return AKZ_ERR_INVALID_ARG;
    std::string s;
    AKZ_TRY(read_file(path, s));
    FeaturesData f;
    try {
        AKZ_TRY(is_json(path) ? parse_features_json(s, f) : parse_features_bincode(s, f));
    } catch (const std::exception& e) {  // bad_alloc / length_error on absurd element counts: nothing throws across the ABI
        set_error(std::string("read_features: ") + e.what());
        return AKZ_ERR_NO_MEMORY;
    }
    const uint64_t nb = f.desc.empty() ? 0 : f.desc[0].size();
    for (const auto& d : f.desc)
        if (d.size() != nb) {
            set_error("descriptors of different lengths");
            return AKZ_ERR_UNSUPPORTED;
        }
    if (n_keypoints) *n_keypoints = f.kps.size();
    if (n_descriptors) *n_descriptors = f.desc.size();
    if (desc_bytes) *desc_bytes = nb;
    if (kps) {
        if (cap_keypoints < f.kps.size() || (desc && cap_desc_bytes < f.desc.size() * nb)) {
            set_error("read_features: buffers too small");
            return AKZ_ERR_BUFFER;
        }
        if (!f.kps.empty()) std::memcpy(kps, f.kps.data(), f.kps.size() * sizeof(akz_keypoint));
        if (desc)
            for (size_t i = 0; i < f.desc.size(); ++i) std::memcpy(desc + i * nb, f.desc[i].data(), nb);
    }
    return AKZ_OK;
}

// akaze_util::serialize_matches_to_file — lib.rs:44-53
int akz_write_matches(const char* path, const akz_match* m, uint64_t n) {
    if (!path || (n && !m)) return AKZ_ERR_INVALID_ARG;
    std::string s;
    if (is_json(path)) {
        s += '[';
        for (uint64_t i = 0; i < n; ++i) {
            if (i) s += ',';
            s += "{\"index_0\":" + std::to_string(m[i].index_0) + ",\"index_1\":" + std::to_string(m[i].index_1) +
                 ",\"distance\":";
            num(s, m[i].distance);
            s += '}';
        }
        s += ']';
    } else {
        put<uint64_t>(s, n);
        for (uint64_t i = 0; i < n; ++i) {
            put<uint64_t>(s, m[i].index_0); put<uint64_t>(s, m[i].index_1); put<double>(s, m[i].distance);
        }
    }
    return write_file(path, s);
}

// akaze_util::deserialize_matches_from_file — lib.rs:56-67.  out == NULL returns the count only.
static int read_matches_impl(const char* path, akz_match* out, uint64_t cap, uint64_t* n_out);
int akz_read_matches(const char* path, akz_match* out, uint64_t cap, uint64_t* n_out) {
    try {
        return read_matches_impl(path, out, cap, n_out);
    } catch (const std::exception& e) {  // bad_alloc / length_error on absurd element counts: nothing throws across the ABI
        set_error(std::string("read_matches: ") + e.what());
        return AKZ_ERR_NO_MEMORY;
    }
}
static int read_matches_impl(const char* path, akz_match* out, uint64_t cap, uint64_t* n_out) {
    if (!path || !n_out) return AKZ_ERR_INVALID_ARG;
    std::string s;
    AKZ_TRY(read_file(path, s));
    std::vector<akz_match> v;
    if (is_json(path)) {
        Json j(s);
        j.expect('[');
        if (!j.eat(']')) {
            do {
                akz_match m{0, 0, 0.0};
                j.expect('{');
                do {
                    const std::string k = j.key();
                    if (k == "index_0") m.index_0 = j.uint(UINT64_MAX);
                    else if (k == "index_1") m.index_1 = j.uint(UINT64_MAX);
                    else if (k == "distance") m.distance = j.number();
                    else j.skip();
                } while (j.ok && j.eat(','));
                j.expect('}');
                v.push_back(m);
            } while (j.ok && j.eat(','));
            j.expect(']');
        }
        if (!j.ok) {
            set_error("malformed JSON matches file");
            return AKZ_ERR_INVALID_ARG;
        }
    } else {
        size_t pos = 0;
        uint64_t n = 0;
        if (!get(s, pos, n) || n > s.size() || n * 24 + 8 != s.size()) {
            set_error("malformed bincode matches file");
            return AKZ_ERR_INVALID_ARG;
        }
        v.resize((size_t)n);
        for (auto& m : v) { get(s, pos, m.index_0); get(s, pos, m.index_1); get(s, pos, m.distance); }
    }
    *n_out = v.size();
    if (out) {
        if (cap < v.size()) return AKZ_ERR_BUFFER;
        if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(akz_match));
    }
    return AKZ_OK;
}

// serde_json::to_string(&Config) / from_str (akaze-util/src/bin/extract_features.rs:66-83); field order and
// names of types::evolution::Config (evolution.rs:8-38)
int akz_config_to_json(const akz_config* cfg, char* buf, uint64_t cap, uint64_t* len) {
    if (!cfg || !len) return AKZ_ERR_INVALID_ARG;
    std::string s = "{\"num_sublevels\":" + std::to_string(cfg->num_sublevels) +
                    ",\"max_octave_evolution\":" + std::to_string(cfg->max_octave_evolution) + ",\"base_scale_offset\":";
    num(s, cfg->base_scale_offset);
    s += ",\"initial_contrast\":";
    num(s, cfg->initial_contrast);
    s += ",\"contrast_percentile\":";
    num(s, cfg->contrast_percentile);
    s += ",\"contrast_factor_num_bins\":" + std::to_string(cfg->contrast_factor_num_bins) + ",\"derivative_factor\":";
    num(s, cfg->derivative_factor);
    s += ",\"detector_threshold\":";
    num(s, cfg->detector_threshold);
    s += ",\"descriptor_channels\":" + std::to_string(cfg->descriptor_channels) +
         ",\"descriptor_pattern_size\":" + std::to_string(cfg->descriptor_pattern_size) + "}";
    *len = s.size();
    if (!buf || cap < s.size() + 1) return AKZ_ERR_BUFFER;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return AKZ_OK;
}

static int config_from_json_impl(const char* json, akz_config* cfg);
int akz_config_from_json(const char* json, akz_config* cfg) {
    try {
        akz_config tmp;
        if (!json || !cfg) return AKZ_ERR_INVALID_ARG;
        tmp = *cfg;
        const int st = config_from_json_impl(json, &tmp);  // *cfg is untouched unless the whole text parsed
        if (st == AKZ_OK) *cfg = tmp;
        return st;
    } catch (const std::exception& e) {
        set_error(std::string("config_from_json: ") + e.what());
        return AKZ_ERR_NO_MEMORY;
    }
}
static int config_from_json_impl(const char* json, akz_config* cfg) {
    if (!json || !cfg) return AKZ_ERR_INVALID_ARG;
    const std::string s(json);
    Json j(s);
    j.expect('{');
    if (!j.eat('}')) {
        do {
            const std::string k = j.key();
            if (k == "num_sublevels") cfg->num_sublevels = (uint32_t)j.uint(UINT32_MAX);
            else if (k == "max_octave_evolution") cfg->max_octave_evolution = (uint32_t)j.uint(UINT32_MAX);
            else if (k == "base_scale_offset") cfg->base_scale_offset = j.number();
            else if (k == "initial_contrast") cfg->initial_contrast = j.number();
            else if (k == "contrast_percentile") cfg->contrast_percentile = j.number();
            else if (k == "contrast_factor_num_bins") cfg->contrast_factor_num_bins = j.uint(UINT64_MAX);
            else if (k == "derivative_factor") cfg->derivative_factor = j.number();
            else if (k == "detector_threshold") cfg->detector_threshold = j.number();
            else if (k == "descriptor_channels") cfg->descriptor_channels = j.uint(UINT64_MAX);
            else if (k == "descriptor_pattern_size") cfg->descriptor_pattern_size = j.uint(UINT64_MAX);
            else j.skip();
        } while (j.ok && j.eat(','));
        j.expect('}');
    }
    if (!j.ok) {
        set_error("akz_config_from_json: malformed JSON");
        return AKZ_ERR_INVALID_ARG;
    }
    return AKZ_OK;
}

}  // extern "C"
