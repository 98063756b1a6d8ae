// ThreadSanitizer driver for akz_remove_outliers (akz_ransac.cpp): the trials of a call run on a process-wide pool of host
// threads; callers that arrive while it is busy start threads of their own.  Four caller threads, each with its own
// random source seeded alike, must all get the result of a lone call.  Built and run by tests/test_fuzz_host.py with
// -fsanitize=thread; the two symbols akz_ransac.cpp takes from akz_api.cpp are stubbed here.
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../akaze-rust_amd/csrc/akz_ransac.cpp"

namespace akz {
void set_error(const std::string&) {}
unsigned host_cpu_share() { return 6; }
}  // namespace akz

int main() {
    const int n = 600;
    std::vector<akz_keypoint> k0(n), k1(n);
    std::vector<akz_match> m(n);
    unsigned s = 9;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((s >> 8) % 100000) / 100.0f; };
    for (int i = 0; i < n; ++i) {
        std::memset(&k0[i], 0, sizeof(akz_keypoint)); std::memset(&k1[i], 0, sizeof(akz_keypoint));
        k0[i].x = rnd() * 3.0f; k0[i].y = rnd() * 2.0f;
        k1[i].x = k0[i].x + 17.0f + (i % 7 == 0 ? rnd() : 0.01f * (float)(i % 5));
        k1[i].y = k0[i].y + 9.0f + 0.01f * (float)(i % 3);
        m[i].index_0 = (uint64_t)i; m[i].index_1 = (uint64_t)i; m[i].distance = 1.0;
    }
    auto call = [&](std::vector<akz_match>& out) {
        akz_random_seed(7, 11);
        out.assign(n, akz_match{});
        uint64_t cnt = 0;
        if (akz_remove_outliers(k0.data(), n, k1.data(), n, m.data(), n, 300, 0.05f, 0.5f, out.data(), &cnt) != AKZ_OK) cnt = 0;
        out.resize((size_t)cnt);
    };
    std::vector<akz_match> alone;
    call(alone);
    bool ok = alone.size() >= 8;
    std::vector<std::vector<akz_match>> got(4);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
        th.emplace_back([&, t] {
            for (int r = 0; r < 6; ++r) call(got[(size_t)t]);
        });
    for (auto& t : th) t.join();
    for (auto& g : got) ok = ok && g.size() == alone.size() && std::memcmp(g.data(), alone.data(), alone.size() * sizeof(akz_match)) == 0;
    printf(ok ? "ransac ok (%zu of %d kept)\n" : "ransac FAILED (%zu of %d)\n", alone.size(), n);
    return ok ? 0 : 1;
}
