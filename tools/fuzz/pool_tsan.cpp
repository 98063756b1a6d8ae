// ThreadSanitizer driver for the host worker pool (akz_pool.hpp): many short runs of varying size from one caller
// thread, each checked for "every index exactly once", pools created and destroyed while idle and right after a
// run, and two pools driven by two caller threads at the same time (two contexts).  Built and run by
// tests/test_fuzz_host.py with -fsanitize=thread.
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "../../akaze-rust_amd/csrc/akz_pool.hpp"

static bool exercise(unsigned workers, int runs, unsigned seed) {
    akz::WorkerPool pool(workers);
    unsigned s = seed;
    for (int r = 0; r < runs; ++r) {
        s = s * 1664525u + 1013904223u;
        const size_t count = (s >> 8) % 97;  // including 0 and 1
        std::vector<int> hits(count, 0);
        std::atomic<long> sum{0};
        pool.run(count, [&](size_t i) {
            hits[i] += 1;  // a second call with the same index would be a data race and a wrong count
            sum += (long)i;
        });
        for (size_t i = 0; i < count; ++i)
            if (hits[i] != 1) return false;
        if (sum.load() != (long)(count * (count - (count ? 1 : 0)) / 2)) return false;
    }
    return true;
}

int main(int argc, char** argv) {
    const int runs = argc > 1 ? atoi(argv[1]) : 2000;
    bool ok = true;
    for (unsigned w : {0u, 1u, 3u, 7u}) ok = ok && exercise(w, runs, 17u + w);
    for (int k = 0; k < 50; ++k) ok = ok && exercise(4, 3, 1000u + (unsigned)k);  // short-lived pools
    bool ok_a = false, ok_b = false;
    std::thread a([&] { ok_a = exercise(5, runs, 5u); }), b([&] { ok_b = exercise(5, runs, 6u); });
    a.join();
    b.join();
    ok = ok && ok_a && ok_b;
    printf(ok ? "pool ok\n" : "pool FAILED\n");
    return ok ? 0 : 1;
}
