// The descriptor exchange of the multi-GPU path (akz_comm_* / akz_gather_*), driven through the C ABI only — what a
// C++ or Rust host does before a cross-image brute-force match.
//
//   gather_selftest                       one rank on device 0
//   gather_selftest RANK NRANKS ID_FILE [DEVICE]
//                                         rank RANK of NRANKS, one process per GPU (device = RANK unless DEVICE is given: the
//                                         one-GPU rehearsal of tests/test_comm_faults.py, whose stand-in for librccl carries
//                                         the blocks between processes); rank 0 writes the 128-byte communicator id to
//                                         ID_FILE, the others wait for it
//
// Every rank extracts frames of its shard (image i -> GPU i mod G), gathers the descriptor rows with the synchronous
// form and with the pipelined form, and checks that (a) its own rows came back bit for bit in its slot of both
// results, (b) the two forms agree, (c) the headers carry the row and image counts.  Prints "gather selftest ok".
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/akaze_hip.h"
#include "../../include/akaze_hip_debug.h"

// An exchange that timed out (AKZ_ERR_TIMEOUT: a peer is missing or hung) leaves a collective on the communicator's stream
// that may never complete.  The host reports it, may still release its objects (akz_gather_free / akz_comm_destroy return
// at once on an abandoned communicator), and ends the process WITHOUT the runtime's exit handlers, which could wait for
// that stream: _exit, non-zero.
static akz_comm* g_comm = nullptr;
#define TRY(expr)                                                                          \
    do {                                                                                   \
        const int st_ = (expr);                                                            \
        if (st_ != AKZ_OK) {                                                               \
            fprintf(stderr, "FAILED: %s: %s\n", #expr, akz_last_error());                  \
            if (st_ == AKZ_ERR_TIMEOUT) {                                                  \
                akz_comm_destroy(g_comm);                                                  \
                fprintf(stderr, "communicator abandoned after a timeout: exiting with status 3\n"); \
                fflush(stderr);                                                            \
                _exit(3);                                                                  \
            }                                                                              \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)
#define CHECK(cond)                                                \
    do {                                                           \
        if (!(cond)) {                                             \
            fprintf(stderr, "CHECK FAILED: %s (line %d)\n", #cond, __LINE__); \
            return 1;                                              \
        }                                                          \
    } while (0)

int main(int argc, char** argv) {
    int rank = 0, nranks = 1;
    const char* id_file = nullptr;
    int device_arg = -1;
    if (argc == 4 || argc == 5) {
        rank = atoi(argv[1]);
        nranks = atoi(argv[2]);
        id_file = argv[3];
        if (argc == 5) device_arg = atoi(argv[4]);
    } else if (argc != 1) {
        fprintf(stderr, "usage: %s [RANK NRANKS ID_FILE [DEVICE]]\n", argv[0]);
        return 2;
    }
    const int device = device_arg >= 0 ? device_arg : rank;
    void* stream = nullptr;
    TRY(akz_stream_create(device, &stream));
    akz_ctx* ctx = nullptr;
    TRY(akz_ctx_create(device, stream, &ctx));

    uint8_t id[AKZ_COMM_ID_BYTES];
    if (rank == 0) {
        TRY(akz_comm_unique_id(id));
        if (id_file) {
            std::string tmp = std::string(id_file) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            CHECK(f && fwrite(id, 1, sizeof(id), f) == sizeof(id));
            fclose(f);
            CHECK(rename(tmp.c_str(), id_file) == 0);
        }
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); ++tries) usleep(100000);
        CHECK(f && fread(id, 1, sizeof(id), f) == sizeof(id));
        fclose(f);
    }
    akz_comm* comm = nullptr;
    TRY(akz_comm_create(device, id, rank, nranks, &comm));
    g_comm = comm;
    if (const char* t = getenv("AKZ_SELFTEST_TIMEOUT_S")) TRY(akz_comm_set_timeout(comm, atof(t)));
    int r2 = -1, n2 = -1;
    TRY(akz_comm_info(comm, &r2, &n2));
    CHECK(r2 == rank && n2 == nranks);

    // this rank's shard: two frames of a 4-frame-per-rank job
    const uint32_t W = 320, H = 240, NIMG = 2;
    akz_config cfg;
    akz_config_default(&cfg);
    std::vector<uint8_t> frames((size_t)W * H * NIMG);
    for (uint32_t i = 0; i < NIMG; ++i) TRY(akz_synth_frame_u8(frames.data() + (size_t)i * W * H, W, H, (uint64_t)rank + (uint64_t)i * nranks, 0, 0));
    void* d_frames = nullptr;
    TRY(akz_device_malloc(ctx, frames.size(), &d_frames));
    TRY(akz_memcpy_h2d(ctx, d_frames, frames.data(), frames.size()));
    akz_result* res = nullptr;
    TRY(akz_extract_device_u8(ctx, (const uint8_t*)d_frames, W, H, NIMG, &cfg, 0, &res));
    uint64_t rows = 0;
    const uint8_t* d_rows = nullptr;
    for (uint32_t i = 0; i < NIMG; ++i) {
        const uint8_t* p = nullptr;
        uint64_t n = 0;
        TRY(akz_result_device_descriptors(res, i, &p, &n));
        if (i == 0) d_rows = p;
        rows += n;
    }
    CHECK(rows > 0);
    std::vector<uint8_t> mine((size_t)rows * 64);
    TRY(akz_memcpy_d2h(ctx, mine.data(), d_rows, mine.size()));

    // (1) Appendix C form
    std::vector<uint64_t> counts((size_t)nranks, 0);
    const uint8_t* d_all = nullptr;
    TRY(akz_gather_descriptors(comm, d_rows, rows, &d_all, counts.data()));
    CHECK(counts[(size_t)rank] == rows);
    uint64_t total = 0, my_off = 0;
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) my_off = total;
        total += counts[(size_t)r];
    }
    std::vector<uint8_t> all((size_t)total * 64);
    TRY(akz_memcpy_d2h(ctx, all.data(), d_all, all.size()));
    CHECK(memcmp(all.data() + my_off * 64, mine.data(), mine.size()) == 0);

    // (2) pipelined form, twice in flight order (begin, begin is not allowed to need the first one finished)
    uint64_t cap = 1;
    for (uint64_t v : counts) cap = v > cap ? v : cap;
    cap += cap / 2;
    const akz_result* rs[1] = {res};
    akz_gather *g1 = nullptr, *g2 = nullptr;
    TRY(akz_gather_begin(comm, rs, 1, cap, &g1));
    TRY(akz_gather_begin_rows(comm, d_rows, rows, cap, nullptr, &g2));
    for (akz_gather* g : {g1, g2}) {
        const uint8_t* blocks = nullptr;
        uint64_t block_rows = 0;
        std::vector<uint64_t> c2((size_t)nranks, 0), im((size_t)nranks, 0);
        TRY(akz_gather_stream_wait(g, stream));
        TRY(akz_gather_finish(g, &blocks, &block_rows, c2.data(), im.data()));
        CHECK(block_rows == cap + 1);
        for (int r = 0; r < nranks; ++r) CHECK(c2[(size_t)r] == counts[(size_t)r]);
        CHECK(im[(size_t)rank] == (g == g1 ? NIMG : 1));
        uint64_t off = 0;
        for (int r = 0; r < nranks; ++r) {
            std::vector<uint8_t> blk((size_t)counts[(size_t)r] * 64);
            if (!blk.empty()) TRY(akz_memcpy_d2h(ctx, blk.data(), blocks + ((uint64_t)r * block_rows + 1) * 64, blk.size()));
            CHECK(memcmp(blk.data(), all.data() + off * 64, blk.size()) == 0);
            off += counts[(size_t)r];
        }
        TRY(akz_gather_free(g));
    }
    // (3) a shard that does not fit the agreed capacity: the rank still takes part in the collective (header only), EVERY
    // rank gets AKZ_ERR_BUFFER from finish together with the counts, and the communicator stays usable
    uint64_t largest = 0;
    for (uint64_t v : counts) largest = v > largest ? v : largest;
    if (largest > 0) {
        akz_gather* g3 = nullptr;
        TRY(akz_gather_begin_rows(comm, d_rows, rows, largest - 1, nullptr, &g3));
        std::vector<uint64_t> c3((size_t)nranks, 0);
        CHECK(akz_gather_finish(g3, nullptr, nullptr, c3.data(), nullptr) == AKZ_ERR_BUFFER);
        for (int r = 0; r < nranks; ++r) CHECK(c3[(size_t)r] == counts[(size_t)r]);
        TRY(akz_gather_free(g3));
        akz_gather* g4 = nullptr;
        TRY(akz_gather_begin_rows(comm, d_rows, rows, largest, nullptr, &g4));
        TRY(akz_gather_finish(g4, nullptr, nullptr, c3.data(), nullptr));
        TRY(akz_gather_free(g4));
    }

    // (4) BASELINE configs[4] through the C ABI: all-pairs match over a gather (every unordered pair once, both directions).
    {
        akz_gather* g5 = nullptr;
        TRY(akz_gather_begin(comm, rs, 1, cap, &g5));
        akz_pairs* pairs = nullptr;
        TRY(akz_match_all_pairs(ctx, g5, 10000, 0.86, &pairs));
        uint64_t n_images = 0, first = 0, owned = 0;
        TRY(akz_pairs_info(pairs, &n_images, &first, &owned));
        CHECK(n_images == (uint64_t)NIMG * nranks && owned == NIMG && first == (uint64_t)rank * NIMG);
        // per-image rows of this rank as the gather reports them == the result's own counts
        uint64_t tab[8] = {0}, tab_n = 0;
        TRY(akz_gather_image_rows(g5, rank, tab, 8, &tab_n));
        CHECK(tab_n == NIMG);
        std::vector<std::vector<uint8_t>> img_rows((size_t)n_images);
        uint64_t off = 0;
        for (uint64_t j = 0; j < n_images; ++j) {
            uint64_t nr = 0;
            int owner = -1;
            TRY(akz_pairs_image_rows(pairs, j, &nr, &owner));
            CHECK(owner == (int)(j / NIMG));
            if (owner == rank) {
                uint64_t nk = 0;
                TRY(akz_result_counts(res, j - first, nullptr, &nk, nullptr));
                CHECK(nk == nr && tab[j - first] == nr);
            }
            img_rows[(size_t)j].assign(all.begin() + (long)(off * 64), all.begin() + (long)((off + nr) * 64));  // `all`: rank-major rows of (1)
            off += nr;
        }
        CHECK(off == total);
        // every ORDERED pair of the job is held by exactly one rank (the owner of the pair's lead image), both directions
        // together; what this rank holds must equal akz_descriptor_match of the two images' rows
        uint64_t checked = 0, held = 0;
        for (uint64_t q = 0; q < n_images; ++q)
            for (uint64_t j = 0; j < n_images; ++j) {
                uint64_t n = 0;
                if (j == q) {
                    TRY(akz_pairs_matches(pairs, q, j, nullptr, 0, &n));
                    CHECK(n == 0);
                    continue;
                }
                int holder = -1, holder_rev = -2;
                TRY(akz_pairs_holder(pairs, q, j, &holder));
                TRY(akz_pairs_holder(pairs, j, q, &holder_rev));
                CHECK(holder == holder_rev && holder >= 0 && holder < nranks);
                if (holder != rank) {
                    CHECK(akz_pairs_matches(pairs, q, j, nullptr, 0, &n) != AKZ_OK);
                    continue;
                }
                ++held;
                TRY(akz_pairs_matches(pairs, q, j, nullptr, 0, &n));
                std::vector<akz_match> got((size_t)n + 1), exp(img_rows[(size_t)q].size() / 64 + 1);
                TRY(akz_pairs_matches(pairs, q, j, got.data(), n, &n));
                uint64_t ne = 0;
                TRY(akz_descriptor_match(ctx, img_rows[(size_t)q].data(), img_rows[(size_t)q].size() / 64, img_rows[(size_t)j].data(),
                                         img_rows[(size_t)j].size() / 64, 64, 10000, 0.86, exp.data(), &ne));
                CHECK(n == ne && memcmp(got.data(), exp.data(), (size_t)n * sizeof(akz_match)) == 0);
                checked += n;
            }
        CHECK(checked > 0 && (nranks > 1 || held == n_images * (n_images - 1)));
        TRY(akz_pairs_free(pairs));
        TRY(akz_gather_free(g5));
    }

    TRY(akz_result_free(res));
    TRY(akz_device_free(ctx, d_frames));
    TRY(akz_comm_destroy(comm));
    TRY(akz_ctx_destroy(ctx));
    TRY(akz_stream_destroy(device, stream));
    printf("gather selftest ok: rank %d of %d, %llu local rows, %llu gathered\n", rank, nranks, (unsigned long long)rows,
           (unsigned long long)total);
    return 0;
}
