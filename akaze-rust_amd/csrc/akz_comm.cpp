// The path's one exchange step (SURVEY.md 8(e), Appendix C `akz_gather_descriptors`): an RCCL all-gather of the
// 64-byte descriptor rows of every rank's shard, so that a brute-force Hamming match can run on any GPU of the job.
// The reference has no counterpart (it is a single-process CPU crate); this is the MI355X-native addition that the
// one-image-per-GPU sharding needs.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already carries an RCCL — PyTorch-ROCm ships its
// own copy under the same soname — gets that one, a plain C++ / Rust host gets /opt/rocm's.  libakaze_hip.so itself
// has no link-time dependency on it, so single-GPU users never load a communication library.
//
// Wire format: ONE all-gather per exchange.  Every rank contributes a block of (1 + cap_rows) 64-byte rows: row 0 is a
// header {u64 rows, u64 images, u64 cap_rows, u64 sequence, u64 overflow, u64 table rows}, rows 1.. are its descriptor
// rows, followed by the table of rows per image (eight u64 per row; it counts against cap_rows), the rest is padding.  A rank whose shard does not fit cap_rows STILL takes part in the collective: it sends the header alone with
// `overflow` set and `rows` = what it needed, and akz_gather_finish reports AKZ_ERR_BUFFER on EVERY rank (with the
// counts filled in, so that all ranks can agree on a larger capacity and repeat the exchange).  What no rank can
// detect before the collective is a DIFFERENT cap_rows on different ranks -- the message sizes then differ, which RCCL
// does not survive: cap_rows is part of the job's contract, and the header check in finish is best effort only.
// A few MB per rank: latency-bound over xGMI, so one fixed-size collective beats exact-size send/recv pairs.
//
// Nothing here synchronises the extraction stream: the local rows are copied on the communicator's copy stream, the
// collective is enqueued behind the copy on its exchange stream, and the caller retires the gather (or makes its
// matcher's stream wait for it) whenever it wants — typically one step later.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "akz_internal.hpp"

using namespace akz;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("librccl.so.1 cannot be loaded: ") + (dlerror() ? dlerror() : "unknown");
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

int rccl_ready() {
    Rccl* r = rccl();
    if (!r->error.empty() || !r->handle) {
        set_error(r->error.empty() ? "RCCL is not available" : r->error);
        return AKZ_ERR_UNSUPPORTED;
    }
    return AKZ_OK;
}

#define AKZ_NCCL_TRY(expr)                                                                        \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess) {                                                                  \
            set_error(std::string(#expr) + " failed: " + rccl()->GetErrorString(_r));             \
            return AKZ_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

constexpr uint64_t kRow = 64;

}  // namespace

struct akz_gather {
    akz_comm* comm = nullptr;
    uint8_t* send = nullptr;   // (1 + cap_rows) rows
    uint8_t* recv = nullptr;   // nranks * (1 + cap_rows) rows
    uint64_t cap_rows = 0;
    size_t send_bytes = 0, recv_bytes = 0;
    hipEvent_t done = nullptr;
    uint64_t* pinned = nullptr;  // header staging: 8 u64 per rank
    bool in_use = false;
    bool overflow = false;     // this rank's shard did not fit: it sent its header only
    bool finished = false;     // akz_gather_finish has read the headers and the per-image tables
    bool delivered = false;    // external transport: akz_gather_deliver has been called (`done` is recorded)
    bool timed_out = false;    // akz_gather_finish gave up on it (the communicator is abandoned while any such gather is unresolved)
    std::vector<uint64_t> hdr_rows, hdr_images;      // per rank, from the headers
    std::vector<std::vector<uint64_t>> image_rows;   // per rank: rows of every image of its shard
    std::vector<uint64_t> table;                     // this rank's per-image table, staged for the send block
    hipEvent_t readers = nullptr;                    // akz_match_all_pairs: its copies out of recv have run (another stream)
    bool readers_pending = false;
};

static void pairs_destroy(akz_pairs* p);  // (defined with akz_pairs, below)
static void delete_pairs_host_only(akz_pairs* p);
static void pairs_orphan_all(akz_comm* c);
struct akz_comm {
    int device = 0, rank = 0, nranks = 1;
    ncclComm_t nccl = nullptr;
    bool external = false;          // akz_comm_create_external: the CALLER moves the blocks between ranks (no RCCL in the process)
    bool host = false;              // ... with device AKZ_COMM_HOST: the blocks (and the caller's rows) live in HOST memory and no
                                    // GPU call is made -- the wire format, the overflow protocol and the all-pairs plan under a
                                    // CPU-only process group (tests/test_distributed.py)
    double timeout_s = 0.0;         // akz_comm_set_timeout: how long akz_gather_finish waits for an exchange (0: without limit)
    bool abandoned = false;         // an exchange timed out: the collective may never complete.  Nothing waits for the
                                    // communicator's streams or events any more (akz_gather_free, akz_comm_destroy return at
                                    // once and LEAK the device objects: hipFree / hipStreamDestroy / ncclCommDestroy would
                                    // wait for the stuck collective), no new exchange is accepted; the process is expected to end
    hipStream_t xs = nullptr;       // exchange stream: the collectives, in order
    hipStream_t cs = nullptr;       // copy stream: local rows -> send block, headers -> host (never behind a collective)
    hipEvent_t ready = nullptr;     // producer-side event the copy stream waits for
    hipEvent_t copied = nullptr;    // the send block is complete
    uint8_t* sync_out = nullptr;    // compacted rows of the last akz_gather_descriptors call (Appendix C form)
    size_t sync_out_bytes = 0;
    uint64_t sequence = 0;
    std::vector<akz_gather*> pool;  // every gather object ever handed out (reused when free and large enough)
    std::vector<akz_pairs*> pairs_pool;  // freed all-pairs results: device block, pinned counts and event are reused (at most kPairsPool)
    std::vector<akz_pairs*> live_pairs;  // handed out and not yet freed: akz_comm_destroy orphans them (their comm pointer is cleared)
    static constexpr size_t kPairsPool = 2;  // two steps' objects alternate in a pipelined job; more would only hold device memory
};

static void gather_release_buffers(akz_gather* g) {
    if (g->comm && g->comm->host) {
        std::free(g->send);
        std::free(g->recv);
        std::free(g->pinned);
        g->send = g->recv = nullptr;
        g->pinned = nullptr;
        g->cap_rows = ~0ull;
        return;
    }
    if (g->send) (void)hipFree(g->send);
    if (g->recv) (void)hipFree(g->recv);
    if (g->pinned) (void)hipHostFree(g->pinned);
    if (g->done) (void)hipEventDestroy(g->done);
    if (g->readers) (void)hipEventDestroy(g->readers);
    g->readers = nullptr;
    g->readers_pending = false;
    g->send = g->recv = nullptr;
    g->pinned = nullptr;
    g->done = nullptr;
}

static int gather_acquire(akz_comm* c, uint64_t cap_rows, akz_gather** out) {
    if (c->abandoned) {
        set_error("gather: an earlier exchange on this communicator timed out (a peer is missing or hung); it accepts no further work");
        return AKZ_ERR_TIMEOUT;
    }
    akz_gather* g = nullptr;
    for (akz_gather* p : c->pool)
        if (!p->in_use && p->send && p->cap_rows == cap_rows) {
            g = p;
            break;
        }
    if (!g) {
        for (akz_gather* p : c->pool)
            if (!p->in_use) {  // a free object of another capacity (or one whose allocation failed earlier): rebuild it
                gather_release_buffers(p);
                g = p;
                break;
            }
        if (!g) {
            g = new akz_gather;
            g->comm = c;
            c->pool.push_back(g);
        }
        // the object matches a capacity only once every buffer exists: a failure part-way leaves it free and empty
        g->cap_rows = ~0ull;
        const size_t send_bytes = (size_t)(1 + cap_rows) * kRow, recv_bytes = send_bytes * (size_t)c->nranks;
        bool ok;
        if (c->host) {
            g->send = (uint8_t*)std::calloc(1, send_bytes);  // no uninitialised bytes on the wire
            g->recv = (uint8_t*)std::malloc(recv_bytes);
            g->pinned = (uint64_t*)std::malloc((size_t)c->nranks * kRow + kRow);
            ok = g->send && g->recv && g->pinned;
        } else {
            ok = hipMalloc((void**)&g->send, send_bytes) == hipSuccess && hipMalloc((void**)&g->recv, recv_bytes) == hipSuccess &&
                 hipMemsetAsync(g->send, 0, send_bytes, c->cs) == hipSuccess &&  // no uninitialised bytes on the wire
                 hipHostMalloc((void**)&g->pinned, (size_t)c->nranks * kRow + kRow, hipHostMallocDefault) == hipSuccess &&
                 hipEventCreateWithFlags(&g->done, hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            gather_release_buffers(g);
            set_error("gather: allocating the exchange buffers failed");
            return AKZ_ERR_HIP;
        }
        g->cap_rows = cap_rows;
        g->send_bytes = send_bytes;
        g->recv_bytes = recv_bytes;
    }
    g->in_use = true;
    g->overflow = false;
    g->finished = false;
    g->delivered = false;
    g->timed_out = false;
    *out = g;
    return AKZ_OK;
}

// rows of `n_src` device blocks -> send block (copy stream), then one all-gather (exchange stream).  With wait_copy the
// call returns once the local rows have been copied (the sources may then be released); it never waits for a collective.
static int gather_enqueue(akz_comm* c, akz_gather* g, const uint8_t* const* d_src, const uint64_t* src_rows, uint64_t n_src,
                          uint64_t images, hipStream_t producer, bool wait_copy, const std::vector<uint64_t>* per_image = nullptr) {
    uint64_t rows = 0;
    for (uint64_t i = 0; i < n_src; ++i) rows += src_rows[i];
    // rows per image travel behind the descriptor rows, eight per 64-byte row (raw rows of akz_gather_begin_rows are one
    // image and need no table)
    g->table.clear();
    if (per_image) g->table = *per_image;
    const uint64_t table_rows = (g->table.size() + 7) / 8;
    g->table.resize(table_rows * 8, 0);
    // a shard that does not fit still takes part (header only, marked): a rank that skipped the collective would leave
    // every other rank waiting in it, and every later collective of the communicator mismatched
    g->overflow = rows + table_rows > g->cap_rows;
    if (c->host) {  // the same block, assembled with memcpy: header row, descriptor rows, per-image table
        uint64_t hdr[8] = {rows, images, g->cap_rows, ++c->sequence, g->overflow ? 1ull : 0ull, g->overflow ? 0ull : table_rows, 0, 0};
        std::memcpy(g->send, hdr, kRow);
        uint64_t at = 1;
        for (uint64_t i = 0; i < n_src && !g->overflow; ++i) {
            if (src_rows[i]) std::memcpy(g->send + at * kRow, d_src[i], src_rows[i] * kRow);
            at += src_rows[i];
        }
        if (!g->overflow && table_rows) std::memcpy(g->send + at * kRow, g->table.data(), table_rows * kRow);
        return AKZ_OK;
    }
    if (g->readers_pending) {  // an all-pairs match of the step before still copies out of these buffers on its own stream
        AKZ_HIP_TRY(hipStreamWaitEvent(c->cs, g->readers, 0));
        g->readers_pending = false;
    }
    if (producer) {  // the rows are complete in the order of this stream
        AKZ_HIP_TRY(hipEventRecord(c->ready, producer));
        AKZ_HIP_TRY(hipStreamWaitEvent(c->cs, c->ready, 0));
    }
    uint64_t* hdr = g->pinned + (size_t)c->nranks * 8;  // this rank's header, staged in pinned memory
    hdr[0] = rows;
    hdr[1] = images;
    hdr[2] = g->cap_rows;
    hdr[3] = ++c->sequence;
    hdr[4] = g->overflow ? 1 : 0;
    hdr[5] = g->overflow ? 0 : table_rows;
    hdr[6] = hdr[7] = 0;
    AKZ_HIP_TRY(hipMemcpyAsync(g->send, hdr, kRow, hipMemcpyHostToDevice, c->cs));
    uint64_t at = 1;
    for (uint64_t i = 0; i < n_src && !g->overflow; ++i) {
        if (src_rows[i] == 0) continue;
        AKZ_HIP_TRY(hipMemcpyAsync(g->send + at * kRow, d_src[i], src_rows[i] * kRow, hipMemcpyDeviceToDevice, c->cs));
        at += src_rows[i];
    }
    if (!g->overflow && table_rows)  // (pageable source: the runtime stages it before the call returns)
        AKZ_HIP_TRY(hipMemcpyAsync(g->send + at * kRow, g->table.data(), table_rows * kRow, hipMemcpyHostToDevice, c->cs));
    AKZ_HIP_TRY(hipEventRecord(c->copied, c->cs));
    if (c->external) {
        // the caller carries the blocks (akz_gather_blocks / akz_gather_deliver): the send block must be complete on return
        AKZ_HIP_TRY(hipEventSynchronize(c->copied));
        return AKZ_OK;
    }
    AKZ_HIP_TRY(hipStreamWaitEvent(c->xs, c->copied, 0));
    AKZ_NCCL_TRY(rccl()->AllGather(g->send, g->recv, g->send_bytes, ncclUint8, c->nccl, c->xs));
    AKZ_HIP_TRY(hipEventRecord(g->done, c->xs));
    if (wait_copy) AKZ_HIP_TRY(hipEventSynchronize(c->copied));
    return AKZ_OK;
}

extern "C" {

int akz_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return AKZ_ERR_INVALID_ARG;
    AKZ_TRY(rccl_ready());
    ncclUniqueId id;
    AKZ_NCCL_TRY(rccl()->GetUniqueId(&id));
    static_assert(sizeof(id) == AKZ_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id_out, &id, sizeof(id));
    return AKZ_OK;
}

int akz_comm_create(int device, const uint8_t* id, int rank, int nranks, akz_comm** out) {
    if (!out) return AKZ_ERR_INVALID_ARG;
    *out = nullptr;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("akz_comm_create: bad rank / nranks / id");
        return AKZ_ERR_INVALID_ARG;
    }
    AKZ_TRY(rccl_ready());
    AKZ_HIP_TRY(hipSetDevice(device));
    akz_comm* c = new akz_comm;
    c->device = device;
    c->rank = rank;
    c->nranks = nranks;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclResult_t st = rccl()->CommInitRank(&c->nccl, nranks, uid, rank);
    if (st != ncclSuccess) {
        set_error(std::string("ncclCommInitRank failed: ") + rccl()->GetErrorString(st));
        delete c;
        return AKZ_ERR_HIP;
    }
    if (hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->copied, hipEventDisableTiming) != hipSuccess) {
        set_error("akz_comm_create: stream / event creation failed");
        (void)rccl()->CommDestroy(c->nccl);
        if (c->xs) (void)hipStreamDestroy(c->xs);
        if (c->cs) (void)hipStreamDestroy(c->cs);
        if (c->ready) (void)hipEventDestroy(c->ready);
        if (c->copied) (void)hipEventDestroy(c->copied);
        delete c;
        return AKZ_ERR_HIP;
    }
    *out = c;
    return AKZ_OK;
}

// A communicator whose blocks the CALLER moves between the ranks (MPI, gloo, shared memory, a test that runs several
// ranks on one GPU -- which RCCL refuses): same wire format, same gather objects, same all-pairs match; RCCL is not loaded.
int akz_comm_create_external(int device, int rank, int nranks, akz_comm** out) {
    if (!out) return AKZ_ERR_INVALID_ARG;
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("akz_comm_create_external: bad rank / nranks");
        return AKZ_ERR_INVALID_ARG;
    }
    if (device == AKZ_COMM_HOST) {  // blocks in host memory: no device, no stream, no event
        akz_comm* c = new akz_comm;
        c->device = device;
        c->rank = rank;
        c->nranks = nranks;
        c->external = c->host = true;
        *out = c;
        return AKZ_OK;
    }
    AKZ_HIP_TRY(hipSetDevice(device));
    akz_comm* c = new akz_comm;
    c->device = device;
    c->rank = rank;
    c->nranks = nranks;
    c->external = true;
    if (hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->copied, hipEventDisableTiming) != hipSuccess) {
        set_error("akz_comm_create_external: stream / event creation failed");
        if (c->xs) (void)hipStreamDestroy(c->xs);
        if (c->cs) (void)hipStreamDestroy(c->cs);
        if (c->ready) (void)hipEventDestroy(c->ready);
        if (c->copied) (void)hipEventDestroy(c->copied);
        delete c;
        return AKZ_ERR_HIP;
    }
    *out = c;
    return AKZ_OK;
}

int akz_gather_blocks(akz_gather* g, const uint8_t** d_send, uint8_t** d_recv, uint64_t* block_bytes) {
    if (!g || !g->in_use || !g->comm->external) {
        set_error("akz_gather_blocks: a gather in flight on a communicator of akz_comm_create_external");
        return AKZ_ERR_INVALID_ARG;
    }
    if (d_send) *d_send = g->send;
    if (d_recv) *d_recv = g->recv;
    if (block_bytes) *block_bytes = g->send_bytes;
    return AKZ_OK;
}
int akz_gather_deliver(akz_gather* g, void* stream) {
    if (!g || !g->in_use || !g->comm->external) {
        set_error("akz_gather_deliver: a gather in flight on a communicator of akz_comm_create_external");
        return AKZ_ERR_INVALID_ARG;
    }
    akz_comm* c = g->comm;
    if (c->host) {
        g->delivered = true;  // (host blocks are complete when the caller says so)
        return AKZ_OK;
    }
    AKZ_HIP_TRY(hipSetDevice(c->device));
    AKZ_HIP_TRY(hipEventRecord(g->done, stream ? (hipStream_t)stream : c->xs));  // (NULL: the blocks are complete now)
    g->delivered = true;
    return AKZ_OK;
}

int akz_comm_destroy(akz_comm* c) {
    if (!c) return AKZ_OK;
    if (c->abandoned) {
        // A collective that will never complete sits on c->xs.  Every call that would wait for it -- synchronising or
        // destroying the streams, hipFree (synchronises the device), ncclCommDestroy -- is left out: the device objects
        // leak, the host objects go, and the caller can end the process with an error instead of hanging in teardown.
        for (akz_gather* g : c->pool) delete g;
        c->pool.clear();
        for (akz_pairs* p : c->pairs_pool) delete_pairs_host_only(p);
        c->pairs_pool.clear();
        pairs_orphan_all(c);
        delete c;
        return AKZ_OK;
    }
    if (c->host) {
        for (akz_gather* g : c->pool) {
            gather_release_buffers(g);
            delete g;
        }
        c->pool.clear();
        for (akz_pairs* p : c->pairs_pool) delete_pairs_host_only(p);
        c->pairs_pool.clear();
        pairs_orphan_all(c);
        delete c;
        return AKZ_OK;
    }
    (void)hipSetDevice(c->device);
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    if (c->cs) (void)hipStreamSynchronize(c->cs);
    if (c->sync_out) (void)hipFree(c->sync_out);
    for (akz_gather* g : c->pool) {
        gather_release_buffers(g);
        delete g;
    }
    c->pool.clear();
    for (akz_pairs* p : c->pairs_pool) pairs_destroy(p);
    c->pairs_pool.clear();
    pairs_orphan_all(c);  // results the caller still holds stay readable; their akz_pairs_free no longer touches this object
    if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->copied) (void)hipEventDestroy(c->copied);
    if (c->xs) (void)hipStreamDestroy(c->xs);
    if (c->cs) (void)hipStreamDestroy(c->cs);
    delete c;
    return AKZ_OK;
}

int akz_comm_set_timeout(akz_comm* c, double seconds) {
    if (!c || !(seconds >= 0.0)) return AKZ_ERR_INVALID_ARG;
    c->timeout_s = seconds;
    return AKZ_OK;
}
int akz_comm_place_streams(akz_comm* c, akz_ctx* ctx) {
    if (!c || !ctx) {
        set_error("akz_comm_place_streams: null argument");
        return AKZ_ERR_INVALID_ARG;
    }
    if (c->host) return AKZ_OK;  // (no streams)
    AKZ_HIP_TRY(hipSetDevice(c->device));
    hipStream_t slots[2] = {c->xs, c->cs};
    int shared = 0;
    const int st = akz::place_streams_beside(ctx, slots, 2, &shared);
    c->xs = slots[0];
    c->cs = slots[1];
    return st;
}

int akz_comm_info(const akz_comm* c, int* rank, int* nranks) {
    if (!c) return AKZ_ERR_INVALID_ARG;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return AKZ_OK;
}

int akz_gather_begin_rows(akz_comm* c, const uint8_t* d_local, uint64_t n_local, uint64_t cap_rows, void* producer_stream,
                          akz_gather** out) {
    if (!c || !out || (n_local && !d_local)) {
        set_error("akz_gather_begin_rows: null argument");
        return AKZ_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (!c->host) AKZ_HIP_TRY(hipSetDevice(c->device));
    akz_gather* g = nullptr;
    AKZ_TRY(gather_acquire(c, cap_rows, &g));
    const int st = gather_enqueue(c, g, &d_local, &n_local, 1, 1, (hipStream_t)producer_stream, false);
    if (st != AKZ_OK) {
        g->in_use = false;
        return st;
    }
    *out = g;
    return AKZ_OK;
}

int akz_gather_begin_image_rows(akz_comm* c, const uint8_t* d_local, const uint64_t* rows_per_image, uint64_t n_images,
                                uint64_t cap_rows, void* producer_stream, akz_gather** out) {
    if (!c || !out || (n_images && !rows_per_image)) {
        set_error("akz_gather_begin_image_rows: null argument");
        return AKZ_ERR_INVALID_ARG;
    }
    *out = nullptr;
    std::vector<uint64_t> per_image(rows_per_image, rows_per_image + n_images);
    uint64_t n_local = 0;
    for (uint64_t v : per_image) n_local += v;
    if (n_local && !d_local) {
        set_error("akz_gather_begin_image_rows: rows without a block");
        return AKZ_ERR_INVALID_ARG;
    }
    if (!c->host) AKZ_HIP_TRY(hipSetDevice(c->device));
    akz_gather* g = nullptr;
    AKZ_TRY(gather_acquire(c, cap_rows, &g));
    const int st = gather_enqueue(c, g, &d_local, &n_local, 1, n_images, (hipStream_t)producer_stream, false, &per_image);
    if (st != AKZ_OK) {
        g->in_use = false;
        return st;
    }
    *out = g;
    return AKZ_OK;
}

int akz_gather_begin(akz_comm* c, const akz_result* const* results, uint64_t n_results, uint64_t cap_rows,
                     akz_gather** out) {
    if (!c || !out || (n_results && !results)) {
        set_error("akz_gather_begin: null argument");
        return AKZ_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (c->host) {
        set_error("akz_gather_begin: extraction results live on a device; a host-memory communicator takes akz_gather_begin_image_rows");
        return AKZ_ERR_UNSUPPORTED;
    }
    AKZ_HIP_TRY(hipSetDevice(c->device));
    std::vector<const uint8_t*> src;
    std::vector<uint64_t> rows, per_image;
    uint64_t images = 0;
    for (uint64_t i = 0; i < n_results; ++i) {
        uint64_t n_img = 0;
        AKZ_TRY(akz_result_num_images(results[i], &n_img));
        const uint8_t* base = nullptr;
        uint64_t total = 0;
        for (uint64_t img = 0; img < n_img; ++img) {  // the images' rows are back to back in one device block
            const uint8_t* p = nullptr;
            uint64_t n = 0;
            AKZ_TRY(akz_result_device_descriptors(results[i], img, &p, &n));
            if (img == 0) base = p;
            total += n;
            per_image.push_back(n);
        }
        images += n_img;
        src.push_back(base);
        rows.push_back(base ? total : 0);
    }
    akz_gather* g = nullptr;
    AKZ_TRY(gather_acquire(c, cap_rows, &g));
    // akz_extract_finish returns with the descriptor rows complete, so there is no producer stream to wait for.  The
    // call returns when the local rows have been copied (on the copy stream, which never queues behind a collective):
    // the results may be freed right away.
    int st = gather_enqueue(c, g, src.data(), rows.data(), src.size(), images, nullptr, true, &per_image);
    if (st != AKZ_OK) {
        g->in_use = false;
        return st;
    }
    *out = g;
    return AKZ_OK;
}

int akz_gather_stream_wait(akz_gather* g, void* stream) {
    if (!g || !g->in_use) return AKZ_ERR_INVALID_ARG;
    if (g->comm->external && !g->delivered) {
        set_error("gather: the blocks of this exchange have not been delivered yet (akz_gather_deliver)");
        return AKZ_ERR_INVALID_ARG;
    }
    if (g->comm->host) return AKZ_OK;  // (delivered host blocks are complete: nothing to wait for)
    AKZ_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, g->done, 0));
    return AKZ_OK;
}

int akz_gather_finish(akz_gather* g, const uint8_t** d_all, uint64_t* block_rows, uint64_t* counts, uint64_t* images) {
    if (!g || !g->in_use) {
        set_error("akz_gather_finish: not a gather in flight");
        return AKZ_ERR_INVALID_ARG;
    }
    akz_comm* c = g->comm;
    if (!c->host) AKZ_HIP_TRY(hipSetDevice(c->device));
    if (c->external && !g->delivered) {
        set_error("gather: the blocks of this exchange have not been delivered yet (akz_gather_deliver)");
        return AKZ_ERR_INVALID_ARG;
    }
    // the headers of all blocks -> host (64 bytes per rank): an overflow anywhere is an error everywhere
    if (c->timeout_s > 0.0 && !g->finished && !c->host) {  // a peer that never joined leaves the collective waiting for ever: give up with a message
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipEventQuery(g->done);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) AKZ_HIP_TRY(q);
            (void)hipGetLastError();
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
                set_error("gather: the exchange did not complete within the communicator's timeout (akz_comm_set_timeout): a peer is missing or hung");
                c->abandoned = true;  // from here on nothing waits for this communicator's streams (akz_gather_free, akz_comm_destroy)
                g->timed_out = true;
                return AKZ_ERR_TIMEOUT;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    if (!c->host) AKZ_HIP_TRY(hipEventSynchronize(g->done));
    if (g->timed_out) {
        // the caller asked again (a longer limit, or none) and the exchange DID complete -- it was slow, not stuck: the stream
        // is healthy, the communicator is back in service unless another gather is still overdue
        g->timed_out = false;
        bool any = false;
        for (akz_gather* o : c->pool) any = any || o->timed_out;
        c->abandoned = any;
    }
    if (!g->finished) {
        if (c->host) {
            for (int r = 0; r < c->nranks; ++r) std::memcpy(g->pinned + (size_t)r * 8, g->recv + (size_t)r * g->send_bytes, kRow);
        } else {
            AKZ_HIP_TRY(hipMemcpy2DAsync(g->pinned, kRow, g->recv, g->send_bytes, kRow, (size_t)c->nranks, hipMemcpyDeviceToHost, c->cs));
            AKZ_HIP_TRY(hipStreamSynchronize(c->cs));
        }
        g->hdr_rows.assign((size_t)c->nranks, 0);
        g->hdr_images.assign((size_t)c->nranks, 0);
        g->image_rows.assign((size_t)c->nranks, {});
        bool overflow = false;
        for (int r = 0; r < c->nranks; ++r) {
            const uint64_t* h = g->pinned + (size_t)r * 8;
            if (h[2] != g->cap_rows) {
                set_error("gather: ranks disagree on the block capacity (every rank must pass the same cap_rows)");
                return AKZ_ERR_INVALID_ARG;
            }
            overflow = overflow || h[4] != 0 || h[0] + h[5] > g->cap_rows;
            g->hdr_rows[(size_t)r] = h[0];
            g->hdr_images[(size_t)r] = h[1];
        }
        if (!overflow)
            for (int r = 0; r < c->nranks; ++r) {  // the per-image tables (a few rows per rank)
                const uint64_t* h = g->pinned + (size_t)r * 8;
                std::vector<uint64_t>& t = g->image_rows[(size_t)r];
                t.assign((size_t)h[5] * 8, 0);
                if (h[5] && c->host)
                    std::memcpy(t.data(), g->recv + (size_t)r * g->send_bytes + (1 + h[0]) * kRow, h[5] * kRow);
                else if (h[5])
                    AKZ_HIP_TRY(hipMemcpy(t.data(), g->recv + (size_t)r * g->send_bytes + (1 + h[0]) * kRow, h[5] * kRow,
                                          hipMemcpyDeviceToHost));
                t.resize((size_t)std::min<uint64_t>(h[1], t.size()));
                if (h[5] == 0 && h[1] == 1) t.assign(1, h[0]);  // raw rows: one image
            }
        g->overflow = overflow;
        g->finished = true;
    }
    for (int r = 0; r < c->nranks; ++r) {
        if (counts) counts[r] = g->hdr_rows[(size_t)r];
        if (images) images[r] = g->hdr_images[(size_t)r];
    }
    if (g->overflow) {
        set_error("gather: a rank's shard has more descriptor rows than the agreed capacity (counts hold what each rank needed; "
                  "repeat the exchange with a larger cap_rows on every rank)");
        return AKZ_ERR_BUFFER;
    }
    if (d_all) *d_all = g->recv;
    if (block_rows) *block_rows = 1 + g->cap_rows;
    return AKZ_OK;
}

int akz_gather_image_rows(akz_gather* g, int rank, uint64_t* rows_per_image, uint64_t cap, uint64_t* n_images) {
    if (!g || !g->in_use || !g->finished || rank < 0 || rank >= g->comm->nranks) {
        set_error("akz_gather_image_rows: not a finished gather / rank out of range");
        return AKZ_ERR_INVALID_ARG;
    }
    const std::vector<uint64_t>& t = g->image_rows[(size_t)rank];
    if (n_images) *n_images = t.size();
    if (rows_per_image)
        for (size_t i = 0; i < t.size() && i < cap; ++i) rows_per_image[i] = t[i];
    return AKZ_OK;
}

int akz_gather_free(akz_gather* g) {
    if (!g) return AKZ_OK;
    if (g->comm->abandoned) {  // its collective may never complete: do not wait, and never hand the buffers out again
        return AKZ_OK;         // (in_use stays set; gather_acquire refuses new work on this communicator anyway)
    }
    if (!g->comm->host && g->in_use && g->done && (!g->comm->external || g->delivered)) {
        (void)hipSetDevice(g->comm->device);
        (void)hipEventSynchronize(g->done);
    }
    g->in_use = false;
    return AKZ_OK;
}

// SURVEY.md Appendix C form: synchronous, exact counts, rows of all ranks compacted in rank order.  Two collectives
// (the counts, then blocks padded to the largest shard); *d_all stays valid until the next call on this communicator.
int akz_gather_descriptors(akz_comm* c, const uint8_t* d_local, uint64_t n_local, const uint8_t** d_all, uint64_t* counts) {
    if (!c || !d_all || !counts || (n_local && !d_local)) {
        set_error("akz_gather_descriptors: null argument");
        return AKZ_ERR_INVALID_ARG;
    }
    *d_all = nullptr;
    if (c->external) {
        set_error("akz_gather_descriptors: the synchronous form needs the library's own transport (akz_comm_create); with "
                  "akz_comm_create_external use akz_gather_begin_rows / akz_gather_blocks / akz_gather_deliver");
        return AKZ_ERR_UNSUPPORTED;
    }
    AKZ_HIP_TRY(hipSetDevice(c->device));
    akz_gather* g0 = nullptr;  // header-only exchange: every rank learns every shard's row count
    AKZ_TRY(gather_acquire(c, 0, &g0));
    const uint8_t* none = nullptr;
    const uint64_t zero = 0;
    int st = gather_enqueue(c, g0, &none, &zero, 1, n_local, nullptr, false);
    std::vector<uint64_t> all((size_t)c->nranks, 0);
    if (st == AKZ_OK) st = akz_gather_finish(g0, nullptr, nullptr, nullptr, all.data());  // n_local travels as `images`
    akz_gather_free(g0);
    AKZ_TRY(st);
    uint64_t cap = 1, total = 0;
    for (uint64_t v : all) {
        cap = std::max(cap, v);
        total += v;
    }
    akz_gather* g = nullptr;
    AKZ_TRY(gather_acquire(c, cap, &g));
    const uint8_t* blocks = nullptr;
    uint64_t block_rows = 0;
    st = gather_enqueue(c, g, &d_local, &n_local, 1, 1, nullptr, false);
    if (st == AKZ_OK) st = akz_gather_finish(g, &blocks, &block_rows, counts, nullptr);
    if (st == AKZ_OK && c->sync_out_bytes < std::max<uint64_t>(1, total) * kRow) {
        if (c->sync_out) (void)hipFree(c->sync_out);
        c->sync_out = nullptr;
        c->sync_out_bytes = 0;
        const size_t want = (size_t)(std::max<uint64_t>(1, total) * kRow * 5 / 4);
        if (hipMalloc((void**)&c->sync_out, want) != hipSuccess) {
            set_error("akz_gather_descriptors: hipMalloc failed");
            st = AKZ_ERR_HIP;
        } else {
            c->sync_out_bytes = want;
        }
    }
    if (st == AKZ_OK) {
        uint64_t at = 0;
        for (int r = 0; r < c->nranks && st == AKZ_OK; ++r) {
            if (counts[r] && hipMemcpyAsync(c->sync_out + at * kRow, blocks + ((uint64_t)r * block_rows + 1) * kRow,
                                            counts[r] * kRow, hipMemcpyDeviceToDevice, c->cs) != hipSuccess) {
                set_error("akz_gather_descriptors: compaction copy failed");
                st = AKZ_ERR_HIP;
            }
            at += counts[r];
        }
        if (st == AKZ_OK && hipStreamSynchronize(c->cs) != hipSuccess) {
            set_error("akz_gather_descriptors: synchronisation failed");
            st = AKZ_ERR_HIP;
        }
    }
    akz_gather_free(g);
    AKZ_TRY(st);
    *d_all = c->sync_out;
    return AKZ_OK;
}

// ---- all-pairs match over a finished gather (BASELINE configs[4]) ------------------------------------------------------
// Every unordered image pair {a, b} of the job is matched ONCE, in both directions, by the rank that owns its LEAD image
// (pairs_lead below: a balanced rule, every image leads about half of its pairs).  The lead image is the query set of one
// both-direction launch (akz_descriptor_match_sets_mutual_device) against the images it leads, taken where they lie in
// the gathered block; hamming(a, b) = hamming(b, a), so the launch's matrix-core work serves descriptor_match(a, b) AND
// descriptor_match(b, a).  Round 3 matched every ORDERED pair (and every image against itself): 2.1 x the work.
}  // extern "C"
struct akz_pairs {
    akz_ctx* ctx = nullptr;
    akz_comm* comm = nullptr;
    int device = 0;
    uint8_t* d_block = nullptr;          // the gathered rows (compacted, rank-major), then every list and count
    size_t block_bytes = 0;
    uint64_t* h_cnt = nullptr;           // pinned: the counts of every list, copied behind the launches
    size_t h_cnt_entries = 0;
    hipEvent_t done = nullptr;           // the counts have arrived
    hipEvent_t fork = nullptr, join = nullptr;  // the matcher's second stream: after the rows are in place / before the counts leave
    hipStream_t last_stream = nullptr;   // the matcher's stream of the step that used these buffers last (reuse on another one waits for `done`)
    bool used = false;                   // `done` has been recorded at least once
    bool waited = false;
    bool plan_only = false;              // akz_pairs_plan: who matches what, nothing matched (no device object behind it)
    std::vector<uint64_t> rows, offset;  // per image: rows, first row in the compacted block
    std::vector<int> owner;
    uint64_t first_owned = 0, n_owned = 0;
    struct Lead {                        // one per owned image
        std::vector<uint64_t> sets;      // the images it leads, ascending
        std::vector<uint64_t> col0;      // first record of set k's opposite-direction list
        akz_match *d_rows = nullptr, *d_cols = nullptr;  // [sets][rows of the lead], [sum of the sets' rows]
        size_t cnt0 = 0;                 // index of its counts in h_cnt: sets.size() (lead -> set) then sets.size() (set -> lead)
    };
    std::vector<Lead> lead;
};
// the image of an unordered pair that serves as the query set of its block
static uint64_t pairs_lead(uint64_t a, uint64_t b) {
    const uint64_t lo = std::min(a, b), hi = std::max(a, b);
    return ((hi - lo) & 1u) ? lo : hi;
}
static void delete_pairs_host_only(akz_pairs* p) { delete p; }  // abandoned communicator: device objects leak (see akz_comm_destroy)
static void pairs_destroy(akz_pairs* p) {
    if (!p) return;
    if (p->d_block) (void)hipFree(p->d_block);
    if (p->h_cnt) (void)hipHostFree(p->h_cnt);
    if (p->done) (void)hipEventDestroy(p->done);
    if (p->fork) (void)hipEventDestroy(p->fork);
    if (p->join) (void)hipEventDestroy(p->join);
    delete p;
}
// Lifetime rule (akaze_hip.h): an akz_pairs may outlive its communicator.  While the communicator lives, a freed object goes
// back to its pool (at most kPairsPool of them; the rest are destroyed); akz_comm_destroy clears the comm pointer of every
// object still held by the caller, whose akz_pairs_free then destroys it on its own.
static void pairs_release(akz_pairs* p) {
    if (!p) return;
    akz_comm* c = p->comm;
    if (c) {
        auto it = std::find(c->live_pairs.begin(), c->live_pairs.end(), p);
        if (it != c->live_pairs.end()) c->live_pairs.erase(it);
        if (c->pairs_pool.size() < akz_comm::kPairsPool) {
            c->pairs_pool.push_back(p);
            return;
        }
    }
    if (p->used && p->done) (void)hipEventSynchronize(p->done);  // launches of its last step may still read the block
    pairs_destroy(p);
}
// The all-pairs plan of a finished gather, host arithmetic only: the job's images numbered rank-major with their row counts
// and owners, this rank's images, and for each of those the images it LEADS (pairs_lead) -- the pairs this rank matches.
static int pairs_plan(akz_comm* c, akz_gather* g, akz_pairs* p, uint64_t* total_rows) {
    p->rows.clear(); p->offset.clear(); p->owner.clear(); p->lead.clear();
    p->first_owned = p->n_owned = 0;
    uint64_t total = 0;
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) p->first_owned = p->rows.size();
        uint64_t in_rank = 0;
        for (uint64_t n : g->image_rows[(size_t)r]) {
            p->offset.push_back(total + in_rank);
            p->rows.push_back(n);
            p->owner.push_back(r);
            in_rank += n;
        }
        if (in_rank != g->hdr_rows[(size_t)r]) {
            set_error("akz_match_all_pairs: a block's per-image table does not add up to its row count");
            return AKZ_ERR_INVALID_ARG;
        }
        if (r == c->rank) p->n_owned = g->image_rows[(size_t)r].size();
        total += in_rank;
    }
    const uint64_t n_images = p->rows.size();
    p->lead.resize((size_t)p->n_owned);
    for (uint64_t k = 0; k < p->n_owned; ++k) {
        akz_pairs::Lead& L = p->lead[(size_t)k];
        const uint64_t q = p->first_owned + k;
        uint64_t col_rows = 0;
        for (uint64_t j = 0; j < n_images; ++j)
            if (j != q && pairs_lead(q, j) == q) {
                L.sets.push_back(j);
                L.col0.push_back(col_rows);
                col_rows += p->rows[(size_t)j];
            }
    }
    if (total_rows) *total_rows = total;
    return AKZ_OK;
}
static void pairs_orphan_all(akz_comm* c) {
    for (akz_pairs* p : c->live_pairs) p->comm = nullptr;
    c->live_pairs.clear();
}
extern "C" {

int akz_match_all_pairs(akz_ctx* ctx, akz_gather* g, uint64_t distance_threshold, double lowes_ratio, akz_pairs** out) {
    if (!out) return AKZ_ERR_INVALID_ARG;
    *out = nullptr;
    if (!ctx || !g || !g->in_use) {
        set_error("akz_match_all_pairs: null context / not a gather in flight");
        return AKZ_ERR_INVALID_ARG;
    }
    akz_comm* c = g->comm;
    if (c->host) {
        set_error("akz_match_all_pairs: the matcher needs the gathered rows on a device; a host-memory communicator offers akz_pairs_plan");
        return AKZ_ERR_UNSUPPORTED;
    }
    const uint8_t* blocks = nullptr;
    uint64_t block_rows = 0;
    AKZ_TRY(akz_gather_finish(g, &blocks, &block_rows, nullptr, nullptr));
    AKZ_HIP_TRY(hipSetDevice(c->device));
    // a pairs object of an earlier step whose buffers can be reused (steady state: no allocation)
    akz_pairs* p = nullptr;
    if (!c->pairs_pool.empty()) {
        p = c->pairs_pool.back();
        c->pairs_pool.pop_back();
    } else {
        p = new akz_pairs;
    }
    struct Back {  // on any error the object returns to the pool
        akz_pairs* p;
        bool armed = true;
        ~Back() { if (armed) pairs_release(p); }
    } back{p};
    p->ctx = ctx;
    p->comm = c;
    c->live_pairs.push_back(p);
    p->device = c->device;
    p->waited = false;
    p->plan_only = false;
    uint64_t total = 0;
    AKZ_TRY(pairs_plan(c, g, p, &total));
    const uint64_t n_images = p->rows.size();
    (void)n_images;
    // layout of the block: rows | per lead: lists lead -> set, lists set -> lead | counts
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t bytes = up(std::max<uint64_t>(1, total) * kRow), n_cnt = 0;
    std::vector<size_t> off_rows((size_t)p->n_owned), off_cols((size_t)p->n_owned);
    for (uint64_t k = 0; k < p->n_owned; ++k) {
        akz_pairs::Lead& L = p->lead[(size_t)k];
        const uint64_t q = p->first_owned + k;
        const uint64_t col_rows = L.sets.empty() ? 0 : L.col0.back() + p->rows[(size_t)L.sets.back()];
        off_rows[(size_t)k] = bytes;
        bytes += up(std::max<uint64_t>(1, L.sets.size() * p->rows[(size_t)q]) * sizeof(akz_match));
        off_cols[(size_t)k] = bytes;
        bytes += up(std::max<uint64_t>(1, col_rows) * sizeof(akz_match));
        L.cnt0 = n_cnt;
        n_cnt += 2 * L.sets.size();
    }
    const size_t off_cnt = bytes;
    bytes += up(std::max<size_t>(1, n_cnt) * sizeof(uint64_t));
    hipStream_t ms = (hipStream_t)akz_ctx_stream(ctx);  // the matcher's stream: everything below is enqueued on it, in order
    // A pooled object's buffers were last used by launches on `last_stream`.  On the same stream this step's copies are
    // ordered behind them; on another one (a second context on this communicator, a context that was recreated) they are
    // not: this step's first write waits for that step's `done`.
    if (p->used && p->last_stream != ms) AKZ_HIP_TRY(hipStreamWaitEvent(ms, p->done, 0));
    if (p->block_bytes < bytes) {
        if (p->d_block) {
            if (p->used) AKZ_HIP_TRY(hipEventSynchronize(p->done));  // (an earlier step's launches may still read the old block)
            AKZ_HIP_TRY(hipFree(p->d_block));
        }
        p->d_block = nullptr;
        p->block_bytes = 0;
        AKZ_HIP_TRY(hipMalloc((void**)&p->d_block, bytes + bytes / 4));
        p->block_bytes = bytes + bytes / 4;
    }
    if (p->h_cnt_entries < std::max<size_t>(1, n_cnt)) {
        if (p->h_cnt && p->used) AKZ_HIP_TRY(hipEventSynchronize(p->done));  // (the last step's count copy targets it)
        if (p->h_cnt) AKZ_HIP_TRY(hipHostFree(p->h_cnt));
        p->h_cnt = nullptr;
        p->h_cnt_entries = 0;
        AKZ_HIP_TRY(hipHostMalloc((void**)&p->h_cnt, std::max<size_t>(1, n_cnt) * 2 * sizeof(uint64_t), hipHostMallocDefault));
        p->h_cnt_entries = std::max<size_t>(1, n_cnt) * 2;
    }
    if (!p->done) AKZ_HIP_TRY(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
    // the gathered rows, compacted: the gather's buffers go back to the communicator as soon as these copies have run
    uint64_t at = 0;
    for (int r = 0; r < c->nranks; ++r) {
        const uint64_t n = g->hdr_rows[(size_t)r];
        if (n)
            AKZ_HIP_TRY(hipMemcpyAsync(p->d_block + at * kRow, blocks + ((uint64_t)r * block_rows + 1) * kRow, n * kRow,
                                       hipMemcpyDeviceToDevice, ms));
        at += n;
    }
    if (!g->readers) AKZ_HIP_TRY(hipEventCreateWithFlags(&g->readers, hipEventDisableTiming));
    AKZ_HIP_TRY(hipEventRecord(g->readers, ms));
    g->readers_pending = true;
    uint64_t* d_cnt = (uint64_t*)(p->d_block + off_cnt);
    if (n_cnt) AKZ_HIP_TRY(hipMemsetAsync(d_cnt, 0, n_cnt * sizeof(uint64_t), ms));
    // Consecutive lead images alternate between the matcher's stream and its side stream (own scratch): around an image's
    // one big pass sit an unpack, a seed launch and two compactions that do not fill the chip (130 of 550 us at 16 x 4K) --
    // they run under the other image's pass.
    hipStream_t side = p->n_owned > 1 ? akz::match_side_stream(ctx) : nullptr;
    if (side) {
        if (!p->fork) AKZ_HIP_TRY(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
        if (!p->join) AKZ_HIP_TRY(hipEventCreateWithFlags(&p->join, hipEventDisableTiming));
        AKZ_HIP_TRY(hipEventRecord(p->fork, ms));
        AKZ_HIP_TRY(hipStreamWaitEvent(side, p->fork, 0));
    }
    struct Join {  // (also on an error return: the side stream's work is ordered before whatever follows on the matcher's)
        hipStream_t side, ms;
        hipEvent_t ev;
        ~Join() {
            if (side && hipEventRecord(ev, side) == hipSuccess) (void)hipStreamWaitEvent(ms, ev, 0);
        }
    };
    {
    Join joiner{side, ms, p->join};
    for (uint64_t k = 0; k < p->n_owned; ++k) {
        akz_pairs::Lead& L = p->lead[(size_t)k];
        const uint64_t q = p->first_owned + k;
        L.d_rows = (akz_match*)(p->d_block + off_rows[(size_t)k]);
        L.d_cols = (akz_match*)(p->d_block + off_cols[(size_t)k]);
        if (L.sets.empty()) continue;
        std::vector<uint64_t> first, rows;
        for (uint64_t j : L.sets) {
            first.push_back(p->offset[(size_t)j]);
            rows.push_back(p->rows[(size_t)j]);
        }
        AKZ_TRY(akz::match_sets_at(ctx, p->d_block + p->offset[(size_t)q] * kRow, p->rows[(size_t)q], p->d_block, first.data(), rows.data(),
                                   L.sets.size(), distance_threshold, lowes_ratio, L.d_rows, d_cnt + L.cnt0, L.d_cols,
                                   d_cnt + L.cnt0 + L.sets.size(), side && (k & 1) ? 1 : 0));
    }
    }
    if (n_cnt) AKZ_HIP_TRY(hipMemcpyAsync(p->h_cnt, d_cnt, n_cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, ms));
    AKZ_HIP_TRY(hipEventRecord(p->done, ms));
    p->last_stream = ms;
    p->used = true;
    back.armed = false;
    *out = p;
    return AKZ_OK;
}
int akz_pairs_plan(akz_gather* g, akz_pairs** out) {
    if (!out) return AKZ_ERR_INVALID_ARG;
    *out = nullptr;
    if (!g || !g->in_use) {
        set_error("akz_pairs_plan: not a gather in flight");
        return AKZ_ERR_INVALID_ARG;
    }
    akz_comm* c = g->comm;
    AKZ_TRY(akz_gather_finish(g, nullptr, nullptr, nullptr, nullptr));
    akz_pairs* p = new akz_pairs;  // (never pooled: it owns nothing on a device)
    p->comm = nullptr;
    p->device = c->device;
    p->plan_only = true;
    const int st = pairs_plan(c, g, p, nullptr);
    if (st != AKZ_OK) {
        delete p;
        return st;
    }
    *out = p;
    return AKZ_OK;
}
int akz_pairs_lead_sets(const akz_pairs* p, uint64_t lead_image, uint64_t* images, uint64_t cap, uint64_t* n) {
    if (!p || !n || lead_image < p->first_owned || lead_image >= p->first_owned + p->n_owned) {
        set_error("akz_pairs_lead_sets: an image this rank owns");
        return AKZ_ERR_INVALID_ARG;
    }
    const akz_pairs::Lead& L = p->lead[(size_t)(lead_image - p->first_owned)];
    *n = L.sets.size();
    for (size_t i = 0; i < L.sets.size() && i < cap && images; ++i) images[i] = L.sets[i];
    return AKZ_OK;
}
int akz_pairs_info(const akz_pairs* p, uint64_t* n_images, uint64_t* first_owned, uint64_t* n_owned) {
    if (!p) return AKZ_ERR_INVALID_ARG;
    if (n_images) *n_images = p->rows.size();
    if (first_owned) *first_owned = p->first_owned;
    if (n_owned) *n_owned = p->n_owned;
    return AKZ_OK;
}
int akz_pairs_image_rows(const akz_pairs* p, uint64_t image, uint64_t* rows, int* owner_rank) {
    if (!p || image >= p->rows.size()) return AKZ_ERR_INVALID_ARG;
    if (rows) *rows = p->rows[(size_t)image];
    if (owner_rank) *owner_rank = p->owner[(size_t)image];
    return AKZ_OK;
}
int akz_pairs_holder(const akz_pairs* p, uint64_t image_a, uint64_t image_b, int* rank) {
    if (!p || !rank || image_a >= p->rows.size() || image_b >= p->rows.size() || image_a == image_b) {
        set_error("akz_pairs_holder: two different images of the job");
        return AKZ_ERR_INVALID_ARG;
    }
    *rank = p->owner[(size_t)pairs_lead(image_a, image_b)];
    return AKZ_OK;
}
int akz_pairs_matches(const akz_pairs* pc, uint64_t query, uint64_t image, akz_match* out, uint64_t cap, uint64_t* n) {
    akz_pairs* p = const_cast<akz_pairs*>(pc);
    if (!p || !n || query >= p->rows.size() || image >= p->rows.size()) {
        set_error("akz_pairs_matches: images of the job");
        return AKZ_ERR_INVALID_ARG;
    }
    *n = 0;
    if (query == image) return AKZ_OK;  // (an image is not matched against itself)
    const uint64_t lead = pairs_lead(query, image);
    if (lead < p->first_owned || lead >= p->first_owned + p->n_owned) {
        set_error("akz_pairs_matches: this pair was matched by the rank akz_pairs_holder names");
        return AKZ_ERR_INVALID_ARG;
    }
    if (p->plan_only) {
        set_error("akz_pairs_matches: a plan (akz_pairs_plan) holds no lists; akz_match_all_pairs does");
        return AKZ_ERR_INVALID_ARG;
    }
    const akz_pairs::Lead& L = p->lead[(size_t)(lead - p->first_owned)];
    const uint64_t other = lead == query ? image : query;
    const size_t k = (size_t)(std::lower_bound(L.sets.begin(), L.sets.end(), other) - L.sets.begin());
    AKZ_HIP_TRY(hipSetDevice(p->device));
    if (!p->waited) {
        AKZ_HIP_TRY(hipEventSynchronize(p->done));
        p->waited = true;
    }
    const bool forward = lead == query;  // the lead's rows are the queries of the list asked for
    const uint64_t cnt = p->h_cnt[L.cnt0 + (forward ? 0 : L.sets.size()) + k];
    *n = cnt;
    const uint64_t take = std::min(cnt, cap);
    if (out && take) {
        const akz_match* src = forward ? L.d_rows + k * p->rows[(size_t)lead] : L.d_cols + L.col0[k];
        AKZ_HIP_TRY(hipMemcpy(out, src, take * sizeof(akz_match), hipMemcpyDeviceToHost));
    }
    return AKZ_OK;
}
int akz_pairs_totals(const akz_pairs* pc, uint64_t* n_lists, uint64_t* n_matches, uint64_t* n_distances) {
    akz_pairs* p = const_cast<akz_pairs*>(pc);
    if (!p) return AKZ_ERR_INVALID_ARG;
    if (p->plan_only) {  // what WOULD be matched: lists and distances, no matches yet
        uint64_t lists = 0, dist = 0;
        for (uint64_t k = 0; k < p->n_owned; ++k)
            for (uint64_t j : p->lead[(size_t)k].sets) {
                lists += 2;
                dist += p->rows[(size_t)(p->first_owned + k)] * p->rows[(size_t)j];
            }
        if (n_lists) *n_lists = lists;
        if (n_matches) *n_matches = 0;
        if (n_distances) *n_distances = dist;
        return AKZ_OK;
    }
    AKZ_HIP_TRY(hipSetDevice(p->device));
    if (!p->waited) {
        AKZ_HIP_TRY(hipEventSynchronize(p->done));
        p->waited = true;
    }
    uint64_t lists = 0, matches = 0, dist = 0;
    for (uint64_t k = 0; k < p->n_owned; ++k) {
        const akz_pairs::Lead& L = p->lead[(size_t)k];
        const uint64_t rq = p->rows[(size_t)(p->first_owned + k)];
        for (size_t i = 0; i < L.sets.size(); ++i) {
            lists += 2;
            matches += p->h_cnt[L.cnt0 + i] + p->h_cnt[L.cnt0 + L.sets.size() + i];
            dist += rq * p->rows[(size_t)L.sets[i]];
        }
    }
    if (n_lists) *n_lists = lists;
    if (n_matches) *n_matches = matches;
    if (n_distances) *n_distances = dist;
    return AKZ_OK;
}
int akz_pairs_free(akz_pairs* p) {
    if (!p) return AKZ_OK;
    if (p->plan_only) {
        delete p;
        return AKZ_OK;
    }
    (void)hipSetDevice(p->device);
    // the buffers go back to the communicator's pool: the next step's launches are ordered behind this step's on the
    // matcher's stream, and a reader of this object has waited for `done`
    pairs_release(p);
    return AKZ_OK;
}

}  // extern "C"
