// A stand-in for librccl.so.1 -- TEST INFRASTRUCTURE, never shipped, never loaded by the product outside tests/.
// akz_comm.cpp binds RCCL at run time (dlopen("librccl.so.1")); a test puts the directory of this library first in
// LD_LIBRARY_PATH of a C++ host (gather_selftest: no PyTorch, so no other RCCL is in the process) to exercise what cannot be
// exercised with the real library on a one-GPU box:
//   * fault paths: AKZ_STUB_RCCL=init_fail (ncclCommInitRank fails), allgather_fail (ncclAllGather returns an error),
//     built with -DSTUB_NO_ALLGATHER (a symbol is missing); stall_after_N (the (N+1)-th ncclAllGather "loses its peer": it
//     returns success like the real call does once the collective is enqueued, and the stream then stays busy for
//     AKZ_STUB_RCCL_STALL_S seconds -- default 25 -- behind a HOST function that sleeps: the stream-side picture of a
//     collective whose peer died, bounded, and without a single GPU wave spinning);
//   * the communicator's logic with MORE THAN ONE rank: AKZ_STUB_RCCL=ok (default) all-gathers between PROCESSES through a
//     POSIX shared-memory segment named by the unique id (host staged, synchronous; a peer that does not arrive within
//     AKZ_STUB_RCCL_TIMEOUT_S -- default 30 -- is an error, not a hang), so that `gather_selftest RANK NRANKS ID_FILE 0`
//     runs N ranks of the RCCL code path of akz_comm.cpp on one device.
// It implements only the five entry points akz_comm.cpp binds, for ncclUint8.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {
constexpr size_t kData = 64u << 20;  // room for nranks blocks (sparse until touched)
struct Shared {
    std::atomic<unsigned> arrived;
    std::atomic<unsigned> generation;
    std::atomic<unsigned> attached;
    char pad[64 - 3 * sizeof(std::atomic<unsigned>)];
    unsigned char data[1];
};
struct Comm {
    int rank = 0, nranks = 1;
    std::string name;
    Shared* sh = nullptr;
};
const char* mode() {
    const char* m = getenv("AKZ_STUB_RCCL");
    return m ? m : "ok";
}
double timeout_s() {
    const char* t = getenv("AKZ_STUB_RCCL_TIMEOUT_S");
    return t ? atof(t) : 30.0;
}
bool barrier(Comm* c) {
    if (c->nranks == 1) return true;
    Shared* s = c->sh;
    const unsigned gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == (unsigned)c->nranks) {
        s->arrived.store(0);
        s->generation.store(gen + 1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/akzstub_%d_%ld", (int)getpid(), (long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!std::strcmp(mode(), "init_fail")) return ncclSystemError;
    Comm* c = new Comm;
    c->rank = rank;
    c->nranks = nranks;
    c->name = id.internal;
    const size_t bytes = sizeof(Shared) + kData;
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
        if (fd >= 0) close(fd);
        delete c;
        return ncclSystemError;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->sh = (Shared*)p;  // (a fresh segment is zero-filled: counters start at 0)
    c->sh->attached.fetch_add(1);
    if (!barrier(c)) {  // the real call is collective too
        munmap(p, bytes);
        delete c;
        return ncclSystemError;
    }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    const unsigned left = c->sh->attached.fetch_sub(1) - 1;
    munmap(c->sh, sizeof(Shared) + kData);
    if (left == 0) shm_unlink(c->name.c_str());
    delete c;
    return ncclSuccess;
}

#ifndef STUB_NO_ALLGATHER
namespace {
std::atomic<int> g_calls{0};
void stall(void* seconds) {
    const double s = *(const double*)seconds;
    std::this_thread::sleep_for(std::chrono::duration<double>(s));
}
double g_stall_s = 25.0;
}  // namespace
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!std::strcmp(mode(), "allgather_fail")) return ncclUnhandledCudaError;
    if (!std::strncmp(mode(), "stall_after_", 12) && g_calls.fetch_add(1) >= atoi(mode() + 12)) {
        const char* t = getenv("AKZ_STUB_RCCL_STALL_S");
        if (t) g_stall_s = atof(t);
        return hipLaunchHostFunc(stream, stall, &g_stall_s) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
    }
    if (dt != ncclUint8 && dt != ncclInt8) return ncclInvalidArgument;
    if (count * (size_t)c->nranks > kData) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->sh->data + (size_t)c->rank * count, send, count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;   // every block is in place
    if (hipMemcpy(recv, c->sh->data, count * (size_t)c->nranks, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;   // every rank has read: the slots may be overwritten
    return ncclSuccess;
}
#endif

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled cuda error (stub)";
        case ncclSystemError: return "unhandled system error (stub)";
        case ncclInvalidArgument: return "invalid argument (stub)";
        default: return "error (stub)";
    }
}

}  // extern "C"
