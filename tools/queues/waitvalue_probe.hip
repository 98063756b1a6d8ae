// Can a stream be made to wait for something that has NOT been enqueued yet?  hipStreamWaitEvent only sees records that
// precede the call; hipStreamWaitValue32 waits for a memory word, which another stream (hipStreamWriteValue32) or the host
// can set later.  Checks support on this pool and measures the release latency.
//   hipcc --offload-arch=gfx950 -O2 -o waitvalue_probe waitvalue_probe.hip && ./waitvalue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_delay(unsigned long long ticks, unsigned* out) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while (wall_clock64() - t0 < ticks && spins < (1u << 28)) ++spins;
    if (out) *out = spins;
}
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    uint32_t* sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc((void**)&sig, 64)); printf("  (plain hipMalloc instead)\n"); }
    CK(hipMemset(sig, 0, 64));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, a, 100ull, nullptr);
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, b, 100ull, nullptr);
    CK(hipDeviceSynchronize());
    for (uint32_t round = 1; round <= 5; ++round) {
        // stream a: wait for sig >= round, then a short kernel; stream b (enqueued LATER): 200 us of work, then the write
        CK(hipEventRecord(e0, a));
        e = hipStreamWaitValue32(a, sig, round, hipStreamWaitValueGte, 0xffffffffu);
        if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 1; }
        hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, a, 100ull, nullptr);
        CK(hipEventRecord(e1, a));
        std::this_thread::sleep_for(std::chrono::microseconds(300));   // the host enqueues b's work later
        hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, b, 20000ull, nullptr);   // 200 us at 100 MHz
        e = hipStreamWriteValue32(b, sig, round, 0);
        if (e != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); return 1; }
        CK(hipEventRecord(e2, b));
        CK(hipEventSynchronize(e1));
        CK(hipEventSynchronize(e2));
        float ms_a = 0, ms_ab = 0;
        CK(hipEventElapsedTime(&ms_a, e0, e1));
        CK(hipEventElapsedTime(&ms_ab, e2, e1));
        printf("round %u: stream a was held %.0f us (host delay 300 + 200 of work on b); a's kernel ended %.1f us after b's write\n", round, ms_a * 1e3f, ms_ab * 1e3f);
    }
    // release from the host: a stream write on a third stream with nothing in front of it
    CK(hipEventRecord(e0, a));
    CK(hipStreamWaitValue32(a, sig, 100, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, a, 100ull, nullptr);
    CK(hipEventRecord(e1, a));
    std::this_thread::sleep_for(std::chrono::microseconds(500));
    hipStream_t c;
    CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    CK(hipStreamWriteValue32(c, sig, 100, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("host-side release after 500 us: a was held %.0f us\n", ms * 1e3f);
    return 0;
}
