// Which streams of a process make progress side by side?  N non-blocking streams; for every ordered pair (A, B): a chain
// of K back-to-back 40 us single-wave spins on A, then one spin on B; printed: when B's spin finished relative to the
// start of A's chain (us).  ~40-100: B ran beside A; >= K * 40: B waited for A's chain (same hardware queue, or -- what
// this tool is for -- queues that share a pipe of the command processor).  Second table: two plain spins (the round-4
// placement probe's first form), which only sees shared queues.
//   hipcc --offload-arch=gfx950 -O2 -o queue_probe queue_probe.hip && GPU_MAX_HW_QUEUES=8 ./queue_probe 8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_delay(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while (wall_clock64() - t0 < ticks && spins < (1u << 26)) ++spins;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8, K = 10;
    std::vector<hipStream_t> s(N);
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto& x : s) { hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, x, 100ull); CK(hipStreamSynchronize(x)); }
    printf("GPU_MAX_HW_QUEUES=%s, %d streams; rows A (busy chain), columns B (one spin): B's finish, us after A's start\n",
           getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)", N);
    for (int mode = 0; mode < 2; ++mode) {
        printf(mode == 0 ? "chain of %d x 40 us on A:\n" : "one 120 us spin on A:\n", K);
        for (int a = 0; a < N; ++a) {
            for (int b = 0; b < N; ++b) {
                if (a == b) { printf("     -"); continue; }
                CK(hipEventRecord(e0, s[a]));
                if (mode == 0) for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s[a], 4000ull);
                else hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s[a], 12000ull);
                hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s[b], mode == 0 ? 4000ull : 12000ull);
                CK(hipEventRecord(e1, s[b]));
                CK(hipEventSynchronize(e1));
                CK(hipStreamSynchronize(s[a]));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %5.0f", ms * 1e3f);
            }
            printf("\n");
        }
    }
    // third table: 40 tiny kernels on each of A and B at the same time; time until both have drained (us).  Queues that share
    // a command-processor pipe (or a hardware queue) dispatch one after the other
    printf("40 x 1 us kernels on A and on B together (diagonal: on A alone):\n");
    for (int a = 0; a < N; ++a) {
        for (int b = 0; b < N; ++b) {
            CK(hipEventRecord(e0, s[a]));
            for (int k = 0; k < 40; ++k) {
                hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s[a], 100ull);
                if (a != b) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s[b], 100ull);
            }
            CK(hipEventRecord(e1, s[b]));
            CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(s[a]));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %5.0f", ms * 1e3f);
        }
        printf("\n");
    }
    // fourth: B's kernel waits (event) for a kernel at the END of a 5 x 40 us chain on A, and A then waits for B (the shape of
    // the pipeline's early stages): time until A's last kernel has finished
    printf("A: 5 x 40 us, event -> B: 40 us, event -> A: 40 us (ideal 280):\n");
    hipEvent_t ea, eb;
    CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    for (int a = 0; a < N; ++a) {
        for (int b = 0; b < N; ++b) {
            if (a == b) { printf("     -"); continue; }
            CK(hipEventRecord(e0, s[a]));
            for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s[a], 4000ull);
            CK(hipEventRecord(ea, s[a]));
            CK(hipStreamWaitEvent(s[b], ea, 0));
            hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s[b], 4000ull);
            CK(hipEventRecord(eb, s[b]));
            CK(hipStreamWaitEvent(s[a], eb, 0));
            hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s[a], 4000ull);
            CK(hipEventRecord(e1, s[a]));
            CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(s[b]));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %5.0f", ms * 1e3f);
        }
        printf("\n");
    }
    return 0;
}
