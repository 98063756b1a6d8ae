// What HBM itself sustains on this MI355X for plain streaming kernels (tools only): every launch works on buffers the
// previous launches did not touch (a rotation over 8 GB), so the 256 MB Infinity Cache cannot serve any of it.
//   hbm_ceiling -> table (GB/s): read-only, write-only, copy, 1 read : 6 writes, with ordinary and nontemporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, int W, int NT>
__global__ void __launch_bounds__(256) k_stream(const f4* __restrict__ in, f4* __restrict__ out, size_t n4, size_t plane4, float* sink) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f4 v = {1, 2, 3, 4};
#pragma unroll
        for (int r = 0; r < R; ++r) v += in[i + r * plane4];
        if (W == 0) acc += v;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (NT) __builtin_nontemporal_store(v, out + i + k * plane4);
            else out[i + k * plane4] = v;
        }
    }
    if (W == 0 && acc.x == 12345.678f) *sink = acc.y;
}

template <int R, int W, int NT>
void run(const char* name, f4* pool, size_t pool4, size_t plane4, float* sink, int grid) {
    // a launch uses R + W planes; consecutive launches use consecutive plane groups of the pool
    const size_t group = (size_t)(R + W) * plane4;
    const int groups = (int)(pool4 / group);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto launch = [&](int i) {
        f4* base = pool + (size_t)(i % groups) * group;
        hipLaunchKernelGGL((k_stream<R, W, NT>), dim3(grid), dim3(256), 0, 0, base, base + (size_t)R * plane4, plane4, plane4, sink);
    };
    for (int i = 0; i < 3; ++i) launch(i);
    hipEventRecord(a);
    const int it = 12;
    for (int i = 0; i < it; ++i) launch(3 + i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %s grid %6d  %5.0f GB/s  (%.0f us per launch of %.2f GB)\n", name, NT ? "nt" : "  ", grid,
           (double)(R + W) * plane4 * 16 * it / ms / 1e6, ms / it * 1e3, (double)(R + W) * plane4 * 16 / 1e9);
    fflush(stdout);
}

int main() {
    const size_t plane4 = (size_t)1920 * 1080 * 32 / 4;  // one 32-frame 1080p plane, as float4
    const size_t pool4 = (size_t)8 << 30 >> 4;            // 8 GB
    f4* pool; float* sink;
    hipMalloc(&pool, pool4 * 16); hipMalloc(&sink, 4);
    hipMemset(pool, 0, pool4 * 16);
    for (int grid : {2048, 8192, 65536}) {
        run<1, 0, 0>("read only", pool, pool4, plane4, sink, grid);
        run<0, 1, 0>("write only", pool, pool4, plane4, sink, grid);
        run<0, 1, 1>("write only", pool, pool4, plane4, sink, grid);
        run<1, 1, 0>("copy 1:1", pool, pool4, plane4, sink, grid);
        run<1, 1, 1>("copy 1:1", pool, pool4, plane4, sink, grid);
        run<1, 6, 0>("1 read : 6 writes", pool, pool4, plane4, sink, grid);
        run<1, 6, 1>("1 read : 6 writes", pool, pool4, plane4, sink, grid);
        run<1, 3, 1>("1 read : 3 writes", pool, pool4, plane4, sink, grid);
        run<2, 1, 1>("2 reads : 1 write", pool, pool4, plane4, sink, grid);
        run<0, 6, 1>("6 writes", pool, pool4, plane4, sink, grid);
    }
    return 0;
}
