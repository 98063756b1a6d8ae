// HBM bandwidth by read:write mix on one GPU (tools only; not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int R, int W>
__global__ void __launch_bounds__(256) k_mix(const float4* __restrict__ in, float4* __restrict__ out, size_t n, size_t stride) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 v = in[i + r * stride];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (W == 0) {
            if (acc.x == 123.456f) out[i] = acc;  // keep the loads alive
        }
#pragma unroll
        for (int wv = 0; wv < W; ++wv) {
            acc.x += 1.0f;
            out[i + wv * stride] = acc;
        }
    }
}
template <int R, int W>
void run(const float4* in, float4* out, size_t n, size_t stride, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_mix<R, W>), dim3(blocks), dim3(256), 0, 0, in, out, n, stride);
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k_mix<R, W>), dim3(blocks), dim3(256), 0, 0, in, out, n, stride);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)n * 16 * (R + W) * it;
    printf("R%d:W%d blocks %5d  %.0f GB/s\n", R, W, blocks, bytes / ms / 1e6);
}
int main() {
    const size_t n = 16u << 20;  // float4 per plane: 256 MiB
    const size_t planes = 4;
    float4 *in, *out;
    hipMalloc(&in, n * planes * 16); hipMalloc(&out, n * planes * 16);
    hipMemset(in, 0, n * planes * 16); hipMemset(out, 0, n * planes * 16);
    const size_t m = n - (1u << 20);  // elements per launch; leaves room for odd plane strides
    for (size_t stride : {n, n - 4096 + 48, n - 77777, (size_t)(1920 * 1080 * 32 / 4) / 4}) {
        printf("plane stride %zu float4 (%zu B)\n", stride, stride * 16);
        for (int blocks : {2048, 65536}) {
            run<2, 0>(in, out, m, stride, blocks); run<0, 2>(in, out, m, stride, blocks); run<0, 4>(in, out, m, stride, blocks);
            run<1, 2>(in, out, m, stride, blocks); run<2, 4>(in, out, m, stride, blocks); run<1, 3>(in, out, m, stride, blocks);
        }
    }
    return 0;
}
