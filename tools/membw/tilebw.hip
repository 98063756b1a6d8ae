// HBM bandwidth of TILED plane traffic (tools only; not part of the product): every workgroup of 512 threads moves
// TW x TH tiles of R input planes to W output planes of a batch of 1920x1080 frames with no arithmetic, in the access
// shapes the stencil kernels use (one dword per lane and row, or one float4 per lane), one tile per workgroup or a
// persistent grid.  What this reaches is the ceiling of a tiled kernel with that read:write mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4_native __attribute__((ext_vector_type(4)));

template <int R, int W, int TW, int TH, int VEC, int NTS = 0>
__global__ void __launch_bounds__(512) k_tile(const float* __restrict__ in, float* __restrict__ out, int w, int h, int n,
                                               size_t plane, int ntx, int nty) {
    const int ntiles = ntx * nty * n;
    constexpr int CW = TW / VEC;          // lanes per tile row
    constexpr int ROWS = 512 / CW;        // tile rows covered per pass
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int bz = t / (ntx * nty), rem = t - bz * ntx * nty, by = rem / ntx, bx = rem - by * ntx;
        const int x = bx * TW + tx * VEC;
        const size_t base = (size_t)bz * w * h;
#pragma unroll
        for (int r0 = 0; r0 < TH; r0 += ROWS) {
            const int y = by * TH + r0 + ty;
            if (ty >= TH || y >= h || x >= w) continue;
            const size_t g = base + (size_t)y * w + x;
            if (VEC == 4) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float4 v;
                    if (NTS & 2) {
                        const f4_native t = __builtin_nontemporal_load(reinterpret_cast<const f4_native*>(in + r * plane + g));
                        v = make_float4(t.x, t.y, t.z, t.w);
                    } else {
                        v = *reinterpret_cast<const float4*>(in + r * plane + g);
                    }
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    acc.x += 1.0f;
                    if (NTS & 1) { f4_native t = {acc.x, acc.y, acc.z, acc.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4_native*>(out + k * plane + g)); }
                    else *reinterpret_cast<float4*>(out + k * plane + g) = acc;
                }
            } else {
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) acc += (NTS & 2) ? __builtin_nontemporal_load(in + r * plane + g) : in[r * plane + g];
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    acc += 1.0f;
                    if (NTS & 1) __builtin_nontemporal_store(acc, out + k * plane + g);
                    else out[k * plane + g] = acc;
                }
            }
        }
    }
}

template <int R, int W, int TW, int TH, int VEC, int NTS = 0>
void run(const float* in, float* out, int w, int h, int n, int blocks) {
    const int ntx = (w + TW - 1) / TW, nty = (h + TH - 1) / TH;
    const size_t plane = (size_t)w * h * n;
    const int grid = blocks > 0 ? blocks : ntx * nty * n;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((k_tile<R, W, TW, TH, VEC, NTS>), dim3(grid), dim3(512), 0, 0, in, out, w, h, n, plane, ntx, nty);
    hipEventRecord(a);
    const int it = 10;
    for (int i = 0; i < it; ++i)
        hipLaunchKernelGGL((k_tile<R, W, TW, TH, VEC, NTS>), dim3(grid), dim3(512), 0, 0, in, out, w, h, n, plane, ntx, nty);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)plane * 4 * (R + W) * it;
    if (NTS) printf("[nontemporal %s%s] ", (NTS & 1) ? "st" : "", (NTS & 2) ? "ld" : "");
    printf("R%d:W%d tile %3dx%-2d vec%d grid %6d  %5.0f GB/s  (%.0f us)\n", R, W, TW, TH, VEC, grid, bytes / ms / 1e6, ms / it * 1e3);
}

template <int R, int W>
void shapes(const float* in, float* out, int w, int h, int n) {
    for (int blocks : {0, 1024, 768, 512}) {
        run<R, W, 64, 32, 1>(in, out, w, h, n, blocks);
        run<R, W, 64, 32, 4>(in, out, w, h, n, blocks);
        run<R, W, 128, 16, 1>(in, out, w, h, n, blocks);
        run<R, W, 128, 16, 4>(in, out, w, h, n, blocks);
        run<R, W, 256, 8, 4>(in, out, w, h, n, blocks);
        run<R, W, 512, 4, 4>(in, out, w, h, n, blocks);
        run<R, W, 1920, 1, 4>(in, out, w, h, n, blocks);
    }
}

// The FED access shape: R planes read as (TW + 2*HX) x (TH + 2*HY) regions in float4 groups (HX a multiple of 4, so the
// region starts 16 bytes before a 256-byte tile row), one plane written as TW x TH tiles.  No arithmetic.
template <int R, int TW, int TH, int HX, int HY>
__global__ void __launch_bounds__(512) k_halo(const float* __restrict__ in, float* __restrict__ out, int w, int h, int n,
                                               size_t plane, int ntx, int nty) {
    constexpr int XG = (TW + 2 * HX) / 4, RH = TH + 2 * HY;
    const int t = blockIdx.x;
    const int bz = t / (ntx * nty), rem = t - bz * ntx * nty, by = rem / ntx, bx = rem - by * ntx;
    const size_t base = (size_t)bz * w * h;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int idx = threadIdx.x; idx < XG * RH; idx += 512) {
        const int ly = idx / XG, g = idx - ly * XG;
        const int gy = by * TH - HY + ly, gx = bx * TW - HX + 4 * g;
        if (gy < 0 || gy >= h || gx < 0 || gx >= w) continue;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(in + r * plane + base + (size_t)gy * w + gx);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    // every thread of the first TW/4 * TH writes one centre group
    const int cw = TW / 4;
    for (int idx = threadIdx.x; idx < cw * TH; idx += 512) {
        const int ly = idx / cw, g = idx - ly * cw;
        const int gy = by * TH + ly, gx = bx * TW + 4 * g;
        if (gy < h && gx < w) *reinterpret_cast<float4*>(out + base + (size_t)gy * w + gx) = acc;
    }
}
template <int R, int TW, int TH, int HX, int HY>
void run_halo(const float* in, float* out, int w, int h, int n) {
    const int ntx = (w + TW - 1) / TW, nty = (h + TH - 1) / TH;
    const size_t plane = (size_t)w * h * n;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((k_halo<R, TW, TH, HX, HY>), dim3(ntx * nty * n), dim3(512), 0, 0, in, out, w, h, n, plane, ntx, nty);
    hipEventRecord(a);
    const int it = 10;
    for (int i = 0; i < it; ++i)
        hipLaunchKernelGGL((k_halo<R, TW, TH, HX, HY>), dim3(ntx * nty * n), dim3(512), 0, 0, in, out, w, h, n, plane, ntx, nty);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("halo R%d:W1 tile %3dx%-2d halo x%d y%d: %.0f us per launch, %.0f GB/s algorithmic (tile bytes only)\n", R, TW, TH, HX, HY,
           ms / it * 1e3, (double)plane * 4 * (R + 1) * it / ms / 1e6);
}

// Sustained rate: the same launch back to back for `seconds`, reported per window of 50 launches (does the rate of a
// cold chip hold under continuous load?)
template <int R, int W>
void sustained(const float* in, float* out, int w, int h, int n, double seconds) {
    constexpr int TW = 64, TH = 32;
    const int ntx = (w + TW - 1) / TW, nty = (h + TH - 1) / TH;
    const size_t plane = (size_t)w * h * n;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    double total_ms = 0;
    printf("sustained R%d:W%d 64x32 vec1, one tile per workgroup:", R, W);
    while (total_ms < seconds * 1e3) {
        hipEventRecord(a);
        for (int i = 0; i < 50; ++i)
            hipLaunchKernelGGL((k_tile<R, W, TW, TH, 1>), dim3(ntx * nty * n), dim3(512), 0, 0, in, out, w, h, n, plane, ntx, nty);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        total_ms += ms;
        printf(" %.0f", (double)plane * 4 * (R + W) * 50 / ms / 1e6);
    }
    printf(" GB/s\n");
}

int main(int argc, char** argv) {
    const int w = 1920, h = 1080, n = 32;
    const size_t plane = (size_t)w * h * n;
    float *in, *out;
    hipMalloc(&in, plane * 4 * 2); hipMalloc(&out, plane * 4 * 6);
    hipMemset(in, 0, plane * 4 * 2); hipMemset(out, 0, plane * 4 * 6);
    if (argc > 1 && argv[1][0] == 'h') {  // tilebw halo
        run_halo<2, 64, 48, 0, 0>(in, out, w, h, n);
        run_halo<2, 64, 48, 4, 0>(in, out, w, h, n);
        run_halo<2, 64, 48, 0, 3>(in, out, w, h, n);
        run_halo<2, 64, 48, 4, 3>(in, out, w, h, n);
        run_halo<2, 64, 32, 8, 8>(in, out, w, h, n);
        run_halo<2, 128, 24, 4, 3>(in, out, w, h, n);
        run_halo<2, 256, 16, 4, 3>(in, out, w, h, n);
        run_halo<1, 64, 48, 4, 3>(in, out, w, h, n);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'n') {  // tilebw nontemporal: do streaming stores / loads lift the ceiling?
        for (int blocks : {0, 1024}) {
            run<1, 2, 64, 32, 1, 0>(in, out, w, h, n, blocks);
            run<1, 2, 64, 32, 1, 1>(in, out, w, h, n, blocks);
            run<1, 2, 64, 32, 1, 3>(in, out, w, h, n, blocks);
            run<1, 2, 64, 32, 4, 0>(in, out, w, h, n, blocks);
            run<1, 2, 64, 32, 4, 1>(in, out, w, h, n, blocks);
            run<2, 4, 64, 32, 1, 0>(in, out, w, h, n, blocks);
            run<2, 4, 64, 32, 1, 1>(in, out, w, h, n, blocks);
            run<2, 4, 64, 32, 1, 3>(in, out, w, h, n, blocks);
            run<2, 4, 64, 32, 4, 0>(in, out, w, h, n, blocks);
            run<2, 4, 64, 32, 4, 1>(in, out, w, h, n, blocks);
            run<2, 4, 64, 32, 4, 3>(in, out, w, h, n, blocks);
            run<1, 6, 64, 32, 1, 0>(in, out, w, h, n, blocks);
            run<1, 6, 64, 32, 1, 1>(in, out, w, h, n, blocks);
        }
        return 0;
    }
    if (argc > 1) {  // tilebw sustained
        sustained<2, 4>(in, out, w, h, n, 4.0);
        sustained<1, 2>(in, out, w, h, n, 4.0);
        return 0;
    }
    shapes<1, 2>(in, out, w, h, n);   // k_deriv1, k_prep
    shapes<2, 4>(in, out, w, h, n);   // k_deriv2, all planes kept
    shapes<1, 6>(in, out, w, h, n);   // one-kernel detector
    shapes<2, 1>(in, out, w, h, n);   // FED launch (Lt, Lflow -> Lt')
    return 0;
}
