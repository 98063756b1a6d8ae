// HBM bandwidth of the COLUMN-MARCH access shape (tools only; not part of the product): a workgroup owns a strip of
// `cols` columns of one image and walks down a band of rows; per step it reads `rpi` rows of one plane and writes them
// to W planes, with no arithmetic.  What limits a marching kernel: the strip width, the rows in flight per step, the
// number of resident workgroups, alignment of the strips, streaming stores?
//   marchbw  -> table
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

template <int VEC> struct V;
template <> struct V<2> { typedef f2 t; typedef f2u u; };
template <> struct V<4> { typedef f4 t; typedef f4u u; };

// grid: n * nbands * nstrips workgroups (strips fastest); block: cols / VEC threads
template <int VEC, int W, int RPI, int NT, int PF>
__global__ void k_march(const float* __restrict__ in, float* __restrict__ out, int w, int h, size_t plane, int cols, int x_shift,
                        int nstrips, int nbands, int band_rows) {
    typedef typename V<VEC>::t vt;
    typedef typename V<VEC>::u vu;
    const int per = nstrips * nbands;
    const int img = __builtin_amdgcn_readfirstlane((int)blockIdx.x / per);
    const int rem = (int)blockIdx.x - img * per;
    const int band = __builtin_amdgcn_readfirstlane(rem / nstrips), strip = rem - band * nstrips;
    const int x = strip * cols + x_shift + VEC * (int)threadIdx.x;
    const bool ok = x >= 0 && x + VEC <= w;
    const int r0 = band * band_rows, r1 = min(r0 + band_rows, h);
    const float* src = in + (size_t)img * w * h;
    float* dst = out + (size_t)img * w * h;
    const unsigned boff = (unsigned)(ok ? x : 0) * 4u;
    vt q[PF][RPI];
    auto ld = [&](int r, vt (&v)[RPI]) {
#pragma unroll
        for (int j = 0; j < RPI; ++j) {
            const int rr = min(r + j, h - 1);
            v[j] = *reinterpret_cast<const vu*>(reinterpret_cast<const char*>(src + (size_t)rr * w) + boff);
        }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) ld(r0 + i * RPI, q[i]);
    for (int r = r0; r < r1; r += RPI) {
        vt nxt[RPI];
        ld(r + PF * RPI, nxt);
        if (ok) {
#pragma unroll
            for (int j = 0; j < RPI; ++j) {
                if (r + j < r1) {
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        vu* p = reinterpret_cast<vu*>(reinterpret_cast<char*>(dst + k * plane + (size_t)(r + j) * w) + boff);
                        if (NT) __builtin_nontemporal_store((vu)q[0][j], p);
                        else *p = q[0][j];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i + 1 < PF; ++i)
#pragma unroll
            for (int j = 0; j < RPI; ++j) q[i][j] = q[i + 1][j];
#pragma unroll
        for (int j = 0; j < RPI; ++j) q[PF - 1][j] = nxt[j];
    }
}

static int g_inputs = 4;  // launches rotate over this many input planes: 1 = the input stays in the 256 MB Infinity Cache
template <int VEC, int W, int RPI, int NT, int PF>
void run(const float* in, float* out, int w, int h, int n, int cols, int x_shift, int wgs_per_cu) {
    const size_t plane = (size_t)w * h * n;
    const int nstrips = (w + cols - 1) / cols;
    const long colsN = (long)n * nstrips;
    long nb = std::max<long>(1, (256L * wgs_per_cu + colsN - 1) / colsN);
    int band_rows = (int)((h + nb - 1) / nb);
    band_rows = (band_rows + RPI - 1) / RPI * RPI;
    const int nbands = (h + band_rows - 1) / band_rows;
    const int grid = (int)(colsN * nbands), block = cols / VEC;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((k_march<VEC, W, RPI, NT, PF>), dim3(grid), dim3(block), 0, 0, in + (size_t)(i % g_inputs) * plane, out, w, h, plane, cols, x_shift, nstrips, nbands, band_rows);
    hipEventRecord(a);
    const int it = 8;
    for (int i = 0; i < it; ++i)
        hipLaunchKernelGGL((k_march<VEC, W, RPI, NT, PF>), dim3(grid), dim3(block), 0, 0, in + (size_t)(i % g_inputs) * plane, out, w, h, plane, cols, x_shift, nstrips, nbands, band_rows);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)plane * 4 * (1 + W) * it;
    printf("march %s R1:W%d vec%d cols %4d shift %3d rows/step %d pf %d %s wg/cu %d (grid %5d x %3d thr, band %4d rows)  %5.0f GB/s  (%.0f us)\n", g_inputs > 1 ? "input from HBM " : "input from cache", W, VEC, cols,
           x_shift, RPI, PF, NT ? "nt" : "  ", wgs_per_cu, grid, block, band_rows, bytes / ms / 1e6, ms / it * 1e3);
    fflush(stdout);
}

int main() {
    const int w = 1920, h = 1080, n = 32;
    const size_t plane = (size_t)w * h * n;
    float *in, *out;
    hipMalloc(&in, plane * 4 * 4); hipMalloc(&out, plane * 4 * 6);
    hipMemset(in, 0, plane * 4 * 4); hipMemset(out, 0, plane * 4 * 6);
    // A 32-frame 1080p plane is 265 MB: launches that re-read ONE input plane find most of it in the Infinity Cache and
    // report 20-35 % more than the pyramid sees, where every launch reads a plane written long ago.
    g_inputs = 1;
    run<2, 6, 1, 1, 2>(in, out, w, h, n, 480, 0, 3);
    run<2, 6, 4, 1, 2>(in, out, w, h, n, 480, 0, 3);
    g_inputs = 4;
    // the shape of k_detector_march today: 512-column strips (492 useful), 2 columns per thread, one row per step
    run<2, 6, 1, 0, 2>(in, out, w, h, n, 492, -10, 3);
    run<2, 6, 1, 0, 2>(in, out, w, h, n, 480, 0, 3);      // aligned strips
    run<2, 6, 1, 1, 2>(in, out, w, h, n, 480, 0, 3);      // + streaming stores
    run<4, 6, 1, 0, 2>(in, out, w, h, n, 480, 0, 3);      // 16 bytes per lane (120 threads)
    run<4, 6, 1, 0, 2>(in, out, w, h, n, 960, 0, 3);      // wider strips
    run<4, 6, 1, 0, 2>(in, out, w, h, n, 1920, 0, 3);     // whole rows
    for (int f : {1, 2, 4, 6, 8}) run<2, 6, 1, 0, 2>(in, out, w, h, n, 480, 0, f);
    run<2, 6, 2, 0, 2>(in, out, w, h, n, 480, 0, 3);      // more rows in flight per step
    run<2, 6, 4, 0, 2>(in, out, w, h, n, 480, 0, 3);
    run<2, 6, 4, 1, 2>(in, out, w, h, n, 480, 0, 3);
    run<2, 6, 8, 0, 1>(in, out, w, h, n, 480, 0, 3);
    run<2, 6, 4, 0, 2>(in, out, w, h, n, 480, 0, 6);
    run<4, 6, 4, 0, 2>(in, out, w, h, n, 1920, 0, 3);
    run<2, 3, 1, 0, 2>(in, out, w, h, n, 480, 0, 3);      // lean mix
    run<2, 3, 4, 0, 2>(in, out, w, h, n, 480, 0, 3);
    return 0;
}
