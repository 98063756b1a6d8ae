// What the FP4 matrix-core instruction of the matcher (v_mfma_scale_f32_32x32x64_f8f6f4, both operands e2m1) costs per
// SIMD on gfx950, alone and with the matcher's LDS operand reads / accumulator reads beside it.  One workgroup of 16
// waves per CU (four per SIMD), chains of eight dependent MFMAs as in k_match_fp4.
//   hipcc -O3 --offload-arch=gfx950 -o fp4_rate fp4_rate.hip && ./fp4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int PITCH = 272, ROWS = 128;

template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_rate(const uint8_t* __restrict__ src, float* __restrict__ sink, unsigned iters,
                                                     unsigned long long* __restrict__ cycles) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[ROWS * PITCH];
    const unsigned tid = threadIdx.x, lane = tid & 63u, r = lane & 31u, h = lane >> 5;
    for (unsigned i = tid; i < ROWS * PITCH / 16; i += blockDim.x)
        reinterpret_cast<uint4*>(s_tile)[i] = reinterpret_cast<const uint4*>(src)[i];
    __syncthreads();
    auto op = [](v4i x) { return v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0}; };
    v4i bq[8];
    for (int s = 0; s < 8; ++s) bq[s] = *reinterpret_cast<const v4i*>(src + (size_t)(tid & 127u) * PITCH + 32 * s);
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(bq[s]));
    float top = -1e30f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            v16f acc;
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            const uint8_t* arow = &s_tile[(32 * sub + r) * PITCH + 16 * h];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                v4i a;
                if (MODE == 0) {
                    a = bq[(s + 1) & 7];
                    asm volatile("" : "+v"(a));
                } else {
                    a = *reinterpret_cast<const v4i*>(arow + 32 * s);
                }
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op(a), op(bq[s]), acc, 4, 4, 0, 127, 0, 127);
            }
            if (MODE >= 1) {
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            if (MODE == 2) {
                float t = acc[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) t = fmaxf(t, acc[i]);
                top = fmaxf(top, t);
            } else {
                asm volatile("" ::"v"(acc[0]));
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (top == 12345.0f) sink[tid] = top;
    if (tid == 0) {  // the workgroup's FIRST wave: the SIMD issues its oldest wave first, so this one runs almost as if alone
        cycles[2 * blockIdx.x] = c1 - c0;
        cycles[2 * blockIdx.x + 1] = r1 - r0;  // 100 MHz ticks
    }
}

template <int MODE, int WAVES>
static void run(const char* what, const uint8_t* d_src, float* d_sink, unsigned long long* d_cyc, int blocks) {
    const unsigned iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d_src, d_sink, iters, d_cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_rate<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d_src, d_sink, iters, d_cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(2 * blocks);
    hipMemcpy(cyc.data(), d_cyc, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0, ticks = 0;
    for (int b = 0; b < blocks; ++b) { sum += (double)cyc[2 * b]; ticks += (double)cyc[2 * b + 1]; }
    const double pf = (double)blocks * WAVES * iters * 32.0 * 32 * 32 * 64 * 2 / (ms * 1e-3) / 1e15, mhz = 100.0 * sum / ticks;
    // every SIMD executes iters * 32 * WAVES / 4 MFMAs in the launch's `ms`: cycles per MFMA per SIMD at the in-kernel clock
    std::printf("%-44s %2d waves/CU: %.2f PFLOP/s; clock %4.0f MHz; %5.1f cycles per MFMA per SIMD (32 = the instruction); first wave alone-ish: %5.1f per own MFMA\n",
                what, WAVES, pf, mhz, ms * 1e-3 * mhz * 1e6 / ((double)iters * 32.0 * (WAVES / 4)), sum / blocks / ((double)iters * 32.0));
}

int main() {
    std::vector<uint8_t> h(ROWS * PITCH);
    unsigned x = 12345;
    for (auto& b : h) {  // +-1 e2m1 nibbles (0x2 = +1, 0xa = -1), as the matcher's operands
        x = x * 1664525u + 1013904223u;
        b = (uint8_t)(((x >> 16) & 1 ? 0x2 : 0xa) | (((x >> 17) & 1 ? 0x2 : 0xa) << 4));
    }
    uint8_t* d_src; float* d_sink; unsigned long long* d_cyc;
    hipMalloc(&d_src, h.size()); hipMalloc(&d_sink, 4096 * 4); hipMalloc(&d_cyc, 2048 * 8);
    hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 16>("MFMA chains, operands in registers", d_src, d_sink, d_cyc, 256);
        run<1, 16>("+ one ds_read_b128 per MFMA (3 ahead)", d_src, d_sink, d_cyc, 256);
        run<2, 16>("+ max over the 16 accumulators per chain", d_src, d_sink, d_cyc, 256);
        run<0, 4>("MFMA chains, registers, one wave per SIMD", d_src, d_sink, d_cyc, 256);
        run<1, 4>("+ ds_read_b128, one wave per SIMD", d_src, d_sink, d_cyc, 256);
        run<1, 8>("+ ds_read_b128, two waves per SIMD", d_src, d_sink, d_cyc, 256);
    }
    return 0;
}
