// k_ransac_trials: the RANSAC trials of match_features on the GPU (akaze/src/ops/estimate_fundamental_matrix.rs:99-165, called from
// akaze/src/lib.rs:267-274).  The reference runs num_trials x (8-point model + inlier count over every match) one after
// the other on one core; the host port of akz_ransac.cpp spreads the trials over host threads (1.3-1.4 ms for the 8 264
// matches of a 4K pair at 1 000 trials on 16 cores, of which 4.9 us per model in the 8 x 9 decomposition).  Here every
// trial is a workgroup: its first wave forms the model -- the same source as the host's (akz_fmatrix.hpp), f64 Jacobi
// rotations on the 8 x 9 matrix in LDS -- and the 256 threads count the inliers.  The samples are drawn on the
// host from the calling thread's random source in trial order, the winner is picked on the host in trial order with the
// reference's strict `>`, and the final filter runs on the host: same result as the host path, bit for bit.
#include <hip/hip_runtime.h>

#include "akz_fmatrix.hpp"
#include "akz_internal.hpp"

namespace akz {
namespace {

constexpr int RT = 256;

// pts: x0 | y0 | x1 | y1, n floats each; samples: 8 match indices per trial; out: per trial 9 floats (model) and the
// inlier count (-1: no model)
__global__ void __launch_bounds__(RT) k_ransac_trials(const float* __restrict__ pts, unsigned n, const unsigned* __restrict__ samples,
                                                      float epsilon_model, float epsilon_inlier, float* __restrict__ models,
                                                      int* __restrict__ inliers) {
    __shared__ float s_f[9];
    __shared__ int s_ok, s_cnt;
    __shared__ double s_m[8 * 9];
    struct LdsMat {
        double* p;
        __device__ double& at(int r, int k) { return p[r * 9 + k]; }
    };
    const unsigned trial = blockIdx.x, tid = threadIdx.x;
    const float *x0 = pts, *y0 = pts + n, *x1 = pts + 2 * (size_t)n, *y1 = pts + 3 * (size_t)n;
    // The model: the host's source (akz_fmatrix.hpp) on the matrix in LDS.  The host rotates the 28 row pairs of a sweep one
    // after the other in row-cyclic order; pairs that share no row commute, and every pair (p, q) depends only on pairs of
    // level p + q - 1 or less -- for two pairs with a common row the cyclic order and the level order agree -- so the 13
    // levels of a sweep run one after the other with the up to four pairs of a level on four lanes: the same rotations on
    // the same operands, bit for bit, along a chain of 13 instead of 28 (a rotation is ~2 000 cycles of dependent f64
    // arithmetic -- three square roots, three divisions -- and nothing else shortens a lone trial).
    if (tid < 64) {  // the workgroup's first wave; LDS operations of one wave execute in order
        LdsMat m{s_m};
        if (tid == 0) {
            float sx0[8], sy0[8], sx1[8], sy1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned j = samples[(size_t)trial * 8 + i];
                sx0[i] = x0[j]; sy0[i] = y0[j]; sx1[i] = x1[j]; sy1[i] = y1[j];
            }
            design_matrix(m, sx0, sy0, sx1, sy1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int sweep = 0; sweep < 60; ++sweep) {
            bool rotated = false;
            for (int level = 0; level <= 12; ++level) {
                const int p = max(0, level - 6) + (int)tid, q = level + 1 - p;  // lanes 0 .. 3: the level's pairs
                if (tid < 4 && p < q) rotated = jacobi_pair(m, p, q) || rotated;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (__ballot(rotated) == 0ull) break;
        }
        if (tid == 0) {
            float f[9];
            const bool ok = model_from_rotated(m, epsilon_model, f);
            s_ok = ok ? 1 : 0;
            s_cnt = 0;
            if (ok) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    s_f[k] = f[k];
                    models[(size_t)trial * 9 + k] = f[k];
                }
            }
        }
    }
    __syncthreads();
    if (!s_ok) {
        if (tid == 0) inliers[trial] = -1;
        return;
    }
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = s_f[k];
    int cnt = 0;
    for (unsigned i = tid; i < n; i += RT) cnt += fundamental_error(f, x0[i], y0[i], x1[i], y1[i]) < epsilon_inlier ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63u) == 0) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (tid == 0) inliers[trial] = s_cnt;
}

}  // namespace

namespace launch {
void ransac_trials(hipStream_t s, const float* d_pts, uint32_t n_matches, const uint32_t* d_samples, uint32_t trials, float epsilon_model,
                   float epsilon_inlier, float* d_models, int32_t* d_inliers) {
    if (trials == 0) return;
    hipLaunchKernelGGL(k_ransac_trials, dim3(trials), dim3(RT), 0, s, d_pts, n_matches, d_samples, epsilon_model, epsilon_inlier, d_models,
                       d_inliers);
}
}  // namespace launch
}  // namespace akz
