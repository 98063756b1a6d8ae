// Host post-filter of match_features: 8-point fundamental matrix + RANSAC
// (akaze/src/ops/estimate_fundamental_matrix.rs:17-165, called from akaze/src/lib.rs:267-274).
// a tiny host step after the GPU matcher, outside the GPU path.
//
// Reference behaviour reproduced here, quirks included:
//   * fewer than 8 matches: returned unchanged (:107-110);
//   * the design matrix row is [x0*x1, x0*y1, x0, y0*x1, y0*y1, y0, x1, y1, 1] (:26-40) with (x0, y0)
//     from keypoints_0 and (x1, y1) from keypoints_1;
//   * nalgebra's SVD of the 8x9 matrix yields 8 singular values / right vectors; the model is the
//     right singular vector of the SMALLEST OF THOSE 8 (:46-53) — not the null vector of the 9x9
//     problem — accepted only if all 8 singular values exceed epsilon_model (rank == 8, :44);
//   * F = [[v0, v3, v6], [v1, v4, v7], [v2, v5, v8]] (:55-66); error = |p_r^T F p_l| (:79-83);
//   * a model replaces the best one only with strictly more inliers; if no trial yields a model the
//     final model is the zero matrix, whose error is 0, so every match is kept (:112-113, :152-163);
//   * every trial takes `random::default()` (:118).  In the `random` crate 0.12 that is a handle to a
//     THREAD-LOCAL Xorshift128+ source seeded [42, 69], not a fresh generator: successive trials (and later
//     calls, and the debug drawings' random_color) continue one stream.  The crate is not in the reference
//     tree; generator, seed and the persistence are pinned by the reference's own published output
//     test-data/keypoints-1.jpg, whose i-th keypoint disc has the colour of stream values 3i..3i+2
//     (tests/test_reference_outputs.py).  akz_random_seed reseeds the calling thread's source.
// Not reproducible bit for bit (and not a parity target): the HashSet iteration order of the
// 8 sampled indices (random per process) and nalgebra's f32 SVD.  Here the sample is taken in ascending
// index order and the decomposition is a one-sided Jacobi SVD of the 8x9 matrix in f64.
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "akz_fmatrix.hpp"
#include "akz_internal.hpp"
#include "akz_pool.hpp"

namespace akz {

// `random::default()` of the `random` crate 0.12: one Xorshift128+ source per thread, seeded [42, 69]
DefaultSource& default_source() {
    static thread_local DefaultSource src;
    return src;
}

namespace {

struct Model {
    float f[9];  // row-major 3 x 3
};

// estimate_fundamental_matrix (:17-69): the model itself is akz_fmatrix.hpp (shared with the device kernel)
bool estimate(const akz_keypoint* k0, const akz_keypoint* k1, const akz_match* sample, float epsilon, Model& out) {
    float x0[8], y0[8], x1[8], y1[8];
    for (int i = 0; i < 8; ++i) {
        x0[i] = k0[sample[i].index_0].x; y0[i] = k0[sample[i].index_0].y;
        x1[i] = k1[sample[i].index_1].x; y1[i] = k1[sample[i].index_1].y;
    }
    return fundamental_from_8(x0, y0, x1, y1, epsilon, out.f);
}

// evaluate_model (:79-83): |p_r^T F p_l|
float model_error(const Model& md, const akz_keypoint& k0, const akz_keypoint& k1) {
    return fundamental_error(md.f, k0.x, k0.y, k1.x, k1.y);
}

}  // namespace
}  // namespace akz

using namespace akz;

// trials_on_device (match_features with a context): runs the trials elsewhere -- px0 .. py1 (n_matches floats each), the
// samples (8 per trial), -> models (9 floats per trial), inliers (-1: no model); AKZ_OK or an error (the host path then
// takes over)
int akz::remove_outliers_impl(const akz_keypoint* keypoints_0, uint64_t n0, const akz_keypoint* keypoints_1, uint64_t n1,
                              const akz_match* matches, uint64_t n_matches, uint64_t num_trials, float epsilon_model,
                              float epsilon_inlier, akz_match* out, uint64_t* n_out, const TrialsOnDevice& trials_on_device) {
    if (!n_out || (n_matches && (!matches || !out))) {
        set_error("remove_outliers: null pointer");
        return AKZ_ERR_INVALID_ARG;
    }
    for (uint64_t i = 0; i < n_matches; ++i)
        if (matches[i].index_0 >= n0 || matches[i].index_1 >= n1 || !keypoints_0 || !keypoints_1) {
            set_error("remove_outliers: match index out of range");
            return AKZ_ERR_INVALID_ARG;
        }
    if (n_matches < 8) {  // "Not enough points to do RANSAC."
        if (n_matches) std::memcpy(out, matches, n_matches * sizeof(akz_match));
        *n_out = n_matches;
        return AKZ_OK;
    }
    // The trials are independent once their samples are drawn: the samples come from the thread's random source in trial
    // order (as the sequential loop of :111-147 draws them), the models and inlier counts are computed on a few host
    // threads, and the winner is picked in trial order with the reference's strict `>` -- the same model as the
    // sequential loop returns (12 ms -> 1.5 ms for the 8 000 matches of a 4K pair at 1 000 trials).
    DefaultSource& src = default_source();
    std::vector<uint64_t> samples((size_t)num_trials * 8);
    {
        std::vector<uint64_t> picked;
        for (uint64_t trial = 0; trial < num_trials; ++trial) {
            picked.clear();  // `set.insert(source.read::<usize>() % matches.len())` until 8 distinct indices (:117-121)
            while (picked.size() < 8) {
                const uint64_t j = src.next() % n_matches;
                if (std::find(picked.begin(), picked.end(), j) == picked.end()) picked.push_back(j);
            }
            std::sort(picked.begin(), picked.end());  // the reference iterates the HashSet: arbitrary order
            std::copy(picked.begin(), picked.end(), samples.begin() + (size_t)trial * 8);
        }
    }
    std::vector<Model> models((size_t)num_trials);
    std::vector<int64_t> inliers((size_t)num_trials, -1);  // -1: no model (rank-deficient sample)
    // the matched point pairs side by side (x0, y0, x1, y1 as four arrays): the inlier count of a trial -- model_error over
    // every match, 8 M evaluations for the 8 000 matches of a 4K pair at 1 000 trials -- is then a streaming loop the
    // compiler vectorises instead of two gathers of 32-byte keypoints per evaluation.  Same expression per match, same
    // operation order (pr[2] = pl[2] = 1.0f: the products with them are exact).
    std::vector<float> px0((size_t)n_matches), py0((size_t)n_matches), px1((size_t)n_matches), py1((size_t)n_matches);
    for (uint64_t i = 0; i < n_matches; ++i) {
        px0[(size_t)i] = keypoints_0[matches[i].index_0].x; py0[(size_t)i] = keypoints_0[matches[i].index_0].y;
        px1[(size_t)i] = keypoints_1[matches[i].index_1].x; py1[(size_t)i] = keypoints_1[matches[i].index_1].y;
    }
    auto run_trials = [&](uint64_t lo, uint64_t hi) {
        const float *x0 = px0.data(), *y0 = py0.data(), *x1 = px1.data(), *y1 = py1.data();
        for (uint64_t trial = lo; trial < hi; ++trial) {
            akz_match sample[8];
            for (int i = 0; i < 8; ++i) sample[i] = matches[samples[(size_t)trial * 8 + i]];
            Model model;
            if (!estimate(keypoints_0, keypoints_1, sample, epsilon_model, model)) continue;
            int64_t inl = 0;
            for (uint64_t i = 0; i < n_matches; ++i)  // model_error(model, k0, k1) < epsilon_inlier
                inl += fundamental_error(model.f, x0[i], y0[i], x1[i], y1[i]) < epsilon_inlier ? 1 : 0;
            models[(size_t)trial] = model;
            inliers[(size_t)trial] = inl;
        }
    };
    bool on_device = false;
    if (trials_on_device && num_trials <= 0x7fffffffull && n_matches <= 0x7fffffffull) {
        std::vector<uint32_t> smp((size_t)num_trials * 8);
        for (size_t i = 0; i < smp.size(); ++i) smp[i] = (uint32_t)samples[i];
        std::vector<float> mdl((size_t)num_trials * 9);
        std::vector<int32_t> inl((size_t)num_trials);
        if (trials_on_device(px0.data(), py0.data(), px1.data(), py1.data(), (uint32_t)n_matches, smp.data(), (uint32_t)num_trials,
                             epsilon_model, epsilon_inlier, mdl.data(), inl.data()) == AKZ_OK) {
            for (uint64_t t = 0; t < num_trials; ++t) {
                inliers[(size_t)t] = inl[(size_t)t];
                if (inl[(size_t)t] >= 0) std::memcpy(models[(size_t)t].f, &mdl[(size_t)t * 9], sizeof(float) * 9);
            }
            on_device = true;
        }
    }
    // (a model costs ~5-9 us -- the 8 x 9 singular value decomposition -- and an inlier count ~0.3-1 ns per match)
    const unsigned nthreads = (unsigned)std::min<uint64_t>(std::min(host_cpu_share(), 16u),
                                                          std::max<uint64_t>(1, num_trials * (n_matches + 9000) / 2000000));  // ~0.1 ms of work per thread at least
    if (on_device) {
        // (done)
    } else if (nthreads <= 1) {
        run_trials(0, num_trials);
    } else {
        // the trials go to a pool of host threads that lives as long as the process (starting 16 threads per call was a
        // third of a 1 000-trial call's 1.3 ms); a second caller at the same time starts its own threads, as before
        // The pool is never destroyed (its threads end with the process) and belongs to the process that made it: after a
        // fork() the child inherits the object but none of its threads, so a pid that differs drops it (leaked: joining
        // threads that do not exist would hang) and starts a new one.  A child forked while another thread held pool_m
        // never gets the lock and takes the per-call threads below.
        static std::mutex pool_m;
        static WorkerPool* pool = nullptr;
        static pid_t pool_pid = 0;
        std::unique_lock<std::mutex> lk(pool_m, std::try_to_lock);
        if (lk.owns_lock()) {
            const pid_t me = getpid();
            if (pool && pool_pid != me) pool = nullptr;
            if (!pool || pool->size() < nthreads) {
                if (pool && pool_pid == me) delete pool;
                pool = new WorkerPool(nthreads - 1);
                pool_pid = me;
            }
            // handed out dynamically, four pieces per thread -- unless the pool is wider than this call wants (it was grown by
            // a larger one): then exactly nthreads pieces, so that a small call does not wake sixteen threads
            const uint64_t per = std::max<uint64_t>(1, num_trials / ((pool->size() > nthreads ? 1ull : 4ull) * nthreads));
            const size_t pieces = (size_t)((num_trials + per - 1) / per);
            pool->run(pieces, [&](size_t i) { run_trials((uint64_t)i * per, std::min<uint64_t>(num_trials, ((uint64_t)i + 1) * per)); });
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthreads; ++t)
                th.emplace_back(run_trials, num_trials * t / nthreads, num_trials * (t + 1) / nthreads);
            for (auto& t : th) t.join();
        }
    }
    int64_t max_inliers = 0;
    Model final_model;
    std::memset(&final_model, 0, sizeof(final_model));
    for (uint64_t trial = 0; trial < num_trials; ++trial)
        if (inliers[(size_t)trial] > max_inliers) {
            max_inliers = inliers[(size_t)trial];
            final_model = models[(size_t)trial];
        }
    uint64_t k = 0;
    for (uint64_t i = 0; i < n_matches; ++i)
        if (model_error(final_model, keypoints_0[matches[i].index_0], keypoints_1[matches[i].index_1]) <
            epsilon_inlier)
            out[k++] = matches[i];
    *n_out = k;
    return AKZ_OK;
}

extern "C" int akz_remove_outliers(const akz_keypoint* keypoints_0, uint64_t n0, const akz_keypoint* keypoints_1,
                                   uint64_t n1, const akz_match* matches, uint64_t n_matches, uint64_t num_trials,
                                   float epsilon_model, float epsilon_inlier, akz_match* out, uint64_t* n_out) {
    return akz::remove_outliers_impl(keypoints_0, n0, keypoints_1, n1, matches, n_matches, num_trials, epsilon_model, epsilon_inlier, out,
                                     n_out, akz::TrialsOnDevice());
}

// ops::estimate_fundamental_matrix::estimate_fundamental_matrix (:17-69) for exactly 8 matches: *found = 0 is the
// reference's `None` (rank < 8 at `epsilon`); f = the 3x3 matrix, row-major
extern "C" int akz_estimate_fundamental_matrix(const akz_keypoint* keypoints_0, uint64_t n0, const akz_keypoint* keypoints_1,
                                               uint64_t n1, const akz_match* matches8, float epsilon, float* f, int* found) {
    if (!keypoints_0 || !keypoints_1 || !matches8 || !f || !found) {
        set_error("estimate_fundamental_matrix: null pointer");
        return AKZ_ERR_INVALID_ARG;
    }
    for (int i = 0; i < 8; ++i)
        if (matches8[i].index_0 >= n0 || matches8[i].index_1 >= n1) {
            set_error("estimate_fundamental_matrix: match index out of range");
            return AKZ_ERR_INVALID_ARG;
        }
    Model md;
    *found = estimate(keypoints_0, keypoints_1, matches8, epsilon, md) ? 1 : 0;
    if (*found) std::memcpy(f, md.f, sizeof(md.f));
    return AKZ_OK;
}

// random::default().seed([s0, s1]) for the calling thread
extern "C" int akz_random_seed(uint64_t s0, uint64_t s1) {
    DefaultSource& src = default_source();
    src.s0 = s0;
    src.s1 = s1;
    return AKZ_OK;
}
