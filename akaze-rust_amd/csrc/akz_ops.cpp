// C ABI of libakaze_hip.so, part 2: the per-op entry points on device planes (mirror of `pub mod ops` / `types::image`) and
// the launch helpers -- blur, Scharr, contrast factor, FED, detector -- that the extraction pipeline shares with them.
#include "akz_ctx.hpp"

// ---------------------------------------------------------------------------------------------
// per-op entry points
// ---------------------------------------------------------------------------------------------
static int check_plane_args(const void* a, const void* b, uint32_t w, uint32_t h, uint32_t n, int hw) {
    if (!a || !b || w == 0 || h == 0 || n == 0) {
        set_error("null plane pointer or empty image");
        return AKZ_ERR_INVALID_ARG;
    }
    if ((int)w < 2 * hw + 1 || (int)h < 2 * hw + 1) {
        set_error("image smaller than the filter kernel");
        return AKZ_ERR_TOO_SMALL;
    }
    return AKZ_OK;
}

template <typename T>
int gaussian_blur_impl(akz_ctx* c, const T* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                              float sigma) {
    const size_t ks = gaussian_kernel_size(sigma);
    if (ks > (size_t)kMaxTaps || !(sigma > 0.0f)) {
        set_error("gaussian_blur: sigma must be in (0, 6]");
        return AKZ_ERR_INVALID_ARG;
    }
    const std::vector<float> k = gaussian_kernel(sigma, ks);
    Taps t;
    AKZ_TRY(taps_from_dense(k.data(), (uint32_t)k.size(), t));
    AKZ_TRY(check_plane_args(d_in, d_out, w, h, n, t.hw));
    constexpr bool is_u8 = std::is_same<T, uint8_t>::value;
    // large batches: the column march (HBM-bound: 0.33 GB of a 32-frame 1080p batch)
    const uint64_t blur_march_min_px = c->launch_min_px;
    if ((const void*)d_in != (const void*)d_out && (c->prep_mode == 3 || (c->prep_mode == 2 && (uint64_t)w * h * n >= blur_march_min_px)) &&
        launch::blur5_march_supported(w, h, (uint32_t)k.size())) {
        if constexpr (is_u8) launch::blur5_march_u8(c->stream, d_in, d_out, w, h, n, k.data());
        else launch::blur5_march_f32(c->stream, d_in, d_out, w, h, n, k.data());
        AKZ_HIP_TRY(hipGetLastError());
        return AKZ_OK;
    }
    if ((const void*)d_in != (const void*)d_out && c->prep_mode != 0 &&
        launch::blur5_stream_supported(w, h, (uint32_t)k.size(), is_u8) &&
        (!is_u8 || ((uintptr_t)d_in & 3u) == 0) && (c->prep_mode == 1 || (uint64_t)w * h * n >= c->stream_min_px)) {
        if constexpr (is_u8) launch::blur5_stream_u8(c->stream, d_in, d_out, w, h, n, k.data());
        else launch::blur5_stream_f32(c->stream, d_in, d_out, w, h, n, k.data());
        AKZ_HIP_TRY(hipGetLastError());
        return AKZ_OK;
    }
    if (launch::blur_fused_supported((uint32_t)k.size()) && (const void*)d_in != (const void*)d_out) {
        if constexpr (std::is_same<T, uint8_t>::value)
            launch::blur_fused_u8(c->stream, d_in, d_out, w, h, n, k.data(), (uint32_t)k.size());
        else
            launch::blur_fused_f32(c->stream, d_in, d_out, w, h, n, k.data(), (uint32_t)k.size());
        AKZ_HIP_TRY(hipGetLastError());
        return AKZ_OK;
    }
    AKZ_TRY(ensure(c, c->scratch[0], plane_bytes(w, h, n)));
    float* tmp = (float*)c->scratch[0].p;
    if constexpr (std::is_same<T, uint8_t>::value) launch::filter_h_u8(c->stream, d_in, tmp, w, h, n, t);
    else launch::filter_h_f32(c->stream, d_in, tmp, w, h, n, t);
    launch::filter_v_f32(c->stream, tmp, d_out, w, h, n, t);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
template int gaussian_blur_impl<float>(akz_ctx*, const float*, float*, uint32_t, uint32_t, uint32_t, float);
template int gaussian_blur_impl<uint8_t>(akz_ctx*, const uint8_t*, float*, uint32_t, uint32_t, uint32_t, float);

// scharr_horizontal / scharr_vertical (derivatives.rs:41-65) through scratch[0]
static int scharr_impl(akz_ctx* c, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n, bool x_order,
                       uint32_t sigma) {
    const Taps tm = taps_scharr_main(sigma), to = taps_scharr_off(sigma);
    AKZ_TRY(ensure(c, c->scratch[0], plane_bytes(w, h, n)));
    float* tmp = (float*)c->scratch[0].p;
    launch::filter_h_f32(c->stream, d_in, tmp, w, h, n, x_order ? tm : to);
    launch::filter_v_f32(c->stream, tmp, d_out, w, h, n, x_order ? to : tm);
    return AKZ_OK;
}

// Level 0 of a job of the tiled family (lib.rs:56-69): Lt0 = gaussian_blur(frame, base_scale_offset) and the contrast factor of
// it in TWO launches -- k_head (both blurs, the Scharr pair, the largest squared gradient magnitude) and k_contrast_hist_final
// (histogram, percentile; leaves the scratch zero) -- instead of six (blur, fill, blur, maximum, histogram, percentile).
// d_gx / d_gy: two planes for the contrast factor's Scharr pair (the caller's level-0 Lx / Ly, which the detector writes much
// later); d_blurred: where the contrast factor's blur of Lt0 goes (level 1's Lsmooth, or NULL: context scratch).
// sched[10] = 1: the separate stages (measurement).
template <typename T>
int head_impl(akz_ctx* c, const T* d_in, float* d_lt0, float* d_blurred, float* d_gx, float* d_gy, uint32_t w, uint32_t h, uint32_t n, float sigma0,
              double percentile, double gscale, uint64_t nbins, double* d_k_out, bool* fused, uint32_t* d_zero_word) {
    *fused = false;
    if (c->prep_mode != 0 || c->sched[10] != 0 || !(sigma0 > 0.0f) || !(gscale > 0.0) || nbins == 0 || nbins > 4096) return AKZ_OK;
    const size_t ks0 = gaussian_kernel_size(sigma0), ks1 = gaussian_kernel_size((float)gscale);
    if (!launch::head_fused_supported(w, h, (uint32_t)ks0, (uint32_t)ks1) || (const void*)d_in == (const void*)d_lt0) return AKZ_OK;
    const std::vector<float> k5 = gaussian_kernel(sigma0, ks0), g3 = gaussian_kernel((float)gscale, ks1);
    const size_t small_bytes = ((size_t)n * (8 + 4 + nbins * 4) + 255) / 256 * 256;  // maxima | bins | tickets
    AKZ_TRY(ensure(c, c->small, small_bytes + (size_t)n * (nbins + 1) * sizeof(double)));
    unsigned long long* d_smax = (unsigned long long*)c->small.p;
    uint32_t* d_hist = (uint32_t*)((char*)c->small.p + (size_t)n * 8);
    uint32_t* d_done = d_hist + (size_t)n * nbins;
    const size_t known_zero = c->small_zero_p == c->small.p ? c->small_zero : 0;
    c->small_zero = 0;  // (until both launches are enqueued: an error in between leaves the scratch in an unknown state)
    if (known_zero < small_bytes)  // (first use, a larger job, or another family used it last)
        AKZ_HIP_TRY(hipMemsetAsync(c->small.p, 0, small_bytes, c->stream));
    float* blurred = d_blurred;  // (level 1's Lsmooth where level 1 continues the octave: the same blur of Lt0, lib.rs:92-95)
    if (!blurred) {
        AKZ_TRY(ensure(c, c->scratch[1], plane_bytes(w, h, n)));
        blurred = (float*)c->scratch[1].p;
    }
    {
        StageTimer st(c, AKZ_ST_BLUR0);
        if constexpr (std::is_same<T, uint8_t>::value)
            launch::head_fused_u8(c->stream, d_in, d_lt0, blurred, d_gx, d_gy, w, h, n, k5.data(), g3.data(), d_smax, d_zero_word);
        else
            launch::head_fused_f32(c->stream, d_in, d_lt0, blurred, d_gx, d_gy, w, h, n, k5.data(), g3.data(), d_smax, d_zero_word);
    }
    {
        StageTimer st(c, AKZ_ST_CONTRAST);
        launch::contrast_hist_final(c->stream, d_gx, d_gy, w, h, n, d_smax, (uint32_t)nbins, d_hist, d_done, percentile, d_k_out);
    }
    AKZ_HIP_TRY(hipGetLastError());
    c->small_zero_p = c->small.p;
    c->small_zero = std::max(known_zero, small_bytes);  // (the histogram pass leaves what it used zero again)
    *fused = true;
    return AKZ_OK;
}
template int head_impl<float>(akz_ctx*, const float*, float*, float*, float*, float*, uint32_t, uint32_t, uint32_t, float, double, double, uint64_t, double*, bool*, uint32_t*);
template int head_impl<uint8_t>(akz_ctx*, const uint8_t*, float*, float*, float*, float*, uint32_t, uint32_t, uint32_t, float, double, double, uint64_t, double*, bool*, uint32_t*);

int contrast_impl(akz_ctx* c, const float* d_in, uint32_t w, uint32_t h, uint32_t n, double percentile,
                         double gscale, uint64_t nbins, double* d_k_out) {
    if (nbins == 0 || nbins > 4096) {
        set_error("contrast_factor: num_bins must be in 1..4096");
        return AKZ_ERR_INVALID_ARG;
    }
    if (w < 5 || h < 5) {
        set_error("contrast_factor: image too small");
        return AKZ_ERR_TOO_SMALL;
    }
    // a multiple of 256 bytes: the runtime then clears it with one fill kernel instead of two (body + tail), which is
    // one dependent dispatch less in a lone frame's launch chain
    const size_t small_bytes = ((size_t)n * (8 + nbins * 4) + 255) / 256 * 256;
    const size_t thr_bytes = (size_t)n * (nbins + 1) * sizeof(double);  // bin thresholds (streaming form), not cleared
    AKZ_TRY(ensure(c, c->small, small_bytes + thr_bytes));
    unsigned long long* d_hmax = (unsigned long long*)c->small.p;
    uint32_t* d_hist = (uint32_t*)((char*)c->small.p + (size_t)n * 8);
    double* d_thr = (double*)((char*)c->small.p + small_bytes);
    AKZ_HIP_TRY(hipMemsetAsync(c->small.p, 0, small_bytes, c->stream));
    c->small_zero = 0;  // (these passes do not clean up after themselves)
    const size_t ks = gaussian_kernel_size((float)gscale);
    const bool stream = c->prep_mode != 0 && gscale > 0.0 && launch::contrast_stream_supported(w, h, (uint32_t)ks, (uint32_t)nbins) &&
                        (c->prep_mode == 1 || (uint64_t)w * h * n >= c->stream_min_px);
    const bool march = gscale > 0.0 && (c->prep_mode == 3 || (c->prep_mode == 2 && (uint64_t)w * h * n >= c->launch_min_px)) &&
                       launch::contrast_march_supported(w, h, (uint32_t)ks, (uint32_t)nbins);
    if (march) {
        const std::vector<float> g3 = gaussian_kernel((float)gscale, ks);
        launch::contrast_march(c->stream, d_in, w, h, n, g3.data(), d_hmax, (uint32_t)nbins, d_hist, d_thr);
    } else if (stream) {
        // both passes recompute blur + Scharr from the input in registers: no blurred plane is written or re-read
        const std::vector<float> g3 = gaussian_kernel((float)gscale, ks);
        launch::contrast_stream(c->stream, d_in, w, h, n, g3.data(), d_hmax, (uint32_t)nbins, d_hist, d_thr);
    } else {
        AKZ_TRY(ensure(c, c->scratch[1], plane_bytes(w, h, n)));
        float* blurred = (float*)c->scratch[1].p;
        AKZ_TRY(gaussian_blur_impl<float>(c, d_in, blurred, w, h, n, (float)gscale));
        launch::contrast_max(c->stream, blurred, w, h, n, d_hmax);
        launch::contrast_hist(c->stream, blurred, w, h, n, d_hmax, (uint32_t)nbins, d_hist);
    }
    launch::contrast_final(c->stream, d_hmax, d_hist, (uint32_t)nbins, percentile, n, d_k_out);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}

// FED launch plan: the level's n_tau steps are cut into ceil(n_tau / 8) launches of balanced size.
// Small launches (a lone frame's coarse octaves: 4 to 72 workgroups, every launch at the floor of a dependent
// dispatch) fuse up to 16 steps on flat 64 x 10 tiles instead: half the launches of a level (launches of at most 512
// workgroups take this form).
static constexpr uint32_t kFedMaxFuse = 8;
uint32_t fed_max_fuse(const akz_ctx* c, uint32_t w, uint32_t h, uint32_t n) {
    constexpr uint64_t deep_wgs = gates::kFedDeepWorkgroups;
    return c->fed_mode == 2 && launch::fed_deep_workgroups(w, h, n) <= deep_wgs ? 2 * kFedMaxFuse : kFedMaxFuse;
}
uint32_t fed_num_launches(const akz_ctx* c, uint32_t n_tau, uint32_t w, uint32_t h, uint32_t n) {
    if (c->fed_mode == 0) return n_tau;
    const uint32_t fuse = fed_max_fuse(c, w, h, n);
    return (n_tau + fuse - 1) / fuse;
}
// calculate_step x n_tau.  The first launch reads `in` (never written), launches alternate between
// the buffers A and B such that the LAST one writes A.  `in` may be B or a third buffer, never A
// unless the number of launches is even (then A is rewritten only after it was consumed).
float* fed_dst(uint32_t launches, uint32_t k /*1-based*/, float* A, float* B) {
    return ((launches - k) % 2 == 0) ? A : B;
}
int fed_impl(akz_ctx* c, const float* in, float* A, float* B, const float* lflow, float* lstep, uint32_t w,
                    uint32_t h, uint32_t n, const double* taus, uint32_t n_tau, const launch::FedNextPrep* next, bool* next_done) {
    if (next_done) *next_done = false;
    const uint32_t launches = fed_num_launches(c, n_tau, w, h, n);
    if (launches == 0) {
        if (in != A) AKZ_HIP_TRY(hipMemcpyAsync(A, in, plane_bytes(w, h, n), hipMemcpyDeviceToDevice, c->stream));
        return AKZ_OK;
    }
    const float* cur = in;
    uint32_t done = 0;
    for (uint32_t k = 1; k <= launches; ++k) {
        float* dst = fed_dst(launches, k, A, B);
        if ((const float*)dst == cur) {
            set_error("internal: FED ping-pong aliasing");
            return AKZ_ERR_INVALID_ARG;
        }
        if (c->fed_mode == 0) {
            const float half_tau = 0.5f * (float)taus[done];
            done += 1;
            launch::fed_step(c->stream, cur, lflow, dst, (done == n_tau) ? lstep : nullptr, w, h, n, half_tau);
        } else {
            const uint32_t cnt = (n_tau - done + (launches - k + 1) - 1) / (launches - k + 1);  // balanced chunks
            float ht[2 * kFedMaxFuse];
            for (uint32_t j = 0; j < cnt; ++j) ht[j] = 0.5f * (float)taus[done + j];
            done += cnt;
            // the level's last launch may carry the next level's preparation (k_fed_own's epilogue)
            const launch::FedNextPrep* np = (k == launches && next && launch::fed_epilogue_supported(w, h, cnt)) ? next : nullptr;
            launch::fed_fused(c->stream, cur, lflow, dst, (done == n_tau) ? lstep : nullptr, w, h, n, ht, cnt, np);
            if (np && next_done) *next_done = true;
        }
        cur = dst;
    }
    if (c->profiling) {
        c->prof.fed_launches += launches;
        c->prof.fed_px_steps += (uint64_t)w * h * n * n_tau;
    }
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}

// Detector kernel family of one launch: 0 = LDS-tiled pair (k_deriv1 + k_deriv2: the fallback for kernel sizes and
// image sizes the other two do not cover), 4 = one LDS-tiled kernel (k_detector_tiled: small launches and single
// frames, where extract_begin also groups levels of equal sigma_size into one launch), 5 = the one-pass column march
// (k_detector_march, akz_march.hip: large launches).  Measured on MI355X per level of a 32-frame batch, all planes
// kept (tools/march_probe.py, microseconds, sigma_size 3): 1920x1080 555 / 503 / 350, 960x540 140 / 124 / 82,
// 480x270 45 / 41 / 66 (pair / tiled / march): the march from 8 Mpx per launch on.
int detector_family(const akz_ctx* c, uint32_t sigma, uint32_t w, uint32_t h, uint32_t n, float border_m,
                           bool keep_second, bool nms) {
    (void)keep_second;
    if (c->det_mode == 0) return 0;
    if (c->det_mode == 4) return launch::detector_tiled_fused_supported(sigma) ? 4 : 0;
    if (c->det_mode == 5) return launch::detector_march_supported(sigma, w, h, border_m, nms) ? 5 : 0;
    const uint64_t march_min = c->launch_min_px;
    const uint64_t px = (uint64_t)w * h * n;
    if (px >= march_min && launch::detector_march_supported(sigma, w, h, border_m, nms)) return 5;
    if (px < march_min && launch::detector_tiled_fused_supported(sigma)) return 4;
    return 0;
}

int detector_impl(akz_ctx* c, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx,
                         float* lyy, float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n) {
    if (sigma == 0 || 2 * sigma + 1 > (uint32_t)kMaxTaps) {
        set_error("detector_response: sigma_size must be in 1..6");
        return AKZ_ERR_INVALID_ARG;
    }
    AKZ_TRY(check_plane_args(lsmooth, ldet_out, w, h, n, (int)sigma));
    if (const int fam = detector_family(c, sigma, w, h, n, 0.0f, lxx && lyy && lxy, false)) {
        (fam == 5 ? launch::detector_march : launch::detector_tiled_fused)(
            c->stream, lsmooth, sigma, lx, ly, lxx, lyy, lxy, ldet_out, w, h, n, 0, 0.0f, 0.0f, nullptr, 0, nullptr);
        AKZ_HIP_TRY(hipGetLastError());
        return AKZ_OK;
    }
    if (launch::detector_fused_supported(sigma)) {
        launch::detector_fused(c->stream, lsmooth, sigma, lx, ly, lxx, lyy, lxy, ldet_out, w, h, n);
        AKZ_HIP_TRY(hipGetLastError());
        return AKZ_OK;
    }
    const size_t pb = plane_bytes(w, h, n);
    if (!lxx) { AKZ_TRY(ensure(c, c->scratch[2], pb)); lxx = (float*)c->scratch[2].p; }
    if (!lyy) { AKZ_TRY(ensure(c, c->scratch[3], pb)); lyy = (float*)c->scratch[3].p; }
    if (!lxy) { AKZ_TRY(ensure(c, c->scratch[4], pb)); lxy = (float*)c->scratch[4].p; }
    AKZ_TRY(scharr_impl(c, lsmooth, lx, w, h, n, true, sigma));   // Lx  = scharr(Lsmooth, x)
    AKZ_TRY(scharr_impl(c, lsmooth, ly, w, h, n, false, sigma));  // Ly  = scharr(Lsmooth, y)
    AKZ_TRY(scharr_impl(c, lx, lxx, w, h, n, true, sigma));       // Lxx = scharr(Lx, x)
    AKZ_TRY(scharr_impl(c, ly, lyy, w, h, n, false, sigma));      // Lyy = scharr(Ly, y)
    AKZ_TRY(scharr_impl(c, lx, lxy, w, h, n, false, sigma));      // Lxy = scharr(Lx, y)
    const uint32_t quat = sigma * sigma * sigma * sigma;
    launch::ldet(c->stream, lxx, lyy, lxy, ldet_out, (uint64_t)w * h * n, (float)quat);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}

extern "C" {

int akz_op_horizontal_filter(akz_ctx* c, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                             const float* taps, uint32_t ntaps) {
    AKZ_TRY(bind(c));
    Taps t;
    AKZ_TRY(taps_from_dense(taps, ntaps, t));
    AKZ_TRY(check_plane_args(d_in, d_out, w, h, n, t.hw));
    launch::filter_h_f32(c->stream, d_in, d_out, w, h, n, t);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
int akz_op_vertical_filter(akz_ctx* c, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                           const float* taps, uint32_t ntaps) {
    AKZ_TRY(bind(c));
    Taps t;
    AKZ_TRY(taps_from_dense(taps, ntaps, t));
    AKZ_TRY(check_plane_args(d_in, d_out, w, h, n, t.hw));
    launch::filter_v_f32(c->stream, d_in, d_out, w, h, n, t);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
int akz_op_gaussian_blur(akz_ctx* c, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                         float sigma) {
    AKZ_TRY(bind(c));
    return gaussian_blur_impl<float>(c, d_in, d_out, w, h, n, sigma);
}
int akz_op_gaussian_blur_u8(akz_ctx* c, const uint8_t* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                            float sigma) {
    AKZ_TRY(bind(c));
    return gaussian_blur_impl<uint8_t>(c, d_in, d_out, w, h, n, sigma);
}
int akz_op_half_size(akz_ctx* c, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n) {
    AKZ_TRY(bind(c));
    AKZ_TRY(check_plane_args(d_in, d_out, w, h, n, 0));
    if (w < 2 || h < 2) {
        set_error("half_size: image smaller than 2x2");
        return AKZ_ERR_TOO_SMALL;
    }
    launch::half_size(c->stream, d_in, d_out, w, h, n);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
int akz_op_scharr(akz_ctx* c, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n, int x_order,
                  int y_order, uint32_t sigma) {
    AKZ_TRY(bind(c));
    if (x_order == 0 && y_order == 0) {
        // derivatives.rs:127-128: neither order is a new (zero) image of the input's size; sigma_size is not looked at
        AKZ_TRY(check_plane_args(d_in, d_out, w, h, n, 0));
        AKZ_HIP_TRY(hipMemsetAsync(d_out, 0, plane_bytes(w, h, n), c->stream));
        return AKZ_OK;
    }
    if (sigma == 0 || 2 * sigma + 1 > (uint32_t)kMaxTaps) {
        set_error("scharr: sigma_size must be in 1..6");
        return AKZ_ERR_INVALID_ARG;
    }
    AKZ_TRY(check_plane_args(d_in, d_out, w, h, n, (int)sigma));
    // derivatives.rs:118-122: with both orders the reference takes the HORIZONTAL derivative twice and adds the two
    // images (`sqrt_squared` is `image_1 += image_2`, image.rs:218-231)
    AKZ_TRY(scharr_impl(c, d_in, d_out, w, h, n, x_order != 0, sigma));
    if (x_order != 0 && y_order != 0) launch::accumulate(c->stream, d_out, d_out, (uint64_t)w * h * n);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
int akz_op_pm_g2(akz_ctx* c, const float* d_lx, const float* d_ly, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                 const double* d_k) {
    AKZ_TRY(bind(c));
    AKZ_TRY(check_plane_args(d_lx, d_out, w, h, n, 0));
    if (!d_ly || !d_k) return AKZ_ERR_INVALID_ARG;
    launch::pm_g2(c->stream, d_lx, d_ly, d_out, w, h, n, d_k, 0);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
int akz_op_contrast_factor(akz_ctx* c, const float* d_in, uint32_t w, uint32_t h, uint32_t n, double percentile,
                           double gscale, uint64_t num_bins, double* d_k_out) {
    AKZ_TRY(bind(c));
    if (!d_in || !d_k_out || n == 0) return AKZ_ERR_INVALID_ARG;
    return contrast_impl(c, d_in, w, h, n, percentile, gscale, num_bins, d_k_out);
}
int akz_op_flow(akz_ctx* c, const float* d_lsmooth, float* d_lflow, uint32_t w, uint32_t h, uint32_t n,
                const double* d_k, uint32_t k_scale_pow) {
    AKZ_TRY(bind(c));
    AKZ_TRY(check_plane_args(d_lsmooth, d_lflow, w, h, n, 1));
    if (!d_k) return AKZ_ERR_INVALID_ARG;
    launch::flow(c->stream, d_lsmooth, d_lflow, w, h, n, d_k, k_scale_pow);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
int akz_op_fed_steps(akz_ctx* c, float* d_lt, const float* d_lflow, float* d_lstep, uint32_t w, uint32_t h,
                     uint32_t n, const double* taus, uint32_t n_tau) {
    AKZ_TRY(bind(c));
    AKZ_TRY(check_plane_args(d_lt, d_lflow, w, h, n, 1));
    if (n_tau && !taus) return AKZ_ERR_INVALID_ARG;
    // in place for the caller: copy the input aside when the launch count is odd
    AKZ_TRY(ensure(c, c->scratch[5], plane_bytes(w, h, n)));
    AKZ_TRY(ensure(c, c->scratch[3], plane_bytes(w, h, n)));
    float* B = (float*)c->scratch[5].p;
    float* in = (float*)c->scratch[3].p;
    AKZ_HIP_TRY(hipMemcpyAsync(in, d_lt, plane_bytes(w, h, n), hipMemcpyDeviceToDevice, c->stream));
    StageTimer st(c, AKZ_ST_FED);  // the launches alone, without the copy above (stand-alone roofline legs of bench.py)
    st.kernel(AKZ_KR_FED_OWN, n_tau, w, h, n, fed_num_launches(c, n_tau, w, h, n), 0, (uint64_t)w * h * n * n_tau);
    return fed_impl(c, in, d_lt, B, d_lflow, d_lstep, w, h, n, taus, n_tau);
}
int akz_op_detector_response(akz_ctx* c, const float* d_lsmooth, uint32_t sigma_size, float* d_lx, float* d_ly,
                             float* d_lxx, float* d_lyy, float* d_lxy, float* d_ldet, uint32_t w, uint32_t h,
                             uint32_t n) {
    AKZ_TRY(bind(c));
    if (!d_lx || !d_ly) return AKZ_ERR_INVALID_ARG;
    return detector_impl(c, d_lsmooth, sigma_size, d_lx, d_ly, d_lxx, d_lyy, d_lxy, d_ldet, w, h, n);
}

}  // extern "C"

