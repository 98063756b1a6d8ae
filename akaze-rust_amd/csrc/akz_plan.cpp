// Host-side planning: everything the reference computes once per (width, height, Config)
// in scalar f32/f64 code — level table, FED step sizes, filter taps.  These values feed the
// HIP kernels as arguments; transcendental functions are evaluated by the host libm so that
// they agree with what a CPU build of the reference would produce on the same machine.
#include <cmath>
#include <cstring>

#include "akz_internal.hpp"

namespace akz {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const std::string& get_error() { return g_error; }

// Rust float -> integer casts saturate (NaN -> 0).
static uint64_t to_usize(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)v;
}
static uint32_t to_u32(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)v;
}

int config_validate(const akz_config& c) {
    if (c.num_sublevels == 0 || c.max_octave_evolution == 0) {
        set_error("Config: num_sublevels and max_octave_evolution must be >= 1");
        return AKZ_ERR_INVALID_ARG;
    }
    if (c.num_sublevels * (uint64_t)c.max_octave_evolution > (uint64_t)kMaxLevels) {
        set_error("Config: more than 64 evolution levels");
        return AKZ_ERR_INVALID_ARG;
    }
    if (c.descriptor_channels < 1 || c.descriptor_channels > 3) {
        set_error("Config: descriptor_channels must be 1, 2 or 3");  // descriptors.rs:48
        return AKZ_ERR_INVALID_ARG;
    }
    if (c.descriptor_pattern_size != 10) {
        // the 2x2/3x3/4x4 grid geometry (steps 10/7/5) is baked into the M-LDB kernel
        set_error("Config: descriptor_pattern_size other than 10 is not supported");
        return AKZ_ERR_UNSUPPORTED;
    }
    if (c.contrast_factor_num_bins == 0 || c.contrast_factor_num_bins > 4096) {
        set_error("Config: contrast_factor_num_bins must be in 1..4096");
        return AKZ_ERR_INVALID_ARG;
    }
    if (!(c.base_scale_offset > 0.0) || std::ceil(c.base_scale_offset) > 6.0) {
        set_error("Config: base_scale_offset must be in (0, 6]");
        return AKZ_ERR_INVALID_ARG;
    }
    return AKZ_OK;
}

// ops::fed_tau — akaze/src/ops/fed_tau.rs:27-106
static bool small_is_prime(uint64_t n) {
    if (n < 2) return false;
    for (uint64_t d = 2; d * d <= n; ++d)
        if (n % d == 0) return false;
    return true;
}
int fed_tau_by_process_time(double T, int M, double tau_max, bool reordering, std::vector<double>& tau) {
    tau.clear();
    const double t = T / (double)M;
    const uint64_t n = to_usize(std::ceil(std::sqrt(3.0 * t / tau_max + 0.25) - 0.5 - 1.0e-8) + 0.5);
    if (n == 0) return AKZ_OK;
    if (n > (1u << 20)) {
        set_error("fed_tau: unreasonable number of steps");
        return AKZ_ERR_INVALID_ARG;
    }
    const double scale = 3.0 * t / (tau_max * (double)(n * (n + 1)));
    const double pi = 3.14159265358979323846264338327950288;
    const double c = 1.0 / (4.0 * (double)n + 2.0);
    const double d = scale * tau_max / 2.0;
    std::vector<double> natural(n);
    for (uint64_t k = 0; k < n; ++k) {
        const double hcos = std::cos(pi * (2.0 * (double)k + 1.0) * c);
        natural[k] = d / (hcos * hcos);
    }
    if (!reordering) {
        tau = natural;
        return AKZ_OK;
    }
    const uint64_t kappa = n / 2;  // kappa-cycle reordering
    if (kappa == 0) {
        set_error("fed_tau: n == 1 with reordering never terminates in the reference (fed_tau.rs:95)");
        return AKZ_ERR_UNSUPPORTED;
    }
    uint64_t prime = n + 1;
    while (!small_is_prime(prime)) ++prime;
    tau.resize(n);
    uint64_t k = 0;
    for (uint64_t slot = 0; slot < n; ++slot) {
        for (;;) {
            const uint64_t r = ((k + 1) * kappa) % prime;
            ++k;
            if (r >= 1 && r - 1 < n) {
                tau[slot] = natural[r - 1];
                break;
            }
        }
    }
    return AKZ_OK;
}

// types::image::gaussian / gaussian_kernel — akaze/src/types/image.rs:341-365 (all f32)
size_t gaussian_kernel_size(float sigma) { return (size_t)to_usize(std::ceil(sigma)) * 2 + 1; }
std::vector<float> gaussian_kernel(float sigma, size_t kernel_size) {
    std::vector<float> k(kernel_size, 0.0f);
    const int hw = (int)(kernel_size / 2);
    const float pi = 3.14159265358979323846f;
    const float recip = 1.0f / (std::sqrt(2.0f * pi) * sigma);
    float sum = 0.0f;
    for (int i = -hw; i <= hw; ++i) {
        const float x = (float)i;
        const float v = recip * expf(-(x * x) / (2.0f * (sigma * sigma)));
        k[(size_t)(i + hw)] = v;
        sum += v;
    }
    for (float& v : k) v /= sum;
    return k;
}

// ops::derivatives — akaze/src/ops/derivatives.rs:74-101
void scharr_kernels(uint32_t scale, std::vector<float>& main_axis, std::vector<float>& off_axis) {
    const size_t size = 3 + 2 * (size_t)(scale - 1);
    const double w = 10.0 / 3.0;
    const double norm = 1.0 / (2.0 * (double)scale * (w + 2.0));
    main_axis.assign(size, 0.0f);
    off_axis.assign(size, 0.0f);
    main_axis[0] = (float)norm;
    main_axis[size / 2] = (float)(w * norm);
    main_axis[size - 1] = (float)norm;
    off_axis[0] = -1.0f;
    off_axis[size / 2] = 0.0f;
    off_axis[size - 1] = 1.0f;
}

int taps_from_dense(const float* k, uint32_t ntaps, Taps& t) {
    if (!k || ntaps == 0 || ntaps % 2 == 0 || ntaps > (uint32_t)kMaxTaps) {
        set_error("filter kernel must have an odd number of taps, at most 13");
        return AKZ_ERR_INVALID_ARG;
    }
    t = Taps();
    t.n = (int)ntaps;
    t.hw = (int)(ntaps / 2);
    for (int i = 0; i < t.n; ++i) {
        t.off[i] = i - t.hw;
        t.wgt[i] = k[i];
    }
    return AKZ_OK;
}
// Scharr kernels are zero except at -s, 0, +s; the zero taps add +/-0 to the running sum and
// are skipped (bit-identical up to the sign of an exact zero).
static Taps sparse3(const std::vector<float>& dense, uint32_t scale) {
    Taps t;
    t.n = 3;
    t.hw = (int)scale;
    t.off[0] = -(int)scale; t.wgt[0] = dense.front();
    t.off[1] = 0;           t.wgt[1] = dense[dense.size() / 2];
    t.off[2] = (int)scale;  t.wgt[2] = dense.back();
    return t;
}
Taps taps_scharr_main(uint32_t scale) {
    std::vector<float> m, o;
    scharr_kernels(scale, m, o);
    return sparse3(m, scale);
}
Taps taps_scharr_off(uint32_t scale) {
    std::vector<float> m, o;
    scharr_kernels(scale, m, o);
    return sparse3(o, scale);
}

// EvolutionStep::new + allocate_evolutions — akaze/src/types/evolution.rs:101-161 ; level
// sizes from the chained half_size of lib.rs:80-90 ; detector sigma from detector_response.rs:21-24.
int build_plan(uint32_t w, uint32_t h, const akz_config& cfg, std::vector<LevelPlan>& plan) {
    plan.clear();
    AKZ_TRY(config_validate(cfg));
    if (w == 0 || h == 0) {
        set_error("empty image");
        return AKZ_ERR_INVALID_ARG;
    }
    uint32_t lw = w, lh = h;
    for (uint32_t o = 0; o < cfg.max_octave_evolution; ++o) {
        const double rfactor = 1.0 / std::pow(2.0, (double)o);
        const uint32_t level_h = to_u32((double)h * rfactor);
        const uint32_t level_w = to_u32((double)w * rfactor);
        if (!((level_w >= 80 && level_h >= 40) || o == 0)) break;
        if (o > 0) {
            lw /= 2;
            lh /= 2;
        }
        for (uint32_t s = 0; s < cfg.num_sublevels; ++s) {
            LevelPlan lv;
            lv.esigma = cfg.base_scale_offset * std::pow(2.0, (double)s / (double)cfg.num_sublevels + (double)o);
            lv.etime = 0.5 * (lv.esigma * lv.esigma);
            lv.octave = o;
            lv.sublevel = s;
            lv.sigma_size = to_u32(std::round(lv.esigma));
            lv.w = lw;
            lv.h = lh;
            lv.det_sigma = to_u32(std::round(lv.esigma * cfg.derivative_factor / std::pow(2.0, (double)o)));
            plan.push_back(lv);
        }
    }
    for (size_t i = 1; i < plan.size(); ++i) {
        const double ttime = plan[i].etime - plan[i - 1].etime;
        AKZ_TRY(fed_tau_by_process_time(ttime, 1, 0.25, true, plan[i].tau));
    }
    // every stencil clamps to [hw, dim-1-hw]; the widest is the detector Scharr (hw = det_sigma)
    // and the keypoint border test needs far more room anyway.
    for (const LevelPlan& lv : plan) {
        const uint32_t need = 2 * std::max<uint32_t>(lv.det_sigma, 2) + 3;
        if (lv.det_sigma == 0 || lv.det_sigma > 6) {
            set_error("detector sigma outside 1..6 (derivative_factor too large/small)");
            return AKZ_ERR_UNSUPPORTED;
        }
        if (lv.w < need || lv.h < need) {
            set_error("image too small for the filter half-widths of its pyramid");
            return AKZ_ERR_TOO_SMALL;
        }
    }
    return AKZ_OK;
}

}  // namespace akz
