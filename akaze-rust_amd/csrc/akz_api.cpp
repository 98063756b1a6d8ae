// C ABI of libakaze_hip.so (declared in include/akaze_hip.h), part 1: context lifecycle, device memory helpers, host-side
// planning, profile, mode setters and measurement hooks.  See akz_ctx.hpp for how the ABI is split over translation units.
#include "akz_ctx.hpp"

// Host cores this process can count on: the affinity mask, cut down by the cgroup CPU quota, divided among the ranks
// of a one-process-per-GPU job on this host (LOCAL_WORLD_SIZE, set by torchrun).
unsigned akz::host_cpu_share() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0)
            n = std::min<unsigned>(n, (unsigned)std::max(1L, atol(q) / period));
        fclose(f);
    }
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
        const int ranks = atoi(e);
        if (ranks > 1) n = std::max(2u, n / (unsigned)ranks);
    }
    return std::max(1u, n);
}

int level_info_out(const std::vector<LevelPlan>& plan, uint64_t level, double* etime, double* esigma,
                          uint32_t* octave, uint32_t* sublevel, uint32_t* sigma_size, uint32_t* lw, uint32_t* lh,
                          uint32_t* det_sigma, uint64_t* n_tau, double* tau, uint64_t tau_cap) {
    if (level >= plan.size()) {
        set_error("level out of range");
        return AKZ_ERR_INVALID_ARG;
    }
    const LevelPlan& lv = plan[(size_t)level];
    if (etime) *etime = lv.etime;
    if (esigma) *esigma = lv.esigma;
    if (octave) *octave = lv.octave;
    if (sublevel) *sublevel = lv.sublevel;
    if (sigma_size) *sigma_size = lv.sigma_size;
    if (lw) *lw = lv.w;
    if (lh) *lh = lv.h;
    if (det_sigma) *det_sigma = lv.det_sigma;
    if (n_tau) *n_tau = lv.tau.size();
    if (tau)
        for (size_t i = 0; i < lv.tau.size() && i < tau_cap; ++i) tau[i] = lv.tau[i];
    return AKZ_OK;
}

extern "C" {

int akz_abi_version(void) { return AKZ_ABI_VERSION; }
const char* akz_last_error(void) { return get_error().c_str(); }

void akz_config_default(akz_config* o) {
    if (!o) return;
    o->num_sublevels = 4;
    o->max_octave_evolution = 4;
    o->base_scale_offset = 1.6;
    o->initial_contrast = 0.001;
    o->contrast_percentile = 0.7;
    o->contrast_factor_num_bins = 300;
    o->derivative_factor = 1.5;
    o->detector_threshold = 0.001;
    o->descriptor_channels = 3;
    o->descriptor_pattern_size = 10;
}

int akz_ctx_create(int device, void* stream, akz_ctx** out) {
    if (!out) {
        set_error("akz_ctx_create: null out");
        return AKZ_ERR_INVALID_ARG;
    }
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device visible: the A-KAZE HIP path cannot run (there is no CPU fallback)");
        return AKZ_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device index out of range");
        return AKZ_ERR_INVALID_ARG;
    }
    AKZ_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    AKZ_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("libakaze_hip.so carries gfx950 code only; device is ") + prop.gcnArchName);
        return AKZ_ERR_NO_DEVICE;
    }
    std::unique_ptr<akz_ctx> c(new akz_ctx);
    c->device = device;
    c->stream = c->main = (hipStream_t)stream;  // NULL == the default stream
    *out = c.release();
    return AKZ_OK;
}
int akz_stream_create(int device, void** stream_out) {
    if (!stream_out) return AKZ_ERR_INVALID_ARG;
    AKZ_HIP_TRY(hipSetDevice(device));
    hipStream_t s = nullptr;
    AKZ_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (void*)s;
    return AKZ_OK;
}
int akz_stream_destroy(int device, void* stream) {
    AKZ_HIP_TRY(hipSetDevice(device));
    if (stream) AKZ_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return AKZ_OK;
}

int akz_ctx_destroy(akz_ctx* c) {
    if (!c) return AKZ_OK;
    (void)hipSetDevice(c->device);
    finisher_stop(c);
    for (akz_ctx* l : c->lanes) (void)akz_ctx_destroy(l);
    c->lanes.clear();
    if (c->lane_in) { (void)hipEventDestroy(c->lane_in); c->lane_in = nullptr; }
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->lazy[0], &c->lazy[1], &c->lazy[2], &c->lazy[3], &c->lazy[4], &c->lazy[5],
                      &c->scratch[0], &c->scratch[1], &c->scratch[2], &c->scratch[3], &c->scratch[4], &c->scratch[5], &c->scratch_coarse,
                      &c->small, &c->cand, &c->cand_sorted, &c->sort_scratch, &c->rel_scratch, &c->sel_scratch, &c->sel_recs, &c->bucket_scratch, &c->kp_in, &c->kp_out, &c->match_a, &c->match_b, &c->match_rec, &c->match_state,
                      &c->match_out, &c->cosi, &c->mm_q8, &c->mm_t8, &c->mm_pop, &c->mm_tab, &c->mm_cols,
                      &c->ms1.q8, &c->ms1.t8, &c->ms1.pop, &c->ms1.tab, &c->ms1.cols, &c->ms1.rec, &c->ransac_dev};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (DevBuf& b : c->pin)
        if (b.p) (void)hipHostFree(b.p);
    if (c->ransac_pin.p) (void)hipHostFree(c->ransac_pin.p);
    if (c->tab_ring) (void)hipHostFree(c->tab_ring);
    c->tab_ring = nullptr;
    if (c->ms1.ring) (void)hipHostFree(c->ms1.ring);
    c->ms1.ring = nullptr;
    for (hipEvent_t& e : c->ms1.ring_ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    for (hipEvent_t& e : c->tab_ring_ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (c->copy) {
        (void)hipStreamSynchronize(c->copy);
        (void)hipStreamDestroy(c->copy);
        c->copy = nullptr;
    }
    for (int i = 0; i < akz_ctx::kSlots; ++i) {
        if (c->cand_slot[i].p) (void)hipFree(c->cand_slot[i].p);
        if (c->count_slot[i].p) (void)hipFree(c->count_slot[i].p);
        if (c->stage[i].p) (void)hipFree(c->stage[i].p);
        if (c->staged[i]) (void)hipEventDestroy(c->staged[i]);
        c->staged[i] = nullptr;
    }
    if (c->aux) {
        (void)hipStreamSynchronize(c->aux);
        (void)hipStreamDestroy(c->aux);
    }
    if (c->coarse) {
        (void)hipStreamSynchronize(c->coarse);
        (void)hipStreamDestroy(c->coarse);
        c->coarse = nullptr;
    }
    if (c->pre) {
        (void)hipStreamSynchronize(c->pre);
        (void)hipStreamDestroy(c->pre);
        c->pre = nullptr;
    }
    if (c->pre_done) { (void)hipEventDestroy(c->pre_done); c->pre_done = nullptr; }
    for (auto& s : c->slab_pool) (void)hipFree(s.second);
    for (auto& sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    c->slab_pool.clear();
    c->spans.clear();
    c->ev_pool.clear();
    c->aux = nullptr;
    c->workers.reset();  // joins the host worker threads
    for (hipEvent_t& e : c->fed_ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    for (hipEvent_t& e : c->pre_ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    for (hipEvent_t& e : c->probe_ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    c->dead = true;  // results that are still alive keep the (now resource-less) struct; see result_delete
    if (c->live_results == 0) delete c;
    return AKZ_OK;
}
int akz_ctx_synchronize(akz_ctx* c) {
    AKZ_TRY(bind(c));
    for (akz_ctx* l : c->lanes) AKZ_TRY(akz_ctx_synchronize(l));
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->coarse) AKZ_HIP_TRY(hipStreamSynchronize(c->coarse));
    if (c->pre) AKZ_HIP_TRY(hipStreamSynchronize(c->pre));
    if (c->copy) AKZ_HIP_TRY(hipStreamSynchronize(c->copy));
    return AKZ_OK;
}
void* akz_ctx_stream(akz_ctx* c) { return c ? (void*)c->main : nullptr; }

int akz_device_malloc(akz_ctx* c, size_t bytes, void** d_out) {
    AKZ_TRY(bind(c));
    if (!d_out) return AKZ_ERR_INVALID_ARG;
    AKZ_HIP_TRY(hipMalloc(d_out, bytes ? bytes : 1));
    return AKZ_OK;
}
int akz_device_free(akz_ctx* c, void* d_ptr) {
    AKZ_TRY(bind(c));
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    AKZ_HIP_TRY(hipFree(d_ptr));
    return AKZ_OK;
}
int akz_memcpy_h2d(akz_ctx* c, void* d_dst, const void* src, size_t bytes) {
    AKZ_TRY(bind(c));
    AKZ_HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    return AKZ_OK;
}
int akz_memcpy_d2h(akz_ctx* c, void* dst, const void* d_src, size_t bytes) {
    AKZ_TRY(bind(c));
    AKZ_HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    return AKZ_OK;
}

// ---------------------------------------------------------------------------------------------
// host-side planning
// ---------------------------------------------------------------------------------------------
int akz_fed_tau_by_process_time(double T, int M, double tau_max, int reordering, double* out, uint64_t cap,
                                uint64_t* n) {
    if (M <= 0 || !(tau_max > 0.0)) {
        set_error("fed_tau: M and tau_max must be positive");
        return AKZ_ERR_INVALID_ARG;
    }
    std::vector<double> tau;
    AKZ_TRY(fed_tau_by_process_time(T, M, tau_max, reordering != 0, tau));
    if (n) *n = tau.size();
    if (out)
        for (size_t i = 0; i < tau.size() && i < cap; ++i) out[i] = tau[i];
    return AKZ_OK;
}
int akz_gaussian_kernel(float sigma, uint64_t kernel_size, float* out) {
    if (!out || kernel_size == 0 || kernel_size % 2 == 0) {
        set_error("gaussian_kernel: odd kernel_size and non-null out required");
        return AKZ_ERR_INVALID_ARG;
    }
    const std::vector<float> k = gaussian_kernel(sigma, (size_t)kernel_size);
    std::memcpy(out, k.data(), k.size() * sizeof(float));
    return AKZ_OK;
}
int akz_scharr_kernels(uint32_t scale, float* main_axis, float* off_axis) {
    if (scale == 0 || !main_axis || !off_axis) {
        set_error("scharr_kernels: scale >= 1 and non-null outputs required");
        return AKZ_ERR_INVALID_ARG;
    }
    std::vector<float> m, o;
    scharr_kernels(scale, m, o);
    std::memcpy(main_axis, m.data(), m.size() * sizeof(float));
    std::memcpy(off_axis, o.data(), o.size() * sizeof(float));
    return AKZ_OK;
}
int akz_plan_num_levels(uint32_t w, uint32_t h, const akz_config* cfg, uint64_t* n_levels) {
    if (!cfg || !n_levels) return AKZ_ERR_INVALID_ARG;
    std::vector<LevelPlan> plan;
    AKZ_TRY(build_plan(w, h, *cfg, plan));
    *n_levels = plan.size();
    return AKZ_OK;
}
int akz_plan_level_info(uint32_t w, uint32_t h, const akz_config* cfg, uint64_t level, double* etime, double* esigma,
                        uint32_t* octave, uint32_t* sublevel, uint32_t* sigma_size, uint32_t* level_w,
                        uint32_t* level_h, uint32_t* detector_sigma, uint64_t* n_tau, double* tau, uint64_t tau_cap) {
    if (!cfg) return AKZ_ERR_INVALID_ARG;
    std::vector<LevelPlan> plan;
    AKZ_TRY(build_plan(w, h, *cfg, plan));
    return level_info_out(plan, level, etime, esigma, octave, sublevel, sigma_size, level_w, level_h, detector_sigma,
                          n_tau, tau, tau_cap);
}

}  // extern "C"


extern "C" {
int akz_ctx_set_profiling(akz_ctx* c, int on) {
    AKZ_TRY(bind(c));
    if (on < 0 || on > 2) return AKZ_ERR_INVALID_ARG;
    c->profiling = on == 1 ? 2 : on == 2 ? 1 : 0;  // API: 1 = all stages, 2 = light (FED spans only)
    for (akz_ctx* l : c->lanes) l->profiling = c->profiling;
    return AKZ_OK;
}
int akz_ctx_warmup(akz_ctx* c) {
    AKZ_TRY(bind(c));
    if (!c->placed) AKZ_TRY(place_streams(c));
    (void)device_libm_mode(c);  // (once per process: the self-test of the device's atan2f / cosf / sinf against this libm)
    return AKZ_OK;
}
static int get_profile_full(akz_ctx* c, akz_profile* out, int reset) {
    AKZ_TRY(bind(c));
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    resolve_spans(c);
    *out = c->prof;
    // what the stream-placement probe decided for this context (not accumulated, not reset)
    out->placement_probed = c->placed ? 1u : 0u;
    out->placement_early_stages = (uint32_t)c->pre_mode;
    out->placement_replaced = (uint32_t)c->place_replaced;
    out->placement_shared = (uint32_t)(c->place_collisions + c->lane_collisions);
    out->placement_retries = (uint32_t)c->place_retries;
    if (reset) c->prof = akz_profile{};
    for (akz_ctx* l : c->lanes) {  // jobs dealt to the lanes are part of this context's profile
        akz_profile p{};
        AKZ_TRY(get_profile_full(l, &p, reset));
        for (size_t i = 0; i < sizeof(p.ms) / sizeof(p.ms[0]); ++i) out->ms[i] += p.ms[i];
        out->calls += p.calls;
        out->pixels += p.pixels;
        out->fed_launches += p.fed_launches;
        out->fed_px_steps += p.fed_px_steps;
        out->det_launches += p.det_launches;
        out->det_px += p.det_px;
        out->fused_px += p.fused_px;
    }
    return AKZ_OK;
}
// The struct has grown at its end (ABI 5: placement_*) and may grow again: akz_ctx_get_profile2 writes at most the bytes the
// caller says its struct has; the original symbol writes the ABI-4 prefix only, so that a host compiled against the
// smaller struct is never written past its end.
int akz_ctx_get_profile2(akz_ctx* c, akz_profile* out, uint64_t struct_size, int reset) {
    if (!out || struct_size < offsetof(akz_profile, placement_probed)) {
        set_error("akz_ctx_get_profile2: null struct, or one smaller than the ABI-4 akz_profile");
        return AKZ_ERR_INVALID_ARG;
    }
    akz_profile full{};
    AKZ_TRY(get_profile_full(c, &full, reset));
    std::memcpy(out, &full, (size_t)std::min<uint64_t>(struct_size, sizeof(full)));
    return AKZ_OK;
}
int akz_ctx_get_profile(akz_ctx* c, akz_profile* out, int reset) {
    return akz_ctx_get_profile2(c, out, offsetof(akz_profile, placement_probed), reset);
}
// akaze::extract_features(input_image_path, options) — akaze/src/lib.rs:167-194
int akz_extract_features_file(akz_ctx* c, const char* path, const akz_config* cfg, uint32_t flags, akz_result** out) {
    if (!c || !path || !out) {
        set_error("akz_extract_features_file: null argument");
        return AKZ_ERR_INVALID_ARG;
    }
    uint32_t w = 0, h = 0;
    uint8_t* luma = nullptr;
    AKZ_TRY(akz_image_load_luma(path, &w, &h, &luma));
    const int st = akz_extract_gray_u8(c, luma, w, h, cfg, flags, out);
    akz_image_free(luma);
    return st;
}

// types::evolution::write_evolutions — evolution.rs:175-218 (file names: build_path, :163-168)
int akz_write_evolutions(const akz_result* r, uint64_t img, const char* dir) {
    if (!r || !dir) {
        set_error("akz_write_evolutions: null argument");
        return AKZ_ERR_INVALID_ARG;
    }
    uint64_t n_levels = 0, nk = 0, nb = 0;
    AKZ_TRY(akz_result_counts(r, img, &n_levels, &nk, &nb));
    static const struct { akz_plane p; const char* label; } order[10] = {
        {AKZ_LT, "Lt_"}, {AKZ_LSMOOTH, "Lsmooth_"}, {AKZ_LX, "Lx_"}, {AKZ_LY, "Ly_"}, {AKZ_LXX, "Lxx_"},
        {AKZ_LYY, "Lyy_"}, {AKZ_LXY, "Lxy_"}, {AKZ_LFLOW, "Lflow_"}, {AKZ_LSTEP, "Lstep_"}, {AKZ_LDET, "Ldet_"}};
    std::vector<float> buf;
    for (uint64_t l = 0; l < n_levels; ++l) {
        uint32_t w = 0, h = 0;
        AKZ_TRY(akz_result_level_info(r, l, nullptr, nullptr, nullptr, nullptr, nullptr, &w, &h, nullptr, nullptr, 0));
        for (const auto& o : order) {
            uint64_t n_px = 0;
            AKZ_TRY(akz_fetch_plane(r, img, l, o.p, nullptr, &n_px));
            if (n_px == 0) continue;  // `save` skips 0x0 images
            buf.resize(n_px);
            AKZ_TRY(akz_fetch_plane(r, img, l, o.p, buf.data(), &n_px));
            char name[64];
            // build_path (evolution.rs:163-168) formats "{label}{idx:05}.png" and then calls set_extension(".png"), which
            // replaces the text after the last dot by ".png" itself: the files are called Lt_00000..png (two dots)
            snprintf(name, sizeof(name), "%s%05llu..png", o.label, (unsigned long long)l);
            const std::string path = std::string(dir) + "/" + name;
            AKZ_TRY(akz_image_save_plane_png(path.c_str(), buf.data(), w, h));
        }
    }
    return AKZ_OK;
}

int akz_ctx_set_detector_mode(akz_ctx* c, int mode) {
    if (!c || (mode != 0 && mode != 2 && mode != 4 && mode != 5)) return AKZ_ERR_INVALID_ARG;
    c->det_mode = mode;
    for (akz_ctx* l : c->lanes) l->det_mode = mode;
    return AKZ_OK;
}

int akz_ctx_set_prep_mode(akz_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 3) return AKZ_ERR_INVALID_ARG;
    c->prep_mode = mode;
    for (akz_ctx* l : c->lanes) l->prep_mode = mode;
    return AKZ_OK;
}

int akz_ctx_set_host_threads(akz_ctx* c, uint32_t threads) {
    AKZ_TRY(bind(c));
    if (threads > 256) return AKZ_ERR_INVALID_ARG;
    for (int k = 0; k < akz_ctx::kSlots; ++k)
        if (c->slot_busy[k]) {
            set_error("akz_ctx_set_host_threads: extractions are in flight on this context");
            return AKZ_ERR_INVALID_ARG;
        }
    c->host_threads = threads;
    c->workers.reset();  // joins the old pool; the next finish starts the new one
    for (akz_ctx* l : c->lanes) AKZ_TRY(akz_ctx_set_host_threads(l, threads));
    return AKZ_OK;
}
int akz_ctx_set_candidate_hint(akz_ctx* c, uint32_t per_image) {
    AKZ_TRY(bind(c));
    c->cand_cap_hint = std::max<uint32_t>(per_image, 16u);
    for (akz_ctx* l : c->lanes) l->cand_cap_hint = c->cand_cap_hint.load();
    return AKZ_OK;
}
int akz_ctx_set_match_mode(akz_ctx* c, int mode) {
    AKZ_TRY(bind(c));
    if (mode < 0 || mode > 3) return AKZ_ERR_INVALID_ARG;
    c->match_mode = mode;
    for (akz_ctx* l : c->lanes) l->match_mode = mode;
    return AKZ_OK;
}
int akz_ctx_set_fed_mode(akz_ctx* c, int mode) {
    AKZ_TRY(bind(c));
    if (mode != 0 && mode != 2) return AKZ_ERR_INVALID_ARG;
    c->fed_mode = mode;
    for (akz_ctx* l : c->lanes) l->fed_mode = mode;
    return AKZ_OK;
}
const char* akz_fed_kernel_name(void) { return "k_level_march + k_fed_own"; }  // (+ k_octave_resident: same profile group)
int akz_debug_march_bands(int kind, uint32_t w, uint32_t h, uint32_t n, int half_width, int32_t* rows, uint32_t cap,
                          uint32_t* n_bands) {
    if (!n_bands || (cap && !rows) || kind < 0 || kind > 1 || w < 16 || h < 16 || n == 0 || half_width < 1 || half_width > 4)
        return AKZ_ERR_INVALID_ARG;
    *n_bands = launch::march_band_rows(kind, w, h, n, half_width, rows, cap);
    return AKZ_OK;
}
const char* akz_detector_kernel_name(void) { return "detector (k_detector_march + k_detector_tiled)"; }
int akz_debug_set_schedule(akz_ctx* c, int key, int value) {
    if (!c || key < 0 || key > 10) return AKZ_ERR_INVALID_ARG;
    AKZ_TRY(bind(c));
    c->sched[key] = value;
    if (key == 7 || key == 8) launch::march_min_band_rows(c->sched[7], c->sched[8]);  // (process-wide: measurement)
    if (key == 4) {
        c->big_px_sync = value > 0 ? (uint64_t)value * 1000u : gates::kBigPxSync;
        c->big_px_async = value > 0 ? (uint64_t)value * 1000u : gates::kBigPxAsync;
        c->lane_px = value > 0 ? (uint64_t)value * 1000u : gates::kLanePx;
        for (akz_ctx* l : c->lanes) {  // (a job that is dealt to a lane runs there as a one-stream chain)
            l->big_px_sync = std::max(c->big_px_sync, c->lane_px);
            l->big_px_async = std::max(c->big_px_async, c->lane_px);
        }
    }
    return AKZ_OK;
}
int akz_debug_placement_verdict(float spin_pair_ms, float tiny_pair_ms, float tiny_alone_ms, float spin_ms) {
    return akz::placement_verdict(spin_pair_ms, tiny_pair_ms, tiny_alone_ms, spin_ms);
}
int akz_debug_stream_placement(akz_ctx* c, int* info) {
    if (!c || !info) return AKZ_ERR_INVALID_ARG;
    AKZ_TRY(bind(c));
    info[0] = c->placed ? 1 : 0;
    info[1] = c->pre_mode;
    info[2] = c->place_replaced;
    info[3] = c->place_collisions + c->lane_collisions;
    return AKZ_OK;
}
int akz_debug_select_info(akz_ctx* c, int* info) {
    if (!c || !info) return AKZ_ERR_INVALID_ARG;
    AKZ_TRY(bind(c));
    const akz_ctx* q = c->lanes.empty() ? c : c->lanes[0];
    info[0] = q->sel_last_mode.load();
    info[1] = (int)q->sel_last_rounds.load();
    info[2] = (int)q->sel_last_fallback.load();
    info[3] = (int)q->last_total_cands.load();
    for (int k = 0; k < 4; ++k) info[4 + k] = (int)q->sel_last_ticks[k].load();
    return AKZ_OK;
}
int akz_debug_set_select(akz_ctx* c, int mode) {
    if (!c) return AKZ_ERR_INVALID_ARG;
    c->dbg_select = mode < 0 ? -1 : (mode >= 2 ? 2 : (mode != 0));
    for (akz_ctx* l : c->lanes) l->dbg_select = c->dbg_select;
    return AKZ_OK;
}
int akz_debug_set_host_sort(akz_ctx* c, int on) {
    if (!c) return AKZ_ERR_INVALID_ARG;
    c->dbg_host_sort = on < 0 ? -1 : (on != 0);
    for (akz_ctx* l : c->lanes) l->dbg_host_sort = c->dbg_host_sort;
    return AKZ_OK;
}
int akz_debug_set_device_libm(akz_ctx* c, int mode) {
    if (!c) return AKZ_ERR_INVALID_ARG;
    c->dbg_device_libm = mode == 0 ? 0 : -1;
    for (akz_ctx* l : c->lanes) l->dbg_device_libm = c->dbg_device_libm;
    return AKZ_OK;
}
int akz_debug_device_libm(akz_ctx* c, int* available, int* last_job) {
    AKZ_TRY(bind(c));
    if (available) *available = device_libm_mode(c);
    if (last_job) {
        *last_job = c->libm_last;
        for (akz_ctx* l : c->lanes) *last_job = std::max(*last_job, l->libm_last);
    }
    return AKZ_OK;
}
int akz_debug_libm_eval(akz_ctx* c, const float* d_a, const float* d_b, float* d_out3, uint64_t n, int fma) {
    AKZ_TRY(bind(c));
    if (!d_a || !d_b || !d_out3) return AKZ_ERR_INVALID_ARG;
    launch::libm_eval(c->stream, d_a, d_b, d_out3, n, fma != 0, nullptr);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
int akz_debug_gates(const akz_gate** rows, uint64_t* n) {
    if (!rows || !n) return AKZ_ERR_INVALID_ARG;
    size_t k = 0;
    *rows = gates::table(&k);
    *n = k;
    return AKZ_OK;
}
int akz_debug_kernel_rows(akz_ctx* c, akz_kernel_row* out, uint32_t cap, uint32_t* n_rows, int reset) {
    AKZ_TRY(bind(c));
    if (!n_rows || (cap && !out)) return AKZ_ERR_INVALID_ARG;
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    resolve_spans(c);
    std::lock_guard<std::mutex> lk(c->ev_m);
    *n_rows = (uint32_t)c->rows.size();
    for (size_t i = 0; i < c->rows.size() && i < cap; ++i) out[i] = c->rows[i];
    if (reset) {
        // spans still in flight keep their row index: the table keeps its rows and only the figures start again
        for (akz_kernel_row& r : c->rows) r.launches = 0, r.px = 0, r.px_steps = 0, r.ms = 0.0;
    }
    return AKZ_OK;
}
int akz_debug_rcp_f64_to_f32(akz_ctx* c, const double* d_x, float* d_out, uint64_t n) {
    AKZ_TRY(bind(c));
    if (!d_x || !d_out) return AKZ_ERR_INVALID_ARG;
    if (n) launch::rcp_f64_to_f32(c->stream, d_x, d_out, n);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
int akz_debug_set_match_chunks(akz_ctx* c, uint32_t pair_chunks, uint32_t set_chunks) {
    if (!c || set_chunks > 16) return AKZ_ERR_INVALID_ARG;
    c->dbg_pair_chunks = pair_chunks;
    c->dbg_set_chunks = set_chunks;
    for (akz_ctx* l : c->lanes) {
        l->dbg_pair_chunks = pair_chunks;
        l->dbg_set_chunks = set_chunks;
    }
    return AKZ_OK;
}

}  // extern "C"

