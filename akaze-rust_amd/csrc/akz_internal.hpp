// Internal declarations shared by the host-side translation units of libakaze_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/akaze_hip.h"
#include "../../include/akaze_hip_debug.h"

namespace akz {

void set_error(const std::string& msg);
// host cores this process can count on: affinity mask, cgroup CPU quota, ranks per host (akz_api.cpp)
unsigned host_cpu_share();

// remove_outliers (estimate_fundamental_matrix.rs:99-165) with the trials optionally run by the caller's hook (akz_ransac.cpp;
// match_features hands in the device kernel of akz_fmatrix.hip): (x0, y0, x1, y1 of every match, n, 8 sample indices per
// trial, trials, epsilon_model, epsilon_inlier) -> 9 model floats and the inlier count (-1: no model) per trial
using TrialsOnDevice = std::function<int(const float*, const float*, const float*, const float*, uint32_t, const uint32_t*, uint32_t,
                                         float, float, float*, int32_t*)>;
int remove_outliers_impl(const akz_keypoint* keypoints_0, uint64_t n0, const akz_keypoint* keypoints_1, uint64_t n1,
                         const akz_match* matches, uint64_t n_matches, uint64_t num_trials, float epsilon_model,
                         float epsilon_inlier, akz_match* out, uint64_t* n_out, const TrialsOnDevice& trials_on_device);

#define AKZ_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::akz::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));            \
            return AKZ_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

// The `random` crate's default source (Xorshift128+), one per thread, seeded [42, 69] (akz_ransac.cpp)
struct DefaultSource {
    uint64_t s0 = 42, s1 = 69;
    uint64_t next() {
        uint64_t x = s0;
        const uint64_t y = s1;
        s0 = y;
        x ^= x << 23;
        x ^= x >> 17;
        x ^= y ^ (y >> 26);
        s1 = x;
        return x + y;
    }
};
DefaultSource& default_source();

#define AKZ_TRY(expr)            \
    do {                         \
        int _s = (expr);         \
        if (_s != AKZ_OK) return _s; \
    } while (0)

// one query set against sets that lie anywhere in one block of 64-byte rows (set k: rows set_first[k] .. + set_rows[k]);
// with d_n_cols also the opposite direction of every block (akz_descriptor_match_sets_mutual_device) -- akz_api.cpp
int match_sets_at(akz_ctx* c, const uint8_t* d_q, uint64_t n0, const uint8_t* d_rows, const uint64_t* set_first,
                  const uint64_t* set_rows, uint64_t n_sets, uint64_t distance_threshold, double lowes_ratio, akz_match* d_out,
                  uint64_t* d_n_out, akz_match* d_out_cols, uint64_t* d_n_cols, int side = 0);
// side 1 of match_sets_at enqueues on this stream with its own scratch (nullptr: the matcher mode has one side only); the
// caller orders it after its inputs and joins it before its outputs are read
hipStream_t match_side_stream(akz_ctx* c);

// streams of another component that run beside a context's: each slot ends up on a hardware queue and a pipe that the
// context's caller, coarse and finish streams do not use (a colliding stream is destroyed and replaced) -- akz_api.cpp
int place_streams_beside(akz_ctx* c, hipStream_t* slots, int n_slots, int* still_shared);
// what the stream-placement probe concludes from its timings (akz_api.cpp; akz_debug_placement_verdict for the tests):
// spin_pair_ms = two spins of spin_ms on two streams, first start to second end; tiny_pair_ms = 24 + 24 interleaved tiny
// kernels on two streams (negative: not measured); tiny_alone_ms = 24 of them on one stream
enum { kPlaceQueue = 1, kPlacePipe = 2, kPlaceAmbiguous = 4 };
int placement_verdict(float spin_pair_ms, float tiny_pair_ms, float tiny_alone_ms, float spin_ms);

// ---- host-side planning (akz_plan.cpp) ---------------------------------------------------
struct LevelPlan {
    double etime = 0, esigma = 0;
    uint32_t octave = 0, sublevel = 0, sigma_size = 0;  // EvolutionStep scalars
    uint32_t w = 0, h = 0;                               // plane size of the level
    uint32_t det_sigma = 0;                              // round(esigma*derivative_factor/2^octave)
    std::vector<double> tau;                             // fed_tau_steps
};
int config_validate(const akz_config& c);
int fed_tau_by_process_time(double T, int M, double tau_max, bool reordering, std::vector<double>& tau);
std::vector<float> gaussian_kernel(float sigma, size_t kernel_size);
size_t gaussian_kernel_size(float sigma);
void scharr_kernels(uint32_t scale, std::vector<float>& main_axis, std::vector<float>& off_axis);
int build_plan(uint32_t w, uint32_t h, const akz_config& cfg, std::vector<LevelPlan>& plan);

// A separable-filter pass as a sparse tap list: out = sum_i wgt[i] * in(.. + off[i]) evaluated
// left to right starting from 0.0f.  Dense Gaussian: 2hw+1 taps; Scharr: taps at -s, 0, +s.
constexpr int kMaxTaps = 13;
struct Taps {
    int n = 0;
    int hw = 0;  // half width of the full kernel (border clamp distance)
    int off[kMaxTaps] = {0};
    float wgt[kMaxTaps] = {0};
};
int taps_from_dense(const float* k, uint32_t ntaps, Taps& t);
Taps taps_scharr_main(uint32_t scale);
Taps taps_scharr_off(uint32_t scale);

// ---- device kernels' launch wrappers (akz_kernels.hip) -----------------------------------
struct Candidate {  // one NMS survivor (scale_space_extrema.rs:32-42 + bounds :80-87)
    uint32_t level;
    uint32_t idx;  // flat index w*y + x in the level
    float v, xp, xm, yp, ym;
    uint32_t img;  // image of the batch
};
struct KpParam {  // per-keypoint input of the orientation / descriptor kernels
    float xf, yf;     // point / ratio
    float scale;      // round(0.5*size/ratio)
    uint32_t level;
    uint32_t img;     // image of the batch
    uint32_t _pad[3];
};
struct OrientOut {
    float sum_x, sum_y;
    uint32_t found;
    uint32_t angle_bits;  // the angle as f32 bits where the device formed it (k_mldb with DeviceAngles); unused otherwise
};
struct MatchRec {  // per query descriptor
    uint32_t min_d, second_d, min_j, _pad;
};
struct LevelPtrs {  // device addresses of the planes one kernel needs, image 0 of the batch
    const float* lt;
    const float* lx;
    const float* ly;
    uint32_t w, h;
    uint64_t stride;  // elements between consecutive images
};
constexpr int kMaxLevels = 64;
constexpr int kRel1 = 12, kRel2 = 6;  // neighbour lists of the keypoint selection (akz_sort.hip, k_relations): 36-byte rows (four next-level places overflowed on enough candidates of a 4K frame for k_select's second pass to spend half its time scanning for them).  Six earlier neighbours overflowed on a third of a 4K frame's fine-level candidates -- the host then scans the partner levels itself -- : host selection 0.79 -> 0.45 ms per 4K frame with twelve (nine suffice; sixteen give nothing more)
struct LevelTable {
    LevelPtrs lv[kMaxLevels];
};

namespace launch {
void filter_h_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const Taps& t);
void filter_h_u8(hipStream_t s, const uint8_t* in, float* out, uint32_t w, uint32_t h, uint32_t n, const Taps& t);
void filter_v_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const Taps& t);
void delay(hipStream_t s, uint32_t microseconds);  // one wave spinning on the 100 MHz counter (stream-placement probe)
void half_size(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n);
void pm_g2(hipStream_t s, const float* lx, const float* ly, float* out, uint32_t w, uint32_t h, uint32_t n,
           const double* d_k, uint32_t k_scale_pow);
void flow(hipStream_t s, const float* lsmooth, float* lflow, uint32_t w, uint32_t h, uint32_t n, const double* d_k,
          uint32_t k_scale_pow);
void fed_step(hipStream_t s, const float* lt_in, const float* lflow, float* lt_out, float* lstep, uint32_t w,
              uint32_t h, uint32_t n, float half_tau);
// n_steps <= 16 explicit steps in one launch of k_fed_own (LDS tile + register ownership, temporally fused)
// next != NULL: the launch also writes the NEXT level's Lsmooth and Lflow from this level's final Lt (k_fed_own's epilogue;
// only where fed_epilogue_supported says so)
struct FedNextPrep {
    float* lsmooth;
    float* lflow;
    const float* g3;     // gaussian_kernel(1.0, 3)
    const double* d_k;
    uint32_t k_pow;
};
bool fed_epilogue_supported(uint32_t w, uint32_t h, uint32_t n_steps);
void fed_fused(hipStream_t s, const float* lt_in, const float* lflow, float* lt_out, float* lstep, uint32_t w, uint32_t h,
               uint32_t n, const float* half_taus, uint32_t n_steps, const FedNextPrep* next = nullptr);
// workgroups of a 9..16-step launch (64 x 10 tiles)
uint64_t fed_deep_workgroups(uint32_t w, uint32_t h, uint32_t n);
void contrast_max(hipStream_t s, const float* blurred, uint32_t w, uint32_t h, uint32_t n,
                  unsigned long long* d_hmax_bits);
void contrast_hist(hipStream_t s, const float* blurred, uint32_t w, uint32_t h, uint32_t n,
                   const unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t* d_hist);
void contrast_final(hipStream_t s, const unsigned long long* d_hmax_bits, const uint32_t* d_hist, uint32_t nbins,
                    double percentile, uint32_t n, double* d_k);
// ---- fused LDS stencil chains (akz_stencil.hip) ----
bool blur_fused_supported(uint32_t ntaps);
void blur_fused_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k,
                    uint32_t ntaps);
void blur_fused_u8(hipStream_t s, const uint8_t* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k,
                   uint32_t ntaps);
// prev: the previous level's final Lt (pw x ph).  half: this level starts a new octave; then the 2x2
// mean is written to lt_out as well.  g3: the 3 Gaussian taps of sigma 1.
void prep_fused(hipStream_t s, const float* prev, bool half, float* lt_out, float* lsmooth, float* lflow, uint32_t w,
                uint32_t h, uint32_t pw, uint32_t ph, uint32_t n, const float* g3, const double* d_k, uint32_t k_pow);
bool detector_fused_supported(uint32_t sigma);
void detector_fused(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx, float* lyy,
                    float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n);
// first + second derivatives, Ldet and the extrema candidates of one level in two launches
bool detector_tiled_fused_supported(uint32_t sigma);
// the same kernel over a set of levels that share sigma_size (one launch for several small levels)
struct DetLevelDesc {
    const float* lsmooth;
    float *lx, *ly, *lxx, *lyy, *lxy, *ldet;
    uint32_t w, h, level;
    float border_m;
};
uint32_t detector_tiled_set_max();
void detector_tiled_set(hipStream_t s, uint32_t sigma, const DetLevelDesc* levels, uint32_t nlevels, uint32_t n, float thr,
                        Candidate* d_cand, uint32_t cap, uint32_t* d_count);
void detector_tiled_fused(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx,
                          float* lyy, float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level,
                          float thr, float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count);
// streaming 5-tap gaussian_blur (akz_stream.hip)
bool blur5_stream_supported(uint32_t w, uint32_t h, uint32_t ntaps, bool is_u8);
void blur5_stream_u8(hipStream_t s, const uint8_t* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k);
void blur5_stream_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k);
// streaming contrast-factor passes (akz_stream.hip): max and histogram of the gradient of blur(in) in two launches
uint32_t march_band_rows(int kind, uint32_t w, uint32_t h, uint32_t n, int S, int32_t* cs_ce, uint32_t cap);  // test hook
bool blur5_march_supported(uint32_t w, uint32_t h, uint32_t ntaps);  // akz_march.hip: the same blur as a column march
void blur5_march_u8(hipStream_t s, const uint8_t* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k);
void blur5_march_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k);
// the same two passes as column marches (akz_march.hip: two columns per thread, bound by the plane read, not by arithmetic)
bool contrast_march_supported(uint32_t w, uint32_t h, uint32_t ntaps, uint32_t nbins);
void contrast_march(hipStream_t s, const float* in, uint32_t w, uint32_t h, uint32_t n, const float* g3,
                    unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t* d_hist, double* d_thr);
void contrast_thresholds(hipStream_t s, const unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t n, double* d_thr);
bool contrast_stream_supported(uint32_t w, uint32_t h, uint32_t ntaps, uint32_t nbins);
void contrast_stream(hipStream_t s, const float* in, uint32_t w, uint32_t h, uint32_t n, const float* g3,
                     unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t* d_hist, double* d_thr);
// streaming form of prep_fused (akz_stream.hip)
bool prep_stream_supported(uint32_t w, uint32_t h);
void prep_stream(hipStream_t s, const float* prev, bool half, float* lt_out, float* lsmooth, float* lflow, uint32_t w,
                 uint32_t h, uint32_t pw, uint32_t ph, uint32_t n, const float* g3, const double* d_k, uint32_t k_pow);
// one-pass column march (akz_march.hip): 256-thread workgroups own 512-column strips, two columns per thread
bool detector_march_supported(uint32_t sigma, uint32_t w, uint32_t h, float border_m, bool nms);
void detector_march(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx, float* lyy,
                    float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level, float thr,
                    float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count);
// level preparation + the first n_steps <= 4 diffusion steps of a level in one launch (akz_march.hip, k_level_march)
bool head_fused_supported(uint32_t w, uint32_t h, uint32_t ntaps0, uint32_t ntaps1);
// (d_zero_word: a 32-bit word the kernel clears -- the job's candidate counter -- or NULL)
void head_fused_u8(hipStream_t s, const uint8_t* in, float* lt0, float* blurred, float* gx, float* gy, uint32_t w, uint32_t h, uint32_t n,
                   const float* k5, const float* g3, unsigned long long* d_smax_bits, uint32_t* d_zero_word = nullptr);
void head_fused_f32(hipStream_t s, const float* in, float* lt0, float* blurred, float* gx, float* gy, uint32_t w, uint32_t h, uint32_t n,
                    const float* k5, const float* g3, unsigned long long* d_smax_bits, uint32_t* d_zero_word = nullptr);
// the histogram + percentile pass over the stored Scharr pair (gx, gy: k_head's), and Lflow = pm_g2 of the same pair
void contrast_hist_final(hipStream_t s, const float* gx, const float* gy, uint32_t w, uint32_t h, uint32_t n, unsigned long long* d_smax_bits,
                         uint32_t nbins, uint32_t* d_hist, uint32_t* d_done, double percentile, double* d_k);
void flow_from_pair(hipStream_t s, const float* gx, const float* gy, float* lflow, uint32_t w, uint32_t h, uint32_t n, const double* d_k,
                    uint32_t k_pow);
void march_min_band_rows(int detector, int level);  // measurement hook (process-wide): 0 = the planners' own rules
bool level_march_supported(uint32_t w, uint32_t h);
// pw, ph != 0: `prev` is the previous octave's last Lt (pw x ph) and the level starts from its 2x2 mean, formed inside the
// kernel (level_march_half_supported)
bool level_march_half_supported(uint32_t w, uint32_t h, uint32_t pw, uint32_t ph, uint32_t n_steps);
void level_march(hipStream_t s, const float* prev, float* lsmooth, float* lflow, float* lt_out, float* lstep, uint32_t w,
                 uint32_t h, uint32_t n, const float* g3, const double* d_k, uint32_t k_pow, const float* half_taus,
                 uint32_t n_steps, uint32_t pw = 0, uint32_t ph = 0);
// the remaining levels of the pyramid, from the first one whose image fits a compute unit, in ONE launch with one
// workgroup per image (akz_resident.hip, k_octave_resident): preparation and every diffusion step of every level
struct ResidentLevel {
    float *lt, *lsmooth, *lflow, *lstep;  // image 0 of the batch; lstep may be null
    uint32_t w, h;
    bool half;              // the level opens an octave (2x2 mean of the previous level's final Lt)
    uint32_t n_tau;         // >= 1
    const float* half_tau;  // 0.5f * (tau as f32) per step
    uint32_t k_pow;         // octave
};
constexpr int kResidentMaxLevels = 10;
constexpr int kResidentMaxSteps = 448;
bool octave_resident_supported(uint32_t w, uint32_t h);
void octave_resident(hipStream_t s, const float* prev, uint32_t pw, uint32_t ph, uint32_t n, const ResidentLevel* levels,
                     uint32_t n_levels, const float* g3, const double* d_k);
bool detector_nms_fused_supported(uint32_t sigma);
void detector_nms_fused(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx,
                        float* lyy, float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level,
                        float thr, float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count);
void rcp_f64_to_f32(hipStream_t s, const double* x, float* out, uint64_t n);  // test hook (akz_pm_g2.hpp)
void accumulate(hipStream_t s, float* a, const float* b, uint64_t count);  // a += b (image.rs:218-231); a may equal b
void ldet(hipStream_t s, const float* lxx, const float* lyy, const float* lxy, float* out, uint64_t count,
          float sigma_quat);
// the candidate list into scan order on the device (akz_sort.hip): by image, level, flat pixel index.  false: the
// batch needs more key bits than the sort takes (the host then sorts)
size_t sort_candidates_scratch(uint32_t cap, uint64_t max_px, uint32_t n_levels, uint32_t n_images);
uint32_t sort_small_capacity();  // one image's list up to this capacity is sorted by one launch of one workgroup:
// false: not this list (too long, too many rows) -- the caller takes sort_candidates_device
// Lists of any length, several images, up to 65 536 (image, level, row) buckets: the same counting sort in four launches
// (scratch: sort_candidates_buckets_scratch(cap) bytes; scratch_is_new: the buffer has just been (re)allocated).  false: too
// many buckets -- the caller takes sort_candidates_device.  rel_scratch as for sort_candidates_rows.
size_t sort_candidates_buckets_scratch(uint32_t cap);
bool sort_candidates_buckets(hipStream_t s, const Candidate* d_cand, uint32_t cap, const uint32_t* d_count, const uint32_t* level_w,
                             const uint32_t* level_h, uint32_t n_levels, uint32_t n_images, void* scratch, bool scratch_is_new,
                             Candidate* d_sorted, uint32_t* d_zero, void* rel_scratch);
// rel_scratch (candidate_relations_bytes for ONE image, or null): the sort leaves candidate_relations' level / row tables there
// (its bucket starts are those tables) -- pass tables_ready to candidate_relations
bool sort_candidates_rows(hipStream_t s, const Candidate* d_cand, uint32_t cap, const uint32_t* d_count, const uint32_t* level_w,
                          const uint32_t* level_h, uint32_t n_levels, Candidate* d_sorted, uint32_t* d_zero, void* rel_scratch);
bool sort_candidates_device(hipStream_t s, const Candidate* d_cand, uint32_t cap, const uint32_t* d_count, uint64_t max_px,
                            uint32_t n_levels, uint32_t n_images, void* scratch, Candidate* d_sorted, uint32_t* d_zero = nullptr);
// who can be within `size` of whom (akz_sort.hip, k_relations): per candidate of the SORTED list kRel1 indices of earlier
// candidates of its own / the previous level and kRel2 of the next level, relative to its image's first candidate
size_t candidate_relations_bytes(uint32_t cap, const uint32_t* level_h, uint32_t n_levels, uint32_t n_images);
void candidate_relations(hipStream_t s, const Candidate* d_sorted, uint32_t cap, const uint32_t* d_count, const float* size, const float* ratio,
                         const uint32_t* level_w, const uint32_t* level_h, uint32_t n_levels, uint32_t n_images, void* scratch,
                         uint16_t** d_rel_out, uint32_t** d_flags_out, void* sel_scratch = nullptr, bool tables_ready = false);
// The selection itself on the device (akz_sort.hip: k_sel_prepare, k_select, k_sel_pack; akz_select.hpp): dependency rounds
// over the neighbour lists, one workgroup per image.  sel_scratch (select_device_bytes) must have been handed to
// sort_candidates_device (d_zero = select_device_revcnt(...)) and to candidate_relations of the same list.  Leaves, in image
// order, the selected keypoints (d_recs: 32-byte records -- x, y, response, level, then room for the keypoint's OrientOut,
// which orientation_counted(..., out_stride 2) fills in: one copy brings both to the host) and the parameters of the keypoint
// kernels (d_pars), room for `cap` of each; *d_hdr_out: per image 16 words {keypoints, extrema, status, looks, four phase
// times, the list's length, the image's flags, its contrast factor (2 words), 4 unused} -- status != 0: the image is for the
// host's selection (an overflowed list, too many candidates) and the whole job should take that path --; *d_total_out: the
// keypoint count of the job
size_t select_device_bytes(uint32_t cap, uint32_t n_images);
uint32_t* select_device_revcnt(void* sel_scratch, uint32_t cap, uint32_t n_images);
void select_device(hipStream_t s, const Candidate* d_sorted, uint32_t cap, const uint32_t* d_count, const float* size, const float* ratio,
                   const uint32_t* level_w, const uint32_t* level_h, uint32_t n_levels, uint32_t n_images, const void* rel_scratch,
                   void* sel_scratch, const double* d_k, void* d_recs, KpParam* d_pars, uint32_t** d_hdr_out, uint32_t** d_total_out);
// candidates of all images are appended to ONE list (d_count is a single counter, cap the list capacity)
void nms(hipStream_t s, const float* ldet, uint32_t w, uint32_t h, uint32_t n, uint64_t img_stride, uint32_t level,
         float thr, float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count);
void orientation(hipStream_t s, const LevelTable& lt, const KpParam* d_kp, uint32_t nkp,
                 unsigned long long window_mask, uint32_t n_windows, OrientOut* d_out);
// the same with the keypoint count still on the device (*d_nkp <= max_kp: the grid is sized for max_kp)
void orientation_counted(hipStream_t s, const LevelTable& lt, const KpParam* d_kp, const uint32_t* d_nkp, uint32_t max_kp,
                         unsigned long long window_mask, uint32_t n_windows, OrientOut* d_out, uint32_t out_stride);
// d_cosi: (cosf(angle), sinf(angle)) per keypoint from the host libm (descriptors.rs:55-56)
void mldb(hipStream_t s, const LevelTable& lt, const KpParam* d_kp, const float* d_cosi, uint32_t nkp,
          uint32_t channels, uint8_t* d_desc64);
// the same with the keypoint count still on the device and the angle, cosf, sinf of every keypoint formed ON the device from
// its orientation sums (akz_libm.hpp; d_sums[i * sums_stride]): the angle is left in d_sums[..].angle_bits, *d_flag is raised
// if an argument is outside what the device forms cover
// (keypoints [first, min(last, *d_nkp)): the grid covers first .. last; first is rounded down to the kernel's group of four)
// `mirror`: the kernel also stores every keypoint's 32-byte record and descriptor row, and the selection's headers, into the
// host's (pinned, device-visible) buffers -- the waited-for job's results without copies behind the kernel
struct MldbMirror {
    void* host_recs;        // 32 bytes per keypoint
    const void* d_hdr;      // the selection's headers on the device, hdr_bytes (a multiple of 16) ...
    void* host_hdr;         // ... and where the host wants them
    uint32_t hdr_bytes;
    uint8_t* host_desc;     // 64 bytes per keypoint, or NULL
};
void mldb_counted(hipStream_t s, const LevelTable& lt, const KpParam* d_kp, const uint32_t* d_nkp, uint32_t first, uint32_t last, OrientOut* d_sums,
                  uint32_t sums_stride, bool libm_fma, uint32_t* d_flag, uint32_t channels, uint8_t* d_desc64, const MldbMirror* mirror = nullptr);
void libm_eval(hipStream_t s, const float* a, const float* b, float* out3, uint64_t n, bool fma, uint32_t* d_flag);
uint32_t match_num_chunks(uint32_t n0, uint32_t n1);
// Both scans write one record per (chunk of the train set, query): d_rec[chunk * n0 + query]; match_compact merges a
// query's chunk records (ascending rows: the lowest row wins among equal minima) and applies the ratio test.
// rows_le_61: rows are M-LDB descriptors (at most 61 bytes): bytes 61..63 are padding and not compared
void match(hipStream_t s, const uint8_t* d0, uint32_t n0, const uint8_t* d1, uint32_t n1, uint32_t threshold,
           bool rows_le_61, uint32_t chunks, MatchRec* d_rec);
// the same scan on the matrix cores (akz_match.hip): descriptor bits unpacked to int8, distances from one integer
// GEMM; identical records.  Rows of the unpacked images are padded (match_mfma_rows).
uint32_t match_mfma_rows(uint32_t n, bool queries);
uint32_t match_mfma_chunks(uint32_t n0, uint32_t n1, uint32_t forced = 0);
uint32_t match_mfma_multi_chunks(uint32_t n0, uint32_t n_sets, uint32_t avg_tiles, uint32_t forced = 0);  // chunks per set of a multi-set launch
// bound (queries only): n_bound arrays of n_pad per-query pruning bounds, set to threshold; d_tiles (train images of
// several sets): per LDS tile {first source row, valid rows}
void unpack_bits(hipStream_t s, const uint8_t* d, uint32_t n, uint32_t n_pad, bool query, uint8_t* out8, uint32_t* pop,
                 uint32_t* bound, uint32_t threshold, uint32_t n_bound, const uint32_t* d_tiles, bool fp4 = false);
// both sets of a pair call in one launch (query form with one bound array, train form)
// one workgroup per RANSAC trial: model of its eight samples (akz_fmatrix.hpp) + inlier count over all matches (akz_fmatrix.hip)
void ransac_trials(hipStream_t s, const float* d_pts, uint32_t n_matches, const uint32_t* d_samples, uint32_t trials, float epsilon_model,
                   float epsilon_inlier, float* d_models, int32_t* d_inliers);
void unpack_pair(hipStream_t s, const uint8_t* dq, uint32_t nq, uint32_t q_pad, uint8_t* outq, uint32_t* popq, uint32_t* bound, uint32_t threshold,
                 const uint8_t* dt, uint32_t nt, uint32_t t_pad, uint8_t* outt, uint32_t* popt, bool fp4);
uint32_t match_mfma_tile_rows();
uint32_t match_mfma_query_block();
struct MatchChunkHost {  // = MatchChunk of akz_match.hip
    uint32_t t_begin, t_end, row0, n_rows, bound_off, record;
};
void match_mfma_multi(hipStream_t s, const uint8_t* q8, const uint32_t* qpop, uint32_t n0, const uint8_t* t8,
                      const void* d_table, uint32_t n_chunks, uint32_t threshold, uint32_t* bound, MatchRec* d_out, bool fp4 = false);
void match_mfma(hipStream_t s, const uint8_t* q8, const uint32_t* qpop, uint32_t n0, const uint8_t* t8, uint32_t n1,
                uint32_t threshold, uint32_t* bound, uint32_t chunks, MatchRec* d_rec, bool fp4 = false);
// both directions of a multi-set launch (FP4 form): see k_match_fp4<.., COLS> in akz_match.hip
struct MatchColSetHost {  // = ColSet of akz_match.hip
    uint32_t row0, rows, out0;
};
uint32_t match_cols_seed_rows(uint32_t n0);
void match_cols_seed(hipStream_t s, const uint8_t* q4, uint32_t n0, const uint8_t* t4, uint32_t t_rows, uint32_t threshold,
                     uint32_t* d_bound, MatchRec* d_seed, unsigned long long* cbest, uint32_t* csecond);
void match_fp4_multi_mutual(hipStream_t s, const uint8_t* q4, uint32_t n0, const uint8_t* t4, const void* d_table, uint32_t n_chunks,
                            uint32_t threshold, uint32_t* bound, MatchRec* d_out, unsigned long long* cbest, uint32_t* csecond);
void match_compact_cols(hipStream_t s, const unsigned long long* cbest, const uint32_t* csecond, const void* d_sets, uint32_t n_sets,
                        uint32_t threshold, double ratio2, akz_match* d_out, unsigned long long* d_n_out);
// set k's `chunks` chunk records at d_rec + k * chunks * n0, its matches at d_out + k * n0, its count at d_n_out[k]
void match_compact_sets(hipStream_t s, const MatchRec* d_rec, uint32_t n0, uint32_t n_sets, uint32_t chunks, uint32_t threshold,
                        double ratio2, akz_match* d_out, unsigned long long* d_n_out);
void match_compact(hipStream_t s, const MatchRec* d_rec, uint32_t n0, uint32_t chunks, uint32_t threshold, double ratio2,
                   akz_match* d_out, unsigned long long* d_n_out);
// the same merge as a kernel of its own, one thread per query (many chunks): d_out[query]
void match_merge(hipStream_t s, const MatchRec* d_part, uint32_t n0, uint32_t chunks, uint32_t threshold, MatchRec* d_out);
// merge + ratio test + ordered compaction of one set in one launch; d_state: match_merge_compact_state_bytes(n0) bytes that
// were zero when allocated and are only ever written by this kernel; epoch: different for every call on that buffer, never 0
size_t match_merge_compact_state_bytes(uint32_t n0);
void match_merge_compact(hipStream_t s, const MatchRec* d_part, uint32_t n0, uint32_t chunks, uint32_t threshold, double ratio2,
                         akz_match* d_out, unsigned long long* d_n_out, void* d_state, uint32_t epoch);
}  // namespace launch

// ---- host keypoint logic (akz_keypoints.cpp) ---------------------------------------------
struct HostKeypoint {
    float x, y, response, size;
    uint32_t octave, class_id;
    float angle;
    // level coordinates + 4-neighbour Ldet values carried from the NMS kernel
    uint32_t lx, ly;
    float xp, xm, yp, ym;
};
// unordered NMS survivors of one image into the reference's scan order: level, then flat index
void sort_candidates(std::vector<Candidate>& cands, const std::vector<LevelPlan>& plan);
// scale_space_extrema.rs:12-132 on raster-ordered candidates, then :141-178 (refinement w/o orientation)
void select_keypoints(const Candidate* cands_sorted, size_t n_cands, const std::vector<LevelPlan>& plan,
                      const akz_config& cfg, std::vector<HostKeypoint>& out, uint64_t* n_extrema);
// the same selection from the device's neighbour lists (launch::candidate_relations; rel: k1 + k2 entries per candidate)
void select_keypoints_rel(const Candidate* cands_sorted, size_t n_cands, const uint16_t* rel, int k1, int k2, const std::vector<LevelPlan>& plan,
                          const akz_config& cfg, std::vector<HostKeypoint>& out, uint64_t* n_extrema);
// the per-level constants of the selection (f32, as select_keypoints forms them)
void selection_level_constants(const std::vector<LevelPlan>& plan, const akz_config& cfg, std::vector<float>& size, std::vector<float>& ratio);
// which of the sliding windows of compute_main_orientation contain atan2f(a, a), a > 0
void orientation_windows(unsigned long long* mask, uint32_t* n_windows);
float border_margin(const LevelPlan& lv, const akz_config& cfg);  // smax * sigma_size (f32)

}  // namespace akz

// ---- plane stores (device code only) --------------------------------------------------------------------------
// The tiled stencil / FED kernels use ordinary stores: streaming (nontemporal) stores measured the same there in
// round 1 (those kernels are bound by their on-chip dependency chains); the column march (akz_march.hip), which IS
// bound by the store path, uses them (+20 %).
#if defined(__HIPCC__)
namespace akz {
typedef float akz_f4a __attribute__((ext_vector_type(4)));                // 16-byte aligned group of four pixels
typedef float akz_f4u __attribute__((ext_vector_type(4), aligned(4)));    // dword-aligned group of four pixels
__device__ __forceinline__ void plane_store(float* p, float v) { *p = v; }
__device__ __forceinline__ void plane_store4(float* p, float a, float b, float c, float d) {  // p 16-byte aligned
    const akz_f4a t = {a, b, c, d};
    *reinterpret_cast<akz_f4a*>(p) = t;
}
__device__ __forceinline__ void plane_store4u(float* p, akz_f4a t) {  // p dword-aligned
    *reinterpret_cast<akz_f4u*>(p) = t;
}
}  // namespace akz
#endif
