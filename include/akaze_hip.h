/* akaze_hip.h — C ABI of the MI355X-native A-KAZE hot path (libakaze_hip.so).
 *
 * The reference crate (indianajohn/akaze-rust) has no FFI/plugin boundary; its
 * drop-in boundary is the public Rust API.  Every entry point below names the
 * reference item it replaces (paths relative to the reference repository root).
 * A Rust shim that re-exports the reference signatures over these symbols is in
 * akaze-rust_amd/rust/ and described in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns AKZ_OK (0) or a negative akz_status; nothing throws
 *     or aborts across the ABI.  akz_last_error() gives a thread-local message.
 *   - an akz_ctx is bound to one GPU and one HIP stream and is NOT thread-safe;
 *     use one context per host thread / per GPU.
 *   - images are row-major, contiguous, index = w*y + x, exactly like
 *     GrayFloatImage (akaze/src/types/image.rs:32-36).  A batch of n images of
 *     equal size is n such planes back to back (plane stride = w*h elements).
 *   - pointers named d_* are DEVICE pointers (hipMalloc / torch CUDA tensors);
 *     all others are host pointers.
 *   - "op" entry points enqueue on the context's stream and return without
 *     synchronising unless they hand a value back to the host.
 */
#ifndef AKAZE_HIP_H
#define AKAZE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 3): additions only -- akz_extract_begin_host_*, akz_ctx_set_host_threads, akz_ctx_set_eager_finish,
   akz_gather_image_rows, akz_match_all_pairs / akz_pairs_*, AKZ_INPUT_READY; the gather's overflow protocol.  Every
   entry point of version 2 keeps its signature and meaning.
   4 (round 4): akz_descriptor_match_sets_mutual_device and akz_pairs_holder added; akz_match_all_pairs matches every
   unordered image pair once and a rank's akz_pairs holds both lists of the pairs whose lead image it owns (version 3: the
   ordered pairs whose query image it owns); akz_ctx_set_eager_finish covers the context's own jobs and defaults to on;
   the kernel-family selectors and the synthetic-frame generator moved to akaze_hip_debug.h (still exported).  No
   signature changed.
   5 (round 5): additions -- akz_comm_create_external / akz_gather_blocks / akz_gather_deliver (the exchange carried by the
   caller), akz_comm_set_timeout, akz_pairs_totals, akz_ctx_warmup; akz_profile grew by the placement_* fields at its END
   (a caller built against version 4 must not pass its smaller struct to akz_ctx_get_profile); an akz_pairs may be freed
   after its communicator; AKZ_ERR_TIMEOUT.  No signature changed.
   6 (round 6): additions -- akz_ctx_get_profile2 (size-checked; akz_ctx_get_profile writes the ABI-4 prefix of akz_profile
   only from now on); akz_op_scharr accepts both and neither order as the reference does; a communicator whose exchange
   timed out is abandoned (akz_comm_set_timeout); AKZ_COMM_HOST communicators, akz_gather_begin_image_rows, akz_pairs_plan,
   akz_pairs_lead_sets.  No signature changed. */
#define AKZ_ABI_VERSION 6

typedef enum akz_status {
    AKZ_OK = 0,
    AKZ_ERR_INVALID_ARG = -1,  /* null pointer, zero size, unsupported Config value    */
    AKZ_ERR_HIP = -2,          /* a HIP runtime call failed (message in akz_last_error) */
    AKZ_ERR_NO_DEVICE = -3,    /* no gfx950 device visible / extension cannot run        */
    AKZ_ERR_TOO_SMALL = -4,    /* image smaller than the stencil supports                */
    AKZ_ERR_OVERFLOW = -5,     /* a fixed-capacity device buffer (candidates) overflowed */
    AKZ_ERR_UNSUPPORTED = -6,  /* reference behaviour that does not terminate / panics   */
    AKZ_ERR_BUFFER = -7,       /* caller buffer too small                                */
    AKZ_ERR_IO = -8,           /* an image file cannot be read / written or is malformed */
    AKZ_ERR_NO_MEMORY = -9,    /* host allocation failed                                 */
    AKZ_ERR_TIMEOUT = -10      /* an exchange did not complete within akz_comm_set_timeout (a peer is missing or hung) */
} akz_status;

/* types::evolution::Config — akaze/src/types/evolution.rs:8-38 (defaults :40-55). */
typedef struct akz_config {
    uint32_t num_sublevels;
    uint32_t max_octave_evolution;
    double base_scale_offset;
    double initial_contrast; /* declared by the reference, never read */
    double contrast_percentile;
    uint64_t contrast_factor_num_bins;
    double derivative_factor;
    double detector_threshold;
    uint64_t descriptor_channels;
    uint64_t descriptor_pattern_size;
} akz_config;

/* types::keypoint::Keypoint — akaze/src/types/keypoint.rs:8-30 (point = (x, y)). */
typedef struct akz_keypoint {
    float x, y;
    float response;
    float size;
    uint64_t octave;
    uint64_t class_id;
    float angle;
    uint32_t _pad;
} akz_keypoint;

/* types::feature_match::Match — akaze/src/types/feature_match.rs:9-16. */
typedef struct akz_match {
    uint64_t index_0;
    uint64_t index_1;
    double distance;
} akz_match;

/* EvolutionStep image fields in declaration order — akaze/src/types/evolution.rs:71-89. */
typedef enum akz_plane {
    AKZ_LT = 0, AKZ_LSMOOTH = 1, AKZ_LX = 2, AKZ_LY = 3, AKZ_LXX = 4,
    AKZ_LYY = 5, AKZ_LXY = 6, AKZ_LFLOW = 7, AKZ_LSTEP = 8, AKZ_LDET = 9
} akz_plane;

/* akz_extract_* flags */
enum {
    /* keep every EvolutionStep plane resident so akz_fetch_plane can return it
       (what extract_features returns, akaze/src/lib.rs:193).  Without it only the
       planes later stages read (Lt, Lsmooth, Lx, Ly, Lflow, Ldet) are materialised and
       Lxx/Lyy/Lxy/Lstep are never written. */
    AKZ_KEEP_ALL_PLANES = 1u << 0,
    /* leave keypoints/descriptors on the device only (no D2H of descriptors) */
    AKZ_NO_HOST_DESCRIPTORS = 1u << 1,
    /* akz_extract_from_planes only: upload the planes, skip the extrema pass (keypoints come from the caller through
       akz_result_describe_keypoints) */
    AKZ_NO_DETECT = 1u << 2,
    /* akz_extract_begin_device_*: the caller promises that the frames are COMPLETE in device memory when the call is made
       -- nothing still pending on the context's stream (or on any other) writes them -- and stay untouched until the job
       is finished.  The first stages of a large batch (level-0 blur, contrast factor) then do not wait for the work the
       context's stream still holds for the batch before and run under that batch's kernels.  Without the flag the
       library has to assume that the frames are produced by whatever the caller enqueued on the stream before the call.
       (akz_extract_begin_host_* knows when its own upload is complete and always runs ahead.)  Results are identical.
       Takes effect only in a process started with GPU_MAX_HW_QUEUES >= 8 in its environment (threading):
       with the runtime's default of 4 in-order hardware queues one more busy stream would serialise the pipeline. */
    AKZ_INPUT_READY = 1u << 3
};

typedef struct akz_ctx akz_ctx;
typedef struct akz_result akz_result;
typedef struct akz_job akz_job; /* an extraction in flight, see akz_extract_begin_* */

/* ---- library ------------------------------------------------------------------------ */
int akz_abi_version(void);
const char* akz_last_error(void);
/* Config::default() — akaze/src/types/evolution.rs:40-55 */
void akz_config_default(akz_config* out);

/* ---- context ------------------------------------------------------------------------ */
/* stream: the hipStream_t every kernel and copy of this context is enqueued on, e.g.
   torch.cuda.current_stream().cuda_stream.  NULL is the device's default (null) stream, as in
   every HIP API.  akz_stream_create/destroy give non-torch hosts a private stream. */
int akz_ctx_create(int device, void* stream, akz_ctx** out);
int akz_stream_create(int device, void** stream_out);
int akz_stream_destroy(int device, void* stream);
/* Results and jobs of the context may be freed after it (they keep their host data; device accessors then
   return AKZ_ERR_INVALID_ARG). */
int akz_ctx_destroy(akz_ctx* ctx);
int akz_ctx_synchronize(akz_ctx* ctx);
void* akz_ctx_stream(akz_ctx* ctx);

/* plain device-memory helpers so a non-torch host (the Rust shim) needs no HIP bindings */
int akz_device_malloc(akz_ctx* ctx, size_t bytes, void** d_out);
int akz_device_free(akz_ctx* ctx, void* d_ptr);
int akz_memcpy_h2d(akz_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int akz_memcpy_d2h(akz_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* ---- host-side planning (scalar code the reference also runs on the CPU) ------------- */
/* ops::fed_tau::fed_tau_by_process_time — akaze/src/ops/fed_tau.rs:27-30 (+ :43-106).
   Writes min(*n, cap) step sizes.  AKZ_ERR_UNSUPPORTED where the reference never
   terminates (n == 1 with reordering, fed_tau.rs:95). */
int akz_fed_tau_by_process_time(double T, int M, double tau_max, int reordering, double* out, uint64_t cap,
                                uint64_t* n);
/* types::image::gaussian_kernel — akaze/src/types/image.rs:352-365 */
int akz_gaussian_kernel(float sigma, uint64_t kernel_size, float* out);
/* ops::derivatives::scharr_{main,off}_axis_kernel — akaze/src/ops/derivatives.rs:74-101 */
int akz_scharr_kernels(uint32_t scale, float* main_axis, float* off_axis);
/* types::evolution::allocate_evolutions — akaze/src/types/evolution.rs:135-161.
   Level sizes follow the chained half_size of create_nonlinear_scale_space (lib.rs:80-90). */
int akz_plan_num_levels(uint32_t w, uint32_t h, const akz_config* cfg, uint64_t* n_levels);
int akz_plan_level_info(uint32_t w, uint32_t h, const akz_config* cfg, uint64_t level, double* etime, double* esigma,
                        uint32_t* octave, uint32_t* sublevel, uint32_t* sigma_size, uint32_t* level_w,
                        uint32_t* level_h, uint32_t* detector_sigma, uint64_t* n_tau, double* tau, uint64_t tau_cap);

/* ---- per-op entry points on device planes (mirror of `pub mod ops` / `types::image`) -- */
/* types::image::horizontal_filter / vertical_filter incl. fill_border — image.rs:239-332 */
int akz_op_horizontal_filter(akz_ctx* ctx, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                             const float* taps, uint32_t ntaps);
int akz_op_vertical_filter(akz_ctx* ctx, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                           const float* taps, uint32_t ntaps);
/* types::image::gaussian_blur — image.rs:374-380 */
int akz_op_gaussian_blur(akz_ctx* ctx, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                         float sigma);
/* create_unit_float_image + gaussian_blur on 8-bit luma — image.rs:127-140, lib.rs:56 */
int akz_op_gaussian_blur_u8(akz_ctx* ctx, const uint8_t* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                            float sigma);
/* ImageFunctions::half_size — image.rs:102-118 ; out is (w/2) x (h/2) */
int akz_op_half_size(akz_ctx* ctx, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n);
/* ops::derivatives::scharr — derivatives.rs:112-130, all four order combinations as the reference has them:
 * x only / y only -> scharr_horizontal / scharr_vertical; both -> the horizontal derivative added to itself (:118-122 via
 * image.rs:218-231); neither -> a zero image (:127-128; sigma_size unused). */
int akz_op_scharr(akz_ctx* ctx, const float* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n, int x_order,
                  int y_order, uint32_t sigma_size);
/* pm_g2 — lib.rs:26-41 ; d_k: n contrast factors (double) on the device */
int akz_op_pm_g2(akz_ctx* ctx, const float* d_lx, const float* d_ly, float* d_out, uint32_t w, uint32_t h, uint32_t n,
                 const double* d_k);
/* ops::contrast_factor::compute_contrast_factor — contrast_factor.rs:18-71 ; writes n doubles to d_k_out */
int akz_op_contrast_factor(akz_ctx* ctx, const float* d_in, uint32_t w, uint32_t h, uint32_t n, double percentile,
                           double gradient_histogram_scale, uint64_t num_bins, double* d_k_out);
/* Lsmooth -> Lflow of one level: scharr(sigma 1) x2 + pm_g2 — lib.rs:98-105.
   d_k: per-image contrast factor of octave 0; k_scale_pow = number of x0.75 octave steps (lib.rs:84). */
int akz_op_flow(akz_ctx* ctx, const float* d_lsmooth, float* d_lflow, uint32_t w, uint32_t h, uint32_t n,
                const double* d_k, uint32_t k_scale_pow);
/* ops::nonlinear_diffusion::calculate_step, n_tau times — nonlinear_diffusion.rs:15-144.
   d_lt is updated in place (as the reference does); d_lstep (may be NULL) receives the
   last step's increment.  taus are the f64 step sizes of the level. */
int akz_op_fed_steps(akz_ctx* ctx, float* d_lt, const float* d_lflow, float* d_lstep, uint32_t w, uint32_t h,
                     uint32_t n, const double* taus, uint32_t n_tau);
/* compute_multiscale_derivatives_for_evolution + Ldet — detector_response.rs:8-14, :40-54.
   d_lxx/d_lyy/d_lxy may be NULL (not materialised). */
int akz_op_detector_response(akz_ctx* ctx, const float* d_lsmooth, uint32_t sigma_size, float* d_lx, float* d_ly,
                             float* d_lxx, float* d_lyy, float* d_lxy, float* d_ldet, uint32_t w, uint32_t h,
                             uint32_t n);

/* ---- the hot path: extract_features -------------------------------------------------- */
/* akaze::extract_features — akaze/src/lib.rs:167-194, minus image::open/to_luma (ingest is
   upstream of the path).  Host-buffer forms copy the frame(s) H2D first. */
int akz_extract_gray_u8(akz_ctx* ctx, const uint8_t* img, uint32_t w, uint32_t h, const akz_config* cfg,
                        uint32_t flags, akz_result** out);
int akz_extract_gray_f32(akz_ctx* ctx, const float* img, uint32_t w, uint32_t h, const akz_config* cfg,
                         uint32_t flags, akz_result** out);
/* n equally sized frames already resident in HBM; one result handle for the batch */
int akz_extract_device_u8(akz_ctx* ctx, const uint8_t* d_imgs, uint32_t w, uint32_t h, uint32_t n,
                          const akz_config* cfg, uint32_t flags, akz_result** out);
int akz_extract_device_f32(akz_ctx* ctx, const float* d_imgs, uint32_t w, uint32_t h, uint32_t n,
                           const akz_config* cfg, uint32_t flags, akz_result** out);

/* Two-phase form for streaming: begin enqueues the scale space, the detector response and the extrema
   candidates of a batch on the context's stream and returns WITHOUT synchronising; finish waits for
   that batch only, runs the host keypoint logic, orientation and descriptors (on an auxiliary stream)
   and hands the result over.  Begin the next batch before finishing the previous one and the host
   phase of one runs under the kernels of the other.  Up to 3 jobs in flight per context; a job must be
   finished (or abandoned) exactly once; the frames must stay valid until then. */
int akz_extract_begin_device_u8(akz_ctx* ctx, const uint8_t* d_imgs, uint32_t w, uint32_t h, uint32_t n,
                                const akz_config* cfg, uint32_t flags, akz_job** out);
int akz_extract_begin_device_f32(akz_ctx* ctx, const float* d_imgs, uint32_t w, uint32_t h, uint32_t n,
                                 const akz_config* cfg, uint32_t flags, akz_job** out);
/* The same for frames in HOST memory (the reference's extract_features starts from host data, lib.rs:171-178): the
   n frames are copied into a staging buffer of the context on its COPY stream, the extraction waits for that copy
   only -- so with batch j+1 begun before batch j is finished, the upload of j+1 runs under the kernels of j.  Pinned
   memory (hipHostMalloc / hipHostRegister) makes the copy asynchronous; pageable memory works and is staged by the
   runtime.  h_imgs must stay valid until the job is finished or abandoned. */
int akz_extract_begin_host_u8(akz_ctx* ctx, const uint8_t* h_imgs, uint32_t w, uint32_t h, uint32_t n,
                              const akz_config* cfg, uint32_t flags, akz_job** out);
int akz_extract_begin_host_f32(akz_ctx* ctx, const float* h_imgs, uint32_t w, uint32_t h, uint32_t n,
                               const akz_config* cfg, uint32_t flags, akz_job** out);
int akz_extract_finish(akz_job* job, akz_result** out);
int akz_job_abandon(akz_job* job);

/* ops::scale_space_extrema::detect_keypoints (scale_space_extrema.rs:199-203) and ops::descriptors::extract_descriptors
   (descriptors.rs:14-27) on evolutions the CALLER holds (the reference's `pub mod ops` takes `&[EvolutionStep]` that
   need not come from extract_features): planes = n_levels x 10 host pointers in akz_plane order (row-major f32 of the
   level's size, NULL where the caller has nothing; Lt, Lx, Ly of every level are required, Ldet unless
   AKZ_NO_DETECT), n_levels = what akz_plan_num_levels returns for (w, h, cfg).  The planes are uploaded, the extrema
   test runs on the uploaded Ldet, and the result carries keypoints (with orientation) and descriptors exactly as if
   the pyramid had been built here.  With AKZ_NO_DETECT the result has no keypoints and serves
   akz_result_describe_keypoints / akz_fetch_plane. */
int akz_extract_from_planes(akz_ctx* ctx, uint32_t w, uint32_t h, const akz_config* cfg, const float* const* planes,
                            uint64_t n_levels, uint32_t flags, akz_result** out);

/* Releases the result; its device blocks (pyramid planes, descriptor rows) go back to the context's pool.  Work of the
   CALLER that reads device pointers obtained from the result (akz_result_device_plane, akz_result_device_descriptors) must
   have completed: the next batch's first stages may run ahead on another stream of the context (AKZ_INPUT_READY,
   akz_extract_begin_host_*) and write a pooled block before anything still queued on the caller's stream has run. */
int akz_result_free(akz_result* res);
int akz_result_num_images(const akz_result* res, uint64_t* n_images);
/* (Vec<EvolutionStep>.len(), Vec<Keypoint>.len(), Descriptor.vector.len()) of image `img` */
int akz_result_counts(const akz_result* res, uint64_t img, uint64_t* n_levels, uint64_t* n_keypoints,
                      uint64_t* desc_bytes);
int akz_result_keypoints(const akz_result* res, uint64_t img, akz_keypoint* out /* n_keypoints */);
/* unpadded: n_keypoints * desc_bytes bytes, Descriptor.vector of each keypoint back to back */
int akz_result_descriptors(const akz_result* res, uint64_t img, uint8_t* out);
/* ops::scale_space_extrema::compute_main_orientation (scale_space_extrema.rs:207-329) and
   ops::descriptors::extract_descriptors (descriptors.rs:14-35) for caller-supplied keypoints of image `img` on the
   retained pyramid: uses point, size, octave and class_id of every keypoint; with compute_orientation != 0 the
   angle field is (re)computed and written back, otherwise the given angle is used.  descriptors: n x desc_bytes.
   Keypoints too close to the image border are sampled with clamped coordinates (the reference would panic). */
int akz_result_describe_keypoints(const akz_result* res, uint64_t img, akz_keypoint* kps, uint64_t n_keypoints,
                                  int compute_orientation, uint8_t* descriptors);
/* device-resident descriptors, 64-byte rows (desc_bytes used, rest zero), for akz_match_device / RCCL gather */
int akz_result_device_descriptors(const akz_result* res, uint64_t img, const uint8_t** d_desc, uint64_t* n_keypoints);
/* D2D copy of ALL images' descriptor rows (image 0 first, 64-byte rows) into a caller buffer of
   capacity_rows rows, e.g. a torch tensor that is then all-gathered over RCCL.  Runs on the context's
   auxiliary stream and is complete on return (it never waits for batches still in flight). */
int akz_result_copy_device_descriptors(const akz_result* res, uint8_t* d_dst, uint64_t capacity_rows,
                                       uint64_t* rows);
/* the contrast factor compute_contrast_factor returned for image `img` (lib.rs:64-69) */
int akz_result_contrast(const akz_result* res, uint64_t img, double* k);
/* scalar fields of EvolutionStep — evolution.rs:59-70, :91 */
int akz_result_level_info(const akz_result* res, uint64_t level, double* etime, double* esigma, uint32_t* octave,
                          uint32_t* sublevel, uint32_t* sigma_size, uint32_t* w, uint32_t* h, uint64_t* n_tau,
                          double* tau, uint64_t tau_cap);
/* lazy D2H of one EvolutionStep image; *n_px = 0 for the 0x0 planes (level 0 Lflow/Lstep).  Planes the
   extraction did not keep (Lxx, Lyy, Lxy, Lstep without AKZ_KEEP_ALL_PLANES) are recomputed for this image
   from the kept ones with the same kernels — bit-identical, at the cost of a few launches per call — so an
   EvolutionStep is complete either way.  out may be NULL to query the size. */
int akz_fetch_plane(const akz_result* res, uint64_t img, uint64_t level, akz_plane plane, float* out,
                    uint64_t* n_px);
/* device address of a resident plane (NULL if not kept) */
int akz_result_device_plane(const akz_result* res, uint64_t img, uint64_t level, akz_plane plane,
                            const float** d_plane);

/* ---- the hot path: match_features (descriptor part) ---------------------------------- */
/* ops::feature_matching::descriptor_match — akaze/src/ops/feature_matching.rs:23-94.
   d0: n0 x desc_bytes, d1: n1 x desc_bytes (host, unpadded).  out must hold n0 entries. */
int akz_descriptor_match(akz_ctx* ctx, const uint8_t* d0, uint64_t n0, const uint8_t* d1, uint64_t n1,
                         uint64_t desc_bytes, uint64_t distance_threshold, double lowes_ratio, akz_match* out,
                         uint64_t* n_out);
/* same on device-resident 64-byte descriptor rows (what akz_result_device_descriptors and the
   RCCL gather produce: an M-LDB descriptor has at most 61 bytes, bytes 61..63 of a row are padding
   and are NOT compared).  d_out: n0 akz_match records on the device, compacted in index_0
   order; *d_n_out (device uint64) receives the count. */
int akz_descriptor_match_device(akz_ctx* ctx, const uint8_t* d_d0, uint64_t n0, const uint8_t* d_d1, uint64_t n1,
                                uint64_t distance_threshold, double lowes_ratio, akz_match* d_out,
                                uint64_t* d_n_out);
/* One query set against n_sets train sets in one launch (all-pairs matching, BASELINE configs[4]): d_train
   holds the sets' 64-byte rows one after the other, set_rows[k] (host) the number of rows of set k.
   Results of set k: matches at d_out + k * n0 (room for n0 each), count at d_n_out[k] (device uint64),
   index_1 relative to the set — exactly what n_sets calls of akz_descriptor_match_device return, at the
   rate of one large product. */
int akz_descriptor_match_sets_device(akz_ctx* ctx, const uint8_t* d_q, uint64_t n0, const uint8_t* d_train,
                                     const uint64_t* set_rows, uint64_t n_sets, uint64_t distance_threshold,
                                     double lowes_ratio, akz_match* d_out, uint64_t* d_n_out);
/* The same launch with the OPPOSITE direction riding along: hamming(a, b) = hamming(b, a), so one pass over the distances of
   a (query set, train set k) block yields both descriptor_match(query set, set k) -- rows as above -- and
   descriptor_match(set k, query set) (feature_matching.rs:23-94 with the arguments exchanged: index_0 a row of set k,
   index_1 a row of the query set, lowest index among equal minima): its matches go to d_out_cols + (number of rows of the
   sets before k) (room for set_rows[k]), its count to d_n_cols[k] (device uint64).  Identical to 2 * n_sets calls of
   akz_descriptor_match_device at the matrix-core work of n_sets.  (On the FP4 matcher, the default; the other matcher
   kernels compute the second direction separately.) */
int akz_descriptor_match_sets_mutual_device(akz_ctx* ctx, const uint8_t* d_q, uint64_t n0, const uint8_t* d_train,
                                            const uint64_t* set_rows, uint64_t n_sets, uint64_t distance_threshold,
                                            double lowes_ratio, akz_match* d_out, uint64_t* d_n_out, akz_match* d_out_cols,
                                            uint64_t* d_n_cols);

/* ---- multi-GPU: the path's one exchange step ----------------------- */
/* extract_features has no cross-image state (akaze/src/lib.rs:167-194): images are sharded one per GPU slot and
   extraction needs no collective.  Before a cross-image brute-force match (feature_matching.rs:23-94 across shards)
   the 64-byte descriptor rows of every rank are all-gathered over RCCL (xGMI).  The reference is a single-process CPU
   crate and has no counterpart.  RCCL is loaded at run time (librccl.so.1); without it these calls return
   AKZ_ERR_UNSUPPORTED and nothing else in the library is affected.
   One process per GPU: rank 0 creates the id and hands its AKZ_COMM_ID_BYTES bytes to every rank by any means (a
   file, MPI, torch.distributed); every rank then calls akz_comm_create (collective). */
#define AKZ_COMM_ID_BYTES 128
typedef struct akz_comm akz_comm;
typedef struct akz_gather akz_gather; /* one exchange in flight */
int akz_comm_unique_id(uint8_t* id_out /* AKZ_COMM_ID_BYTES */);
int akz_comm_create(int device, const uint8_t* id, int rank, int nranks, akz_comm** out);
/* The same communicator with the transport left to the CALLER (MPI, gloo, shared memory; a rehearsal of several ranks on
   ONE GPU, which RCCL refuses): RCCL is not loaded, no id is needed, the call is not collective.  An exchange is then
       akz_gather_begin / akz_gather_begin_rows  -> this rank's block is complete in device memory on return
       akz_gather_blocks(g, &d_send, &d_recv, &bytes)  -> move every rank r's d_send block to d_recv + r * bytes on every
                                                          rank (this rank's own block included)
       akz_gather_deliver(g, stream)                   -> the blocks are complete in the order of `stream` (NULL: now)
   and everything behind it (akz_gather_finish, akz_gather_stream_wait, akz_match_all_pairs, ...) as with RCCL.  The wire
   format is the one described below; akz_gather_descriptors (the synchronous two-collective form) is not available. */
int akz_comm_create_external(int device, int rank, int nranks, akz_comm** out);
/* device = AKZ_COMM_HOST: the same object with its blocks (and the caller's descriptor rows) in HOST memory -- no GPU call is
   made by any akz_comm_* / akz_gather_* function on it.  It serves the wire format, the overflow protocol and the all-pairs
   plan (akz_pairs_plan) to hosts that stage descriptors in host memory and to CPU-only rehearsals of the multi-rank job
   (tests/test_distributed.py under gloo); rows come in through akz_gather_begin_rows / akz_gather_begin_image_rows as host
   pointers, akz_gather_blocks returns host pointers, akz_gather_begin and akz_match_all_pairs (device work) are
   AKZ_ERR_UNSUPPORTED. */
#define AKZ_COMM_HOST (-1)
int akz_comm_destroy(akz_comm* comm);
/* How long akz_gather_finish (and what calls it: akz_gather_descriptors, akz_match_all_pairs) waits for an exchange before
   it gives up with AKZ_ERR_TIMEOUT -- a peer that crashed or never joined leaves a collective waiting for ever, and a host
   that can report that is worth more than one that hangs.  0 (default): wait without limit.  After a timeout the
   communicator is ABANDONED: the stuck collective still sits on its stream, so nothing waits for that stream any more --
   akz_gather_free and akz_comm_destroy return at once and leak the device buffers, streams and the RCCL communicator (freeing
   them would wait for the collective), new exchanges are refused with AKZ_ERR_TIMEOUT.  (An exchange that was only slow: a
   later akz_gather_finish of the same gather -- after a longer akz_comm_set_timeout, or 0 -- that sees it complete puts the
   communicator back in service.)  The host is expected to report the
   error and end the process with a non-zero status WITHOUT running the GPU runtime's exit handlers (`_exit` / `os._exit`:
   they, too, may wait for the stuck stream). */
int akz_comm_set_timeout(akz_comm* comm, double seconds);
/* Optional, once, with idle streams (before the first step): moves the communicator's two streams onto hardware queues and
   command-processor pipes that `ctx`'s busy streams (the caller's, the coarse chain's, the finish half's) do not use -- the
   collective of a step runs beside the next batch's kernels, and two busy streams on one queue or pipe slow both (see
   akz_ctx / INTEGRATION.md, threading).  Triggers the context's own stream-placement probe if it has not run. */
int akz_comm_place_streams(akz_comm* comm, akz_ctx* ctx);
int akz_comm_info(const akz_comm* comm, int* rank, int* nranks);
/* Appendix C form, synchronous: every rank contributes n_local rows (device, 64 bytes each, as returned by
   akz_result_device_descriptors); on return *d_all points at the rows of all ranks in rank order (device memory owned
   by the communicator, valid until the next call) and counts[r] is rank r's row count (host, nranks entries). */
int akz_gather_descriptors(akz_comm* comm, const uint8_t* d_local, uint64_t n_local, const uint8_t** d_all,
                           uint64_t* counts /* nranks */);
/* Pipelined form: ONE fixed-size all-gather per call, enqueued on the communicator's own streams without touching
   the extraction stream.  Every rank contributes a block of 1 + cap_rows rows: a header row {u64 rows, u64 images,
   u64 cap_rows, u64 sequence, u64 overflow, u64 table rows, 0, 0} followed by the descriptor rows of all images of all
   `results` (in order) and, for akz_gather_begin, a table of the rows per image (eight u64 per row; it counts against
   cap_rows: allow one row per eight images);
   cap_rows MUST be the same on every rank (the message sizes of the collective depend on it: a mismatch is fatal
   for the communicator and cannot be detected before the collective is issued) and should be at least the largest
   shard.  A rank whose shard does not fit still takes part — it sends its header alone, marked — and
   akz_gather_finish then returns AKZ_ERR_BUFFER on EVERY rank, with counts[] holding what each rank needed, so that
   all ranks can repeat the exchange with the same larger capacity; the communicator stays usable.
   akz_gather_begin returns as soon as the local rows have been copied — the results may be freed, the next batch
   begun — and never waits for a collective; akz_gather_begin_rows takes raw device rows that are complete in the
   order of producer_stream (may be NULL: complete now) and must stay valid until the gather is finished. */
int akz_gather_begin(akz_comm* comm, const akz_result* const* results, uint64_t n_results, uint64_t cap_rows,
                     akz_gather** out);
int akz_gather_begin_rows(akz_comm* comm, const uint8_t* d_local, uint64_t n_local, uint64_t cap_rows,
                          void* producer_stream, akz_gather** out);
/* external transport only (akz_comm_create_external): the block this rank sends, where the blocks of all ranks go, bytes per block */
int akz_gather_blocks(akz_gather* g, const uint8_t** d_send, uint8_t** d_recv, uint64_t* block_bytes);
int akz_gather_deliver(akz_gather* g, void* stream);
/* make `stream` (e.g. the matcher's) wait for the gathered blocks without a host synchronisation */
int akz_gather_stream_wait(akz_gather* g, void* stream);
/* host wait.  *d_all: nranks blocks of *block_rows (= 1 + cap_rows) rows, rank r's descriptor rows start one row into
   block r; counts / images (host, nranks entries, may be NULL) are read from the headers.  Rows beyond a rank's count
   are unspecified. */
int akz_gather_finish(akz_gather* g, const uint8_t** d_all, uint64_t* block_rows, uint64_t* counts, uint64_t* images);
/* rows of every image of rank `rank`'s shard, in its shard order (host, after akz_gather_finish): a block carries them
   behind its descriptor rows, eight u64 per row (akz_gather_begin_rows sends its rows as ONE image).  *n_images = the
   rank's image count; rows_per_image receives min(cap, *n_images) entries. */
int akz_gather_image_rows(akz_gather* g, int rank, uint64_t* rows_per_image, uint64_t cap, uint64_t* n_images);
/* hand the blocks back to the communicator (waits for the collective if it is still running) */
int akz_gather_free(akz_gather* g);
/* akz_gather_begin_rows for the rows of SEVERAL images held by the caller (device memory, or host memory on an
   AKZ_COMM_HOST communicator): n_images images back to back, rows_per_image[i] rows each; the per-image table travels
   behind the rows exactly as with akz_gather_begin. */
int akz_gather_begin_image_rows(akz_comm* comm, const uint8_t* d_local, const uint64_t* rows_per_image, uint64_t n_images,
                                uint64_t cap_rows, void* producer_stream, akz_gather** out);

/* BASELINE configs[4] — the cross-GPU all-pairs match for hosts that are not Python: after an
   exchange every rank holds the descriptor rows of every image of the job (numbered rank-major).  Every UNORDERED image
   pair {a, b} is matched once, in both directions, by the rank that owns the pair's lead image (lo if hi - lo is odd,
   else hi: every image leads about half of its pairs): the lead image is the query set of one both-direction launch
   (akz_descriptor_match_sets_mutual_device) against the images it leads, so the matrix-core work of a block serves
   descriptor_match(a, b) and descriptor_match(b, a) (ops::feature_matching::descriptor_match, feature_matching.rs:23-94;
   every list identical to akz_descriptor_match of the pair).  The call enqueues on the context's stream and returns; it
   allocates only when the job grows (buffers are pooled in the communicator) and finishes the gather if the caller has
   not; the gather may be freed afterwards (a later exchange that reuses its buffers is ordered behind this call's copies). */
typedef struct akz_pairs akz_pairs;
int akz_match_all_pairs(akz_ctx* ctx, akz_gather* gather, uint64_t distance_threshold, double lowes_ratio, akz_pairs** out);
/* The PLAN of akz_match_all_pairs without the match: finishes the gather and returns an akz_pairs that answers
   akz_pairs_info / _image_rows / _holder / _lead_sets / _totals (lists and distances this rank would compute; no matches) --
   host arithmetic only, also on an AKZ_COMM_HOST communicator.  akz_pairs_matches on it is an error. */
int akz_pairs_plan(akz_gather* gather, akz_pairs** out);
/* the images that `lead_image` (one this rank owns) is matched against by this rank, ascending: with akz_pairs_holder
   these say who computes every unordered pair of the job */
int akz_pairs_lead_sets(const akz_pairs* p, uint64_t lead_image, uint64_t* images, uint64_t cap, uint64_t* n);
/* *n_images: images of the whole job; this rank's are first_owned .. first_owned + n_owned - 1 */
int akz_pairs_info(const akz_pairs* p, uint64_t* n_images, uint64_t* first_owned, uint64_t* n_owned);
int akz_pairs_image_rows(const akz_pairs* p, uint64_t image, uint64_t* rows, int* owner_rank);
/* the rank whose akz_pairs holds both lists of the pair {image_a, image_b} (the owner of its lead image) */
int akz_pairs_holder(const akz_pairs* p, uint64_t image_a, uint64_t image_b, int* rank);
/* descriptor_match(query, image) for a pair this rank holds (akz_pairs_holder; either image may be the query): index_0
   into the query's rows, index_1 into `image`'s; ordered by index_0.  out may be NULL (count only); at most cap records are
   written.  The pair (query, query) is empty.  The first call waits for the launches of akz_match_all_pairs. */
int akz_pairs_matches(const akz_pairs* p, uint64_t query, uint64_t image, akz_match* out, uint64_t cap, uint64_t* n);
/* Everything this rank holds at once (waits for the launches like the first akz_pairs_matches): *n_lists = match lists held
   (two per unordered pair it leads), *n_matches = their records in all, *n_distances = descriptor pairs whose distance was
   formed (rows(a) x rows(b) per unordered pair; each serves both directions).  Any pointer may be NULL. */
int akz_pairs_totals(const akz_pairs* p, uint64_t* n_lists, uint64_t* n_matches, uint64_t* n_distances);
/* Lifetime: an akz_pairs may be freed before or after akz_comm_destroy of the communicator it came from (the communicator
   keeps a list of the objects it has handed out and cuts them loose when it is destroyed; until then a freed object's
   buffers are pooled there, at most two of them).  It must be freed before akz_ctx_destroy of the context it ran on. */
int akz_pairs_free(akz_pairs* p);

/* ---- the rest of match_features (host post-filter) -------------------- */
/* ops::estimate_fundamental_matrix::remove_outliers — estimate_fundamental_matrix.rs:99-165: 8-point
   fundamental matrix + RANSAC over the matches; fewer than 8 matches are returned unchanged.  Host code.
   The reference's own output is not reproducible (HashSet order, `random` crate), so this is a
   behavioural restatement, not a parity target.  out must hold n_matches entries. */
int akz_remove_outliers(const akz_keypoint* keypoints_0, uint64_t n0, const akz_keypoint* keypoints_1, uint64_t n1,
                        const akz_match* matches, uint64_t n_matches, uint64_t num_trials, float epsilon_model,
                        float epsilon_inlier, akz_match* out, uint64_t* n_out);
/* ops::estimate_fundamental_matrix::estimate_fundamental_matrix — estimate_fundamental_matrix.rs:17-69: the 8-point
   model of exactly 8 matches; *found = 0 where the reference returns None (fewer than 8 singular values above
   epsilon); f receives the 3x3 matrix row by row. */
int akz_estimate_fundamental_matrix(const akz_keypoint* keypoints_0, uint64_t n0, const akz_keypoint* keypoints_1,
                                    uint64_t n1, const akz_match* matches8, float epsilon, float* f /* 9 */, int* found);
/* akaze::match_features — akaze/src/lib.rs:252-275: descriptor_match(d0, d1, 10000, lowes_ratio) on the
   GPU, then remove_outliers(kp0, kp1, matches, ransac_trials, 0.05, ransac_epsilon_inliers): samples, winner and final
   filter on the host, the trials' models and inlier counts on the device (the host's arithmetic, bit for bit).
   Descriptors are host arrays of n_descriptors x desc_bytes (both sets the same desc_bytes); keypoints and
   descriptors of a set are counted separately, as the reference's slices are — a set with more descriptors than
   keypoints is AKZ_ERR_INVALID_ARG (the reference panics once such a match reaches RANSAC).  out must hold
   n_descriptors_0 entries. */
int akz_match_features(akz_ctx* ctx, const akz_keypoint* keypoints_0, uint64_t n_keypoints_0, const uint8_t* descriptors_0,
                       uint64_t n_descriptors_0, const akz_keypoint* keypoints_1, uint64_t n_keypoints_1,
                       const uint8_t* descriptors_1, uint64_t n_descriptors_1, uint64_t desc_bytes, double lowes_ratio,
                       uint64_t ransac_trials, float ransac_epsilon_inliers, akz_match* out, uint64_t* n_out);

/* ---- on-disk formats of akaze-util ---------------------------------- */
/* akaze_util::{serialize,deserialize}_{features,matches}_{to,from}_file — akaze-util/src/lib.rs:17-67.
   A path ending in ".json" is serde_json, anything else bincode 1.x (little-endian, u64 lengths), exactly
   as the reference chooses (lib.rs:24-28).  descriptors: n x desc_bytes.  The read functions return the
   counts when the output pointers are NULL. */
int akz_write_features(const char* path, const akz_keypoint* kps, uint64_t n, const uint8_t* descriptors,
                       uint64_t desc_bytes);
int akz_read_features(const char* path, akz_keypoint* kps, uint8_t* descriptors, uint64_t cap_keypoints,
                      uint64_t cap_desc_bytes, uint64_t* n_keypoints, uint64_t* n_descriptors, uint64_t* desc_bytes);
int akz_write_matches(const char* path, const akz_match* matches, uint64_t n);
int akz_read_matches(const char* path, akz_match* out, uint64_t cap, uint64_t* n_out);

/* ---- host-only pieces, callable without a GPU --------------------------------------------- */
/* find_scale_space_extrema's order-dependent cache logic + the sub-pixel step
   (scale_space_extrema.rs:43-178) on NMS candidates {u32 level, u32 flat_idx, f32 v, xp, xm, yp, ym,
   u32 pad} that already passed threshold, 4-neighbour maximum and the border test.  Keypoints are
   returned without orientation (angle = 0). */
int akz_host_select_keypoints(uint32_t w, uint32_t h, const akz_config* cfg, const void* cand, uint64_t n_cand,
                              akz_keypoint* out, uint64_t cap, uint64_t* n_out, uint64_t* n_extrema);

/* ---- measurement hooks --------------------------------------------------------------- */
/* Stage timing with HIP events recorded on the context's stream (device stages) and the host
   clock (host stages).  Off by default; when on, every extract call adds its stage times. */
typedef enum akz_stage {
    AKZ_ST_BLUR0 = 0,    /* level-0 Gaussian blur (lib.rs:56)                            */
    AKZ_ST_CONTRAST = 1, /* compute_contrast_factor (lib.rs:64-69)                       */
    AKZ_ST_PREP = 2,     /* per level: clone/half_size, Lsmooth, Lflow (lib.rs:80-105)   */
    AKZ_ST_FED = 3,      /* per level: the calculate_step launches (lib.rs:109-118)      */
    AKZ_ST_DETECTOR = 4, /* detector_response (detector_response.rs:38-55)               */
    AKZ_ST_NMS = 5,      /* stand-alone NMS launches (if any) + D2H of the candidate lists */
    AKZ_ST_HOST_KP = 6,  /* host: sort, cache logic, refinement                          */
    AKZ_ST_ORIENT = 7,   /* orientation kernel + host atan2f/cosf/sinf                   */
    AKZ_ST_MLDB = 8,     /* descriptor kernel + D2H                                      */
    AKZ_ST_TOTAL = 9,    /* wall time of the whole extract call (host clock)             */
    AKZ_STAGE_COUNT = 10
} akz_stage;
typedef struct akz_profile {
    double ms[AKZ_STAGE_COUNT];
    uint64_t fed_launches;   /* FED kernel launches inside AKZ_ST_FED                        */
    uint64_t fed_px_steps;   /* sum over launches of pixels x steps advanced (x batch)       */
    uint64_t calls;          /* extract calls accumulated                                     */
    uint64_t pixels;         /* input pixels accumulated (w*h*n per call)                     */
    uint64_t det_launches;   /* detector kernel launches inside AKZ_ST_DETECTOR               */
    uint64_t det_px;         /* sum over those launches of level pixels (x batch)             */
    uint64_t fused_px;       /* level pixels (x batch) whose preparation ran inside a FED launch (k_level_march) */
    /* What the context's stream-placement probe found (akz_ctx_warmup / the first large batch; state, not accumulated):
       probed 0/1; early_stages 2 = a batch's level-0 stages run ahead on the copy stream, 0 = on the caller's stream (the
       copy stream could not be given a hardware queue and a pipe of its own: perf only, -5 %); replaced = streams the probe
       re-created; shared = streams that still share a queue or pipe with another busy one; retries = measurements repeated
       because the first answer was "shared" or ambiguous */
    uint32_t placement_probed, placement_early_stages, placement_replaced, placement_shared, placement_retries, placement_reserved;
} akz_profile;
/* on: 0 = off, 1 = every stage (two HIP events per stage and level), 2 = light: only the FED and detector spans and
   the host-clock stages (what bench.py uses inside its timed region) */
int akz_ctx_set_profiling(akz_ctx* ctx, int on);
/* akz_profile has grown at its end (ABI 5) and may again.  akz_ctx_get_profile2 writes min(struct_size, sizeof(akz_profile))
   bytes -- pass sizeof(akz_profile) of the header the caller was compiled against; akz_ctx_get_profile, the original symbol,
   writes the ABI-4 prefix only (everything before placement_probed), so a host built against the smaller struct is never
   written past its end. */
int akz_ctx_get_profile2(akz_ctx* ctx, akz_profile* out, uint64_t struct_size, int reset);
int akz_ctx_get_profile(akz_ctx* ctx, akz_profile* out, int reset);
/* Optional, once, with the caller's stream idle and NOT being captured: runs the stream-placement probe now (~2 ms: spins
   and tiny kernels on the context's streams, which are synchronised) instead of inside the first large akz_extract_begin_*
   -- that call then stays asynchronous.  The probe measures which of the context's streams share a hardware queue or a
   command-processor pipe and replaces those that do; every timing that says "shared" is repeated and the shortest decides.
   Results never depend on it; akz_ctx_get_profile reports what it found. */
int akz_ctx_warmup(akz_ctx* ctx);
/* Extrema candidates per image that the next extraction reserves room for (default 32768; it grows to 1.25x
   the largest count seen).  A list that overflows is detected by akz_extract_finish, which enlarges it and
   repeats the extrema pass on the stored Ldet planes — results are the same, the call is slower; the setter
   exists so that this path can be tested and so that callers with very dense frames can skip the retry. */
int akz_ctx_set_candidate_hint(akz_ctx* ctx, uint32_t per_image);
/* Lanes for small jobs.  A lone 1080p frame is a chain of ~45 launches of a few hundred workgroups each: the chip is busy
   but only a fraction of it at a time.  With lanes = k (2..8) the extract_begin calls of jobs below 2.4 Mpx (the lane gate of DESIGN.md 6.1) are dealt in
   turn to k child contexts with their own streams, scratch planes, candidate buffers and HOST THREAD: the finish half of
   a lane's job (candidate round trip, keypoint selection, orientation / descriptor kernels and their copies) starts on
   the lane's thread as soon as the job has been begun, so that the chains of consecutive frames overlap on the chip and
   their host halves on the host; akz_extract_finish waits for the lane and hands the result over.  Larger jobs, and
   everything at lanes = 1 (default), run on the context itself.  The caller's stream is respected (a lane starts behind
   what the caller enqueued before the call); results are bit-identical and are used through the same calls. */
int akz_ctx_set_lanes(akz_ctx* ctx, uint32_t lanes);
/* The finish half on the context's own thread (default ON).  The finish half of every job begun through
   akz_extract_begin_* -- candidate fetch, keypoint selection, orientation / descriptor kernels, copies -- is started by the
   begin call on a thread the library owns; akz_extract_finish waits for it and hands the result over (results, error
   reporting through akz_extract_finish and akz_job_abandon are the same either way; the synchronous akz_extract_* calls
   run both halves on the caller's thread).  The caller's thread then only enqueues: with two batches begun ahead, whatever
   else it does between the calls (the exchange of a multi-GPU job, file I/O) no longer delays the keypoint half of the
   batches in flight -- worth +13 % on a rank with two host cores, nothing on sixteen.  Entry points that share state with
   the finish half (result queries that launch kernels, the setters) wait until the thread is idle; akz_extract_begin_*, the
   matcher and akz_result_free do not.  on = 0: akz_extract_finish runs the finish half itself (jobs dealt to lanes are
   always finished by their lane's thread). */
int akz_ctx_set_eager_finish(akz_ctx* ctx, int on);
/* Host threads of the finish half of an extraction (candidate bucketing, per-image keypoint selection, libm calls):
   0 (default) = automatic -- the affinity mask of the process, cut down by the cgroup CPU quota and divided by
   LOCAL_WORLD_SIZE (one process per GPU: torchrun sets it), at most 16.  A launcher that has already pinned each rank
   to its own cores passes that number here.  Not while extractions are in flight. */
int akz_ctx_set_host_threads(akz_ctx* ctx, uint32_t threads);
/* ---- image ingest, options files (host code) -------------------- */
/* What `image::open(path)` hands to the crate (akaze/src/lib.rs:171): JPEG (baseline / progressive
   Huffman, 8 bit, 1 or 3 components), PNG (non-interlaced) and binary PNM, decoded to 8-bit luma
   (*channels = 1) or RGB (*channels = 3) as stored.  The pixel buffers returned by the akz_image_*
   loaders are released with akz_image_free.  Decoding of lossy formats is not bit-pinned against the
   `image` / `jpeg-decoder` crates (their sources are not in the reference tree). */
int akz_image_load(const char* path, uint32_t* width, uint32_t* height, uint32_t* channels, uint8_t** pixels);
/* image::open(path).to_luma() — the input of create_unit_float_image (types/image.rs:127-140) */
int akz_image_load_luma(const char* path, uint32_t* width, uint32_t* height, uint8_t** luma);
/* image::open(path).to_rgb() — what the debug drawings are made on (extract_features.rs:94-96) */
int akz_image_load_rgb(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgb);
void akz_image_free(void* pixels);
/* akaze::extract_features(input_image_path, options) — akaze/src/lib.rs:167-194: decode, to_luma,
   unit float image, scale space, keypoints, descriptors.  Equivalent to akz_image_load_luma +
   akz_extract_gray_u8. */
int akz_extract_features_file(akz_ctx* ctx, const char* path, const akz_config* cfg, uint32_t flags, akz_result** out);
/* serde_json form of Config (what `-o options.json` reads and writes, extract_features.rs:66-83).
   to_json: writes a NUL-terminated string, *len = its length (AKZ_ERR_BUFFER if cap is too small, *len
   then holds the size needed).  from_json: fields that are absent keep the value already in *cfg
   (serde itself would reject the file: every field is required there). */
int akz_config_to_json(const akz_config* cfg, char* buf, uint64_t cap, uint64_t* len);
int akz_config_from_json(const char* json, akz_config* cfg);

/* `random::default().seed([s0, s1])`: reseeds the calling thread's default random source — one Xorshift128+
   stream per thread, initially seeded [42, 69], shared by the RANSAC sampling of akz_remove_outliers /
   akz_match_features (estimate_fundamental_matrix.rs:118-121) and the colours of the debug drawings
   (types/image.rs:385-392), exactly as the `random` crate 0.12 behaves inside one process. */
int akz_random_seed(uint64_t s0, uint64_t s1);

/* ---- debug output (host code) ----------------------------------- */
/* 8-bit PNG, channels = 1 (luma) or 3 (RGB) */
int akz_image_save_png(const char* path, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels);
/* types::image::save — normalize to [0,1] (min/max), `(v * 255) as u8`, write (types/image.rs:168-197);
   a 0x0 plane writes nothing */
int akz_image_save_plane_png(const char* path, const float* plane, uint32_t width, uint32_t height);
/* types::evolution::write_evolutions (evolution.rs:175-218): Lt_00000..png ... Ldet_000NN..png (sic: build_path's
   set_extension(".png"), evolution.rs:163-168, keeps the dot of its argument) of image `img`
   into directory `dir` (must exist).  Needs a result extracted with AKZ_KEEP_ALL_PLANES to contain every
   plane; planes that were not kept are skipped like the reference skips 0x0 images. */
int akz_write_evolutions(const akz_result* res, uint64_t img, const char* dir);
/* types::image::random_color (image.rs:385-392): the next colour of the calling thread's default random source */
int akz_random_color(uint8_t* rgb /* 3 */);
/* types::image::draw_circle (image.rs:418-443) / draw_line (image.rs:453-480) on an 8-bit RGB image; pixels outside the
   image are skipped (the reference would panic) */
int akz_draw_circle(uint8_t* rgb, uint32_t width, uint32_t height, float x, float y, const uint8_t* color /* 3 */,
                    float radius);
int akz_draw_line(uint8_t* rgb, uint32_t width, uint32_t height, float x0, float y0, float x1, float y1,
                  const uint8_t* color /* 3 */, float radius);
/* types::keypoint::draw_keypoints_to_image (keypoint.rs:52-56): blends a disc of radius `size` at every
   keypoint into the RGB image, each in the next random_color() of the calling thread's default source
   (image.rs:385-392; see akz_random_seed).  Pixels outside the image are skipped (the reference would panic). */
int akz_draw_keypoints(uint8_t* rgb, uint32_t width, uint32_t height, const akz_keypoint* kps, uint64_t n);
/* types::feature_match::draw_matches (feature_match.rs:32-82): the two images side by side (each half as wide as
   the wider one), one line per match.  *out_rgb is released with akz_image_free. */
int akz_draw_matches(const uint8_t* rgb0, uint32_t w0, uint32_t h0, const uint8_t* rgb1, uint32_t w1, uint32_t h1,
                     const akz_keypoint* kp0, uint64_t n0, const akz_keypoint* kp1, uint64_t n1,
                     const akz_match* matches, uint64_t n_matches, uint32_t* out_w, uint32_t* out_h, uint8_t** out_rgb);

#ifdef __cplusplus
}
#endif
#endif /* AKAZE_HIP_H */
