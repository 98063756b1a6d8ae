/* Measurement and test hooks of libakaze_hip.so: NOT part of the drop-in boundary (include/akaze_hip.h).  A host that
   replaces the reference crate never needs these; the bench, the tools and the tests do.  Same conventions (status-code
   returns, no exceptions across the ABI). */
#ifndef AKAZE_HIP_DEBUG_H
#define AKAZE_HIP_DEBUG_H
#include "akaze_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Is the begin phase of an extraction (scale space + detector + extrema of a batch: ~45 dependent launches for a lone
   1080p frame) shorter as ONE hipGraph launch?  Captures it for (d_imgs, w, h, n, cfg) and times `reps` graph launches
   against `reps` plain enqueues, each from an idle stream (HIP events): *ms_graph, *ms_plain per launch, *graph_nodes
   (may be NULL) the node count.  Results of the probe's extractions are discarded. */
int akz_ctx_graph_probe(akz_ctx* ctx, const uint8_t* d_imgs, uint32_t w, uint32_t h, uint32_t n, const akz_config* cfg,
                        uint32_t flags, uint32_t reps, double* ms_graph, double* ms_plain, uint64_t* graph_nodes);
/* Diagnostics (host only, no GPU work): the row bands the column-march planners cut an n-image batch of w x h into.
   kind 0: detector / blur march with filter half width `half_width` (interior rows half_width .. h-1-half_width),
   kind 1: level march (interior rows 1 .. h-2; first and last band shorter).  Writes up to `cap` (first, end) row pairs
   to rows, the number of bands to *n_bands.  Used by the CPU tests to check that the bands tile the rows exactly. */
int akz_debug_march_bands(int kind, uint32_t w, uint32_t h, uint32_t n, int half_width, int32_t* rows, uint32_t cap,
                          uint32_t* n_bands);
/* names of the default FED kernel and of the detector kernel that large launches take (for bench / profiles) */
const char* akz_fed_kernel_name(void);
const char* akz_detector_kernel_name(void);
/* Test hook: force the chunk counts of the matrix-core matcher (train rows per workgroup column): `pair_chunks` for a
   pair call, `set_chunks` (1..16) per set of a multi-set call; 0 = automatic (the default).  Results are identical for
   every value -- which is what the tests that use this check. */
int akz_debug_set_match_chunks(akz_ctx* ctx, uint32_t pair_chunks, uint32_t set_chunks);
/* Measurement hook: schedule variants of a large batch (results are identical).  key 0: where the early stages
   (level-0 blur, contrast factor) of a batch whose input is complete run: 0 = the copy stream if the context's
   stream-placement probe found it a hardware queue and a pipe of its own (default), 1 = the copy stream regardless, 2 = a
   stream of their own (a fifth busy stream), 3 = the context's stream (no running ahead); key 1: they are held back until
   the batch before has finished its fine-level diffusion (1, default) or start at once (0); key 2: 1 = no placement probe;
   key 3: octave at which the coarse chain forks (0 = default 2); key 4: the job size in thousands of pixels (w*h*n) from
   which a job takes the batch path -- column-march kernels, forked coarse chain, resident tail -- 0 = default (1 400 for the synchronous entry points, 300 for the begin / finish interface since the end of round 6;
   6 000 / 3 000 in round 5; the same value also replaces lane_px, the size below which a context with lanes deals its jobs to them);
   key 5: 1 = the detector of a fine level right behind its level kernel (profiles/r06_interleave.txt: measured, not the default);
   key 6: 1 = every level's preparation as a launch of its own (default 0: on the tiled path the last diffusion launch of a level
   also prepares the next level of the octave -- k_fed_own's epilogue);
   keys 7, 8 (process-wide): the least interior rows of a band of the detector march / the level march, 0 = the planners' rules
   (40; 20 below 48 Mpx per launch) -- how finely a small job's column marches are cut into workgroups;
   key 9: bit 0 = the waited-for job's headers, keypoint records and descriptor rows come to the host as three copies behind the
   descriptor kernel (default: that kernel stores them into the host's pinned buffers itself); bit 1 = the keypoint kernels of
   a job that is alone in the context's hands on the auxiliary stream, as every other job's (default: on the main stream,
   in order behind its detectors, when its chain ended there);
   key 10: 1 = level 0 of a tiled-family job as separate launches -- blur, clearing of the contrast scratch, blur, maximum,
   histogram, percentile, and level 1's preparation as a k_prep launch (default 0: k_head + k_contrast_hist_final, and
   level 1's Lflow from k_head's Scharr pair). */
int akz_debug_set_schedule(akz_ctx* ctx, int key, int value);
/* What the stream-placement probe of the context's first large batch found: info[0] = it has run, info[1] = early stages on
   the context's stream (0) or the copy stream (2), info[2] = streams it re-created because they shared a hardware queue
   or a command-processor pipe with another busy stream of the context, info[3] = streams that still share one. */
int akz_debug_stream_placement(akz_ctx* ctx, int* info);
/* The probe's decision as a pure function of its timings (host only, no GPU): bit 0 = the two streams share a hardware
   queue, bit 1 = a command-processor pipe, bit 2 = too close to a threshold to trust a single measurement (the probe then
   repeats it and lets the shortest decide).  spin_pair_ms: two spins of spin_ms each on two streams, first start to second
   end; tiny_pair_ms: 24 + 24 interleaved tiny kernels on the two streams (negative: not measured); tiny_alone_ms: 24 of
   them on one stream. */
int akz_debug_placement_verdict(float spin_pair_ms, float tiny_pair_ms, float tiny_alone_ms, float spin_ms);
/* The orientation's libm calls ON the device.  The reference's angle is atan2f of two sums, its descriptor pattern is rotated
   by cosf / sinf of that angle -- the platform's libm, i.e. glibc's float routines on the machine the reference runs on.
   csrc/akz_libm.hpp states those routines (glibc 2.35: fdlibm atan2f / atanf, sincosf.h in its FMA and SSE2 builds) as IEEE
   arithmetic; a job whose keypoint selection ran on the device then needs no host round trip between orientation and
   descriptors.  Used only where it is PROVEN equal: on x86-64, with the build glibc's own selector picks for this CPU, after
   a self-test of ~2 million arguments of each function against the process's libm (once per process).
   akz_debug_device_libm: *available = 0 (the host's libm is used), 1 (device, FMA build), 2 (device, SSE2 build); *last_job =
   the same for the last finished job (0 also when that job's selection ran on the host).  akz_debug_set_device_libm(ctx, 0)
   switches it off, (ctx, -1) back to automatic.  akz_debug_libm_eval: d_out3[3 i .. 3 i + 2] = atan2f(d_a[i], d_b[i]),
   cosf(d_a[i]), sinf(d_a[i]) as the device forms them (fma != 0: the FMA build) -- NaN for |d_a[i]| >= 120. */
int akz_debug_device_libm(akz_ctx* ctx, int* available, int* last_job);
int akz_debug_set_device_libm(akz_ctx* ctx, int mode);
int akz_debug_libm_eval(akz_ctx* ctx, const float* d_a, const float* d_b, float* d_out3, uint64_t n, int fma);
/* The gate table (csrc/akz_gates.hpp): every size / host-thread threshold that chooses between two equivalent kernel families
   or paths -- name, value, what it counts, what lies on either side.  Results never depend on a gate.  *rows points at a
   static table of *n entries.  (big_px_sync / big_px_async are the compiled-in values; akz_ctx_calibrate_gates and
   akz_debug_set_schedule(ctx, 4, ...) change a context's own copies.) */
typedef struct akz_gate {
    const char* name;
    double value;
    const char* unit;
    const char* meaning;
} akz_gate;
int akz_debug_gates(const akz_gate** rows, uint64_t* n);
/* Re-derives the two JOB gates of this context from timings on the machine at hand instead of the compiled-in values (which
   were tuned on a 16-core host of one pool): lone synthetic frames of 0.9 / 1.4 / 2.1 / 4.1 / 8.3 Mpx are extracted through
   the synchronous call and through the begin / finish interface (two jobs in flight), each with the batch path forced on
   and off; a gate becomes the smallest size from which the batch path is the faster one at that size and at every larger
   one (the largest size + 1 if it never is).  ~0.2 s, allocates what such jobs allocate; the caller's stream must be idle.
   *sync_px / *async_px (may be NULL): the gates chosen; ms (may be NULL): 5 sizes x {sync lone, sync batch, streamed lone,
   streamed batch} = 20 medians in milliseconds.  Results of extractions never depend on the outcome. */
int akz_ctx_calibrate_gates(akz_ctx* ctx, uint64_t* sync_px, uint64_t* async_px, double* ms);
/* Measurement hook: the HIP-event spans of the diffusion and detector stages (akz_ctx_set_profiling) broken down by kernel
   variant and launch shape -- the rows behind bench.py's roofline.kernel, which names a group of kernels.  kind: see below;
   param: fused FED steps of a k_level_march launch (+16: the octave's 2x2 mean folded in, +32: Lstep written), FED steps
   covered by a group of k_fed_own launches, levels inside a k_octave_resident launch, sigma_size of a detector launch;
   (w, h, n): the launch's level size and batch (the largest level of a k_detector_tiled launch over several levels);
   launches, px = level pixels x batch, px_steps = pixel-steps advanced, ms = sum of the spans.  Rows accumulate like
   akz_profile; reset != 0 zeroes the figures.  *n_rows = rows there are (may exceed cap). */
enum { AKZ_KR_LEVEL_MARCH = 1, AKZ_KR_FED_OWN = 2, AKZ_KR_OCTAVE_RESIDENT = 3, AKZ_KR_DETECTOR_TILED = 4, AKZ_KR_DETECTOR_MARCH = 5 };
typedef struct akz_kernel_row {
    uint32_t stage, kind, param, w, h, n;
    uint64_t launches, px, px_steps;
    double ms;
} akz_kernel_row;
int akz_debug_kernel_rows(akz_ctx* ctx, akz_kernel_row* out, uint32_t cap, uint32_t* n_rows, int reset);
/* Test hook: pm_g2's reciprocal on its own -- d_out[i] = (1.0 / d_x[i]) as f32 the way every level kernel forms it
   (csrc/akz_pm_g2.hpp: refined hardware reciprocal, the full f64 division only where the f32 rounding could depend on it). */
int akz_debug_rcp_f64_to_f32(akz_ctx* ctx, const double* d_x, float* d_out, uint64_t n);
/* Test hook: where the extrema candidates are put into scan order: 1 = bucketed and sorted on the HOST (also the fallback
   that a candidate-list overflow and over-wide sort keys take), 0 = device sort, -1 = automatic (the default: device
   sort for contexts with fewer than four host threads).  Results are identical. */
int akz_debug_set_host_sort(akz_ctx* ctx, int on);
/* Test hook: where and how the order-dependent keypoint selection runs: 2 = on the device (dependency rounds over the
   neighbour lists, k_select; a job with an image whose lists overflowed falls back to 1), 1 = on the host from the device's
   neighbour lists (k_relations; needs the device sort, which it switches on), 0 = on the host from its spatial grids,
   -1 = automatic (the default: on the device for contexts with fewer than four host threads, for jobs that do not take the
   batch path, for jobs of fewer than four images, for the job that is waited for and for images of 6 Mpx and more; the grids otherwise).  Results are identical. */
int akz_debug_set_select(akz_ctx* ctx, int mode);
/* The selection of the last finished job: info[0] = where it ran (0 / 1 / 2 as above), info[1] = the most dependency rounds
   an image took on the device, info[2] = images that sent the job back to the host's selection, info[3] = candidates,
   info[4 .. 7] = 10 ns ticks of k_select's phases on image 0 (first states, turns, second pass, output).  info: 8 ints. */
int akz_debug_select_info(akz_ctx* ctx, int* info);

/* ---- kernel-family selectors and the synthetic frame generator (tests, bench, tools): every mode gives bit-identical
   results; a drop-in host never calls these ---------------------------------------------------------------------- */
/* Deterministic synthetic 8-bit luma frame (integer-only, SplitMix64-seeded):
   gradient background + w*h/2048 random rectangles/discs + +-8 noise.  (shift_x, shift_y)
   translates the shapes, giving a second view of the same frame for match tests.  Host code. */
int akz_synth_frame_u8(uint8_t* out, uint32_t w, uint32_t h, uint64_t frame_index, int32_t shift_x,
                       int32_t shift_y);
/* FED kernel variant: 2 (default) = k_fed_own, LDS tile + register ownership, up to 8 explicit steps per launch (16
   for launches of a few workgroups); 0 = k_fed_step, one launch per step.  Results are bit-identical. */
int akz_ctx_set_fed_mode(akz_ctx* ctx, int mode);
/* Matcher kernel: 2 (default) and 3 = matrix cores on FP4 operands (k_match_fp4: descriptor bits as +-1 in e2m1,
   v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales, hamming = (488 - dot) / 2, exact in f32), 1 = matrix cores
   on int8 operands (k_match_mfma), 0 = popcount kernel (k_match).  Results are identical. */
int akz_ctx_set_match_mode(akz_ctx* ctx, int mode);
/* Detector kernel variant: 2 (default) = automatic (the one-pass column march k_detector_march for launches of
   8 Mpx and more, the one-kernel LDS-tiled form k_detector_tiled below that); 5 = column march wherever it is
   supported (sigma_size <= 4); 4 = the LDS-tiled kernel; 0 = the LDS-tiled kernel pair (the fallback for other
   kernel sizes).  Results are bit-identical. */
int akz_ctx_set_detector_mode(akz_ctx* ctx, int mode);
/* Level preparation (Lsmooth, Lflow of a level): 2 (default) = automatic — for launches of 8 Mpx and more the
   preparation and the level's first (up to four) diffusion steps run in ONE kernel (k_level_march), smaller launches take
   the streaming or the LDS-tiled preparation kernel; 3 = the fused kernel wherever it is supported; 1 = streaming
   preparation, 0 = LDS-tiled preparation (both without fusion).  Results are bit-identical. */
int akz_ctx_set_prep_mode(akz_ctx* ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif /* AKAZE_HIP_DEBUG_H */
