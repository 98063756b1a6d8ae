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
/* Test hook: where the extrema candidates are put into scan order: 1 = bucketed and sorted on the HOST (also the fallback
   that a candidate-list overflow and over-wide sort keys take), 0 = device sort, -1 = automatic (the default: device
   sort for contexts with fewer than four host threads).  Results are identical. */
int akz_debug_set_host_sort(akz_ctx* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* AKAZE_HIP_DEBUG_H */
