// Every size / host-thread gate that chooses between kernel families or paths of the C ABI, in ONE table.
//
// Results never depend on a gate (each side of each gate is held to the oracle bit for bit; tests/test_gpu_gates.py forces
// every path on every BASELINE geometry through both entry points): a gate only says which of two equivalent forms is
// expected to be faster.  The values were measured on MI355X boxes of this pool (profiles/r04_*, r05_job_gate*.txt,
// r05_lone_select.txt); akz_debug_gates() hands the table out (DESIGN.md section 6 is rendered from it by
// tools/render_gates.py), akz_debug_set_schedule(ctx, 4, ...) overrides the job gates for measurements, and
// akz_ctx_calibrate_gates() re-derives the job gates from timings on the machine at hand.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/akaze_hip_debug.h"

namespace akz {
namespace gates {

// ---- which path a JOB takes (w * h * n input pixels) ----
// The batch path -- column-march kernels on launches that are large enough, forked coarse chain, resident tail -- against the
// chain of tiled launches on one stream.  Rounds 4-5 had the gate at 6 Mpx for a synchronous call and 3 Mpx for a job of the
// begin / finish interface; round 6 found why a call liked the batch path so much later than a stream: its keypoint
// selection ran on the host's grids (one thread per image), which a stream hides and a call does not.  With the selection of
// the job that is waited for on the device, the batch path won from ~2.4 Mpx on through both entry points (at 1 x 1080p and
// 2 x 720p the two were equal): profiles/r06_job_gates.txt.  Later in round 6 the diffusion kernel stopped paying for the
// image's border (akz_kernels.hip: k_fed_own) and the fork of the coarse chain began to pay for a lone 1080p frame as well:
// 0.633 -> 0.617 ms per call, 0.470 -> 0.443 per streamed frame; 1760 x 990: 0.628 -> 0.597 / 0.455 -> 0.426; at 1600 x 900 the two
// are equal with all planes kept and the batch path 3-5 % ahead without, at 2 x 960 x 540 and 1280 x 720 the one-stream chain is
// 5-8 % ahead (profiles/r06_lone_chain.txt sections 12, 22): 1.4 Mpx.  Two constants remain because akz_ctx_calibrate_gates
// measures both.
constexpr uint64_t kBigPxSync = 1400000;    // akz_extract_*
constexpr uint64_t kBigPxAsync = 300000;    // akz_extract_begin_*: a STREAM of jobs is ahead on the batch path from VGA frames on, with all planes
                                            // kept or without (640 x 480: 0.324 -> 0.305 ms per frame, 1280 x 720: 0.365 -> 0.332, 1600 x 900:
                                            // 0.41 -> 0.37; 480 x 360: 0.216 against 0.299 the other way) -- since the detector march of a job of few
                                            // strip columns is cut into 24-row bands (akz_march.hip); r06_lone_chain.txt sections 22-24
constexpr uint64_t kLanePx = 2400000;       // akz_ctx_set_lanes deals the jobs below this to its lanes (where they run as one-stream chains:
                                            // four lanes carry 1080p frames at 0.34-0.37 ms each, the batch path on one context at 0.44)
constexpr uint64_t kTiledPrepPx = 11000000; // jobs below this: the tiled preparation family for every launch (k_head, k_contrast_hist_final,
                                            // k_prep riding on k_fed_own's last launch, no resident tail) whatever the per-launch gates
                                            // below say.  8 Mpx when the family was introduced (r06_lone_libm.txt); after the rest of
                                            // round 6 (level 0 in two launches, k_fed_own's borders) it is ahead up to 10.4 Mpx --
                                            // 4 x 1080p 1.51 -> 1.36 ms per call, 5 x 1080p 1.72 -> 1.58, a lone 4K frame 1.38 -> 1.37 --,
                                            // mixed at 12-17 Mpx and behind from 25 (r06_lone_chain.txt section 19)
// ---- which kernel family a LAUNCH takes (level w * h * n pixels) ----
constexpr uint64_t kMarchPx = 8u << 20;     // blur / contrast / detector marches instead of the tiled kernels (also every
                                            // full-resolution launch of a batch-path job, whatever its size)
constexpr uint64_t kLevelMarchPx = 4000000; // k_level_march (preparation + <= 4 FED steps in one launch) instead of k_prep + k_fed_own
constexpr uint64_t kStreamPx = 2u << 20;    // the streaming preparation / blur kernels instead of the tiled ones
constexpr uint64_t kFedDeepWorkgroups = 512;  // launches of at most this many workgroups fuse 16 FED steps instead of 8
// ---- where the order-dependent keypoint selection and the candidate sort run ----
constexpr unsigned kFewHostThreads = 4;     // contexts with fewer host threads keep selection and sort on the device
constexpr uint64_t kSelectDevicePx = 6000000;  // images of this size and more: neighbour lists + device selection when waited for
constexpr uint32_t kSortBuckets = 65536;    // (image, level, row) buckets the four-launch counting sort takes; rocPRIM beyond
// ---- the matcher ----
constexpr uint64_t kMergeCompactMinRows = 2048;     // query sets from here on: merge + ratio test + compaction in one launch
constexpr uint64_t kMergeCompactMaxRows = 262144;   // ... up to here (every workgroup of its look-back chain resident at once)
constexpr uint32_t kPremergeChunks = 4;             // more train chunks than this: a parallel merge before the compaction

// the table akz_debug_gates() returns: name, value, unit, what lies on either side
inline const akz_gate* table(size_t* n) {
    static const akz_gate rows[] = {
        {"big_px_sync", (double)kBigPxSync, "input px per job", "synchronous akz_extract_*: the chain of tiled launches on one stream below, the batch path (forked coarse chain; marches and resident tail where tiled_prep_px and the per-launch gates allow) from here on"},
        {"big_px_async", (double)kBigPxAsync, "input px per job", "akz_extract_begin_*: the same choice for a job of the begin / finish interface (a stream of jobs overlaps the two chains of consecutive jobs: the batch path pays from much smaller jobs on than for a call that is waited for)"},
        {"lane_px", (double)kLanePx, "input px per job", "a context with lanes (akz_ctx_set_lanes) deals the jobs below this to them, as one-stream chains; larger jobs stay on the context"},
        {"tiled_prep_px", (double)kTiledPrepPx, "input px per job", "below: the tiled preparation family for every launch of the job (level-0 blur, contrast passes, k_prep as an epilogue of the previous level's last k_fed_own launch, no resident tail); from here on the per-launch gates below decide"},
        {"march_px", (double)kMarchPx, "level px per launch", "k_blur5_march / k_contrast_march / k_detector_march instead of the tiled kernels (batch-path jobs: their full-resolution launches regardless)"},
        {"level_march_px", (double)kLevelMarchPx, "level px per launch", "k_level_march instead of k_prep + k_fed_own"},
        {"stream_px", (double)kStreamPx, "level px per launch", "k_prep_stream / k_blur5_stream instead of the tiled k_prep / k_blur"},
        {"fed_deep_workgroups", (double)kFedDeepWorkgroups, "workgroups per launch", "k_fed_own fuses 16 steps per launch at or below, 8 above"},
        {"few_host_threads", (double)kFewHostThreads, "host threads of the context", "fewer: candidate sort and keypoint selection on the device; from here on the host's counting sort and grids serve large batches"},
        {"select_device_px", (double)kSelectDevicePx, "px per image", "images from here on: device neighbour lists (+ k_select when the job is waited for) even on many-core hosts"},
        {"sort_buckets", (double)kSortBuckets, "(image, level, row) buckets", "k_sort_rows / k_bucket_* up to here, rocPRIM radix sort beyond"},
        {"merge_compact_min_rows", (double)kMergeCompactMinRows, "query rows", "single-workgroup compaction below, k_match_merge_compact from here on"},
        {"merge_compact_max_rows", (double)kMergeCompactMaxRows, "query rows", "k_match_merge_compact up to here (its look-back needs the whole grid resident), merge + compaction beyond"},
        {"premerge_chunks", (double)kPremergeChunks, "train chunks", "more chunks than this: k_match_merge before the compaction"},
    };
    if (n) *n = sizeof(rows) / sizeof(rows[0]);
    return rows;
}

}  // namespace gates
}  // namespace akz
