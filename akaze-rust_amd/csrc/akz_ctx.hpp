// The context of libakaze_hip.so and what every translation unit of the C ABI shares: akz_ctx (streams, scratch, pools, profile),
// the stage timer, buffer helpers and the entry-point prologue (bind).  The C ABI itself is split by boundary area:
//   akz_api.cpp      context lifecycle, host-side planning, profile, modes and measurement hooks
//   akz_ops.cpp      the per-op entry points (`pub mod ops` / `types::image`) and the launch helpers the pipeline shares with them
//   akz_extract.cpp  extract_features: begin / finish halves, jobs, results, stream placement
//   akz_match_api.cpp  descriptor_match in all its forms, match_features
// Every size / host-thread gate that picks between kernel families or paths is a named constant of akz_gates.hpp.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <sched.h>
#include <time.h>
#include <type_traits>

#include "akz_internal.hpp"
#include "akz_select.hpp"
#include "akz_pool.hpp"
#include "akz_gates.hpp"

// functions shared by the library's translation units, none of them exported
#define AKZ_LOCAL __attribute__((visibility("hidden")))

namespace akz {
const std::string& get_error();
}
using namespace akz;

namespace {
struct SelKpHost {  // a keypoint of the device's selection as the host fetches it (akz_sort.hip: SelKp)
    sel::KpRec rec;
    OrientOut sums;
};
static_assert(sizeof(SelKpHost) == 32, "the device writes 32-byte records");
}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};


// Pixels per launch (w * h * batch) from which a launch fills the chip on its own: the column-march kernels, the fork of
// the coarse chain and the resident coarse octave engage from here on (swept again in round 3: 4 Mpx is -7 ... -9 % on
// 8-frame batches, 8 000 000 -- so that a lone 4K frame qualifies -- changes nothing for it: that frame is bound by the
// host's per-image selection and the finish round trips, not by its kernels).
// From which job size (w*h*n pixels) a job takes the BATCH path -- column-march kernels, the coarse chain forked onto its own
// stream, the resident tail -- instead of the chain of tiled launches (profiles/r05_lone_ab.txt, one MI355X):
//   job            latency of one call   per job in a stream of jobs
//   1 x 1080p      0.98 / 1.21 ms        0.60 / 0.62 ms      (tiled chain / batch path)
//   1 x 2016x1512  1.21 / 1.28           0.77 / 0.67
//   2 x 1080p      1.42 / 1.51           0.90 / 0.78
//   3 x 1080p      1.76 / 1.78           1.09 / 0.92
//   1 x 4K         2.65 / 2.40           1.50 / 1.37
// Which jobs take the batch path (column marches, forked coarse chain, resident tail) is decided by job size per entry point:
// (the two values: akz_gates.hpp, gates::kBigPxSync / kBigPxAsync)
// The finish half of a lane's jobs on a thread of the library (akz_ctx_set_eager_finish): started by begin, so that the
// candidate round trip, the host keypoint logic and the keypoint kernels of frame i run while the caller's thread
// enqueues frame i + 1 on another lane; akz_extract_finish then only collects the result.  One thread per lane, jobs
// in the order they were begun.  Every other use of the lane (bind) first waits until the thread is idle, so the lane's
// state is never touched from two threads at once.
struct akz_job;
struct Finisher {
    std::thread th;
    std::mutex m;
    std::condition_variable wake, done;
    std::vector<akz_job*> queue;
    unsigned in_flight = 0;  // queued + running
    bool quit = false;
};

struct akz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;            // the stream the enqueueing helpers use: the caller's stream, except while extract_begin
                                             // puts a batch's early stages or its coarse chain on a stream of their own
    hipStream_t main = nullptr;              // the caller's stream (never changes; what other threads synchronise with)
    bool own_stream = false;
    DevBuf scratch[6];                       // f32 plane temporaries (largest level x batch)
    DevBuf scratch_coarse;                   // the coarse chain's own diffusion scratch: it outlives the batch's join (see extract_begin)
    DevBuf lazy[6];                          // one-image planes of akz_fetch_plane's recomputation (never shared with scratch users)
    DevBuf small;                            // hmax bits / histogram / counters
    void* small_zero_p = nullptr;            // ... of which the first small_zero bytes are known to be zero (head_impl: its histogram
    size_t small_zero = 0;                   //     pass leaves them so; 0: not known)
    DevBuf cand;                             // NMS candidates
    DevBuf cand_sorted, sort_scratch;        // the list in scan order (device sort of extract_finish) and the sort's scratch
    DevBuf rel_scratch;                      // the selection's neighbour lists (launch::candidate_relations)
    DevBuf sel_scratch, sel_recs;            // the selection on the device (launch::select_device): its scratch, the selected keypoints
    DevBuf bucket_scratch;                   // launch::sort_candidates_buckets (its counters are zero between jobs)
    int dbg_select = -1;                     // akz_debug_set_select: 2 / 1 / 0 force the device / the neighbour-list / the grid selection, -1 automatic
    std::atomic<uint32_t> last_total_kp{0};  // keypoints of the previous finished job (speculative fetch size of the device selection)
    std::atomic<int> sel_last_mode{-1};      // akz_debug_select_info: how the last finished job was selected (0 grids, 1 lists, 2 device), the
    std::atomic<int> sel_skip{0};            // jobs of shape sel_skip_shape that leave the device's selection out (the last one fell back)
    std::atomic<uint64_t> sel_skip_shape{0};
    std::atomic<uint32_t> sel_last_ticks[4];
    std::atomic<uint32_t> sel_last_rounds{0}, sel_last_fallback{0};  // device's longest run of rounds, images that sent it back to the host
    DevBuf kp_in, kp_out;                    // keypoint params / orientation sums
    DevBuf match_a, match_b, match_rec, match_out;
    DevBuf match_state;                      // k_match_merge_compact's per-workgroup counts (zeroed when allocated, then told apart by epoch);
                                             // single-stream: only ever touched by launches on `stream` (see match_device_impl)
    uint32_t match_epoch = 0;
    DevBuf ransac_dev, ransac_pin;           // match_features: the trials' inputs and outputs on the device / pinned staging of both
    DevBuf mm_q8, mm_t8, mm_pop, mm_tab;     // MFMA matcher: unpacked int8 images of the two sets, bit counts, set tables
    DevBuf mm_cols;                          // both-direction launches: the train rows' (best, second) state, seed records and bound
    void* tab_ring = nullptr;                // pinned staging ring of the multi-set matcher's tables
    size_t tab_ring_bytes = 0;
    uint64_t tab_ring_next = 0;
    hipEvent_t tab_ring_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // A second set of the multi-set matcher's scratch: akz_match_all_pairs runs the launches of consecutive lead images
    // alternately on the caller's stream and on the finish stream (akz::match_sets_at, side 1), so that the small launches
    // around one image's pass (unpack, seed, compactions: 130 us of 550) run under the other's
    struct MatchSide {
        DevBuf q8, t8, pop, tab, cols, rec;
        void* ring = nullptr;
        size_t ring_bytes = 0;
        uint64_t ring_next = 0;
        hipEvent_t ring_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    } ms1;
    int match_mode = 2;                      // 0: popcount kernel, 1: matrix cores on int8 operands, 2 (default) / 3: on FP4 operands (akz_ctx_set_match_mode)
    uint32_t dbg_pair_chunks = 0, dbg_set_chunks = 0;  // akz_debug_set_match_chunks (0: automatic)
    int dbg_device_libm = -1;   // akz_debug_set_device_libm: 0 = the host's libm always, -1 = automatic (device_libm_mode)
    int libm_last = 0;          // how the last finished job got its angles: 0 host libm, 1 / 2 the device's FMA / SSE2 forms
    int dbg_host_sort = -1;                            // akz_debug_set_host_sort: 1 / 0 force the host / the device sort, -1 automatic
    DevBuf cosi;                             // (cos, sin) per keypoint
    DevBuf pin[10];                          // pinned host staging: candidates, orientation sums, descriptor
                                             // rows, keypoint params, (cos, sin), contrast factors, neighbour lists, their flags,
                                             // the device selection's headers, its orientation sums
    std::vector<std::pair<size_t, void*>> slab_pool;  // freed device blocks (pyramid slabs, descriptor rows)
    std::mutex slab_m;                                // results are freed by the caller while a lane's finisher thread allocates
    // extractions in flight (akz_extract_begin_* / akz_extract_finish)
    static constexpr int kSlots = 3;
    DevBuf cand_slot[kSlots], count_slot[kSlots];
    std::atomic<bool> slot_busy[kSlots] = {{false}, {false}, {false}};  // (begin on the caller's thread, finish possibly on the finisher's)
    std::atomic<uint32_t> cand_cap_hint{1u << 15};  // grows to 1.25x the largest candidate count seen
    std::atomic<int> live_results{0};   // akz_result objects (also inside jobs) that still point at this context
    bool dead = false;                  // akz_ctx_destroy was called; the struct lives until the last result is freed
    std::atomic<uint32_t> last_total_cands{0};  // candidates of the previous finished job (speculative fetch size)
    std::atomic<uint64_t> last_cand_shape{0};   // ... and its shape (w << 40 | h << 16 | n)
    hipStream_t aux = nullptr;          // finish-side copies and keypoint kernels (created under aux_m: the caller's thread and the finisher's may both be first)
    std::mutex aux_m;
    hipStream_t coarse = nullptr;       // the coarse octaves' chain (diffusion + detectors), next to the fine detectors
    hipStream_t pre = nullptr;          // level-0 blur + contrast factor of a batch whose input is known to be complete:
    hipEvent_t pre_done = nullptr;      // they run ahead, under the kernels of the batch before (extract_begin)
    hipStream_t copy = nullptr;         // uploads of host frames (akz_extract_begin_host_*), under the kernels of the batch before
    DevBuf stage[kSlots];               // staging buffers of those uploads, one per job slot
    hipEvent_t staged[kSlots] = {nullptr, nullptr, nullptr};
    // Lanes: child contexts (own streams, scratch planes, candidate slots) that small jobs are dealt to in turn, so that
    // the launch chains of consecutive single frames -- 44 back-to-back launches of a few hundred workgroups each --
    // overlap on the chip instead of queueing on one stream (akz_ctx_set_lanes).
    std::vector<akz_ctx*> lanes;
    unsigned next_lane = 0;
    bool is_lane = false;               // this context is a lane of another one
    bool eager_finish = true;           // akz_ctx_set_eager_finish (default on): the finish half of every job begun through
                                        // akz_extract_begin_* runs on the context's own thread (a lane's on the lane's)
    std::shared_ptr<Finisher> fin;      // (a lane's) finisher thread (shared with its jobs: one may outlive the context), started with its first eager job
    hipEvent_t lane_in = nullptr;       // inputs of the job are ready on the caller's stream
    // stage profiling (akz_ctx_set_profiling)
    uint64_t stream_min_px = gates::kStreamPx;  // pixels per launch (w*h*n) from which the streaming kernels pay off
    int prep_mode = 2;  // level preparation: 0 LDS-tiled, 1 streaming, 2 auto (fused with the first diffusion steps for large launches), 3 fused wherever supported
    int det_mode = 2;  // 0: tiled pair, 2: auto, 4: one tiled kernel, 5: column march (akz_ctx_set_detector_mode)
    int fed_mode = 2;  // 0: k_fed_step (1 step/launch), 2: k_fed_own (<= 8 steps/launch; <= 16 for small launches)
    // schedule variants (akz_debug_set_schedule): [0] where the early stages of a batch run -- 0 the copy stream if the
    // placement probe found it a queue and a pipe of its own (below), 1 the copy stream regardless, 2 a stream of their own
    // (a fifth busy stream), 3 the context's stream (no running ahead); [1] early stages held back until the batch before
    // has finished its fine-level diffusion (default) or not; [2] no placement probe
    int sched[11] = {0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t big_px = gates::kBigPxAsync;  // the gate of the job being begun (set by extract_begin from the two below; the begin half's helpers read it)
    uint64_t big_px_sync = gates::kBigPxSync, big_px_async = gates::kBigPxAsync;  // (sched[4] sets both: measurement)
    uint64_t lane_px = gates::kLanePx;  // jobs below it go to the lanes, if the context has any (sched[4] sets it too)
    // pixels per LAUNCH (level w*h*n) from which the blur, the contrast passes and the detectors take their column-march
    // form: 8 Mpx (a 32 x 480x270 level: 41 us tiled against 66 for the march, which would run one strip per image row);
    // inside a batch-path job smaller than that, the job's own size -- its full-resolution launches march, the rest is tiled
    static constexpr uint64_t kLaunchMarchPx = gates::kMarchPx;
    uint64_t launch_min_px = kLaunchMarchPx;
    // Stream placement (place_streams): the runtime multiplexes a process's streams onto a few in-order hardware queues
    // (GPU_MAX_HW_QUEUES, 4 by default); two busy streams of a context on one queue serialise the whole pipeline, so the
    // first large batch measures which of the context's streams actually run side by side and replaces those that do not
    bool placed = false;
    int pre_mode = 0;            // early stages: 0 on the context's stream, 2 on the copy stream
    int place_replaced = 0;      // streams that were re-created because they shared a queue with another one
    int place_collisions = 0;    // pairs that still share a queue (no free queue was found)
    int place_retries = 0;       // probe measurements that were repeated because the first answer was "shared" or ambiguous
    int lane_collisions = 0;     // lanes whose stream still shares a queue or a pipe with another lane's
    hipEvent_t probe_ev[2] = {nullptr, nullptr};
    int profiling = 0;  // 0 off, 1 FED spans + host-clock stages, 2 every stage
    akz_profile prof{};
    struct Span { int stage; hipEvent_t a, b; int row; };
    std::vector<Span> spans;          // recorded, not yet resolved
    // akz_debug_kernel_rows: the spans of the two dominant stages by kernel variant and launch shape (what bench.py lists
    // behind roofline.kernel); a span's `row` indexes this table (-1: not broken down)
    std::vector<akz_kernel_row> rows;
    std::vector<hipEvent_t> ev_pool;  // recycled events
    std::mutex ev_m;                  // guards spans and ev_pool (begin records on the caller's thread while a finish resolves)
    // Every extract_begin records, behind its last fine-level diffusion launch, the event fed_ev[seq % kFedRing] of its
    // sequence number: the keypoint kernels of job i are held back until job i + 1 has passed that point, and the early
    // stages of job i + 1 until job i has (see extract_begin).  Four slots: at most kSlots jobs are in flight.
    static constexpr int kFedRing = 4;
    hipEvent_t fed_ev[kFedRing] = {nullptr, nullptr, nullptr, nullptr};
    // ... and pre_ev[seq % kFedRing] behind the last diffusion launch of its FIRST octave: from there to the detectors the
    // main stream carries the half-resolution octave's launches, which are bound by latency and leave most of the chip's
    // bandwidth unused -- that is where the next job's blur and contrast passes go (sched[1] = 2)
    hipEvent_t pre_ev[kFedRing] = {nullptr, nullptr, nullptr, nullptr};
    std::atomic<uint64_t> begin_seq{0};  // sequence number of the job begun last
    std::shared_ptr<std::atomic<int>> in_hand = std::make_shared<std::atomic<int>>(0);  // akz_job::in_hand
    std::unique_ptr<WorkerPool> workers;  // host threads of the finish half (started on first use)
    unsigned host_threads = 0;            // akz_ctx_set_host_threads; 0 = sized by host_cpu_share()
    WorkerPool& pool() {
        if (!workers) workers.reset(new WorkerPool(std::min(host_threads ? host_threads : host_cpu_share(), 16u) - 1));
        return *workers;
    }
};

// RAII stage timer: device stages bracket the enqueued work with two events on the stream; they
// are resolved (hipEventElapsedTime) at the end of the extract call, after the final sync.
struct StageTimer {
    akz_ctx* c;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(akz_ctx* c) {
        {
            std::lock_guard<std::mutex> lk(c->ev_m);
            if (!c->ev_pool.empty()) {
                hipEvent_t e = c->ev_pool.back();
                c->ev_pool.pop_back();
                return e;
            }
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    bool on;
    hipStream_t s;
    int row = -1;
    StageTimer(akz_ctx* ctx, int st, hipStream_t stream = nullptr) : c(ctx), stage(st), s(stream ? stream : ctx->stream) {
        on = c->profiling >= 2 || (c->profiling == 1 && (st == AKZ_ST_FED || st == AKZ_ST_DETECTOR));
        if (!on) return;
        a = get(c);
        b = get(c);
        (void)hipEventRecord(a, s);
    }
    // the span belongs to kernel variant (kind, param) on launches of shape (w, h, n): `launches` launches over `px` level
    // pixels (x batch) that advance `px_steps` pixel-steps (FED kinds)
    void kernel(uint32_t kind, uint32_t param, uint32_t w, uint32_t h, uint32_t n, uint64_t launches, uint64_t px, uint64_t px_steps = 0) {
        if (!on) return;
        std::lock_guard<std::mutex> lk(c->ev_m);
        for (size_t i = 0; i < c->rows.size() && row < 0; ++i) {
            const akz_kernel_row& r = c->rows[i];
            if (r.stage == (uint32_t)stage && r.kind == kind && r.param == param && r.w == w && r.h == h && r.n == n) row = (int)i;
        }
        if (row < 0) {
            akz_kernel_row r{};
            r.stage = (uint32_t)stage, r.kind = kind, r.param = param, r.w = w, r.h = h, r.n = n;
            c->rows.push_back(r);
            row = (int)c->rows.size() - 1;
        }
        c->rows[(size_t)row].launches += launches;
        c->rows[(size_t)row].px += px;
        c->rows[(size_t)row].px_steps += px_steps;
    }
    ~StageTimer() {
        if (!on) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(c->ev_m);
        c->spans.push_back({stage, a, b, row});
    }
};
AKZ_LOCAL inline void ev_put(akz_ctx* c, hipEvent_t e) {
    if (!e) return;
    std::lock_guard<std::mutex> lk(c->ev_m);
    c->ev_pool.push_back(e);
}
AKZ_LOCAL inline void resolve_spans(akz_ctx* c) {
    std::lock_guard<std::mutex> lk(c->ev_m);
    std::vector<akz_ctx::Span> pending;
    for (auto& sp : c->spans) {
        if (hipEventQuery(sp.b) != hipSuccess) {  // still in flight (a later job): keep for the next call
            pending.push_back(sp);
            continue;
        }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            c->prof.ms[sp.stage] += (double)ms;
            if (sp.row >= 0 && (size_t)sp.row < c->rows.size()) c->rows[(size_t)sp.row].ms += (double)ms;
        }
        c->ev_pool.push_back(sp.a);
        c->ev_pool.push_back(sp.b);
    }
    c->spans.swap(pending);
}
AKZ_LOCAL inline double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

AKZ_LOCAL inline int ensure(akz_ctx* c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return AKZ_OK;
    if (b.p) {
        AKZ_HIP_TRY(hipStreamSynchronize(c->main));
        if (c->aux) AKZ_HIP_TRY(hipStreamSynchronize(c->aux));
        if (c->coarse) AKZ_HIP_TRY(hipStreamSynchronize(c->coarse));
        if (c->pre) AKZ_HIP_TRY(hipStreamSynchronize(c->pre));
        if (c->copy) AKZ_HIP_TRY(hipStreamSynchronize(c->copy));
        AKZ_HIP_TRY(hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    AKZ_HIP_TRY(hipMalloc(&b.p, want));
    b.bytes = want;
    return AKZ_OK;
}
AKZ_LOCAL inline int ensure_pinned(akz_ctx* c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return AKZ_OK;
    if (b.p) {
        AKZ_HIP_TRY(hipStreamSynchronize(c->main));
        if (c->aux) AKZ_HIP_TRY(hipStreamSynchronize(c->aux));
        AKZ_HIP_TRY(hipHostFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    AKZ_HIP_TRY(hipHostMalloc(&b.p, want, hipHostMallocDefault));
    b.bytes = want;
    return AKZ_OK;
}
// The auxiliary stream carries the finish-side copies and the per-keypoint kernels.  (A lowest-priority
// stream was measured and made no difference to the main-stream kernels, so it is a plain stream.)
AKZ_LOCAL inline int ensure_aux(akz_ctx* c) {
    // (the matcher's entry points bind without draining the finisher thread, whose finish half creates the stream too)
    std::lock_guard<std::mutex> lk(c->aux_m);
    if (!c->aux) AKZ_HIP_TRY(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    return AKZ_OK;
}
// wait until the context's finisher thread (if any) has nothing queued or running; a no-op on that thread itself
AKZ_LOCAL inline void finisher_drain(akz_ctx* c) {
    Finisher* f = c->fin.get();
    if (!f || f->th.get_id() == std::this_thread::get_id()) return;
    std::unique_lock<std::mutex> lk(f->m);
    f->done.wait(lk, [&] { return f->in_flight == 0; });
}
AKZ_LOCAL inline void finisher_stop(akz_ctx* c) {
    Finisher* f = c->fin.get();
    if (!f) return;
    {
        std::unique_lock<std::mutex> lk(f->m);
        f->done.wait(lk, [&] { return f->in_flight == 0; });
        f->quit = true;
    }
    f->wake.notify_all();
    if (f->th.joinable()) f->th.join();
    c->fin.reset();
}
// drain_self = false: extract_begin on a context whose jobs are finished by its own thread -- the two halves touch disjoint
// state (see akz_ctx) and run side by side; every other entry point waits until that thread is idle
AKZ_LOCAL inline int bind(akz_ctx* c, bool lanes_too = true, bool drain_self = true) {
    if (!c) {
        set_error("null context");
        return AKZ_ERR_INVALID_ARG;
    }
    if (c->dead) {
        set_error("the context of this object was destroyed");
        return AKZ_ERR_INVALID_ARG;
    }
    if (drain_self) finisher_drain(c);
    if (lanes_too)  // (every call but the one that deals a job to a lane: setters forward to the lanes, queries read them)
        for (akz_ctx* l : c->lanes) finisher_drain(l);
    AKZ_HIP_TRY(hipSetDevice(c->device));
    // The HIP runtime keeps ONE last-error slot per thread, shared with every other user of the runtime in the
    // process (PyTorch probes that fail on purpose, ...): drop whatever is in it so that the hipGetLastError()
    // checks after this entry point's launches report this entry point's errors only.
    (void)hipGetLastError();
    return AKZ_OK;
}


// ---- shared between the translation units of the C ABI ----------------------------------------------------------------
AKZ_LOCAL inline size_t plane_bytes(uint32_t w, uint32_t h, uint32_t n) { return (size_t)w * h * n * sizeof(float); }
// akz_api.cpp
AKZ_LOCAL int level_info_out(const std::vector<LevelPlan>& plan, uint64_t level, double* etime, double* esigma, uint32_t* octave,
                   uint32_t* sublevel, uint32_t* sigma_size, uint32_t* lw, uint32_t* lh, uint32_t* det_sigma, uint64_t* n_tau,
                   double* tau, uint64_t tau_cap);
// akz_ops.cpp: the launch helpers behind the per-op entry points, which the extraction pipeline calls as well
template <typename T>
AKZ_LOCAL int gaussian_blur_impl(akz_ctx* c, const T* d_in, float* d_out, uint32_t w, uint32_t h, uint32_t n, float sigma);
// level 0 of a small job in two launches (k_head, k_contrast_hist_final); *fused = false: not this job (the separate stages follow)
template <typename T>
AKZ_LOCAL int head_impl(akz_ctx* c, const T* d_in, float* d_lt0, float* d_blurred, float* d_gx, float* d_gy, uint32_t w, uint32_t h, uint32_t n,
                        float sigma0, double percentile, double gscale, uint64_t nbins, double* d_k_out, bool* fused, uint32_t* d_zero_word = nullptr);
AKZ_LOCAL int contrast_impl(akz_ctx* c, const float* d_in, uint32_t w, uint32_t h, uint32_t n, double percentile, double gscale, uint64_t nbins,
                  double* d_k_out);
AKZ_LOCAL uint32_t fed_max_fuse(const akz_ctx* c, uint32_t w, uint32_t h, uint32_t n);
AKZ_LOCAL uint32_t fed_num_launches(const akz_ctx* c, uint32_t n_tau, uint32_t w, uint32_t h, uint32_t n);
AKZ_LOCAL float* fed_dst(uint32_t launches, uint32_t k /*1-based*/, float* A, float* B);
// next / next_done: the last launch may also prepare the NEXT level (launch::fed_fused's epilogue); *next_done says whether it did
AKZ_LOCAL int fed_impl(akz_ctx* c, const float* in, float* A, float* B, const float* lflow, float* lstep, uint32_t w, uint32_t h, uint32_t n,
             const double* taus, uint32_t n_tau, const launch::FedNextPrep* next = nullptr, bool* next_done = nullptr);
AKZ_LOCAL int detector_family(const akz_ctx* c, uint32_t sigma, uint32_t w, uint32_t h, uint32_t n, float border_m, bool keep_second,
                    bool nms = true);
AKZ_LOCAL int detector_impl(akz_ctx* c, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx, float* lyy, float* lxy,
                  float* ldet_out, uint32_t w, uint32_t h, uint32_t n);
// akz_extract.cpp
AKZ_LOCAL int place_streams(akz_ctx* c);
AKZ_LOCAL int device_libm_mode(akz_ctx* c);  // 0: host libm; 1 / 2: the device's FMA / SSE2 forms reproduce it (akz_libm.hpp)
