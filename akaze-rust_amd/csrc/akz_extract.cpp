// C ABI of libakaze_hip.so, part 3: extract_features -- the begin half (scale space, detectors, candidates), the finish half
// (selection, orientation, descriptors), jobs and lanes, results and their accessors, stream placement.
#include "akz_ctx.hpp"

// ---------------------------------------------------------------------------------------------
// extract_features
// ---------------------------------------------------------------------------------------------
struct akz_result {
    akz_ctx* ctx = nullptr;
    akz_config cfg;
    uint32_t w = 0, h = 0, n = 0, flags = 0;
    uint64_t big_px = 0;            // the job-size gate this job was begun under (gates::kBigPxSync / kBigPxAsync)
    std::vector<LevelPlan> plan;
    void* slab = nullptr;
    size_t slab_bytes = 0;
    float* planes[kMaxLevels][10];  // image 0 of the batch; stride = level w*h
    double* d_k = nullptr;          // inside the slab
    std::vector<double> k_host;
    std::vector<std::vector<akz_keypoint>> kps;
    std::vector<uint8_t> rows64;             // host copy of the 64-byte rows (all images)
    uint8_t* d_desc64 = nullptr;             // all images back to back, 64-byte rows
    size_t desc_block_bytes = 0;             // pooled device block behind d_desc64
    std::vector<uint64_t> desc_off;          // first row of each image in d_desc64
    std::vector<uint64_t> n_extrema;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int slab_acquire(akz_ctx* c, size_t bytes, void** p, size_t* got) {
    bytes = align_up(std::max<size_t>(bytes, 256), 256);
    std::lock_guard<std::mutex> lk(c->slab_m);
    for (size_t i = 0; i < c->slab_pool.size(); ++i)
        if (c->slab_pool[i].first >= bytes && c->slab_pool[i].first <= bytes + bytes / 4 + 65536) {
            *p = c->slab_pool[i].second;
            *got = c->slab_pool[i].first;
            c->slab_pool.erase(c->slab_pool.begin() + (long)i);
            return AKZ_OK;
        }
    bytes += bytes / 8;  // head-room so that the next, slightly larger request can reuse the block
    AKZ_HIP_TRY(hipMalloc(p, bytes));
    *got = bytes;
    return AKZ_OK;
}
static void slab_release(akz_ctx* c, void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(c->slab_m);
    if (c->slab_pool.size() >= 8) {
        (void)hipStreamSynchronize(c->main);
        if (c->aux) (void)hipStreamSynchronize(c->aux);
        if (c->coarse) (void)hipStreamSynchronize(c->coarse);
        if (c->pre) (void)hipStreamSynchronize(c->pre);
        if (c->copy) (void)hipStreamSynchronize(c->copy);
        (void)hipFree(c->slab_pool.front().second);
        c->slab_pool.erase(c->slab_pool.begin());
    }
    c->slab_pool.emplace_back(bytes, p);
}

// An extraction in flight: everything up to the NMS candidates is enqueued on the context's
// stream by extract_begin (no host synchronisation); extract_finish picks the candidates up on
// the auxiliary stream once `nms_done` fires and runs the host keypoint logic, orientation and
// descriptors.  With two jobs in flight the host phase of one batch runs under the kernels of the
// next while the scale-space kernels of both stay serialised on one stream.
struct ResultDeleter {
    void operator()(akz_result* r) const;
};
struct akz_job {
    std::unique_ptr<akz_result, ResultDeleter> r;
    int slot = -1;            // candidate / counter buffers used by this job
    uint32_t cap = 0;         // candidate capacity per image
    hipEvent_t nms_done = nullptr;
    uint64_t seq = 0;         // position in the context's order of begins (fed_ev ring)
    double t_begin_ms = 0.0;
    // eager finish: the lane's thread runs the finish half and leaves its outcome here (guarded by fin->m)
    std::shared_ptr<Finisher> fin;
    bool finished = false;
    int rc = 0;
    akz_result* out = nullptr;
    std::string err;
    // jobs of the context that the caller has begun and not collected yet (this one included), counted until the job object
    // goes: a job begun with none other in the caller's hand is being waited for, one begun with company is part of a stream
    std::shared_ptr<std::atomic<int>> in_hand;
    bool alone_at_begin = true;
    hipStream_t done_stream = nullptr;  // the stream the begin chain ended on (the main stream, or the forked coarse chain's)
    ~akz_job() {
        if (in_hand) --*in_hand;
    }
};

static void result_release_device(akz_result* r) {
    if (r->ctx && r->ctx->dead) {  // the pools went away with the context: hand the blocks back to the runtime
        if (r->slab) (void)hipFree(r->slab);
        if (r->d_desc64) (void)hipFree(r->d_desc64);
    } else {
        if (r->slab) slab_release(r->ctx, r->slab, r->slab_bytes);
        if (r->d_desc64) slab_release(r->ctx, r->d_desc64, r->desc_block_bytes);
    }
    r->slab = nullptr;
    r->d_desc64 = nullptr;
}
// Every akz_result is deleted through here: a result may outlive akz_ctx_destroy (a caller that frees in the
// "wrong" order); the context struct itself is then released with its last result.
static void result_delete(akz_result* r) {
    if (!r) return;
    akz_ctx* c = r->ctx;
    result_release_device(r);
    delete r;
    if (c && --c->live_results == 0 && c->dead) delete c;
}
void ResultDeleter::operator()(akz_result* r) const { result_delete(r); }
// a job that will not produce a result hands back what it holds on its context (the shell itself is deleted by the caller)
static void job_release(akz_job* j) {
    akz_ctx* c = j->r ? j->r->ctx : nullptr;
    if (!c) return;
    (void)hipStreamSynchronize(c->main);
    if (c->coarse) (void)hipStreamSynchronize(c->coarse);  // a forked batch completes on the coarse stream
    if (c->pre) (void)hipStreamSynchronize(c->pre);
    if (c->copy) (void)hipStreamSynchronize(c->copy);
    if (j->slot >= 0) c->slot_busy[j->slot] = false;
    j->slot = -1;
    ev_put(c, j->nms_done);
    j->nms_done = nullptr;
    result_release_device(j->r.get());
}
// the outcome of an eagerly finished job, once its lane's thread is through with it
static void job_wait(akz_job* j) {
    if (!j->fin) return;
    std::unique_lock<std::mutex> lk(j->fin->m);
    j->fin->done.wait(lk, [&] { return j->finished; });
}
static void job_destroy(akz_job* j) {
    if (!j) return;
    if (j->fin) {
        job_wait(j);
        if (j->out) result_delete(j->out);
    } else {
        job_release(j);
    }
    delete j;
}

// ---- stream placement ----------------------------------------------------------------------------------------------
// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES in-order hardware queues (4 by default), and the
// command processor of the chip has FOUR pipes: hardware queues k and k + 4 share one, and a pipe switches between its
// queues at ~25 us a switch (tools/queues/queue_probe.hip: 40 tiny kernels on each of two streams drain in 0.23 ms on
// different pipes, in 1.0 ms on one; a dependency across two queues of one pipe costs +55 us).  A batch pipeline that
// keeps four streams busy -- the caller's, the coarse chain's, the finish half's, the uploads' / early stages' -- therefore
// wants exactly four queues on four pipes: two of its streams on one QUEUE serialise everything behind everything
// (13.4 -> 7.7 Gpix/s, round 3), two on one PIPE cost 15 % (11.7 against 13.8 Gpix/s, round 4).  Which queue a stream got
// cannot be asked, so it is measured.
// Do streams a and b get in each other's way?  (1) a 120 us single-wave spin on each, from idle: on one hardware queue the
// second starts when the first has finished; (2) 24 tiny kernels on each, interleaved: on one pipe they drain several
// times slower than `alone_ms`, what 24 of them take on one stream.
// The verdict is a pure function of the probe's timings (akz::placement_verdict, unit-tested on the recorded timings of
// profiles/r04_queue_probe.txt).  Everything that disturbs a measurement -- the host thread preempted between two launches,
// a profiler that serialises dispatches, a neighbour's kernels -- can only make it LONGER, so a measurement that says
// "shared" is repeated (up to three in all) and the shortest one decides: a stream is only given up on evidence that
// repeats.
int akz::placement_verdict(float spin_pair_ms, float tiny_pair_ms, float tiny_alone_ms, float spin_ms) {
    int v = 0;
    // one hardware queue: the second spin starts when the first has finished (2 x; side by side 1.0-1.3 x)
    if (spin_pair_ms > 1.6f * spin_ms) v |= kPlaceQueue;
    // one pipe of the command processor: 24 + 24 interleaved tiny kernels drain ~8 x slower than 24 on one stream (different
    // pipes: 1.6-2.5 x); between 3 x and 6 x a single measurement is not trusted either way
    if (tiny_pair_ms >= 0.0f) {
        if (tiny_pair_ms > 4.0f * tiny_alone_ms) v |= kPlacePipe;
        if (tiny_pair_ms > 3.0f * tiny_alone_ms && tiny_pair_ms < 6.0f * tiny_alone_ms) v |= kPlaceAmbiguous;
    }
    if (spin_pair_ms > 1.35f * spin_ms && spin_pair_ms < 1.9f * spin_ms) v |= kPlaceAmbiguous;
    return v;
}
static int streams_interfere(akz_ctx* c, hipStream_t a, hipStream_t b, float alone_ms, bool* bad) {
    constexpr uint32_t kDelayUs = 120;
    constexpr float kSpinMs = (float)kDelayUs * 1e-3f;
    *bad = false;
    float spin_best = 1e30f, tiny_best = 1e30f;
    for (int attempt = 0; attempt < 3; ++attempt) {
        float ms = 0.0f;
        AKZ_HIP_TRY(hipEventRecord(c->probe_ev[0], a));
        launch::delay(a, kDelayUs);
        launch::delay(b, kDelayUs);
        AKZ_HIP_TRY(hipEventRecord(c->probe_ev[1], b));
        AKZ_HIP_TRY(hipGetLastError());
        AKZ_HIP_TRY(hipEventSynchronize(c->probe_ev[1]));
        AKZ_HIP_TRY(hipStreamSynchronize(a));
        AKZ_HIP_TRY(hipEventElapsedTime(&ms, c->probe_ev[0], c->probe_ev[1]));
        spin_best = std::min(spin_best, ms);
        if (!(akz::placement_verdict(spin_best, -1.0f, alone_ms, kSpinMs) & (akz::kPlaceQueue | akz::kPlaceAmbiguous))) break;
        if (attempt < 2) ++c->place_retries;
    }
    if (akz::placement_verdict(spin_best, -1.0f, alone_ms, kSpinMs) & akz::kPlaceQueue) {
        *bad = true;
        return AKZ_OK;
    }
    for (int attempt = 0; attempt < 3; ++attempt) {
        float ms = 0.0f;
        AKZ_HIP_TRY(hipEventRecord(c->probe_ev[0], a));
        for (int k = 0; k < 24; ++k) {
            launch::delay(a, 1);
            launch::delay(b, 1);
        }
        AKZ_HIP_TRY(hipEventRecord(c->probe_ev[1], b));
        AKZ_HIP_TRY(hipGetLastError());
        AKZ_HIP_TRY(hipEventSynchronize(c->probe_ev[1]));
        AKZ_HIP_TRY(hipStreamSynchronize(a));
        AKZ_HIP_TRY(hipEventElapsedTime(&ms, c->probe_ev[0], c->probe_ev[1]));
        tiny_best = std::min(tiny_best, ms);
        if (!(akz::placement_verdict(0.0f, tiny_best, alone_ms, kSpinMs) & (akz::kPlacePipe | akz::kPlaceAmbiguous))) break;
        if (attempt < 2) ++c->place_retries;
    }
    *bad = (akz::placement_verdict(0.0f, tiny_best, alone_ms, kSpinMs) & akz::kPlacePipe) != 0;  // (different pipes: 1.8 x, one pipe: 8 x)
    return AKZ_OK;
}
// The first large batch of a context checks that its busy streams do not share a hardware queue or a pipe.  A stream of the
// library that does is replaced by a fresh one (up to eight tries: the runtime hands a new stream the least-used queue,
// and the rejected ones stay alive until the end so that they keep theirs occupied).  The early stages of a batch run
// on the copy stream (idle for resident frames; for host frames the blur has to follow the upload anyway) -- a fifth busy
// stream would have to share a pipe with one of the four.  About 0.4 ms per pair, once per context.
// the probe's working set: streams already accepted, rejected ones (kept alive until the end so that they keep their queues
// occupied), the time 24 tiny kernels take on one stream
struct StreamPlacer {
    akz_ctx* c;
    float alone_ms = 0.0f;
    std::vector<hipStream_t> accepted, rejected;
    explicit StreamPlacer(akz_ctx* ctx) : c(ctx) {}
    ~StreamPlacer() {
        for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    }
    int calibrate(hipStream_t on) {
        for (hipEvent_t& e : c->probe_ev)
            if (!e) AKZ_HIP_TRY(hipEventCreate(&e));
        launch::delay(on, 1);  // (the first launch of a kernel loads its code object: not part of a measurement)
        AKZ_HIP_TRY(hipStreamSynchronize(on));
        alone_ms = 1e30f;
        for (int attempt = 0; attempt < 3; ++attempt) {  // (the shortest of three: see placement_verdict)
            float ms = 0.0f;
            AKZ_HIP_TRY(hipEventRecord(c->probe_ev[0], on));
            for (int k = 0; k < 24; ++k) launch::delay(on, 1);
            AKZ_HIP_TRY(hipEventRecord(c->probe_ev[1], on));
            AKZ_HIP_TRY(hipEventSynchronize(c->probe_ev[1]));
            AKZ_HIP_TRY(hipEventElapsedTime(&ms, c->probe_ev[0], c->probe_ev[1]));
            alone_ms = std::min(alone_ms, ms);
        }
        return AKZ_OK;
    }
    int collides(hipStream_t x, bool* hit) {
        *hit = false;
        for (hipStream_t a : accepted) {
            AKZ_TRY(streams_interfere(c, a, x, alone_ms, hit));
            if (*hit) return AKZ_OK;
        }
        return AKZ_OK;
    }
    // *slot ends up a stream that interferes with none of `accepted` (and joins them), or keeps its value (free = false)
    int settle(hipStream_t* slot, bool* free) {
        bool hit = false;
        AKZ_TRY(collides(*slot, &hit));
        for (int attempt = 0; hit && attempt < 8; ++attempt) {
            hipStream_t fresh = nullptr;
            AKZ_HIP_TRY(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
            bool fresh_hit = false;
            const int st = collides(fresh, &fresh_hit);
            if (st != AKZ_OK) {
                rejected.push_back(fresh);
                return st;
            }
            if (!fresh_hit) {
                rejected.push_back(*slot);
                *slot = fresh;
                hit = false;
                ++c->place_replaced;
            } else {
                rejected.push_back(fresh);
            }
        }
        *free = !hit;
        accepted.push_back(*slot);
        return AKZ_OK;
    }
};
int place_streams(akz_ctx* c) {
    if (c->is_lane) {
        c->placed = true;
        return AKZ_OK;
    }
    // a stream that is being captured into a graph cannot be synchronised or timed: the probe waits for a call outside
    // the capture (akz_ctx_warmup is the place to run it once, up front)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (c->main && hipStreamIsCapturing(c->main, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return AKZ_OK;
    (void)hipGetLastError();
    c->placed = true;
    finisher_drain(c);  // (the finish half uses c->aux)
    AKZ_TRY(ensure_aux(c));
    if (!c->coarse) AKZ_HIP_TRY(hipStreamCreateWithFlags(&c->coarse, hipStreamNonBlocking));
    if (!c->copy) AKZ_HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    if (c->sched[2]) {  // (measurement: no probe -- streams as the runtime placed them)
        c->pre_mode = 2;
        return AKZ_OK;
    }
    AKZ_HIP_TRY(hipStreamSynchronize(c->main));
    AKZ_HIP_TRY(hipStreamSynchronize(c->aux));
    AKZ_HIP_TRY(hipStreamSynchronize(c->coarse));
    AKZ_HIP_TRY(hipStreamSynchronize(c->copy));
    StreamPlacer sp(c);
    AKZ_TRY(sp.calibrate(c->main));
    sp.accepted.push_back(c->main);
    bool free_coarse = false, free_aux = false, free_copy = false;
    AKZ_TRY(sp.settle(&c->coarse, &free_coarse));
    AKZ_TRY(sp.settle(&c->aux, &free_aux));
    AKZ_TRY(sp.settle(&c->copy, &free_copy));
    c->place_collisions = (free_coarse ? 0 : 1) + (free_aux ? 0 : 1) + (free_copy ? 0 : 1);
    c->pre_mode = free_copy ? 2 : 0;
    return AKZ_OK;
}
// Lanes: a lane enqueues both halves of its jobs on its one stream, and the point of lanes is that their launch chains run
// side by side -- the same check for the lanes' streams (among themselves: the caller's stream carries only the events
// that order a lane behind the caller's work).  Up to four lanes can have a pipe each.
static int place_lanes(akz_ctx* c) {
    if (c->sched[2] || c->lanes.empty()) return AKZ_OK;
    for (akz_ctx* l : c->lanes) AKZ_HIP_TRY(hipStreamSynchronize(l->main));
    StreamPlacer sp(c);
    AKZ_TRY(sp.calibrate(c->lanes[0]->main));
    c->lane_collisions = 0;
    for (akz_ctx* l : c->lanes) {
        bool free = false;
        hipStream_t st = l->main;
        AKZ_TRY(sp.settle(&st, &free));
        if (st != l->main) {  // (the replaced stream is destroyed with the placer's rejects; the lane owns the new one)
            l->main = l->stream = st;
        }
        if (!free) ++c->lane_collisions;
    }
    return AKZ_OK;
}

// Streams of another component of the process that are busy beside a context's (the exchange stream of akz_comm: one RCCL
// collective per step): they get queues and pipes that the caller's stream, the coarse chain's and the finish half's do not
// use -- with four pipes that leaves the copy stream's, which carries the least.
int akz::place_streams_beside(akz_ctx* c, hipStream_t* slots, int n_slots, int* still_shared) {
    AKZ_TRY(bind(c));
    if (still_shared) *still_shared = 0;
    if (c->is_lane || c->sched[2] || n_slots <= 0) return AKZ_OK;
    if (!c->placed) AKZ_TRY(place_streams(c));
    AKZ_HIP_TRY(hipStreamSynchronize(c->main));
    AKZ_HIP_TRY(hipStreamSynchronize(c->coarse));
    AKZ_HIP_TRY(hipStreamSynchronize(c->aux));
    for (int i = 0; i < n_slots; ++i) AKZ_HIP_TRY(hipStreamSynchronize(slots[i]));
    StreamPlacer sp(c);
    AKZ_TRY(sp.calibrate(c->main));
    sp.accepted = {c->main, c->coarse, c->aux};
    for (int i = 0; i < n_slots; ++i) {
        bool free = false;
        AKZ_TRY(sp.settle(&slots[i], &free));
        sp.accepted.pop_back();  // (the component's own streams may share among themselves)
        if (!free && still_shared) ++*still_shared;
    }
    return AKZ_OK;
}

template <typename T>
static int extract_begin(akz_ctx* c, const T* d_imgs, uint32_t w, uint32_t h, uint32_t n, const akz_config* cfgp,
                         uint32_t flags, akz_job** out, int want_slot = -1, hipEvent_t input_ready = nullptr, bool sync_call = false) {
    if (!out) return AKZ_ERR_INVALID_ARG;
    *out = nullptr;
    AKZ_TRY(bind(c, true, c && c->is_lane));  // (a lane's finish half shares the lane's one stream: begin waits for it)
    c->big_px = sync_call ? c->big_px_sync : c->big_px_async;  // (akz_gates.hpp)
    struct GateRestore {  // the per-op entry points (akz_op_*) use the same helpers: they see the begin / finish interface's gate
        akz_ctx* c;
        ~GateRestore() {
            c->big_px = c->big_px_async;
            c->launch_min_px = akz_ctx::kLaunchMarchPx;
        }
    } gate_restore{c};
    if (!d_imgs || !cfgp || n == 0) {
        set_error("extract: null image/config or empty batch");
        return AKZ_ERR_INVALID_ARG;
    }
    int slot = want_slot;
    for (int i = 0; i < akz_ctx::kSlots && slot < 0; ++i)
        if (!c->slot_busy[i]) slot = i;
    if (slot < 0 || c->slot_busy[slot]) {
        set_error("extract_begin: too many extractions in flight on this context (finish one first)");
        return AKZ_ERR_INVALID_ARG;
    }
    std::unique_ptr<akz_job> job(new akz_job);
    job->in_hand = c->in_hand;
    job->alone_at_begin = job->in_hand->fetch_add(1) == 0;
    job->r.reset(new akz_result);
    akz_result* r = job->r.get();
    r->ctx = c;
    ++c->live_results;
    r->cfg = *cfgp;
    r->w = w; r->h = h; r->n = n; r->flags = flags;
    r->big_px = c->big_px;  // (the finish half may run on another thread while the next job is begun with another gate)
    AKZ_TRY(build_plan(w, h, r->cfg, r->plan));
    const akz_config& cfg = r->cfg;
    const std::vector<LevelPlan>& plan = r->plan;
    const size_t L = plan.size();
    const bool keep_all = (flags & AKZ_KEEP_ALL_PLANES) != 0;
    hipStream_t s = c->stream;

    // ---- pyramid slab layout ----
    std::memset(r->planes, 0, sizeof(r->planes));
    size_t off = 0;
    std::vector<std::pair<float**, size_t>> fix;  // (slot, offset)
    auto want = [&](size_t lvl, int plane) {
        const size_t bytes = align_up(plane_bytes(plan[lvl].w, plan[lvl].h, n), 256);
        fix.emplace_back(&r->planes[lvl][plane], off);
        off += bytes;
    };
    for (size_t l = 0; l < L; ++l) {
        want(l, AKZ_LT);
        if (l > 0) want(l, AKZ_LSMOOTH);  // level 0: Lsmooth is a clone of Lt (lib.rs:58) -> alias
        want(l, AKZ_LX);
        want(l, AKZ_LY);
        want(l, AKZ_LDET);
        if (l > 0) want(l, AKZ_LFLOW);
        if (keep_all) {
            want(l, AKZ_LXX);
            want(l, AKZ_LYY);
            want(l, AKZ_LXY);
            if (l > 0) want(l, AKZ_LSTEP);
        }
    }
    const size_t k_off = off;
    off += align_up((size_t)n * sizeof(double), 256);
    AKZ_TRY(slab_acquire(c, off, &r->slab, &r->slab_bytes));
    for (auto& f : fix) *f.first = (float*)((char*)r->slab + f.second);
    r->planes[0][AKZ_LSMOOTH] = r->planes[0][AKZ_LT];
    r->d_k = (double*)((char*)r->slab + k_off);
    auto P = [&](size_t l, int p) { return r->planes[l][p]; };
    struct Guard {  // return the device blocks to the pool on any early error exit
        akz_result* r;
        bool armed = true;
        ~Guard() {
            if (!armed) return;
            // work already enqueued (possibly on the coarse stream, which nothing has joined yet) still writes the slab
            akz_ctx* c = r->ctx;
            if (c && c->coarse) (void)hipStreamSynchronize(c->coarse);
            if (c && c->pre) (void)hipStreamSynchronize(c->pre);
            if (c && c->copy) (void)hipStreamSynchronize(c->copy);
            result_release_device(r);
        }
    } guard{r};

    job->t_begin_ms = now_ms();
    // ---- detector response (detector_response.rs:38-55) + extrema candidates ----
    // One append list for the whole batch (image id stored per candidate): a single D2H later.  The detector of
    // level l needs only Lsmooth_l; its launches follow the whole diffusion chain on the same stream (running them on
    // a side stream next to the diffusion was +3 % with the round-1 kernels and is -15 % with the column march, which
    // saturates the store path on its own; with only the half-resolution octave's detectors on the side stream it is
    // still -5 %: removed).
    uint32_t cap = (uint32_t)std::min<uint64_t>((uint64_t)n * std::max<uint32_t>(c->cand_cap_hint.load(), 16u),
                                                0x7fffffffull / sizeof(Candidate));
    // A job like the one before it (same shape) whose list was short gets a list no longer than the one-launch sort takes
    // (launch::sort_small_capacity): should this image have more candidates after all, the overflow path of the finish half
    // redoes the extrema with room for them.
    {
        const uint32_t last = c->last_total_cands.load();
        if (c->last_cand_shape.load() == (((uint64_t)w << 40) | ((uint64_t)h << 16) | n) && last > 0 &&
            (uint64_t)last * 5 / 4 + 64 <= launch::sort_small_capacity())
            cap = std::min(cap, launch::sort_small_capacity());
    }
    AKZ_TRY(ensure(c, c->cand_slot[slot], (size_t)cap * sizeof(Candidate)));
    AKZ_TRY(ensure(c, c->count_slot[slot], 256));
    uint32_t* d_count = (uint32_t*)c->count_slot[slot].p;
    Candidate* d_cand = (Candidate*)c->cand_slot[slot].p;
    if (!c->fed_ev[0])
        for (hipEvent_t& e : c->fed_ev) AKZ_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!c->pre_ev[0])
        for (hipEvent_t& e : c->pre_ev) AKZ_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const uint64_t seq = c->begin_seq.load() + 1;  // published when this job's event has been recorded
    // derivatives, Ldet and extrema candidates of level l in one or two launches on stream `st_`; false when the
    // level's kernel size has no fused form (then the multi-kernel fallback runs on the main stream at the end)
    auto detector_one_pass = [&](size_t l, hipStream_t st_) -> bool {
        const LevelPlan& lv = plan[l];
        const float thr = (float)cfg.detector_threshold, bm = border_margin(lv, cfg);
        if (c->profiling) {
            c->prof.det_launches += 1;
            c->prof.det_px += (uint64_t)lv.w * lv.h * n;
        }
        if (const int fam = detector_family(c, lv.det_sigma, lv.w, lv.h, n, bm, keep_all)) {
            StageTimer st(c, AKZ_ST_DETECTOR, st_);
            st.kernel(fam == 5 ? AKZ_KR_DETECTOR_MARCH : AKZ_KR_DETECTOR_TILED, lv.det_sigma, lv.w, lv.h, n, 1, (uint64_t)lv.w * lv.h * n);
            (fam == 5 ? launch::detector_march : launch::detector_tiled_fused)(
                st_, P(l, AKZ_LSMOOTH), lv.det_sigma, P(l, AKZ_LX), P(l, AKZ_LY), P(l, AKZ_LXX), P(l, AKZ_LYY),
                P(l, AKZ_LXY), P(l, AKZ_LDET), lv.w, lv.h, n, (uint32_t)l, thr, bm, d_cand, cap, d_count);
            return true;
        }
        if (launch::detector_nms_fused_supported(lv.det_sigma)) {
            StageTimer st(c, AKZ_ST_DETECTOR, st_);
            launch::detector_nms_fused(st_, P(l, AKZ_LSMOOTH), lv.det_sigma, P(l, AKZ_LX), P(l, AKZ_LY), P(l, AKZ_LXX),
                                       P(l, AKZ_LYY), P(l, AKZ_LXY), P(l, AKZ_LDET), lv.w, lv.h, n, (uint32_t)l, thr,
                                       bm, d_cand, cap, d_count);
            return true;
        }
        return false;
    };
    // ---- level 0: Lt0 = gaussian_blur(img, base_scale_offset); contrast factor (lib.rs:56-69) ----
    // Running ahead.  These two stages need nothing but the frames, and the contrast passes are bound by arithmetic, not
    // by bandwidth: when the frames are known to be complete -- the caller says so (AKZ_INPUT_READY) or this library
    // uploaded them itself (akz_extract_begin_host_*: `input_ready` is the upload's event) -- a large batch enqueues them
    // on the context's copy stream, which does NOT wait for what the context's stream still has to do for the batch before,
    // and the context's stream picks up behind them.  They then run under the previous batch's detectors instead of in front
    // of this batch's first level: 0.3-0.4 ms less on the critical path of a 5 ms step (+3.7 %, 5 x 80 steps each way).
    // (place_streams, above: on the context's stream instead when the copy stream could not be given a hardware queue and
    // a pipe of its own.)
    // The contrast scratch (c->small) is shared by the jobs of a context: a job's early stages wait for the level-0 stages
    // of the job before, on whichever stream those ran (pre_done).  Only with the march kernels (they use no other
    // context scratch).
    struct StreamRestore {  // the helpers (gaussian_blur_impl, fed_impl, StageTimer, ...) enqueue on c->stream
        akz_ctx* c;
        hipStream_t main;
        ~StreamRestore() { c->stream = main; }
    } stream_restore{c, s};
    // Jobs below gates::kTiledPrepPx take the TILED preparation family whatever their launches' sizes -- k_blur, k_contrast_max /
    // _hist, k_prep riding on the previous level's last k_fed_own launch, no resident tail: since that epilogue exists the
    // chain of few-microsecond launches beats the streaming kernels and k_level_march up to ~11 Mpx per job (akz_gates.hpp: kTiledPrepPx; profiles/
    // r06_lone_libm.txt: 2-6 x 1080p, 4-8 x 720p, lone 2-5 Mpx frames 3-9 % faster per call, 0-9 % as a stream).  The helpers
    // read c->prep_mode: it is swapped for the duration of this begin half (only the automatic mode 2 is overridden).
    struct PrepModeRestore {
        akz_ctx* c;
        int mode;
        ~PrepModeRestore() { c->prep_mode = mode; }
    } prep_mode_restore{c, c->prep_mode};
    if (c->prep_mode == 2 && c->sched[6] == 0 && (uint64_t)w * h * n < gates::kTiledPrepPx) c->prep_mode = 0;
    bool early = false;
    const bool big = (uint64_t)w * h * n >= c->big_px;
    c->launch_min_px = big ? std::min<uint64_t>(akz_ctx::kLaunchMarchPx, (uint64_t)w * h * n) : akz_ctx::kLaunchMarchPx;
    if (big && !c->placed) AKZ_TRY(place_streams(c));
    const int pre_mode = c->sched[0] == 0 ? c->pre_mode : c->sched[0] == 1 ? 2 : c->sched[0] == 2 ? 1 : 0;  // (1: a stream of its own, measurement only)
    if ((input_ready || (flags & AKZ_INPUT_READY)) && pre_mode != 0 && c->profiling < 2 && c->prep_mode == 2 && big &&
        launch::blur5_march_supported(w, h, (uint32_t)gaussian_kernel_size((float)cfg.base_scale_offset)) &&
        launch::contrast_march_supported(w, h, (uint32_t)gaussian_kernel_size(1.0f), (uint32_t)cfg.contrast_factor_num_bins)) {
        hipStream_t ps = nullptr;
        if (pre_mode == 2) {
            if (!c->copy) AKZ_HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
            ps = c->copy;
        } else {
            if (!c->pre) AKZ_HIP_TRY(hipStreamCreateWithFlags(&c->pre, hipStreamNonBlocking));
            ps = c->pre;
        }
        if (input_ready) AKZ_HIP_TRY(hipStreamWaitEvent(ps, input_ready, 0));
        if (c->pre_done) AKZ_HIP_TRY(hipStreamWaitEvent(ps, c->pre_done, 0));
        // however early the caller begins this batch, its first two stages start when the batch before goes from its
        // (VALU-bound) diffusion launches to its (bandwidth-bound) detectors: that is what they are meant to run under
        if (c->sched[1] && seq > 1)
            AKZ_HIP_TRY(hipStreamWaitEvent(ps, (c->sched[1] == 2 ? c->pre_ev : c->fed_ev)[(seq - 1) % akz_ctx::kFedRing], 0));
        c->stream = ps;
        early = true;
    }
    // the job's candidate counter is cleared on the stream of its first stage (every detector launch comes behind that):
    // by that stage's kernel itself where it is k_head, by a fill otherwise
    // (a small job: both stages in two launches -- akz_ops.cpp: head_impl; level 1, where it continues the octave, finds its Lsmooth
    // written -- the contrast factor's blur of Lt0 is the same image -- and the Scharr pair of it in level 0's Lx / Ly planes)
    bool head_fused = false;
    const bool level1_clone = L > 1 && plan[1].octave == plan[0].octave && plan[1].w == w && plan[1].h == h;
    AKZ_TRY(head_impl<T>(c, d_imgs, P(0, AKZ_LT), level1_clone ? P(1, AKZ_LSMOOTH) : nullptr, P(0, AKZ_LX), P(0, AKZ_LY), w, h, n,
                         (float)cfg.base_scale_offset, cfg.contrast_percentile, 1.0, cfg.contrast_factor_num_bins, r->d_k, &head_fused, d_count));
    const bool head_level1 = head_fused && level1_clone;
    if (!head_fused) {
        AKZ_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), c->stream));
        {
            StageTimer st(c, AKZ_ST_BLUR0);
            AKZ_TRY(gaussian_blur_impl<T>(c, d_imgs, P(0, AKZ_LT), w, h, n, (float)cfg.base_scale_offset));
        }
        {
            StageTimer st(c, AKZ_ST_CONTRAST);
            AKZ_TRY(contrast_impl(c, P(0, AKZ_LSMOOTH), w, h, n, cfg.contrast_percentile, 1.0,
                                  cfg.contrast_factor_num_bins, r->d_k));
        }
    }
    // every job marks the end of its level-0 stages (the last use of the context's contrast scratch): a later job that
    // runs ahead waits for exactly that, whichever stream it was recorded on
    if (!c->pre_done) AKZ_HIP_TRY(hipEventCreateWithFlags(&c->pre_done, hipEventDisableTiming));
    AKZ_HIP_TRY(hipEventRecord(c->pre_done, c->stream));
    if (early) {
        c->stream = s;
        AKZ_HIP_TRY(hipStreamWaitEvent(s, c->pre_done, 0));
    }

    // ---- levels 1..L-1 (lib.rs:78-119) ----
    AKZ_TRY(ensure(c, c->scratch[5], plane_bytes(w, h, n)));
    const std::vector<float> g1 = gaussian_kernel(1.0f, gaussian_kernel_size(1.0f));  // Lsmooth taps (lib.rs:95)
    // Fork.  From octave `fork_octave` on the levels are small: their launches (diffusion, preparation, detectors) do not
    // fill the chip and are bound by launch-to-launch latency -- about 1 ms of the step for 8 % of its pixels.  That
    // chain moves to a second stream when octave fork_octave - 1 is finished, and the main stream goes straight to the
    // detectors of the fine octaves (bandwidth-bound, 2.2 ms): the two run side by side and join before the candidate
    // list is read.  (Running two BIG kernels side by side is a loss -- see above -- so the fork is at octave 2.)
    const int fork_octave = c->sched[3] > 0 ? c->sched[3] : 2;  // (forking at octave 3 instead, octave 2 on the main stream: -4 %; sched[3]: measurement)
    // (a lone 1080p frame is a chain of dependent launches either way and only pays for the two events: measured
    // 0.596 -> 0.625 ms per streamed frame; batch-path jobs (gates::kBigPxSync / kBigPxAsync) fork)
    const uint64_t fork_min_px = c->big_px;
    hipStream_t ls = s;  // the stream the level loop enqueues on
    size_t fork_level = L;  // first level of the coarse chain
    // Resident tail: from the first level whose image fits one compute unit, ALL remaining levels (preparation and
    // every diffusion step, across octaves) are one launch with one workgroup per image (akz_resident.hip).
    // One workgroup advances an image by one diffusion step in ~2 us whatever the batch size, so a lone frame, whose
    // launch chain is bound by latency, keeps the separate launches (octave 3 of a 1080p frame: 0.11 ms as 12 launches
    // against 0.32 ms resident); a batch that forks its coarse chain onto the second stream hides that latency under
    // the fine detectors and gains what the 17 small launches cost those detectors (5.9 -> 5.3 ms per 32-frame step).
    size_t res_first = L;
    if (c->fed_mode == 2 && (c->prep_mode == 3 || (c->prep_mode == 2 && (uint64_t)w * h * n >= fork_min_px))) {
        size_t f = 1;
        while (f < L && !launch::octave_resident_supported(plan[f].w, plan[f].h)) ++f;
        f = std::max(f, L > (size_t)launch::kResidentMaxLevels ? L - (size_t)launch::kResidentMaxLevels : (size_t)1);
        size_t steps = 0;
        bool ok = true;
        for (size_t l = L; l-- > f;) {
            if (plan[l].tau.empty()) ok = false;
            steps += plan[l].tau.size();
            if (steps > (size_t)launch::kResidentMaxSteps) {  // keep the tail that fits
                steps -= plan[l].tau.size();
                f = l + 1;
                break;
            }
        }
        if (ok && f < L) res_first = f;
    }
    // (full stage profiling attributes time to stages: it keeps everything on one stream)
    if (fork_octave > 0 && c->profiling < 2 && (uint64_t)w * h * n >= fork_min_px)
        for (size_t i = 1; i < L && fork_level == L; ++i)
            if ((int)plan[i].octave >= fork_octave) fork_level = i;
    // small frames in a large batch: the resident tail may start before octave 2 -- the chain then forks where the tail
    // starts (run_levels stops at the resident launch, which covers every level behind it: a fork behind that point
    // would run those levels a second time as separate launches)
    if (fork_level < L && res_first < fork_level) fork_level = res_first;
    // sched[5] (measurement, profiles/r06_interleave.txt): the detector of a fine level right behind the kernel that wrote its
    // Lsmooth instead of after the whole fine chain -- does the detector then find (part of) the plane in the Infinity Cache?
    std::vector<char> det_done(L, 0);
    auto interleave_detector = [&](size_t l) {
        if (c->sched[5] && l < fork_level && !det_done[l] &&
            detector_family(c, plan[l].det_sigma, plan[l].w, plan[l].h, n, border_margin(plan[l], cfg), keep_all) == 5 && detector_one_pass(l, ls))
            det_done[l] = 1;
    };
    // prepared[l]: level l's Lsmooth and Lflow have been written by the last diffusion launch of level l - 1 (k_fed_own's epilogue)
    std::vector<char> prepared(L + 1, 0);
    const uint64_t level_min_px = gates::kLevelMarchPx;
    auto takes_level_march = [&](size_t l) {
        return !plan[l].tau.empty() && c->fed_mode == 2 && launch::level_march_supported(plan[l].w, plan[l].h) &&
               (c->prep_mode == 3 || (c->prep_mode == 2 && (uint64_t)plan[l].w * plan[l].h * n >= level_min_px));
    };
    auto takes_stream_prep = [&](size_t l) {
        const bool half_l = plan[l].octave > plan[l - 1].octave;
        return c->prep_mode != 0 && launch::prep_stream_supported(plan[l].w, plan[l].h) &&
               (c->prep_mode == 1 || (c->prep_mode >= 2 && !half_l && (uint64_t)plan[l].w * plan[l].h * n >= c->stream_min_px));
    };
    auto run_levels = [&](size_t lo, size_t hi) -> int {
    for (size_t i = lo; i < hi; ++i) {
        const LevelPlan& lv = plan[i];
        const LevelPlan& pv = plan[i - 1];
        if (i >= 2) interleave_detector(i - 1);  // (level i - 1 is complete; its Lsmooth was written one launch group ago)

        if (i == res_first) {
            std::vector<launch::ResidentLevel> rl;
            std::vector<std::vector<float>> ht(L);
            uint64_t px_steps = 0;
            for (size_t l = i; l < L; ++l) {
                for (double t : plan[l].tau) ht[l].push_back(0.5f * (float)t);
                rl.push_back(launch::ResidentLevel{P(l, AKZ_LT), P(l, AKZ_LSMOOTH), P(l, AKZ_LFLOW), keep_all ? P(l, AKZ_LSTEP) : nullptr,
                                                   plan[l].w, plan[l].h, plan[l].octave > plan[l - 1].octave,
                                                   (uint32_t)plan[l].tau.size(), ht[l].data(), plan[l].octave});
                px_steps += (uint64_t)plan[l].w * plan[l].h * n * plan[l].tau.size();
            }
            StageTimer st(c, AKZ_ST_FED);
            st.kernel(AKZ_KR_OCTAVE_RESIDENT, (uint32_t)rl.size(), plan[i].w, plan[i].h, n, 1, 0, px_steps);
            launch::octave_resident(ls, P(i - 1, AKZ_LT), pv.w, pv.h, n, rl.data(), (uint32_t)rl.size(), g1.data(), r->d_k);
            if (c->profiling) {
                c->prof.fed_launches += 1;
                c->prof.fed_px_steps += px_steps;
            }
            AKZ_HIP_TRY(hipGetLastError());
            break;
        }
        float* A = P(i, AKZ_LT);
        const bool on_coarse = i >= fork_level;  // the coarse chain has its own ping-pong plane (it outlives the batch's join)
        if (on_coarse) AKZ_TRY(ensure(c, c->scratch_coarse, plane_bytes(plan[fork_level].w, plan[fork_level].h, n)));
        float* B = (float*)(on_coarse ? c->scratch_coarse.p : c->scratch[5].p);
        const uint32_t n_tau = (uint32_t)lv.tau.size();
        const bool half = lv.octave > pv.octave;
        // FED input: the previous level's final Lt (clone, lib.rs:92, no copy needed) or its 2x2 mean
        // (lib.rs:82) materialised by k_prep into a buffer the first FED launch does not write.
        const float* fed_in = P(i - 1, AKZ_LT);
        float* half_buf = nullptr;
        if (half) {
            const uint32_t launches = fed_num_launches(c, n_tau, lv.w, lv.h, n);
            half_buf = launches == 0 ? A : (fed_dst(launches, 1, A, B) == A ? B : A);
            fed_in = half_buf;
        }
        // Large launches of levels that diffuse: preparation and the first (up to four) diffusion steps in ONE launch of
        // k_level_march (akz_march.hip) — Lt is read once for both, 4 B read + 12 (+4) B written per pixel instead of
        // 12 + 12 (+4); a new octave's 2x2 mean is materialised first.  Remaining steps follow in k_fed_own launches.
        // (from 4 Mpx per launch -- the third octave of a 32-frame 1080p batch -- on: one launch less per level in the
        // coarse chain that runs next to the fine detectors, +1.0 % throughput, measured 4 x 80 steps each way)
        const bool fuse_level = takes_level_march(i);
        if (fuse_level) {
            const uint32_t n1 = std::min<uint32_t>(n_tau, 4u), rem = n_tau - n1;
            const uint32_t rest = rem ? fed_num_launches(c, rem, lv.w, lv.h, n) : 0;
            float* d1 = fed_dst(rest + 1, 1, A, B);
            const float* level_in = P(i - 1, AKZ_LT);
            // a new octave: the 2x2 mean of the previous Lt is formed inside the level kernel where the widths allow it
            // (one launch and one plane round trip less per octave), materialised first otherwise
            const bool fold_half = half && launch::level_march_half_supported(lv.w, lv.h, pv.w, pv.h, n1);
            if (half && !fold_half) {
                StageTimer st(c, AKZ_ST_PREP);
                float* hb = d1 == A ? B : A;
                launch::half_size(ls, P(i - 1, AKZ_LT), hb, pv.w, pv.h, n);
                level_in = hb;
            }
            float ht[4];
            for (uint32_t j = 0; j < n1; ++j) ht[j] = 0.5f * (float)lv.tau[j];
            {
                StageTimer st(c, AKZ_ST_FED);
                st.kernel(AKZ_KR_LEVEL_MARCH, n1 | (fold_half ? 16u : 0u) | ((rem == 0 && keep_all) ? 32u : 0u), lv.w, lv.h, n, 1,
                          (uint64_t)lv.w * lv.h * n, (uint64_t)lv.w * lv.h * n * n1);
                launch::level_march(ls, level_in, P(i, AKZ_LSMOOTH), P(i, AKZ_LFLOW), d1,
                                    (rem == 0 && keep_all) ? P(i, AKZ_LSTEP) : nullptr, lv.w, lv.h, n, g1.data(), r->d_k,
                                    lv.octave, ht, n1, fold_half ? pv.w : 0u, fold_half ? pv.h : 0u);
                if (c->profiling) {
                    c->prof.fed_launches += 1;
                    c->prof.fed_px_steps += (uint64_t)lv.w * lv.h * n * n1;
                    c->prof.fused_px += (uint64_t)lv.w * lv.h * n;
                }
            }
            if (rem) {  // (a span of its own: the rows of akz_debug_kernel_rows tell the two kernels apart)
                StageTimer st(c, AKZ_ST_FED);
                st.kernel(AKZ_KR_FED_OWN, rem, lv.w, lv.h, n, rest, 0, (uint64_t)lv.w * lv.h * n * rem);
                AKZ_TRY(fed_impl(c, d1, A, B, P(i, AKZ_LFLOW), keep_all ? P(i, AKZ_LSTEP) : nullptr, lv.w, lv.h, n,
                                 lv.tau.data() + n1, rem));
            }
            AKZ_HIP_TRY(hipGetLastError());
            continue;
        }
        // (Preparation + the first eight diffusion steps as ONE tiled launch -- k_prep and k_fed_own fused, tile + halo 8 + 2 --
        // was built and measured in round 3: 20 us per launch at best against 6-8 + 8-10 for the pair (the preparation then runs
        // on the whole diffusion region, 2.3 x the tile); a lone 1080p frame 0.59 -> 0.86 ms, batches -1 ... -4 %.  Removed.)
        if (i == 1 && head_level1) {  // Lsmooth is k_head's; Lflow = pm_g2 of the Scharr pair k_head left in level 0's Lx / Ly
            StageTimer st(c, AKZ_ST_PREP);
            launch::flow_from_pair(ls, P(0, AKZ_LX), P(0, AKZ_LY), P(1, AKZ_LFLOW), lv.w, lv.h, n, r->d_k, lv.octave);
        } else if (!prepared[i]) {
            StageTimer st(c, AKZ_ST_PREP);
            // measured on MI355X: the streaming kernel is ~2x faster for cloned levels of a batch (a single
            // frame is launch-latency bound and stays on the tiled kernel); for the first
            // level of an octave (2x2 mean of a 4x larger input) the two are equal, the tiled one stays
            const bool stream_prep = takes_stream_prep(i);
            if (stream_prep)
                launch::prep_stream(ls, P(i - 1, AKZ_LT), half, half_buf, P(i, AKZ_LSMOOTH), P(i, AKZ_LFLOW), lv.w, lv.h,
                                    pv.w, pv.h, n, g1.data(), r->d_k, lv.octave);
            else
                launch::prep_fused(ls, P(i - 1, AKZ_LT), half, half_buf, P(i, AKZ_LSMOOTH), P(i, AKZ_LFLOW), lv.w, lv.h,
                                   pv.w, pv.h, n, g1.data(), r->d_k, lv.octave);
        }
        if (keep_all && n_tau == 0) AKZ_HIP_TRY(hipMemsetAsync(P(i, AKZ_LSTEP), 0, plane_bytes(lv.w, lv.h, n), ls));
        {
            // The next level of the octave starts from this level's final Lt: where it would take the tiled preparation
            // (k_prep), the last diffusion launch of this level writes its Lsmooth and Lflow as well -- one dependent launch
            // less per level of a lone frame's chain (sched[6] = 1: a launch of its own, as before)
            launch::FedNextPrep np{};
            const bool ride = c->sched[6] == 0 && c->fed_mode == 2 && i + 1 < L && i + 1 != res_first && plan[i + 1].octave == lv.octave &&
                              !takes_level_march(i + 1) && !takes_stream_prep(i + 1);
            if (ride) np = launch::FedNextPrep{P(i + 1, AKZ_LSMOOTH), P(i + 1, AKZ_LFLOW), g1.data(), r->d_k, plan[i + 1].octave};
            bool rode = false;
            StageTimer st(c, AKZ_ST_FED);
            st.kernel(AKZ_KR_FED_OWN, n_tau, lv.w, lv.h, n, fed_num_launches(c, n_tau, lv.w, lv.h, n), 0, (uint64_t)lv.w * lv.h * n * n_tau);
            AKZ_TRY(fed_impl(c, fed_in, A, B, P(i, AKZ_LFLOW), keep_all ? P(i, AKZ_LSTEP) : nullptr, lv.w, lv.h, n,
                             lv.tau.data(), n_tau, ride ? &np : nullptr, &rode));
            prepared[i + 1] = rode ? 1 : 0;
        }
    }
    return AKZ_OK;
    };

    // ---- detectors: levels [lo, hi) on stream st (c->stream is st while this runs) ----
    // levels whose detector is the one-kernel tiled form are grouped by sigma_size: one launch per group
    auto detectors = [&](size_t lo, size_t hi, hipStream_t st_) -> int {
        std::map<uint32_t, std::vector<launch::DetLevelDesc>> sets;
        for (size_t l = lo; l < hi; ++l) {
            if (det_done[l]) continue;  // (sched[5]: enqueued behind its level kernel already)
            const LevelPlan& lv = plan[l];
            const float thr = (float)cfg.detector_threshold, bm = border_margin(lv, cfg);
            if (detector_family(c, lv.det_sigma, lv.w, lv.h, n, bm, keep_all) == 4) {
                sets[lv.det_sigma].push_back(launch::DetLevelDesc{P(l, AKZ_LSMOOTH), P(l, AKZ_LX), P(l, AKZ_LY), P(l, AKZ_LXX),
                                                                 P(l, AKZ_LYY), P(l, AKZ_LXY), P(l, AKZ_LDET), lv.w, lv.h,
                                                                 (uint32_t)l, bm});
                continue;
            }
            if (detector_one_pass(l, st_)) continue;
            {
                StageTimer st(c, AKZ_ST_DETECTOR);
                AKZ_TRY(detector_impl(c, P(l, AKZ_LSMOOTH), lv.det_sigma, P(l, AKZ_LX), P(l, AKZ_LY), P(l, AKZ_LXX),
                                      P(l, AKZ_LYY), P(l, AKZ_LXY), P(l, AKZ_LDET), lv.w, lv.h, n));
            }
            StageTimer st(c, AKZ_ST_NMS);
            launch::nms(st_, P(l, AKZ_LDET), lv.w, lv.h, n, (uint64_t)lv.w * lv.h, (uint32_t)l, thr, bm, d_cand, cap,
                        d_count);
        }
        for (auto& kv : sets) {
            const uint32_t maxn = launch::detector_tiled_set_max();
            for (size_t i = 0; i < kv.second.size(); i += maxn) {
                StageTimer st(c, AKZ_ST_DETECTOR, st_);
                if (c->profiling) {
                    c->prof.det_launches += 1;
                    uint64_t set_px = 0;
                    for (size_t j = i; j < std::min(kv.second.size(), i + maxn); ++j) set_px += (uint64_t)kv.second[j].w * kv.second[j].h * n;
                    c->prof.det_px += set_px;
                    // one launch over several levels of one sigma_size: the row carries the largest level's shape
                    st.kernel(AKZ_KR_DETECTOR_TILED, kv.first, kv.second[i].w, kv.second[i].h, n, 1, set_px);
                }
                launch::detector_tiled_set(st_, kv.first, kv.second.data() + i, (uint32_t)std::min<size_t>(maxn, kv.second.size() - i),
                                           n, (float)cfg.detector_threshold, d_cand, cap, d_count);
            }
        }
        return AKZ_OK;
    };
    // (A small job's first-octave detectors on a second stream, under the remaining octaves' chain of small launches, was built
    // and measured three times in round 6.  With the persistent detector launches: lone 1080p call 0.795 / 0.799 ms with /
    // without, 720p 0.634 / 0.620 -- persistent workgroups hold their compute units until the launch ends and the chain's
    // launches wait for places.  With one tile per workgroup on a lowest-priority stream the two do run side by side -- under
    // rocprofv3 the begin chain ends 40 us earlier -- but unprofiled, where the chain's launches follow each other without the
    // profiler's gaps, there is nothing to fill: 0.745-0.751 / 0.749-0.750 ms, 720p 0.588 / 0.576.  On a stream with a CU mask
    // (hipExtStreamCreateWithCUMask: 64 / 128 / 192 of the 256 compute units): 0.730 / 0.690 / 0.670 against 0.682 without, a
    // stream of frames 0.57 / 0.54 / 0.514 against 0.519, 720p 0.54 against 0.53.  Not kept.)
    // the fine levels (all levels when the batch does not fork) on the main stream
    {
        size_t oct1 = 1;  // first level past the first octave (fork_level if there is none on the main stream)
        while (oct1 < fork_level && plan[oct1].octave == plan[0].octave) ++oct1;
        AKZ_TRY(run_levels(1, oct1));
        AKZ_HIP_TRY(hipEventRecord(c->pre_ev[seq % akz_ctx::kFedRing], s));
        AKZ_TRY(run_levels(oct1, fork_level));
    }
    // The keypoint kernels of the batch that is finished next (orientation, M-LDB: gather-bound, on the auxiliary
    // stream) wait for this point: next to the VALU-bound diffusion launches they cost more than next to the
    // bandwidth-bound detector launches that follow, and the diffusion launches stay individually timeable.
    AKZ_HIP_TRY(hipEventRecord(c->fed_ev[seq % akz_ctx::kFedRing], s));
    c->begin_seq.store(seq);

    hipStream_t done_on = s;  // the stream behind whose work the batch's candidate list is complete
    if (fork_level < L) {
        // The coarse chain (levels from fork_level on, then their detectors) runs on the second stream; the main stream
        // takes the fine detectors.  The JOIN is on the coarse stream: it waits for the fine detectors and records the
        // batch's completion, and the main stream goes straight on to the next batch.  (Joined on the main stream, that
        // stream sat idle for 0.35-0.5 ms per 32-frame step: next to the bandwidth-bound fine detectors the coarse
        // chain's small launches are starved -- HBM latency grows several-fold -- and finish well after them.  Now that
        // tail runs under the next batch's level-0 kernels; the chain has its own diffusion scratch, and the chains of
        // consecutive batches follow each other on one stream.)
        bool own_kernels = true;  // the multi-kernel detector fallback borrows context scratch planes: then join on the main stream
        for (size_t l = fork_level; l < L; ++l) {
            const int fam = detector_family(c, plan[l].det_sigma, plan[l].w, plan[l].h, n, border_margin(plan[l], cfg), keep_all);
            own_kernels = own_kernels && (fam != 0 || launch::detector_nms_fused_supported(plan[l].det_sigma));
        }
        if (!c->coarse) AKZ_HIP_TRY(hipStreamCreateWithFlags(&c->coarse, hipStreamNonBlocking));
        hipEvent_t fine_done = StageTimer::get(c);
        AKZ_HIP_TRY(hipEventRecord(fine_done, s));
        AKZ_HIP_TRY(hipStreamWaitEvent(c->coarse, fine_done, 0));
        ev_put(c, fine_done);
        // (holding the chain back until the full-resolution detectors, or all fine detectors, have finished: -2 ... -5 %)
        // The fine detectors are ENQUEUED first: the coarse chain is dozens of small launches, and a caller that is not
        // ahead of the chip -- one synchronous call on a 4K pair -- kept the main stream idle for the 0.19 ms it took to
        // enqueue them (the two streams run side by side either way).
        AKZ_TRY(detectors(0, fork_level, s));
        ls = c->coarse;
        c->stream = c->coarse;
        AKZ_TRY(run_levels(fork_level, L));
        AKZ_TRY(detectors(fork_level, L, c->coarse));
        c->stream = s;
        hipEvent_t ev = StageTimer::get(c);
        if (own_kernels) {
            AKZ_HIP_TRY(hipEventRecord(ev, s));
            AKZ_HIP_TRY(hipStreamWaitEvent(c->coarse, ev, 0));
            done_on = c->coarse;
        } else {
            AKZ_HIP_TRY(hipEventRecord(ev, c->coarse));
            AKZ_HIP_TRY(hipStreamWaitEvent(s, ev, 0));
        }
        ev_put(c, ev);
    } else {
        AKZ_TRY(detectors(0, L, s));
    }
    AKZ_HIP_TRY(hipGetLastError());
    job->nms_done = StageTimer::get(c);
    AKZ_HIP_TRY(hipEventRecord(job->nms_done, done_on));
    job->done_stream = done_on;
    job->slot = slot;
    job->cap = cap;
    job->seq = seq;
    c->slot_busy[slot] = true;
    guard.armed = false;
    *out = job.release();
    return AKZ_OK;
}

// Can the device stand in for this process's libm (akz_libm.hpp)?  Decided once per process: the build of sinf / cosf that
// glibc's selector takes on this CPU (FMA + AVX2: the FMA build; FMA4: a third build the header does not state; otherwise
// SSE2), then every bit of atan2f, cosf, sinf on kN arguments each -- orientation-like angles, every binade of small and
// large ratios, zeros, infinities, NaNs, the reduction's breakpoints -- against the host's functions.
int device_libm_mode(akz_ctx* c) {
    if (c->dbg_device_libm == 0) return 0;
    static std::mutex m;
    static int mode = -1;
    std::lock_guard<std::mutex> lk(m);
    if (mode >= 0) return mode;
    mode = 0;
#if defined(__x86_64__) && defined(__GLIBC__)
    __builtin_cpu_init();
    const bool fma = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
    if (!fma && __builtin_cpu_supports("fma4")) return mode;
    constexpr size_t kN = 1u << 21;
    std::vector<float> a(kN), b(kN);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {  // SplitMix64
        uint64_t z = (st += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 0x1p-126f, -0x1p-126f, 0x1p-149f, 0x1.fffffep127f, 0x1p-12f,
                             0x1.921fb6p-1f, 0x1.921fb6p0f, 0x1.921fb6p1f, -0x1.921fb6p1f, 0.4375f, 0.6875f, 1.1875f, 2.4375f, 0x1p25f, 0x1p-29f,
                             119.99f, 120.0f, 1e10f};
    const size_t n_sp = sizeof(special) / sizeof(special[0]);
    for (size_t i = 0; i < kN; ++i) {
        const uint64_t r = next(), q = next();
        if (i < n_sp * n_sp) {
            a[i] = special[i / n_sp];
            b[i] = special[i % n_sp];
        } else if (i & 1) {  // an orientation-like angle and a ratio near it: |a| <= pi (+ a little), b of any magnitude
            a[i] = (float)((double)(int64_t)(r >> 11) * (1.0 / 9007199254740992.0) * 6.6 - 3.3);
            uint32_t u = (uint32_t)q;
            u = (u & 0x807fffffu) | ((64u + (u >> 23) % 128u) << 23);  // exponents 2^-63 .. 2^64
            std::memcpy(&b[i], &u, 4);
        } else {  // both of any magnitude: every reduction branch of atanf, the k > 60 / k < -60 shortcuts
            uint32_t u = (uint32_t)r, v = (uint32_t)q;
            u = (u & 0x807fffffu) | ((32u + (u >> 23) % 192u) << 23);
            v = (v & 0x807fffffu) | ((32u + (v >> 23) % 192u) << 23);
            if ((i & 6) == 2) v = (v & 0x80000000u) | (u & 0x7f800000u) | (v & 0x007fffffu);  // comparable magnitudes
            std::memcpy(&a[i], &u, 4);
            std::memcpy(&b[i], &v, 4);
        }
    }
    float *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    std::vector<float> got(3 * kN);
    const bool ok_dev = hipMalloc((void**)&d_a, kN * 4) == hipSuccess && hipMalloc((void**)&d_b, kN * 4) == hipSuccess &&
                        hipMalloc((void**)&d_o, 3 * kN * 4) == hipSuccess &&
                        hipMemcpy(d_a, a.data(), kN * 4, hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(d_b, b.data(), kN * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok_dev) {
        launch::libm_eval(nullptr, d_a, d_b, d_o, kN, fma, nullptr);
        const bool ran = hipGetLastError() == hipSuccess && hipMemcpy(got.data(), d_o, 3 * kN * 4, hipMemcpyDeviceToHost) == hipSuccess;
        std::atomic<uint64_t> bad{ran ? 0u : 1u};
        if (ran) {
            const size_t pieces = 64;
            c->pool().run(pieces, [&](size_t p) {
                uint64_t mine = 0;
                auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
                auto same = [&](float x, float y) { return bits(x) == bits(y) || (x != x && y != y); };
                for (size_t i = kN * p / pieces; i < kN * (p + 1) / pieces; ++i) {
                    mine += !same(got[3 * i], atan2f(a[i], b[i]));
                    if (std::fabs(a[i]) < 119.9f) mine += !same(got[3 * i + 1], cosf(a[i])) + !same(got[3 * i + 2], sinf(a[i]));
                }
                bad += mine;
            });
        }
        if (bad.load() == 0) mode = fma ? 1 : 2;
    }
    (void)hipGetLastError();
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (d_o) (void)hipFree(d_o);
#endif
    return mode;
}

// The finish half proper.  The job shell stays with the caller; on failure everything the job held is released.
static int extract_finish_body(akz_job* jobp, akz_result** out) {
    *out = nullptr;
    std::unique_ptr<akz_job, void (*)(akz_job*)> job(jobp, job_release);
    akz_result* r = job->r.get();
    akz_ctx* c = r->ctx;
    AKZ_TRY(bind(c, false));
    // everything below waits only for THIS job's kernels: on the context's auxiliary stream, behind the job's event.  An
    // eagerly finished job of a lane stays on the lane's own stream (its finish half is enqueued right behind its begin
    // half): the streams of a process share a few hardware queues, each of which runs its packets in order, so every
    // further stream of a lane queues its keypoint kernels behind the launch chain of some other lane
    hipStream_t s = c->main;  // (never c->stream: extract_begin may be swapping it on the caller's thread right now)
    // ... and the job that is alone in the context's hands takes the stream its chain ended on: its keypoint kernels follow the
    // last detector in order, without a dependency between two hardware queues (lone 1080p call 0.721 -> 0.694 ms)
    const bool in_order = !c->is_lane && job->done_stream && job->alone_at_begin && job->in_hand && job->in_hand->load() == 1 && (c->sched[9] & 2) == 0;
    if (in_order) {
        s = job->done_stream;
    } else if (!(c->is_lane && job->fin)) {
        AKZ_TRY(ensure_aux(c));
        s = c->aux;
    }
    const akz_config& cfg = r->cfg;
    const std::vector<LevelPlan>& plan = r->plan;
    const size_t L = plan.size();
    const uint32_t n = r->n;
    auto P = [&](size_t l, int p) { return r->planes[l][p]; };
    AKZ_HIP_TRY(hipStreamWaitEvent(s, job->nms_done, 0));
    const double t_fetch0 = now_ms();
    double t_counts = t_fetch0;

    // ---- candidates: sorted into scan order on the device, then the list length and exactly the used part ----
    uint32_t cap = job->cap;
    uint32_t* d_count = (uint32_t*)c->count_slot[job->slot].p;
    const uint64_t max_px = (uint64_t)plan[0].w * plan[0].h;
    // Where the list is put into scan order: with four or more host threads bucketing + sorting are 0.15 ms of wall time
    // per 32-frame batch and the dozen small launches of the device sort cost the kernels of the next batch more
    // (-1.5 % throughput); with the two threads a rank has on a node whose cores are shared by eight ranks they are a
    // fifth of a host phase that no longer hides under the GPU step (+7 % throughput with the device sort).
    // The selection itself (round 4): with the list sorted on the device, the device also answers which candidates can be
    // within `size` of which (k_relations) and the host's order-dependent logic walks those short lists on three small
    // arrays instead of querying a spatial grid per image (select_keypoints_rel: the same answers, a quarter of the host
    // time).  Taken where the host is the bottleneck -- contexts with fewer than four host threads, and lone frames, whose
    // latency is the sum of the two halves -- and not for large batches on many-core hosts, where the extra small launches
    // cost the kernels of the next batch more than the idle host threads gain.
    // (... and for images of 6 Mpx and more, whose sequential selection -- 1.4 ms per 4K frame on the grids -- is the longest
    // single piece of a synchronous call)
    // (... and for the job that is WAITED FOR, and for jobs of fewer images than the host's selection needs threads: the host
    // walks one image per thread, which a stream hides and a call does not -- a lone 3-6 Mpx frame on the batch path spent
    // 0.5 ms on the grids, 1.93 against 1.29 ms per call either side of 6 Mpx; 4 x 1080p per call 2.07 against 1.75)
    // (a lane's jobs run next to the other lanes': part of a stream)
    const bool waited_for = !c->is_lane && job->alone_at_begin;
    const bool want_rel = c->dbg_select == 1 || (c->dbg_select < 0 && (c->pool().size() < gates::kFewHostThreads || (uint64_t)r->w * r->h * n < r->big_px ||
                                                                        (uint64_t)r->w * r->h >= gates::kSelectDevicePx || n < gates::kFewHostThreads || waited_for));
    // The selection ITSELF on the device (round 5; akz_select.hpp, launch::select_device): the order-dependent walk as
    // dependency rounds over the same neighbour lists, one workgroup per image, and the orientation sums right behind it on
    // the keypoints it leaves -- the host neither fetches the candidate list nor selects, and one of the two round trips of
    // the finish half is gone.  Taken where the neighbour lists are (the host's selection is the longest piece of a lone
    // image's call); an image whose lists overflowed sends the job down the host's path below.
    // ... and where the call waits for it: a job begun while another one was still in the caller's hand is part of a
    // stream, whose rate the chip bounds, not the host -- and k_select's workgroup wants nearly all of a compute unit's LDS, which a chip busy with the
    // next job's kernels frees only at their ends (pairs of 4K frames streamed: 2.9 ms per pair with the host's selection,
    // 4.4 with the device's; the synchronous pair 4.3 -> 3.7 ms, a lone 4K frame 2.06 -> 1.81 ms, a lone 1080p frame
    // 0.98 -> 0.95 ms)
    // (a job whose lists overflowed went back to the host's selection after the device's attempt: the next eight jobs of that
    // shape do not try -- dense texture stays dense)
    const uint64_t shape = ((uint64_t)r->w << 40) | ((uint64_t)r->h << 16) | n;
    bool skip_dev = false;
    if (c->sel_skip.load() > 0 && c->sel_skip_shape.load() == shape) {
        skip_dev = true;
        --c->sel_skip;
    }
    const bool want_dev = c->dbg_select == 2 || (c->dbg_select < 0 && want_rel && !skip_dev && (c->pool().size() < gates::kFewHostThreads || waited_for));
    if (!want_dev) {  // (akz_debug_select_info: nothing was tried on the device)
        c->sel_last_rounds = 0;
        c->sel_last_fallback = 0;
    }
    bool sorted = false, dev_sel = false;
    uint32_t mldb_spec = 0;  // keypoints the device's own descriptor launch covered
    int libm_dev = 0;        // 1 / 2: angles and descriptors were enqueued on the device behind the selection (device_libm_mode)
    bool dev_angles_ok = true;  // ... and every angle was one the device forms cover
    bool mirrored = false;      // ... and that launch stored the host's copies itself (headers, records, descriptor rows)
    uint16_t* d_rel = nullptr;
    uint32_t* d_rel_flags = nullptr;
    uint32_t *d_sel_hdr = nullptr, *d_sel_total = nullptr;
    std::vector<float> lsize, lratio;
    selection_level_constants(plan, cfg, lsize, lratio);
    LevelTable tab;
    std::memset(&tab, 0, sizeof(tab));
    for (size_t l = 0; l < L; ++l) {
        tab.lv[l].lt = P(l, AKZ_LT);
        tab.lv[l].lx = P(l, AKZ_LX);
        tab.lv[l].ly = P(l, AKZ_LY);
        tab.lv[l].w = plan[l].w;
        tab.lv[l].h = plan[l].h;
        tab.lv[l].stride = (uint64_t)plan[l].w * plan[l].h;
    }
    unsigned long long wmask = 0;
    uint32_t nwin = 0;
    orientation_windows(&wmask, &nwin);
    if (c->dbg_host_sort == 0 || (c->dbg_host_sort < 0 && (c->pool().size() < gates::kFewHostThreads || want_rel)) || want_dev) {
        AKZ_TRY(ensure(c, c->cand_sorted, (size_t)cap * sizeof(Candidate)));
        void* selp = nullptr;
        if (want_dev) {
            AKZ_TRY(ensure(c, c->sel_scratch, launch::select_device_bytes(cap, n)));
            selp = c->sel_scratch.p;
        }
        std::vector<uint32_t> lw(L), lh(L);
        for (size_t l = 0; l < L; ++l) {
            lw[l] = plan[l].w;
            lh[l] = plan[l].h;
        }
        uint32_t* d_zero = selp ? launch::select_device_revcnt(selp, cap, n) : nullptr;
        const bool want_lists = want_rel || want_dev;
        if (want_lists) AKZ_TRY(ensure(c, c->rel_scratch, launch::candidate_relations_bytes(cap, lh.data(), (uint32_t)L, n)));
        // (the one-launch sort is one workgroup with 112 KB of LDS: as k_select, for the job that is waited for)
        const bool rows_sort = n == 1 && (waited_for || c->dbg_select == 2) &&
                               launch::sort_candidates_rows(s, (const Candidate*)c->cand_slot[job->slot].p, cap, d_count, lw.data(), lh.data(), (uint32_t)L,
                                                            (Candidate*)c->cand_sorted.p, d_zero, want_lists ? c->rel_scratch.p : nullptr);
        sorted = rows_sort;
        bool buckets_sort = false;
        if (!sorted) {  // several images or a longer list: the same sort in four launches, if the job has few enough rows
            const size_t before = c->bucket_scratch.bytes;  // (a buffer that has just been (re)allocated: its counters are not zero yet)
            AKZ_TRY(ensure(c, c->bucket_scratch, launch::sort_candidates_buckets_scratch(cap)));
            buckets_sort = launch::sort_candidates_buckets(s, (const Candidate*)c->cand_slot[job->slot].p, cap, d_count, lw.data(), lh.data(), (uint32_t)L,
                                                           n, c->bucket_scratch.p, c->bucket_scratch.bytes != before, (Candidate*)c->cand_sorted.p, d_zero,
                                                           want_lists ? c->rel_scratch.p : nullptr);
            sorted = buckets_sort;
        }
        if (!sorted) {
            AKZ_TRY(ensure(c, c->sort_scratch, launch::sort_candidates_scratch(cap, max_px, (uint32_t)L, n)));
            sorted = launch::sort_candidates_device(s, (const Candidate*)c->cand_slot[job->slot].p, cap, d_count, max_px, (uint32_t)L,
                                                    n, c->sort_scratch.p, (Candidate*)c->cand_sorted.p, d_zero);
        }
        AKZ_HIP_TRY(hipGetLastError());
        if (sorted && (want_rel || want_dev)) {
            launch::candidate_relations(s, (const Candidate*)c->cand_sorted.p, cap, d_count, lsize.data(), lratio.data(), lw.data(), lh.data(),
                                        (uint32_t)L, n, c->rel_scratch.p, &d_rel, &d_rel_flags, selp, rows_sort || buckets_sort);
            AKZ_HIP_TRY(hipGetLastError());
            if (want_dev) {
                AKZ_TRY(ensure(c, c->sel_recs, (size_t)cap * sizeof(SelKpHost)));
                AKZ_TRY(ensure(c, c->kp_in, (size_t)cap * sizeof(KpParam)));
                launch::select_device(s, (const Candidate*)c->cand_sorted.p, cap, d_count, lsize.data(), lratio.data(), lw.data(), lh.data(),
                                      (uint32_t)L, n, c->rel_scratch.p, selp, r->d_k, c->sel_recs.p, (KpParam*)c->kp_in.p, &d_sel_hdr, &d_sel_total);
                if ((uint64_t)r->w * r->h * n >= r->big_px && c->begin_seq.load() > job->seq)  // (as the host path's orientation below)
                    AKZ_HIP_TRY(hipStreamWaitEvent(s, c->fed_ev[(job->seq + 1) % akz_ctx::kFedRing], 0));
                launch::orientation_counted(s, tab, (const KpParam*)c->kp_in.p, d_sel_total, cap, wmask, nwin, &((SelKpHost*)c->sel_recs.p)->sums, 2);
                AKZ_HIP_TRY(hipGetLastError());
                dev_sel = true;
                // ... and, where the device reproduces this process's atan2f / cosf / sinf (akz_libm.hpp), the descriptors right
                // behind the orientation sums: no host round trip between the two.  Speculative like the selection itself: a job
                // that goes back to the host's selection, or an angle outside what the device forms cover, discards them.
                libm_dev = device_libm_mode(c);
                if (libm_dev) {
                    void* blk = nullptr;
                    AKZ_TRY(slab_acquire(c, (size_t)cap * 64, &blk, &r->desc_block_bytes));
                    r->d_desc64 = (uint8_t*)blk;
                    // (the grid for as many keypoints as the last job had + 25 %, like the speculative fetch below: a list's capacity
                    // is 3-15 x its keypoints, and 8 000 workgroups that find nothing to do cost 15 us; the rest, should there be
                    // one, follows when the count is known)
                    mldb_spec = std::min<uint32_t>(cap, c->last_total_kp.load() + c->last_total_kp.load() / 4 + 256u);
                    // ... and what the host wants of them -- headers, keypoint records, descriptor rows -- stored into its pinned
                    // buffers by the same launch (k_mldb's host mirror; sched[9] = 1: three copies behind the kernel, as before)
                    launch::MldbMirror mir{};
                    if ((c->sched[9] & 1) == 0) {
                        const bool host_rows = !(r->flags & AKZ_NO_HOST_DESCRIPTORS);
                        AKZ_TRY(ensure_pinned(c, c->pin[8], (size_t)n * 64));
                        AKZ_TRY(ensure_pinned(c, c->pin[0], (size_t)mldb_spec * sizeof(SelKpHost)));
                        if (host_rows) AKZ_TRY(ensure_pinned(c, c->pin[2], (size_t)mldb_spec * 64));
                        mir = launch::MldbMirror{c->pin[0].p, d_sel_hdr, c->pin[8].p, n * 64u, host_rows ? (uint8_t*)c->pin[2].p : nullptr};
                        mirrored = true;
                    }
                    launch::mldb_counted(s, tab, (const KpParam*)c->kp_in.p, d_sel_total, 0, mldb_spec, &((SelKpHost*)c->sel_recs.p)->sums, 2,
                                         libm_dev == 1, nullptr, (uint32_t)cfg.descriptor_channels, r->d_desc64, mirrored ? &mir : nullptr);
                    AKZ_HIP_TRY(hipGetLastError());
                }
            }
        }
    }
    constexpr size_t kRelRow = (size_t)(kRel1 + kRel2) * sizeof(uint16_t);
    const Candidate* hc = nullptr;   // the whole list on the host (pinned)
    uint32_t total_c = 0, spec_kp = 0;
    uint64_t total_kp = 0;
    for (int attempt = 0;; ++attempt) {
        const Candidate* d_list = sorted ? (const Candidate*)c->cand_sorted.p : (const Candidate*)c->cand_slot[job->slot].p;
        AKZ_TRY(ensure_pinned(c, c->pin[1], 256));
        uint32_t* total_p = (uint32_t*)c->pin[1].p;
        const bool dev_fetch = attempt == 0 && dev_sel;  // (the device's selection brings the count with its headers)
        if (!dev_fetch) AKZ_HIP_TRY(hipMemcpyAsync(total_p, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        // Small jobs are bound by the latency of these round trips: fetch, with the count, as many candidates as the
        // previous job of this context had (+25 %) and the contrast factors, so that one synchronisation serves all.
        uint32_t spec = 0;
        if (dev_fetch) {  // the device's selection: two copies -- the headers (with the list's length, the contrast factors and the
                          // images' flags) and as many keypoints with their orientation sums as the last job had (+25 %)
            const uint32_t last = c->last_total_kp.load();
            spec_kp = libm_dev ? mldb_spec : std::min<uint32_t>(cap, last + last / 4 + 256u);  // (with the device's descriptors: exactly the keypoints their launch covered)
            if (!mirrored) {
                AKZ_TRY(ensure_pinned(c, c->pin[8], (size_t)n * 64));
                AKZ_HIP_TRY(hipMemcpyAsync(c->pin[8].p, d_sel_hdr, (size_t)n * 64, hipMemcpyDeviceToHost, s));
                AKZ_TRY(ensure_pinned(c, c->pin[0], (size_t)spec_kp * sizeof(SelKpHost)));
                AKZ_HIP_TRY(hipMemcpyAsync(c->pin[0].p, c->sel_recs.p, (size_t)spec_kp * sizeof(SelKpHost), hipMemcpyDeviceToHost, s));
                if (libm_dev && !(r->flags & AKZ_NO_HOST_DESCRIPTORS)) {  // the descriptor rows of those keypoints with the same synchronisation
                    AKZ_TRY(ensure_pinned(c, c->pin[2], (size_t)spec_kp * 64));
                    AKZ_HIP_TRY(hipMemcpyAsync(c->pin[2].p, r->d_desc64, (size_t)spec_kp * 64, hipMemcpyDeviceToHost, s));
                }
            }  // (else: k_mldb's host mirror has stored all three)
            AKZ_TRY(ensure_pinned(c, c->pin[5], (size_t)n * sizeof(double)));
            AKZ_TRY(ensure_pinned(c, c->pin[7], (size_t)n * sizeof(uint32_t)));
        } else if (attempt == 0) {
            const uint32_t last = c->last_total_cands.load();
            spec = std::min<uint32_t>(cap, last + last / 4 + 64u);
            if ((size_t)spec * sizeof(Candidate) > (1u << 20)) spec = 0;  // large lists: exactly the used part, below
            AKZ_TRY(ensure_pinned(c, c->pin[5], (size_t)n * sizeof(double)));
            AKZ_HIP_TRY(hipMemcpyAsync(c->pin[5].p, r->d_k, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
            if (spec) {
                AKZ_TRY(ensure_pinned(c, c->pin[0], (size_t)spec * sizeof(Candidate)));
                AKZ_HIP_TRY(hipMemcpyAsync(c->pin[0].p, d_list, (size_t)spec * sizeof(Candidate), hipMemcpyDeviceToHost, s));
                if (d_rel) {
                    AKZ_TRY(ensure_pinned(c, c->pin[6], (size_t)spec * kRelRow));
                    AKZ_HIP_TRY(hipMemcpyAsync(c->pin[6].p, d_rel, (size_t)spec * kRelRow, hipMemcpyDeviceToHost, s));
                }
            }
            if (d_rel) {
                AKZ_TRY(ensure_pinned(c, c->pin[7], (size_t)n * sizeof(uint32_t)));
                AKZ_HIP_TRY(hipMemcpyAsync(c->pin[7].p, d_rel_flags, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            }
        }
        AKZ_HIP_TRY(hipStreamSynchronize(s));
        t_counts = now_ms();
        if (dev_fetch) {  // (into the places where the host's path looks for them, should the job go back to it)
            const uint32_t* hdr = (const uint32_t*)c->pin[8].p;
            *total_p = hdr[8];
            for (uint32_t img = 0; img < n; ++img) {
                ((uint32_t*)c->pin[7].p)[img] = hdr[img * 16 + 9];
                std::memcpy((double*)c->pin[5].p + img, &hdr[img * 16 + 10], sizeof(double));
            }
        }
        total_c = *total_p;
        if (attempt == 0) r->k_host.assign((const double*)c->pin[5].p, (const double*)c->pin[5].p + n);
        c->last_total_cands = total_c;
        c->last_cand_shape = ((uint64_t)r->w << 40) | ((uint64_t)r->h << 16) | n;
        c->cand_cap_hint = std::max(c->cand_cap_hint.load(), (uint32_t)((uint64_t)total_c * 5 / 4 / n) + 64u);
        if (total_c > cap) {  // overflow: grow and redo the NMS pass alone on the stored Ldet planes (the host sorts that list)
            if (attempt >= 3) {
                set_error("NMS candidate buffer overflow");
                return AKZ_ERR_OVERFLOW;
            }
            cap = total_c + total_c / 8;
            sorted = false;
            dev_sel = false;
            d_rel = nullptr;  // (the lists belong to the truncated list)
            AKZ_HIP_TRY(hipStreamSynchronize(c->main));
            AKZ_TRY(ensure(c, c->cand_slot[job->slot], (size_t)cap * sizeof(Candidate)));
            AKZ_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
            for (size_t l = 0; l < L; ++l)
                launch::nms(s, P(l, AKZ_LDET), plan[l].w, plan[l].h, n, (uint64_t)plan[l].w * plan[l].h, (uint32_t)l,
                            (float)cfg.detector_threshold, border_margin(plan[l], cfg),
                            (Candidate*)c->cand_slot[job->slot].p, cap, d_count);
            AKZ_HIP_TRY(hipGetLastError());
            continue;
        }
        if (dev_sel) {
            const uint32_t* hdr = (const uint32_t*)c->pin[8].p;
            uint32_t max_rounds = 0, fallen = 0;
            for (uint32_t img = 0; img < n; ++img) {
                if (hdr[img * 16 + 2] != 0) {  // an image for the host's selection: the whole job takes it (its lists are fetched below)
                    dev_sel = false;
                    ++fallen;
                    fallen |= hdr[img * 16 + 2] << 16;
                }
                total_kp += hdr[img * 16];
                max_rounds = std::max(max_rounds, hdr[img * 16 + 3]);
            }
            for (int k = 0; k < 4; ++k) c->sel_last_ticks[k] = hdr[4 + k];  // (image 0's phases)
            c->sel_last_rounds = max_rounds;
            c->sel_last_fallback = fallen;
            if (fallen) {
                c->sel_skip = 8;
                c->sel_skip_shape = shape;
            }
            if (dev_sel && libm_dev && total_kp > mldb_spec) {  // ... than the device's descriptor launch covered: the rest of them
                launch::mldb_counted(s, tab, (const KpParam*)c->kp_in.p, d_sel_total, mldb_spec, (uint32_t)total_kp, &((SelKpHost*)c->sel_recs.p)->sums, 2,
                                     libm_dev == 1, nullptr, (uint32_t)cfg.descriptor_channels, r->d_desc64);
                AKZ_HIP_TRY(hipGetLastError());
            }
            if (dev_sel && total_kp > spec_kp) {  // more keypoints than last time: the rest
                std::vector<uint8_t> keep_r((const uint8_t*)c->pin[0].p, (const uint8_t*)c->pin[0].p + (size_t)spec_kp * sizeof(SelKpHost));
                AKZ_TRY(ensure_pinned(c, c->pin[0], (size_t)total_kp * sizeof(SelKpHost)));
                std::memcpy(c->pin[0].p, keep_r.data(), keep_r.size());
                AKZ_HIP_TRY(hipMemcpyAsync((SelKpHost*)c->pin[0].p + spec_kp, (const SelKpHost*)c->sel_recs.p + spec_kp,
                                           (size_t)(total_kp - spec_kp) * sizeof(SelKpHost), hipMemcpyDeviceToHost, s));
                if (libm_dev && !(r->flags & AKZ_NO_HOST_DESCRIPTORS)) {
                    std::vector<uint8_t> keep_d((const uint8_t*)c->pin[2].p, (const uint8_t*)c->pin[2].p + (size_t)spec_kp * 64);
                    AKZ_TRY(ensure_pinned(c, c->pin[2], (size_t)total_kp * 64));
                    std::memcpy(c->pin[2].p, keep_d.data(), keep_d.size());
                    AKZ_HIP_TRY(hipMemcpyAsync((uint8_t*)c->pin[2].p + (size_t)spec_kp * 64, r->d_desc64 + (size_t)spec_kp * 64,
                                               (size_t)(total_kp - spec_kp) * 64, hipMemcpyDeviceToHost, s));
                }
                AKZ_HIP_TRY(hipStreamSynchronize(s));
            }
            if (dev_sel) break;
            total_kp = 0;
        }
        const uint32_t have = std::min(spec, total_c);  // already on the host
        if (total_c > have) {
            std::vector<Candidate> keep;
            if (have) keep.assign((const Candidate*)c->pin[0].p, (const Candidate*)c->pin[0].p + have);  // ensure_pinned may move the buffer
            AKZ_TRY(ensure_pinned(c, c->pin[0], (size_t)total_c * sizeof(Candidate)));
            if (have) std::memcpy(c->pin[0].p, keep.data(), (size_t)have * sizeof(Candidate));
            AKZ_HIP_TRY(hipMemcpyAsync((Candidate*)c->pin[0].p + have, d_list + have, (size_t)(total_c - have) * sizeof(Candidate),
                                       hipMemcpyDeviceToHost, s));
            if (d_rel) {
                std::vector<uint8_t> keep_rel;
                if (have) keep_rel.assign((const uint8_t*)c->pin[6].p, (const uint8_t*)c->pin[6].p + (size_t)have * kRelRow);
                AKZ_TRY(ensure_pinned(c, c->pin[6], (size_t)total_c * kRelRow));
                if (have) std::memcpy(c->pin[6].p, keep_rel.data(), keep_rel.size());
                AKZ_HIP_TRY(hipMemcpyAsync((uint8_t*)c->pin[6].p + (size_t)have * kRelRow, (const uint8_t*)d_rel + (size_t)have * kRelRow,
                                           (size_t)(total_c - have) * kRelRow, hipMemcpyDeviceToHost, s));
            }
            AKZ_HIP_TRY(hipStreamSynchronize(s));
        }
        AKZ_TRY(ensure_pinned(c, c->pin[0], sizeof(Candidate)));
        hc = (const Candidate*)c->pin[0].p;
        break;
    }
    // per image: a range of the sorted list, or (host fallback) a bucket of the unordered one, sorted below
    std::vector<std::vector<Candidate>> cands;
    std::vector<std::pair<const Candidate*, size_t>> span(n, {nullptr, 0});
    if (dev_sel) {
        // (nothing of the candidate list is on the host)
    } else if (sorted) {
        size_t at = 0;
        for (uint32_t img = 0; img < n; ++img) {
            const Candidate* b = hc + at;
            const Candidate* e = std::partition_point(b, hc + total_c, [&](const Candidate& x) { return x.img <= img; });
            span[img] = {b, (size_t)(e - b)};
            at += (size_t)(e - b);
        }
    } else {
        cands.resize(n);
        // bucket the unordered list by image: slices of the list are counted and scattered by separate threads
        const unsigned slices = (unsigned)std::min<size_t>((size_t)total_c / 16384 + 1, (size_t)c->pool().size());
        std::vector<std::vector<uint32_t>> at(slices, std::vector<uint32_t>(n, 0));  // counts, then write offsets
        auto slice_range = [&](unsigned t, size_t* b, size_t* e) {
            *b = (size_t)total_c * t / slices;
            *e = (size_t)total_c * (t + 1) / slices;
        };
        auto run_slices = [&](auto&& fn) { c->pool().run(slices, [&](size_t t) { fn((unsigned)t); }); };
        run_slices([&](unsigned t) {
            size_t b, e;
            slice_range(t, &b, &e);
            for (size_t i = b; i < e; ++i)
                if (hc[i].img < n) at[t][hc[i].img]++;
        });
        for (uint32_t img = 0; img < n; ++img) {
            uint32_t run = 0;
            for (unsigned t = 0; t < slices; ++t) {
                const uint32_t cnt = at[t][img];
                at[t][img] = run;
                run += cnt;
            }
            cands[img].resize(run);
        }
        run_slices([&](unsigned t) {
            size_t b, e;
            slice_range(t, &b, &e);
            for (size_t i = b; i < e; ++i)
                if (hc[i].img < n) cands[hc[i].img][at[t][hc[i].img]++] = hc[i];
        });
    }
    c->sel_last_mode = dev_sel ? 2 : (d_rel && sorted ? 1 : 0);
    c->slot_busy[job->slot] = false;  // the candidate buffers may be reused by the next begin
    job->slot = -1;
    if (c->profiling) c->prof.ms[AKZ_ST_NMS] += now_ms() - t_counts;  // candidate D2H after the counts arrived
    (void)t_fetch0;

    // ---- host: raster order, sequential cache logic, refinement ----
    const double t_host0 = now_ms();
    std::vector<std::vector<HostKeypoint>> hk(n);
    r->n_extrema.assign(n, 0);
    r->desc_off.assign(n + 1, 0);
    if (dev_sel) {  // the device's selection: the records into the host's form (size and octave follow from the level)
        const uint32_t* hdr = (const uint32_t*)c->pin[8].p;
        const SelKpHost* recs = (const SelKpHost*)c->pin[0].p;
        std::vector<uint64_t> first(n + 1, 0);
        for (uint32_t img = 0; img < n; ++img) first[img + 1] = first[img] + hdr[img * 16];
        c->pool().run(n, [&](size_t img) {
            r->n_extrema[img] = hdr[img * 16 + 1];
            hk[img].resize(hdr[img * 16]);
            for (size_t i = 0; i < hk[img].size(); ++i) {
                const sel::KpRec& q = recs[first[img] + i].rec;
                HostKeypoint& k = hk[img][i];
                k.x = q.x; k.y = q.y; k.response = q.response;
                k.size = lsize[q.level];
                k.octave = plan[q.level].octave;
                k.class_id = q.level;
                k.angle = 0.0f;
                if (libm_dev) std::memcpy(&k.angle, &recs[first[img] + i].sums.angle_bits, 4);  // (formed on the device: k_mldb)
                k.lx = k.ly = 0;
                k.xp = k.xm = k.yp = k.ym = 0.0f;
            }
        });
        if (libm_dev)
            for (uint32_t img = 0; img < n && dev_angles_ok; ++img)
                for (const HostKeypoint& k : hk[img])
                    if (!(std::fabs(k.angle) < 119.9f)) {  // NaN, or beyond the device's sinf / cosf: the host's libm takes the job's angles
                        dev_angles_ok = false;
                        break;
                    }
        c->last_total_kp = (uint32_t)total_kp;
    } else {
        c->pool().run(n, [&](size_t img) {  // images are independent
            if (!sorted) {
                sort_candidates(cands[img], plan);
                span[img] = {cands[img].data(), cands[img].size()};
            }
            if (d_rel && sorted && (((const uint32_t*)c->pin[7].p)[img] & 1u) == 0)  // (bit 0: > 65 533 candidates; bit 1 only keeps the image from the device's selection)
                select_keypoints_rel(span[img].first, span[img].second, (const uint16_t*)c->pin[6].p + (size_t)(span[img].first - hc) * (kRel1 + kRel2),
                                     kRel1, kRel2, plan, cfg, hk[img], &r->n_extrema[img]);
            else
                select_keypoints(span[img].first, span[img].second, plan, cfg, hk[img], &r->n_extrema[img]);
        });
    }
    total_kp = 0;
    for (uint32_t img = 0; img < n; ++img) {
        r->desc_off[img] = total_kp;
        total_kp += hk[img].size();
    }
    r->desc_off[n] = total_kp;
    if (c->profiling) c->prof.ms[AKZ_ST_HOST_KP] += now_ms() - t_host0;

    // ---- orientation (device sums + host atan2f) and M-LDB descriptors ----
    // keypoint parameters are built directly in pinned memory: pageable H2D copies above ~1 MiB make the
    // runtime pin user pages in place, which serialises concurrent contexts
    KpParam* params = nullptr;
    if (!dev_sel) {
        AKZ_TRY(ensure_pinned(c, c->pin[3], std::max<size_t>(1, total_kp) * sizeof(KpParam)));
        params = (KpParam*)c->pin[3].p;
    }
    if (!dev_sel) c->pool().run(n, [&](size_t img) {  // (per image on the workers: with few host threads every serial loop over 7 x 10^4 keypoints counts)
        for (size_t i = 0; i < hk[img].size(); ++i) {
            const HostKeypoint& k = hk[img][i];
            KpParam& p = params[r->desc_off[img] + i];
            const float ratio = (float)(1u << k.octave);
            p.xf = k.x / ratio;
            p.yf = k.y / ratio;
            p.scale = std::round(0.5f * k.size / ratio);
            p.level = k.class_id;
            p.img = (uint32_t)img;
            p._pad[0] = p._pad[1] = p._pad[2] = 0;
        }
    });
    r->kps.assign(n, {});
    const bool dev_desc = dev_sel && libm_dev != 0 && dev_angles_ok;  // angles in hk, descriptor rows in r->d_desc64 (and pin[2]) already
    c->libm_last = dev_desc ? libm_dev : 0;
    if (total_kp && dev_desc) {
        const double t_ml0 = now_ms();
        if (!(r->flags & AKZ_NO_HOST_DESCRIPTORS)) {
            const uint8_t* rows = (const uint8_t*)c->pin[2].p;
            r->rows64.resize(total_kp * 64);
            const size_t kRowChunk = 8192;
            c->pool().run((total_kp + kRowChunk - 1) / kRowChunk, [&](size_t j) {
                const size_t b = j * kRowChunk * 64, e = std::min<size_t>(total_kp * 64, b + kRowChunk * 64);
                std::memcpy(r->rows64.data() + b, rows + b, e - b);
            });
        }
        if (c->profiling) c->prof.ms[AKZ_ST_MLDB] += now_ms() - t_ml0;
    } else if (total_kp) {
        const double t_or0 = now_ms();
        KpParam* d_kp = (KpParam*)c->kp_in.p;
        OrientOut* oo = nullptr;
        size_t oo_stride = 1;
        AKZ_TRY(ensure(c, c->cosi, total_kp * 2 * sizeof(float)));
        if (dev_sel) {  // (parameters and sums are the device selection's: already here, next to the keypoints' records)
            oo = &((SelKpHost*)c->pin[0].p)->sums;
            oo_stride = 2;
        } else {
            AKZ_TRY(ensure(c, c->kp_in, total_kp * sizeof(KpParam)));
            AKZ_TRY(ensure(c, c->kp_out, total_kp * sizeof(OrientOut)));
            AKZ_TRY(ensure_pinned(c, c->pin[1], total_kp * std::max(sizeof(OrientOut), 2 * sizeof(float))));
            d_kp = (KpParam*)c->kp_in.p;
            OrientOut* d_oo = (OrientOut*)c->kp_out.p;
            AKZ_HIP_TRY(hipMemcpyAsync(d_kp, params, total_kp * sizeof(KpParam), hipMemcpyHostToDevice, s));
            // behind the fine octaves' diffusion of the batch begun right after this one, if there is one (next to the
            // VALU-bound diffusion launches these gather-bound kernels cost more than next to the bandwidth-bound detectors that
            // follow; the NEXT batch, not the one begun last: with two batches begun ahead that one is a whole step away); small
            // jobs are bound by the latency of this chain, not by the chip, and do not wait
            if ((uint64_t)r->w * r->h * n >= r->big_px && c->begin_seq.load() > job->seq)
                AKZ_HIP_TRY(hipStreamWaitEvent(s, c->fed_ev[(job->seq + 1) % akz_ctx::kFedRing], 0));
            launch::orientation(s, tab, d_kp, (uint32_t)total_kp, wmask, nwin, d_oo);
            AKZ_HIP_TRY(hipGetLastError());
            oo = (OrientOut*)c->pin[1].p;
            AKZ_HIP_TRY(hipMemcpyAsync(oo, d_oo, total_kp * sizeof(OrientOut), hipMemcpyDeviceToHost, s));
            AKZ_HIP_TRY(hipStreamSynchronize(s));
        }
        AKZ_TRY(ensure_pinned(c, c->pin[4], total_kp * 2 * sizeof(float)));
        float* cosi = (float*)c->pin[4].p;
        std::vector<float> angles(total_kp);
        const size_t kAngleChunk = 4096;  // keypoints per libm job (1 024 or 512: a 4K frame's call +3 ... +7 %: waking more workers costs more than it saves)
        c->pool().run((total_kp + kAngleChunk - 1) / kAngleChunk, [&](size_t j) {
            const size_t b = j * kAngleChunk, e = std::min<size_t>(total_kp, b + kAngleChunk);
            for (size_t g = b; g < e; ++g) {
                const OrientOut& og = oo[g * oo_stride];
                const float ang = og.found ? atan2f(og.sum_y, og.sum_x) : 0.0f;  // scale_space_extrema.rs:326
                angles[g] = ang;
                cosi[2 * g] = cosf(ang);                                                  // descriptors.rs:55-56
                cosi[2 * g + 1] = sinf(ang);
            }
        });
        c->pool().run(n, [&](size_t img) {
            for (size_t i = 0; i < hk[img].size(); ++i) hk[img][i].angle = angles[r->desc_off[img] + i];
        });
        AKZ_HIP_TRY(hipMemcpyAsync(c->cosi.p, cosi, total_kp * 2 * sizeof(float), hipMemcpyHostToDevice, s));
        const double t_ml0 = now_ms();
        if (c->profiling) c->prof.ms[AKZ_ST_ORIENT] += t_ml0 - t_or0;
        // descriptor rows live in a pooled device block owned by the result
        if (r->d_desc64 && r->desc_block_bytes < total_kp * 64) {  // (a block acquired for the device's own attempt, too small now)
            slab_release(c, r->d_desc64, r->desc_block_bytes);
            r->d_desc64 = nullptr;
        }
        if (!r->d_desc64) {
            void* blk = nullptr;
            AKZ_TRY(slab_acquire(c, total_kp * 64, &blk, &r->desc_block_bytes));
            r->d_desc64 = (uint8_t*)blk;
        }
        launch::mldb(s, tab, d_kp, (const float*)c->cosi.p, (uint32_t)total_kp, (uint32_t)cfg.descriptor_channels,
                     r->d_desc64);
        AKZ_HIP_TRY(hipGetLastError());
        if (!(r->flags & AKZ_NO_HOST_DESCRIPTORS)) {
            AKZ_TRY(ensure_pinned(c, c->pin[2], total_kp * 64));
            uint8_t* rows = (uint8_t*)c->pin[2].p;
            AKZ_HIP_TRY(hipMemcpyAsync(rows, r->d_desc64, total_kp * 64, hipMemcpyDeviceToHost, s));
            AKZ_HIP_TRY(hipStreamSynchronize(s));
            r->rows64.resize(total_kp * 64);  // un-padded lazily by akz_result_descriptors
            const size_t kRowChunk = 8192;
            c->pool().run((total_kp + kRowChunk - 1) / kRowChunk, [&](size_t j) {
                const size_t b = j * kRowChunk * 64, e = std::min<size_t>(total_kp * 64, b + kRowChunk * 64);
                std::memcpy(r->rows64.data() + b, rows + b, e - b);
            });
        } else {
            AKZ_HIP_TRY(hipStreamSynchronize(s));
        }
        if (c->profiling) c->prof.ms[AKZ_ST_MLDB] += now_ms() - t_ml0;
    }
    c->pool().run(n, [&](size_t img) {
        r->kps[img].resize(hk[img].size());
        for (size_t i = 0; i < hk[img].size(); ++i) {
            const HostKeypoint& k = hk[img][i];
            r->kps[img][i] = akz_keypoint{k.x, k.y, k.response, k.size, k.octave, k.class_id, k.angle, 0};
        }
    });
    if (c->profiling) {
        resolve_spans(c);  // only spans whose events have completed are resolved
        c->prof.ms[AKZ_ST_TOTAL] += now_ms() - job->t_begin_ms;
        c->prof.calls += 1;
        c->prof.pixels += (uint64_t)r->w * r->h * n;
    }
    ev_put(c, job->nms_done);
    job->nms_done = nullptr;
    *out = job->r.release();
    (void)job.release();
    return AKZ_OK;
}
static int extract_finish(akz_job* jobp, akz_result** out) {
    if (!jobp || !out) return AKZ_ERR_INVALID_ARG;
    *out = nullptr;
    int rc;
    if (jobp->fin) {  // finished (or being finished) by its lane's thread
        job_wait(jobp);
        rc = jobp->rc;
        *out = jobp->out;
        if (rc != AKZ_OK) set_error(jobp->err);
    } else {
        rc = extract_finish_body(jobp, out);
    }
    delete jobp;
    return rc;
}
static void finisher_loop(akz_ctx* c, Finisher* f) {
    (void)hipSetDevice(c->device);
    for (;;) {
        akz_job* j = nullptr;
        {
            std::unique_lock<std::mutex> lk(f->m);
            f->wake.wait(lk, [&] { return f->quit || !f->queue.empty(); });
            if (f->queue.empty()) return;  // quit is only set with nothing in flight
            j = f->queue.front();
            f->queue.erase(f->queue.begin());
        }
        akz_result* res = nullptr;
        int rc;
        try {
            rc = extract_finish_body(j, &res);
        } catch (const std::exception& e) {  // (allocation failure on this thread must not take the process down)
            set_error(std::string("extract_finish: ") + e.what());
            rc = AKZ_ERR_NO_MEMORY;
            res = nullptr;
        }
        std::string err = rc != AKZ_OK ? get_error() : std::string();
        {
            std::lock_guard<std::mutex> lk(f->m);
            j->rc = rc;
            j->out = res;
            j->err.swap(err);
            j->finished = true;
            --f->in_flight;
        }
        f->done.notify_all();
    }
}
static void finisher_post(akz_ctx* lane, akz_job* j) {
    if (!lane->fin) {
        lane->fin.reset(new Finisher);
        Finisher* f = lane->fin.get();
        f->th = std::thread([lane, f] { finisher_loop(lane, f); });
    }
    Finisher* f = lane->fin.get();
    {
        std::lock_guard<std::mutex> lk(f->m);
        j->fin = lane->fin;
        f->queue.push_back(j);
        ++f->in_flight;
    }
    f->wake.notify_one();
}

// `pub mod ops` on CALLER-PROVIDED evolutions (ops::scale_space_extrema::detect_keypoints, scale_space_extrema.rs:199-203,
// and ops::descriptors::extract_descriptors, descriptors.rs:14-27, take a `Vec<EvolutionStep>` that the caller may have
// built or modified itself): the host planes of one image are uploaded into a result's slab, the extrema pass runs on
// the uploaded Ldet planes and the usual finish half (host keypoint logic, orientation, descriptors) follows.
static int extract_from_planes(akz_ctx* c, uint32_t w, uint32_t h, const akz_config* cfgp, const float* const* planes,
                               uint64_t n_levels, uint32_t flags, akz_result** out) {
    if (!out) return AKZ_ERR_INVALID_ARG;
    *out = nullptr;
    AKZ_TRY(bind(c));
    if (!cfgp || !planes) {
        set_error("extract_from_planes: null config / plane table");
        return AKZ_ERR_INVALID_ARG;
    }
    int slot = -1;
    for (int i = 0; i < akz_ctx::kSlots; ++i)
        if (!c->slot_busy[i]) {
            slot = i;
            break;
        }
    if (slot < 0) {
        set_error("extract_from_planes: too many extractions in flight on this context (finish one first)");
        return AKZ_ERR_INVALID_ARG;
    }
    std::unique_ptr<akz_job> job(new akz_job);
    job->in_hand = c->in_hand;
    job->alone_at_begin = job->in_hand->fetch_add(1) == 0;
    job->r.reset(new akz_result);
    akz_result* r = job->r.get();
    r->ctx = c;
    ++c->live_results;
    r->cfg = *cfgp;
    r->w = w; r->h = h; r->n = 1;
    r->flags = (flags & ~(uint32_t)AKZ_NO_DETECT) | AKZ_KEEP_ALL_PLANES;
    AKZ_TRY(build_plan(w, h, r->cfg, r->plan));
    const std::vector<LevelPlan>& plan = r->plan;
    const size_t L = plan.size();
    if (n_levels != L) {
        set_error("extract_from_planes: the number of evolutions does not match allocate_evolutions(width, height, options)");
        return AKZ_ERR_INVALID_ARG;
    }
    const bool detect = !(flags & AKZ_NO_DETECT);
    for (size_t l = 0; l < L; ++l)
        for (int p : {(int)AKZ_LT, (int)AKZ_LX, (int)AKZ_LY, (int)AKZ_LDET})
            if (!planes[l * 10 + p] && (detect || p != AKZ_LDET)) {
                set_error("extract_from_planes: Lt, Lx, Ly (and Ldet for detection) of every evolution are required");
                return AKZ_ERR_INVALID_ARG;
            }
    hipStream_t s = c->stream;
    std::memset(r->planes, 0, sizeof(r->planes));
    size_t off = 0;
    std::vector<std::pair<float**, size_t>> fix;
    for (size_t l = 0; l < L; ++l)
        for (int p = 0; p < 10; ++p) {
            if (!planes[l * 10 + p]) continue;
            fix.emplace_back(&r->planes[l][p], off);
            off += align_up(plane_bytes(plan[l].w, plan[l].h, 1), 256);
        }
    const size_t k_off = off;
    off += 256;
    AKZ_TRY(slab_acquire(c, off, &r->slab, &r->slab_bytes));
    for (auto& f : fix) *f.first = (float*)((char*)r->slab + f.second);
    if (!r->planes[0][AKZ_LSMOOTH]) r->planes[0][AKZ_LSMOOTH] = r->planes[0][AKZ_LT];
    r->d_k = (double*)((char*)r->slab + k_off);
    struct Guard {
        akz_result* r;
        bool armed = true;
        ~Guard() {
            if (armed) result_release_device(r);
        }
    } guard{r};
    job->t_begin_ms = now_ms();
    AKZ_HIP_TRY(hipMemsetAsync(r->d_k, 0, sizeof(double), s));  // the contrast factor is not part of the inputs
    for (size_t l = 0; l < L; ++l)
        for (int p = 0; p < 10; ++p)
            if (planes[l * 10 + p] && r->planes[l][p] && !(l == 0 && p == AKZ_LSMOOTH && r->planes[0][AKZ_LSMOOTH] == r->planes[0][AKZ_LT]))
                AKZ_HIP_TRY(hipMemcpyAsync(r->planes[l][p], planes[l * 10 + p], plane_bytes(plan[l].w, plan[l].h, 1),
                                           hipMemcpyHostToDevice, s));
    AKZ_HIP_TRY(hipStreamSynchronize(s));  // the caller's planes are pageable host memory: complete before returning
    const uint32_t cap = (uint32_t)std::min<uint64_t>((uint64_t)std::max<uint32_t>(c->cand_cap_hint.load(), 16u), 0x7fffffffull / sizeof(Candidate));
    AKZ_TRY(ensure(c, c->cand_slot[slot], (size_t)cap * sizeof(Candidate)));
    AKZ_TRY(ensure(c, c->count_slot[slot], 256));
    uint32_t* d_count = (uint32_t*)c->count_slot[slot].p;
    AKZ_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
    if (detect)
        for (size_t l = 0; l < L; ++l)
            launch::nms(s, r->planes[l][AKZ_LDET], plan[l].w, plan[l].h, 1, (uint64_t)plan[l].w * plan[l].h, (uint32_t)l,
                        (float)r->cfg.detector_threshold, border_margin(plan[l], r->cfg), (Candidate*)c->cand_slot[slot].p, cap,
                        d_count);
    AKZ_HIP_TRY(hipGetLastError());
    job->nms_done = StageTimer::get(c);
    AKZ_HIP_TRY(hipEventRecord(job->nms_done, s));
    job->slot = slot;
    job->cap = cap;
    c->slot_busy[slot] = true;
    guard.armed = false;
    return extract_finish(job.release(), out);
}

template <typename T>
static int extract_impl(akz_ctx* c, const T* d_imgs, uint32_t w, uint32_t h, uint32_t n, const akz_config* cfgp,
                        uint32_t flags, akz_result** out) {
    if (!out) return AKZ_ERR_INVALID_ARG;
    *out = nullptr;
    akz_job* job = nullptr;
    AKZ_TRY(extract_begin<T>(c, d_imgs, w, h, n, cfgp, flags, &job, -1, nullptr, /*sync_call=*/true));
    return extract_finish(job, out);
}

template <typename T>
static int extract_host(akz_ctx* c, const T* img, uint32_t w, uint32_t h, const akz_config* cfg, uint32_t flags,
                        akz_result** out) {
    AKZ_TRY(bind(c));
    if (!img || w == 0 || h == 0) {
        set_error("extract: null or empty image");
        return AKZ_ERR_INVALID_ARG;
    }
    const size_t bytes = (size_t)w * h * sizeof(T);
    AKZ_TRY(ensure(c, c->scratch[4], bytes));
    AKZ_HIP_TRY(hipMemcpyAsync(c->scratch[4].p, img, bytes, hipMemcpyHostToDevice, c->stream));
    // the frame is consumed by level 0 on the same stream before anything else touches scratch[4]
    return extract_impl<T>(c, (const T*)c->scratch[4].p, w, h, 1, cfg, flags, out);
}

extern "C" {

int akz_extract_gray_u8(akz_ctx* c, const uint8_t* img, uint32_t w, uint32_t h, const akz_config* cfg, uint32_t flags,
                        akz_result** out) {
    return extract_host<uint8_t>(c, img, w, h, cfg, flags, out);
}
int akz_extract_gray_f32(akz_ctx* c, const float* img, uint32_t w, uint32_t h, const akz_config* cfg, uint32_t flags,
                         akz_result** out) {
    return extract_host<float>(c, img, w, h, cfg, flags, out);
}
int akz_extract_device_u8(akz_ctx* c, const uint8_t* d_imgs, uint32_t w, uint32_t h, uint32_t n,
                          const akz_config* cfg, uint32_t flags, akz_result** out) {
    return extract_impl<uint8_t>(c, d_imgs, w, h, n, cfg, flags, out);
}
int akz_extract_device_f32(akz_ctx* c, const float* d_imgs, uint32_t w, uint32_t h, uint32_t n,
                           const akz_config* cfg, uint32_t flags, akz_result** out) {
    return extract_impl<float>(c, d_imgs, w, h, n, cfg, flags, out);
}

// lanes = 1 (default): every job runs on the context's own stream.  lanes = k > 1: jobs below gates::kLanePx are dealt to k child
// contexts in turn (larger jobs fill the chip on their own and stay on the context).  A job's result belongs to the
// lane it ran on; nothing else changes for the caller (same begin / finish / result calls, bit-identical results).
int akz_ctx_set_lanes(akz_ctx* c, uint32_t lanes) {
    AKZ_TRY(bind(c));
    if (lanes < 1 || lanes > 8) return AKZ_ERR_INVALID_ARG;
    const size_t want = lanes == 1 ? 0 : lanes;
    // a lane that goes away must have no extraction in flight: its job would be finished on destroyed streams
    for (size_t i = want; i < c->lanes.size(); ++i) {
        for (int k = 0; k < akz_ctx::kSlots; ++k)
            if (c->lanes[i]->slot_busy[k]) {
                set_error("akz_ctx_set_lanes: a lane that would be removed has an extraction in flight (finish or abandon it first)");
                return AKZ_ERR_INVALID_ARG;
            }
    }
    while (c->lanes.size() > want) {
        AKZ_TRY(akz_ctx_destroy(c->lanes.back()));
        c->lanes.pop_back();
    }
    while (c->lanes.size() < want) {
        void* st = nullptr;
        AKZ_TRY(akz_stream_create(c->device, &st));
        akz_ctx* l = nullptr;
        const int rc = akz_ctx_create(c->device, st, &l);
        if (rc != AKZ_OK) {
            (void)akz_stream_destroy(c->device, st);
            return rc;
        }
        l->own_stream = true;
        l->is_lane = true;
        l->det_mode = c->det_mode; l->prep_mode = c->prep_mode; l->match_mode = c->match_mode; l->fed_mode = c->fed_mode;
        l->cand_cap_hint = c->cand_cap_hint.load();
        l->stream_min_px = c->stream_min_px;
        l->big_px_sync = std::max(c->big_px_sync, c->lane_px);  // (a job that is dealt to a lane runs there as a one-stream chain)
        l->big_px_async = std::max(c->big_px_async, c->lane_px);
        l->host_threads = c->host_threads;
        l->profiling = c->profiling;
        l->dbg_pair_chunks = c->dbg_pair_chunks;
        l->dbg_set_chunks = c->dbg_set_chunks;
        l->dbg_host_sort = c->dbg_host_sort;
        l->dbg_select = c->dbg_select;
        c->lanes.push_back(l);
    }
    c->next_lane = 0;
    return place_lanes(c);
}
// on != 0: the finish half of every job that is dealt to a lane starts on the lane's own thread as soon as the job has
// been begun; akz_extract_finish waits for it and hands the result over (bit-identical; errors of the finish half are
// reported there as before).  Jobs that stay on the context itself (no lanes, or batch-path jobs) are not affected.
int akz_ctx_set_eager_finish(akz_ctx* c, int on) {
    AKZ_TRY(bind(c));
    c->eager_finish = on != 0;
    return AKZ_OK;
}
static int extract_begin_dispatch(akz_ctx* c, const void* imgs, bool is_u8, uint32_t w, uint32_t h, uint32_t n,
                                  const akz_config* cfg, uint32_t flags, akz_job** out, bool on_host = false) {
    akz_ctx* on = c;
    if (c && !c->lanes.empty() && (uint64_t)w * h * n < c->lane_px) {
        AKZ_TRY(bind(c, false));
        on = c->lanes[c->next_lane++ % c->lanes.size()];
        // the lane starts when the caller's stream has reached this point (its inputs are complete)
        if (!c->lane_in) AKZ_HIP_TRY(hipEventCreateWithFlags(&c->lane_in, hipEventDisableTiming));
        AKZ_HIP_TRY(hipEventRecord(c->lane_in, c->stream));
        AKZ_HIP_TRY(hipStreamWaitEvent(on->stream, c->lane_in, 0));
    }
    int slot = -1;
    const void* d_imgs = imgs;
    if (on_host) {
        // Frames in host memory: upload on the context's copy stream into the staging buffer of the job slot this
        // extraction will hold (exclusive until its finish); only this job's kernels wait for the copy, so it runs
        // under whatever the main stream is doing for the batch before.
        if (out) *out = nullptr;
        AKZ_TRY(bind(on, true, on->is_lane));
        if (!imgs || w == 0 || h == 0 || n == 0) {
            set_error("extract_begin_host: null or empty frames");
            return AKZ_ERR_INVALID_ARG;
        }
        for (int i = 0; i < akz_ctx::kSlots && slot < 0; ++i)
            if (!on->slot_busy[i]) slot = i;
        if (slot < 0) {
            set_error("extract_begin: too many extractions in flight on this context (finish one first)");
            return AKZ_ERR_INVALID_ARG;
        }
        const size_t bytes = (size_t)w * h * n * (is_u8 ? 1 : sizeof(float));
        AKZ_TRY(ensure(on, on->stage[slot], bytes));
        if (!on->copy) AKZ_HIP_TRY(hipStreamCreateWithFlags(&on->copy, hipStreamNonBlocking));
        if (!on->staged[slot]) AKZ_HIP_TRY(hipEventCreateWithFlags(&on->staged[slot], hipEventDisableTiming));
        AKZ_HIP_TRY(hipMemcpyAsync(on->stage[slot].p, imgs, bytes, hipMemcpyHostToDevice, on->copy));
        AKZ_HIP_TRY(hipEventRecord(on->staged[slot], on->copy));
        AKZ_HIP_TRY(hipStreamWaitEvent(on->stream, on->staged[slot], 0));
        d_imgs = on->stage[slot].p;
    }
    hipEvent_t ready = on_host ? on->staged[slot] : nullptr;  // frames this library uploaded: complete behind that event
    const int rc = is_u8 ? extract_begin<uint8_t>(on, (const uint8_t*)d_imgs, w, h, n, cfg, flags, out, slot, ready)
                         : extract_begin<float>(on, (const float*)d_imgs, w, h, n, cfg, flags, out, slot, ready);
    // a job dealt to a lane is always finished by the lane's own thread (lanes whose jobs wait for the caller's finish call
    // measured SLOWER than no lanes: the chains overlap on the chip but the finish halves queue on one host thread)
    if (rc == AKZ_OK && (c->eager_finish || on != c)) finisher_post(on, *out);
    return rc;
}
int akz_extract_begin_host_u8(akz_ctx* c, const uint8_t* h_imgs, uint32_t w, uint32_t h, uint32_t n, const akz_config* cfg,
                              uint32_t flags, akz_job** out) {
    return extract_begin_dispatch(c, h_imgs, true, w, h, n, cfg, flags, out, true);
}
int akz_extract_begin_host_f32(akz_ctx* c, const float* h_imgs, uint32_t w, uint32_t h, uint32_t n, const akz_config* cfg,
                               uint32_t flags, akz_job** out) {
    return extract_begin_dispatch(c, h_imgs, false, w, h, n, cfg, flags, out, true);
}
int akz_extract_begin_device_u8(akz_ctx* c, const uint8_t* d_imgs, uint32_t w, uint32_t h, uint32_t n,
                                const akz_config* cfg, uint32_t flags, akz_job** out) {
    return extract_begin_dispatch(c, d_imgs, true, w, h, n, cfg, flags, out);
}
int akz_extract_begin_device_f32(akz_ctx* c, const float* d_imgs, uint32_t w, uint32_t h, uint32_t n,
                                 const akz_config* cfg, uint32_t flags, akz_job** out) {
    return extract_begin_dispatch(c, d_imgs, false, w, h, n, cfg, flags, out);
}
int akz_extract_finish(akz_job* job, akz_result** out) { return extract_finish(job, out); }

// The job gates from timings on this machine (akaze_hip_debug.h).  Five lone frames; each through the synchronous call and
// through the begin / finish interface with two jobs in flight, with the batch path forced off (gate above every size) and
// on (gate 0); medians of kReps.  A gate = the smallest size from which the batch path wins at that size and every larger one.
int akz_ctx_calibrate_gates(akz_ctx* c, uint64_t* sync_px, uint64_t* async_px, double* ms_out) {
    AKZ_TRY(bind(c));
    static const uint32_t kShapes[5][2] = {{1280, 720}, {1600, 900}, {1920, 1080}, {2688, 1512}, {3840, 2160}};  // (0.9 .. 8.3 Mpx: both sides of the gates)
    constexpr int kReps = 5;
    akz_config cfg;
    akz_config_default(&cfg);
    const uint64_t keep_sync = c->big_px_sync, keep_async = c->big_px_async;
    struct Restore {
        akz_ctx* c;
        uint64_t s, a;
        bool armed = true;
        ~Restore() { if (armed) { c->big_px_sync = s; c->big_px_async = a; } }
    } restore{c, keep_sync, keep_async};
    double ms[5][4];
    auto median = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    for (int si = 0; si < 5; ++si) {
        const uint32_t w = kShapes[si][0], h = kShapes[si][1];
        std::vector<uint8_t> frame((size_t)w * h);
        AKZ_TRY(akz_synth_frame_u8(frame.data(), w, h, 0, 0, 0));
        void* d = nullptr;
        AKZ_TRY(akz_device_malloc(c, frame.size(), &d));
        struct Free { akz_ctx* c; void* d; ~Free() { (void)akz_device_free(c, d); } } fr{c, d};
        AKZ_TRY(akz_memcpy_h2d(c, d, frame.data(), frame.size()));
        for (int path = 0; path < 2; ++path) {
            c->big_px_sync = c->big_px_async = path ? 0 : ~0ull;
            // synchronous call
            std::vector<double> t;
            for (int r = 0; r < kReps + 1; ++r) {
                akz_result* res = nullptr;
                const double t0 = now_ms();
                AKZ_TRY(akz_extract_device_u8(c, (const uint8_t*)d, w, h, 1, &cfg, AKZ_NO_HOST_DESCRIPTORS, &res));
                if (r) t.push_back(now_ms() - t0);  // (the first call of a shape allocates)
                akz_result_free(res);
            }
            ms[si][path] = median(t);
            // begin / finish with two jobs in flight: time per frame in steady state
            t.clear();
            akz_job* jobs[2] = {nullptr, nullptr};
            AKZ_TRY(akz_extract_begin_device_u8(c, (const uint8_t*)d, w, h, 1, &cfg, AKZ_NO_HOST_DESCRIPTORS | AKZ_INPUT_READY, &jobs[0]));
            double last = now_ms();
            for (int r = 0; r < 2 * kReps + 2; ++r) {
                akz_job* next = nullptr;
                int st = akz_extract_begin_device_u8(c, (const uint8_t*)d, w, h, 1, &cfg, AKZ_NO_HOST_DESCRIPTORS | AKZ_INPUT_READY, &next);
                akz_result* res = nullptr;
                if (st == AKZ_OK) st = akz_extract_finish(jobs[0], &res);
                else (void)akz_job_abandon(jobs[0]);
                if (st != AKZ_OK) {
                    if (next) (void)akz_job_abandon(next);
                    return st;
                }
                akz_result_free(res);
                jobs[0] = next;
                const double now = now_ms();
                if (r >= 2) t.push_back(now - last);
                last = now;
            }
            akz_result* res = nullptr;
            AKZ_TRY(akz_extract_finish(jobs[0], &res));
            akz_result_free(res);
            ms[si][2 + path] = median(t);
        }
    }
    auto gate = [&](int lone, int batch) -> uint64_t {
        int first = 5;  // smallest index from which the batch path wins everywhere above
        for (int si = 4; si >= 0 && ms[si][batch] < ms[si][lone]; --si) first = si;
        return first == 5 ? (uint64_t)kShapes[4][0] * kShapes[4][1] + 1 : (uint64_t)kShapes[first][0] * kShapes[first][1];
    };
    restore.armed = false;
    c->big_px_sync = gate(0, 1);
    c->big_px_async = gate(2, 3);
    for (akz_ctx* l : c->lanes) {  // (a job that is dealt to a lane runs there as a one-stream chain)
        l->big_px_sync = std::max(c->big_px_sync, c->lane_px);
        l->big_px_async = std::max(c->big_px_async, c->lane_px);
    }
    if (sync_px) *sync_px = c->big_px_sync;
    if (async_px) *async_px = c->big_px_async;
    if (ms_out)
        for (int si = 0; si < 5; ++si)
            for (int k = 0; k < 4; ++k) ms_out[si * 4 + k] = ms[si][k];
    return AKZ_OK;
}
// Measurement hook (bench.py `single_frame.graph`): is a lone frame's begin phase — a chain of ~45 dependent
// launches — shorter as ONE hipGraph launch?  The begin phase of (d_imgs, w, h, n, cfg) is stream-captured into a
// graph (same kernels, same buffers), then `reps` graph launches and `reps` plain enqueues of the same chain are
// timed from an idle stream with HIP events.  Results of the captured chain are discarded.
int akz_ctx_graph_probe(akz_ctx* c, const uint8_t* d_imgs, uint32_t w, uint32_t h, uint32_t n, const akz_config* cfg,
                        uint32_t flags, uint32_t reps, double* ms_graph, double* ms_plain, uint64_t* graph_nodes) {
    AKZ_TRY(bind(c));
    if (!d_imgs || !cfg || !ms_graph || !ms_plain || reps == 0) return AKZ_ERR_INVALID_ARG;
    if ((uint64_t)w * h * n >= c->big_px_async) {  // such a batch forks its coarse chain and completes on that stream: not one capture
        set_error("akz_ctx_graph_probe: jobs that take the batch path fork onto a second stream and cannot be captured from one");
        return AKZ_ERR_INVALID_ARG;
    }
    const int prof = c->profiling;
    c->profiling = 0;
    struct Restore { akz_ctx* c; int p; ~Restore() { c->profiling = p; } } restore{c, prof};
    // 1. warm: every buffer the chain uses exists afterwards (no allocation may happen while capturing)
    for (int i = 0; i < 2; ++i) {
        akz_result* r = nullptr;
        AKZ_TRY(extract_impl<uint8_t>(c, d_imgs, w, h, n, cfg, flags, &r));
        result_delete(r);
    }
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    AKZ_HIP_TRY(hipEventCreate(&e0));
    AKZ_HIP_TRY(hipEventCreate(&e1));
    // 2. plain chain, from an idle stream each time
    double plain = 0.0;
    for (uint32_t i = 0; i < reps; ++i) {
        akz_job* job = nullptr;
        AKZ_HIP_TRY(hipEventRecord(e0, c->stream));
        AKZ_TRY(extract_begin<uint8_t>(c, d_imgs, w, h, n, cfg, flags, &job));
        AKZ_HIP_TRY(hipEventRecord(e1, c->stream));
        AKZ_HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.0f;
        AKZ_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        plain += ms;
        job_destroy(job);
    }
    // 3. the same chain as a graph
    akz_job* cap = nullptr;
    AKZ_HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    const int st = extract_begin<uint8_t>(c, d_imgs, w, h, n, cfg, flags, &cap);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
    if (st != AKZ_OK || ce != hipSuccess || !graph) {
        if (cap) job_destroy(cap);
        if (st == AKZ_OK) set_error(std::string("stream capture of the begin phase failed: ") + hipGetErrorString(ce));
        return st != AKZ_OK ? st : AKZ_ERR_HIP;
    }
    size_t nodes = 0;
    (void)hipGraphGetNodes(graph, nullptr, &nodes);
    if (graph_nodes) *graph_nodes = nodes;
    hipGraphExec_t exec = nullptr;
    AKZ_HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    AKZ_HIP_TRY(hipGraphLaunch(exec, c->stream));  // warm
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    double gms = 0.0;
    for (uint32_t i = 0; i < reps; ++i) {
        AKZ_HIP_TRY(hipEventRecord(e0, c->stream));
        AKZ_HIP_TRY(hipGraphLaunch(exec, c->stream));
        AKZ_HIP_TRY(hipEventRecord(e1, c->stream));
        AKZ_HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.0f;
        AKZ_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        gms += ms;
    }
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    job_destroy(cap);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_graph = gms / reps;
    *ms_plain = plain / reps;
    return AKZ_OK;
}

int akz_extract_from_planes(akz_ctx* c, uint32_t w, uint32_t h, const akz_config* cfg, const float* const* planes,
                            uint64_t n_levels, uint32_t flags, akz_result** out) {
    return extract_from_planes(c, w, h, cfg, planes, n_levels, flags, out);
}
int akz_job_abandon(akz_job* job) {
    if (!job) return AKZ_OK;
    job_wait(job);
    const akz_result* r = job->out ? job->out : job->r.get();
    if (r) (void)hipSetDevice(r->ctx->device);
    job_destroy(job);
    return AKZ_OK;
}

int akz_result_free(akz_result* r) {
    if (!r) return AKZ_OK;
    (void)hipSetDevice(r->ctx->device);
    result_delete(r);
    return AKZ_OK;
}
int akz_result_num_images(const akz_result* r, uint64_t* n) {
    if (!r || !n) return AKZ_ERR_INVALID_ARG;
    *n = r->n;
    return AKZ_OK;
}
static int check_img(const akz_result* r, uint64_t img) {
    if (!r || img >= r->n) {
        set_error("null result or image index out of range");
        return AKZ_ERR_INVALID_ARG;
    }
    return AKZ_OK;
}
int akz_result_counts(const akz_result* r, uint64_t img, uint64_t* n_levels, uint64_t* n_keypoints,
                      uint64_t* desc_bytes) {
    AKZ_TRY(check_img(r, img));
    if (n_levels) *n_levels = r->plan.size();
    if (n_keypoints) *n_keypoints = r->kps[(size_t)img].size();
    if (desc_bytes) *desc_bytes = ((6 + 36 + 120) * r->cfg.descriptor_channels + 7) / 8;
    return AKZ_OK;
}
int akz_result_keypoints(const akz_result* r, uint64_t img, akz_keypoint* out) {
    AKZ_TRY(check_img(r, img));
    const auto& k = r->kps[(size_t)img];
    if (!k.empty()) {
        if (!out) return AKZ_ERR_INVALID_ARG;
        std::memcpy(out, k.data(), k.size() * sizeof(akz_keypoint));
    }
    return AKZ_OK;
}
int akz_result_descriptors(const akz_result* r, uint64_t img, uint8_t* out) {
    AKZ_TRY(check_img(r, img));
    if (r->flags & AKZ_NO_HOST_DESCRIPTORS) {
        set_error("descriptors were kept on the device (AKZ_NO_HOST_DESCRIPTORS)");
        return AKZ_ERR_INVALID_ARG;
    }
    const size_t nk = r->kps[(size_t)img].size();
    if (nk) {
        if (!out) return AKZ_ERR_INVALID_ARG;
        const size_t nb = ((6 + 36 + 120) * r->cfg.descriptor_channels + 7) / 8;
        const uint8_t* rows = r->rows64.data() + r->desc_off[(size_t)img] * 64;
        for (size_t i = 0; i < nk; ++i) std::memcpy(out + i * nb, rows + i * 64, nb);
    }
    return AKZ_OK;
}
// ops::scale_space_extrema::compute_main_orientation (scale_space_extrema.rs:207-329) and
// ops::descriptors::extract_descriptors (descriptors.rs:14-35) for CALLER-SUPPLIED keypoints of image `img`, on the
// pyramid the result retains: what the reference's two public ops do when they are handed a keypoint list that did
// not come out of detect_keypoints (re-description, externally detected points).
int akz_result_describe_keypoints(const akz_result* r, uint64_t img, akz_keypoint* kps, uint64_t n_kp,
                                  int compute_orientation, uint8_t* descriptors) {
    AKZ_TRY(check_img(r, img));
    akz_ctx* c = r->ctx;
    AKZ_TRY(bind(c));
    if (n_kp == 0) return AKZ_OK;
    if (!kps || !descriptors) {
        set_error("akz_result_describe_keypoints: null argument");
        return AKZ_ERR_INVALID_ARG;
    }
    const size_t L = r->plan.size();
    LevelTable tab;
    std::memset(&tab, 0, sizeof(tab));
    for (size_t l = 0; l < L; ++l) {
        tab.lv[l].lt = r->planes[l][AKZ_LT];
        tab.lv[l].lx = r->planes[l][AKZ_LX];
        tab.lv[l].ly = r->planes[l][AKZ_LY];
        tab.lv[l].w = r->plan[l].w;
        tab.lv[l].h = r->plan[l].h;
        tab.lv[l].stride = (uint64_t)r->plan[l].w * r->plan[l].h;
    }
    AKZ_TRY(ensure_aux(c));
    hipStream_t s = c->aux;
    AKZ_TRY(ensure_pinned(c, c->pin[3], n_kp * sizeof(KpParam)));
    KpParam* params = (KpParam*)c->pin[3].p;
    for (uint64_t i = 0; i < n_kp; ++i) {
        const akz_keypoint& k = kps[i];
        if (k.class_id >= L || k.octave > 30) {  // the reference indexes evolutions[class_id] and would panic
            set_error("akz_result_describe_keypoints: keypoint class_id / octave out of range");
            return AKZ_ERR_INVALID_ARG;
        }
        KpParam& p = params[i];
        const float ratio = (float)(1u << k.octave);
        p.xf = k.x / ratio;
        p.yf = k.y / ratio;
        p.scale = std::round(0.5f * k.size / ratio);
        p.level = (uint32_t)k.class_id;
        p.img = (uint32_t)img;
        p._pad[0] = p._pad[1] = p._pad[2] = 0;
    }
    AKZ_TRY(ensure(c, c->kp_in, n_kp * sizeof(KpParam)));
    AKZ_TRY(ensure(c, c->kp_out, n_kp * sizeof(OrientOut)));
    AKZ_TRY(ensure(c, c->cosi, n_kp * 2 * sizeof(float)));
    AKZ_TRY(ensure_pinned(c, c->pin[1], n_kp * std::max(sizeof(OrientOut), 2 * sizeof(float))));
    KpParam* d_kp = (KpParam*)c->kp_in.p;
    AKZ_HIP_TRY(hipMemcpyAsync(d_kp, params, n_kp * sizeof(KpParam), hipMemcpyHostToDevice, s));
    if (compute_orientation) {
        unsigned long long wmask = 0;
        uint32_t nwin = 0;
        orientation_windows(&wmask, &nwin);
        OrientOut* d_oo = (OrientOut*)c->kp_out.p;
        launch::orientation(s, tab, d_kp, (uint32_t)n_kp, wmask, nwin, d_oo);
        AKZ_HIP_TRY(hipGetLastError());
        OrientOut* oo = (OrientOut*)c->pin[1].p;
        AKZ_HIP_TRY(hipMemcpyAsync(oo, d_oo, n_kp * sizeof(OrientOut), hipMemcpyDeviceToHost, s));
        AKZ_HIP_TRY(hipStreamSynchronize(s));
        for (uint64_t i = 0; i < n_kp; ++i)  // no window sum above zero: the angle keeps its value (scale_space_extrema.rs:322-327)
            if (oo[i].found) kps[i].angle = atan2f(oo[i].sum_y, oo[i].sum_x);
    }
    AKZ_TRY(ensure_pinned(c, c->pin[4], n_kp * 2 * sizeof(float)));
    float* cosi = (float*)c->pin[4].p;
    for (uint64_t i = 0; i < n_kp; ++i) {
        cosi[2 * i] = cosf(kps[i].angle);  // descriptors.rs:55-56
        cosi[2 * i + 1] = sinf(kps[i].angle);
    }
    AKZ_HIP_TRY(hipMemcpyAsync(c->cosi.p, cosi, n_kp * 2 * sizeof(float), hipMemcpyHostToDevice, s));
    AKZ_TRY(ensure(c, c->match_a, n_kp * 64));
    uint8_t* d_rows = (uint8_t*)c->match_a.p;
    launch::mldb(s, tab, d_kp, (const float*)c->cosi.p, (uint32_t)n_kp, (uint32_t)r->cfg.descriptor_channels, d_rows);
    AKZ_HIP_TRY(hipGetLastError());
    AKZ_TRY(ensure_pinned(c, c->pin[2], n_kp * 64));
    uint8_t* rows = (uint8_t*)c->pin[2].p;
    AKZ_HIP_TRY(hipMemcpyAsync(rows, d_rows, n_kp * 64, hipMemcpyDeviceToHost, s));
    AKZ_HIP_TRY(hipStreamSynchronize(s));
    const size_t nb = ((6 + 36 + 120) * r->cfg.descriptor_channels + 7) / 8;
    for (uint64_t i = 0; i < n_kp; ++i) std::memcpy(descriptors + i * nb, rows + i * 64, nb);
    return AKZ_OK;
}
int akz_result_device_descriptors(const akz_result* r, uint64_t img, const uint8_t** d_desc, uint64_t* n_keypoints) {
    AKZ_TRY(check_img(r, img));
    if (d_desc) *d_desc = r->d_desc64 ? r->d_desc64 + r->desc_off[(size_t)img] * 64 : nullptr;
    if (n_keypoints) *n_keypoints = r->kps[(size_t)img].size();
    return AKZ_OK;
}
int akz_result_copy_device_descriptors(const akz_result* r, uint8_t* d_dst, uint64_t capacity_rows, uint64_t* rows) {
    if (!r) return AKZ_ERR_INVALID_ARG;
    const uint64_t total = r->desc_off.empty() ? 0 : r->desc_off.back();
    if (rows) *rows = total;
    if (total == 0) return AKZ_OK;
    if (!d_dst || capacity_rows < total) {
        set_error("copy_device_descriptors: destination too small");
        return AKZ_ERR_BUFFER;
    }
    AKZ_TRY(bind(r->ctx, true, false));
    // on the auxiliary stream and complete on return: the context's main stream may already be busy
    // with the next batch, and the caller typically hands d_dst to a collective on yet another stream
    akz_ctx* c = r->ctx;
    AKZ_TRY(ensure_aux(c));
    AKZ_HIP_TRY(hipMemcpyAsync(d_dst, r->d_desc64, total * 64, hipMemcpyDeviceToDevice, c->aux));
    AKZ_HIP_TRY(hipStreamSynchronize(c->aux));
    return AKZ_OK;
}
int akz_result_contrast(const akz_result* r, uint64_t img, double* k) {
    AKZ_TRY(check_img(r, img));
    if (!k) return AKZ_ERR_INVALID_ARG;
    *k = r->k_host[(size_t)img];
    return AKZ_OK;
}
int akz_result_level_info(const akz_result* r, uint64_t level, double* etime, double* esigma, uint32_t* octave,
                          uint32_t* sublevel, uint32_t* sigma_size, uint32_t* w, uint32_t* h, uint64_t* n_tau,
                          double* tau, uint64_t tau_cap) {
    if (!r) return AKZ_ERR_INVALID_ARG;
    return level_info_out(r->plan, level, etime, esigma, octave, sublevel, sigma_size, w, h, nullptr, n_tau, tau,
                          tau_cap);
}
int akz_result_device_plane(const akz_result* r, uint64_t img, uint64_t level, akz_plane plane,
                            const float** d_plane) {
    AKZ_TRY(check_img(r, img));
    if (level >= r->plan.size() || (int)plane < 0 || (int)plane > 9 || !d_plane) {
        set_error("level/plane out of range");
        return AKZ_ERR_INVALID_ARG;
    }
    const float* base = r->planes[(size_t)level][(int)plane];
    const LevelPlan& lv = r->plan[(size_t)level];
    *d_plane = base ? base + (size_t)img * lv.w * lv.h : nullptr;
    return AKZ_OK;
}
// A plane that the extraction did not keep (Lxx, Lyy, Lxy, Lstep without AKZ_KEEP_ALL_PLANES) is recomputed for one
// image from planes that are always kept, with the kernels and in the order of the extraction: second derivatives
// from the level's Lsmooth (detector_response.rs:9-13), Lstep by repeating the level's diffusion from the previous
// level's Lt (lib.rs:80-92, :109-118).  Bit-identical to the kept planes; *d_out points into context scratch memory
// that the next call overwrites.
static int recompute_plane(const akz_result* r, uint64_t img, uint64_t level, akz_plane plane, const float** d_out) {
    akz_ctx* c = r->ctx;
    AKZ_TRY(bind(c));
    const LevelPlan& lv = r->plan[(size_t)level];
    const size_t px = (size_t)lv.w * lv.h, pb = px * sizeof(float);
    auto img_plane = [&](uint64_t l, int p) { return r->planes[(size_t)l][p] + (size_t)img * r->plan[(size_t)l].w * r->plan[(size_t)l].h; };
    *d_out = nullptr;
    if (plane == AKZ_LXX || plane == AKZ_LYY || plane == AKZ_LXY) {
        for (int k = 0; k < 6; ++k) AKZ_TRY(ensure(c, c->lazy[k], pb));
        float* b[6];
        for (int k = 0; k < 6; ++k) b[k] = (float*)c->lazy[k].p;
        AKZ_TRY(detector_impl(c, img_plane(level, AKZ_LSMOOTH), lv.det_sigma, b[0], b[1], b[2], b[3], b[4], b[5], lv.w, lv.h, 1));
        *d_out = plane == AKZ_LXX ? b[2] : plane == AKZ_LYY ? b[3] : b[4];
        return AKZ_OK;
    }
    if (plane == AKZ_LSTEP && level > 0) {
        const LevelPlan& pv = r->plan[(size_t)level - 1];
        for (int k = 0; k < 4; ++k) AKZ_TRY(ensure(c, c->lazy[k], std::max(pb, (size_t)4)));
        float *A = (float*)c->lazy[0].p, *B = (float*)c->lazy[1].p, *step = (float*)c->lazy[2].p;
        const float* in = img_plane(level - 1, AKZ_LT);
        if (lv.octave > pv.octave) {  // first level of an octave: the 2x2 mean of the previous level's Lt
            launch::half_size(c->stream, in, (float*)c->lazy[3].p, pv.w, pv.h, 1);
            in = (const float*)c->lazy[3].p;
        }
        AKZ_HIP_TRY(hipMemsetAsync(step, 0, pb, c->stream));  // a level without diffusion steps keeps the zero plane (lib.rs:107)
        AKZ_TRY(fed_impl(c, in, A, B, img_plane(level, AKZ_LFLOW), step, lv.w, lv.h, 1, lv.tau.data(), (uint32_t)lv.tau.size()));
        *d_out = step;
        return AKZ_OK;
    }
    return AKZ_OK;  // level 0 has no Lflow / Lstep (0 x 0 in the reference)
}

int akz_fetch_plane(const akz_result* r, uint64_t img, uint64_t level, akz_plane plane, float* out, uint64_t* n_px) {
    const float* d = nullptr;
    AKZ_TRY(akz_result_device_plane(r, img, level, plane, &d));
    const LevelPlan& lv = r->plan[(size_t)level];
    const bool lazy = !d && (plane == AKZ_LXX || plane == AKZ_LYY || plane == AKZ_LXY || (plane == AKZ_LSTEP && level > 0));
    if (lazy && !out) {  // size query
        if (n_px) *n_px = (uint64_t)lv.w * lv.h;
        return AKZ_OK;
    }
    if (lazy) AKZ_TRY(recompute_plane(r, img, level, plane, &d));
    const uint64_t npx = d ? (uint64_t)lv.w * lv.h : 0;
    if (n_px) *n_px = npx;
    if (out && npx) {
        AKZ_TRY(bind(r->ctx));
        AKZ_HIP_TRY(hipMemcpyAsync(out, d, npx * sizeof(float), hipMemcpyDeviceToHost, r->ctx->stream));
        AKZ_HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    }
    return AKZ_OK;
}


}  // extern "C"
