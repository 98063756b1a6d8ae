// Host part of keypoint selection: the order-dependent scalar logic of
// akaze/src/ops/scale_space_extrema.rs that follows the NMS kernel.  The kernel hands over
// candidates that already passed threshold + 4-neighbour maximum + descriptor-border test; the
// host sorts them into the reference's raster order and replays the sequential cache logic.
#include <algorithm>
#include <cmath>

#include "akz_internal.hpp"

namespace akz {

// smax * sigma_size of the border test (scale_space_extrema.rs:14, :52, :80-83), all f32
float border_margin(const LevelPlan& lv, const akz_config& cfg) {
    const float smax = 10.0f * std::sqrt(2.0f);
    const float size = (float)(lv.esigma * cfg.derivative_factor);
    const float ratio = powf(2.0f, (float)lv.octave);
    const float sigma_size = std::round(size / ratio);
    return smax * sigma_size;
}

// The sliding-window loop of compute_main_orientation (scale_space_extrema.rs:298-328) replayed
// for a sample whose angle is atan2f(a, a), a > 0 — the only angle value that can ever fall in a
// window (the others are <= 0).  Bit i of mask = window i takes the samples.
void orientation_windows(unsigned long long* mask, uint32_t* n_windows) {
    volatile float one = 1.0f;  // keep atan2f a run-time libm call
    const float ang = atan2f(one, one);
    const float pi = 3.14159265358979323846f;
    unsigned long long m = 0;
    uint32_t n = 0;
    float ang1 = 0.0f;
    while (ang1 < 2.0f * pi && n < 64) {
        const float ang2 = (ang1 + pi / 3.0f > 2.0f * pi) ? (ang1 - 5.0f * pi / 3.0f) : (ang1 + pi / 3.0f);
        ang1 += 0.15f;
        const bool in = (ang1 < ang2 && ang1 < ang && ang < ang2) ||
                        (ang2 < ang1 && ((ang > 0.0f && ang < ang2) || (ang > ang1 && ang < 2.0f * pi)));
        if (in) m |= 1ull << n;
        ++n;
    }
    *mask = m;
    *n_windows = n;
}

namespace {

// Uniform grids (one per level) over full-resolution coordinates holding cache slots as
// intrusive singly linked lists (head per cell, next per slot; no per-cell allocation).  They let
// the "first cache entry within `size` on level e or e-1" query of the reference (a linear scan,
// :57-76) be answered from a few cells.  The answer is the MINIMUM slot index among qualifying
// entries, which is exactly what the reference's front-to-back scan with `break` returns.
// A level's cells are twice the largest query radius it ever sees (its own entries are looked up by
// candidates of the same and of the next level, radius = their size + 1), so a query touches at
// most 2 x 2 cells; the grid keeps its own copy of every slot's position (8 bytes) so that the
// distance test does not pull the 56-byte cache entries through the cache.
class SlotGrids {
    struct G {
        float cell, inv;
        int nx, ny;
        size_t base;
    };
    struct P {
        float x, y;
    };

public:
    SlotGrids(const std::vector<LevelPlan>& plan, const akz_config& cfg, size_t max_slots) {
        const float width = (float)plan[0].w + 16.0f, height = (float)plan[0].h + 16.0f;
        size_t total = 0;
        for (size_t l = 0; l < plan.size(); ++l) {
            const LevelPlan& up = plan[std::min(l + 1, plan.size() - 1)];
            G g;
            // (strictly wider than the widest query -- radius = size(l+1) + 1 -- so that the rounding of x - radius and
            // x + radius can never spread a query over three cells; cells() still checks)
            g.cell = std::max(16.0f, 2.0f * ((float)(up.esigma * cfg.derivative_factor) + 1.0f) * 1.001f);
            g.inv = 1.0f / g.cell;
            g.nx = std::max(1, (int)std::ceil(width / g.cell) + 1);
            g.ny = std::max(1, (int)std::ceil(height / g.cell) + 1);
            g.base = total;
            total += (size_t)g.nx * g.ny;
            grids_.push_back(g);
        }
        // the arrays live in per-thread buffers that keep their capacity between calls (a 4K frame needs ~2 MB of
        // cell heads; fresh allocations of that size are page-faulted in on every call).  The cell heads are all -1
        // between calls: the destructor resets the cells this call touched (a few thousand) instead of every call
        // clearing all of them (135 K for a 1080p frame: a sixth of the selection's time)
        static thread_local std::vector<int32_t> head_buf, next_buf;
        static thread_local std::vector<P> pos_buf;
        static thread_local std::vector<uint32_t> touched_buf;
        if (head_buf.size() < total) head_buf.resize(total, -1);
        if (next_buf.size() < max_slots) next_buf.resize(max_slots);
        if (pos_buf.size() < max_slots) pos_buf.resize(max_slots);
        touched_buf.clear();
        head_ = head_buf.data();
        next_ = next_buf.data();
        pos_ = pos_buf.data();
        touched_ = &touched_buf;
    }
    ~SlotGrids() {
        for (uint32_t cell : *touched_) head_[cell] = -1;
    }
    SlotGrids(const SlotGrids&) = delete;
    SlotGrids& operator=(const SlotGrids&) = delete;
    void insert(uint32_t level, int32_t slot, float x, float y) {
        const size_t cell = index(level, x, y);
        int32_t& h = head_[cell];
        if (h == -1) touched_->push_back((uint32_t)cell);  // (a cell emptied by remove() and filled again is listed twice: harmless)
        next_[slot] = h;
        h = slot;
        pos_[slot] = P{x, y};
    }
    void remove(uint32_t level, int32_t slot, float x, float y) {
        int32_t* link = &head_[index(level, x, y)];
        while (*link != -1) {
            if (*link == slot) {
                *link = next_[slot];
                return;
            }
            link = &next_[*link];
        }
    }
    // lowest slot index below `best` whose position is within sqrt(d2max) of (x, y); `best` if none
    uint32_t min_slot_within(uint32_t level, float x, float y, float radius, float d2max, uint32_t best) const {
        const G& g = grids_[level];
        const int x0 = cx(g, x - radius), x1 = cx(g, x + radius), y0 = cy(g, y - radius), y1 = cy(g, y + radius);
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx)
                for (int32_t s = head_[g.base + (size_t)yy * g.nx + xx]; s != -1; s = next_[s]) {
                    if ((uint32_t)s >= best) continue;
                    const P p = pos_[s];
                    const float dist = (x - p.x) * (x - p.x) + (y - p.y) * (y - p.y);
                    if (dist <= d2max) best = (uint32_t)s;
                }
        return best;
    }
    // The cells a query of (x, y, radius) walks on `level`, computed ahead of the walk (select_keypoints does it for a
    // block of candidates at a time: straight-line code the compiler vectorises, and the cell heads can be prefetched):
    // first cell + whether the range spans a second column / row.
    struct Cells {
        uint32_t first;
        uint32_t dx, dy;  // 0 / 1 and 0 / nx
        bool wide;        // the range spans more than 2 x 2 cells (cannot happen with the cell sizes above): general walk
    };
    Cells cells(uint32_t level, float x, float y, float radius) const {
        const G& g = grids_[level];
        const int x0 = cx(g, x - radius), x1 = cx(g, x + radius), y0 = cy(g, y - radius), y1 = cy(g, y + radius);
        const bool wide = x1 - x0 > 1 || y1 - y0 > 1;
        return Cells{(uint32_t)(g.base + (size_t)y0 * g.nx + x0), (uint32_t)std::min(1, x1 - x0), (uint32_t)(std::min(1, y1 - y0) * g.nx), wide};
    }
    void prefetch(const Cells& c) const {
        __builtin_prefetch(head_ + c.first);
        __builtin_prefetch(head_ + c.first + c.dy);
    }
    uint32_t min_slot_in(const Cells& c, float x, float y, float d2max, uint32_t best) const {
        const int32_t h[4] = {head_[c.first], head_[c.first + c.dx], head_[c.first + c.dy], head_[c.first + c.dy + c.dx]};
        if ((h[0] & h[1] & h[2] & h[3]) == -1) return best;
        for (int k = 0; k < 4; ++k) {
            if (((k & 1) && !c.dx) || ((k & 2) && !c.dy)) continue;  // the range spans one column / row: the cell repeats
            for (int32_t s = h[k]; s != -1; s = next_[s]) {
                if ((uint32_t)s >= best) continue;
                const P p = pos_[s];
                const float dist = (x - p.x) * (x - p.x) + (y - p.y) * (y - p.y);
                if (dist <= d2max) best = (uint32_t)s;
            }
        }
        return best;
    }
    // is there a slot >= from within sqrt(d2max) of (x, y)?
    bool any_slot_from(uint32_t level, float x, float y, float radius, float d2max, uint32_t from) const {
        const G& g = grids_[level];
        const int x0 = cx(g, x - radius), x1 = cx(g, x + radius), y0 = cy(g, y - radius), y1 = cy(g, y + radius);
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx)
                for (int32_t s = head_[g.base + (size_t)yy * g.nx + xx]; s != -1; s = next_[s]) {
                    if ((uint32_t)s < from) continue;
                    const P p = pos_[s];
                    const float dist = (x - p.x) * (x - p.x) + (y - p.y) * (y - p.y);
                    if (dist <= d2max) return true;
                }
        return false;
    }

private:
    // the cell of a coordinate only has to be monotone and identical for insert and query: x * (1 / cell) will do
    static int cx(const G& g, float x) { return std::min(g.nx - 1, std::max(0, (int)(x * g.inv))); }
    static int cy(const G& g, float y) { return std::min(g.ny - 1, std::max(0, (int)(y * g.inv))); }
    size_t index(uint32_t level, float x, float y) const {
        const G& g = grids_[level];
        return g.base + (size_t)cy(g, y) * g.nx + cx(g, x);
    }
    std::vector<G> grids_;
    int32_t *head_ = nullptr, *next_ = nullptr;
    P* pos_ = nullptr;
    std::vector<uint32_t>* touched_ = nullptr;
};

}  // namespace

// Candidates into the reference's scan order (level, then flat index w*y + x): a counting sort over (level, row)
// buckets followed by an insertion sort inside each row, which holds a handful of candidates.  About 4x faster
// than std::sort on the 32-byte records (levels and indices must have been validated against the plan).
void sort_candidates(std::vector<Candidate>& c, const std::vector<LevelPlan>& plan) {
    if (c.size() < 2) return;
    std::vector<uint32_t> row0(plan.size() + 1, 0);
    for (size_t l = 0; l < plan.size(); ++l) row0[l + 1] = row0[l] + plan[l].h;
    std::vector<uint32_t> start(row0.back() + 1, 0);
    std::vector<uint32_t> bucket(c.size());
    for (size_t i = 0; i < c.size(); ++i) {
        bucket[i] = row0[c[i].level] + c[i].idx / plan[c[i].level].w;
        ++start[bucket[i] + 1];
    }
    for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
    std::vector<Candidate> out(c.size());
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (size_t i = 0; i < c.size(); ++i) out[fill[bucket[i]]++] = c[i];
    for (size_t b = 0; b + 1 < start.size(); ++b)
        for (uint32_t i = start[b] + 1; i < start[b + 1]; ++i) {
            const Candidate v = out[i];
            uint32_t j = i;
            for (; j > start[b] && out[j - 1].idx > v.idx; --j) out[j] = out[j - 1];
            out[j] = v;
        }
    c.swap(out);
}

void select_keypoints(const Candidate* cands, size_t n_cands, const std::vector<LevelPlan>& plan,
                      const akz_config& cfg, std::vector<HostKeypoint>& out, uint64_t* n_extrema) {
    out.clear();
    static thread_local std::vector<HostKeypoint> cache_tl;  // keeps its capacity between calls (~0.4 MB per 1080p frame)
    std::vector<HostKeypoint>& cache = cache_tl;              // (one TLS lookup, not one per access: this is a shared library)
    cache.clear();
    cache.reserve(n_cands);
    if (plan.empty()) {
        if (n_extrema) *n_extrema = 0;
        return;
    }
    SlotGrids grids(plan, cfg, n_cands + 1);

    // per-level constants of the loop below (the same expressions, evaluated once per level)
    struct LevelConst {
        float size, ratio;
    };
    std::vector<LevelConst> lc(plan.size());
    for (size_t l = 0; l < plan.size(); ++l)
        lc[l] = LevelConst{(float)(plan[l].esigma * cfg.derivative_factor), powf(2.0f, (float)plan[l].octave)};

    // ---- first pass: scale_space_extrema.rs:43-100 ----
    // Candidates arrive sorted by level.  Per block of a level's candidates, first everything that does not depend on
    // the cache (coordinates, the grid cells of the two look-ups; their heads are prefetched), then the sequential logic.
    constexpr size_t kBlock = 64;
    float bqx[kBlock], bqy[kBlock];
    uint32_t blx[kBlock], bly[kBlock];
    SlotGrids::Cells bc0[kBlock], bc1[kBlock];
    for (size_t c0 = 0; c0 < n_cands;) {
        const uint32_t level = cands[c0].level;
        size_t nb = 1;
        while (nb < kBlock && c0 + nb < n_cands && cands[c0 + nb].level == level) ++nb;
        const LevelPlan& lv = plan[level];
        const float size = lc[level].size, ratio = lc[level].ratio, radius = size + 1.0f, size2 = size * size;
        const uint32_t w = lv.w;
        const double inv_w = 1.0 / (double)w;
        for (size_t j = 0; j < nb; ++j) {
            const uint32_t idx = cands[c0 + j].idx;
            uint32_t ry = (uint32_t)((double)idx * inv_w);  // idx / w without the integer division: exact after the fix-up
            if (ry * w > idx) --ry;
            else if ((ry + 1) * w <= idx) ++ry;
            bly[j] = ry;
            blx[j] = idx - ry * w;
            bqx[j] = (float)blx[j] * ratio;
            bqy[j] = (float)bly[j] * ratio;
            bc0[j] = grids.cells(level, bqx[j], bqy[j], radius);
            grids.prefetch(bc0[j]);
            if (level > 0) {
                bc1[j] = grids.cells(level - 1, bqx[j], bqy[j], radius);
                grids.prefetch(bc1[j]);
            }
        }
        for (size_t j = 0; j < nb; ++j) {
            const Candidate& c = cands[c0 + j];
            const float response = std::fabs(c.v);
            // first (lowest-index) cache entry on this or the previous level within `size`
            uint32_t hit = bc0[j].wide ? grids.min_slot_within(level, bqx[j], bqy[j], radius, size2, UINT32_MAX)
                                       : grids.min_slot_in(bc0[j], bqx[j], bqy[j], size2, UINT32_MAX);
            if (level > 0)
                hit = bc1[j].wide ? grids.min_slot_within(level - 1, bqx[j], bqy[j], radius, size2, hit)
                                  : grids.min_slot_in(bc1[j], bqx[j], bqy[j], size2, hit);
            bool is_repeated = false;
            if (hit != UINT32_MAX) {
                if (response > cache[hit].response) is_repeated = true;
                else continue;  // not an extremum
            }
            // (the border test already ran on the device)
            HostKeypoint kp;
            kp.lx = blx[j];
            kp.ly = bly[j];
            kp.response = response;
            kp.size = size;
            kp.octave = lv.octave;
            kp.class_id = level;
            kp.x = (float)kp.lx * ratio + 0.5f * (ratio - 1.0f);
            kp.y = (float)kp.ly * ratio + 0.5f * (ratio - 1.0f);
            kp.angle = 0.0f;
            kp.xp = c.xp; kp.xm = c.xm; kp.yp = c.yp; kp.ym = c.ym;
            if (!is_repeated) {
                cache.push_back(kp);
                grids.insert(kp.class_id, (int32_t)(cache.size() - 1), kp.x, kp.y);
            } else {
                const HostKeypoint old = cache[hit];
                grids.remove(old.class_id, (int32_t)hit, old.x, old.y);
                cache[hit] = kp;
                grids.insert(kp.class_id, (int32_t)hit, kp.x, kp.y);
            }
        }
        c0 += nb;
    }

    // ---- second pass: drop points repeated on the next level LATER in the cache (:109-129), and on the survivors
    // the "sub-pixel" step (:141-178; the LU solve result is discarded by the reference) ----
    uint64_t extrema = 0;
    out.reserve(cache.size());
    for (uint32_t i = 0; i < cache.size(); ++i) {
        const HostKeypoint& k = cache[i];
        const bool repeated = (size_t)k.class_id + 1 < plan.size() &&
                              grids.any_slot_from(k.class_id + 1, k.x, k.y, k.size + 1.0f, k.size * k.size, i);
        if (repeated) continue;
        ++extrema;
        // the reference re-derives the level coordinates as round(k.x / ratio): p*ratio + 0.5(ratio-1) rounds back
        // to p, i.e. to k.lx, k.ly
        const float ratio = lc[k.class_id].ratio;
        const float d_x = 0.5f * (k.xp - k.xm);
        const float d_y = 0.5f * (k.yp - k.ym);
        const float b0 = -d_x, b1 = -d_y;
        if (std::fabs(b0) <= 1.0f && std::fabs(b1) <= 1.0f) {
            out.push_back(k);
            HostKeypoint& r = out.back();
            r.x = ((float)k.lx + b0) * ratio + 0.5f * (ratio - 1.0f);
            r.y = ((float)k.ly + b1) * ratio + 0.5f * (ratio - 1.0f);
        }
    }
    if (n_extrema) *n_extrema = extrema;
}

void selection_level_constants(const std::vector<LevelPlan>& plan, const akz_config& cfg, std::vector<float>& size, std::vector<float>& ratio) {
    size.resize(plan.size());
    ratio.resize(plan.size());
    for (size_t l = 0; l < plan.size(); ++l) {
        size[l] = (float)(plan[l].esigma * cfg.derivative_factor);
        ratio[l] = powf(2.0f, (float)plan[l].octave);
    }
}

// The same selection as select_keypoints (scale_space_extrema.rs:43-178) from the device's neighbour lists: rel holds, per
// candidate, kRel1 candidates of its own (earlier) / the previous level within `size` of its query point and kRel2
// candidates of the next level within `size` of its stored position (launch::candidate_relations; indices into this
// image's list, 0xffff = none).  A cache entry is represented by the candidate that occupies it: alive[c], slot[c]; a
// replacement in place (:70-76 / :95-99) hands the slot to the new candidate.  "First cache entry within size" = the
// lowest slot among the alive neighbours; "a later entry of the next level within size" = an alive next-level neighbour
// with a slot >= the entry's own.
void select_keypoints_rel(const Candidate* cands, size_t n, const uint16_t* rel, int k1, int k2, const std::vector<LevelPlan>& plan,
                          const akz_config& cfg, std::vector<HostKeypoint>& out, uint64_t* n_extrema) {
    out.clear();
    if (n_extrema) *n_extrema = 0;
    if (plan.empty() || n == 0) return;
    static thread_local std::vector<uint32_t> slot_tl, occ_tl;
    static thread_local std::vector<uint8_t> alive_tl;
    std::vector<uint32_t>&slot = slot_tl, &occ = occ_tl;
    std::vector<uint8_t>& alive = alive_tl;
    slot.assign(n, 0);
    alive.assign(n, 0);
    occ.clear();
    occ.reserve(n);
    const size_t K = (size_t)k1 + (size_t)k2;
    // the rare candidate whose list overflowed (0xfffe): its partner levels are scanned here, with the selection's own
    // expressions.  first[l] = the image's first candidate of level l (the list is sorted by level)
    std::vector<uint32_t> first(plan.size() + 2, (uint32_t)n);
    {
        uint32_t at = 0;
        for (size_t l = 0; l <= plan.size(); ++l) {
            while (at < n && cands[at].level < l) ++at;
            first[l] = at;
        }
        first[plan.size() + 1] = (uint32_t)n;
    }
    // per level: 2^octave and esigma * derivative_factor (the expressions of scale_space_extrema.rs:52-60 / :150-170, evaluated
    // once per level instead of once per keypoint: powf was a fifth of this function on a 4K image)
    std::vector<float> ratio_l, size_l;
    selection_level_constants(plan, cfg, size_l, ratio_l);
    struct Pos { float qx, qy, px, py, size2; };
    auto pos_of = [&](size_t c) {
        const LevelPlan& lv = plan[cands[c].level];
        const float ratio = ratio_l[cands[c].level], size = size_l[cands[c].level];
        const uint32_t ly = cands[c].idx / lv.w, lx = cands[c].idx - ly * lv.w;
        Pos p;
        p.qx = (float)lx * ratio; p.qy = (float)ly * ratio;
        p.px = p.qx + 0.5f * (ratio - 1.0f); p.py = p.qy + 0.5f * (ratio - 1.0f);
        p.size2 = size * size;
        return p;
    };
    for (size_t c = 0; c < n; ++c) {
        const uint16_t* r = rel + c * K;
        uint32_t hit = UINT32_MAX, hc = 0;
        if (r[0] == 0xfffeu) {
            const uint32_t l = cands[c].level;
            const Pos me = pos_of(c);
            for (size_t q = l > 0 ? first[l - 1] : first[l]; q < c; ++q) {
                if (!alive[q] || slot[q] >= hit) continue;
                const Pos o = pos_of(q);
                const float dist = (me.qx - o.px) * (me.qx - o.px) + (me.qy - o.py) * (me.qy - o.py);
                if (dist <= me.size2) {
                    hit = slot[q];
                    hc = (uint32_t)q;
                }
            }
        } else {
            for (int j = 0; j < k1 && r[j] != 0xffffu; ++j) {
                const uint32_t q = r[j];
                if (alive[q] && slot[q] < hit) {
                    hit = slot[q];
                    hc = q;
                }
            }
        }
        if (hit != UINT32_MAX) {
            if (std::fabs(cands[c].v) > std::fabs(cands[hc].v)) {  // replaces the entry in place
                alive[hc] = 0;
                alive[c] = 1;
                slot[c] = hit;
                occ[hit] = (uint32_t)c;
            }
            continue;  // (else: not an extremum)
        }
        alive[c] = 1;
        slot[c] = (uint32_t)occ.size();
        occ.push_back((uint32_t)c);
    }
    uint64_t extrema = 0;
    out.reserve(occ.size());
    for (uint32_t i = 0; i < occ.size(); ++i) {
        const uint32_t k = occ[i];
        const uint16_t* r = rel + (size_t)k * K + k1;
        bool repeated = false;
        if (r[0] == 0xfffeu) {
            const uint32_t l = cands[k].level;
            const Pos me = pos_of(k);
            for (size_t q = first[l + 1]; q < first[l + 2] && !repeated; ++q) {
                if (!alive[q] || slot[q] < i) continue;
                const Pos o = pos_of(q);
                const float dist = (me.px - o.px) * (me.px - o.px) + (me.py - o.py) * (me.py - o.py);
                repeated = dist <= me.size2;
            }
        } else {
            for (int j = 0; j < k2 && r[j] != 0xffffu; ++j) {
                const uint32_t q = r[j];
                if (alive[q] && slot[q] >= i) {
                    repeated = true;
                    break;
                }
            }
        }
        if (repeated) continue;
        ++extrema;
        const Candidate& cd = cands[k];
        const LevelPlan& lv = plan[cd.level];
        const float ratio = ratio_l[cd.level];
        const uint32_t ly = cd.idx / lv.w, lx = cd.idx - ly * lv.w;
        const float d_x = 0.5f * (cd.xp - cd.xm), d_y = 0.5f * (cd.yp - cd.ym);
        const float b0 = -d_x, b1 = -d_y;
        if (std::fabs(b0) <= 1.0f && std::fabs(b1) <= 1.0f) {
            HostKeypoint kp;
            kp.lx = lx;
            kp.ly = ly;
            kp.response = std::fabs(cd.v);
            kp.size = size_l[cd.level];
            kp.octave = lv.octave;
            kp.class_id = cd.level;
            kp.angle = 0.0f;
            kp.xp = cd.xp; kp.xm = cd.xm; kp.yp = cd.yp; kp.ym = cd.ym;
            kp.x = ((float)lx + b0) * ratio + 0.5f * (ratio - 1.0f);
            kp.y = ((float)ly + b1) * ratio + 0.5f * (ratio - 1.0f);
            out.push_back(kp);
        }
    }
    if (n_extrema) *n_extrema = extrema;
}

}  // namespace akz
