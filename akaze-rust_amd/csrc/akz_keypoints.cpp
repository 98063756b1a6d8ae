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
class SlotGrids {
public:
    SlotGrids(const std::vector<LevelPlan>& plan, const akz_config& cfg, size_t max_slots) {
        const float width = (float)plan[0].w + 16.0f, height = (float)plan[0].h + 16.0f;
        size_t total = 0;
        for (const LevelPlan& lv : plan) {
            G g;
            g.cell = std::max(8.0f, (float)(lv.esigma * cfg.derivative_factor));
            g.nx = std::max(1, (int)std::ceil(width / g.cell) + 1);
            g.ny = std::max(1, (int)std::ceil(height / g.cell) + 1);
            g.base = total;
            total += (size_t)g.nx * g.ny;
            grids_.push_back(g);
        }
        head_.assign(total, -1);
        next_.assign(max_slots, -1);
    }
    void insert(uint32_t level, int32_t slot, float x, float y) {
        int32_t& h = head_[index(level, x, y)];
        next_[slot] = h;
        h = slot;
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
    template <typename F>
    void for_each_near(uint32_t level, float x, float y, float radius, F&& f) const {
        const G& g = grids_[level];
        const int x0 = cx(g, x - radius), x1 = cx(g, x + radius), y0 = cy(g, y - radius), y1 = cy(g, y + radius);
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx)
                for (int32_t s = head_[g.base + (size_t)yy * g.nx + xx]; s != -1; s = next_[s]) f((uint32_t)s);
    }

private:
    struct G {
        float cell;
        int nx, ny;
        size_t base;
    };
    static int cx(const G& g, float x) { return std::min(g.nx - 1, std::max(0, (int)std::floor(x / g.cell))); }
    static int cy(const G& g, float y) { return std::min(g.ny - 1, std::max(0, (int)std::floor(y / g.cell))); }
    size_t index(uint32_t level, float x, float y) const {
        const G& g = grids_[level];
        return g.base + (size_t)cy(g, y) * g.nx + cx(g, x);
    }
    std::vector<G> grids_;
    std::vector<int32_t> head_, next_;
};

}  // namespace

void select_keypoints(const std::vector<Candidate>& cands, const std::vector<LevelPlan>& plan,
                      const akz_config& cfg, std::vector<HostKeypoint>& out, uint64_t* n_extrema) {
    out.clear();
    std::vector<HostKeypoint> cache;
    cache.reserve(cands.size());
    if (plan.empty()) {
        if (n_extrema) *n_extrema = 0;
        return;
    }
    SlotGrids grids(plan, cfg, cands.size() + 1);

    // ---- first pass: scale_space_extrema.rs:43-100 ----
    for (const Candidate& c : cands) {
        const LevelPlan& lv = plan[c.level];
        HostKeypoint kp;
        kp.lx = c.idx % lv.w;
        kp.ly = c.idx / lv.w;
        kp.response = std::fabs(c.v);
        kp.size = (float)(lv.esigma * cfg.derivative_factor);
        kp.octave = lv.octave;
        kp.class_id = c.level;
        kp.x = (float)kp.lx;
        kp.y = (float)kp.ly;
        kp.angle = 0.0f;
        kp.xp = c.xp; kp.xm = c.xm; kp.yp = c.yp; kp.ym = c.ym;
        const float ratio = powf(2.0f, (float)lv.octave);
        const float qx = kp.x * ratio, qy = kp.y * ratio;
        const float size2 = kp.size * kp.size;
        // first (lowest-index) cache entry on this or the previous level within `size`
        uint32_t hit = UINT32_MAX;
        auto visit = [&](uint32_t s) {
            if (s >= hit) return;
            const HostKeypoint& p = cache[s];
            const float dist = (qx - p.x) * (qx - p.x) + (qy - p.y) * (qy - p.y);
            if (dist <= size2) hit = s;
        };
        grids.for_each_near(c.level, qx, qy, kp.size + 1.0f, visit);
        if (c.level > 0) grids.for_each_near(c.level - 1, qx, qy, kp.size + 1.0f, visit);
        bool is_repeated = false, is_extremum = true;
        if (hit != UINT32_MAX) {
            if (kp.response > cache[hit].response) is_repeated = true;
            else is_extremum = false;
        }
        if (!is_extremum) continue;
        // (the border test already ran on the device)
        kp.x = kp.x * ratio + 0.5f * (ratio - 1.0f);
        kp.y = kp.y * ratio + 0.5f * (ratio - 1.0f);
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

    // ---- second pass: drop points repeated on the next level LATER in the cache (:109-129) ----
    std::vector<HostKeypoint> extrema;
    extrema.reserve(cache.size());
    for (uint32_t i = 0; i < cache.size(); ++i) {
        const HostKeypoint& a = cache[i];
        bool repeated = false;
        if ((size_t)a.class_id + 1 < plan.size()) {
            const float size2 = a.size * a.size;
            grids.for_each_near(a.class_id + 1, a.x, a.y, a.size + 1.0f, [&](uint32_t s) {
                if (repeated || s < i) return;
                const HostKeypoint& b = cache[s];
                const float dist = (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y);
                if (dist <= size2) repeated = true;
            });
        }
        if (!repeated) extrema.push_back(a);
    }
    if (n_extrema) *n_extrema = extrema.size();

    // ---- "sub-pixel" step (:141-178): the LU solve result is discarded by the reference ----
    for (const HostKeypoint& k : extrema) {
        const float ratio = powf(2.0f, (float)k.octave);
        const float fx = std::round(k.x / ratio), fy = std::round(k.y / ratio);
        const uint32_t x = fx > 0.0f ? (uint32_t)fx : 0u, y = fy > 0.0f ? (uint32_t)fy : 0u;
        (void)x; (void)y;  // equal k.lx, k.ly: p*ratio + 0.5(ratio-1) rounds back to p
        const float d_x = 0.5f * (k.xp - k.xm);
        const float d_y = 0.5f * (k.yp - k.ym);
        const float b0 = -d_x, b1 = -d_y;
        if (std::fabs(b0) <= 1.0f && std::fabs(b1) <= 1.0f) {
            HostKeypoint r = k;
            r.x = ((float)k.lx + b0) * ratio + 0.5f * (ratio - 1.0f);
            r.y = ((float)k.ly + b1) * ratio + 0.5f * (ratio - 1.0f);
            out.push_back(r);
        }
    }
}

}  // namespace akz
