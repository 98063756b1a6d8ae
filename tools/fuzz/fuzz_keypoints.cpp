// Sanitizer + cross-check driver for the host keypoint selection (akz_keypoints.cpp): random candidate sets with dense
// clusters, exact ties and repeated positions are put into scan order by sort_candidates and run through
// select_keypoints (uniform grids, counting sort), and the result is compared with a direct restatement of the
// reference's linear scans (scale_space_extrema.rs:43-129, :141-178).  CPU build only; built and run by
// tests/test_fuzz_host.py with -fsanitize=address,undefined.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../include/akaze_hip.h"

#include "../../akaze-rust_amd/csrc/akz_internal.hpp"
#include "../../akaze-rust_amd/csrc/akz_select.hpp"

using namespace akz;

// the reference's two passes and the sub-pixel step with plain linear scans over the cache
static void naive(const std::vector<Candidate>& cands, const std::vector<LevelPlan>& plan, const akz_config& cfg,
                  std::vector<HostKeypoint>& out, uint64_t* n_extrema) {
    std::vector<HostKeypoint> cache;
    for (const Candidate& c : cands) {
        const LevelPlan& lv = plan[c.level];
        HostKeypoint kp{};
        kp.lx = c.idx % lv.w;
        kp.ly = c.idx / lv.w;
        kp.response = std::fabs(c.v);
        kp.size = (float)(lv.esigma * cfg.derivative_factor);
        kp.octave = lv.octave;
        kp.class_id = c.level;
        kp.x = (float)kp.lx;
        kp.y = (float)kp.ly;
        kp.xp = c.xp; kp.xm = c.xm; kp.yp = c.yp; kp.ym = c.ym;
        const float ratio = powf(2.0f, (float)lv.octave);
        bool is_extremum = true, is_repeated = false;
        size_t id_repeated = 0;
        for (size_t ik = 0; ik < cache.size(); ++ik) {
            const HostKeypoint& p = cache[ik];
            if (kp.class_id == p.class_id || (kp.class_id > 0 && kp.class_id - 1 == p.class_id)) {
                const float dist = (kp.x * ratio - p.x) * (kp.x * ratio - p.x) + (kp.y * ratio - p.y) * (kp.y * ratio - p.y);
                if (dist <= kp.size * kp.size) {
                    if (kp.response > p.response) { id_repeated = ik; is_repeated = true; }
                    else is_extremum = false;
                    break;
                }
            }
        }
        if (!is_extremum) continue;
        kp.x = kp.x * ratio + 0.5f * (ratio - 1.0f);
        kp.y = kp.y * ratio + 0.5f * (ratio - 1.0f);
        if (!is_repeated) cache.push_back(kp);
        else cache[id_repeated] = kp;
    }
    std::vector<HostKeypoint> extrema;
    for (size_t i = 0; i < cache.size(); ++i) {
        bool repeated = false;
        const HostKeypoint& a = cache[i];
        for (size_t j = i; j < cache.size(); ++j) {
            const HostKeypoint& b = cache[j];
            if (a.class_id + 1 == b.class_id) {
                const float dist = (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y);
                if (dist <= a.size * a.size) { repeated = true; break; }
            }
        }
        if (!repeated) extrema.push_back(a);
    }
    *n_extrema = extrema.size();
    out.clear();
    for (const HostKeypoint& k : extrema) {
        const float ratio = powf(2.0f, (float)k.octave);
        const float b0 = -(0.5f * (k.xp - k.xm)), b1 = -(0.5f * (k.yp - k.ym));
        if (std::fabs(b0) <= 1.0f && std::fabs(b1) <= 1.0f) {
            HostKeypoint r = k;
            r.x = ((float)k.lx + b0) * ratio + 0.5f * (ratio - 1.0f);
            r.y = ((float)k.ly + b1) * ratio + 0.5f * (ratio - 1.0f);
            out.push_back(r);
        }
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    std::mt19937 rng(20261003);
    auto uni = [&](int lo, int hi) { return (int)(rng() % (unsigned)(hi - lo + 1)) + lo; };
    long total_c = 0, total_k = 0, rel_rounds = 0, overflowed = 0, dep_rounds = 0, max_rounds = 0;
    for (int it = 0; it < rounds; ++it) {
        akz_config cfg{};  // Config::default() (types/evolution.rs:40-55)
        cfg.num_sublevels = 4; cfg.max_octave_evolution = 4; cfg.base_scale_offset = 1.6; cfg.initial_contrast = 0.001;
        cfg.contrast_percentile = 0.7; cfg.contrast_factor_num_bins = 300; cfg.derivative_factor = 1.5;
        cfg.detector_threshold = 0.001; cfg.descriptor_channels = 3; cfg.descriptor_pattern_size = 10;
        if (it % 5 == 4) { cfg.num_sublevels = 5; cfg.max_octave_evolution = 5; }
        const uint32_t w = (uint32_t)uni(64, 700), h = (uint32_t)uni(64, 500);
        std::vector<LevelPlan> plan;
        if (build_plan(w, h, cfg, plan) != AKZ_OK) continue;
        // candidates: a few cluster centres in full-resolution coordinates, every level draws points around them
        // (every other round sparser: lists that fit the device's neighbour lists, which the dependency rounds below need)
        const bool sparse = it % 2 == 1;
        const int n_centres = uni(1, 12), per_level = sparse ? uni(0, 120) : uni(0, 400);
        std::vector<std::pair<float, float>> centres;
        for (int c = 0; c < n_centres; ++c) centres.emplace_back((float)uni(0, (int)w - 1), (float)uni(0, (int)h - 1));
        std::vector<Candidate> cands;
        for (size_t l = 0; l < plan.size(); ++l) {
            const LevelPlan& lv = plan[l];
            const float ratio = (float)(1u << lv.octave);
            for (int k = 0; k < per_level; ++k) {
                const auto& ce = centres[(size_t)uni(0, n_centres - 1)];
                const int spread = sparse ? uni(8, 160) : (uni(0, 3) == 0 ? 40 : 4);
                const int x = std::min<int>((int)lv.w - 1, std::max(0, (int)(ce.first / ratio) + uni(-spread, spread)));
                const int y = std::min<int>((int)lv.h - 1, std::max(0, (int)(ce.second / ratio) + uni(-spread, spread)));
                Candidate c{};
                c.level = (uint32_t)l;
                c.idx = (uint32_t)(y * (int)lv.w + x);
                c.v = (float)uni(1, 6) * 0.25f;  // few distinct responses: many exact ties
                c.xp = (float)uni(-3, 3) * 0.5f; c.xm = (float)uni(-3, 3) * 0.5f;
                c.yp = (float)uni(-3, 3) * 0.5f; c.ym = (float)uni(-3, 3) * 0.5f;
                cands.push_back(c);
            }
        }
        // one candidate per (level, pixel), as the extrema kernels emit them
        std::sort(cands.begin(), cands.end(), [](const Candidate& a, const Candidate& b) {
            return a.level != b.level ? a.level < b.level : a.idx < b.idx;
        });
        cands.erase(std::unique(cands.begin(), cands.end(), [](const Candidate& a, const Candidate& b) {
            return a.level == b.level && a.idx == b.idx;
        }), cands.end());
        std::vector<Candidate> shuffled = cands;
        std::shuffle(shuffled.begin(), shuffled.end(), rng);
        sort_candidates(shuffled, plan);
        for (size_t i = 0; i < cands.size(); ++i)
            if (shuffled[i].level != cands[i].level || shuffled[i].idx != cands[i].idx || shuffled[i].v != cands[i].v) {
                fprintf(stderr, "sort_candidates differs at %zu (round %d)\n", i, it);
                return 1;
            }
        std::vector<HostKeypoint> got, exp;
        uint64_t ne_got = 0, ne_exp = 0;
        select_keypoints(shuffled.data(), shuffled.size(), plan, cfg, got, &ne_got);
        naive(cands, plan, cfg, exp, &ne_exp);
        if (ne_got != ne_exp || got.size() != exp.size()) {
            fprintf(stderr, "round %d: %zu/%llu keypoints/extrema, expected %zu/%llu\n", it, got.size(),
                    (unsigned long long)ne_got, exp.size(), (unsigned long long)ne_exp);
            return 1;
        }
        for (size_t i = 0; i < got.size(); ++i)
            if (got[i].x != exp[i].x || got[i].y != exp[i].y || got[i].response != exp[i].response ||
                got[i].class_id != exp[i].class_id || got[i].size != exp[i].size || got[i].octave != exp[i].octave) {
                fprintf(stderr, "round %d: keypoint %zu differs\n", it, i);
                return 1;
            }
        // the same selection from neighbour lists (select_keypoints_rel; on the GPU box they come from k_relations): who
        // lies within `size` of whom, by brute force with the selection's own expressions
        {
            const int K1 = kRel1, K2 = kRel2;  // the device's list lengths: dense clusters overflow them, which marks the candidate
                                               // (0xfffe) for the host's own scan of its partner levels
            const size_t n = cands.size();
            std::vector<uint16_t> rel(n * (size_t)(K1 + K2), 0xffffu);
            bool overflow = n > 65533;
            std::vector<uint8_t> o1(n, 0), o2(n, 0);
            std::vector<float> px(n), py(n), qx(n), qy(n), sz(n);
            for (size_t i = 0; i < n; ++i) {
                const LevelPlan& lv = plan[cands[i].level];
                const float ratio = powf(2.0f, (float)lv.octave);
                const uint32_t ly = cands[i].idx / lv.w, lx = cands[i].idx - ly * lv.w;
                qx[i] = (float)lx * ratio; qy[i] = (float)ly * ratio;
                px[i] = qx[i] + 0.5f * (ratio - 1.0f); py[i] = qy[i] + 0.5f * (ratio - 1.0f);
                sz[i] = (float)(lv.esigma * cfg.derivative_factor);
            }
            for (size_t i = 0; i < n && !overflow; ++i) {
                int u1 = 0, u2 = 0;
                for (size_t j = 0; j < n; ++j) {
                    if (j == i) continue;
                    const uint32_t li = cands[i].level, lj = cands[j].level;
                    if ((lj == li && j < i) || (li > 0 && lj == li - 1)) {
                        const float d = (qx[i] - px[j]) * (qx[i] - px[j]) + (qy[i] - py[j]) * (qy[i] - py[j]);
                        if (d <= sz[i] * sz[i]) {
                            if (u1 < K1) rel[i * (size_t)(K1 + K2) + (size_t)u1++] = (uint16_t)j;
                            else o1[i] = 1;
                        }
                    } else if (lj == li + 1) {
                        const float d = (px[i] - px[j]) * (px[i] - px[j]) + (py[i] - py[j]) * (py[i] - py[j]);
                        if (d <= sz[i] * sz[i]) {
                            if (u2 < K2) rel[i * (size_t)(K1 + K2) + (size_t)K1 + (size_t)u2++] = (uint16_t)j;
                            else o2[i] = 1;
                        }
                    }
                }
            }
            long overflowed_now = 0;
            for (size_t i = 0; i < n; ++i) {
                if (o1[i]) rel[i * (size_t)(K1 + K2)] = 0xfffeu;
                if (o2[i]) rel[i * (size_t)(K1 + K2) + (size_t)K1] = 0xfffeu;
                overflowed_now += o1[i] + o2[i];
            }
            overflowed += overflowed_now;
            if (!overflow) {
                std::vector<HostKeypoint> got2;
                uint64_t ne2 = 0;
                select_keypoints_rel(cands.data(), n, rel.data(), K1, K2, plan, cfg, got2, &ne2);
                if (ne2 != ne_exp || got2.size() != exp.size()) {
                    fprintf(stderr, "round %d (neighbour lists): %zu/%llu keypoints/extrema, expected %zu/%llu\n", it, got2.size(),
                            (unsigned long long)ne2, exp.size(), (unsigned long long)ne_exp);
                    return 1;
                }
                for (size_t i = 0; i < got2.size(); ++i)
                    if (got2[i].x != exp[i].x || got2[i].y != exp[i].y || got2[i].response != exp[i].response ||
                        got2[i].class_id != exp[i].class_id || got2[i].size != exp[i].size || got2[i].octave != exp[i].octave ||
                        got2[i].lx != exp[i].lx || got2[i].ly != exp[i].ly) {
                        fprintf(stderr, "round %d (neighbour lists): keypoint %zu differs\n", it, i);
                        return 1;
                    }
                ++rel_rounds;
                // the same selection as dependency rounds (akz_select.hpp; on the GPU box akz_select.hip runs them): reverse lists,
                // ranks, then rounds in which every ready candidate takes its turn -- here in a random order with every write
                // visible at once, which is the worst interleaving a workgroup could produce
                bool dev_ok = overflowed_now == 0;
                std::vector<std::vector<uint16_t>> rev(n);
                for (size_t i = 0; i < n && dev_ok; ++i)
                    for (int j = 0; j < K1; ++j) {
                        const uint16_t q = rel[i * (size_t)(K1 + K2) + (size_t)j];
                        if (q == sel::kNone) break;
                        rev[q].push_back((uint16_t)i);
                    }
                for (size_t i = 0; i < n; ++i) dev_ok = dev_ok && rev[i].size() <= (size_t)sel::kRev;
                if (dev_ok) {
                    // rows as k_sel_prepare forms them, then the turns as k_select takes them: `threads` owners, each walking its
                    // candidates (c % threads == owner) in index order, interleaved at random, every write visible at once
                    typedef sel::Row<kRel1, kRel2> Row;
                    std::vector<Row> rows(n);
                    std::vector<uint16_t> word(n, sel::kUndecided);
                    const size_t threads = (size_t)uni(1, 96);
                    std::vector<std::vector<uint16_t>> own(threads);
                    size_t undecided = 0;
                    for (size_t i = 0; i < n; ++i) {
                        const uint16_t* r1 = &rel[i * (size_t)(K1 + K2)];
                        Row& rw = rows[i];
                        rw.wins = 0;
                        unsigned len = 0;
                        for (int j = 0; j < K1; ++j) {
                            rw.rel1[j] = r1[j];
                            rw.pred[j] = sel::kNone;
                            if (r1[j] == sel::kNone) continue;
                            ++len;
                            for (uint16_t m : rev[r1[j]])
                                if (m < i && (rw.pred[j] == sel::kNone || m > rw.pred[j])) rw.pred[j] = m;
                            if (std::fabs(cands[i].v) > std::fabs(cands[r1[j]].v)) rw.wins |= (uint16_t)(1u << j);
                        }
                        for (int j = 0; j < K2; ++j) rw.rel2[j] = r1[K1 + j];
                        rw.refined = (uint16_t)(len << 8);
                        if (r1[0] == sel::kNone) word[i] = (uint16_t)i;
                        else { own[i % threads].push_back((uint16_t)i); ++undecided; }
                    }
                    auto WORD = [&](uint16_t q) { return word[q]; };
                    std::vector<size_t> at_of(threads, 0);
                    std::vector<sel::Progress> prog(threads);
                    for (auto& pr : prog) sel::start(&pr);
                    long looks = 0, idle = 0;
                    while (undecided) {
                        const size_t t = (size_t)uni(0, (int)threads - 1);
                        if (at_of[t] >= own[t].size()) continue;
                        const uint16_t c = own[t][at_of[t]];
                        const Row& rw = rows[c];
                        const int len = rw.refined >> 8;
                        ++looks;
                        // one neighbour per look (the device goes on while they are through: any interleaving in between is allowed)
                        bool moved = false;
                        if (prog[t].through < len) {
                            const int j = prog[t].through;
                            const bool has_pred = rw.pred[j] != sel::kNone;
                            const uint16_t wp = has_pred ? word[rw.pred[j]] : (uint16_t)0;
                            moved = sel::advance(&prog[t], has_pred, wp, word[rw.rel1[j]]);
                        }
                        if (prog[t].through < len) {
                            if (!moved && ++idle > 4000000) { fprintf(stderr, "round %d: the turns stalled with %zu candidates\n", it, undecided); return 1; }
                            continue;
                        }
                        idle = 0;
                        bool replaces;
                        const uint16_t mine = sel::turn(prog[t], c, rw.wins, &replaces);
                        if (replaces) word[rw.rel1[prog[t].at]] = sel::kGone;
                        word[c] = mine;
                        ++at_of[t];
                        sel::start(&prog[t]);
                        --undecided;
                    }
                    max_rounds = std::max(max_rounds, looks);
                    // second pass, refinement, cache order = order of the positions' creators
                    std::vector<int32_t> at(n, -1);
                    uint64_t ne3 = 0;
                    for (size_t k = 0; k < n; ++k) {
                        if (!sel::alive(word[k])) continue;
                        if (sel::repeated_later<kRel2>(word[k], rows[k].rel2, WORD)) continue;
                        ++ne3;
                        at[word[k]] = (int32_t)k;
                    }
                    std::vector<sel::KpRec> got3;
                    for (size_t o = 0; o < n; ++o) {
                        if (at[o] < 0) continue;
                        const Candidate& cd = cands[(size_t)at[o]];
                        const LevelPlan& lv = plan[cd.level];
                        sel::KpRec rec{};
                        rec.level = cd.level;
                        if (sel::refine(cd.idx % lv.w, cd.idx / lv.w, cd.v, cd.xp, cd.xm, cd.yp, cd.ym, powf(2.0f, (float)lv.octave), &rec)) got3.push_back(rec);
                    }
                    if (ne3 != ne_exp || got3.size() != exp.size()) {
                        fprintf(stderr, "round %d (dependency rounds): %zu/%llu keypoints/extrema, expected %zu/%llu\n", it, got3.size(),
                                (unsigned long long)ne3, exp.size(), (unsigned long long)ne_exp);
                        return 1;
                    }
                    for (size_t i = 0; i < got3.size(); ++i)
                        if (got3[i].x != exp[i].x || got3[i].y != exp[i].y || got3[i].response != exp[i].response || got3[i].level != exp[i].class_id) {
                            fprintf(stderr, "round %d (dependency rounds): keypoint %zu differs\n", it, i);
                            return 1;
                        }
                    ++dep_rounds;
                }
            }
        }
        total_c += (long)cands.size();
        total_k += (long)got.size();
    }
    printf("selected %ld keypoints from %ld candidates, identical to the linear scans (%ld rounds also through the neighbour lists, %ld overflowed lists; "
           "%ld rounds also as the device's data flow of turns, at most %ld looks)\n", total_k, total_c, rel_rounds, overflowed, dep_rounds, max_rounds);
    return 0;
}
