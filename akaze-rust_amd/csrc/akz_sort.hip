// Extrema candidates into the reference's scan order ON THE DEVICE (scale_space_extrema.rs:32-42 walks the levels in
// order and each level's pixels in raster order; the detector kernels append their candidates to one list in whatever
// order their waves finish).  Key = (image, level, flat pixel index), packed into as few bits as the batch needs; a
// device radix sort of (key, list position) pairs and one gather leave the list sorted by image, level and pixel, so
// that the host neither buckets nor sorts: with the host cores of a node shared by 8 ranks those two passes were a fifth
// of the host work of a batch (the other four fifths, the order-dependent selection, cannot move).
// The sort is rocPRIM's device radix sort (a library primitive); keys are unique, so the result is deterministic.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "akz_internal.hpp"

namespace akz {
namespace {

struct KeyBits {
    unsigned idx_bits, level_bits, total;
};
__host__ __device__ inline unsigned bits_for(unsigned long long n) {  // bits that hold 0 .. n-1
    unsigned b = 1;
    while (b < 63 && (1ull << b) < n) ++b;
    return b;
}

// list entries beyond the count get the largest key: they sort to the end
__global__ void k_candidate_keys(const Candidate* __restrict__ cand, unsigned cap, const unsigned* __restrict__ d_count,
                                 KeyBits kb, unsigned long long* __restrict__ keys, unsigned* __restrict__ pos) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const unsigned n = min(*d_count, cap);
    unsigned long long k = ~0ull >> (64 - kb.total);
    if (i < n) {
        const Candidate c = cand[i];
        k = ((unsigned long long)c.img << (kb.idx_bits + kb.level_bits)) | ((unsigned long long)c.level << kb.idx_bits) | c.idx;
    }
    keys[i] = k;
    pos[i] = i;
}
__global__ void k_candidate_gather(const Candidate* __restrict__ cand, unsigned cap, const unsigned* __restrict__ d_count,
                                   const unsigned* __restrict__ pos, Candidate* __restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_count, cap)) return;
    out[i] = cand[pos[i]];
}

// ---- who can meet whom in the keypoint selection (round 4) ------------------------------------------------------------
// find_scale_space_extrema (scale_space_extrema.rs:43-129) asks, for every candidate in scan order, for the FIRST cache
// entry on its own or the previous level within `size` of it (:57-76), and afterwards drops the entries that have a LATER
// entry of the next level within `size` (:109-129).  Which entries are in the cache at that moment is order-dependent and
// stays on the host; WHO can be within `size` of whom is pure geometry, the same f32 expressions on every candidate pair,
// and is answered here for the whole batch: per candidate the (at most kRel1) candidates of its own level before it and
// of the previous level that lie within its `size` of it, and the (at most kRel2) candidates of the NEXT level within
// `size` of its stored position.  The host's selection then walks these short lists on three small per-image arrays
// (alive, slot, response) instead of querying a 2 MB spatial grid per image: the same answers (tests), a quarter of the
// host time.  Indices are relative to the image's first candidate (u16; 0xffff = none).  A candidate with more neighbours
// than a list holds (coarse levels: `size` spans several level pixels there, and those levels have few candidates) gets
// 0xfffe in the list's first place and the host scans that candidate's partner levels itself; an image with more than
// 65 533 candidates is flagged and takes the grid path.
struct RelLevels {  // per level, as select_keypoints computes them
    float size[kMaxLevels], ratio[kMaxLevels];
    unsigned w[kMaxLevels];
    unsigned n_levels, n_images;
};
// first list position of every (image, level): offs[img * (L + 1) + level], level == L: the image's end
__global__ void k_rel_offsets(const Candidate* __restrict__ sorted, const unsigned* __restrict__ d_count, unsigned cap, unsigned n_levels,
                              unsigned n_images, unsigned* __restrict__ offs) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_images * (n_levels + 1)) return;
    const unsigned img = t / (n_levels + 1), level = t - img * (n_levels + 1);
    const unsigned n = min(*d_count, cap);
    unsigned lo = 0, hi = n;  // first entry with (img', level') >= (img, level)
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        const Candidate c = sorted[mid];
        const bool before = c.img < img || (c.img == img && c.level < level);
        if (before) lo = mid + 1;
        else hi = mid;
    }
    offs[t] = lo;
}
__device__ __forceinline__ unsigned rel_lower_bound(const Candidate* __restrict__ s, unsigned lo, unsigned hi, unsigned idx) {
    while (lo < hi) {  // first entry of [lo, hi) with .idx >= idx
        const unsigned mid = (lo + hi) >> 1;
        if (s[mid].idx < idx) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__global__ void __launch_bounds__(256) k_relations(const Candidate* __restrict__ sorted, const unsigned* __restrict__ d_count, unsigned cap,
                                                   RelLevels lv, const unsigned* __restrict__ offs, unsigned short* __restrict__ rel,
                                                   unsigned* __restrict__ img_flags) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_count, cap)) return;
    const Candidate c = sorted[i];
    if (c.img >= lv.n_images || c.level >= lv.n_levels) return;
    const unsigned L = lv.n_levels, l = c.level;
    const unsigned* o = offs + (size_t)c.img * (L + 1);
    const unsigned img0 = o[0];
    unsigned short* out = rel + (size_t)i * (kRel1 + kRel2);
#pragma unroll
    for (int k = 0; k < kRel1 + kRel2; ++k) out[k] = 0xffffu;
    const bool too_many = o[L] - img0 > 65533u;
    bool over1 = false, over2 = false;
    const float size = lv.size[l], size2 = size * size, ratio = lv.ratio[l];
    const unsigned ly = c.idx / lv.w[l], lx = c.idx - ly * lv.w[l];
    const float qx = (float)lx * ratio, qy = (float)ly * ratio;                  // the query point of the first pass (:52-53 of the host code)
    const float px = qx + 0.5f * (ratio - 1.0f), py = qy + 0.5f * (ratio - 1.0f);  // the candidate's stored position
    // partners of level pl in [b, e): rows whose stored y can lie within `reach` of y0 -- a conservative band, the exact
    // test decides
    auto scan = [&](unsigned pl, unsigned b, unsigned e, float x0, float y0, int first, int room, int& used, bool& overflow) {
        if (b >= e) return;
        const float pr = lv.ratio[pl], off = 0.5f * (pr - 1.0f);
        const unsigned pw = lv.w[pl];
        const float reach = size + 1.0f;
        const float ylo = (y0 - reach - off) / pr - 1.0f, yhi = (y0 + reach - off) / pr + 1.0f;
        const unsigned r0 = ylo <= 0.0f ? 0u : (unsigned)ylo;
        if (yhi < 0.0f) return;
        const unsigned long long last = (unsigned long long)((unsigned)yhi + 1u) * pw;
        unsigned j = rel_lower_bound(sorted, b, e, r0 * pw);
        for (; j < e; ++j) {
            const Candidate p = sorted[j];
            if ((unsigned long long)p.idx >= last) break;
            const unsigned py_i = p.idx / pw, px_i = p.idx - py_i * pw;
            const float sx = (float)px_i * pr + off, sy = (float)py_i * pr + off;  // its stored position
            const float dist = (x0 - sx) * (x0 - sx) + (y0 - sy) * (y0 - sy);
            if (dist <= size2) {
                if (used < room) out[first + used] = (unsigned short)(j - img0);
                else overflow = true;
                ++used;
            }
        }
    };
    int used1 = 0, used2 = 0;
    if (l > 0) scan(l - 1, o[l - 1], o[l], qx, qy, 0, kRel1, used1, over1);   // every entry of the previous level comes before this one
    scan(l, o[l], i, qx, qy, 0, kRel1, used1, over1);                          // its own level: the ones before it
    if (l + 1 < L) scan(l + 1, o[l + 1], o[l + 2], px, py, kRel1, kRel2, used2, over2);
    if (over1) out[0] = 0xfffeu;
    if (over2) out[kRel1] = 0xfffeu;
    if (too_many) img_flags[c.img] = 1u;
}

}  // namespace

namespace launch {

size_t candidate_relations_bytes(uint32_t cap, uint32_t n_levels, uint32_t n_images) {
    const size_t offs = ((size_t)n_images * (n_levels + 1) * 4 + 255) / 256 * 256, flags = ((size_t)n_images * 4 + 255) / 256 * 256;
    return offs + flags + (size_t)cap * (kRel1 + kRel2) * sizeof(uint16_t) + 256;
}
// d_rel_out / d_flags_out: where the lists and the per-image overflow flags are inside `scratch`
void candidate_relations(hipStream_t s, const Candidate* d_sorted, uint32_t cap, const uint32_t* d_count, const float* size, const float* ratio,
                         const uint32_t* level_w, uint32_t n_levels, uint32_t n_images, void* scratch, uint16_t** d_rel_out,
                         uint32_t** d_flags_out) {
    const size_t offs_b = ((size_t)n_images * (n_levels + 1) * 4 + 255) / 256 * 256, flags_b = ((size_t)n_images * 4 + 255) / 256 * 256;
    unsigned* offs = (unsigned*)scratch;
    unsigned* flags = (unsigned*)((char*)scratch + offs_b);
    unsigned short* rel = (unsigned short*)((char*)scratch + offs_b + flags_b);
    *d_rel_out = rel;
    *d_flags_out = flags;
    RelLevels lv;
    std::memset(&lv, 0, sizeof(lv));
    for (uint32_t l = 0; l < n_levels && l < (uint32_t)kMaxLevels; ++l) {
        lv.size[l] = size[l];
        lv.ratio[l] = ratio[l];
        lv.w[l] = level_w[l];
    }
    lv.n_levels = n_levels;
    lv.n_images = n_images;
    (void)hipMemsetAsync(flags, 0, (size_t)n_images * 4, s);
    const unsigned no = n_images * (n_levels + 1);
    hipLaunchKernelGGL(k_rel_offsets, dim3((no + 255) / 256), dim3(256), 0, s, d_sorted, d_count, cap, n_levels, n_images, offs);
    hipLaunchKernelGGL(k_relations, dim3((cap + 255) / 256), dim3(256), 0, s, d_sorted, d_count, cap, lv, offs, rel, flags);
}

// scratch layout: keys in | keys out | positions in | positions out | rocPRIM's temporary storage
static size_t sort_temp(uint32_t cap, unsigned end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, cap, 0, end_bit, nullptr);
    return bytes;
}
static KeyBits key_bits(uint64_t max_px, uint32_t n_levels, uint32_t n_images) {
    KeyBits kb;
    kb.idx_bits = bits_for(max_px);
    kb.level_bits = bits_for(n_levels);
    kb.total = kb.idx_bits + kb.level_bits + bits_for((unsigned long long)n_images + 1);  // + 1: the all-ones padding key
    return kb;
}
size_t sort_candidates_scratch(uint32_t cap, uint64_t max_px, uint32_t n_levels, uint32_t n_images) {
    const KeyBits kb = key_bits(max_px, n_levels, n_images);
    const size_t a = ((size_t)cap * 8 + 255) / 256 * 256, b = ((size_t)cap * 4 + 255) / 256 * 256;
    return 2 * a + 2 * b + sort_temp(cap, kb.total) + 256;
}
bool sort_candidates_device(hipStream_t s, const Candidate* d_cand, uint32_t cap, const uint32_t* d_count, uint64_t max_px,
                            uint32_t n_levels, uint32_t n_images, void* scratch, Candidate* d_sorted) {
    if (cap == 0) return true;
    const KeyBits kb = key_bits(max_px, n_levels, n_images);
    if (kb.total > 63) return false;
    const size_t a = ((size_t)cap * 8 + 255) / 256 * 256, b = ((size_t)cap * 4 + 255) / 256 * 256;
    char* p = (char*)scratch;
    unsigned long long* k_in = (unsigned long long*)p;
    unsigned long long* k_out = (unsigned long long*)(p + a);
    unsigned* p_in = (unsigned*)(p + 2 * a);
    unsigned* p_out = (unsigned*)(p + 2 * a + b);
    void* tmp = p + 2 * a + 2 * b;
    size_t tmp_bytes = sort_temp(cap, kb.total);
    const unsigned nb = (cap + 255) / 256;
    hipLaunchKernelGGL(k_candidate_keys, dim3(nb), dim3(256), 0, s, d_cand, cap, d_count, kb, k_in, p_in);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, p_in, p_out, cap, 0, kb.total, s) != hipSuccess) return false;
    hipLaunchKernelGGL(k_candidate_gather, dim3(nb), dim3(256), 0, s, d_cand, cap, d_count, p_out, d_sorted);
    return true;
}

}  // namespace launch
}  // namespace akz
