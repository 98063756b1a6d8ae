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

}  // namespace

namespace launch {

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
