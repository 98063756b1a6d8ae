// The order-dependent keypoint selection (scale_space_extrema.rs:43-129) as a data flow of turns -- the pieces that the
// device kernel (akz_sort.hip, k_select) and the CPU cross-check (tools/fuzz/fuzz_keypoints.cpp) share.
//
// The reference walks the candidates in scan order and keeps a cache: a candidate looks for the FIRST cache entry of its
// own or the previous level within `size` of it; if there is one it either replaces that entry in place (larger response)
// or is dropped, otherwise it is appended.  Afterwards entries that have a LATER entry of the next level within `size` are
// dropped.  Two observations turn the walk into something a GPU can do:
//   * "first" and "later" compare cache positions, and a cache position is created by exactly one candidate (an in-place
//     replacement inherits it): positions are ordered like the candidates that created them.  The creator of the position
//     a candidate sits in replaces the running counter: no global state is left, and one 16-bit word per candidate holds
//     its whole state.
//   * what candidate c sees of a neighbour q (k_relations' lists, rel1(c)) is q's own outcome plus whether an EARLIER
//     candidate that also has q in its list replaced q before c's turn.  With rev(q) = {c' : q in rel1(c')}, c's view of q
//     is final as soon as q and every member of rev(q) before c have had their turns -- and the members of rev(q) wait for
//     each other in index order, so "every earlier member" is "the member right before c": pred(c, q).
// A neighbour that is through in this sense stays through, so a candidate waits for its neighbours one after the other and
// folds each one's contribution (is it alive, where is its cache position) into a running minimum the moment it is; when
// the last one is through it takes its turn.  Two candidates whose turns can come at the same time never share a neighbour
// (both would be members of its rev list, and the later one waits for the earlier), so their reads and writes do not meet,
// and the lowest undecided candidate of the image can always go: no barrier, no rounds.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define AKZ_SEL_HD __host__ __device__ __forceinline__
#define AKZ_SEL_UNROLL _Pragma("unroll")
#define AKZ_SEL_KEEP(v) __asm__ volatile("" : "+v"(v))  // the value is wanted HERE: its load is not sunk into a later branch
#else
#define AKZ_SEL_HD inline
#define AKZ_SEL_UNROLL
#define AKZ_SEL_KEEP(v) (void)0
#endif

namespace akz {
namespace sel {

constexpr int kRev = 12;  // members of a reverse list the device keeps (more: the image goes to the host's selection)
constexpr uint16_t kNone = 0xffffu, kListOverflow = 0xfffeu;
// A candidate's word says it all: kUndecided until its turn; afterwards the creator of the cache position it sits in (a
// candidate index: below kGone) while it is in the cache, kGone if it was dropped or has been replaced.  Written at the
// candidate's own turn and by the one member of its reverse list that replaces it, never from two sides at once.
constexpr uint16_t kUndecided = 0xffffu, kGone = 0xfffeu;

struct KpRec {  // a selected keypoint as the host needs it (size and octave follow from the level)
    float x, y, response;
    uint32_t level;
};

// what a candidate's turn needs, gathered once (k_sel_prepare): its earlier neighbours (rel1, the first `len` places), for each
// the member of its reverse list right before this candidate (pred, kNone: this is the first), whether its response beats
// each neighbour's (bit j of wins), its next-level neighbours (second pass)
template <int K1, int K2>
struct alignas(16) Row {
    uint16_t rel1[K1];
    uint16_t pred[K1];
    uint16_t wins;
    uint16_t rel2[K2];
    uint16_t refined;  // bit 0: passes the refinement's gradient test (:141-178); bits 8 .. 11: entries of rel1
};

AKZ_SEL_HD bool alive(uint16_t word) { return word < kGone; }

// A candidate's progress through its list (scale_space_extrema.rs:57-99): how many neighbours are through, and of those the
// live one with the first cache position (at: its place in the list, -1: none yet)
struct Progress {
    int through;
    int at;
    uint16_t hit;  // the creator of that cache position
};
AKZ_SEL_HD void start(Progress* p) { p->through = 0; p->at = -1; p->hit = kNone; }
// Is neighbour `through` through?  word_pred MUST have been read before word_q (the predecessor stores "q is gone" before its
// own word).  If so the neighbour's contribution is folded in and the function returns true.
AKZ_SEL_HD bool advance(Progress* p, bool has_pred, uint16_t word_pred, uint16_t word_q) {
    if (word_q == kUndecided || (has_pred && word_pred == kUndecided)) return false;
    if (alive(word_q) && word_q < p->hit) {  // (the creators of live entries are distinct)
        p->hit = word_q;
        p->at = p->through;
    }
    ++p->through;
    return true;
}
// the turn, once every neighbour is through: the candidate's word, and whether the neighbour at p.at is replaced (the
// caller stores kGone into that neighbour's word first, then the candidate's word)
AKZ_SEL_HD uint16_t turn(const Progress& p, uint16_t c, uint16_t wins, bool* replaces) {
    *replaces = false;
    if (p.at < 0) return c;  // a cache position of its own
    if ((wins >> p.at) & 1u) {
        *replaces = true;
        return p.hit;        // takes the neighbour's position over
    }
    return kGone;            // not an extremum
}

// second pass (:109-129): a live entry of the next level, at or after this entry's cache position, within `size`
template <int K2, class W>
AKZ_SEL_HD bool repeated_later(uint16_t mine, const uint16_t* rel2, W word) {
    bool rep = false, open = true;
    AKZ_SEL_UNROLL
    for (int j = 0; j < K2; ++j) {
        const uint16_t q = rel2[j];
        open = open && q != kNone;
        if (open) {
            const uint16_t w = word(q);
            if (alive(w) && w >= mine) rep = true;
        }
    }
    return rep;
}

// the "sub-pixel" step (:141-178; only the gradient test and the shift survive in the reference) and the record / kernel
// parameters of a survivor.  lx, ly: level coordinates; ratio = 2^octave; size = esigma * derivative_factor (f32)
AKZ_SEL_HD bool refine(uint32_t lx, uint32_t ly, float v, float xp, float xm, float yp, float ym, float ratio, KpRec* rec) {
    const float d_x = 0.5f * (xp - xm), d_y = 0.5f * (yp - ym);
    const float b0 = -d_x, b1 = -d_y;
    if (!(__builtin_fabsf(b0) <= 1.0f && __builtin_fabsf(b1) <= 1.0f)) return false;
    rec->x = ((float)lx + b0) * ratio + 0.5f * (ratio - 1.0f);
    rec->y = ((float)ly + b1) * ratio + 0.5f * (ratio - 1.0f);
    rec->response = __builtin_fabsf(v);
    return true;
}

}  // namespace sel
}  // namespace akz
