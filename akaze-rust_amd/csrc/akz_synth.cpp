// Deterministic synthetic 8-bit luma frames for benchmarks and parity tests.
// Integer-only, seeded with SplitMix64, so every language binding produces identical bytes.
// This is host utility code (input generation); it is not part of the measured path.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/akaze_hip.h"
#include "../../include/akaze_hip_debug.h"

namespace {
struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};
}  // namespace

extern "C" int akz_synth_frame_u8(uint8_t* out, uint32_t w, uint32_t h, uint64_t frame_index, int32_t shift_x,
                                  int32_t shift_y);

// frame = gradient background blended with W*H/2048 random rectangles/discs, plus +-8 noise.
// (shift_x, shift_y) translates every shape: frame k shifted is the "second view" of frame k.
extern "C" int akz_synth_frame_u8(uint8_t* out, uint32_t w, uint32_t h, uint64_t frame_index, int32_t shift_x,
                                  int32_t shift_y) {
    if (!out || w == 0 || h == 0) return AKZ_ERR_INVALID_ARG;
    SplitMix64 rng{0xA4A2E000ull + frame_index};
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint32_t g = ((37u * (x - (uint32_t)shift_x) + 91u * (y - (uint32_t)shift_y)) >> 5) & 0xFFu;
            out[(size_t)y * w + x] = (uint8_t)((g + 128u) >> 1);
        }
    const uint32_t n_shapes = std::max<uint32_t>(8, (uint32_t)(((uint64_t)w * h) / 2048));
    for (uint32_t i = 0; i < n_shapes; ++i) {
        const int64_t cx = (int64_t)rng.below(w) + shift_x, cy = (int64_t)rng.below(h) + shift_y;
        const int64_t r = 4 + rng.below(125);  // side / radius 4..128
        const uint32_t val = rng.below(256);
        const bool disc = (rng.next() & 1) != 0;
        const int64_t x0 = std::max<int64_t>(0, cx - r), x1 = std::min<int64_t>((int64_t)w - 1, cx + r);
        const int64_t y0 = std::max<int64_t>(0, cy - r), y1 = std::min<int64_t>((int64_t)h - 1, cy + r);
        for (int64_t y = y0; y <= y1; ++y)
            for (int64_t x = x0; x <= x1; ++x) {
                if (disc && (x - cx) * (x - cx) + (y - cy) * (y - cy) > r * r) continue;
                uint8_t& p = out[(size_t)y * w + (size_t)x];
                p = (uint8_t)((p + val + 1u) >> 1);
            }
    }
    // noise is a function of the pixel position and the frame only (not shifted), like sensor noise
    SplitMix64 noise{0x5EED0000ull + frame_index * 0x100000001B3ull};
    for (size_t i = 0; i < (size_t)w * h; i += 8) {
        uint64_t bits = noise.next();
        for (size_t j = i; j < std::min((size_t)w * h, i + 8); ++j, bits >>= 8) {
            const int d = (int)(bits & 0xFF) % 17 - 8;
            const int v = (int)out[j] + d;
            out[j] = (uint8_t)std::min(255, std::max(0, v));
        }
    }
    return AKZ_OK;
}
