// Frames for the C++ programs that need pixels with corners: a soft gradient with seeded grey rectangles, shifted by two
// pixels to the right and one down per frame.
#ifndef VSA_GPU_TEST_FRAMES_HPP
#define VSA_GPU_TEST_FRAMES_HPP
#include <cstddef>
#include <cstdint>
#include <vector>

inline std::vector<uint8_t> testFrame(int w, int h, int t, uint32_t seed = 12345u) {
    std::vector<uint8_t> img((std::size_t)w * h);
    const int ox = 2 * t, oy = t;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(std::size_t)y * w + x] = (uint8_t)(64 + ((x - ox) / 16 + (y - oy) / 12) % 32);
    uint32_t s = seed;
    auto next = [&s]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int r = 0; r < 180; r++) {
        const int rw = 12 + (int)(next() % 60), rh = 12 + (int)(next() % 50);
        const int x0 = (int)(next() % (uint32_t)(w + 100)) - 50 + ox, y0 = (int)(next() % (uint32_t)(h + 100)) - 50 + oy;
        const uint8_t g = (uint8_t)(next() % 256);
        for (int y = y0 < 0 ? 0 : y0; y < y0 + rh && y < h; y++)
            for (int x = x0 < 0 ? 0 : x0; x < x0 + rw && x < w; x++) img[(std::size_t)y * w + x] = g;
    }
    return img;
}
#endif
