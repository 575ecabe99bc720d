/*
 * Seeded synthetic inputs for the front-end: the reference's own fixture feeds
 * 640x480 uniform-random frames from an unseeded std::random_device
 * (tests/e2e/e2e_test_fixture.cpp:200-215); these generators keep that shape but are
 * reproducible, and add a textured moving scene with real corners so that detection,
 * tracking and matching have something to find.  Host-only code (no GPU needed); frames
 * are generated once and uploaded before any timed region.
 */
#include <cstdint>
#include <cmath>
#include <vector>
#include <cstring>
#include <algorithm>

namespace {

inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
inline uint32_t hash2(uint32_t seed, uint32_t idx) { return mix32(seed ^ mix32(idx + 0x9E3779B9u)); }

/* Static toroidal world: 4 octaves of value noise (cells 64/32/16/8 px, weights 8:4:2:1)
 * with axis-aligned rectangles of random grey painted on top (one per 1536 px^2, at least 256). */
struct Canvas {
    int cw, ch; std::vector<uint8_t> px;
    Canvas(uint32_t seed, int w, int h) {
        cw = ((w + 63) / 64) * 64; ch = ((h + 63) / 64) * 64;
        px.resize((size_t)cw * ch);
        const int cells[4] = {64, 32, 16, 8}; const int wts[4] = {8, 4, 2, 1};
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++) {
                double acc = 0;
                for (int o = 0; o < 4; o++) {
                    int c = cells[o], gx = cw / c, gy = ch / c;
                    int i0 = x / c, j0 = y / c, i1 = (i0 + 1) % gx, j1 = (j0 + 1) % gy;
                    double fx = (double)(x % c) / c, fy = (double)(y % c) / c;
                    uint32_t s = seed + 0x1000u * (o + 1);
                    double v00 = hash2(s, j0 * gx + i0) & 255, v01 = hash2(s, j0 * gx + i1) & 255;
                    double v10 = hash2(s, j1 * gx + i0) & 255, v11 = hash2(s, j1 * gx + i1) & 255;
                    acc += wts[o] * ((v00 * (1 - fx) + v01 * fx) * (1 - fy) + (v10 * (1 - fx) + v11 * fx) * fy);
                }
                px[(size_t)y * cw + x] = (uint8_t)std::floor(acc / 15.0 + 0.5);
            }
        const int n_rect = std::max(256, cw * ch / 1536); /* same rectangle density at every size */
        for (int r = 0; r < n_rect; r++) {
            uint32_t a = hash2(seed + 0x7777u, 4 * r), b = hash2(seed + 0x7777u, 4 * r + 1),
                     c = hash2(seed + 0x7777u, 4 * r + 2), d = hash2(seed + 0x7777u, 4 * r + 3);
            int x0 = a % cw, y0 = b % ch, rw = 8 + (c & 63), rh = 8 + ((c >> 8) & 63);
            uint8_t g = (uint8_t)(d & 255);
            for (int y = 0; y < rh; y++)
                for (int x = 0; x < rw; x++) px[(size_t)((y0 + y) % ch) * cw + (x0 + x) % cw] = g;
        }
    }
    double at(int x, int y) const {
        x %= cw; if (x < 0) x += cw;
        y %= ch; if (y < 0) y += ch;
        return px[(size_t)y * cw + x];
    }
};

}  // namespace

extern "C" {

/* Uniform noise frame: pixel = low byte of a hash of (seed, linear index). */
void vfe_synth_noise(uint32_t seed, int w, int h, uint8_t* out, int step) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[(size_t)y * step + x] = (uint8_t)(hash2(seed, (uint32_t)(y * w + x)) & 255);
}

/* Frame t of a moving view of the static world: translation (vx, vy) px/frame and rotation
 * rot_deg degrees/frame about the image centre, bilinear sampling, rounded to 8 bits.
 * The flow of frame t -> t+1 is known analytically. */
void vfe_synth_scene(uint32_t seed, int w, int h, int t, double vx, double vy, double rot_deg,
                     uint8_t* out, int step) {
    /* one-entry cache per calling thread: consecutive frames of one sequence share the world */
    static thread_local Canvas* cache = nullptr; static thread_local uint32_t c_seed; static thread_local int c_w, c_h;
    if (!cache || c_seed != seed || c_w != w || c_h != h) {
        delete cache; cache = new Canvas(seed, w, h); c_seed = seed; c_w = w; c_h = h;
    }
    const Canvas& C = *cache;
    const double th = rot_deg * t * (3.14159265358979323846 / 180.0);
    const double cs = std::cos(th), sn = std::sin(th);
    const double cx = (w - 1) * 0.5, cy = (h - 1) * 0.5;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double dx = x - cx, dy = y - cy;
            double u = cx + cs * dx - sn * dy + vx * t, v = cy + sn * dx + cs * dy + vy * t;
            double fu = std::floor(u), fv = std::floor(v);
            int iu = (int)fu, iv = (int)fv;
            double a = u - fu, b = v - fv;
            double val = (C.at(iu, iv) * (1 - a) + C.at(iu + 1, iv) * a) * (1 - b) +
                         (C.at(iu, iv + 1) * (1 - a) + C.at(iu + 1, iv + 1) * a) * b;
            out[(size_t)y * step + x] = (uint8_t)std::floor(val + 0.5);
        }
}

/* n random 256-bit descriptors from the hash stream. */
void vfe_synth_descriptors(uint32_t seed, long long first, long long n, uint8_t* out) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    for (long long i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) o[i * 8 + k] = hash2(seed + (uint32_t)k * 0x01000193u, (uint32_t)(first + i));
}

/* Copy of `src` with `nflip` distinct pseudo-random bits flipped (planted near-duplicates). */
void vfe_synth_flip_bits(uint32_t seed, const uint8_t* src, int nflip, uint8_t* dst) {
    std::memcpy(dst, src, 32);
    uint8_t used[256] = {0};
    int done = 0; uint32_t k = 0;
    while (done < nflip && done < 256) {
        uint32_t b = hash2(seed, k++) & 255;
        if (used[b]) continue;
        used[b] = 1; dst[b >> 3] ^= (uint8_t)(1u << (b & 7)); ++done;
    }
}

}  // extern "C"
