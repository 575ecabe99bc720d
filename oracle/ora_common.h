/*
 * CPU oracle for the visual front-end -- TEST INFRASTRUCTURE ONLY.
 *
 * This directory restates, on the CPU and with no dependencies, the arithmetic
 * that the reference delegates to OpenCV (imgproc / features2d / video) at
 *   src/slam/adapters/openvins_adapter.cpp:70-80   (cvtColor + goodFeaturesToTrack)
 *   src/slam/adapters/basalt_adapter.cpp:42-53     (cvtColor + goodFeaturesToTrack)
 *   src/slam/adapters/orbslam3_adapter.cpp:69-79   (cvtColor + ORB::detectAndCompute)
 * plus the stages the reference only names in its configuration
 *   config/openvins.yaml:25-39 (use_klt, fast_threshold, grid_x/y, min_px_dist, num_klt_levels)
 *   config/basalt.yaml:85-108  (pyramid_levels, patch_size)
 *   config/vins_mono.yaml:91-104 (max_cnt, min_dist, equalize)
 * i.e. pyrDown pyramids, pyramidal Lucas-Kanade, FAST and brute-force Hamming kNN.
 *
 * PARITY UNPINNED: OpenCV (the reference's pinned dependency, 4.5.x via
 * CMakeLists.txt:32) is not present in this build environment and the reference
 * holds no golden vectors for this path, so this restatement follows the published
 * OpenCV 4.5 algorithms and is itself the normative checker for the HIP kernels.
 *
 * Nothing in the shipped library may include, link or call this code; it is used
 * by tests/, by the smoke check and by the benchmark's cpu_baseline leg.
 *
 * All floating-point code here must be compiled with -ffp-contract=off: the HIP
 * kernels are built the same way, so every float expression has one rounding per
 * operation on both sides and results can be compared bit for bit.
 */
#ifndef VFE_ORACLE_COMMON_H
#define VFE_ORACLE_COMMON_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>

namespace ora {

/* cvRound: round half to even (OpenCV uses cvtsd2si / lrint). */
static inline int cv_round(double v) { return (int)std::lrint(v); }
static inline int cv_roundf(float v) { return (int)std::lrintf(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_floorf(float v) { int i = (int)v; return i - ((float)i > v); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

/* BORDER_REFLECT_101:  ... c b | a b c ... y z | y x ...  */
static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - 2 - p;
    }
    return p;
}

struct Img {
    const uint8_t* d; int w, h, step;
    uint8_t at(int y, int x) const { return d[(size_t)y * step + x]; }
    uint8_t at_r(int y, int x) const { return d[(size_t)reflect101(y, h) * step + reflect101(x, w)]; }
};

/* murmur3 finaliser, the hash behind every synthetic input in this repository */
static inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}

/* Deterministic cos/sin of a float angle in radians (|x| <= ~7), evaluated in double
 * with a fixed operation order and rounded once to float.  Used for the steered
 * BRIEF rotation so that the CPU oracle and the GPU kernel produce the same bits;
 * agrees with (float)cos((double)x) except for 1-in-1e8 double-rounding cases. */
static inline void sincos_det(float xf, float* s_out, float* c_out) {
    const double x = (double)xf;
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11; /* pi/2 - pio2_hi */
    const double kd = std::nearbyint(x * two_over_pi);
    const int k = (int)kd;
    double r = x - kd * pio2_hi;
    r = r - kd * pio2_lo;
    const double z = r * r;
    /* Taylor/minimax coefficients (fdlibm __kernel_sin / __kernel_cos) */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S6;
    ps = ps * z + S5; ps = ps * z + S4; ps = ps * z + S3; ps = ps * z + S2; ps = ps * z + S1;
    const double sn = r + (r * z) * ps;
    double pc = C6;
    pc = pc * z + C5; pc = pc * z + C4; pc = pc * z + C3; pc = pc * z + C2; pc = pc * z + C1;
    const double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s; *c_out = (float)c;
}

}  // namespace ora
#endif
