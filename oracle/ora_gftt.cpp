/* CPU oracle (test infrastructure only): Shi-Tomasi corner detection.
 * Restates cv::goodFeaturesToTrack / cv::cornerMinEigenVal (OpenCV 4.5) as called at
 * src/slam/adapters/openvins_adapter.cpp:79-80 and src/slam/adapters/basalt_adapter.cpp:50-53
 * (mask none, blockSize 3, gradientSize 3, useHarrisDetector false). */
#include "ora_common.h"
#include "vfe_oracle.h"

using namespace ora;

/* Sobel 3x3 with scale 1/(4*3*255) folded into the smoothing tap, 3x3 un-normalised box
 * sum of the gradient products (horizontal 3-sum, then vertical 3-sum, reflect-101 on the
 * product planes), then the smaller eigenvalue.  One float rounding per operation. */
extern "C" void ora_min_eigen_map(const uint8_t* img, int w, int h, int step, float* eig) {
    Img I{img, w, h, step};
    const double scale = 1.0 / ((1 << 2) * 3 * 255.0);
    const float k_o = (float)scale, k_c = (float)(2.0 * scale);
    std::vector<float> cxx((size_t)w * h), cxy((size_t)w * h), cyy((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            /* Dx: row pass [-1 0 1] (exact), column pass k_c*r1 + k_o*(r0 + r2) */
            float rx[3], ry[3];
            for (int k = 0; k < 3; k++) {
                int yy = reflect101(y + k - 1, h);
                int a = I.at(yy, reflect101(x - 1, w)), b = I.at(yy, x), c = I.at(yy, reflect101(x + 1, w));
                rx[k] = (float)(c - a);
                /* Dy: row pass k_c*b + k_o*(a + c) */
                ry[k] = k_c * (float)b + k_o * (float)(a + c);
            }
            float dx = k_c * rx[1] + k_o * (rx[0] + rx[2]);
            float dy = ry[2] - ry[0];
            size_t i = (size_t)y * w + x;
            cxx[i] = dx * dx; cxy[i] = dx * dy; cyy[i] = dy * dy;
        }
    std::vector<float> hx((size_t)w * h), hxy((size_t)w * h), hy((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t l = (size_t)y * w + reflect101(x - 1, w), c = (size_t)y * w + x, r = (size_t)y * w + reflect101(x + 1, w);
            hx[c] = (cxx[l] + cxx[c]) + cxx[r];
            hxy[c] = (cxy[l] + cxy[c]) + cxy[r];
            hy[c] = (cyy[l] + cyy[c]) + cyy[r];
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t t = (size_t)reflect101(y - 1, h) * w + x, c = (size_t)y * w + x, b = (size_t)reflect101(y + 1, h) * w + x;
            float sxx = (hx[t] + hx[c]) + hx[b];
            float sxy = (hxy[t] + hxy[c]) + hxy[b];
            float syy = (hy[t] + hy[c]) + hy[b];
            float a = sxx * 0.5f, bb = sxy, cc = syy * 0.5f;
            float d = a - cc;
            eig[c] = (a + cc) - std::sqrt(d * d + bb * bb);
        }
}

static int select_corners(std::vector<float>& eig, int w, int h, int max_corners, double quality, double min_distance,
                          const float* seeds, int n_seed, float* corners_xy, float* scores, int cap);

extern "C" int ora_good_features(const uint8_t* img, int w, int h, int step,
                                 int max_corners, double quality, double min_distance,
                                 const float* seeds, int n_seed,
                                 float* corners_xy, float* scores, int cap) {
    std::vector<float> eig((size_t)w * h);
    ora_min_eigen_map(img, w, h, step, eig.data());
    return select_corners(eig, w, h, max_corners, quality, min_distance, seeds, n_seed, corners_xy, scores, cap);
}

/* the same call on the row-streaming eigenvalue map (ora_opt.cpp): identical results, less memory traffic */
extern "C" int ora_good_features_opt(const uint8_t* img, int w, int h, int step,
                                     int max_corners, double quality, double min_distance,
                                     const float* seeds, int n_seed,
                                     float* corners_xy, float* scores, int cap) {
    std::vector<float> eig((size_t)w * h);
    ora_min_eigen_map_opt(img, w, h, step, eig.data());
    return select_corners(eig, w, h, max_corners, quality, min_distance, seeds, n_seed, corners_xy, scores, cap);
}

static int select_corners(std::vector<float>& eig, int w, int h, int max_corners, double quality, double min_distance,
                          const float* seeds, int n_seed, float* corners_xy, float* scores, int cap) {
    float max_val = eig[0];
    for (size_t i = 1; i < eig.size(); i++) max_val = std::max(max_val, eig[i]);
    /* threshold(eig, eig, maxVal*quality, 0, THRESH_TOZERO): double product, compared as float */
    const float thr = (float)((double)max_val * quality);
    for (auto& v : eig) v = (v > thr) ? v : 0.f;
    /* 3x3 dilate; neighbours outside the image are ignored */
    std::vector<const float*> cand;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            float v = eig[(size_t)y * w + x];
            if (v == 0.f) continue;
            float m = v;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) m = std::max(m, eig[(size_t)(y + dy) * w + (x + dx)]);
            if (v == m) cand.push_back(&eig[(size_t)y * w + x]);
        }
    /* greaterThanPtr: value descending, ties by larger address first */
    std::sort(cand.begin(), cand.end(), [](const float* a, const float* b) {
        return (*a > *b) ? true : (*a < *b) ? false : (a > b);
    });
    int n = 0;
    if (min_distance >= 1) {
        const int cell = cv_round(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<std::vector<std::pair<float, float>>> grid((size_t)gw * gh);
        const double md2 = min_distance * min_distance;
        for (int s = 0; s < n_seed; s++) {
            int sx = cv_roundf(seeds[2 * s]), sy = cv_roundf(seeds[2 * s + 1]);
            if (sx < 0 || sy < 0 || sx >= w || sy >= h) continue;
            grid[(size_t)(sy / cell) * gw + sx / cell].push_back({(float)sx, (float)sy});
        }
        for (const float* p : cand) {
            int ofs = (int)(p - eig.data());
            int y = ofs / w, x = ofs - y * w;
            int xc = x / cell, yc = y / cell;
            int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
            int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            bool good = true;
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (auto& m : grid[(size_t)yy * gw + xx]) {
                        float dx = x - m.first, dy = y - m.second;
                        if ((double)(dx * dx + dy * dy) < md2) { good = false; break; }
                    }
            if (good) {
                if (n >= cap) return -1;
                grid[(size_t)yc * gw + xc].push_back({(float)x, (float)y});
                corners_xy[2 * n] = (float)x; corners_xy[2 * n + 1] = (float)y;
                if (scores) scores[n] = *p;
                ++n;
                if (max_corners > 0 && n == max_corners) break;
            }
        }
    } else {
        for (const float* p : cand) {
            int ofs = (int)(p - eig.data());
            int y = ofs / w, x = ofs - y * w;
            if (n >= cap) return -1;
            corners_xy[2 * n] = (float)x; corners_xy[2 * n + 1] = (float)y;
            if (scores) scores[n] = *p;
            ++n;
            if (max_corners > 0 && n == max_corners) break;
        }
    }
    return n;
}
