/* CPU oracle (test infrastructure only): steps behind the tracker -- sub-pixel corner refinement and
 * fundamental-matrix RANSAC.  Restates OpenCV 4.5 imgproc/calib3d semantics for the parameters the
 * reference's configurations name (`subpixel: true`, config/basalt.yaml:102; `F_threshold: 1.0`,
 * config/vins_mono.yaml:101).  The library is absent here and the reference holds no golden vectors for
 * these steps: parity unpinned, this file is normative for the GPU path. */
#include "ora_common.h"
#include "vfe_oracle.h"
#include <cfloat>
#include <cmath>

using namespace ora;

/* The weights cv::cornerSubPix gives the (2*win+1)^2 window: exp(-(x/win)^2) * exp(-(y/win)^2), formed in float;
 * an optional dead square of half-size zero_zone in the middle (zero_zone < 0: none). */
extern "C" int ora_subpix_mask(int win, int zero_zone, float* mask) {
    if (win < 1) return -1;
    const int n = 2 * win + 1;
    for (int i = 0; i < n; i++) {
        const float y = (float)(i - win) / (float)win;
        const float vy = std::exp(-y * y);
        for (int j = 0; j < n; j++) {
            const float x = (float)(j - win) / (float)win;
            mask[i * n + j] = (float)(vy * std::exp(-x * x));
        }
    }
    if (zero_zone >= 0 && zero_zone * 2 + 1 < n)
        for (int i = win - zero_zone; i <= win + zero_zone; i++)
            for (int j = win - zero_zone; j <= win + zero_zone; j++) mask[i * n + j] = 0.f;
    return 0;
}

/* cv::cornerSubPix(img, pts, Size(win, win), Size(zero_zone, zero_zone), {COUNT+EPS, max_iter, eps}).
 * Per iteration: a (2*win+3)^2 float patch around the current estimate by bilinear interpolation (indices
 * clamped to the image), central differences, the weighted normal equations accumulated in double in
 * row-major order, one Newton step.  Stops after max_iter steps, when the squared step is <= eps^2, when the
 * system is singular or the estimate leaves the image; a result further than `win` from the start in x or y is
 * discarded (the start is kept). */
extern "C" int ora_corner_subpix(const uint8_t* img, int w, int h, int step, float* pts, int n, int win, int zero_zone,
                                 int max_iter, double eps) {
    if (win < 1 || w < 1 || h < 1) return -1;
    const int wn = 2 * win + 1, pw = wn + 2;
    std::vector<float> mask((size_t)wn * wn), patch((size_t)pw * pw);
    ora_subpix_mask(win, zero_zone, mask.data());
    max_iter = std::min(std::max(max_iter, 1), 100);
    eps = std::max(eps, 0.0);
    eps *= eps;
    for (int p = 0; p < n; p++) {
        const float tx = pts[2 * p], ty = pts[2 * p + 1];
        float cx = tx, cy = ty;
        int iter = 0;
        double err = 0;
        do {
            /* patch centred on (cx, cy) */
            const float ox = cx - (float)(pw - 1) * 0.5f, oy = cy - (float)(pw - 1) * 0.5f;
            const int ix = cv_floorf(ox), iy = cv_floorf(oy);
            const float fa = ox - (float)ix, fb = oy - (float)iy;
            const float a11 = (1.f - fa) * (1.f - fb), a12 = fa * (1.f - fb), a21 = (1.f - fa) * fb, a22 = fa * fb;
            for (int i = 0; i < pw; i++) {
                const int y0 = std::min(std::max(iy + i, 0), h - 1), y1 = std::min(std::max(iy + i + 1, 0), h - 1);
                for (int j = 0; j < pw; j++) {
                    const int x0 = std::min(std::max(ix + j, 0), w - 1), x1 = std::min(std::max(ix + j + 1, 0), w - 1);
                    patch[(size_t)i * pw + j] = (float)img[(size_t)y0 * step + x0] * a11 + (float)img[(size_t)y0 * step + x1] * a12 +
                                                (float)img[(size_t)y1 * step + x0] * a21 + (float)img[(size_t)y1 * step + x1] * a22;
                }
            }
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            for (int i = 0; i < wn; i++) {
                const double py = i - win;
                for (int j = 0; j < wn; j++) {
                    const float* s = &patch[(size_t)(i + 1) * pw + (j + 1)];
                    const double m = mask[i * wn + j];
                    const double tgx = s[1] - s[-1];
                    const double tgy = s[pw] - s[-pw];
                    const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    const double px = j - win;
                    a += gxx; b += gxy; c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            const double det = a * c - b * b;
            if (std::fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            const double scale = 1.0 / det;
            const float nx = (float)((double)cx + c * scale * bb1 - b * scale * bb2);
            const float ny = (float)((double)cy - b * scale * bb1 + a * scale * bb2);
            err = (double)(nx - cx) * (double)(nx - cx) + (double)(ny - cy) * (double)(ny - cy);
            cx = nx; cy = ny;
            if (cx < 0 || cx >= (float)w || cy < 0 || cy >= (float)h) break;
        } while (++iter < max_iter && err > eps);
        if (std::fabs(cx - tx) > (float)win || std::fabs(cy - ty) > (float)win) { cx = tx; cy = ty; }
        pts[2 * p] = cx; pts[2 * p + 1] = cy;
    }
    return 0;
}
