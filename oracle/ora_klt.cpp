/* CPU oracle (test infrastructure only): pyramidal Lucas-Kanade.
 * Restates cv::calcOpticalFlowPyrLK (OpenCV 4.5 LKTrackerInvoker, flags 0 or
 * OPTFLOW_USE_INITIAL_FLOW) -- the tracker the reference's configuration selects
 * (config/openvins.yaml:26,39 use_klt/num_klt_levels; config/basalt.yaml:93,96
 * pyramid_levels/patch_size; config/vins_mono.yaml:91-104) but never calls.
 *
 * Accumulators: OpenCV carries the Gram matrix and mismatch sums either in float (generic
 * x86 build) or in int64 (its NEON build).  The int64 variant is restated here because its
 * sums are exact, hence independent of summation order, which is what lets a parallel
 * reduction reproduce them bit for bit; the float variant differs only in the last bits
 * of those sums (well inside the 1e-3 px tolerance the front-end is held to). */
#include "ora_common.h"
#include "vfe_oracle.h"
#include <cfloat>

using namespace ora;

namespace {

struct Level { std::vector<uint8_t> buf; int w, h; };

/* Scharr derivative of the (reflect-101 extended) image at an in-image pixel. */
inline void scharr(const Img& I, int x, int y, int& ix, int& iy) {
    int y0 = reflect101(y - 1, I.h), y2 = reflect101(y + 1, I.h);
    int x0 = reflect101(x - 1, I.w), x2 = reflect101(x + 1, I.w);
    int a00 = I.at(y0, x0), a01 = I.at(y0, x), a02 = I.at(y0, x2);
    int a10 = I.at(y, x0), a12 = I.at(y, x2);
    int a20 = I.at(y2, x0), a21 = I.at(y2, x), a22 = I.at(y2, x2);
    ix = 3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20);
    iy = 3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02);
}
/* derivative image is zero-padded outside the image; the image itself is reflect-101 padded */
inline void deriv_at(const Img& I, int x, int y, int& ix, int& iy) {
    if (x < 0 || y < 0 || x >= I.w || y >= I.h) { ix = iy = 0; return; }
    scharr(I, x, y, ix, iy);
}
inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

}  // namespace

extern "C" int ora_klt_track(const uint8_t* prev, const uint8_t* next, int w, int h, int step,
                             const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
                             int win, int max_level, int max_iter, double eps, double min_eig_thr,
                             int use_initial_flow) {
    max_level = ora_lk_max_level(w, h, win, max_level);
    std::vector<Level> P(max_level + 1), N(max_level + 1);
    auto build = [&](std::vector<Level>& L, const uint8_t* img) {
        L[0].w = w; L[0].h = h; L[0].buf.resize((size_t)w * h);
        for (int y = 0; y < h; y++) std::memcpy(&L[0].buf[(size_t)y * w], img + (size_t)y * step, w);
        for (int l = 1; l <= max_level; l++) {
            L[l].w = (L[l - 1].w + 1) / 2; L[l].h = (L[l - 1].h + 1) / 2;
            L[l].buf.resize((size_t)L[l].w * L[l].h);
            ora_pyr_down(L[l - 1].buf.data(), L[l - 1].w, L[l - 1].h, L[l - 1].w, L[l].buf.data(), L[l].w);
        }
    };
    build(P, prev); build(N, next);

    const double epsilon = eps * eps;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (win - 1) * 0.5f;
    std::vector<short> Iw((size_t)win * win), dIw((size_t)win * win * 2);

    for (int i = 0; i < n; i++) { status[i] = 1; if (err) err[i] = 0; }

    for (int level = max_level; level >= 0; level--) {
        Img I{P[level].buf.data(), P[level].w, P[level].h, P[level].w};
        Img J{N[level].buf.data(), N[level].w, N[level].h, N[level].w};
        for (int pt = 0; pt < n; pt++) {
            float px = prev_pts[2 * pt] * (float)(1. / (1 << level));
            float py = prev_pts[2 * pt + 1] * (float)(1. / (1 << level));
            float nx, ny;
            if (level == max_level) {
                if (use_initial_flow) {
                    nx = next_pts[2 * pt] * (float)(1. / (1 << level));
                    ny = next_pts[2 * pt + 1] * (float)(1. / (1 << level));
                } else { nx = px; ny = py; }
            } else { nx = next_pts[2 * pt] * 2.f; ny = next_pts[2 * pt + 1] * 2.f; }
            next_pts[2 * pt] = nx; next_pts[2 * pt + 1] = ny;

            px -= half; py -= half;
            int ipx = cv_floorf(px), ipy = cv_floorf(py);
            if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
                if (level == 0) { status[pt] = 0; if (err) err[pt] = 0; }
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = cv_roundf((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_roundf(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_roundf((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    int X = ipx + x, Y = ipy + y;
                    int s00 = I.at_r(Y, X), s01 = I.at_r(Y, X + 1), s10 = I.at_r(Y + 1, X), s11 = I.at_r(Y + 1, X + 1);
                    int ival = descale(s00 * iw00 + s01 * iw01 + s10 * iw10 + s11 * iw11, W_BITS - 5);
                    int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
                    deriv_at(I, X, Y, dx00, dy00); deriv_at(I, X + 1, Y, dx01, dy01);
                    deriv_at(I, X, Y + 1, dx10, dy10); deriv_at(I, X + 1, Y + 1, dx11, dy11);
                    int ixval = descale(dx00 * iw00 + dx01 * iw01 + dx10 * iw10 + dx11 * iw11, W_BITS);
                    int iyval = descale(dy00 * iw00 + dy01 * iw01 + dy10 * iw10 + dy11 * iw11, W_BITS);
                    Iw[(size_t)y * win + x] = (short)ival;
                    dIw[((size_t)y * win + x) * 2] = (short)ixval;
                    dIw[((size_t)y * win + x) * 2 + 1] = (short)iyval;
                    iA11 += (int64_t)ixval * ixval; iA12 += (int64_t)ixval * iyval; iA22 += (int64_t)iyval * iyval;
                }
            float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
            if ((double)minEig < min_eig_thr || D < FLT_EPSILON) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0, pdy = 0;
            for (int j = 0; j < max_iter; j++) {
                int inx = cv_floorf(nx), iny = cv_floorf(ny);
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nx - inx; b = ny - iny;
                iw00 = cv_roundf((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_roundf(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_roundf((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t ib1 = 0, ib2 = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        int X = inx + x, Y = iny + y;
                        int diff = descale(J.at_r(Y, X) * iw00 + J.at_r(Y, X + 1) * iw01 +
                                           J.at_r(Y + 1, X) * iw10 + J.at_r(Y + 1, X + 1) * iw11, W_BITS - 5) -
                                   Iw[(size_t)y * win + x];
                        ib1 += (int64_t)diff * dIw[((size_t)y * win + x) * 2];
                        ib2 += (int64_t)diff * dIw[((size_t)y * win + x) * 2 + 1];
                    }
                float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
                float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
                nx += dx; ny += dy;
                next_pts[2 * pt] = nx + half; next_pts[2 * pt + 1] = ny + half;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && std::fabs(dx + pdx) < 0.01 && std::fabs(dy + pdy) < 0.01) {
                    next_pts[2 * pt] -= dx * 0.5f; next_pts[2 * pt + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[pt] && err && level == 0) {
                float fx = next_pts[2 * pt] - half, fy = next_pts[2 * pt + 1] - half;
                int inx = cv_floorf(fx), iny = cv_floorf(fy);
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { status[pt] = 0; continue; }
                float aa = fx - inx, bb = fy - iny;
                iw00 = cv_roundf((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = cv_roundf(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = cv_roundf((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t esum = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        int X = inx + x, Y = iny + y;
                        int diff = descale(J.at_r(Y, X) * iw00 + J.at_r(Y, X + 1) * iw01 +
                                           J.at_r(Y + 1, X) * iw10 + J.at_r(Y + 1, X + 1) * iw11, W_BITS - 5) -
                                   Iw[(size_t)y * win + x];
                        esum += std::abs(diff);
                    }
                err[pt] = (float)esum * 1.f / (float)(32 * win * win);
            }
        }
    }
    return 0;
}

/* cv::undistortPoints for the radtan model (config/vins_mono.yaml:32-35 camera.k1..p2), no R / P:
 * x0 = (u - cx)/fx; five iterations of x = (x0 - delta(x, y)) * icdist(x, y) in double. */
extern "C" void ora_undistort_points(const float* pts, float* out, long long n, const double* cam4, const double* d5) {
    const double fx = cam4[0], fy = cam4[1], cx = cam4[2], cy = cam4[3];
    const double k1 = d5[0], k2 = d5[1], p1 = d5[2], p2 = d5[3], k3 = d5[4];
    for (long long i = 0; i < n; i++) {
        const double x0 = ((double)pts[2 * i] - cx) / fx, y0 = ((double)pts[2 * i + 1] - cy) / fy;
        double x = x0, y = y0;
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
            const double dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        out[2 * i] = (float)x; out[2 * i + 1] = (float)y;
    }
}
