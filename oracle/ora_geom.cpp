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

/* ---- fundamental matrix by RANSAC -------------------------------------------------------------
 * cv::findFundamentalMat(p1, p2, FM_RANSAC, threshold, confidence, mask) as VINS-style front-ends call it to
 * reject outlier tracks (`F_threshold`, config/vins_mono.yaml:101).  Same structure as OpenCV's
 * RANSAC registrator: 7-point minimal solver (up to three models per sample), error = larger of the two squared
 * point-to-epipolar-line distances, inlier iff (float)error <= threshold^2, a model replaces the best one when it
 * has more inliers than it and more than 6, and the iteration budget shrinks with the best inlier ratio.
 * Where OpenCV's result depends on its own generator and on LAPACK-style SVD, this restatement fixes:
 *   - sample k = seven distinct indices drawn from mix32(seed, 64 k + attempt) mod n;
 *   - the null space of the 7x9 system by Gauss-Jordan elimination with row pivoting (free columns give the two
 *     basis vectors, each scaled to unit largest entry);
 *   - the cubic's real roots by bracketing between the critical points and 100 bisection steps;
 *   - log() of the budget formula by an atanh series;
 * so that every step uses only +, -, *, / and sqrt in double and a GPU implementation can reproduce it bit for
 * bit. */
namespace {
inline uint32_t mix32(uint32_t a, uint32_t b) {
    uint32_t h = a ^ (b * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

void ransac_sample(uint32_t seed, int k, int n, int* idx) {
    uint32_t attempt = 0;
    for (int j = 0; j < 7; j++) {
        for (;;) {
            const int c = (int)(mix32(seed, (uint32_t)k * 64u + attempt) % (uint32_t)n);
            attempt++;
            bool dup = false;
            for (int i = 0; i < j; i++) dup = dup || idx[i] == c;
            if (!dup || attempt >= 64u) { idx[j] = c; break; }
        }
    }
}

inline double det3(const double* u, const double* v, const double* w) {
    return u[0] * (v[1] * w[2] - v[2] * w[1]) - u[1] * (v[0] * w[2] - v[2] * w[0]) + u[2] * (v[0] * w[1] - v[1] * w[0]);
}

inline double cubic_at(const double* c, double x) { return ((c[3] * x + c[2]) * x + c[1]) * x + c[0]; }

/* root of the cubic in [lo, hi] given a sign change; 100 halvings */
double bisect(const double* c, double lo, double hi) {
    double flo = cubic_at(c, lo);
    for (int i = 0; i < 100; i++) {
        const double mid = 0.5 * (lo + hi);
        const double fm = cubic_at(c, mid);
        if ((flo <= 0 && fm <= 0) || (flo >= 0 && fm >= 0)) { lo = mid; flo = fm; } else hi = mid;
    }
    return 0.5 * (lo + hi);
}

/* real roots of c[3] x^3 + c[2] x^2 + c[1] x + c[0]; returns their number (0..3), ascending */
int cubic_roots(const double* cin, double* roots) {
    double c[4] = {cin[0], cin[1], cin[2], cin[3]};
    double big = 0;
    for (int i = 0; i < 4; i++) big = std::max(big, std::fabs(c[i]));
    if (!(big > 0)) return 0;
    for (int i = 0; i < 4; i++) c[i] = c[i] / big;
    if (std::fabs(c[3]) < 1e-14) {
        if (std::fabs(c[2]) < 1e-14) {
            if (std::fabs(c[1]) < 1e-14) return 0;
            roots[0] = -c[0] / c[1];
            return 1;
        }
        const double disc = c[1] * c[1] - 4.0 * c[2] * c[0];
        if (disc < 0) return 0;
        const double s = std::sqrt(disc);
        const double r0 = (-c[1] - s) / (2.0 * c[2]), r1 = (-c[1] + s) / (2.0 * c[2]);
        roots[0] = std::min(r0, r1); roots[1] = std::max(r0, r1);
        return 2;
    }
    /* all roots lie within the Cauchy bound */
    const double B = 1.0 + std::max(std::fabs(c[0] / c[3]), std::max(std::fabs(c[1] / c[3]), std::fabs(c[2] / c[3])));
    const double disc = c[2] * c[2] - 3.0 * c[3] * c[1]; /* of the derivative 3 c3 x^2 + 2 c2 x + c1, over 4 */
    double edge[4];
    int ne = 0;
    edge[ne++] = -B;
    if (disc > 0) {
        const double s = std::sqrt(disc);
        const double x0 = (-c[2] - s) / (3.0 * c[3]), x1 = (-c[2] + s) / (3.0 * c[3]);
        edge[ne++] = std::min(x0, x1); edge[ne++] = std::max(x0, x1);
    }
    edge[ne++] = B;
    int nr = 0;
    for (int i = 0; i + 1 < ne; i++) {
        const double lo = edge[i], hi = edge[i + 1];
        const double flo = cubic_at(c, lo), fhi = cubic_at(c, hi);
        if ((flo < 0 && fhi > 0) || (flo > 0 && fhi < 0) || (flo == 0 && i == 0)) roots[nr++] = bisect(c, lo, hi);
        else if (fhi == 0) roots[nr++] = hi;
    }
    return nr;
}

/* m1, m2: seven correspondences (x, y); F: up to three row-major 3x3 models */
int seven_point(const double (*m1)[2], const double (*m2)[2], double (*F)[9]) {
    double A[7][9];
    double big = 0;
    for (int i = 0; i < 7; i++) {
        const double x1 = m1[i][0], y1 = m1[i][1], x2 = m2[i][0], y2 = m2[i][1];
        A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2; A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2;
        A[i][6] = x1; A[i][7] = y1; A[i][8] = 1.0;
        for (int j = 0; j < 9; j++) big = std::max(big, std::fabs(A[i][j]));
    }
    const double tiny = big * 1e-10;
    int piv[7], freec[2], nfree = 0, r = 0;
    for (int col = 0; col < 9; col++) {
        if (r == 7) { if (nfree == 2) return 0; freec[nfree++] = col; continue; }
        int p = r;
        double best = std::fabs(A[r][col]);
        for (int i = r + 1; i < 7; i++)
            if (std::fabs(A[i][col]) > best) { best = std::fabs(A[i][col]); p = i; }
        if (!(best > tiny)) { if (nfree == 2) return 0; freec[nfree++] = col; continue; }
        if (p != r) for (int j = 0; j < 9; j++) std::swap(A[p][j], A[r][j]);
        const double inv = 1.0 / A[r][col];
        for (int j = 0; j < 9; j++) A[r][j] = A[r][j] * inv;
        for (int i = 0; i < 7; i++) {
            if (i == r) continue;
            const double f = A[i][col];
            for (int j = 0; j < 9; j++) A[i][j] = A[i][j] - f * A[r][j];
        }
        piv[r++] = col;
    }
    if (r != 7 || nfree != 2) return 0;
    double f1[9], f2[9];
    for (int v = 0; v < 2; v++) {
        double* f = v == 0 ? f1 : f2;
        for (int j = 0; j < 9; j++) f[j] = 0;
        f[freec[v]] = 1.0;
        for (int i = 0; i < 7; i++) f[piv[i]] = -A[i][freec[v]];
        double m = 0;
        for (int j = 0; j < 9; j++) m = std::max(m, std::fabs(f[j]));
        for (int j = 0; j < 9; j++) f[j] = f[j] / m;
    }
    /* det(f2 + x (f1 - f2)) = 0 */
    double d[9];
    for (int j = 0; j < 9; j++) d[j] = f1[j] - f2[j];
    double c[4];
    c[0] = det3(f2, f2 + 3, f2 + 6);
    c[1] = det3(d, f2 + 3, f2 + 6) + det3(f2, d + 3, f2 + 6) + det3(f2, f2 + 3, d + 6);
    c[2] = det3(d, d + 3, f2 + 6) + det3(d, f2 + 3, d + 6) + det3(f2, d + 3, d + 6);
    c[3] = det3(d, d + 3, d + 6);
    double roots[3];
    const int nr = cubic_roots(c, roots);
    for (int k = 0; k < nr; k++) {
        double* Fk = F[k];
        for (int j = 0; j < 9; j++) Fk[j] = f2[j] + roots[k] * d[j];
        if (std::fabs(Fk[8]) > DBL_EPSILON) {
            const double mu = 1.0 / Fk[8];
            for (int j = 0; j < 8; j++) Fk[j] = Fk[j] * mu;
            Fk[8] = 1.0;
        }
    }
    return nr;
}

inline bool f_inlier(const double* F, double x1, double y1, double x2, double y2, float thr2) {
    double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1.0 / (a * a + b * b);
    const double d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1.0 / (a * a + b * b);
    const double d1 = x1 * a + y1 * b + c;
    const float err = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
    return err <= thr2;
}

/* natural logarithm from +, -, *, / only: x = m 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m-1)/(m+1)) */
double det_log(double x) {
    int e = 0;
    while (x >= 1.4142135623730951) { x = x * 0.5; e++; }
    while (x < 0.7071067811865476) { x = x * 2.0; e--; }
    const double s = (x - 1.0) / (x + 1.0), s2 = s * s;
    double term = s, sum = 0;
    for (int k = 0; k < 16; k++) { sum = sum + term / (double)(2 * k + 1); term = term * s2; }
    return 2.0 * sum + (double)e * 0.6931471805599453;
}

/* cv::RANSACUpdateNumIters for 7-point samples; ep = outlier ratio */
int ransac_update_iters(double p, double ep, int max_iters) {
    p = std::min(std::max(p, 0.0), 1.0);
    ep = std::min(std::max(ep, 0.0), 1.0);
    const double num = std::max(1.0 - p, DBL_MIN);
    const double q = 1.0 - ep;
    const double denom = 1.0 - q * q * q * q * q * q * q;
    if (denom < DBL_MIN) return 0;
    const double ln = det_log(num), ld = det_log(denom);
    if (ld >= 0 || -ln >= (double)max_iters * (-ld)) return max_iters;
    return cv_round(ln / ld);
}
}  // namespace

/* p1, p2 [n][2]; status [n] in/out: non-zero entries take part, 1 on return = inlier.  Returns the inlier count;
 * fewer than 8 participants: nothing is rejected, F = 0; no model found: everything is rejected, F = 0. */
extern "C" int ora_fundamental_ransac(const float* p1, const float* p2, uint8_t* status, int n, double threshold,
                                      double confidence, int max_iters, uint32_t seed, double* F_out) {
    std::vector<int> part;
    for (int i = 0; i < n; i++) if (status[i]) part.push_back(i);
    const int np = (int)part.size();
    for (int j = 0; j < 9; j++) F_out[j] = 0;
    if (np < 8) return np;
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    if (threshold <= 0) threshold = 3;
    const float thr2 = (float)(threshold * threshold);
    max_iters = std::min(std::max(max_iters, 1), 1024);
    int niters = max_iters, best_count = 0;
    double best[9] = {0};
    for (int k = 0; k < niters; k++) {
        int idx[7];
        ransac_sample(seed, k, np, idx);
        double m1[7][2], m2[7][2], F[3][9];
        for (int j = 0; j < 7; j++) {
            const int i = part[idx[j]];
            m1[j][0] = p1[2 * i]; m1[j][1] = p1[2 * i + 1]; m2[j][0] = p2[2 * i]; m2[j][1] = p2[2 * i + 1];
        }
        const int nm = seven_point(m1, m2, F);
        for (int m = 0; m < nm; m++) {
            int good = 0;
            for (int q = 0; q < np; q++) {
                const int i = part[q];
                good += f_inlier(F[m], p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], thr2) ? 1 : 0;
            }
            if (good > std::max(best_count, 6)) {
                best_count = good;
                for (int j = 0; j < 9; j++) best[j] = F[m][j];
                niters = ransac_update_iters(confidence, (double)(np - good) / (double)np, niters);
            }
        }
    }
    if (best_count == 0) {
        for (int i = 0; i < n; i++) status[i] = 0;
        return 0;
    }
    for (int j = 0; j < 9; j++) F_out[j] = best[j];
    for (int q = 0; q < np; q++) {
        const int i = part[q];
        status[i] = f_inlier(best, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], thr2) ? 1 : 0;
    }
    return best_count;
}
