/* CPU oracle (test infrastructure only): the same results as ora_klt.cpp / ora_gftt.cpp, organised the way an optimised
 * CPU library organises the work, for the benchmark's CPU baseline.
 *
 * cv::calcOpticalFlowPyrLK builds ONE Scharr derivative image per pyramid level and reads windows from border-extended
 * images (OpenCV 4.5 lkpyramid.cpp: calcSharrDeriv + buildOpticalFlowPyramid with a winSize border); ora_klt.cpp instead
 * evaluates four 3x3 stencils per window pixel per point and reflects every coordinate, which is simple to check and
 * several times slower.  cv::cornerMinEigenVal streams rows; ora_gftt.cpp materialises six full-size float planes.
 * The functions here do the former with the arithmetic of the latter, operation for operation, so their outputs are
 * bit-identical to the plain oracle's (tests/test_oracle_opt.py), and they are what `cpu_baseline.port_fast` times. */
#include "ora_common.h"
#include "vfe_oracle.h"
#include <cfloat>

using namespace ora;

namespace {

inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

/* pyrDown with a ring of five horizontally filtered rows: every source row is filtered once */
void pyr_down_ring(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    std::vector<int> ring((size_t)5 * dw);
    std::vector<int> have(5, -(1 << 30)); /* source row held by each ring slot */
    auto filter_row = [&](int sy) -> const int* {
        const int slot = ((sy % 5) + 5) % 5;
        int* r = &ring[(size_t)slot * dw];
        if (have[slot] == sy) return r;
        have[slot] = sy;
        const uint8_t* s = src + (size_t)sy * step;
        /* interior: no reflection needed for 2x-2 >= 0 and 2x+2 < w */
        const int x_lo = 1, x_hi = (w - 3) / 2; /* inclusive range of x with all five taps inside */
        for (int x = 0; x < dw; x++) {
            if (x >= x_lo && x <= x_hi) {
                const uint8_t* p = s + 2 * x;
                r[x] = p[-2] + 4 * p[-1] + 6 * p[0] + 4 * p[1] + p[2];
            } else {
                const int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = reflect101(2 * x, w),
                          x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                r[x] = s[x0] + 4 * s[x1] + 6 * s[x2] + 4 * s[x3] + s[x4];
            }
        }
        return r;
    };
    for (int y = 0; y < dh; y++) {
        const int* r[5];
        /* distinct reflected rows may share a ring slot only when the image is tiny; filter in order of use and copy */
        int rows[5];
        for (int k = 0; k < 5; k++) rows[k] = reflect101(2 * y + k - 2, h);
        bool clash = false;
        for (int a = 0; a < 5 && !clash; a++)
            for (int b = a + 1; b < 5; b++)
                if (rows[a] != rows[b] && ((rows[a] % 5) == (rows[b] % 5))) { clash = true; break; }
        if (clash) { /* fall back to filtering the five rows into private storage */
            std::vector<int> tmp((size_t)5 * dw);
            for (int k = 0; k < 5; k++) {
                std::fill(have.begin(), have.end(), -(1 << 30));
                const int* f = filter_row(rows[k]);
                std::copy(f, f + dw, &tmp[(size_t)k * dw]);
            }
            std::fill(have.begin(), have.end(), -(1 << 30));
            for (int x = 0; x < dw; x++) {
                const int v = tmp[x] + 4 * tmp[dw + x] + 6 * tmp[2 * dw + x] + 4 * tmp[3 * dw + x] + tmp[4 * dw + x];
                dst[(size_t)y * dstep + x] = (uint8_t)((v + 128) >> 8);
            }
            continue;
        }
        for (int k = 0; k < 5; k++) r[k] = filter_row(rows[k]);
        uint8_t* d = dst + (size_t)y * dstep;
        for (int x = 0; x < dw; x++) d[x] = (uint8_t)((r[0][x] + 4 * r[1][x] + 6 * r[2][x] + 4 * r[3][x] + r[4][x] + 128) >> 8);
    }
}

/* one pyramid level: the image with a reflect-101 border of `pad` pixels and (for the previous frame) its Scharr derivative
 * image, interleaved (Ix, Iy) shorts, zero outside the image, with the same border */
struct LevelOpt {
    int w = 0, h = 0, pad = 0, pitch = 0;
    std::vector<uint8_t> raw;   /* w x h, no border: source of the next level */
    std::vector<uint8_t> img;   /* (w + 2 pad) x (h + 2 pad) */
    std::vector<short> der;     /* 2 shorts per pixel, same geometry; empty for the next frame */
    const uint8_t* px(int y, int x) const { return &img[(size_t)(y + pad) * pitch + (x + pad)]; }
    const short* dx(int y, int x) const { return &der[((size_t)(y + pad) * pitch + (x + pad)) * 2]; }
};

void make_border(LevelOpt& L) {
    L.pitch = L.w + 2 * L.pad;
    L.img.resize((size_t)L.pitch * (L.h + 2 * L.pad));
    std::vector<int> mapx(L.pitch);
    for (int x = 0; x < L.pitch; x++) mapx[x] = reflect101(x - L.pad, L.w);
    for (int y = -L.pad; y < L.h + L.pad; y++) {
        const uint8_t* s = &L.raw[(size_t)reflect101(y, L.h) * L.w];
        uint8_t* d = &L.img[(size_t)(y + L.pad) * L.pitch];
        for (int x = 0; x < L.pad; x++) d[x] = s[mapx[x]];
        std::memcpy(d + L.pad, s, L.w);
        for (int x = L.pad + L.w; x < L.pitch; x++) d[x] = s[mapx[x]];
    }
}

void make_deriv(LevelOpt& L) {
    L.der.assign((size_t)L.pitch * (L.h + 2 * L.pad) * 2, 0);
    for (int y = 0; y < L.h; y++) {
        const uint8_t *r0 = L.px(y - 1, 0), *r1 = L.px(y, 0), *r2 = L.px(y + 1, 0); /* the border is the reflection */
        short* d = &L.der[((size_t)(y + L.pad) * L.pitch + L.pad) * 2];
        for (int x = 0; x < L.w; x++) {
            const int a00 = r0[x - 1], a01 = r0[x], a02 = r0[x + 1], a10 = r1[x - 1], a12 = r1[x + 1];
            const int a20 = r2[x - 1], a21 = r2[x], a22 = r2[x + 1];
            d[2 * x] = (short)(3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20));
            d[2 * x + 1] = (short)(3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02));
        }
    }
}

void build_levels(std::vector<LevelOpt>& L, const uint8_t* img, int w, int h, int step, int max_level, int pad, bool deriv) {
    L.resize(max_level + 1);
    L[0].w = w; L[0].h = h; L[0].raw.resize((size_t)w * h);
    for (int y = 0; y < h; y++) std::memcpy(&L[0].raw[(size_t)y * w], img + (size_t)y * step, w);
    for (int l = 1; l <= max_level; l++) {
        L[l].w = (L[l - 1].w + 1) / 2; L[l].h = (L[l - 1].h + 1) / 2;
        L[l].raw.resize((size_t)L[l].w * L[l].h);
        pyr_down_ring(L[l - 1].raw.data(), L[l - 1].w, L[l - 1].h, L[l - 1].w, L[l].raw.data(), L[l].w);
    }
    for (int l = 0; l <= max_level; l++) {
        L[l].pad = pad;
        make_border(L[l]);
        if (deriv) make_deriv(L[l]);
    }
}

}  // namespace

extern "C" int ora_klt_track_opt(const uint8_t* prev, const uint8_t* next, int w, int h, int step,
                                 const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
                                 int win, int max_level, int max_iter, double eps, double min_eig_thr,
                                 int use_initial_flow) {
    max_level = ora_lk_max_level(w, h, win, max_level);
    /* windows start at -win .. size-1 and reach win + 1 further; reflections of reflections (levels smaller than the
     * border) are handled by reflect101 when the border is filled */
    const int pad = win + 2;
    std::vector<LevelOpt> P, N;
    build_levels(P, prev, w, h, step, max_level, pad, true);
    build_levels(N, next, w, h, step, max_level, pad, false);

    const double epsilon = eps * eps;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (win - 1) * 0.5f;
    std::vector<short> Iw((size_t)win * win), dIw((size_t)win * win * 2);

    for (int i = 0; i < n; i++) { status[i] = 1; if (err) err[i] = 0; }

    for (int level = max_level; level >= 0; level--) {
        const LevelOpt& I = P[level];
        const LevelOpt& J = N[level];
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
            const int ipx = cv_floorf(px), ipy = cv_floorf(py);
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
            for (int y = 0; y < win; y++) {
                const uint8_t *s0 = I.px(ipy + y, ipx), *s1 = I.px(ipy + y + 1, ipx);
                const short *d0 = I.dx(ipy + y, ipx), *d1 = I.dx(ipy + y + 1, ipx);
                short* Ip = &Iw[(size_t)y * win];
                short* dp = &dIw[(size_t)y * win * 2];
                int64_t r11 = 0, r12 = 0, r22 = 0;
                for (int x = 0; x < win; x++) {
                    const int ival = descale(s0[x] * iw00 + s0[x + 1] * iw01 + s1[x] * iw10 + s1[x + 1] * iw11, W_BITS - 5);
                    const int ixval = descale(d0[2 * x] * iw00 + d0[2 * x + 2] * iw01 + d1[2 * x] * iw10 + d1[2 * x + 2] * iw11, W_BITS);
                    const int iyval = descale(d0[2 * x + 1] * iw00 + d0[2 * x + 3] * iw01 + d1[2 * x + 1] * iw10 + d1[2 * x + 3] * iw11, W_BITS);
                    Ip[x] = (short)ival; dp[2 * x] = (short)ixval; dp[2 * x + 1] = (short)iyval;
                    r11 += ixval * ixval; r12 += ixval * iyval; r22 += iyval * iyval; /* |value| < 2^15: the products fit int */
                }
                iA11 += r11; iA12 += r12; iA22 += r22;
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
                const int inx = cv_floorf(nx), iny = cv_floorf(ny);
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
                for (int y = 0; y < win; y++) {
                    const uint8_t *s0 = J.px(iny + y, inx), *s1 = J.px(iny + y + 1, inx);
                    const short* Ip = &Iw[(size_t)y * win];
                    const short* dp = &dIw[(size_t)y * win * 2];
                    int64_t r1 = 0, r2 = 0;
                    for (int x = 0; x < win; x++) {
                        const int diff = descale(s0[x] * iw00 + s0[x + 1] * iw01 + s1[x] * iw10 + s1[x + 1] * iw11, W_BITS - 5) - Ip[x];
                        r1 += diff * dp[2 * x]; r2 += diff * dp[2 * x + 1];
                    }
                    ib1 += r1; ib2 += r2;
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
                const int inx = cv_floorf(fx), iny = cv_floorf(fy);
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { status[pt] = 0; continue; }
                float aa = fx - inx, bb = fy - iny;
                iw00 = cv_roundf((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = cv_roundf(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = cv_roundf((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t esum = 0;
                for (int y = 0; y < win; y++) {
                    const uint8_t *s0 = J.px(iny + y, inx), *s1 = J.px(iny + y + 1, inx);
                    const short* Ip = &Iw[(size_t)y * win];
                    for (int x = 0; x < win; x++)
                        esum += std::abs(descale(s0[x] * iw00 + s0[x + 1] * iw01 + s1[x] * iw10 + s1[x + 1] * iw11, W_BITS - 5) - Ip[x]);
                }
                err[pt] = (float)esum * 1.f / (float)(32 * win * win);
            }
        }
    }
    return 0;
}

/* cornerMinEigenVal, row-streaming: three-row rings of the Sobel row filters, of the product rows' horizontal 3-sums,
 * nothing of full frame size but the output.  Same operations in the same order as ora_min_eigen_map. */
extern "C" void ora_min_eigen_map_opt(const uint8_t* img, int w, int h, int step, float* eig) {
    const double scale = 1.0 / ((1 << 2) * 3 * 255.0);
    const float k_o = (float)scale, k_c = (float)(2.0 * scale);
    /* rx / ry of source row r live in slot r mod 3 (rows are visited in order, reflected rows recomputed on demand) */
    std::vector<float> rx((size_t)3 * w), ry((size_t)3 * w);
    std::vector<float> hx((size_t)3 * w), hxy((size_t)3 * w), hy((size_t)3 * w);
    std::vector<float> pxx(w), pxy(w), pyy(w);
    std::vector<float> trx(w), try_(w);
    auto row_filters = [&](int yy, float* ox, float* oy) {
        const uint8_t* s = img + (size_t)yy * step;
        for (int x = 0; x < w; x++) {
            const int a = s[reflect101(x - 1, w)], b = s[x], c = s[reflect101(x + 1, w)];
            ox[x] = (float)(c - a);
            oy[x] = k_c * (float)b + k_o * (float)(a + c);
        }
    };
    /* product row y -> horizontal 3-sums into slot y mod 3 of hx / hxy / hy */
    auto product_row = [&](int y) {
        const float *x0, *x1, *x2, *y0, *y2;
        const int ya = reflect101(y - 1, h), yb = reflect101(y + 1, h);
        /* the ring holds rows y-1, y, y+1 when they are distinct in-image rows; reflected ones equal y+1 / y-1 */
        x0 = &rx[(size_t)(ya % 3) * w]; x1 = &rx[(size_t)(y % 3) * w]; x2 = &rx[(size_t)(yb % 3) * w];
        y0 = &ry[(size_t)(ya % 3) * w]; y2 = &ry[(size_t)(yb % 3) * w];
        for (int x = 0; x < w; x++) {
            const float dx = k_c * x1[x] + k_o * (x0[x] + x2[x]);
            const float dy = y2[x] - y0[x];
            pxx[x] = dx * dx; pxy[x] = dx * dy; pyy[x] = dy * dy;
        }
        float* ohx = &hx[(size_t)(y % 3) * w]; float* ohxy = &hxy[(size_t)(y % 3) * w]; float* ohy = &hy[(size_t)(y % 3) * w];
        for (int x = 0; x < w; x++) {
            const int l = reflect101(x - 1, w), r = reflect101(x + 1, w);
            ohx[x] = (pxx[l] + pxx[x]) + pxx[r];
            ohxy[x] = (pxy[l] + pxy[x]) + pxy[r];
            ohy[x] = (pyy[l] + pyy[x]) + pyy[r];
        }
    };
    auto eig_row = [&](int y) {
        const int ya = reflect101(y - 1, h), yb = reflect101(y + 1, h);
        const float *tx = &hx[(size_t)(ya % 3) * w], *cx = &hx[(size_t)(y % 3) * w], *bx = &hx[(size_t)(yb % 3) * w];
        const float *txy = &hxy[(size_t)(ya % 3) * w], *cxy = &hxy[(size_t)(y % 3) * w], *bxy = &hxy[(size_t)(yb % 3) * w];
        const float *ty = &hy[(size_t)(ya % 3) * w], *cy = &hy[(size_t)(y % 3) * w], *by = &hy[(size_t)(yb % 3) * w];
        float* e = eig + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            const float sxx = (tx[x] + cx[x]) + bx[x], sxy = (txy[x] + cxy[x]) + bxy[x], syy = (ty[x] + cy[x]) + by[x];
            const float a = sxx * 0.5f, bb = sxy, cc = syy * 0.5f, d = a - cc;
            e[x] = (a + cc) - std::sqrt(d * d + bb * bb);
        }
    };
    if (h < 3) { ora_min_eigen_map(img, w, h, step, eig); return; } /* rings of three distinct rows need three rows */
    /* the vertical filters at row 0 use row 1 twice (reflection): rows -1 and +1 share slot 1 */
    row_filters(0, &rx[0], &ry[0]);
    row_filters(1, &rx[(size_t)w], &ry[(size_t)w]);
    product_row(0);
    for (int y = 1; y < h; y++) {
        if (y + 1 < h) row_filters(y + 1, &rx[(size_t)((y + 1) % 3) * w], &ry[(size_t)((y + 1) % 3) * w]);
        /* at y = h-1 row y+1 reflects to y-1, still in its slot */
        product_row(y);
        if (y >= 2) eig_row(y - 1);
        else { /* y == 1: row 0's vertical sum uses product rows 1, 0, 1 */ eig_row(0); }
    }
    eig_row(h - 1); /* product rows h-2, h-1, h-2 */
}
