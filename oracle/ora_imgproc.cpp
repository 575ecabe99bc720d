/* CPU oracle (test infrastructure only): colour conversion, pyrDown, exact resize, 7x7 Gaussian.
 * Restates OpenCV 4.5 imgproc as used through the reference's call sites
 * (src/slam/adapters/openvins_adapter.cpp:70-75, orbslam3_adapter.cpp:69-79). */
#include "ora_common.h"
#include "vfe_oracle.h"

using namespace ora;

/* cvtColor(COLOR_BGR2GRAY) for 8-bit: 15-bit fixed-point BT.601 weights,
 * gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15. */
extern "C" void ora_bgr2gray(const uint8_t* bgr, int w, int h, int step, uint8_t* gray, int gstep) {
    for (int y = 0; y < h; y++) {
        const uint8_t* s = bgr + (size_t)y * step;
        uint8_t* d = gray + (size_t)y * gstep;
        for (int x = 0; x < w; x++) {
            int b = s[3 * x], g = s[3 * x + 1], r = s[3 * x + 2];
            d[x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
        }
    }
}

/* equalizeHist: lut[v] = saturate(round(cdf'(v) * 255/(total - hist[first]))). */
extern "C" void ora_equalize_hist(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep) {
    int hist[256] = {0};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) hist[src[(size_t)y * step + x]]++;
    int total = w * h;
    int i = 0;
    while (!hist[i]) ++i;
    uint8_t lut[256];
    if (hist[i] == total) {
        for (int k = 0; k < 256; k++) lut[k] = (uint8_t)i;
    } else {
        float scale = (256 - 1.f) / (total - hist[i]);
        int sum = 0;
        for (int k = 0; k <= i; k++) lut[k] = 0;
        for (++i; i < 256; ++i) {
            sum += hist[i];
            int v = cv_roundf(sum * scale);
            lut[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(size_t)y * dstep + x] = lut[src[(size_t)y * step + x]];
}

/* CLAHE (cv::createCLAHE(clipLimit, Size(tilesX, tilesY))->apply) on 8-bit gray, as OpenVINS-style front-ends
 * use it (`histogram_method: "CLAHE"`, docs/reference/15_openvins_smartphone_config.md:111).
 *   - if either dimension is not a multiple of the tile count the image is extended at the right/bottom by
 *     tiles - (size % tiles) pixels in BOTH dimensions (a dimension that divides evenly gets a full `tiles`
 *     extra) with reflect-101, for the histograms only;
 *   - per tile: histogram, clip at max(int(clip * area / 256), 1), spread the excess (equal share to every bin,
 *     the remainder one count to every `256 / remainder`-th bin from bin 0), LUT = saturate(round(cdf * 255 / area));
 *   - every pixel: bilinear blend of the four surrounding tiles' LUT entries in float, tile centres at
 *     (t + 0.5) * tile size, rounded half-to-even. */
extern "C" int ora_clahe(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep, double clip, int tiles_x,
                         int tiles_y) {
    if (tiles_x < 1 || tiles_y < 1 || w < 1 || h < 1) return -1;
    int ew = w, eh = h;
    if (w % tiles_x != 0 || h % tiles_y != 0) { ew = w + tiles_x - (w % tiles_x); eh = h + tiles_y - (h % tiles_y); }
    if (ew - w >= w || eh - h >= h) return -1; /* reflect-101 needs the extension to be smaller than the image */
    const int tw = ew / tiles_x, th = eh / tiles_y, area = tw * th;
    const float lut_scale = (float)(256 - 1) / (float)area;
    int clip_limit = 0;
    if (clip > 0.0) { clip_limit = (int)(clip * area / 256); clip_limit = std::max(clip_limit, 1); }
    std::vector<uint8_t> lut((size_t)tiles_x * tiles_y * 256);
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            int hist[256] = {0};
            for (int y = ty * th; y < (ty + 1) * th; y++)
                for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[src[(size_t)reflect101(y, h) * step + reflect101(x, w)]]++;
            if (clip_limit > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; i++)
                    if (hist[i] > clip_limit) { clipped += hist[i] - clip_limit; hist[i] = clip_limit; }
                const int batch = clipped / 256;
                int residual = clipped - batch * 256;
                for (int i = 0; i < 256; i++) hist[i] += batch;
                if (residual != 0) {
                    const int rstep = std::max(256 / residual, 1);
                    for (int i = 0; i < 256 && residual > 0; i += rstep, residual--) hist[i]++;
                }
            }
            uint8_t* L = &lut[((size_t)ty * tiles_x + tx) * 256];
            int sum = 0;
            for (int i = 0; i < 256; i++) {
                sum += hist[i];
                const int v = cv_roundf((float)sum * lut_scale);
                L[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < h; y++) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = cv_floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        ty1 = std::max(ty1, 0); ty2 = std::min(ty2, tiles_y - 1);
        for (int x = 0; x < w; x++) {
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = cv_floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            tx1 = std::max(tx1, 0); tx2 = std::min(tx2, tiles_x - 1);
            const int v = src[(size_t)y * step + x];
            const float p11 = lut[((size_t)ty1 * tiles_x + tx1) * 256 + v], p12 = lut[((size_t)ty1 * tiles_x + tx2) * 256 + v];
            const float p21 = lut[((size_t)ty2 * tiles_x + tx1) * 256 + v], p22 = lut[((size_t)ty2 * tiles_x + tx2) * 256 + v];
            const float res = (p11 * xa1 + p12 * xa) * ya1 + (p21 * xa1 + p22 * xa) * ya;
            const int r = cv_roundf(res);
            dst[(size_t)y * dstep + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
    return 0;
}

/* pyrDown: separable [1 4 6 4 1] on source samples 2x-2..2x+2 (reflect-101),
 * integer accumulation, one rounding: (sum + 128) >> 8. */
extern "C" void ora_pyr_down(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    Img I{src, w, h, step};
    std::vector<int> rowsum((size_t)5 * dw);
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            int sy = reflect101(2 * y + k - 2, h);
            int* r = &rowsum[(size_t)k * dw];
            for (int x = 0; x < dw; x++) {
                int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = reflect101(2 * x, w),
                    x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                r[x] = I.at(sy, x0) + 4 * I.at(sy, x1) + 6 * I.at(sy, x2) + 4 * I.at(sy, x3) + I.at(sy, x4);
            }
        }
        for (int x = 0; x < dw; x++) {
            int v = rowsum[x] + 4 * rowsum[dw + x] + 6 * rowsum[2 * dw + x] + 4 * rowsum[3 * dw + x] + rowsum[4 * dw + x];
            dst[(size_t)y * dstep + x] = (uint8_t)((v + 128) >> 8);
        }
    }
}

/* buildOpticalFlowPyramid stops when the NEXT level would be <= winSize in either dimension. */
extern "C" int ora_lk_max_level(int w, int h, int win, int max_level) {
    int level = 0;
    for (; level < max_level; level++) {
        int nw = (w + 1) / 2, nh = (h + 1) / 2;
        if (nw <= win || nh <= win) break;
        w = nw; h = nh;
    }
    return level;
}

/* INTER_LINEAR_EXACT coefficient of destination index `val`: source offset and the
 * 8-bit fractional weight of the right/lower neighbour (w0 = 256 - w1). */
static void linear_exact_coeffs(int ssize, int dsize, std::vector<int>& ofs, std::vector<int>& w1) {
    ofs.resize(dsize); w1.resize(dsize);
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1.0 / inv_scale;
    for (int v = 0; v < dsize; v++) {
        double f = scale * ((double)v + 0.5) - 0.5;
        int i = cv_floor(f);
        if (i >= 0 && ssize > 1) {
            if (i < ssize - 1) { ofs[v] = i; w1[v] = cv_round((f - (double)i) * 256.0); }
            else { ofs[v] = ssize - 1; w1[v] = 0; }
        } else { ofs[v] = 0; w1[v] = 0; }
    }
}

extern "C" void ora_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstep,
                                        uint8_t* dst, int dw, int dh, int dstep) {
    std::vector<int> xo, xw, yo, yw;
    linear_exact_coeffs(sw, dw, xo, xw);
    linear_exact_coeffs(sh, dh, yo, yw);
    for (int y = 0; y < dh; y++) {
        const uint8_t* r0 = src + (size_t)yo[y] * sstep;
        const uint8_t* r1 = src + (size_t)std::min(yo[y] + 1, sh - 1) * sstep;
        const uint32_t wy1 = yw[y], wy0 = 256 - wy1;
        for (int x = 0; x < dw; x++) {
            int o = xo[x], o1 = std::min(o + 1, sw - 1);
            uint32_t wx1 = xw[x], wx0 = 256 - wx1;
            uint32_t h0 = r0[o] * wx0 + r0[o1] * wx1; /* 8.8 */
            uint32_t h1 = r1[o] * wx0 + r1[o1] * wx1;
            uint32_t v = h0 * wy0 + h1 * wy1;          /* 16.16 */
            dst[(size_t)y * dstep + x] = (uint8_t)((v + (1u << 15)) >> 16);
        }
    }
}

/* Bit-exact 8.8 fixed-point Gaussian kernel, n=7, sigma=2: softdouble kernel normalised to
 * sum 1, quantised with error diffusion from the edge inwards, centre tap takes the rest. */
extern "C" void ora_gauss7_kernel(int* k7) {
    const int n = 7; const double sigma = 2.0;
    double k[7], sum = 0;
    const double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) { double x = i - (n - 1) * 0.5; k[i] = std::exp(scale2x * x * x); sum += k[i]; }
    for (int i = 0; i < n; i++) k[i] = k[i] * (1.0 / sum);
    double err = 0; long long s = 0;
    for (int i = 0; i < n / 2; i++) {
        double adj = k[i] * 256.0 + err;
        int v = cv_round(adj);
        err = adj - (double)v;
        k7[i] = v; k7[n - 1 - i] = v; s += v;
    }
    k7[n / 2] = (int)(256 - 2 * s);
}

extern "C" void ora_gauss7(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep) {
    int k[7]; ora_gauss7_kernel(k);
    Img I{src, w, h, step};
    std::vector<uint32_t> hb((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int i = 0; i < 7; i++) s += (uint32_t)k[i] * I.at(y, reflect101(x + i - 3, w));
            hb[(size_t)y * w + x] = s; /* 8.8, <= 65280 */
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int i = 0; i < 7; i++) s += (uint32_t)k[i] * hb[(size_t)reflect101(y + i - 3, h) * w + x];
            dst[(size_t)y * dstep + x] = (uint8_t)((s + (1u << 15)) >> 16);
        }
}
