/* CPU oracle (test infrastructure only): ORB keypoints + steered rBRIEF descriptors.
 * Restates cv::ORB::detectAndCompute (OpenCV 4.5 orb.cpp) with the parameters the
 * reference passes at src/slam/adapters/orbslam3_adapter.cpp:76-79: everything default
 * except nfeatures (scale 1.2, 8 levels, edgeThreshold 31, firstLevel 0, WTA_K 2,
 * HARRIS_SCORE, patchSize 31, fastThreshold 20). */
#include "ora_common.h"
#include "vfe_oracle.h"
#include <cfloat>

using namespace ora;

namespace {

const int PATTERN[256 * 4] = {
#include "orb_pattern.inc"
};

const int EDGE = 31, PATCH = 31, HALF_PATCH = 15, HARRIS_BLOCK = 7;
const float HARRIS_K = 0.04f;

float get_scale(int level, float scale_factor) { return (float)std::pow((double)scale_factor, (double)level); }

/* cv::fastAtan2: 7th-order odd polynomial, degrees in [0, 360) */
float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI),
                p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON); c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON); c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

void make_umax(int* umax /*HALF_PATCH+2*/) {
    int v, v0, vmax = cv_floor(HALF_PATCH * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0; ++v0;
    }
}

struct Kp { int x, y, fast; float harris; };

/* KeyPointsFilter::retainBest: keep all with response >= the n-th largest (ties kept) */
template <class T, class F>
void retain_best(std::vector<Kp>& k, int n, F resp, T /*tag*/) {
    if (n < 0 || (int)k.size() <= n) return;
    if (n == 0) { k.clear(); return; }
    std::vector<T> r(k.size());
    for (size_t i = 0; i < k.size(); i++) r[i] = resp(k[i]);
    std::nth_element(r.begin(), r.begin() + (n - 1), r.end(), std::greater<T>());
    T thr = r[n - 1];
    std::vector<Kp> out;
    for (auto& p : k) if (resp(p) >= thr) out.push_back(p);
    k.swap(out);
}

}  // namespace

extern "C" int ora_orb_level_sizes(int w, int h, float scale_factor, int nlevels, int* wh, float* scales) {
    for (int l = 0; l < nlevels; l++) {
        float scale = get_scale(l, scale_factor);
        float inv = 1.0f / scale;
        wh[2 * l] = cv_roundf(w * inv); wh[2 * l + 1] = cv_roundf(h * inv);
        if (scales) scales[l] = scale;
    }
    return 0;
}

extern "C" int ora_orb_quotas(int nfeatures, float scale_factor, int nlevels, int* quota) {
    float factor = (float)(1.0 / scale_factor);
    float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { quota[l] = cv_roundf(nd); sum += quota[l]; nd *= factor; }
    quota[nlevels - 1] = std::max(nfeatures - sum, 0);
    return 0;
}

extern "C" int ora_orb_extract(const uint8_t* img, int w, int h, int step,
                               int nfeatures, float scale_factor, int nlevels, int fast_threshold,
                               ora_orb_kp* kps, uint8_t* desc, int cap) {
    std::vector<int> wh(2 * nlevels), quota(nlevels);
    std::vector<float> scales(nlevels);
    ora_orb_level_sizes(w, h, scale_factor, nlevels, wh.data(), scales.data());
    ora_orb_quotas(nfeatures, scale_factor, nlevels, quota.data());
    int umax[HALF_PATCH + 2]; make_umax(umax);

    /* pyramid: level l = resize(level l-1, INTER_LINEAR_EXACT) */
    std::vector<std::vector<uint8_t>> pyr(nlevels);
    pyr[0].resize((size_t)w * h);
    for (int y = 0; y < h; y++) std::memcpy(&pyr[0][(size_t)y * w], img + (size_t)y * step, w);
    for (int l = 1; l < nlevels; l++) {
        pyr[l].resize((size_t)wh[2 * l] * wh[2 * l + 1]);
        ora_resize_linear_exact(pyr[l - 1].data(), wh[2 * l - 2], wh[2 * l - 1], wh[2 * l - 2],
                                pyr[l].data(), wh[2 * l], wh[2 * l + 1], wh[2 * l]);
    }

    const float hscale = 1.f / ((1 << 2) * HARRIS_BLOCK * 255.f);
    const float hscale4 = hscale * hscale * hscale * hscale;

    int total = 0;
    for (int l = 0; l < nlevels; l++) {
        const int lw = wh[2 * l], lh = wh[2 * l + 1];
        Img I{pyr[l].data(), lw, lh, lw};
        std::vector<Kp> k;
        if (lw > 2 * EDGE && lh > 2 * EDGE) {
            std::vector<int> raw((size_t)lw * lh * 3 / 4 + 16);
            int cnt = ora_fast9(I.d, lw, lh, lw, fast_threshold, 1, raw.data(), (int)raw.size() / 3);
            for (int i = 0; i < cnt; i++) {
                int x = raw[3 * i], y = raw[3 * i + 1];
                /* runByImageBorder: keep inside [EDGE, dim-EDGE) */
                if (x >= EDGE && x < lw - EDGE && y >= EDGE && y < lh - EDGE) k.push_back({x, y, raw[3 * i + 2], 0.f});
            }
        }
        retain_best(k, 2 * quota[l], [](const Kp& p) { return p.fast; }, int());
        /* Harris response, 7x7 block of 3x3 Sobel on the 8-bit level (reflect-101 outside) */
        for (auto& p : k) {
            int a = 0, b = 0, c = 0;
            for (int i = 0; i < HARRIS_BLOCK; i++)
                for (int j = 0; j < HARRIS_BLOCK; j++) {
                    int y = p.y - HARRIS_BLOCK / 2 + i, x = p.x - HARRIS_BLOCK / 2 + j;
                    int Ix = (I.at_r(y, x + 1) - I.at_r(y, x - 1)) * 2 + (I.at_r(y - 1, x + 1) - I.at_r(y - 1, x - 1)) +
                             (I.at_r(y + 1, x + 1) - I.at_r(y + 1, x - 1));
                    int Iy = (I.at_r(y + 1, x) - I.at_r(y - 1, x)) * 2 + (I.at_r(y + 1, x - 1) - I.at_r(y - 1, x - 1)) +
                             (I.at_r(y + 1, x + 1) - I.at_r(y - 1, x + 1));
                    a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
                }
            float fa = (float)a, fb = (float)b, fc = (float)c;
            float s = fa + fb;
            p.harris = ((fa * fb - fc * fc) - (HARRIS_K * s) * s) * hscale4;
        }
        retain_best(k, quota[l], [](const Kp& p) { return p.harris; }, float());
        std::sort(k.begin(), k.end(), [](const Kp& a, const Kp& b) { return a.y != b.y ? a.y < b.y : a.x < b.x; });

        /* descriptors are sampled from the blurred level */
        std::vector<uint8_t> blur((size_t)lw * lh);
        if (!k.empty()) ora_gauss7(I.d, lw, lh, lw, blur.data(), lw);
        const float sf = scales[l];
        for (auto& p : k) {
            if (total >= cap) return -1;
            /* intensity-centroid orientation over the circular patch of radius 15 */
            int m01 = 0, m10 = 0;
            for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m10 += u * I.at(p.y, p.x + u);
            for (int v = 1; v <= HALF_PATCH; ++v) {
                int vsum = 0, d = umax[v];
                for (int u = -d; u <= d; ++u) {
                    int vp = I.at(p.y + v, p.x + u), vm = I.at(p.y - v, p.x + u);
                    vsum += (vp - vm); m10 += u * (vp + vm);
                }
                m01 += v * vsum;
            }
            ora_orb_kp& o = kps[total];
            o.angle = fast_atan2((float)m01, (float)m10);
            o.octave = l; o.lx = p.x; o.ly = p.y; o.response = p.harris; o.size = PATCH * sf;
            o.x = (float)p.x * sf; o.y = (float)p.y * sf;
            /* steered BRIEF: centre recomputed from the scaled position as OpenCV does */
            float inv = 1.f / sf;
            int cx = cv_roundf(o.x * inv), cy = cv_roundf(o.y * inv);
            float ang = o.angle * (float)(M_PI / 180.f);
            float sa, ca; sincos_det(ang, &sa, &ca);
            uint8_t* dsc = desc + (size_t)total * 32;
            for (int i = 0; i < 32; i++) {
                int val = 0;
                for (int j = 0; j < 8; j++) {
                    const int* pt = &PATTERN[(i * 8 + j) * 4];
                    float x0 = pt[0] * ca - pt[1] * sa, y0 = pt[0] * sa + pt[1] * ca;
                    float x1 = pt[2] * ca - pt[3] * sa, y1 = pt[2] * sa + pt[3] * ca;
                    int t0 = blur[(size_t)(cy + cv_roundf(y0)) * lw + cx + cv_roundf(x0)];
                    int t1 = blur[(size_t)(cy + cv_roundf(y1)) * lw + cx + cv_roundf(x1)];
                    val |= (t0 < t1) << j;
                }
                dsc[i] = (uint8_t)val;
            }
            ++total;
        }
    }
    return total;
}

/* the deterministic sine / cosine behind the steered pattern, exported so that tests can compare it with the C library's
 * cosf / sinf (what cv::ORB calls) */
extern "C" void ora_sincos_det(float x, float* s, float* c) { ora::sincos_det(x, s, c); }
