/* Steps directly behind the tracker: forward-backward consistency check and point undistortion. */
#include "vfe_common.h"
#include <cfloat>
#include <cmath>

/* status[i] survives only if the backward track succeeded and came back to within sqrt(max_d2) of the
 * point it started from (squared distance formed in float, one rounding per operation) */
__global__ __launch_bounds__(256) void k_fb_check(const float* __restrict__ orig, const float* __restrict__ back,
                                                  const uint8_t* __restrict__ back_status, const int* __restrict__ n_pts,
                                                  int cap, float max_d2, uint8_t* __restrict__ status) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_pts ? max(0, min(n_pts[b], cap)) : cap;
    if (i >= n) return;
    const size_t o = (size_t)b * cap + i;
    if (!status[o]) return;
    const float dx = back[2 * o] - orig[2 * o], dy = back[2 * o + 1] - orig[2 * o + 1];
    const float d2 = dx * dx + dy * dy;
    if (!back_status[o] || !(d2 <= max_d2)) status[o] = 0;
}

extern "C" int vfe_dev_klt_track_bidir(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                       const float* prev_pts, const int* n_pts, int pts_cap,
                                       float* next_pts, uint8_t* status, float* err,
                                       float* back_pts, uint8_t* back_status,
                                       int win, int max_iter, double eps, double min_eig_thr, double max_recovered_dist2,
                                       void* stream) {
    VFE_REQUIRE(back_pts && back_status, "NULL back_pts / back_status");
    VFE_REQUIRE(max_recovered_dist2 >= 0, "max_recovered_dist2 must be >= 0");
    int rc = vfe_dev_klt_track(prev_levels, next_levels, nlevels, prev_pts, n_pts, pts_cap, next_pts, status, err, win, max_iter,
                               eps, min_eig_thr, 0, stream);
    if (rc) return rc;
    const int batch = prev_levels[0].batch;
    hipStream_t st = (hipStream_t)stream;
    /* the backward track starts from the original positions (OPTFLOW_USE_INITIAL_FLOW) */
    VFE_HIP(hipMemcpyAsync(back_pts, prev_pts, (size_t)batch * pts_cap * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    rc = vfe_dev_klt_track(next_levels, prev_levels, nlevels, next_pts, n_pts, pts_cap, back_pts, back_status, nullptr, win, max_iter,
                           eps, min_eig_thr, 1, stream);
    if (rc) return rc;
    ProfScope prof(VFE_K_MISC, st);
    hipLaunchKernelGGL(k_fb_check, dim3(div_up(pts_cap, 256), batch), dim3(256), 0, st, prev_pts, back_pts, back_status, n_pts,
                       pts_cap, (float)max_recovered_dist2, status);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

/* cv::undistortPoints for the plumb-bob (radtan) model, no rectification, normalised output:
 * five fixed-point iterations in double, as OpenCV's default criteria (5 iterations) perform */
__global__ __launch_bounds__(256) void k_undistort(const float* __restrict__ in, float* __restrict__ out, long long n,
                                                   double fx, double fy, double cx, double cy,
                                                   double k1, double k2, double p1, double p2, double k3) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x0 = ((double)in[2 * i] - cx) / fx, y0 = ((double)in[2 * i + 1] - cy) / fy;
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

extern "C" int vfe_dev_undistort_points(const float* pts, float* out, long long n, const double* camera4,
                                        const double* dist5, void* stream) {
    VFE_REQUIRE(pts && out && camera4 && dist5 && n >= 0, "NULL buffer or n < 0");
    VFE_REQUIRE(camera4[0] != 0.0 && camera4[1] != 0.0, "focal lengths must be non-zero");
    if (n == 0) return VFE_OK;
    ProfScope prof(VFE_K_MISC, (hipStream_t)stream);
    hipLaunchKernelGGL(k_undistort, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts, out, n,
                       camera4[0], camera4[1], camera4[2], camera4[3], dist5[0], dist5[1], dist5[2], dist5[3], dist5[4]);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

/* ---- cv::cornerSubPix ---------------------------------------------------------------------------
 * One wavefront per corner.  Per iteration the lanes interpolate the (2*win+3)^2 float patch around the
 * current estimate into LDS, form the five weighted products of every window position in double, and five
 * lanes add them up in row-major order (the order of a sequential implementation, so results do not depend
 * on the lane count); every lane then takes the same Newton step. */
#define SUBPIX_MAX_WIN 10
struct SubpixMask { float m[(2 * SUBPIX_MAX_WIN + 1) * (2 * SUBPIX_MAX_WIN + 1)]; };

__global__ __launch_bounds__(64) void k_corner_subpix(DImg img, float* __restrict__ pts, const int* __restrict__ n_pts, int cap,
                                                      int win, int max_iter, double eps2, SubpixMask M) {
    extern __shared__ double s_terms[]; /* [5][wn*wn] doubles, then the patch [pw*pw] floats */
    const int lane = threadIdx.x, b = blockIdx.y, p = blockIdx.x;
    const int n = n_pts ? max(0, min(n_pts[b], cap)) : cap;
    if (p >= n) return;
    const int wn = 2 * win + 1, pw = wn + 2, nw = wn * wn;
    float* s_patch = reinterpret_cast<float*>(s_terms + 5 * nw);
    const uint8_t* ip = img.frame(b);
    const int w = img.w, h = img.h;
    float* out = pts + ((size_t)b * cap + p) * 2;
    const float tx = out[0], ty = out[1];
    float cx = tx, cy = ty;
    int iter = 0;
    double err = 0;
    do {
        const float ox = cx - (float)(pw - 1) * 0.5f, oy = cy - (float)(pw - 1) * 0.5f;
        const int ix = (int)floorf(ox), iy = (int)floorf(oy);
        const float fa = ox - (float)ix, fb = oy - (float)iy;
        const float a11 = (1.f - fa) * (1.f - fb), a12 = fa * (1.f - fb), a21 = (1.f - fa) * fb, a22 = fa * fb;
        for (int k = lane; k < pw * pw; k += 64) {
            const int i = k / pw, j = k - i * pw;
            const int y0 = min(max(iy + i, 0), h - 1), y1 = min(max(iy + i + 1, 0), h - 1);
            const int x0 = min(max(ix + j, 0), w - 1), x1 = min(max(ix + j + 1, 0), w - 1);
            const uint8_t* r0 = ip + (size_t)y0 * img.pitch;
            const uint8_t* r1 = ip + (size_t)y1 * img.pitch;
            s_patch[k] = (float)r0[x0] * a11 + (float)r0[x1] * a12 + (float)r1[x0] * a21 + (float)r1[x1] * a22;
        }
        __syncthreads();
        for (int k = lane; k < nw; k += 64) {
            const int i = k / wn, j = k - i * wn;
            const float* s = s_patch + (i + 1) * pw + (j + 1);
            const double m = (double)M.m[k];
            const double tgx = (double)(s[1] - s[-1]);
            const double tgy = (double)(s[pw] - s[-pw]);
            const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
            const double px = (double)(j - win), py = (double)(i - win);
            s_terms[k] = gxx; s_terms[nw + k] = gxy; s_terms[2 * nw + k] = gyy;
            s_terms[3 * nw + k] = gxx * px + gxy * py;
            s_terms[4 * nw + k] = gxy * px + gyy * py;
        }
        __syncthreads();
        double r = 0;
        if (lane < 5) {
            const double* t = s_terms + lane * nw;
            for (int k = 0; k < nw; k++) r += t[k];
        }
        const double a = __shfl(r, 0), bq = __shfl(r, 1), c = __shfl(r, 2), bb1 = __shfl(r, 3), bb2 = __shfl(r, 4);
        __syncthreads();
        const double det = a * c - bq * bq;
        if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
        const double scale = 1.0 / det;
        const float nx = (float)((double)cx + c * scale * bb1 - bq * scale * bb2);
        const float ny = (float)((double)cy - bq * scale * bb1 + a * scale * bb2);
        err = (double)(nx - cx) * (double)(nx - cx) + (double)(ny - cy) * (double)(ny - cy);
        cx = nx; cy = ny;
        if (cx < 0 || cx >= (float)w || cy < 0 || cy >= (float)h) break;
    } while (++iter < max_iter && err > eps2);
    if (fabsf(cx - tx) > (float)win || fabsf(cy - ty) > (float)win) { cx = tx; cy = ty; }
    if (lane == 0) { out[0] = cx; out[1] = cy; }
}

extern "C" int vfe_dev_corner_subpix(const vfe_dimg* img, float* pts, const int* n_pts, int pts_cap, int win, int zero_zone,
                                     int max_iter, double eps, void* stream) {
    int rc = vfe_check_dimg(img, "img");
    if (rc) return rc;
    VFE_REQUIRE(pts && pts_cap > 0, "NULL pts or pts_cap <= 0");
    VFE_REQUIRE(win >= 1 && win <= SUBPIX_MAX_WIN, "win (half window) must be 1..10");
    VFE_REQUIRE(img->batch <= 65535, "batch too large");
    SubpixMask M;
    const int wn = 2 * win + 1;
    /* window weights exp(-(x/win)^2) * exp(-(y/win)^2), formed in float on the host */
    for (int i = 0; i < wn; i++) {
        const float y = (float)(i - win) / (float)win;
        const float vy = std::exp(-y * y);
        for (int j = 0; j < wn; j++) {
            const float x = (float)(j - win) / (float)win;
            M.m[i * wn + j] = (float)(vy * std::exp(-x * x));
        }
    }
    if (zero_zone >= 0 && zero_zone * 2 + 1 < wn)
        for (int i = win - zero_zone; i <= win + zero_zone; i++)
            for (int j = win - zero_zone; j <= win + zero_zone; j++) M.m[i * wn + j] = 0.f;
    max_iter = max_iter < 1 ? 1 : (max_iter > 100 ? 100 : max_iter);
    if (eps < 0) eps = 0;
    const size_t lds = (size_t)5 * wn * wn * sizeof(double) + (size_t)(wn + 2) * (wn + 2) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(VFE_K_MISC, st);
    hipLaunchKernelGGL(k_corner_subpix, dim3(pts_cap, img->batch), dim3(64), lds, st, to_dimg(img), pts, n_pts, pts_cap, win, max_iter,
                       eps * eps, M);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

/* ---- fundamental matrix by RANSAC (outlier filter behind the tracker) ---------------------------------
 * cv::findFundamentalMat(FM_RANSAC)-shaped: 7-point minimal solver, larger of the two squared epipolar
 * distances as the error, best model by inlier count, iteration budget shrinking with the inlier ratio.
 * Sampling and the numerical steps are fixed so that the result does not depend on the schedule:
 * sample k = seven distinct indices from mix32(seed, 64 k + attempt) mod n; null space of the 7x9 system by
 * Gauss-Jordan elimination with row pivoting; real roots of the cubic by bracketing between its critical
 * points and 100 bisection steps; the logarithms of the budget formula by an atanh series -- only
 * +, -, *, / and sqrt in double.
 * One workgroup of 16 waves per frame: every thread solves one sample (all samples up front), then the waves
 * count inliers for 16 hypotheses at a time and thread 0 replays the sequential best-model / budget update in
 * hypothesis order, so the loop ends exactly where a sequential RANSAC would. */
#define FR_MAX_PTS 2048
#define FR_MAX_ITERS 1024

namespace {
__device__ inline uint32_t fr_mix32(uint32_t a, uint32_t b) {
    uint32_t h = a ^ (b * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

__device__ inline void fr_sample(uint32_t seed, int k, int n, int* idx) {
    uint32_t attempt = 0;
    for (int j = 0; j < 7; j++) {
        for (;;) {
            const int c = (int)(fr_mix32(seed, (uint32_t)k * 64u + attempt) % (uint32_t)n);
            attempt++;
            bool dup = false;
            for (int i = 0; i < j; i++) dup = dup || idx[i] == c;
            if (!dup || attempt >= 64u) { idx[j] = c; break; }
        }
    }
}

__device__ inline double fr_det3(const double* u, const double* v, const double* w) {
    return u[0] * (v[1] * w[2] - v[2] * w[1]) - u[1] * (v[0] * w[2] - v[2] * w[0]) + u[2] * (v[0] * w[1] - v[1] * w[0]);
}

__device__ inline double fr_cubic_at(const double* c, double x) { return ((c[3] * x + c[2]) * x + c[1]) * x + c[0]; }

__device__ inline double fr_bisect(const double* c, double lo, double hi) {
    double flo = fr_cubic_at(c, lo);
    for (int i = 0; i < 100; i++) {
        const double mid = 0.5 * (lo + hi);
        const double fm = fr_cubic_at(c, mid);
        if ((flo <= 0 && fm <= 0) || (flo >= 0 && fm >= 0)) { lo = mid; flo = fm; } else hi = mid;
    }
    return 0.5 * (lo + hi);
}

__device__ inline int fr_cubic_roots(const double* cin, double* roots) {
    double c[4] = {cin[0], cin[1], cin[2], cin[3]};
    double big = 0;
    for (int i = 0; i < 4; i++) big = fmax(big, fabs(c[i]));
    if (!(big > 0)) return 0;
    for (int i = 0; i < 4; i++) c[i] = c[i] / big;
    if (fabs(c[3]) < 1e-14) {
        if (fabs(c[2]) < 1e-14) {
            if (fabs(c[1]) < 1e-14) return 0;
            roots[0] = -c[0] / c[1];
            return 1;
        }
        const double disc = c[1] * c[1] - 4.0 * c[2] * c[0];
        if (disc < 0) return 0;
        const double s = sqrt(disc);
        const double r0 = (-c[1] - s) / (2.0 * c[2]), r1 = (-c[1] + s) / (2.0 * c[2]);
        roots[0] = fmin(r0, r1); roots[1] = fmax(r0, r1);
        return 2;
    }
    const double B = 1.0 + fmax(fabs(c[0] / c[3]), fmax(fabs(c[1] / c[3]), fabs(c[2] / c[3])));
    const double disc = c[2] * c[2] - 3.0 * c[3] * c[1];
    double edge[4];
    int ne = 0;
    edge[ne++] = -B;
    if (disc > 0) {
        const double s = sqrt(disc);
        const double x0 = (-c[2] - s) / (3.0 * c[3]), x1 = (-c[2] + s) / (3.0 * c[3]);
        edge[ne++] = fmin(x0, x1); edge[ne++] = fmax(x0, x1);
    }
    edge[ne++] = B;
    int nr = 0;
    for (int i = 0; i + 1 < ne; i++) {
        const double lo = edge[i], hi = edge[i + 1];
        const double flo = fr_cubic_at(c, lo), fhi = fr_cubic_at(c, hi);
        if ((flo < 0 && fhi > 0) || (flo > 0 && fhi < 0) || (flo == 0 && i == 0)) roots[nr++] = fr_bisect(c, lo, hi);
        else if (fhi == 0) roots[nr++] = hi;
    }
    return nr;
}

__device__ inline int fr_seven_point(const double (*m1)[2], const double (*m2)[2], double (*F)[9]) {
    double A[7][9];
    double big = 0;
    for (int i = 0; i < 7; i++) {
        const double x1 = m1[i][0], y1 = m1[i][1], x2 = m2[i][0], y2 = m2[i][1];
        A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2; A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2;
        A[i][6] = x1; A[i][7] = y1; A[i][8] = 1.0;
        for (int j = 0; j < 9; j++) big = fmax(big, fabs(A[i][j]));
    }
    const double tiny = big * 1e-10;
    int piv[7], freec[2], nfree = 0, r = 0;
    for (int col = 0; col < 9; col++) {
        if (r == 7) { if (nfree == 2) return 0; freec[nfree++] = col; continue; }
        int p = r;
        double best = fabs(A[r][col]);
        for (int i = r + 1; i < 7; i++)
            if (fabs(A[i][col]) > best) { best = fabs(A[i][col]); p = i; }
        if (!(best > tiny)) { if (nfree == 2) return 0; freec[nfree++] = col; continue; }
        if (p != r)
            for (int j = 0; j < 9; j++) { const double t = A[p][j]; A[p][j] = A[r][j]; A[r][j] = t; }
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
        for (int j = 0; j < 9; j++) m = fmax(m, fabs(f[j]));
        for (int j = 0; j < 9; j++) f[j] = f[j] / m;
    }
    double d[9];
    for (int j = 0; j < 9; j++) d[j] = f1[j] - f2[j];
    double c[4];
    c[0] = fr_det3(f2, f2 + 3, f2 + 6);
    c[1] = fr_det3(d, f2 + 3, f2 + 6) + fr_det3(f2, d + 3, f2 + 6) + fr_det3(f2, f2 + 3, d + 6);
    c[2] = fr_det3(d, d + 3, f2 + 6) + fr_det3(d, f2 + 3, d + 6) + fr_det3(f2, d + 3, d + 6);
    c[3] = fr_det3(d, d + 3, d + 6);
    double roots[3];
    const int nr = fr_cubic_roots(c, roots);
    for (int k = 0; k < nr; k++) {
        double* Fk = F[k];
        for (int j = 0; j < 9; j++) Fk[j] = f2[j] + roots[k] * d[j];
        if (fabs(Fk[8]) > DBL_EPSILON) {
            const double mu = 1.0 / Fk[8];
            for (int j = 0; j < 8; j++) Fk[j] = Fk[j] * mu;
            Fk[8] = 1.0;
        }
    }
    return nr;
}

__device__ inline bool fr_inlier(const double* F, double x1, double y1, double x2, double y2, float thr2) {
    double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1.0 / (a * a + b * b);
    const double d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1.0 / (a * a + b * b);
    const double d1 = x1 * a + y1 * b + c;
    const float err = (float)fmax(d1 * d1 * s1, d2 * d2 * s2);
    return err <= thr2;
}

__device__ inline double fr_log(double x) {
    int e = 0;
    while (x >= 1.4142135623730951) { x = x * 0.5; e++; }
    while (x < 0.7071067811865476) { x = x * 2.0; e--; }
    const double s = (x - 1.0) / (x + 1.0), s2 = s * s;
    double term = s, sum = 0;
    for (int k = 0; k < 16; k++) { sum = sum + term / (double)(2 * k + 1); term = term * s2; }
    return 2.0 * sum + (double)e * 0.6931471805599453;
}

__device__ inline int fr_update_iters(double p, double ep, int max_iters) {
    p = fmin(fmax(p, 0.0), 1.0);
    ep = fmin(fmax(ep, 0.0), 1.0);
    const double num = fmax(1.0 - p, DBL_MIN);
    const double q = 1.0 - ep;
    const double denom = 1.0 - q * q * q * q * q * q * q;
    if (denom < DBL_MIN) return 0;
    const double ln = fr_log(num), ld = fr_log(denom);
    if (ld >= 0 || -ln >= (double)max_iters * (-ld)) return max_iters;
    return __double2int_rn(ln / ld);
}
}  // namespace

__global__ __launch_bounds__(1024) void k_fundamental_ransac(const float* __restrict__ p1, const float* __restrict__ p2,
                                                             const int* __restrict__ n_pts, int cap, uint8_t* __restrict__ status,
                                                             float thr2, double confidence, int max_iters, uint32_t seed,
                                                             double* __restrict__ models, double* __restrict__ F_out,
                                                             int* __restrict__ inliers) {
    __shared__ float4 s_pts[FR_MAX_PTS];         /* participating points (x1, y1, x2, y2), in index order */
    __shared__ unsigned short s_orig[FR_MAX_PTS]; /* their indices */
    __shared__ uint8_t s_nm[FR_MAX_ITERS];
    __shared__ int s_wave[16], s_cnt[16][3];
    __shared__ int s_np, s_niters, s_best_count, s_best_k, s_best_m;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    const int n = n_pts ? max(0, min(n_pts[b], cap)) : cap;
    const float* a1 = p1 + (size_t)b * cap * 2;
    const float* a2 = p2 + (size_t)b * cap * 2;
    uint8_t* st = status + (size_t)b * cap;
    double* mdl = models + (size_t)b * FR_MAX_ITERS * 27;

    /* ordered compaction of the participating points: thread t looks at points 2t, 2t+1 */
    {
        const int i0 = 2 * tid, i1 = 2 * tid + 1;
        const bool f0 = i0 < n && st[i0] != 0, f1 = i1 < n && st[i1] != 0;
        int c = (f0 ? 1 : 0) + (f1 ? 1 : 0), incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wv; w++) base += s_wave[w];
        int pos = base + incl - c;
        if (f0) { s_pts[pos] = make_float4(a1[2 * i0], a1[2 * i0 + 1], a2[2 * i0], a2[2 * i0 + 1]); s_orig[pos] = (unsigned short)i0; pos++; }
        if (f1) { s_pts[pos] = make_float4(a1[2 * i1], a1[2 * i1 + 1], a2[2 * i1], a2[2 * i1 + 1]); s_orig[pos] = (unsigned short)i1; }
        if (tid == 1023) s_np = base + incl;
        __syncthreads();
    }
    const int np = s_np;
    if (np < 8) { /* nothing to reject */
        if (tid < 9) F_out[(size_t)b * 9 + tid] = 0.0;
        if (tid == 0) inliers[b] = np;
        return;
    }
    /* every thread solves one sample */
    if (tid < max_iters) {
        int idx[7];
        fr_sample(seed, tid, np, idx);
        double m1[7][2], m2[7][2], F[3][9];
        for (int j = 0; j < 7; j++) {
            const float4 q = s_pts[idx[j]];
            m1[j][0] = q.x; m1[j][1] = q.y; m2[j][0] = q.z; m2[j][1] = q.w;
        }
        const int nm = fr_seven_point(m1, m2, F);
        s_nm[tid] = (uint8_t)nm;
        for (int m = 0; m < nm; m++)
            for (int j = 0; j < 9; j++) mdl[((size_t)tid * 3 + m) * 9 + j] = F[m][j];
    }
    if (tid == 0) { s_niters = max_iters; s_best_count = 0; s_best_k = -1; s_best_m = 0; }
    __threadfence_block();
    __syncthreads(); /* the models written above are visible to the whole group */
    for (int base = 0; base < s_niters; base += 16) {
        const int k = base + wv;
        if (k < max_iters && k < s_niters) {
            const int nm = s_nm[k];
            for (int m = 0; m < nm; m++) {
                double F[9];
#pragma unroll
                for (int j = 0; j < 9; j++) F[j] = mdl[((size_t)k * 3 + m) * 9 + j];
                int good = 0;
                for (int q = lane; q < np; q += 64) {
                    const float4 v = s_pts[q];
                    good += fr_inlier(F, v.x, v.y, v.z, v.w, thr2) ? 1 : 0;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) good += __shfl_xor(good, o);
                if (lane == 0) s_cnt[wv][m] = good;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int niters = s_niters, best_count = s_best_count;
            for (int w = 0; w < 16; w++) {
                const int kk = base + w;
                if (kk >= niters || kk >= max_iters) break;
                const int nm = s_nm[kk];
                for (int m = 0; m < nm; m++) {
                    const int good = s_cnt[w][m];
                    if (good > max(best_count, 6)) {
                        best_count = good; s_best_k = kk; s_best_m = m;
                        niters = fr_update_iters(confidence, (double)(np - good) / (double)np, niters);
                    }
                }
            }
            s_niters = niters; s_best_count = best_count;
        }
        __syncthreads();
    }
    if (s_best_count == 0) {
        for (int i = tid; i < n; i += 1024) st[i] = 0;
        if (tid < 9) F_out[(size_t)b * 9 + tid] = 0.0;
        if (tid == 0) inliers[b] = 0;
        return;
    }
    double F[9];
#pragma unroll
    for (int j = 0; j < 9; j++) F[j] = mdl[((size_t)s_best_k * 3 + s_best_m) * 9 + j];
    for (int q = tid; q < np; q += 1024) {
        const float4 v = s_pts[q];
        st[s_orig[q]] = fr_inlier(F, v.x, v.y, v.z, v.w, thr2) ? 1 : 0;
    }
    if (tid < 9) F_out[(size_t)b * 9 + tid] = F[tid];
    if (tid == 0) inliers[b] = s_best_count;
}

extern "C" size_t vfe_fundamental_workspace(int batch) {
    if (batch <= 0) return 0;
    return align_up((size_t)batch * FR_MAX_ITERS * 27 * sizeof(double), 256);
}

extern "C" int vfe_dev_fundamental_ransac(const float* pts1, const float* pts2, const int* n_pts, int pts_cap, int batch,
                                          uint8_t* status, double threshold, double confidence, int max_iters, unsigned seed,
                                          double* F_out, int* inliers, void* workspace, size_t workspace_bytes, void* stream) {
    VFE_REQUIRE(pts1 && pts2 && status && F_out && inliers, "NULL buffer");
    VFE_REQUIRE(batch > 0 && pts_cap > 0 && pts_cap <= FR_MAX_PTS, "pts_cap must be 1..2048");
    const size_t need = vfe_fundamental_workspace(batch);
    if (!workspace || workspace_bytes < need) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
        return VFE_E_NOMEM;
    }
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    if (threshold <= 0) threshold = 3;
    max_iters = max_iters < 1 ? 1 : (max_iters > FR_MAX_ITERS ? FR_MAX_ITERS : max_iters);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(VFE_K_MISC, st);
    hipLaunchKernelGGL(k_fundamental_ransac, dim3(batch), dim3(1024), 0, st, pts1, pts2, n_pts, pts_cap, status,
                       (float)(threshold * threshold), confidence, max_iters, (uint32_t)seed, (double*)workspace, F_out, inliers);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
