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
