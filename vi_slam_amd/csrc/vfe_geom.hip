/* Steps directly behind the tracker: forward-backward consistency check and point undistortion. */
#include "vfe_common.h"

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
