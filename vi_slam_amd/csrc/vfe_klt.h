/* Shared between the generic and the 21x21 Lucas-Kanade kernels. */
#pragma once
#include "vfe_common.h"

#define KLT_MAX_LEVELS 8

struct KltParams {
    DImg prev[KLT_MAX_LEVELS], next[KLT_MAX_LEVELS];
    int nlevels;
    const float* prev_pts;
    const int* n_pts;
    int cap;
    float* next_pts;
    uint8_t* status;
    float* err;
    int win, max_iter;
    double epsilon; /* squared */
    double min_eig_thr;
    int use_initial;
    unsigned total_work;
    /* optional processing order [batch][cap]: the wave that would take points 4k .. 4k+3 of a frame takes points
     * order[4k .. 4k+3] instead (a permutation of 0 .. n-1; entries past a frame's count are not read).  Results do
     * not depend on it; neighbours in the image share cache lines and tend to need the same number of iterations. */
    const unsigned short* order;
};

__device__ static inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

/* 14-bit integer bilinear weights, rounded as cv::calcOpticalFlowPyrLK does (cvRound, last weight by difference) */
__device__ static inline void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
    w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << 14));
    w01 = __float2int_rn(a * (1.f - b) * (float)(1 << 14));
    w10 = __float2int_rn((1.f - a) * b * (float)(1 << 14));
    w11 = (1 << 14) - w00 - w01 - w10;
}

/* enqueue the 21x21 kernel for p.total_work (frame, point) items */
void klt21_launch(const KltParams& p, hipStream_t stream);
/* whether klt21_launch takes its one-wave-per-point kernel for this many (frame, point) slots; that kernel has a
 * workgroup per point, so a processing order buys nothing there */
bool klt21_one_wave_per_point(unsigned total_work);

/* vfe_dev_klt_track with an optional processing order (see KltParams::order) */
int klt_track_ordered(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                      const float* prev_pts, const int* n_pts, int pts_cap, float* next_pts, uint8_t* status, float* err,
                      int win, int max_iter, double eps, double min_eig_thr, int use_initial_flow,
                      const unsigned short* order, hipStream_t stream);
/* fills order[batch][cap] with every frame's points sorted by rim class (the coarsest of the `nlevels` pyramid levels at
 * which the point's window overhangs the level), then by image tile (counting sort, one workgroup per frame); cap <= 65536 */
int klt_order_launch(const float* pts, const int* n_pts, int cap, int batch, int w, int h, int nlevels, unsigned short* order, hipStream_t stream);
