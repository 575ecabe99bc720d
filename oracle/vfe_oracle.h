/*
 * C entry points of the CPU oracle (test infrastructure only, see ora_common.h).
 * Every function works on host memory, is single-threaded and deterministic.
 * Images are 8-bit, row-major, `step` bytes per row.
 */
#ifndef VFE_ORACLE_H
#define VFE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* cvtColor(BGR2GRAY), 8-bit fixed point (call sites openvins_adapter.cpp:72,
 * basalt_adapter.cpp:44, orbslam3_adapter.cpp:71). */
void ora_bgr2gray(const uint8_t* bgr, int w, int h, int step, uint8_t* gray, int gstep);

/* equalizeHist on 8-bit gray (config/vins_mono.yaml:104 `equalize`). */
void ora_equalize_hist(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep);
/* CLAHE (`histogram_method: "CLAHE"`, docs/reference/15_openvins_smartphone_config.md:111); 0 on success. */
int ora_clahe(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep, double clip, int tiles_x, int tiles_y);

/* pyrDown: dst is ((w+1)/2) x ((h+1)/2). */
void ora_pyr_down(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep);

/* Number of the last pyramid level buildOpticalFlowPyramid would produce. */
int ora_lk_max_level(int w, int h, int win, int max_level);

/* resize(INTER_LINEAR_EXACT) and GaussianBlur(7x7, sigma=2, reflect-101), as ORB uses them. */
void ora_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstep,
                             uint8_t* dst, int dw, int dh, int dstep);
void ora_gauss7(const uint8_t* src, int w, int h, int step, uint8_t* dst, int dstep);
void ora_gauss7_kernel(int* k7);

/* cornerMinEigenVal(blockSize=3, ksize=3) -> eig (w*h floats, dense). */
void ora_min_eigen_map(const uint8_t* img, int w, int h, int step, float* eig);

/* goodFeaturesToTrack(img, corners, max_corners, quality, min_distance) with blockSize 3,
 * no mask, Shi-Tomasi score (openvins_adapter.cpp:79-80, basalt_adapter.cpp:50-53).
 * seeds (n_seed points, integer-valued x,y pairs; may be NULL) are treated as already
 * accepted corners for the min-distance test but are not returned and do not count
 * towards max_corners.  Returns the number of corners written (x,y pairs) or -1 if
 * `cap` is too small. */
int ora_good_features(const uint8_t* img, int w, int h, int step,
                      int max_corners, double quality, double min_distance,
                      const float* seeds, int n_seed,
                      float* corners_xy, float* scores, int cap);

/* calcOpticalFlowPyrLK on pyramids built with ora_pyr_down.  pts are (x,y) pairs.
 * status/err follow OpenCV (err = mean absolute patch difference / 32). Returns 0. */
int ora_klt_track(const uint8_t* prev, const uint8_t* next, int w, int h, int step,
                  const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
                  int win, int max_level, int max_iter, double eps, double min_eig_thr,
                  int use_initial_flow);

/* The same three functions with the work organised as an optimised CPU library organises it (one derivative image per
 * pyramid level, border-extended images, row-streaming eigenvalue map; ora_opt.cpp): bit-identical outputs, several
 * times faster.  They are what the benchmark's CPU baseline times as `port_fast`. */
void ora_min_eigen_map_opt(const uint8_t* img, int w, int h, int step, float* eig);
int ora_good_features_opt(const uint8_t* img, int w, int h, int step,
                          int max_corners, double quality, double min_distance,
                          const float* seeds, int n_seed,
                          float* corners_xy, float* scores, int cap);
int ora_klt_track_opt(const uint8_t* prev, const uint8_t* next, int w, int h, int step,
                      const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
                      int win, int max_level, int max_iter, double eps, double min_eig_thr,
                      int use_initial_flow);

void ora_sincos_det(float x, float* s, float* c);

/* cv::undistortPoints, radtan model, normalised output; cam4 = fx fy cx cy, d5 = k1 k2 p1 p2 k3 */
void ora_undistort_points(const float* pts, float* out, long long n, const double* cam4, const double* d5);

/* FAST-9/16 with 3x3 non-max suppression; keypoints in raster order as (x, y, score)
 * int triples.  Returns the count, or -1 if cap is too small. */
int ora_fast9(const uint8_t* img, int w, int h, int step, int threshold, int nms,
              int* kps_xys, int cap);

/* FAST per grid cell, best `per-cell` by score, then greedy min-distance
 * (config/openvins.yaml:29-37: fast_threshold, grid_x, grid_y, min_px_dist, num_pts).
 * Output (x, y) float pairs + scores; returns count or -1. */
int ora_fast_grid(const uint8_t* img, int w, int h, int step, int threshold,
                  int grid_x, int grid_y, int num_features, int min_px_dist,
                  const float* seeds, int n_seed,
                  float* pts_xy, int* scores, int cap);

/* One ORB keypoint as the oracle and the library both report it. */
typedef struct ora_orb_kp {
    float x, y;        /* level-0 coordinates (level coords * scale) */
    float size;        /* 31 * scale */
    float angle;       /* degrees [0,360) */
    float response;    /* Harris response */
    int   octave;
    int   lx, ly;      /* integer coordinates inside the pyramid level */
} ora_orb_kp;

/* ORB::create(nfeatures, scale, nlevels, edge=31, first=0, WTA_K=2, HARRIS, patch=31,
 * fast_threshold)->detectAndCompute (orbslam3_adapter.cpp:76-79).  Keypoints come out in
 * canonical order (octave, ly, lx); desc is 32 bytes per keypoint. Returns count or -1. */
int ora_orb_extract(const uint8_t* img, int w, int h, int step,
                    int nfeatures, float scale_factor, int nlevels, int fast_threshold,
                    ora_orb_kp* kps, uint8_t* desc, int cap);
int ora_orb_level_sizes(int w, int h, float scale_factor, int nlevels, int* wh, float* scales);
int ora_orb_quotas(int nfeatures, float scale_factor, int nlevels, int* quota);

/* BFMatcher(NORM_HAMMING).knnMatch(k=2) on 256-bit descriptors.
 * out_idx/out_dist hold 2 entries per query (idx -1 / dist 0x7fffffff if absent);
 * train indices are reported as base_index + row. */
void ora_match_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt,
                    int base_index, int* out_idx, int* out_dist);
/* Lowe ratio test on knn2 output: good[i] = 1 iff 2 neighbours exist and
 * (float)d0 < ratio * (float)d1. */
void ora_ratio_test(const int* idx, const int* dist, int nq, float ratio, uint8_t* good);

/* cv::cornerSubPix (`subpixel: true`, config/basalt.yaml:102); pts [n][2] refined in place; 0 on success.
 * ora_subpix_mask fills the (2*win+1)^2 window weights. */
int ora_subpix_mask(int win, int zero_zone, float* mask);
int ora_corner_subpix(const uint8_t* img, int w, int h, int step, float* pts, int n, int win, int zero_zone,
                      int max_iter, double eps);

/* cv::findFundamentalMat(FM_RANSAC) as an outlier filter (`F_threshold`, config/vins_mono.yaml:101); see
 * ora_geom.cpp for the sampling and solver conventions.  Returns the inlier count. */
int ora_fundamental_ransac(const float* p1, const float* p2, uint8_t* status, int n, double threshold,
                           double confidence, int max_iters, uint32_t seed, double* F_out);

#ifdef __cplusplus
}
#endif
#endif
