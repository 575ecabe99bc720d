/*
 * vfe.h -- C ABI of the MI355X (gfx950) visual front-end library, libvfe.so.
 *
 * The reference (kcenon/vi_slam) has no FFI for this path: its adapters call OpenCV
 * in-process behind `vi_slam::ISLAMFramework::processImage(const cv::Mat&, int64_t)`
 * (include/slam/i_slam_framework.hpp:63).  This header is the boundary a maintainer binds
 * instead of those OpenCV calls; every entry point cites the call it replaces.
 *
 * Conventions
 *  - plain C: pointers and sizes only, no C++/torch types.
 *  - every function returns a vfe_status (0 = OK, negative = error), never throws, never
 *    aborts; vfe_last_error() gives a thread-local message for the last failure.
 *  - `vfe_dev_*` functions take DEVICE pointers (hipMalloc / torch.cuda memory) and a
 *    `stream` (hipStream_t passed as void*, NULL = default stream); they only enqueue work.
 *    They are batched: one launch processes `batch` independent frames.
 *  - `vfe_ctx` functions take HOST pointers shaped like the fields read from a cv::Mat
 *    (data, rows, cols, step, channels) and block until results are in the caller's
 *    buffers.  A context is bound to one GPU and is not thread-safe (the reference
 *    serialises calls under SLAMEngine::frameworkMutex_, src/slam/slam_engine.cpp:95).
 *  - images are 8-bit, row-major.  Points are (x, y) float pairs.
 *  - there is no CPU fallback: without a usable GPU every compute entry point fails with
 *    VFE_E_NODEVICE / VFE_E_HIP.
 */
#ifndef VFE_H
#define VFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFE_MAX_LEVELS 16

typedef enum vfe_status {
    VFE_OK = 0,
    VFE_E_INVALID = -1,   /* bad argument */
    VFE_E_HIP = -2,       /* HIP runtime error */
    VFE_E_NOMEM = -3,     /* allocation failed / workspace too small */
    VFE_E_CAPACITY = -4,  /* an output or internal buffer overflowed */
    VFE_E_NODEVICE = -5   /* no usable gfx950 device */
} vfe_status;

const char* vfe_status_string(int status);
const char* vfe_last_error(void);
const char* vfe_version(void);
int vfe_device_count(int* count);

/* ------------------------------------------------------------------------------------------
 * Batched device-resident images: `batch` frames of width x height bytes; frame b starts at
 * data + b*frame_stride, row y at + y*pitch.
 *
 * `border` > 0 promises that `border` pixels on every side of every frame are allocated (rows
 * -border .. height+border-1, columns -border .. width+border-1 relative to `data`) and hold the
 * BORDER_REFLECT_101 extension of the frame, the layout cv::buildOpticalFlowPyramid gives its
 * levels.  vfe_dev_pyr_down / vfe_dev_build_pyramid fill the border of every level they write;
 * for images the caller fills (level 0) leave it 0 unless the caller has written the extension.
 * The tracker reads windows that overhang the frame straight from the border when it is at
 * least VFE_LK_BORDER wide, and falls back to per-pixel reflection otherwise: results are
 * identical, padded levels are faster.
 * ---------------------------------------------------------------------------------------- */
#define VFE_LK_BORDER 32
typedef struct vfe_dimg {
    uint8_t* data;
    int width, height;
    int pitch;            /* bytes per row */
    int batch;
    size_t frame_stride;  /* bytes between consecutive frames */
    int border;           /* reflect-101 padding present around every frame, in pixels (0 = none) */
} vfe_dimg;

/* One ORB keypoint (mirrors the fields of cv::KeyPoint that ORB fills, plus the integer
 * position inside its pyramid level). */
typedef struct vfe_orb_kp {
    float x, y;      /* level-0 coordinates */
    float size;      /* 31 * scale(level) */
    float angle;     /* degrees, [0, 360) */
    float response;  /* Harris response */
    int octave;
    int lx, ly;      /* integer coordinates in level `octave` */
} vfe_orb_kp;

/* ---- colour / intensity pre-processing ------------------------------------------------- */

/* cv::cvtColor(image, gray, COLOR_BGR2GRAY) on 8-bit BGR
 * (src/slam/adapters/openvins_adapter.cpp:72, basalt_adapter.cpp:44, orbslam3_adapter.cpp:71).
 * bgr has 3*width bytes per row. */
int vfe_dev_bgr2gray(const vfe_dimg* bgr, const vfe_dimg* gray, void* stream);

/* cv::equalizeHist on 8-bit gray frames (`equalize: 1`, reference config/vins_mono.yaml:104; the
 * reference's VINS placeholder src/slam/adapters/vins_mono_adapter.cpp:19-23 never reads it).  src and
 * dst may be the same image.  workspace: vfe_equalize_workspace(batch) bytes, 256-byte aligned. */
size_t vfe_equalize_workspace(int batch);
int vfe_dev_equalize_hist(const vfe_dimg* src, const vfe_dimg* dst, void* workspace, size_t workspace_bytes, void* stream);

/* cv::createCLAHE(clip_limit, Size(tiles_x, tiles_y))->apply on 8-bit gray frames (`histogram_method: "CLAHE"`,
 * reference docs/reference/15_openvins_smartphone_config.md:111; OpenVINS uses clip 10.0 on 8 x 8 tiles).
 * Sizes that are not multiples of the tile grid are extended by reflection for the histograms, as
 * OpenCV does.  dst may be the same image.  1..64 tiles per dimension.
 * workspace: vfe_clahe_workspace(batch, tiles_x, tiles_y) bytes, 256-byte aligned. */
size_t vfe_clahe_workspace(int batch, int tiles_x, int tiles_y);
int vfe_dev_clahe(const vfe_dimg* src, const vfe_dimg* dst, double clip_limit, int tiles_x, int tiles_y,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- pyramids -------------------------------------------------------------------------- */

/* cv::pyrDown: dst must be ((w+1)/2) x ((h+1)/2), same batch.  The stage the reference's
 * configuration names as `pyramid_levels` / `num_klt_levels` (config/basalt.yaml:93,
 * config/openvins.yaml:39). */
int vfe_dev_pyr_down(const vfe_dimg* src, const vfe_dimg* dst, void* stream);

/* cv::buildOpticalFlowPyramid: fills levels[1..nlevels-1] from levels[0]. */
int vfe_dev_build_pyramid(const vfe_dimg* levels, int nlevels, void* stream);

/* Number of the last level buildOpticalFlowPyramid(win, max_level) produces for w x h. */
int vfe_lk_max_level(int width, int height, int win, int max_level);

/* ---- Shi-Tomasi corners ---------------------------------------------------------------- */

/* Bytes of device workspace vfe_dev_good_features needs. */
size_t vfe_good_features_workspace(int width, int height, int batch, int max_corners, int seed_cap);

/* cv::goodFeaturesToTrack(gray, corners, max_corners, quality, min_distance) with OpenCV's
 * defaults (no mask, blockSize 3, Shi-Tomasi score), per frame of the batch
 * (src/slam/adapters/openvins_adapter.cpp:79-80, src/slam/adapters/basalt_adapter.cpp:50-53).
 *  seeds_xy [batch][seed_cap][2], n_seeds [batch] (device; NULL = none): points already being
 *    tracked; they occupy the min-distance grid first, so new corners keep away from them,
 *    and frame b returns at most max_corners - n_seeds[b] new corners (the VINS-style
 *    "refill" the reference's max_cnt/min_dist keys imply, config/vins_mono.yaml:92,95).
 *    Seeds themselves are not returned.
 *  corners_xy [batch][max_corners][2], scores [batch][max_corners] (may be NULL),
 *  counts [batch]: corners found per frame, in OpenCV's order (score descending).
 *  max_corners must be > 0. */
int vfe_dev_good_features(const vfe_dimg* img, int max_corners, double quality, double min_distance,
                          const float* seeds_xy, const int* n_seeds, int seed_cap,
                          float* corners_xy, float* scores, int* counts,
                          void* workspace, size_t workspace_bytes, void* stream);

/* cv::cornerMinEigenVal(gray, eig, 3, 3): dense float map [batch][height][width] (debug/tests). */
int vfe_dev_min_eigen_map(const vfe_dimg* img, float* eig, void* stream);

/* ---- pyramidal Lucas-Kanade ------------------------------------------------------------ */

/* cv::calcOpticalFlowPyrLK on pyramids from vfe_dev_build_pyramid -- the tracker selected by
 * `use_klt` (config/openvins.yaml:26) / Basalt's optical flow (config/basalt.yaml:85-108);
 * the reference names it but never calls it.
 *  prev_levels/next_levels [nlevels]; prev_pts/next_pts [batch][pts_cap][2];
 *  n_pts [batch] (device; NULL = pts_cap points in every frame);
 *  status [batch][pts_cap] (1 = tracked), err [batch][pts_cap] (may be NULL).
 *  use_initial_flow != 0: next_pts holds the initial guess (OPTFLOW_USE_INITIAL_FLOW). */
int vfe_dev_klt_track(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                      const float* prev_pts, const int* n_pts, int pts_cap,
                      float* next_pts, uint8_t* status, float* err,
                      int win, int max_iter, double eps, double min_eig_thr, int use_initial_flow,
                      void* stream);

/* The same call (same cv::calcOpticalFlowPyrLK, same results at every point's own index) with a caller workspace of
 * vfe_klt_track_workspace(pts_cap, batch) bytes, 256-byte aligned: the points of a frame are then tracked in an order
 * of the library's choosing -- rim points first, then image tile by image tile -- instead of the caller's.  Neighbouring
 * points share cache lines and waves, and only the waves of rim points pay for windows that overhang a level:
 * 7-12 % faster on 256 x 1000 points of 1920x1080 frames.  Applies to win = 21 and pts_cap <= 65536 (otherwise
 * the workspace is ignored). */
size_t vfe_klt_track_workspace(int pts_cap, int batch);
int vfe_dev_klt_track_ws(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                         const float* prev_pts, const int* n_pts, int pts_cap,
                         float* next_pts, uint8_t* status, float* err,
                         int win, int max_iter, double eps, double min_eig_thr, int use_initial_flow,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Forward-backward check on top of vfe_dev_klt_track (`optical_flow_max_recovered_dist2`, reference
 * docs/reference/17_basalt_smartphone_config.md:257; the reverse check of VINS-style front-ends): after
 * the forward track, every point is tracked back next -> prev with its original position as the initial
 * guess; status stays 1 only if both tracks succeeded and the squared distance between the recovered and
 * the original position is <= max_recovered_dist2.  back_pts [batch][cap][2] (recovered positions) and
 * back_status [batch][cap] are outputs/scratch. */
int vfe_dev_klt_track_bidir(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                            const float* prev_pts, const int* n_pts, int pts_cap,
                            float* next_pts, uint8_t* status, float* err, float* back_pts, uint8_t* back_status,
                            int win, int max_iter, double eps, double min_eig_thr, double max_recovered_dist2,
                            void* stream);

/* cv::cornerSubPix(img, pts, Size(win, win), Size(zero_zone, zero_zone), {COUNT+EPS, max_iter, eps}) on every
 * frame's points, in place (`subpixel: true`, reference config/basalt.yaml:102).  win = half window (1..10),
 * zero_zone < 0 = none.  pts [batch][pts_cap][2]; n_pts [batch] (device; NULL = pts_cap points per frame). */
int vfe_dev_corner_subpix(const vfe_dimg* img, float* pts, const int* n_pts, int pts_cap, int win, int zero_zone,
                          int max_iter, double eps, void* stream);

/* Outlier rejection with the epipolar constraint, cv::findFundamentalMat(pts1, pts2, FM_RANSAC, threshold,
 * confidence, mask)-shaped (`F_threshold: 1.0`, reference config/vins_mono.yaml:101): 7-point minimal solver,
 * error = larger squared point-to-epipolar-line distance, adaptive iteration count (at most max_iters <= 1024).
 * Deterministic for a given seed and independent of scheduling.
 *  pts1/pts2 [batch][pts_cap][2] (pts_cap <= 2048); n_pts [batch] (device; NULL = pts_cap per frame);
 *  status [batch][pts_cap] in/out: non-zero entries take part, on return 1 = inlier, 0 = rejected;
 *  F_out [batch][9] row-major (zeros when fewer than 8 points take part -- nothing is rejected then -- or when
 *  no model was found -- everything is rejected); inliers [batch].
 *  workspace: vfe_fundamental_workspace(batch) bytes, 256-byte aligned. */
size_t vfe_fundamental_workspace(int batch);
int vfe_dev_fundamental_ransac(const float* pts1, const float* pts2, const int* n_pts, int pts_cap, int batch,
                               uint8_t* status, double threshold, double confidence, int max_iters, unsigned seed,
                               double* F_out, int* inliers, void* workspace, size_t workspace_bytes, void* stream);

/* cv::undistortPoints (no R, no P): pixel positions -> normalised, undistorted coordinates for the
 * radial-tangential model the configs carry (`camera.k1, k2, p1, p2`, reference
 * config/vins_mono.yaml:32-35).  camera4 = {fx, fy, cx, cy}, dist5 = {k1, k2, p1, p2, k3} (host
 * arrays); pts/out [n][2] device, may alias. */
int vfe_dev_undistort_points(const float* pts, float* out, long long n, const double* camera4, const double* dist5,
                             void* stream);

/* ---- batched front-end step ------------------------------------------------------------- */

/* After tracking: keep the points with status != 0 whose rounded position lies inside
 * [border, width-border) x [border, height-border), preserving order.
 * pts/status [batch][cap], n_pts [batch] (NULL = cap); out_pts [batch][cap][2], out_n [batch]. */
int vfe_dev_compact_tracks(const float* pts, const uint8_t* status, const int* n_pts, int cap, int batch,
                           int width, int height, int border, float* out_pts, int* out_n, void* stream);

/* pts[b][n_pts[b] + i] = extra[b][i] for i < n_extra[b]; n_pts[b] += n_extra[b] (clamped to cap). */
int vfe_dev_append_points(float* pts, int* n_pts, int cap, int batch,
                          const float* extra, const int* n_extra, int extra_cap, void* stream);

size_t vfe_frontend_workspace(int width, int height, int batch, int max_corners);

/* One steady-state front-end step for `batch` independent streams, everything on the GPU:
 *   1. pyramid of the new frame (next_levels[1..] from next_levels[0]),
 *   2. KLT from prev_levels to next_levels for the pts/n_pts of each stream,
 *   3. drop lost / out-of-border tracks,
 *   4. refill with Shi-Tomasi corners of the new frame up to max_corners, min_distance away
 *      from surviving tracks.
 * pts [batch][max_corners][2] and n_pts [batch] are updated in place to the new point set
 * (tracked points first, in order, then new corners by descending score).
 * n_tracked [batch] (may be NULL) receives the number of surviving tracks.
 * This is the batched form of what the adapters do per processImage call (reference
 * include/slam/i_slam_framework.hpp:63, src/slam/adapters/vins_mono_adapter.cpp:19-23, 109). */
int vfe_dev_frontend_step(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                          float* pts, int* n_pts, int* n_tracked, int max_corners,
                          double quality, double min_distance,
                          int win, int max_iter, double eps, double min_eig_thr,
                          void* workspace, size_t workspace_bytes, void* stream);

/* vfe_dev_frontend_step runs the detector's eigenvalue pass on a helper stream of the library (one per host thread and
 * device, forked from and joined back into the caller's stream) beside pyramid + tracker: mode 1 (default).  mode 0
 * keeps every kernel on the caller's stream, in order -- what a profiler wants to see.  The environment variable
 * VFE_OVERLAP=0/1 sets the initial mode.  Process-wide (a context overrides it for its own calls with vfe_ctx_set_overlap).
 * The helper stream and its events are created by the first overlapped call of a host thread on a device; a call whose
 * stream is being captured into a graph before that has happened runs on the caller's stream alone, so warm up with one
 * un-captured step if the captured graph should contain the fork.  No counterpart in the reference, whose adapters make one
 * blocking OpenCV call per frame (src/slam/adapters/openvins_adapter.cpp:78-80). */
int vfe_set_overlap(int mode);

/* The same step with the epipolar outlier filter of VINS-style front-ends (`F_threshold`, reference
 * config/vins_mono.yaml:101) between tracking and refill: tracks that left the image are dropped, the rest is
 * filtered by vfe_dev_fundamental_ransac on (previous, tracked) positions -- lifted to a virtual pinhole camera of
 * focal length virtual_focal centred on the image when a calibration is given (what VINS-Mono's rejectWithF does with
 * FOCAL_LENGTH 460), on raw pixel coordinates otherwise -- and only inliers survive.  Frames with fewer than 8 tracks
 * are not filtered.  reject == NULL or f_threshold <= 0: exactly vfe_dev_frontend_step.  max_corners <= 2048. */
typedef struct vfe_reject_opts {
    double f_threshold;     /* pixels; <= 0 disables the filter */
    double confidence;      /* 0.99 in VINS-Mono */
    int max_iters;          /* 1..1024 */
    unsigned seed;
    const double* camera4;  /* fx, fy, cx, cy (host pointer) or NULL */
    const double* dist5;    /* k1, k2, p1, p2, k3 (host pointer), required with camera4 */
    double virtual_focal;   /* used with camera4 */
    void* workspace;        /* vfe_reject_workspace(batch, max_corners) bytes, 256-byte aligned, device memory */
    size_t workspace_bytes;
} vfe_reject_opts;
size_t vfe_reject_workspace(int batch, int max_corners);
int vfe_dev_frontend_step_reject(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                 float* pts, int* n_pts, int* n_tracked, int max_corners,
                                 double quality, double min_distance,
                                 int win, int max_iter, double eps, double min_eig_thr,
                                 void* workspace, size_t workspace_bytes, const vfe_reject_opts* reject, void* stream);

/* vfe_dev_frontend_step_reject stage by stage, for a pipeline that holds the next frame before the current step has
 * ended (a batch of recorded sequences, a server whose frames queue up):
 *   vfe_dev_build_pyramid        next_levels[1..] from next_levels[0]                    needs the new frame
 *   vfe_dev_frontend_candidates  the detector's eigenvalue pass and candidate lists      needs the new frame
 *   vfe_dev_frontend_track       tracker and optional epipolar filter                    needs both pyramids and the points
 *   vfe_dev_frontend_refill      drops lost tracks, selects new corners, appends them    needs the two stages before it
 * Each stage runs on the stream it is given and nowhere else, and leaves what the later ones read in `workspace`
 * (vfe_frontend_workspace bytes, one per frame in flight); the caller orders them with events.  With two workspaces and
 * three pyramids the first two stages of frame t + 1 run beside the last two of frame t: a chain of dependent launches
 * per step becomes two chains side by side, worth 6-10 % of a step between 4 and 32 streams per GPU (nothing above: there
 * the chip is busy either way).  Same arguments and the same results, bit for bit, as the one call.  No counterpart in the
 * reference, whose adapters make one blocking call per frame (src/slam/adapters/openvins_adapter.cpp:78-80). */
int vfe_dev_frontend_candidates(const vfe_dimg* next_levels, int nlevels, int max_corners, double quality, double min_distance,
                                void* workspace, size_t workspace_bytes, void* stream);
int vfe_dev_frontend_track(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                           const float* pts, const int* n_pts, int max_corners,
                           int win, int max_iter, double eps, double min_eig_thr,
                           void* workspace, size_t workspace_bytes, const vfe_reject_opts* reject, void* stream);
int vfe_dev_frontend_refill(const vfe_dimg* next_levels, float* pts, int* n_pts, int* n_tracked, int max_corners,
                            double quality, double min_distance, void* workspace, size_t workspace_bytes, void* stream);

/* The step with OpenVINS's own detector in place of Shi-Tomasi: tracks are carried by KLT (`use_klt: true`, reference
 * config/openvins.yaml:26) and the frame is refilled with FAST corners bucketed over a grid_x x grid_y grid (`fast_threshold`,
 * `grid_x`, `grid_y`, `min_px_dist`, `num_pts`: config/openvins.yaml:29-37; struct fields
 * include/slam/adapters/openvins_adapter.hpp:29-32) -- the keys the reference parses and then ignores in favour of one
 * goodFeaturesToTrack call per frame (src/slam/adapters/openvins_adapter.cpp:78-80).  Steps 1-3 as vfe_dev_frontend_step,
 * then vfe_dev_fast_grid with the surviving tracks as seeds and num_features = max_corners; the new points are appended
 * in the order the grid offers them (cell by cell, best score first) until max_corners is reached.
 * workspace: vfe_frontend_fast_workspace(...) bytes, 256-byte aligned. */
size_t vfe_frontend_fast_workspace(int width, int height, int batch, int max_corners, int grid_x, int grid_y, int min_px_dist);
int vfe_dev_frontend_step_fast(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                               float* pts, int* n_pts, int* n_tracked, int max_corners,
                               int fast_threshold, int grid_x, int grid_y, int min_px_dist,
                               int win, int max_iter, double eps, double min_eig_thr,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ---- kernel timing (HIP events recorded around every kernel the library launches) ------- */
enum { VFE_K_PYR_DOWN = 0, VFE_K_GFTT_EIG, VFE_K_GFTT_SELECT, VFE_K_KLT, VFE_K_COMPACT, VFE_K_APPEND,
       VFE_K_FAST, VFE_K_ORB_RESIZE, VFE_K_ORB_SELECT, VFE_K_ORB_DESC, VFE_K_MATCH,
       VFE_K_MISC, VFE_K_COUNT };
/* enable != 0: from now on every launch is bracketed by events on its stream. */
int vfe_prof_enable(int enable);
/* Synchronise the recorded events, add their durations to the totals, and copy the totals:
 * ms[VFE_K_COUNT] = summed milliseconds, launches[VFE_K_COUNT] = launch counts.  reset != 0
 * clears the totals afterwards. */
int vfe_prof_read(double* ms, long long* launches, int reset);
const char* vfe_prof_kernel_name(int id);
/* vfe_set_overlap and vfe_prof_enable / vfe_prof_read are process-wide: they govern the vfe_dev_* entry points and every
 * context that has not made a choice of its own with vfe_ctx_set_overlap / vfe_ctx_prof_enable (below). */

/* ---- FAST ------------------------------------------------------------------------------ */

size_t vfe_fast_workspace(int width, int height, int batch);

/* cv::FAST(img, kps, threshold, nonmaxSuppression=true, TYPE_9_16) per frame; keypoints as
 * (x, y, score) int triples in raster order, kps [batch][cap][3], counts [batch]
 * (`fast_threshold`, config/openvins.yaml:29; run on every ORB level by
 * src/slam/adapters/orbslam3_adapter.cpp:79). */
int vfe_dev_fast9(const vfe_dimg* img, int threshold, int* kps, int cap, int* counts,
                  void* workspace, size_t workspace_bytes, void* stream);

/* FAST per grid cell + best-per-cell + occupancy min distance (config/openvins.yaml:29-37:
 * fast_threshold, grid_x, grid_y, min_px_dist, num_pts; struct fields
 * include/slam/adapters/openvins_adapter.hpp:29-32).  FAST runs on each of the grid_x*grid_y
 * cells as an image of its own; the (num_features - n_seeds)/(cells) + 1 best of each cell by
 * score (ties: raster order) are offered cell by cell to an occupancy grid of min_px_dist pixels
 * that the seeds (existing tracks) have marked first.  pts_xy [batch][cap][2],
 * scores [batch][cap], counts [batch]; cap >= num_features + grid_x*grid_y. */
size_t vfe_fast_grid_workspace(int width, int height, int batch, int grid_x, int grid_y, int min_px_dist);
int vfe_dev_fast_grid(const vfe_dimg* img, int threshold, int grid_x, int grid_y, int num_features,
                      int min_px_dist, const float* seeds_xy, const int* n_seeds, int seed_cap,
                      float* pts_xy, int* scores, int cap, int* counts,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- ORB ------------------------------------------------------------------------------- */

size_t vfe_orb_workspace(int width, int height, int batch, float scale_factor, int nlevels);
int vfe_orb_level_sizes(int width, int height, float scale_factor, int nlevels, int* wh, float* scales);
int vfe_orb_quotas(int nfeatures, float scale_factor, int nlevels, int* quota);

/* cv::ORB::create(nfeatures, scale_factor, nlevels, 31, 0, 2, HARRIS_SCORE, 31, fast_threshold)
 *   ->detectAndCompute(gray, noArray(), keypoints, descriptors)
 * (src/slam/adapters/orbslam3_adapter.cpp:76-79).  Keypoints of a frame come out in
 * canonical order (octave, ly, lx); OpenCV's own order is unspecified (nth_element).
 * kps [batch][cap], desc [batch][cap][32], counts [batch]. */
int vfe_dev_orb_extract(const vfe_dimg* img, int nfeatures, float scale_factor, int nlevels,
                        int fast_threshold, vfe_orb_kp* kps, uint8_t* desc, int cap, int* counts,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- brute-force Hamming matching -------------------------------------------------------- */

/* cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, 2) on 256-bit descriptors (the matcher
 * behind `knn_ratio`, docs/reference/15_openvins_smartphone_config.md:105).
 *  out_idx/out_dist [nq][2]; train indices are reported as base_index + row, so a shard of a
 *  larger map reports global indices; absent neighbours: idx -1, dist 0x7fffffff. */
size_t vfe_match_workspace(int nq, long long nt);
int vfe_dev_match_knn2(const uint8_t* query, int nq, const uint8_t* train, long long nt,
                       int base_index, int* out_idx, int* out_dist,
                       void* workspace, size_t workspace_bytes, void* stream);

/* vfe_dev_match_knn2 for callers that cannot allocate device memory or bring a workspace: code under HIP graph
 * capture, tight serving loops (one relocalisation query set after the other against a resident map), harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference.  Same arguments, results and error behaviour as
 * vfe_dev_match_knn2 minus the workspace: the per-chunk top-2 lists (for train sets of 500 000 rows and more also the
 * lists of the sampled rows and one floor per query) go to a VFE_MATCH_NOALLOC_BYTES array that the library's code
 * object reserves on every GPU it runs on (held until the process ends).  The call enqueues its kernels (search and
 * merge; sample search and merge before them for large train sets) on `stream` and calls no allocating routine; a device's copy of the array is looked up once, by vfe_create
 * or by the first call on that device (make that call before a graph capture).  Sizes whose partial lists do not fit
 * (vfe_match_workspace(nq, nt) > VFE_MATCH_NOALLOC_BYTES: from about 5000 queries against a 10 M-row train set, which
 * is searched in a few hundred chunks; ask vfe_match_workspace)
 * are rejected with VFE_E_CAPACITY.  All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (different streams or threads) race; serialise them or use vfe_dev_match_knn2 with separate workspaces. */
#define VFE_MATCH_NOALLOC_BYTES (16u << 20)
int vfe_dev_match_knn2_noalloc(const uint8_t* query, int nq, const uint8_t* train, long long nt,
                               int base_index, int* out_idx, int* out_dist, void* stream);

/* The same for a batch of independent (query, train) pairs, e.g. the descriptors of consecutive frames
 * of `batch` streams as vfe_dev_orb_extract leaves them: set b has nq[b] (nt[b]) descriptors at
 * query + b*q_cap*32 (train + b*t_cap*32); nq and nt are DEVICE arrays, so extraction and matching can
 * be queued without a host round trip.  out_idx/out_dist [batch][q_cap][2], train indices 0-based per
 * set; rows >= nq[b] get the 'absent' record.  good (optional, [batch][q_cap]) receives Lowe's ratio
 * test d0 < ratio * d1 (`knn_ratio`).  No workspace. */
int vfe_dev_match_knn2_batch(const uint8_t* query, const int* nq, int q_cap,
                             const uint8_t* train, const int* nt, int t_cap, int batch,
                             int* out_idx, int* out_dist, float ratio, uint8_t* good, void* stream);

/* Merge per-shard top-2 lists (as all-gathered from `nshards` ranks, [nshards][nq][2]) into
 * the global top-2 by (dist, idx) -- the order a single scan over the whole map produces. */
int vfe_dev_merge_knn2(const int* shard_idx, const int* shard_dist, int nshards, int nq,
                       int* out_idx, int* out_dist, void* stream);

/* Lowe ratio test: good[q] = 1 iff both neighbours exist and d0 < ratio * d1 (float). */
int vfe_dev_ratio_test(const int* idx, const int* dist, int nq, float ratio, uint8_t* good, void* stream);

/* ------------------------------------------------------------------------------------------
 * Host-pointer context API (what the framework adapters call from processImage).
 * ---------------------------------------------------------------------------------------- */
typedef struct vfe_ctx vfe_ctx;

typedef struct vfe_config {
    int device;           /* HIP device ordinal */
    int max_width;        /* largest frame the context will see */
    int max_height;
    int max_features;     /* capacity of every point list */
    int pyramid_levels;   /* LK pyramid levels including level 0 (maxLevel + 1) */
    int win_size;         /* LK window (21) */
    int orb_levels;       /* 0 = ORB not used */
} vfe_config;

typedef struct vfe_frame_stats {
    int n_tracked;        /* points carried over from the previous frame by KLT */
    int n_detected;       /* new corners added this frame */
    int n_total;          /* points alive after this frame */
} vfe_frame_stats;

int vfe_create(const vfe_config* cfg, vfe_ctx** out);
void vfe_destroy(vfe_ctx* ctx);
/* forget the previous frame and all tracks (ISLAMFramework::reset, i_slam_framework.hpp:103) */
int vfe_reset(vfe_ctx* ctx);
/* Per-context forms of vfe_set_overlap and vfe_prof_enable / vfe_prof_read, so that two engines in one process do not
 * switch each other's settings (an engine is one context: reference src/slam/slam_engine.cpp:94-119 serialises one
 * framework under one mutex).  mode 0 / 1: this context's calls run with the setting off / on whatever the process-wide
 * switch says; -1 (the state after vfe_create): follow the process-wide switch.  Launches timed for a context are
 * summed apart from everybody else's and read with vfe_ctx_prof_read (same arrays as vfe_prof_read). */
int vfe_ctx_set_overlap(vfe_ctx* ctx, int mode);
int vfe_ctx_prof_enable(vfe_ctx* ctx, int mode);
int vfe_ctx_prof_read(vfe_ctx* ctx, double* ms, long long* launches, int reset);
/* Every frame handed to the calls below is equalised on the GPU after upload / BGR conversion and before
 * anything else reads it.  mode VFE_EQ_HIST: cv::equalizeHist (`equalize: 1` of reference
 * config/vins_mono.yaml:104); VFE_EQ_CLAHE: CLAHE with clip 10.0 on 8 x 8 tiles (`histogram_method: "CLAHE"`,
 * reference docs/reference/15_openvins_smartphone_config.md:111); VFE_EQ_NONE (default): off. */
#define VFE_EQ_NONE 0
#define VFE_EQ_HIST 1
#define VFE_EQ_CLAHE 2
int vfe_set_equalize(vfe_ctx* ctx, int mode);

/* Detector behind vfe_process_frame: VFE_DET_SHI_TOMASI (default; quality / min_distance arguments of the call) or
 * VFE_DET_FAST_GRID (OpenVINS's: FAST with `fast_threshold` on a grid_x x grid_y grid, occupancy cells of the call's
 * min_distance rounded down to whole pixels; the call's quality argument is unused).  Switching forgets the stream. */
#define VFE_DET_SHI_TOMASI 0
#define VFE_DET_FAST_GRID 1
int vfe_set_detector(vfe_ctx* ctx, int mode, int fast_threshold, int grid_x, int grid_y);

/* vfe_process_frame filters tracks with the epipolar constraint before refilling (see vfe_reject_opts):
 * threshold <= 0 turns it off (default).  camera4 / dist5 may be NULL (raw pixel coordinates). */
int vfe_set_f_rejection(vfe_ctx* ctx, double threshold, const double* camera4, const double* dist5, double virtual_focal);

/* goodFeaturesToTrack on a host image (channels 1 or 3; 3 = BGR, converted on the GPU): replaces the cvtColor +
 * goodFeaturesToTrack pair of reference src/slam/adapters/openvins_adapter.cpp:72-80 and basalt_adapter.cpp:44-53. */
int vfe_detect_shi_tomasi(vfe_ctx* ctx, const uint8_t* data, int rows, int cols, size_t step, int channels,
                          int max_corners, double quality, double min_distance,
                          float* corners_xy, int* n_out);

/* calcOpticalFlowPyrLK between two host images (`use_klt: true`, reference config/openvins.yaml:26; the optical-flow
 * keys of config/basalt.yaml:85-108). */
int vfe_klt_track(vfe_ctx* ctx, const uint8_t* prev, const uint8_t* next, int rows, int cols, size_t step,
                  const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
                  int max_level, int max_iter, double eps, double min_eig_thr);

/* ORB::detectAndCompute on a host image: replaces reference src/slam/adapters/orbslam3_adapter.cpp:71-79. */
int vfe_orb_extract(vfe_ctx* ctx, const uint8_t* data, int rows, int cols, size_t step, int channels,
                    int nfeatures, float scale_factor, int nlevels, int fast_threshold,
                    vfe_orb_kp* kps, uint8_t* desc, int cap, int* n_out);

/* knnMatch(k=2) + ratio test on host descriptors (`knn_ratio: 0.70`, reference
 * docs/reference/15_openvins_smartphone_config.md:105). */
int vfe_match_knn2(vfe_ctx* ctx, const uint8_t* query, int nq, const uint8_t* train, int nt,
                   float ratio, int* idx, int* dist, uint8_t* good);

/* One front-end step of a live stream: pyramid of the new frame, KLT from the previous
 * frame's points, refill with Shi-Tomasi corners up to max_corners.  The first frame after
 * create/reset only detects.  This is the body of ISLAMFramework::processImage (reference
 * include/slam/i_slam_framework.hpp:63) for a tracking front-end, which the reference leaves empty in
 * src/slam/adapters/vins_mono_adapter.cpp:19-23. */
int vfe_process_frame(vfe_ctx* ctx, const uint8_t* data, int rows, int cols, size_t step, int channels,
                      int64_t timestamp_ns, int max_corners, double quality, double min_distance,
                      vfe_frame_stats* stats);
/* current points of the stream (after vfe_process_frame) */
int vfe_get_points(vfe_ctx* ctx, float* pts_xy, int cap, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* VFE_H */
