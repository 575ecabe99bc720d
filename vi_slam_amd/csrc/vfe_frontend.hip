/* Batched steady-state front-end step: pyramid -> KLT -> drop lost tracks -> Shi-Tomasi refill.
 * The batched form of what a framework adapter does in one processImage call
 * (reference call path: src/slam/slam_engine.cpp:94-119 -> adapters' feedImage). */
#include "vfe_common.h"
#include "vfe_gftt.h"
#include "vfe_klt.h"
#include <cstdlib>

/* stable compaction of surviving tracks, one workgroup per stream */
__global__ __launch_bounds__(256) void k_compact(const float* __restrict__ pts, const uint8_t* __restrict__ status,
                                                 const int* __restrict__ n_pts, int cap, int W, int H, int border,
                                                 float* __restrict__ out, int* __restrict__ out_n, int* __restrict__ out_n2) {
    __shared__ int s_w[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = n_pts ? max(0, min(n_pts[b], cap)) : cap;
    int base = 0;
    for (int start = 0; start < n; start += 256) {
        const int i = start + tid;
        bool keep = false;
        float x = 0.f, y = 0.f;
        if (i < n && status[(size_t)b * cap + i]) {
            x = pts[((size_t)b * cap + i) * 2]; y = pts[((size_t)b * cap + i) * 2 + 1];
            const int ix = __float2int_rn(x), iy = __float2int_rn(y);
            keep = ix >= border && ix < W - border && iy >= border && iy < H - border;
        }
        const unsigned long long m = __ballot(keep);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wv] = __popcll(m);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { if (k < wv) off += s_w[k]; tot += s_w[k]; }
        /* in-place use (out == pts) is safe: element base+off+pre <= i and each chunk is read before it is written */
        __syncthreads();
        if (keep) {
            out[((size_t)b * cap + base + off + pre) * 2] = x;
            out[((size_t)b * cap + base + off + pre) * 2 + 1] = y;
        }
        base += tot;
    }
    if (tid == 0) { out_n[b] = base; if (out_n2) out_n2[b] = base; }
}

__global__ __launch_bounds__(256) void k_append(float* __restrict__ pts, int* __restrict__ n_pts, int cap,
                                                const float* __restrict__ extra, const int* __restrict__ n_extra,
                                                int extra_cap) {
    const int b = blockIdx.x;
    const int n0 = max(0, min(n_pts[b], cap));
    const int ne = max(0, min(min(n_extra[b], extra_cap), cap - n0));
    for (int i = threadIdx.x; i < ne; i += 256) {
        pts[((size_t)b * cap + n0 + i) * 2] = extra[((size_t)b * extra_cap + i) * 2];
        pts[((size_t)b * cap + n0 + i) * 2 + 1] = extra[((size_t)b * extra_cap + i) * 2 + 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) n_pts[b] = n0 + ne;
}

extern "C" int vfe_dev_compact_tracks(const float* pts, const uint8_t* status, const int* n_pts, int cap, int batch,
                                      int width, int height, int border, float* out_pts, int* out_n, void* stream) {
    VFE_REQUIRE(pts && status && out_pts && out_n, "NULL buffer");
    VFE_REQUIRE(cap > 0 && batch > 0 && width > 0 && height > 0 && border >= 0, "bad sizes");
    VFE_REQUIRE(out_pts != pts, "compaction is out of place");
    ProfScope prof(VFE_K_COMPACT, (hipStream_t)stream);
    hipLaunchKernelGGL(k_compact, dim3(batch), dim3(256), 0, (hipStream_t)stream, pts, status, n_pts, cap, width, height,
                       border, out_pts, out_n, (int*)nullptr);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_append_points(float* pts, int* n_pts, int cap, int batch,
                                     const float* extra, const int* n_extra, int extra_cap, void* stream) {
    VFE_REQUIRE(pts && n_pts && extra && n_extra, "NULL buffer");
    VFE_REQUIRE(cap > 0 && batch > 0 && extra_cap > 0, "bad sizes");
    ProfScope prof(VFE_K_APPEND, (hipStream_t)stream);
    hipLaunchKernelGGL(k_append, dim3(batch), dim3(256), 0, (hipStream_t)stream, pts, n_pts, cap, extra, n_extra, extra_cap);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

/* tracks that left the image (same rule as the compaction: rounded position inside [border, size - border)) */
__global__ __launch_bounds__(256) void k_border_status(const float* __restrict__ pts, const int* __restrict__ n_pts, int cap, int W, int H,
                                                       int border, uint8_t* __restrict__ status) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_pts ? max(0, min(n_pts[b], cap)) : cap;
    if (i >= n) return;
    const size_t o = (size_t)b * cap + i;
    if (!status[o]) return;
    const int ix = __float2int_rn(pts[2 * o]), iy = __float2int_rn(pts[2 * o + 1]);
    if (!(ix >= border && ix < W - border && iy >= border && iy < H - border)) status[o] = 0;
}

/* normalised coordinates -> pixels of a virtual pinhole camera, in place */
__global__ __launch_bounds__(256) void k_virtual_pinhole(float* __restrict__ pts, long long n, double f, double cx, double cy) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    pts[2 * i] = (float)(f * (double)pts[2 * i] + cx);
    pts[2 * i + 1] = (float)(f * (double)pts[2 * i + 1] + cy);
}

struct RjWs { size_t u1_off, u2_off, F_off, inl_off, models_off, models_bytes, total; };
static RjWs rj_layout(int batch, int cap) {
    RjWs L; size_t o = 0;
    L.u1_off = o; o += align_up((size_t)batch * cap * 2 * sizeof(float), 256);
    L.u2_off = o; o += align_up((size_t)batch * cap * 2 * sizeof(float), 256);
    L.F_off = o; o += align_up((size_t)batch * 9 * sizeof(double), 256);
    L.inl_off = o; o += align_up((size_t)batch * sizeof(int), 256);
    L.models_bytes = vfe_fundamental_workspace(batch);
    L.models_off = o; o += align_up(L.models_bytes, 256);
    L.total = o;
    return L;
}
extern "C" size_t vfe_reject_workspace(int batch, int max_corners) {
    if (batch <= 0 || max_corners <= 0 || max_corners > 2048) return 0;
    return rj_layout(batch, max_corners).total;
}

static int fe_reject(const vfe_reject_opts* rj, const float* prev_pts, const float* next_pts, const int* n_pts, int cap, int batch,
                     int w, int h, uint8_t* status, hipStream_t st) {
    VFE_REQUIRE(cap <= 2048, "the epipolar filter takes at most 2048 points per frame");
    VFE_REQUIRE(rj->workspace && (uintptr_t)rj->workspace % 256 == 0, "reject workspace must be 256-byte aligned");
    VFE_REQUIRE(!rj->camera4 || rj->dist5, "camera4 needs dist5");
    const RjWs L = rj_layout(batch, cap);
    if (rj->workspace_bytes < L.total) {
        vfe_set_error("reject workspace too small: need %zu bytes, got %zu", L.total, rj->workspace_bytes);
        return VFE_E_NOMEM;
    }
    char* ws = (char*)rj->workspace;
    int rc;
    hipLaunchKernelGGL(k_border_status, dim3(div_up(cap, 256), batch), dim3(256), 0, st, next_pts, n_pts, cap, w, h, 1, status);
    VFE_LAUNCH_CHECK();
    const float* a = prev_pts;
    const float* b = next_pts;
    if (rj->camera4) {
        VFE_REQUIRE(rj->virtual_focal > 0, "virtual_focal must be > 0");
        float* u1 = (float*)(ws + L.u1_off);
        float* u2 = (float*)(ws + L.u2_off);
        const long long n = (long long)batch * cap;
        if ((rc = vfe_dev_undistort_points(prev_pts, u1, n, rj->camera4, rj->dist5, st))) return rc;
        if ((rc = vfe_dev_undistort_points(next_pts, u2, n, rj->camera4, rj->dist5, st))) return rc;
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_virtual_pinhole, dim3(blocks), dim3(256), 0, st, u1, n, rj->virtual_focal, w / 2.0, h / 2.0);
        hipLaunchKernelGGL(k_virtual_pinhole, dim3(blocks), dim3(256), 0, st, u2, n, rj->virtual_focal, w / 2.0, h / 2.0);
        VFE_LAUNCH_CHECK();
        a = u1; b = u2;
    }
    return vfe_dev_fundamental_ransac(a, b, n_pts, cap, batch, status, rj->f_threshold, rj->confidence, rj->max_iters, rj->seed,
                                      (double*)(ws + L.F_off), (int*)(ws + L.inl_off), ws + L.models_off, L.models_bytes, st);
}

struct FeWs { size_t gf_off, gf_bytes, next_off, status_off, corners_off, counts_off, order_off, total; };

static FeWs fe_layout(int w, int h, int batch, int cap) {
    FeWs L; size_t o = 0;
    L.gf_off = o; L.gf_bytes = vfe_good_features_workspace(w, h, batch, cap, cap); o += align_up(L.gf_bytes, 256);
    L.next_off = o; o += align_up((size_t)batch * cap * 2 * sizeof(float), 256);
    L.status_off = o; o += align_up((size_t)batch * cap, 256);
    L.corners_off = o; o += align_up((size_t)batch * cap * 2 * sizeof(float), 256);
    L.counts_off = o; o += align_up((size_t)batch * sizeof(int), 256);
    L.order_off = o; o += align_up((size_t)batch * cap * sizeof(unsigned short), 256); /* tracker's processing order */
    L.total = o;
    return L;
}

extern "C" size_t vfe_frontend_workspace(int width, int height, int batch, int max_corners) {
    if (width <= 0 || height <= 0 || batch <= 0 || max_corners <= 0) return 0;
    return fe_layout(width, height, batch, max_corners).total;
}

/* helper stream + fork/join events, one set per host thread and device, created on first use */
AuxStream* vfe_aux_stream(hipStream_t caller) {
    static thread_local AuxStream pool[32];
    static thread_local bool made[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
    if (!made[dev]) {
        /* streams and events cannot be created while the caller's stream is being captured into a graph: such a call
         * runs on the caller's stream alone (the helper exists from the first call outside a capture on) */
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (caller && (hipStreamIsCapturing(caller, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) return nullptr;
        AuxStream a;
        /* Same priority as the caller's streams.  A lower one looks attractive (freed wave slots go to the caller's short
         * dependent launches first, worth about 1 % of a step) but is a priority inversion waiting to happen: the caller's
         * stream sits at a barrier until the helper's kernel has run, and when a process keeps more streams busy than the
         * device has hardware queues, the scheduler does not get round to a lower-priority queue while higher-priority
         * ones still hold packets -- blocked ones included.  Seen as a hang with six single-stream contexts on one thread
         * and a batched pipeline on another. */
        if (hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&a.join, hipEventDisableTiming) != hipSuccess) return nullptr;
        pool[dev] = a; made[dev] = true;
    }
    return &pool[dev];
}
/* Independent kernels side by side.  The big kernels of the library are bound by vector-instruction issue, yet the waves
 * of two of them fill each other's stalls and tails: the detector's eigenvalue pass beside pyramid + tracker gains 6-11 %
 * of a front-end step from one stream to 256.  On by default for every
 * batch size; vfe_set_overlap(0) or VFE_OVERLAP=0 in the environment keeps everything on the caller's stream (e.g. to time
 * the kernels one at a time). */
static std::atomic<int> g_overlap{-1}; /* -1: not set yet, take the environment's word */
bool vfe_overlap_on() {
    if (t_vfe_call && t_vfe_call->overlap >= 0) return t_vfe_call->overlap != 0; /* the calling context's own choice */
    int m = g_overlap.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("VFE_OVERLAP");
        m = e ? (atoi(e) != 0) : 1;
        g_overlap.store(m, std::memory_order_relaxed);
    }
    return m != 0;
}
extern "C" int vfe_set_overlap(int mode) {
    VFE_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (one stream) or 1 (detector beside the tracker)");
    g_overlap.store(mode, std::memory_order_relaxed);
    return VFE_OK;
}

/* ---- the same step with OpenVINS's detector: FAST per grid cell instead of Shi-Tomasi ------------------------------------ */
struct FeFastWs { size_t next_off, status_off, corners_off, scores_off, counts_off, order_off, grid_off, grid_bytes, total; int ccap; };
static FeFastWs fe_fast_layout(int w, int h, int batch, int cap, int grid_x, int grid_y, int min_px_dist) {
    FeFastWs L; size_t o = 0;
    L.ccap = cap + grid_x * grid_y + 16; /* a cell may offer one point more than its share */
    L.next_off = o; o += align_up((size_t)batch * cap * 2 * sizeof(float), 256);
    L.status_off = o; o += align_up((size_t)batch * cap, 256);
    L.corners_off = o; o += align_up((size_t)batch * L.ccap * 2 * sizeof(float), 256);
    L.scores_off = o; o += align_up((size_t)batch * L.ccap * sizeof(int), 256);
    L.counts_off = o; o += align_up((size_t)batch * sizeof(int), 256);
    L.order_off = o; o += align_up((size_t)batch * cap * sizeof(unsigned short), 256);
    L.grid_bytes = vfe_fast_grid_workspace(w, h, batch, grid_x, grid_y, min_px_dist);
    L.grid_off = o; o += align_up(L.grid_bytes, 256);
    L.total = o;
    return L;
}
extern "C" size_t vfe_frontend_fast_workspace(int width, int height, int batch, int max_corners, int grid_x, int grid_y, int min_px_dist) {
    if (width <= 0 || height <= 0 || batch <= 0 || max_corners <= 0 || grid_x <= 0 || grid_y <= 0 || min_px_dist <= 0) return 0;
    return fe_fast_layout(width, height, batch, max_corners, grid_x, grid_y, min_px_dist).total;
}

extern "C" int vfe_dev_frontend_step_fast(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                          float* pts, int* n_pts, int* n_tracked, int max_corners,
                                          int fast_threshold, int grid_x, int grid_y, int min_px_dist,
                                          int win, int max_iter, double eps, double min_eig_thr,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    VFE_REQUIRE(prev_levels && next_levels && nlevels >= 1 && nlevels <= 8, "nlevels must be 1..8");
    VFE_REQUIRE(pts && n_pts && workspace, "NULL buffer");
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    VFE_REQUIRE(grid_x > 0 && grid_y > 0 && grid_x * grid_y <= 4096 && min_px_dist > 0 && max_corners > 0, "bad grid parameters");
    int rc = vfe_check_dimg(&next_levels[0], "next level 0");
    if (rc) return rc;
    const int w = next_levels[0].width, h = next_levels[0].height, batch = next_levels[0].batch, cap = max_corners;
    const FeFastWs L = fe_fast_layout(w, h, batch, cap, grid_x, grid_y, min_px_dist);
    if (workspace_bytes < L.total) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
        return VFE_E_NOMEM;
    }
    char* ws = (char*)workspace;
    float* next_pts = (float*)(ws + L.next_off);
    uint8_t* status = (uint8_t*)(ws + L.status_off);
    float* corners = (float*)(ws + L.corners_off);
    int* scores = (int*)(ws + L.scores_off);
    int* counts = (int*)(ws + L.counts_off);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = vfe_dev_build_pyramid(next_levels, nlevels, stream))) return rc;
    unsigned short* order = (cap <= 65536 && (long long)batch * cap >= 16384) ? (unsigned short*)(ws + L.order_off) : nullptr;
    if (order && (rc = klt_order_launch(pts, n_pts, cap, batch, w, h, nlevels, order, st))) return rc;
    if ((rc = klt_track_ordered(prev_levels, next_levels, nlevels, pts, n_pts, cap, next_pts, status, nullptr,
                                win, max_iter, eps, min_eig_thr, 0, order, st))) return rc;
    if ((rc = vfe_dev_compact_tracks(next_pts, status, n_pts, cap, batch, w, h, 1, pts, n_pts, stream))) return rc;
    if (n_tracked) VFE_HIP(hipMemcpyAsync(n_tracked, n_pts, (size_t)batch * sizeof(int), hipMemcpyDeviceToDevice, st));
    /* the surviving tracks mark the occupancy grid; every cell then offers its best corners */
    if ((rc = vfe_dev_fast_grid(&next_levels[0], fast_threshold, grid_x, grid_y, max_corners, min_px_dist, pts, n_pts, cap,
                                corners, scores, L.ccap, counts, ws + L.grid_off, L.grid_bytes, stream))) return rc;
    return vfe_dev_append_points(pts, n_pts, cap, batch, corners, counts, L.ccap, stream);
}

/* The step stage by stage, for pipelines that hold the next frame before the current step has ended: the pyramid
 * (vfe_dev_build_pyramid) and the detector's first pass touch the new frame only, the tracker needs both pyramids and the
 * points, the refill needs the tracker's and the detector's results.  Same kernels, same arguments, same results as the one
 * call; every stage runs on the stream it is given and the caller orders them. */
static int fe_stage_args(const vfe_dimg* next_levels, int max_corners, void* workspace, size_t workspace_bytes, FeWs* L) {
    VFE_REQUIRE(next_levels && workspace && max_corners > 0, "NULL buffer");
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    int rc = vfe_check_dimg(&next_levels[0], "next level 0");
    if (rc) return rc;
    *L = fe_layout(next_levels[0].width, next_levels[0].height, next_levels[0].batch, max_corners);
    if (workspace_bytes < L->total) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", L->total, workspace_bytes);
        return VFE_E_NOMEM;
    }
    return VFE_OK;
}

extern "C" int vfe_dev_frontend_candidates(const vfe_dimg* next_levels, int nlevels, int max_corners, double quality,
                                           double min_distance, void* workspace, size_t workspace_bytes, void* stream) {
    VFE_REQUIRE(nlevels >= 1 && nlevels <= 8, "nlevels must be 1..8");
    FeWs L;
    int rc = fe_stage_args(next_levels, max_corners, workspace, workspace_bytes, &L);
    if (rc) return rc;
    const int w = next_levels[0].width, h = next_levels[0].height, batch = next_levels[0].batch;
    /* this pass runs beside the tracker of the frame before: the same sharing rule as inside the one call */
    const EigBeside beside = eig_beside_policy(w, h, batch, vfe_pyramid_three_levels_per_launch(next_levels, nlevels));
    return gf_eig_phase(&next_levels[0], max_corners, quality, min_distance, max_corners, (char*)workspace + L.gf_off, L.gf_bytes,
                        (hipStream_t)stream, beside.wgs_per_cu);
}

extern "C" int vfe_dev_frontend_track(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                      const float* pts, const int* n_pts, int max_corners,
                                      int win, int max_iter, double eps, double min_eig_thr,
                                      void* workspace, size_t workspace_bytes, const vfe_reject_opts* reject, void* stream) {
    VFE_REQUIRE(prev_levels && nlevels >= 1 && nlevels <= 8, "nlevels must be 1..8");
    VFE_REQUIRE(pts && n_pts, "NULL buffer");
    FeWs L;
    int rc = fe_stage_args(next_levels, max_corners, workspace, workspace_bytes, &L);
    if (rc) return rc;
    const int w = next_levels[0].width, h = next_levels[0].height, batch = next_levels[0].batch, cap = max_corners;
    char* ws = (char*)workspace;
    float* next_pts = (float*)(ws + L.next_off);
    uint8_t* status = (uint8_t*)(ws + L.status_off);
    hipStream_t st = (hipStream_t)stream;
    unsigned short* order = (cap <= 65536 && (long long)batch * cap >= 16384) ? (unsigned short*)(ws + L.order_off) : nullptr;
    if (order && (rc = klt_order_launch(pts, n_pts, cap, batch, w, h, nlevels, order, st))) return rc;
    if ((rc = klt_track_ordered(prev_levels, next_levels, nlevels, pts, n_pts, cap, next_pts, status, nullptr,
                                win, max_iter, eps, min_eig_thr, 0, order, st))) return rc;
    if (reject && reject->f_threshold > 0 && (rc = fe_reject(reject, pts, next_pts, n_pts, cap, batch, w, h, status, st))) return rc;
    return VFE_OK;
}

extern "C" int vfe_dev_frontend_refill(const vfe_dimg* next_levels, float* pts, int* n_pts, int* n_tracked, int max_corners,
                                       double quality, double min_distance, void* workspace, size_t workspace_bytes, void* stream) {
    VFE_REQUIRE(pts && n_pts, "NULL buffer");
    FeWs L;
    int rc = fe_stage_args(next_levels, max_corners, workspace, workspace_bytes, &L);
    if (rc) return rc;
    char* ws = (char*)workspace;
    return gf_select_phase(&next_levels[0], max_corners, quality, min_distance, pts, n_pts, max_corners, (float*)(ws + L.corners_off),
                           nullptr, (int*)(ws + L.counts_off), ws + L.gf_off, L.gf_bytes, (hipStream_t)stream, true,
                           (const float*)(ws + L.next_off), (const uint8_t*)(ws + L.status_off), n_tracked);
}

extern "C" int vfe_dev_frontend_step(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                     float* pts, int* n_pts, int* n_tracked, int max_corners,
                                     double quality, double min_distance,
                                     int win, int max_iter, double eps, double min_eig_thr,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return vfe_dev_frontend_step_reject(prev_levels, next_levels, nlevels, pts, n_pts, n_tracked, max_corners, quality, min_distance,
                                        win, max_iter, eps, min_eig_thr, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int vfe_dev_frontend_step_reject(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                            float* pts, int* n_pts, int* n_tracked, int max_corners,
                                            double quality, double min_distance,
                                            int win, int max_iter, double eps, double min_eig_thr,
                                            void* workspace, size_t workspace_bytes, const vfe_reject_opts* reject, void* stream) {
    VFE_REQUIRE(prev_levels && next_levels && nlevels >= 1 && nlevels <= 8, "nlevels must be 1..8");
    VFE_REQUIRE(pts && n_pts && workspace, "NULL buffer");
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    int rc = vfe_check_dimg(&next_levels[0], "next level 0");
    if (rc) return rc;
    const int w = next_levels[0].width, h = next_levels[0].height, batch = next_levels[0].batch, cap = max_corners;
    const FeWs L = fe_layout(w, h, batch, cap);
    if (workspace_bytes < L.total) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
        return VFE_E_NOMEM;
    }
    char* ws = (char*)workspace;
    float* next_pts = (float*)(ws + L.next_off);
    uint8_t* status = (uint8_t*)(ws + L.status_off);
    float* corners = (float*)(ws + L.corners_off);
    int* counts = (int*)(ws + L.counts_off);
    /* The detector's first half (eigenvalue map + candidates) depends on the new frame only, the tracker on both
     * pyramids, so they run side by side -- the first half on a helper stream forked from the caller's stream here and
     * joined before the selection. */
    hipStream_t st = (hipStream_t)stream;
    AuxStream* aux = vfe_overlap_on() ? vfe_aux_stream(st) : nullptr;
    ForkGuard guard{nullptr, st, false};
    const EigBeside beside = eig_beside_policy(w, h, batch, vfe_pyramid_three_levels_per_launch(next_levels, nlevels));
    const bool fork_late = beside.after_pyramid;
    auto fork_eig = [&]() -> int {
        VFE_HIP(hipEventRecord(aux->fork, st));
        VFE_HIP(hipStreamWaitEvent(aux->s, aux->fork, 0));
        guard.a = aux;
        int r = gf_eig_phase(&next_levels[0], cap, quality, min_distance, cap, ws + L.gf_off, L.gf_bytes, aux->s, beside.wgs_per_cu);
        if (r) return r;
        VFE_HIP(hipEventRecord(aux->join, aux->s));
        return VFE_OK;
    };
    if (aux && !fork_late && (rc = fork_eig())) return rc;
    /* the tracker takes each frame's points tile by tile (neighbours share cache lines and tend to converge together);
     * every point's result lands at its own index, so nothing downstream sees the order */
    /* (not worth an extra launch when a handful of frames leaves most of the chip idle anyway; it needs the points only,
     * so it goes in front of the pyramid instead of between the pyramid and the tracker) */
    unsigned short* order = (cap <= 65536 && (long long)batch * cap >= 16384) ? (unsigned short*)(ws + L.order_off) : nullptr;
    if (order && (rc = klt_order_launch(pts, n_pts, cap, batch, w, h, nlevels, order, st))) return rc;
    if ((rc = vfe_dev_build_pyramid(next_levels, nlevels, stream))) return rc;
    if (aux && fork_late && (rc = fork_eig())) return rc;
    if ((rc = klt_track_ordered(prev_levels, next_levels, nlevels, pts, n_pts, cap, next_pts, status, nullptr,
                                win, max_iter, eps, min_eig_thr, 0, order, st))) return rc;
    if (reject && reject->f_threshold > 0 && (rc = fe_reject(reject, pts, next_pts, n_pts, cap, batch, w, h, status, st))) return rc;
    /* the selection gathers the surviving tracks itself (no compaction launch between the tracker and the selection) */
    if (aux) {
        VFE_HIP(hipStreamWaitEvent(st, aux->join, 0));
        guard.joined = true;
    } else if ((rc = gf_eig_phase(&next_levels[0], cap, quality, min_distance, cap, ws + L.gf_off, L.gf_bytes, st))) return rc;
    /* the selection appends what it accepts to the tracked points itself (one launch less at the serial end of the step) */
    return gf_select_phase(&next_levels[0], cap, quality, min_distance, pts, n_pts, cap, corners, nullptr, counts,
                           ws + L.gf_off, L.gf_bytes, st, true, next_pts, status, n_tracked);
}
