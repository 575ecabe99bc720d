/* Host-pointer context API: what a framework adapter calls from processImage
 * (reference boundary: include/slam/i_slam_framework.hpp:63; adapters' feedImage at
 * src/slam/adapters/openvins_adapter.cpp:59-86, basalt_adapter.cpp:32-64,
 * orbslam3_adapter.cpp:51-116).  Owns one stream, the device images of the previous and the
 * current frame with their LK pyramids, the track list and all workspaces; every call copies
 * the caller's host image in, enqueues the batched device entry points with batch = 1 and
 * blocks until the outputs are back in the caller's buffers. */
#include "vfe_common.h"
#include <new>
#include <string.h>

#define CTX_MAX_LK_LEVELS 8

extern "C" size_t vfe_match_workspace(int nq, long long nt);
int vfe_match_scratch_init(void** out); /* vfe_match.hip: resolves this device's copy of the built-in match scratch */

struct vfe_ctx {
    vfe_config cfg;
    hipStream_t stream;
    int pitch[CTX_MAX_LK_LEVELS], lw[CTX_MAX_LK_LEVELS], lh[CTX_MAX_LK_LEVELS];
    uint8_t* pyr[2][CTX_MAX_LK_LEVELS]; /* [slot][level]; level 0 = the uploaded gray frame */
    uint8_t* bgr;                       /* staging for 3-channel input */
    float *pts, *next_pts, *corners, *err;
    uint8_t* status;
    int *n_pts, *n_tracked, *counts;
    void* ws; size_t ws_bytes;
    vfe_orb_kp* orb_kps; uint8_t* orb_desc; int orb_cap;
    void* mws; size_t mws_bytes;        /* growable: match descriptors + workspace */
    int cur, have_prev, w, h;           /* stream state */
    double f_thr, f_cam[4], f_dist[5], f_focal; int f_have_cam; void* f_ws; size_t f_ws_bytes; /* epipolar track filter */
    int det_mode, det_thr, det_gx, det_gy; void* det_ws; size_t det_ws_bytes; int det_min_px; /* FAST-grid detector */
    float* det_pts; int* det_scores; int det_cap;
    int equalize; unsigned* eq_hist;    /* optional histogram equalisation of every uploaded frame */
    /* pinned host memory: frames are repacked here row by row and leave in ONE contiguous copy (a pitched
     * copy from pageable memory is done row by row by the runtime, ~4 ms per MB); results come back
     * through h_io / h_small */
    uint8_t* h_stage[2]; size_t stage_bytes;
    hipEvent_t stage_ev[2];             /* recorded behind the last copy out of each staging buffer */
    char* h_io; size_t io_bytes;
    int* h_small;
    VfeCallOpts opts;                   /* this context's overlap / timing choice, seen by the launch sites during its calls */
};

static int ctx_enter(vfe_ctx* c) {
    VFE_REQUIRE(c, "ctx is NULL");
    VFE_HIP(hipSetDevice(c->cfg.device));
    return VFE_OK;
}

static vfe_dimg ctx_level(vfe_ctx* c, int slot, int level, int w, int h) {
    /* levels the library writes itself carry a reflect-101 border for the tracker; level 0 is
     * uploaded as is */
    const int bd = level > 0 ? VFE_LK_BORDER : 0;
    vfe_dimg d; d.width = w; d.height = h; d.pitch = c->pitch[level]; d.batch = 1; d.border = bd;
    d.data = c->pyr[slot][level] + (size_t)bd * d.pitch + bd;
    d.frame_stride = (size_t)d.pitch * (h + 2 * bd);
    return d;
}

/* fill levels[] for a w x h frame in `slot`; returns number of levels (<= cfg.pyramid_levels) */
static int ctx_levels(vfe_ctx* c, int slot, int w, int h, int max_level, vfe_dimg* levels) {
    const int ml = vfe_lk_max_level(w, h, c->cfg.win_size, max_level);
    int lw = w, lh = h;
    for (int l = 0; l <= ml; l++) {
        levels[l] = ctx_level(c, slot, l, lw, lh);
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
    }
    return ml + 1;
}

static int ctx_upload(vfe_ctx* c, int slot, const uint8_t* data, int rows, int cols, size_t step, int channels) {
    VFE_REQUIRE(data && rows > 0 && cols > 0, "empty image");
    VFE_REQUIRE(channels == 1 || channels == 3, "channels must be 1 (gray) or 3 (BGR)");
    VFE_REQUIRE(cols <= c->cfg.max_width && rows <= c->cfg.max_height, "image larger than the context was created for");
    VFE_REQUIRE(step >= (size_t)cols * channels, "step smaller than a row");
    /* repack into the pinned staging buffer with the device pitch, then one contiguous copy.  Every successful call
     * ends with a stream synchronisation, but an entry point that returned an error may have left a copy from this
     * buffer in flight: wait for the last copy out of THIS buffer before it is overwritten (the other buffer's copy,
     * e.g. the first image of vfe_klt_track, keeps running beside this repack). */
    VFE_HIP(hipEventSynchronize(c->stage_ev[slot & 1]));
    uint8_t* hs = c->h_stage[slot & 1];
    const size_t row_bytes = (size_t)cols * channels;
    const size_t dpitch = channels == 1 ? (size_t)c->pitch[0] : (size_t)c->cfg.max_width * 3;
    /* in bands of rows: the copy engine moves a band while the host repacks the next one */
    uint8_t* dst = channels == 1 ? c->pyr[slot][0] : c->bgr;
    const size_t total = dpitch * (size_t)(rows - 1) + row_bytes;
    int nbands = total >= (1u << 20) ? 4 : total >= (1u << 18) ? 2 : 1;
    if (nbands > rows) nbands = rows; /* a band holds at least one row */
    for (int k = 0; k < nbands; k++) {
        const int r0 = (int)((long long)rows * k / nbands), r1 = (int)((long long)rows * (k + 1) / nbands);
        const size_t off = (size_t)r0 * dpitch, bytes = dpitch * (size_t)(r1 - r0 - 1) + row_bytes;
        if (step == dpitch) memcpy(hs + off, data + off, bytes);
        else for (int y = r0; y < r1; y++) memcpy(hs + (size_t)y * dpitch, data + (size_t)y * step, row_bytes);
        VFE_HIP(hipMemcpyAsync(dst + off, hs + off, bytes, hipMemcpyHostToDevice, c->stream));
    }
    VFE_HIP(hipEventRecord(c->stage_ev[slot & 1], c->stream));
    if (channels != 1) {
        vfe_dimg s; s.data = c->bgr; s.width = cols; s.height = rows; s.pitch = (int)dpitch; s.batch = 1; s.frame_stride = dpitch * rows; s.border = 0;
        vfe_dimg g = ctx_level(c, slot, 0, cols, rows);
        int rc = vfe_dev_bgr2gray(&s, &g, c->stream);
        if (rc) return rc;
    }
    if (c->equalize) {
        vfe_dimg g = ctx_level(c, slot, 0, cols, rows);
        int rc = c->equalize == VFE_EQ_CLAHE
                     ? vfe_dev_clahe(&g, &g, 10.0, 8, 8, c->eq_hist, vfe_clahe_workspace(1, 8, 8), c->stream)
                     : vfe_dev_equalize_hist(&g, &g, c->eq_hist, vfe_equalize_workspace(1), c->stream);
        if (rc) return rc;
    }
    return VFE_OK;
}

extern "C" void vfe_destroy(vfe_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int s = 0; s < 2; s++)
        for (int l = 0; l < CTX_MAX_LK_LEVELS; l++) (void)hipFree(c->pyr[s][l]);
    void* bufs[] = {c->bgr, c->pts, c->next_pts, c->corners, c->err, c->status, c->n_pts, c->counts,
                    c->ws, c->orb_kps, c->orb_desc, c->mws, c->eq_hist};
    for (void* p : bufs) (void)hipFree(p);
    (void)hipFree(c->f_ws);
    (void)hipFree(c->det_ws); (void)hipFree(c->det_pts); (void)hipFree(c->det_scores);
    (void)hipHostFree(c->h_stage[0]); (void)hipHostFree(c->h_stage[1]); (void)hipHostFree(c->h_io); (void)hipHostFree(c->h_small);
    for (int s = 0; s < 2; s++) if (c->stage_ev[s]) (void)hipEventDestroy(c->stage_ev[s]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    vfe_prof_forget_owner(c);
    delete c;
}

extern "C" int vfe_create(const vfe_config* cfg, vfe_ctx** out) {
    VFE_REQUIRE(cfg && out, "NULL argument");
    *out = nullptr;
    VFE_REQUIRE(cfg->max_width >= 8 && cfg->max_height >= 8 && cfg->max_width < 32768 && cfg->max_height < 32768, "bad max image size");
    VFE_REQUIRE(cfg->max_features > 0 && cfg->max_features <= (1 << 20), "bad max_features");
    VFE_REQUIRE(cfg->pyramid_levels >= 1 && cfg->pyramid_levels <= CTX_MAX_LK_LEVELS, "pyramid_levels must be 1..8");
    VFE_REQUIRE(cfg->win_size >= 3 && cfg->win_size <= 41 && (cfg->win_size & 1), "win_size must be odd, 3..41");
    VFE_REQUIRE(cfg->orb_levels >= 0 && cfg->orb_levels <= 12, "orb_levels must be 0..12");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        vfe_set_error("no HIP device is visible: the front-end has no CPU fallback");
        return VFE_E_NODEVICE;
    }
    VFE_REQUIRE(cfg->device >= 0 && cfg->device < ndev, "device ordinal out of range");
    VFE_HIP(hipSetDevice(cfg->device));
    {
        /* the library carries gfx950 code objects only: on any other device the first launch would fail with a
         * generic HIP error, so say so here */
        hipDeviceProp_t prop;
        VFE_HIP(hipGetDeviceProperties(&prop, cfg->device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            vfe_set_error("device %d is %s: this library is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);
            return VFE_E_NODEVICE;
        }
        const int rcs = vfe_match_scratch_init(nullptr);
        if (rcs != VFE_OK) return rcs;
    }
    vfe_ctx* c = new (std::nothrow) vfe_ctx;
    if (!c) { vfe_set_error("out of host memory"); return VFE_E_NOMEM; }
    memset(c, 0, sizeof(*c));
    c->cfg = *cfg;
    c->opts.owner = c; c->opts.overlap = -1; c->opts.prof = -1;
#define CTX_TRY(call)                                                           \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            vfe_set_error("%s failed: %s", #call, hipGetErrorString(e_));       \
            vfe_destroy(c);                                                     \
            return e_ == hipErrorOutOfMemory ? VFE_E_NOMEM : VFE_E_HIP;         \
        }                                                                       \
    } while (0)
    CTX_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int s = 0; s < 2; s++) CTX_TRY(hipEventCreateWithFlags(&c->stage_ev[s], hipEventDisableTiming));
    int lw = cfg->max_width, lh = cfg->max_height;
    for (int l = 0; l < cfg->pyramid_levels; l++) {
        const int bd = l > 0 ? VFE_LK_BORDER : 0;
        c->lw[l] = lw; c->lh[l] = lh; c->pitch[l] = (int)align_up((size_t)lw + 2 * bd, 256);
        for (int s = 0; s < 2; s++) CTX_TRY(hipMalloc((void**)&c->pyr[s][l], (size_t)c->pitch[l] * (lh + 2 * bd)));
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
    }
    const size_t N = (size_t)cfg->max_features;
    CTX_TRY(hipMalloc((void**)&c->bgr, (size_t)cfg->max_width * 3 * cfg->max_height));
    {
        const size_t a = vfe_equalize_workspace(1), b = vfe_clahe_workspace(1, 8, 8);
        CTX_TRY(hipMalloc((void**)&c->eq_hist, a > b ? a : b));
    }
    CTX_TRY(hipMalloc((void**)&c->pts, N * 2 * sizeof(float)));
    CTX_TRY(hipMalloc((void**)&c->next_pts, N * 2 * sizeof(float)));
    CTX_TRY(hipMalloc((void**)&c->corners, N * 2 * sizeof(float)));
    CTX_TRY(hipMalloc((void**)&c->err, N * sizeof(float)));
    CTX_TRY(hipMalloc((void**)&c->status, N));
    CTX_TRY(hipMalloc((void**)&c->n_pts, 256));
    c->n_tracked = c->n_pts + 1; /* next to each other: one copy brings both counts back */
    CTX_TRY(hipMalloc((void**)&c->counts, 256));
    CTX_TRY(hipMemsetAsync(c->n_pts, 0, 256, c->stream));
    size_t ws = vfe_frontend_workspace(cfg->max_width, cfg->max_height, 1, cfg->max_features);
    if (cfg->orb_levels > 0) {
        const size_t o = vfe_orb_workspace(cfg->max_width, cfg->max_height, 1, 1.2f, cfg->orb_levels);
        if (o > ws) ws = o;
        c->orb_cap = 2 * cfg->max_features + 256;
        CTX_TRY(hipMalloc((void**)&c->orb_kps, (size_t)c->orb_cap * sizeof(vfe_orb_kp)));
        CTX_TRY(hipMalloc((void**)&c->orb_desc, (size_t)c->orb_cap * 32));
    }
    c->ws_bytes = ws;
    CTX_TRY(hipMalloc(&c->ws, ws));
    {
        const size_t g = (size_t)c->pitch[0] * cfg->max_height, b3 = (size_t)cfg->max_width * 3 * cfg->max_height;
        c->stage_bytes = g > b3 ? g : b3;
        for (int s = 0; s < 2; s++) CTX_TRY(hipHostMalloc((void**)&c->h_stage[s], c->stage_bytes, hipHostMallocDefault));
        c->io_bytes = N * 32 + (size_t)c->orb_cap * (sizeof(vfe_orb_kp) + 32) + 256;
        CTX_TRY(hipHostMalloc((void**)&c->h_io, c->io_bytes, hipHostMallocDefault));
        CTX_TRY(hipHostMalloc((void**)&c->h_small, 256, hipHostMallocDefault));
    }
    CTX_TRY(hipStreamSynchronize(c->stream));
#undef CTX_TRY
    *out = c;
    return VFE_OK;
}

extern "C" int vfe_set_f_rejection(vfe_ctx* c, double threshold, const double* camera4, const double* dist5, double virtual_focal) {
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    if (!(threshold > 0)) { c->f_thr = 0; return VFE_OK; }
    VFE_REQUIRE(c->cfg.max_features <= 2048, "the epipolar filter takes at most 2048 points: create the context with max_features <= 2048");
    VFE_REQUIRE(!camera4 || (dist5 && virtual_focal > 0 && camera4[0] != 0.0 && camera4[1] != 0.0), "camera4 needs dist5 and a virtual focal length");
    if (!c->f_ws) {
        const size_t need = vfe_reject_workspace(1, c->cfg.max_features);
        VFE_HIP(hipMalloc(&c->f_ws, need));
        c->f_ws_bytes = need;
    }
    c->f_have_cam = camera4 ? 1 : 0;
    if (camera4) { memcpy(c->f_cam, camera4, sizeof(c->f_cam)); memcpy(c->f_dist, dist5, sizeof(c->f_dist)); c->f_focal = virtual_focal; }
    c->f_thr = threshold;
    return VFE_OK;
}

extern "C" int vfe_set_detector(vfe_ctx* c, int mode, int fast_threshold, int grid_x, int grid_y) {
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    VFE_REQUIRE(mode == VFE_DET_SHI_TOMASI || mode == VFE_DET_FAST_GRID, "mode must be VFE_DET_SHI_TOMASI or VFE_DET_FAST_GRID");
    if (mode == VFE_DET_FAST_GRID) {
        VFE_REQUIRE(fast_threshold >= 1 && fast_threshold <= 254, "fast_threshold must be 1..254");
        VFE_REQUIRE(grid_x > 0 && grid_y > 0 && grid_x * grid_y <= 4096, "grid must have 1..4096 cells");
    }
    VFE_HIP(hipStreamSynchronize(c->stream));
    c->det_mode = mode; c->det_thr = fast_threshold; c->det_gx = grid_x; c->det_gy = grid_y;
    c->have_prev = 0; c->cur = 0;
    VFE_HIP(hipMemsetAsync(c->n_pts, 0, sizeof(int), c->stream));
    return VFE_OK;
}

/* buffers of the FAST-grid detector, sized for the context's largest frame and the finest occupancy grid asked for so far */
static int ctx_det_buffers(vfe_ctx* c, int min_px) {
    const int cap2 = c->cfg.max_features + c->det_gx * c->det_gy + 16;
    const size_t a = vfe_frontend_fast_workspace(c->cfg.max_width, c->cfg.max_height, 1, c->cfg.max_features, c->det_gx, c->det_gy, min_px);
    if (c->det_ws && a <= c->det_ws_bytes && cap2 <= c->det_cap && min_px >= c->det_min_px) return VFE_OK;
    VFE_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->det_ws); (void)hipFree(c->det_pts); (void)hipFree(c->det_scores);
    c->det_ws = nullptr; c->det_pts = nullptr; c->det_scores = nullptr; c->det_ws_bytes = 0; c->det_cap = 0;
    VFE_HIP(hipMalloc(&c->det_ws, a));
    VFE_HIP(hipMalloc((void**)&c->det_pts, (size_t)cap2 * 2 * sizeof(float)));
    VFE_HIP(hipMalloc((void**)&c->det_scores, (size_t)cap2 * sizeof(int)));
    c->det_ws_bytes = a; c->det_cap = cap2; c->det_min_px = min_px;
    return VFE_OK;
}

extern "C" int vfe_ctx_set_overlap(vfe_ctx* c, int mode) {
    VFE_REQUIRE(c, "ctx is NULL");
    VFE_REQUIRE(mode >= -1 && mode <= 1, "mode must be -1 (follow vfe_set_overlap), 0 or 1");
    c->opts.overlap = mode;
    return VFE_OK;
}

extern "C" int vfe_ctx_prof_enable(vfe_ctx* c, int mode) {
    VFE_REQUIRE(c, "ctx is NULL");
    VFE_REQUIRE(mode >= -1 && mode <= 1, "mode must be -1 (follow vfe_prof_enable), 0 or 1");
    c->opts.prof = mode;
    return VFE_OK;
}

extern "C" int vfe_ctx_prof_read(vfe_ctx* c, double* ms, long long* launches, int reset) {
    VFE_REQUIRE(c, "ctx is NULL");
    return vfe_prof_read_owner(c, ms, launches, reset);
}

extern "C" int vfe_set_equalize(vfe_ctx* c, int enable) {
    VFE_REQUIRE(c, "ctx is NULL");
    VFE_REQUIRE(enable == VFE_EQ_NONE || enable == VFE_EQ_HIST || enable == VFE_EQ_CLAHE, "mode must be 0, 1 or 2");
    c->equalize = enable;
    return VFE_OK;
}

extern "C" int vfe_reset(vfe_ctx* c) {
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    c->have_prev = 0; c->cur = 0; c->w = c->h = 0;
    VFE_HIP(hipMemsetAsync(c->n_pts, 0, sizeof(int), c->stream));
    VFE_HIP(hipStreamSynchronize(c->stream));
    return VFE_OK;
}

extern "C" int vfe_detect_shi_tomasi(vfe_ctx* c, const uint8_t* data, int rows, int cols, size_t step, int channels,
                                     int max_corners, double quality, double min_distance,
                                     float* corners_xy, int* n_out) {
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    VFE_REQUIRE(corners_xy && n_out, "NULL output");
    VFE_REQUIRE(max_corners > 0 && max_corners <= c->cfg.max_features, "max_corners exceeds the context's max_features");
    const int slot = c->cur ^ 1; /* scratch slot: does not disturb the stream's previous frame */
    if ((rc = ctx_upload(c, slot, data, rows, cols, step, channels))) return rc;
    vfe_dimg img = ctx_level(c, slot, 0, cols, rows);
    if ((rc = vfe_dev_good_features(&img, max_corners, quality, min_distance, nullptr, nullptr, 0, c->corners, nullptr,
                                    c->counts, c->ws, c->ws_bytes, c->stream))) return rc;
    /* the whole (small) corner buffer rides along with the count: one synchronisation */
    VFE_HIP(hipMemcpyAsync(c->h_small, c->counts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipMemcpyAsync(c->h_io, c->corners, (size_t)max_corners * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipStreamSynchronize(c->stream));
    const int n = c->h_small[0];
    if (n > 0) memcpy(corners_xy, c->h_io, (size_t)n * 2 * sizeof(float));
    *n_out = n;
    return VFE_OK;
}

extern "C" int vfe_klt_track(vfe_ctx* c, const uint8_t* prev, const uint8_t* next, int rows, int cols, size_t step,
                             const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
                             int max_level, int max_iter, double eps, double min_eig_thr) {
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    VFE_REQUIRE(prev_pts && next_pts && status, "NULL point buffers");
    VFE_REQUIRE(n >= 0 && n <= c->cfg.max_features, "n exceeds the context's max_features");
    VFE_REQUIRE(max_level >= 0 && max_level < c->cfg.pyramid_levels, "max_level exceeds the context's pyramid_levels");
    if (n == 0) return VFE_OK;
    /* uses both slots: the stream's previous frame and its tracks are forgotten, also when the call fails below */
    c->have_prev = 0;
    VFE_HIP(hipMemsetAsync(c->n_pts, 0, sizeof(int), c->stream));
    if ((rc = ctx_upload(c, 0, prev, rows, cols, step, 1)) || (rc = ctx_upload(c, 1, next, rows, cols, step, 1))) return rc;
    vfe_dimg pl[CTX_MAX_LK_LEVELS], nl[CTX_MAX_LK_LEVELS];
    const int nlev = ctx_levels(c, 0, cols, rows, max_level, pl);
    ctx_levels(c, 1, cols, rows, max_level, nl);
    if ((rc = vfe_dev_build_pyramid(pl, nlev, c->stream)) || (rc = vfe_dev_build_pyramid(nl, nlev, c->stream))) return rc;
    const size_t pb = (size_t)n * 2 * sizeof(float);
    memcpy(c->h_io, prev_pts, pb);
    c->h_small[8] = n;
    VFE_HIP(hipMemcpyAsync(c->pts, c->h_io, pb, hipMemcpyHostToDevice, c->stream));
    VFE_HIP(hipMemcpyAsync(c->n_tracked, &c->h_small[8], sizeof(int), hipMemcpyHostToDevice, c->stream));
    if ((rc = vfe_dev_klt_track(pl, nl, nlev, c->pts, c->n_tracked, c->cfg.max_features, c->next_pts, c->status, c->err,
                                c->cfg.win_size, max_iter, eps, min_eig_thr, 0, c->stream))) return rc;
    char* o_pts = c->h_io + align_up(pb, 64);
    char* o_err = o_pts + align_up(pb, 64);
    char* o_st = o_err + align_up((size_t)n * sizeof(float), 64);
    VFE_HIP(hipMemcpyAsync(o_pts, c->next_pts, pb, hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipMemcpyAsync(o_st, c->status, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipMemcpyAsync(o_err, c->err, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipStreamSynchronize(c->stream));
    memcpy(next_pts, o_pts, pb);
    memcpy(status, o_st, (size_t)n);
    if (err) memcpy(err, o_err, (size_t)n * sizeof(float));
    return VFE_OK;
}

extern "C" int vfe_orb_extract(vfe_ctx* c, const uint8_t* data, int rows, int cols, size_t step, int channels,
                               int nfeatures, float scale_factor, int nlevels, int fast_threshold,
                               vfe_orb_kp* kps, uint8_t* desc, int cap, int* n_out) {
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    VFE_REQUIRE(kps && desc && n_out && cap > 0, "NULL output");
    VFE_REQUIRE(c->cfg.orb_levels > 0, "context was created without ORB (orb_levels = 0)");
    VFE_REQUIRE(nlevels >= 1 && nlevels <= c->cfg.orb_levels, "nlevels exceeds the context's orb_levels");
    VFE_REQUIRE(scale_factor >= 1.2f, "contexts size their ORB workspace for scale_factor >= 1.2");
    const int slot = c->cur ^ 1;
    if ((rc = ctx_upload(c, slot, data, rows, cols, step, channels))) return rc;
    vfe_dimg img = ctx_level(c, slot, 0, cols, rows);
    const int dcap = cap < c->orb_cap ? cap : c->orb_cap;
    if ((rc = vfe_dev_orb_extract(&img, nfeatures, scale_factor, nlevels, fast_threshold, c->orb_kps, c->orb_desc, dcap,
                                  c->counts, c->ws, c->ws_bytes, c->stream))) return rc;
    VFE_HIP(hipMemcpyAsync(c->h_small, c->counts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipStreamSynchronize(c->stream));
    const int n = c->h_small[0];
    if (n > 0) {
        char* o_d = c->h_io + align_up((size_t)n * sizeof(vfe_orb_kp), 64);
        VFE_HIP(hipMemcpyAsync(c->h_io, c->orb_kps, (size_t)n * sizeof(vfe_orb_kp), hipMemcpyDeviceToHost, c->stream));
        VFE_HIP(hipMemcpyAsync(o_d, c->orb_desc, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
        VFE_HIP(hipStreamSynchronize(c->stream));
        memcpy(kps, c->h_io, (size_t)n * sizeof(vfe_orb_kp));
        memcpy(desc, o_d, (size_t)n * 32);
    }
    *n_out = n;
    return VFE_OK;
}

extern "C" int vfe_match_knn2(vfe_ctx* c, const uint8_t* query, int nq, const uint8_t* train, int nt,
                              float ratio, int* idx, int* dist, uint8_t* good) {
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    VFE_REQUIRE(query && idx && dist && nq > 0 && nt >= 0 && (nt == 0 || train), "NULL buffer or bad sizes");
    const size_t qb = align_up((size_t)nq * 32, 256), tb = align_up((size_t)nt * 32 + 32, 256);
    const size_t ob = align_up((size_t)nq * 2 * sizeof(int), 256), gb = align_up((size_t)nq, 256);
    const size_t wb = vfe_match_workspace(nq, nt);
    const size_t need = qb + tb + 2 * ob + gb + wb;
    if (need > c->mws_bytes) {
        VFE_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->mws); c->mws = nullptr; c->mws_bytes = 0;
        VFE_HIP(hipMalloc(&c->mws, need));
        c->mws_bytes = need;
    }
    char* p = (char*)c->mws;
    uint8_t* dq = (uint8_t*)p; uint8_t* dt = (uint8_t*)(p + qb);
    int* di = (int*)(p + qb + tb); int* dd = (int*)(p + qb + tb + ob);
    uint8_t* dg = (uint8_t*)(p + qb + tb + 2 * ob);
    void* w = p + qb + tb + 2 * ob + gb;
    VFE_HIP(hipMemcpyAsync(dq, query, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    if (nt) VFE_HIP(hipMemcpyAsync(dt, train, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    if ((rc = vfe_dev_match_knn2(dq, nq, dt, nt, 0, di, dd, w, wb, c->stream))) return rc;
    if (good && (rc = vfe_dev_ratio_test(di, dd, nq, ratio, dg, c->stream))) return rc;
    VFE_HIP(hipMemcpyAsync(idx, di, (size_t)nq * 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipMemcpyAsync(dist, dd, (size_t)nq * 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (good) VFE_HIP(hipMemcpyAsync(good, dg, (size_t)nq, hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipStreamSynchronize(c->stream));
    return VFE_OK;
}

extern "C" int vfe_process_frame(vfe_ctx* c, const uint8_t* data, int rows, int cols, size_t step, int channels,
                                 int64_t timestamp_ns, int max_corners, double quality, double min_distance,
                                 vfe_frame_stats* stats) {
    (void)timestamp_ns;
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    VFE_REQUIRE(max_corners > 0 && max_corners <= c->cfg.max_features, "max_corners exceeds the context's max_features");
    if (c->have_prev && (c->w != cols || c->h != rows)) c->have_prev = 0; /* size change restarts the stream */
    const int nxt = c->have_prev ? (c->cur ^ 1) : c->cur;
    if ((rc = ctx_upload(c, nxt, data, rows, cols, step, channels))) return rc;
    vfe_dimg pl[CTX_MAX_LK_LEVELS], nl[CTX_MAX_LK_LEVELS];
    const int nlev = ctx_levels(c, nxt, cols, rows, c->cfg.pyramid_levels - 1, nl);
    int h_n[2] = {0, 0};
    if (c->det_mode == VFE_DET_FAST_GRID) {
        const int min_px = (int)min_distance;
        VFE_REQUIRE(min_px >= 1, "the FAST-grid detector needs min_distance >= 1 pixel");
        VFE_REQUIRE(!(c->f_thr > 0), "the epipolar filter is not available with the FAST-grid detector");
        if ((rc = ctx_det_buffers(c, min_px))) return rc;
        if (!c->have_prev) {
            if ((rc = vfe_dev_build_pyramid(nl, nlev, c->stream))) return rc;
            /* first frame: every cell offers its best corners, nothing is occupied yet */
            VFE_HIP(hipMemsetAsync(c->n_pts, 0, sizeof(int), c->stream));
            const size_t gb = vfe_fast_grid_workspace(cols, rows, 1, c->det_gx, c->det_gy, min_px);
            if ((rc = vfe_dev_fast_grid(&nl[0], c->det_thr, c->det_gx, c->det_gy, max_corners, min_px, nullptr, nullptr, 0,
                                        c->det_pts, c->det_scores, c->det_cap, c->counts, c->det_ws, gb, c->stream))) return rc;
            if ((rc = vfe_dev_append_points(c->pts, c->n_pts, max_corners, 1, c->det_pts, c->counts, c->det_cap, c->stream))) return rc;
            VFE_HIP(hipMemsetAsync(c->n_tracked, 0, sizeof(int), c->stream));
        } else {
            ctx_levels(c, c->cur, cols, rows, c->cfg.pyramid_levels - 1, pl);
            const size_t need = vfe_frontend_fast_workspace(cols, rows, 1, max_corners, c->det_gx, c->det_gy, min_px);
            if ((rc = vfe_dev_frontend_step_fast(pl, nl, nlev, c->pts, c->n_pts, c->n_tracked, max_corners, c->det_thr, c->det_gx,
                                                 c->det_gy, min_px, c->cfg.win_size, 30, 0.01, 1e-4, c->det_ws, need, c->stream))) return rc;
        }
    } else if (!c->have_prev) {
        if ((rc = vfe_dev_build_pyramid(nl, nlev, c->stream))) return rc;
        if ((rc = vfe_dev_good_features(&nl[0], max_corners, quality, min_distance, nullptr, nullptr, 0, c->pts, nullptr,
                                        c->n_pts, c->ws, c->ws_bytes, c->stream))) return rc;
        VFE_HIP(hipMemsetAsync(c->n_tracked, 0, sizeof(int), c->stream));
    } else {
        ctx_levels(c, c->cur, cols, rows, c->cfg.pyramid_levels - 1, pl);
        vfe_reject_opts rj;
        memset(&rj, 0, sizeof(rj));
        rj.f_threshold = c->f_thr; rj.confidence = 0.99; rj.max_iters = 1000; rj.seed = 0x5EEDu;
        if (c->f_have_cam) { rj.camera4 = c->f_cam; rj.dist5 = c->f_dist; rj.virtual_focal = c->f_focal; }
        rj.workspace = c->f_ws; rj.workspace_bytes = c->f_ws_bytes;
        /* the filter works on arrays of max_features points, the capacity its workspace was sized for */
        VFE_REQUIRE(!(c->f_thr > 0) || max_corners <= 2048, "the epipolar filter takes at most 2048 points");
        if ((rc = vfe_dev_frontend_step_reject(pl, nl, nlev, c->pts, c->n_pts, c->n_tracked, max_corners, quality, min_distance,
                                               c->cfg.win_size, 30, 0.01, 1e-4, c->ws, c->ws_bytes, c->f_thr > 0 ? &rj : nullptr,
                                               c->stream))) return rc;
    }
    VFE_HIP(hipMemcpyAsync(&c->h_small[0], c->n_pts, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream)); /* n_pts, n_tracked */
    VFE_HIP(hipStreamSynchronize(c->stream));
    h_n[0] = c->h_small[0]; h_n[1] = c->h_small[1];
    c->cur = nxt; c->have_prev = 1; c->w = cols; c->h = rows;
    if (stats) { stats->n_total = h_n[0]; stats->n_tracked = h_n[1]; stats->n_detected = h_n[0] - h_n[1]; }
    return VFE_OK;
}

extern "C" int vfe_get_points(vfe_ctx* c, float* pts_xy, int cap, int* n_out) {
    int rc = ctx_enter(c);
    if (rc) return rc;
    VfeCallScope call_scope(&c->opts);
    VFE_REQUIRE(pts_xy && n_out && cap >= 0, "NULL output");
    VFE_HIP(hipMemcpyAsync(c->h_small, c->n_pts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipMemcpyAsync(c->h_io, c->pts, (size_t)c->cfg.max_features * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    VFE_HIP(hipStreamSynchronize(c->stream));
    int n = c->h_small[0];
    if (n > cap) n = cap;
    if (n > 0) memcpy(pts_xy, c->h_io, (size_t)n * 2 * sizeof(float));
    *n_out = n;
    return VFE_OK;
}
