#include "slam/gpu_frontend.hpp"
#include "common/logging.hpp"

namespace vi_slam {

GpuFrontEnd::~GpuFrontEnd() { release(); }

void GpuFrontEnd::configure(int maxFeatures, int pyramidLevels, int winSize, int orbLevels) {
    release();
    cfg_.max_features = maxFeatures > 0 ? maxFeatures : 1;
    cfg_.pyramid_levels = pyramidLevels < 1 ? 1 : (pyramidLevels > 8 ? 8 : pyramidLevels);
    cfg_.win_size = winSize;
    cfg_.orb_levels = orbLevels;
}

void GpuFrontEnd::setDevice(int device) {
    if (device == cfg_.device) return;
    release();
    cfg_.device = device;
}

void GpuFrontEnd::setStageTiming(bool on) {
    stageTiming_ = on;
    if (ctx_) {
        (void)vfe_ctx_prof_enable(ctx_, on ? 1 : -1);
        (void)vfe_ctx_prof_read(ctx_, nullptr, nullptr, 1);
    }
}

bool GpuFrontEnd::readStageTimes(double* ms, int count) {
    static_assert(VFE_K_COUNT >= 1, "timing slots");
    if (!ctx_ || !stageTiming_ || !ms) return false;
    double all[VFE_K_COUNT];
    if (vfe_ctx_prof_read(ctx_, all, nullptr, 1) != VFE_OK) return false;
    for (int i = 0; i < count; i++) ms[i] = i < VFE_K_COUNT ? all[i] : 0.0;
    return true;
}

void GpuFrontEnd::setOverlap(int mode) {
    overlap_ = mode < 0 ? -1 : (mode ? 1 : 0);
    if (ctx_) (void)vfe_ctx_set_overlap(ctx_, overlap_);
}

void GpuFrontEnd::setEqualize(bool on) { setEqualizeMode(on ? VFE_EQ_HIST : VFE_EQ_NONE); }

void GpuFrontEnd::setEqualizeMode(int mode) {
    equalize_ = (mode == VFE_EQ_HIST || mode == VFE_EQ_CLAHE) ? mode : VFE_EQ_NONE;
    if (ctx_) {
        const int rc = vfe_set_equalize(ctx_, equalize_);
        if (rc != VFE_OK) { release(); (void)fail(rc); } /* the next frame rebuilds the context and reports again */
    }
}

void GpuFrontEnd::setFRejection(double threshold, const double* camera4, const double* dist5, double virtualFocal) {
    fThr_ = threshold > 0 ? threshold : 0;
    fHaveCam_ = camera4 && dist5;
    if (fHaveCam_) {
        for (int i = 0; i < 4; i++) fCam_[i] = camera4[i];
        for (int i = 0; i < 5; i++) fDist_[i] = dist5[i];
        fFocal_ = virtualFocal;
    }
    if (ctx_) {
        const int rc = vfe_set_f_rejection(ctx_, fThr_, fHaveCam_ ? fCam_ : nullptr, fHaveCam_ ? fDist_ : nullptr, fFocal_);
        if (rc != VFE_OK) { release(); (void)fail(rc); }
    }
}

void GpuFrontEnd::setFastGridDetector(bool on, int fastThreshold, int gridX, int gridY) {
    detFast_ = on; detThr_ = fastThreshold; detGx_ = gridX; detGy_ = gridY;
    if (ctx_) {
        const int rc = vfe_set_detector(ctx_, on ? VFE_DET_FAST_GRID : VFE_DET_SHI_TOMASI, detThr_, detGx_, detGy_);
        if (rc != VFE_OK) { release(); (void)fail(rc); }
    }
}

void GpuFrontEnd::release() {
    if (ctx_) { vfe_destroy(ctx_); ctx_ = nullptr; }
}

bool GpuFrontEnd::fail(int rc) {
    error_ = std::string(vfe_status_string(rc)) + ": " + vfe_last_error();
    LOG_ERROR("GpuFrontEnd", "%s", error_.c_str());
    return false;
}

bool GpuFrontEnd::ensure(const ImageView& img) {
    if (ctx_ && img.cols <= cfg_.max_width && img.rows <= cfg_.max_height) return true;
    release();
    cfg_.max_width = img.cols > cfg_.max_width ? img.cols : cfg_.max_width;
    cfg_.max_height = img.rows > cfg_.max_height ? img.rows : cfg_.max_height;
    const int rc = vfe_create(&cfg_, &ctx_);
    if (rc != VFE_OK) { ctx_ = nullptr; return fail(rc); }
    /* a context that could not take its configuration must not survive: the next frame would find it alive and run
     * with the option silently off */
    const int rcE = vfe_set_equalize(ctx_, equalize_);
    if (rcE != VFE_OK) { release(); return fail(rcE); }
    if (detFast_) {
        const int rcD = vfe_set_detector(ctx_, VFE_DET_FAST_GRID, detThr_, detGx_, detGy_);
        if (rcD != VFE_OK) { release(); return fail(rcD); }
    }
    if (fThr_ > 0) {
        const int rcF = vfe_set_f_rejection(ctx_, fThr_, fHaveCam_ ? fCam_ : nullptr, fHaveCam_ ? fDist_ : nullptr, fFocal_);
        if (rcF != VFE_OK) { release(); return fail(rcF); }
    }
    (void)vfe_ctx_set_overlap(ctx_, overlap_);
    if (stageTiming_) (void)vfe_ctx_prof_enable(ctx_, 1);
    error_.clear();
    return true;
}

bool GpuFrontEnd::detectShiTomasi(const ImageView& img, int maxCorners, double quality, double minDistance, std::vector<float>& xy) {
    xy.clear();
    if (maxCorners > cfg_.max_features) { release(); cfg_.max_features = maxCorners; }
    if (!ensure(img)) return false;
    xy.resize((size_t)maxCorners * 2);
    int n = 0;
    const int rc = vfe_detect_shi_tomasi(ctx_, img.data, img.rows, img.cols, img.step, img.channels(), maxCorners, quality,
                                         minDistance, xy.data(), &n);
    if (rc != VFE_OK) { xy.clear(); return fail(rc); }
    xy.resize((size_t)n * 2);
    return true;
}

bool GpuFrontEnd::orbExtract(const ImageView& img, int nFeatures, float scaleFactor, int nLevels, int fastThreshold,
                             std::vector<vfe_orb_kp>& kps, std::vector<uint8_t>& desc) {
    kps.clear(); desc.clear();
    if (nFeatures > cfg_.max_features || nLevels > cfg_.orb_levels) {
        release();
        if (nFeatures > cfg_.max_features) cfg_.max_features = nFeatures;
        if (nLevels > cfg_.orb_levels) cfg_.orb_levels = nLevels;
    }
    if (!ensure(img)) return false;
    const int cap = 2 * nFeatures + 256;
    kps.resize(cap); desc.resize((size_t)cap * 32);
    int n = 0;
    const int rc = vfe_orb_extract(ctx_, img.data, img.rows, img.cols, img.step, img.channels(), nFeatures, scaleFactor,
                                   nLevels, fastThreshold, kps.data(), desc.data(), cap, &n);
    if (rc != VFE_OK) { kps.clear(); desc.clear(); return fail(rc); }
    kps.resize(n); desc.resize((size_t)n * 32);
    return true;
}

bool GpuFrontEnd::processFrame(const ImageView& img, int64_t ts, int maxCorners, double quality, double minDistance,
                               vfe_frame_stats& st) {
    if (maxCorners > cfg_.max_features) { release(); cfg_.max_features = maxCorners; }
    if (!ensure(img)) return false;
    const int rc = vfe_process_frame(ctx_, img.data, img.rows, img.cols, img.step, img.channels(), ts, maxCorners, quality,
                                     minDistance, &st);
    if (rc != VFE_OK) return fail(rc);
    return true;
}

bool GpuFrontEnd::points(std::vector<float>& xy) {
    xy.clear();
    if (!ctx_) return true;
    xy.resize((size_t)cfg_.max_features * 2);
    int n = 0;
    const int rc = vfe_get_points(ctx_, xy.data(), cfg_.max_features, &n);
    if (rc != VFE_OK) { xy.clear(); return fail(rc); }
    xy.resize((size_t)n * 2);
    return true;
}

void GpuFrontEnd::reset() {
    if (ctx_) (void)vfe_reset(ctx_);
}

}  // namespace vi_slam
