#ifndef VI_SLAM_AMD_GPU_FRONTEND_HPP
#define VI_SLAM_AMD_GPU_FRONTEND_HPP
/* RAII owner of a vfe_ctx (include/vfe.h): the only place the adapters touch the GPU library.
 * Created lazily on the first image that reaches the front-end, so adapter lifecycle calls
 * work on machines without a GPU, as the reference's do without touching OpenCV. */
#include "slam/image_view.hpp"
#include "vfe.h"
#include <string>
#include <vector>

namespace vi_slam {

class GpuFrontEnd {
public:
    GpuFrontEnd() = default;
    ~GpuFrontEnd();
    GpuFrontEnd(const GpuFrontEnd&) = delete;
    GpuFrontEnd& operator=(const GpuFrontEnd&) = delete;

    void configure(int maxFeatures, int pyramidLevels, int winSize, int orbLevels);
    /* HIP device the context lives on (`gpu_device`); a change takes effect with the next frame */
    void setDevice(int device);
    int device() const { return cfg_.device; }
    /* per-stage GPU times of this context's launches (HIP events around each kernel); readStageTimes hands out and
     * clears what has accumulated since the last read, in milliseconds per stage */
    void setStageTiming(bool on);
    bool stageTiming() const { return stageTiming_; }
    bool readStageTimes(double* ms, int count);
    /* the detector's image-only pass beside pyramid + tracker on a helper stream: 1 on, 0 off, -1 the library's default */
    void setOverlap(int mode);
    /* histogram-equalise every frame before it is used (config/vins_mono.yaml:104 `equalize`) */
    void setEqualize(bool on);
    /* VFE_EQ_NONE / VFE_EQ_HIST / VFE_EQ_CLAHE (`histogram_method`, docs/reference/15_openvins_smartphone_config.md:111) */
    void setEqualizeMode(int mode);
    /* epipolar outlier filter between tracking and refill (`F_threshold`, config/vins_mono.yaml:101); threshold <= 0: off;
     * camera4 / dist5 may be null (raw pixel coordinates) */
    void setFRejection(double threshold, const double* camera4, const double* dist5, double virtualFocal);
    /* detector behind processFrame: Shi-Tomasi (default) or OpenVINS's FAST per grid cell (`fast_threshold`, `grid_x`,
     * `grid_y`: config/openvins.yaml:29-33) */
    void setFastGridDetector(bool on, int fastThreshold, int gridX, int gridY);
    /* each returns false and records lastError() on failure; never throws */
    bool detectShiTomasi(const ImageView& img, int maxCorners, double quality, double minDistance, std::vector<float>& xy);
    bool orbExtract(const ImageView& img, int nFeatures, float scaleFactor, int nLevels, int fastThreshold,
                    std::vector<vfe_orb_kp>& kps, std::vector<uint8_t>& desc);
    bool processFrame(const ImageView& img, int64_t ts, int maxCorners, double quality, double minDistance, vfe_frame_stats& st);
    bool points(std::vector<float>& xy);
    void reset();
    void release();
    const std::string& lastError() const { return error_; }

private:
    bool ensure(const ImageView& img);
    bool fail(int rc);
    vfe_ctx* ctx_ = nullptr;
    vfe_config cfg_{0, 0, 0, 1000, 4, 21, 0};
    int equalize_ = 0;
    double fThr_ = 0, fCam_[4] = {0, 0, 0, 0}, fDist_[5] = {0, 0, 0, 0, 0}, fFocal_ = 460.0;
    bool fHaveCam_ = false;
    bool stageTiming_ = false;
    int overlap_ = -1;
    bool detFast_ = false;
    int detThr_ = 20, detGx_ = 5, detGy_ = 5;
    std::string error_;
};

}  // namespace vi_slam
#endif
