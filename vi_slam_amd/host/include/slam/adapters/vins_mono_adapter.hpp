#ifndef VI_SLAM_AMD_VINS_MONO_ADAPTER_HPP
#define VI_SLAM_AMD_VINS_MONO_ADAPTER_HPP
#include "slam/adapters/adapter_base.hpp"

namespace vi_slam {

/* Front-end keys of config/vins_mono.yaml:38-39,91-104.  The reference adapter opens the file
 * but reads none of them (src/slam/adapters/vins_mono_adapter.cpp:192-201); they are parsed
 * here so that the file drives the GPU front-end. */
struct VINSMonoConfig {
    int imageWidth = 752;
    int imageHeight = 480;
    int maxCnt = 150;        /* max_cnt: features to keep */
    int minDist = 30;        /* min_dist: pixels between features */
    int freq = 10;
    double fThreshold = 1.0; /* F_threshold: epipolar RANSAC threshold in pixels of the virtual camera; <= 0: no filter */
    /* camera.fx .. camera.cy, camera.k1 .. camera.p2 (config/vins_mono.yaml:26-35): with them the filter works on
     * undistorted points lifted to a virtual pinhole camera of focal length 460, as VINS-Mono's rejectWithF does;
     * without them on raw pixel coordinates */
    double camera[4] = {0, 0, 0, 0};
    double distortion[5] = {0, 0, 0, 0, 0};
    bool equalize = true;
    int pyramidLevels = 4;   /* upstream VINS-Mono calls calcOpticalFlowPyrLK with maxLevel 3 */
};

/* VINS-Mono adapter (reference: include/slam/adapters/vins_mono_adapter.hpp,
 * src/slam/adapters/vins_mono_adapter.cpp:94-122).  The reference estimator ignores the image;
 * this one runs the VINS-style front-end on it: KLT from the previous frame, refill with
 * Shi-Tomasi corners up to max_cnt at min_dist.  Pose and status keep the reference's
 * placeholder semantics (no pose is ever produced). */
class VINSMonoAdapter : public AdapterBase {
public:
    VINSMonoAdapter();
    ~VINSMonoAdapter() override;
    VINSMonoConfig getConfig() const { return config_; }
    /* current tracked points (x, y pairs) */
    std::vector<float> getTrackedPoints();

protected:
    void configKey(const std::string& section, const std::string& key, const std::string& value) override;
    void configDone() override;
    int feedImage(const ImageView& image, int64_t timestampNs, const std::vector<IMUSample>& imu) override;
    bool estimatePose(Pose6DoF& pose, int64_t timestampNs) override { (void)pose; (void)timestampNs; return false; }

private:
    VINSMonoConfig config_;
};

}  // namespace vi_slam
#endif
