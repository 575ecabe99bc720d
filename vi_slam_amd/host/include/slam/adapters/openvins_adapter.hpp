#ifndef VI_SLAM_AMD_OPENVINS_ADAPTER_HPP
#define VI_SLAM_AMD_OPENVINS_ADAPTER_HPP
#include "slam/adapters/adapter_base.hpp"

namespace vi_slam {

/* Same fields and defaults as the reference's OpenVINSConfig
 * (include/slam/adapters/openvins_adapter.hpp:24-49). */
struct OpenVINSConfig {
    int numFeatures = 200;
    int numAruco = 0;
    bool useKLT = true;
    int fastThreshold = 20;
    int gridX = 5;
    int gridY = 5;
    int minPxDist = 10;
    double sigmaAcc = 0.1, sigmaGyro = 0.01, sigmaAccBias = 0.001, sigmaGyroBias = 0.0001;
    int maxClones = 11, maxSlam = 50, maxMsckfUpdate = 999;
    bool useZeroVelocityUpdate = true;
    double imuInitWindow = 1.0;
    int maxImuWaiting = 3;
    int numKltLevels = 3;  /* num_klt_levels (config/openvins.yaml:39); the reference has no parser for it */
    /* `gpu_frontend: track` (a key of this build, absent from the reference's files): make use_klt / fast_threshold /
     * grid_x / grid_y / min_px_dist / num_klt_levels drive the front-end they describe -- KLT tracking with a FAST-grid
     * refill, OpenVINS's own scheme -- instead of the reference adapter's one goodFeaturesToTrack call per frame, which
     * stays the default (`gpu_frontend: detect`) */
    bool trackFrontEnd = false;
};

/* OpenVINS adapter: feedImage is goodFeaturesToTrack(gray, numFeatures, 0.01, minPxDist) as at
 * src/slam/adapters/openvins_adapter.cpp:59-86, computed on the GPU; with `gpu_frontend: track` and `use_klt: true` it is
 * pyramidal KLT of the previous frame's points plus a FAST-grid refill (vfe_process_frame with VFE_DET_FAST_GRID). */
class OpenVINSAdapter : public AdapterBase {
public:
    OpenVINSAdapter();
    ~OpenVINSAdapter() override;
    OpenVINSConfig getConfig() const { return config_; }
    const std::vector<float>& getCorners() const { return corners_; }

protected:
    void configKey(const std::string& section, const std::string& key, const std::string& value) override;
    void configDone() override;
    double imuInitWindow() const override { return config_.imuInitWindow; }
    int feedImage(const ImageView& image, int64_t timestampNs, const std::vector<IMUSample>& imu) override;
    bool estimatePose(Pose6DoF& pose, int64_t timestampNs) override;
    void feedIMU(const IMUSample& imu) override;
    void onReset() override;

private:
    OpenVINSConfig config_;
    std::vector<float> corners_;
    IMUSample lastImu_;
    int imuCount_ = 0;
    double px_ = 0, py_ = 0, pz_ = 0;
};

}  // namespace vi_slam
#endif
