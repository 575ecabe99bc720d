#ifndef VI_SLAM_AMD_ADAPTER_BASE_HPP
#define VI_SLAM_AMD_ADAPTER_BASE_HPP
/* Lifecycle shared by the four framework adapters: configuration / calibration files, the
 * IMU ring buffer with the 1-second initialisation gate, status and pose bookkeeping.  The
 * sequence of guards in processImage and the IMU sanity limits are those of the reference
 * adapters (src/slam/adapters/openvins_adapter.cpp:228-305, basalt_adapter.cpp:205-290,
 * orbslam3_adapter.cpp:235-345, vins_mono_adapter.cpp:94-140); what differs per framework is
 * supplied by the derived class. */
#include "slam/gpu_frontend.hpp"
#include "slam/i_slam_framework.hpp"
#include <atomic>
#include <deque>
#include <mutex>
#include <string>

namespace vi_slam {

class AdapterBase : public ISLAMFramework {
public:
    ~AdapterBase() override = default;
    bool initialize(const std::string& configPath) override;
    bool loadCalibration(const std::string& calibPath) override;
    void processImage(const ImageView& image, int64_t timestampNs) override;
    void processIMU(const IMUSample& imu) override;
    bool getPose(Pose6DoF& pose) const override;
    TrackingStatus getStatus() const override { return status_.load(); }
    std::vector<MapPoint> getMapPoints() const override { return {}; }
    void reset() override;
    void shutdown() override;
    int getTrackedFeatureCount() const override { return trackedFeatureCount_.load(); }
    std::string getFrontEndError() const override;
    bool getLastFrameStats(FrontEndFrameStats& out) const override;
    void enableStageTiming(bool on) override;
    bool isReadyForVision() const;
    /* `enable_gpu` / `gpu_device` of the configuration (every adapter reads them): with the GPU off the adapter keeps its
     * lifecycle, IMU handling and status logic and hands no image to the front-end -- there is no CPU fallback */
    bool gpuEnabled() const { return gpuEnabled_; }
    int gpuDevice() const { return frontEnd_.device(); }

protected:
    struct Traits {
        const char* name;
        bool needsCalibration;   /* processImage refuses to run before loadCalibration */
        bool imuGate;            /* wait imuInitWindow seconds of IMU before the first image */
        bool imuSanity;          /* reject NaN / out-of-range IMU samples */
        double maxGyro;          /* rad/s */
        size_t maxImuBuffer;
        bool trackingAfterCalibration; /* status becomes TRACKING right after calibration */
    };
    explicit AdapterBase(const Traits& t) : traits_(t) {}

    /* one `key: value` line of the configuration (section = last `name:` header seen, for
     * section-structured files); return value ignored */
    virtual void configKey(const std::string& section, const std::string& key, const std::string& value) = 0;
    virtual bool isSectionHeader(const std::string& line, std::string& section) const { (void)line; (void)section; return false; }
    virtual void configDone() {}
    virtual double imuInitWindow() const { return 1.0; }
    /* pixels -> features; returns the tracked-feature count (negative: front-end failed) */
    virtual int feedImage(const ImageView& image, int64_t timestampNs, const std::vector<IMUSample>& imuSinceLast) = 0;
    /* placeholder state estimate: true when a pose is available */
    virtual bool estimatePose(Pose6DoF& pose, int64_t timestampNs) = 0;
    virtual void feedIMU(const IMUSample& imu) { (void)imu; }
    virtual void onReset() {}
    /* called when estimatePose fails; default: TRACKING -> LOST */
    virtual bool lostWhenNoPose() const { return true; }

    /* feedImage reports how the frame's features split up (tracked from the previous frame / new); detectors that
     * start from scratch every frame report everything as detected */
    void reportFrame(int tracked, int detected, int total) { frameTracked_ = tracked; frameDetected_ = detected; frameTotal_ = total; }

    GpuFrontEnd frontEnd_;
    Traits traits_;

private:
    bool parseConfigFile(const std::string& path);
    bool checkImuInitialization(int64_t imageTimestampNs);

    std::atomic<TrackingStatus> status_{TrackingStatus::UNINITIALIZED};
    mutable std::mutex statusMutex_, poseMutex_, imuMutex_;
    Pose6DoF latestPose_;
    std::deque<IMUSample> imuBuffer_;
    int64_t lastImageTimestampNs_ = 0, lastImuTimestampNs_ = 0, initStartTimeNs_ = 0;
    std::string configPath_, calibPath_;
    std::atomic<bool> initialized_{false}, calibrated_{false};
    std::atomic<int> trackedFeatureCount_{0};
    bool imuInitialized_ = false;
    bool gpuEnabled_ = true, gpuOffLogged_ = false;
    int pendingDevice_ = -1, pendingOverlap_ = -1;
    int frameTracked_ = 0, frameDetected_ = 0, frameTotal_ = 0; /* set by feedImage through reportFrame */
    mutable std::mutex frameStatsMutex_;
    FrontEndFrameStats lastFrame_;
    uint64_t frameIndex_ = 0;
};

/* helpers for the line-oriented `key: value` files the reference parses by hand */
bool parseBool(const std::string& v);
bool splitKeyValue(const std::string& line, std::string& key, std::string& value);

}  // namespace vi_slam
#endif
