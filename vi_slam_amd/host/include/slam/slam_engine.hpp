#ifndef VI_SLAM_AMD_SLAM_ENGINE_HPP
#define VI_SLAM_AMD_SLAM_ENGINE_HPP
/* Facade with the reference's public surface (include/slam/slam_engine.hpp:35-242): framework
 * selection, mutex-serialised dispatch, pose / status callbacks.  Unlike the reference, whose
 * factory returns nullptr for three of the four types (src/slam/slam_engine.cpp:207-229),
 * every SLAMFrameworkType constructs here.  ROS / ZMQ fan-out is not part of the front-end. */
#include "slam/i_slam_framework.hpp"
#include <functional>
#include <memory>
#include <mutex>
#include <string>

namespace vi_slam {

enum class SLAMFrameworkType { VINS_MONO, OPENVINS, ORB_SLAM3, BASALT };
using PoseCallback = std::function<void(const Pose6DoF&)>;
using StatusCallback = std::function<void(TrackingStatus)>;

class SLAMEngine {
public:
    SLAMEngine();
    ~SLAMEngine();
    SLAMEngine(const SLAMEngine&) = delete;
    SLAMEngine& operator=(const SLAMEngine&) = delete;

    bool selectFramework(SLAMFrameworkType type);
    SLAMFrameworkType getCurrentFramework() const;
    bool initialize(const std::string& configPath);
    bool loadCalibration(const std::string& calibPath);
    void processImage(const ImageView& image, int64_t timestampNs);
    void processIMU(const IMUSample& imu);
    bool getPose(Pose6DoF& pose) const;
    TrackingStatus getStatus() const;
    std::vector<MapPoint> getMapPoints() const;
    void reset();
    void shutdown();
    void setPoseCallback(PoseCallback callback);
    void setStatusCallback(StatusCallback callback);
    /* additions of this build */
    int getTrackedFeatureCount() const;
    std::string getFrontEndError() const;
    /* Front-end statistics for observers (the egress the reference has only for poses, zmq_publisher.hpp:85-118):
     * the callback fires after every processImage that reached an adapter, on the caller's thread with the engine's
     * mutexes held -- like the pose and status callbacks it must not call back into the engine; frontEndStats() keeps
     * the running window (average / P99 latency, counts) and can be read from any thread. */
    void setFrontEndStatsCallback(FrontEndStatsCallback callback);
    const FrontEndStats& frontEndStats() const { return stats_; }
    FrontEndStats& frontEndStats() { return stats_; }
    /* per-stage GPU times in every FrontEndFrameStats (HIP events around each kernel of this engine's own front-end
     * context: engines in one process do not see or switch each other's timing); survives selectFramework */
    void enableStageTiming(bool on);
    ISLAMFramework* framework() { return framework_.get(); }

private:
    std::unique_ptr<ISLAMFramework> createFramework(SLAMFrameworkType type);
    void invokePoseCallback(const Pose6DoF& pose);
    void invokeStatusCallback(TrackingStatus status);
    std::unique_ptr<ISLAMFramework> framework_;
    SLAMFrameworkType currentType_;
    PoseCallback poseCallback_;
    StatusCallback statusCallback_;
    FrontEndStatsCallback statsCallback_;
    FrontEndStats stats_;
    bool stageTiming_ = false;
    mutable std::mutex frameworkMutex_, callbackMutex_;
    bool initialized_;
};

}  // namespace vi_slam
#endif
