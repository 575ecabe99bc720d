#include "slam/slam_engine.hpp"
#include "common/logging.hpp"
#include "slam/adapters/basalt_adapter.hpp"
#include "slam/adapters/openvins_adapter.hpp"
#include "slam/adapters/orbslam3_adapter.hpp"
#include "slam/adapters/vins_mono_adapter.hpp"

namespace vi_slam {

SLAMEngine::SLAMEngine() : currentType_(SLAMFrameworkType::VINS_MONO), initialized_(false) {}
SLAMEngine::~SLAMEngine() { shutdown(); }

std::unique_ptr<ISLAMFramework> SLAMEngine::createFramework(SLAMFrameworkType type) {
    switch (type) {
        case SLAMFrameworkType::VINS_MONO: return std::make_unique<VINSMonoAdapter>();
        case SLAMFrameworkType::OPENVINS: return std::make_unique<OpenVINSAdapter>();
        case SLAMFrameworkType::ORB_SLAM3: return std::make_unique<ORBSLAM3Adapter>();
        case SLAMFrameworkType::BASALT: return std::make_unique<BasaltAdapter>();
    }
    LOG_ERROR("SLAMEngine", "Unknown framework type");
    return nullptr;
}

bool SLAMEngine::selectFramework(SLAMFrameworkType type) {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    /* already on this framework: nothing to tear down, and it stays initialised (reference src/slam/slam_engine.cpp:25-29) */
    if (framework_ && currentType_ == type) return true;
    if (framework_) framework_->shutdown();
    framework_ = createFramework(type);
    if (!framework_) return false;
    currentType_ = type;
    initialized_ = false;
    if (stageTiming_) framework_->enableStageTiming(true);
    return true;
}

SLAMFrameworkType SLAMEngine::getCurrentFramework() const {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    return currentType_;
}

bool SLAMEngine::initialize(const std::string& configPath) {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    if (!framework_) {
        framework_ = createFramework(currentType_); /* default framework when none was selected */
        if (!framework_) return false;
        if (stageTiming_) framework_->enableStageTiming(true);
    }
    if (!framework_->initialize(configPath)) {
        LOG_ERROR("SLAMEngine", "Framework initialization failed");
        return false;
    }
    initialized_ = true;
    return true;
}

bool SLAMEngine::loadCalibration(const std::string& calibPath) {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    if (!framework_) { LOG_ERROR("SLAMEngine", "No framework selected"); return false; }
    return framework_->loadCalibration(calibPath);
}

void SLAMEngine::processImage(const ImageView& image, int64_t timestampNs) {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    if (!framework_) { LOG_ERROR("SLAMEngine", "No framework available for image processing"); return; }
    if (!initialized_) { LOG_WARN("SLAMEngine", "Framework not initialized"); return; }
    FrontEndFrameStats before;
    const bool hadFrame = framework_->getLastFrameStats(before);
    framework_->processImage(image, timestampNs);
    Pose6DoF pose;
    if (framework_->getPose(pose) && pose.valid) invokePoseCallback(pose);
    invokeStatusCallback(framework_->getStatus());
    /* a frame the adapter turned away at one of its guards (no calibration, IMU gate, empty image) never reached the
     * front-end and leaves no record */
    FrontEndFrameStats fs;
    if (framework_->getLastFrameStats(fs) && (!hadFrame || fs.frameIndex != before.frameIndex)) {
        stats_.record(fs);
        std::lock_guard<std::mutex> cb(callbackMutex_);
        if (statsCallback_) statsCallback_(fs);
    }
}

void SLAMEngine::setFrontEndStatsCallback(FrontEndStatsCallback callback) {
    std::lock_guard<std::mutex> lock(callbackMutex_);
    statsCallback_ = std::move(callback);
}

void SLAMEngine::enableStageTiming(bool on) {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    stageTiming_ = on;
    if (framework_) framework_->enableStageTiming(on);
}

void SLAMEngine::processIMU(const IMUSample& imu) {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    if (!framework_ || !initialized_) return;
    framework_->processIMU(imu);
}

bool SLAMEngine::getPose(Pose6DoF& pose) const {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    return framework_ ? framework_->getPose(pose) : false;
}

TrackingStatus SLAMEngine::getStatus() const {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    return framework_ ? framework_->getStatus() : TrackingStatus::UNINITIALIZED;
}

std::vector<MapPoint> SLAMEngine::getMapPoints() const {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    return framework_ ? framework_->getMapPoints() : std::vector<MapPoint>();
}

int SLAMEngine::getTrackedFeatureCount() const {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    return framework_ ? framework_->getTrackedFeatureCount() : 0;
}

std::string SLAMEngine::getFrontEndError() const {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    return framework_ ? framework_->getFrontEndError() : std::string();
}

void SLAMEngine::reset() {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    if (!framework_) { LOG_WARN("SLAMEngine", "No framework to reset"); return; }
    framework_->reset();
    invokeStatusCallback(TrackingStatus::INITIALIZING);
}

void SLAMEngine::shutdown() {
    std::lock_guard<std::mutex> lock(frameworkMutex_);
    if (framework_) { framework_->shutdown(); framework_.reset(); }
    initialized_ = false;
    invokeStatusCallback(TrackingStatus::UNINITIALIZED);
}

void SLAMEngine::setPoseCallback(PoseCallback callback) {
    std::lock_guard<std::mutex> lock(callbackMutex_);
    poseCallback_ = std::move(callback);
}

void SLAMEngine::setStatusCallback(StatusCallback callback) {
    std::lock_guard<std::mutex> lock(callbackMutex_);
    statusCallback_ = std::move(callback);
}

void SLAMEngine::invokePoseCallback(const Pose6DoF& pose) {
    std::lock_guard<std::mutex> lock(callbackMutex_);
    if (poseCallback_) poseCallback_(pose);
}

void SLAMEngine::invokeStatusCallback(TrackingStatus status) {
    std::lock_guard<std::mutex> lock(callbackMutex_);
    if (statusCallback_) statusCallback_(status);
}

}  // namespace vi_slam
