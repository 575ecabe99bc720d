#include "slam/adapters/adapter_base.hpp"
#include <chrono>
#include "common/logging.hpp"
#include <cmath>
#include <fstream>

namespace vi_slam {

bool parseBool(const std::string& v) { return v == "true" || v == "1" || v == "True"; }

bool splitKeyValue(const std::string& line, std::string& key, std::string& value) {
    const size_t colon = line.find(':');
    if (colon == std::string::npos) return false;
    auto trim = [](std::string s) {
        const size_t a = s.find_first_not_of(" \t\r");
        if (a == std::string::npos) return std::string();
        const size_t b = s.find_last_not_of(" \t\r");
        return s.substr(a, b - a + 1);
    };
    key = trim(line.substr(0, colon));
    value = trim(line.substr(colon + 1));
    /* trailing comment, then quotes */
    const size_t hash = value.find(" #");
    if (hash != std::string::npos) value = trim(value.substr(0, hash));
    if (value.size() >= 2 && (value.front() == '"' || value.front() == '\'')) value = value.substr(1, value.size() - 2);
    return !key.empty();
}

bool AdapterBase::parseConfigFile(const std::string& path) {
    std::ifstream file(path);
    if (!file.is_open()) return false;
    std::string line, section;
    gpuEnabled_ = true; gpuOffLogged_ = false; pendingDevice_ = -1; pendingOverlap_ = -1;
    while (std::getline(file, line)) {
        if (line.empty() || line[0] == '#') continue;
        if (isSectionHeader(line, section)) continue;
        std::string key, value;
        if (!splitKeyValue(line, key, value)) continue;
        try {
            /* keys every adapter understands (the reference's only GPU key, `processing.enable_gpu` of
             * config/pc_client_config.yaml:15, is read by nothing there) */
            if (key == "enable_gpu" || key == "processing.enable_gpu") { gpuEnabled_ = parseBool(value); continue; }
            if (key == "gpu_device") { pendingDevice_ = std::stoi(value); continue; }
            if (key == "gpu_overlap") { pendingOverlap_ = parseBool(value) ? 1 : 0; continue; }
            configKey(section, key, value);
        } catch (const std::exception& e) {
            LOG_WARN(traits_.name, "ignoring malformed value for '%s': %s", key.c_str(), e.what());
        }
    }
    configDone();
    /* after configDone: the adapters (re)configure their front-end there */
    frontEnd_.setDevice(pendingDevice_ < 0 ? 0 : pendingDevice_);
    frontEnd_.setOverlap(pendingOverlap_);
    return true;
}

bool AdapterBase::initialize(const std::string& configPath) {
    std::lock_guard<std::mutex> lock(statusMutex_);
    configPath_ = configPath;
    if (!parseConfigFile(configPath)) {
        LOG_ERROR(traits_.name, "Failed to open config file: %s", configPath.c_str());
        return false;
    }
    status_ = TrackingStatus::INITIALIZING;
    initialized_ = true;
    LOG_INFO(traits_.name, "Adapter initialized with %s", configPath.c_str());
    return true;
}

bool AdapterBase::loadCalibration(const std::string& calibPath) {
    std::lock_guard<std::mutex> lock(statusMutex_);
    calibPath_ = calibPath;
    std::ifstream f(calibPath);
    if (!f.is_open()) {
        LOG_ERROR(traits_.name, "Failed to open calibration file: %s", calibPath.c_str());
        return false;
    }
    calibrated_ = true;
    if (traits_.trackingAfterCalibration && initialized_ && !configPath_.empty()) status_ = TrackingStatus::TRACKING;
    return true;
}

bool AdapterBase::checkImuInitialization(int64_t imageTimestampNs) {
    if (imuInitialized_) return true;
    std::lock_guard<std::mutex> lock(imuMutex_);
    if (initStartTimeNs_ == 0 || imuBuffer_.empty()) return false;
    const double elapsed = (imageTimestampNs - initStartTimeNs_) * 1e-9;
    if (elapsed >= imuInitWindow()) {
        imuInitialized_ = true;
        LOG_INFO(traits_.name, "IMU initialization complete after %.3f s", elapsed);
        return true;
    }
    return false;
}

bool AdapterBase::isReadyForVision() const {
    std::lock_guard<std::mutex> lock(imuMutex_);
    if (initStartTimeNs_ == 0 || imuBuffer_.empty()) return false;
    return (lastImuTimestampNs_ - initStartTimeNs_) * 1e-9 >= imuInitWindow() && imuInitialized_;
}

void AdapterBase::processImage(const ImageView& image, int64_t timestampNs) {
    if (!initialized_) { LOG_WARN(traits_.name, "Adapter not initialized"); return; }
    if (traits_.needsCalibration && !calibrated_) { LOG_WARN(traits_.name, "Calibration not loaded"); return; }
    if (image.empty()) { LOG_WARN(traits_.name, "Received empty image"); return; }
    if (traits_.imuGate && !checkImuInitialization(timestampNs)) return;

    /* IMU samples up to this image leave the buffer; those after the previous image are handed on */
    std::vector<IMUSample> since;
    {
        std::lock_guard<std::mutex> lock(imuMutex_);
        while (!imuBuffer_.empty()) {
            const IMUSample& s = imuBuffer_.front();
            if (traits_.imuGate && s.timestampNs > timestampNs) break;
            if (!(lastImageTimestampNs_ > 0 && s.timestampNs <= lastImageTimestampNs_)) since.push_back(s);
            imuBuffer_.pop_front();
        }
    }
    frameTracked_ = frameDetected_ = frameTotal_ = 0;
    if (!gpuEnabled_) {
        if (!gpuOffLogged_) { LOG_WARN(traits_.name, "enable_gpu is false: images are accepted and not processed (no CPU fallback)"); gpuOffLogged_ = true; }
        lastImageTimestampNs_ = timestampNs;
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int tracked = feedImage(image, timestampNs, since);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    lastImageTimestampNs_ = timestampNs;
    if (tracked >= 0) trackedFeatureCount_ = tracked;

    Pose6DoF pose;
    if (estimatePose(pose, timestampNs)) {
        { std::lock_guard<std::mutex> lock(poseMutex_); latestPose_ = pose; }
        if (status_ != TrackingStatus::TRACKING) { status_ = TrackingStatus::TRACKING; LOG_INFO(traits_.name, "Tracking established"); }
    } else if (status_ == TrackingStatus::TRACKING && lostWhenNoPose()) {
        status_ = TrackingStatus::LOST;
        LOG_WARN(traits_.name, "Tracking lost");
    }
    {
        std::lock_guard<std::mutex> lock(frameStatsMutex_);
        lastFrame_ = FrontEndFrameStats();
        lastFrame_.timestampNs = timestampNs;
        lastFrame_.frameIndex = ++frameIndex_;
        lastFrame_.ok = tracked >= 0;
        lastFrame_.tracked = frameTracked_; lastFrame_.detected = frameDetected_;
        lastFrame_.total = tracked >= 0 ? (frameTotal_ ? frameTotal_ : tracked) : 0;
        lastFrame_.latencyUs = us;
        lastFrame_.status = status_.load();
        static_assert(kFrontEndStages == VFE_K_COUNT, "FrontEndFrameStats::stageUs has one slot per timing slot of the GPU library");
        double ms[kFrontEndStages];
        if (frontEnd_.stageTiming() && frontEnd_.readStageTimes(ms, kFrontEndStages)) {
            lastFrame_.haveStages = true;
            for (int i = 0; i < kFrontEndStages; i++) lastFrame_.stageUs[i] = ms[i] * 1e3;
        }
    }
}

void AdapterBase::enableStageTiming(bool on) { frontEnd_.setStageTiming(on); }

bool AdapterBase::getLastFrameStats(FrontEndFrameStats& out) const {
    std::lock_guard<std::mutex> lock(frameStatsMutex_);
    if (frameIndex_ == 0) return false;
    out = lastFrame_;
    return true;
}

void AdapterBase::processIMU(const IMUSample& imu) {
    if (!initialized_) return;
    if (traits_.imuSanity) {
        double maxAcc = 0, maxGyro = 0;
        for (int i = 0; i < 3; i++) {
            if (!std::isfinite(imu.acceleration[i]) || !std::isfinite(imu.angularVelocity[i])) {
                LOG_WARN(traits_.name, "Received invalid IMU data (NaN or Inf values)");
                return;
            }
            maxAcc = std::fmax(maxAcc, std::fabs(imu.acceleration[i]));
            maxGyro = std::fmax(maxGyro, std::fabs(imu.angularVelocity[i]));
        }
        if (maxAcc > 100.0 || maxGyro > traits_.maxGyro) {
            LOG_WARN(traits_.name, "IMU values out of reasonable range");
            return;
        }
    }
    {
        std::lock_guard<std::mutex> lock(imuMutex_);
        if (initStartTimeNs_ == 0) initStartTimeNs_ = imu.timestampNs;
        imuBuffer_.push_back(imu);
        lastImuTimestampNs_ = imu.timestampNs;
        while (imuBuffer_.size() > traits_.maxImuBuffer) imuBuffer_.pop_front();
    }
    feedIMU(imu);
}

bool AdapterBase::getPose(Pose6DoF& pose) const {
    std::lock_guard<std::mutex> lock(poseMutex_);
    pose = latestPose_;
    return latestPose_.valid;
}

std::string AdapterBase::getFrontEndError() const { return frontEnd_.lastError(); }

void AdapterBase::reset() {
    std::lock_guard<std::mutex> lock(statusMutex_);
    onReset();
    frontEnd_.reset();
    { std::lock_guard<std::mutex> l(imuMutex_); imuBuffer_.clear(); }
    { std::lock_guard<std::mutex> l(poseMutex_); latestPose_ = Pose6DoF(); }
    lastImageTimestampNs_ = lastImuTimestampNs_ = initStartTimeNs_ = 0;
    imuInitialized_ = false;
    trackedFeatureCount_ = 0;
    status_ = TrackingStatus::INITIALIZING;
}

void AdapterBase::shutdown() {
    std::lock_guard<std::mutex> lock(statusMutex_);
    onReset();
    frontEnd_.release();
    initialized_ = false;
    calibrated_ = false;
    status_ = TrackingStatus::UNINITIALIZED;
}

}  // namespace vi_slam
