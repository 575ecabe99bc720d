/* The four framework adapters on top of AdapterBase and the GPU front-end. */
#include "slam/adapters/basalt_adapter.hpp"
#include "slam/adapters/openvins_adapter.hpp"
#include "slam/adapters/orbslam3_adapter.hpp"
#include "slam/adapters/vins_mono_adapter.hpp"
#include "common/logging.hpp"
#include <algorithm>
#include <sstream>

namespace vi_slam {

/* ---- VINS-Mono --------------------------------------------------------------------------- */
VINSMonoAdapter::VINSMonoAdapter() : AdapterBase({"VINSMono", false, false, false, 1e30, 1000, true}) {}
VINSMonoAdapter::~VINSMonoAdapter() { shutdown(); }

void VINSMonoAdapter::configKey(const std::string& section, const std::string& key, const std::string& value) {
    const std::string full = (section == "camera" && key.find('.') == std::string::npos) ? "camera." + key : key;
    static const char* const camKeys[4] = {"camera.fx", "camera.fy", "camera.cx", "camera.cy"};
    static const char* const distKeys[4] = {"camera.k1", "camera.k2", "camera.p1", "camera.p2"};
    for (int i = 0; i < 4; i++) {
        if (full == camKeys[i]) { config_.camera[i] = std::stod(value); return; }
        if (full == distKeys[i]) { config_.distortion[i] = std::stod(value); return; }
    }
    if (key == "image_width") config_.imageWidth = std::stoi(value);
    else if (key == "image_height") config_.imageHeight = std::stoi(value);
    else if (key == "max_cnt") config_.maxCnt = std::stoi(value);
    else if (key == "min_dist") config_.minDist = std::stoi(value);
    else if (key == "freq") config_.freq = std::stoi(value);
    else if (key == "F_threshold") config_.fThreshold = std::stod(value);
    else if (key == "equalize") config_.equalize = parseBool(value);
    else if (key == "pyramid_levels") config_.pyramidLevels = std::stoi(value);
}

void VINSMonoAdapter::configDone() {
    frontEnd_.configure(std::max(config_.maxCnt, 1), config_.pyramidLevels, 21, 0);
    frontEnd_.setEqualize(config_.equalize);
    const bool haveCamera = config_.camera[0] != 0.0 && config_.camera[1] != 0.0;
    frontEnd_.setFRejection(config_.fThreshold, haveCamera ? config_.camera : nullptr, haveCamera ? config_.distortion : nullptr, 460.0);
}

int VINSMonoAdapter::feedImage(const ImageView& image, int64_t timestampNs, const std::vector<IMUSample>&) {
    vfe_frame_stats st{0, 0, 0};
    if (!frontEnd_.processFrame(image, timestampNs, std::max(config_.maxCnt, 1), 0.01, (double)config_.minDist, st)) return -1;
    reportFrame(st.n_tracked, st.n_detected, st.n_total);
    return st.n_total;
}

std::vector<float> VINSMonoAdapter::getTrackedPoints() {
    std::vector<float> xy;
    frontEnd_.points(xy);
    return xy;
}

/* ---- OpenVINS ---------------------------------------------------------------------------- */
OpenVINSAdapter::OpenVINSAdapter() : AdapterBase({"OpenVINS", true, true, true, 10.0, 2000, false}) {
    frontEnd_.configure(config_.numFeatures, 4, 21, 0);
}
OpenVINSAdapter::~OpenVINSAdapter() { shutdown(); }

void OpenVINSAdapter::configKey(const std::string&, const std::string& key, const std::string& value) {
    if (key == "num_features" || key == "num_pts") config_.numFeatures = std::stoi(value);
    else if (key == "num_aruco") config_.numAruco = std::stoi(value);
    else if (key == "use_klt") config_.useKLT = parseBool(value);
    else if (key == "fast_threshold") config_.fastThreshold = std::stoi(value);
    else if (key == "grid_x") config_.gridX = std::stoi(value);
    else if (key == "grid_y") config_.gridY = std::stoi(value);
    else if (key == "min_px_dist") config_.minPxDist = std::stoi(value);
    else if (key == "num_klt_levels") config_.numKltLevels = std::stoi(value);
    else if (key == "sigma_a" || key == "accelerometer_noise_density") config_.sigmaAcc = std::stod(value);
    else if (key == "sigma_g" || key == "gyroscope_noise_density") config_.sigmaGyro = std::stod(value);
    else if (key == "sigma_ab" || key == "accelerometer_random_walk") config_.sigmaAccBias = std::stod(value);
    else if (key == "sigma_gb" || key == "gyroscope_random_walk") config_.sigmaGyroBias = std::stod(value);
    else if (key == "max_clones") config_.maxClones = std::stoi(value);
    else if (key == "max_slam") config_.maxSlam = std::stoi(value);
    else if (key == "max_msckf_update") config_.maxMsckfUpdate = std::stoi(value);
    else if (key == "use_zupt" || key == "use_zero_velocity_update") config_.useZeroVelocityUpdate = parseBool(value);
    else if (key == "imu_init_window") config_.imuInitWindow = std::stod(value);
    else if (key == "gpu_frontend") config_.trackFrontEnd = (value == "track" || value == "\"track\"");
}

void OpenVINSAdapter::configDone() {
    const bool track = config_.trackFrontEnd && config_.useKLT;
    const int levels = std::max(1, std::min(8, config_.numKltLevels));
    frontEnd_.configure(std::max(config_.numFeatures, 1), track ? levels : 4, 21, 0);
    frontEnd_.setFastGridDetector(track, config_.fastThreshold, config_.gridX, config_.gridY);
}

int OpenVINSAdapter::feedImage(const ImageView& image, int64_t timestampNs, const std::vector<IMUSample>&) {
    if (config_.numFeatures <= 0) { corners_.clear(); return 0; }
    if (config_.trackFrontEnd && config_.useKLT) {
        vfe_frame_stats st{0, 0, 0};
        if (!frontEnd_.processFrame(image, timestampNs, config_.numFeatures, 0.01, (double)config_.minPxDist, st)) return -1;
        frontEnd_.points(corners_);
        reportFrame(st.n_tracked, st.n_detected, st.n_total);
        return st.n_total;
    }
    if (!frontEnd_.detectShiTomasi(image, config_.numFeatures, 0.01, (double)config_.minPxDist, corners_)) return -1;
    reportFrame(0, (int)(corners_.size() / 2), (int)(corners_.size() / 2));
    return (int)(corners_.size() / 2);
}

void OpenVINSAdapter::feedIMU(const IMUSample& imu) { lastImu_ = imu; imuCount_++; }

bool OpenVINSAdapter::estimatePose(Pose6DoF& pose, int64_t) {
    /* placeholder estimate of the reference (openvins_adapter.cpp:118-135): drifts with the last IMU sample */
    if (imuCount_ < 10) return false;
    pose.timestampNs = lastImu_.timestampNs;
    px_ += lastImu_.accX() * 0.00001; py_ += lastImu_.accY() * 0.00001; pz_ += lastImu_.accZ() * 0.00001;
    pose.position[0] = px_; pose.position[1] = py_; pose.position[2] = pz_;
    pose.valid = true;
    return true;
}

void OpenVINSAdapter::onReset() { imuCount_ = 0; corners_.clear(); }

/* ---- Basalt ------------------------------------------------------------------------------ */
BasaltAdapter::BasaltAdapter() : AdapterBase({"Basalt", true, true, true, 10.0, 2000, false}) {
    frontEnd_.configure(config_.maxFlowPoints, 4, 21, 0);
}
BasaltAdapter::~BasaltAdapter() { shutdown(); }

bool BasaltAdapter::isSectionHeader(const std::string& line, std::string& section) const {
    static const char* names[] = {"optical_flow", "vio", "initialization", "solver", "imu", "output", "cam0"};
    for (const char* n : names)
        if (line.find(std::string(n) + ":") != std::string::npos) { section = n; return true; }
    return false;
}

void BasaltAdapter::configKey(const std::string& section, const std::string& key, const std::string& value) {
    if (key == "vio_mode") config_.vioMode = value;
    else if (key == "debug_mode") config_.debugMode = parseBool(value);
    else if (section == "optical_flow") {
        if (key == "max_points") config_.maxFlowPoints = std::stoi(value);
        else if (key == "quality_threshold") config_.flowQualityThreshold = std::stod(value);
        else if (key == "pyramid_levels") config_.pyramidLevels = std::stoi(value);
        else if (key == "patch_size") config_.patchSize = std::stoi(value);
        else if (key == "max_flow") config_.maxFlow = std::stoi(value);
        else if (key == "subpixel") config_.subpixel = parseBool(value);
        else if (key == "fast_threshold") config_.fastThreshold = std::stoi(value);
        else if (key == "min_distance") config_.minDistance = std::stoi(value);
    } else if (section == "vio") {
        if (key == "max_frames") config_.maxFrames = std::stoi(value);
        else if (key == "max_keyframes") config_.maxKeyframes = std::stoi(value);
        else if (key == "min_parallax") config_.minParallax = std::stod(value);
        else if (key == "use_imu_preintegration") config_.useImuPreintegration = parseBool(value);
        else if (key == "loop_closure") config_.loopClosure = parseBool(value);
        else if (key == "marginalization") config_.marginalization = value;
    } else if (section == "initialization") {
        if (key == "imu_window") config_.imuInitWindow = std::stod(value);
        else if (key == "min_features") config_.minInitFeatures = std::stoi(value);
        else if (key == "static_init") config_.staticInit = parseBool(value);
        else if (key == "max_gyro_norm") config_.maxGyroNorm = std::stod(value);
        else if (key == "max_acc_deviation") config_.maxAccDeviation = std::stod(value);
    } else if (section == "solver") {
        if (key == "max_iterations") config_.maxIterations = std::stoi(value);
        else if (key == "convergence_threshold") config_.convergenceThreshold = std::stod(value);
        else if (key == "lm_damping") config_.lmDamping = std::stod(value);
        else if (key == "use_huber") config_.useHuber = parseBool(value);
        else if (key == "huber_threshold") config_.huberThreshold = std::stod(value);
    } else if (section == "imu") {
        if (key == "rate") config_.imuRate = std::stoi(value);
        else if (key == "acc_noise") config_.accNoise = std::stod(value);
        else if (key == "gyro_noise") config_.gyroNoise = std::stod(value);
        else if (key == "acc_bias_random_walk") config_.accBiasWalk = std::stod(value);
        else if (key == "gyro_bias_random_walk") config_.gyroBiasWalk = std::stod(value);
        else if (key == "gravity") config_.gravity = std::stod(value);
    } else if (section == "output") {
        if (key == "save_trajectory") config_.saveTrajectory = parseBool(value);
        else if (key == "trajectory_format") config_.trajectoryFormat = value;
        else if (key == "save_map") config_.saveMap = parseBool(value);
        else if (key == "verbose") config_.verbosity = std::stoi(value);
    }
}

int BasaltAdapter::feedImage(const ImageView& image, int64_t, const std::vector<IMUSample>&) {
    if (config_.maxFlowPoints <= 0) { corners_.clear(); return 0; }
    if (!frontEnd_.detectShiTomasi(image, config_.maxFlowPoints, config_.flowQualityThreshold * 0.01,
                                   (double)config_.minDistance, corners_)) return -1;
    reportFrame(0, (int)(corners_.size() / 2), (int)(corners_.size() / 2));
    return (int)(corners_.size() / 2);
}

void BasaltAdapter::feedIMU(const IMUSample& imu) { lastImu_ = imu; imuCount_++; }

bool BasaltAdapter::estimatePose(Pose6DoF& pose, int64_t) {
    /* placeholder dead-reckoning of the reference (basalt_adapter.cpp:86-113) */
    if (imuCount_ < 10) return false;
    pose.timestampNs = lastImu_.timestampNs;
    const double dt = 0.005;
    const double a[3] = {lastImu_.accX(), lastImu_.accY(), lastImu_.accZ() - config_.gravity};
    for (int i = 0; i < 3; i++) {
        v_[i] += a[i] * dt;
        p_[i] = std::max(-100.0, std::min(100.0, p_[i] + v_[i] * dt));
        pose.position[i] = p_[i] * 0.001;
    }
    pose.valid = true;
    return true;
}

void BasaltAdapter::onReset() { imuCount_ = 0; corners_.clear(); }

/* ---- ORB-SLAM3 --------------------------------------------------------------------------- */
ORBSLAM3Adapter::ORBSLAM3Adapter() : AdapterBase({"ORB-SLAM3", true, true, true, 20.0, 2000, false}) {
    frontEnd_.configure(config_.nFeatures, 1, 21, 8);
}
ORBSLAM3Adapter::~ORBSLAM3Adapter() { shutdown(); }

void ORBSLAM3Adapter::configKey(const std::string&, const std::string& key, const std::string& value) {
    /* a `Tbc:` header is followed by up to four rows of the 4x4 matrix */
    if (key == "Vocabulary") config_.vocabularyPath = value;
    else if (key == "sensor_type") config_.sensorType = value;
    else if (key == "Camera.type") config_.cameraType = value;
    else if (key == "Camera.fx") config_.fx = std::stod(value);
    else if (key == "Camera.fy") config_.fy = std::stod(value);
    else if (key == "Camera.cx") config_.cx = std::stod(value);
    else if (key == "Camera.cy") config_.cy = std::stod(value);
    else if (key == "Camera.k1") config_.k1 = std::stod(value);
    else if (key == "Camera.k2") config_.k2 = std::stod(value);
    else if (key == "Camera.p1") config_.p1 = std::stod(value);
    else if (key == "Camera.p2") config_.p2 = std::stod(value);
    else if (key == "Camera.k3") config_.k3 = std::stod(value);
    else if (key == "Camera.width") config_.imageWidth = std::stoi(value);
    else if (key == "Camera.height") config_.imageHeight = std::stoi(value);
    else if (key == "Camera.fps") config_.fps = std::stod(value);
    else if (key == "Camera.RGB") config_.isRGB = parseBool(value);
    else if (key == "Camera.bf") config_.bf = std::stod(value);
    else if (key == "ThDepth") config_.thDepth = std::stod(value);
    else if (key == "DepthMapFactor") config_.depthMapFactor = std::stod(value);
    else if (key == "IMU.Frequency") config_.imuFrequency = std::stod(value);
    else if (key == "IMU.NoiseGyro") config_.noiseGyro = std::stod(value);
    else if (key == "IMU.NoiseAcc") config_.noiseAcc = std::stod(value);
    else if (key == "IMU.GyroWalk") config_.gyroWalk = std::stod(value);
    else if (key == "IMU.AccWalk") config_.accWalk = std::stod(value);
    else if (key == "IMU.TimeOffset") config_.imuTimeOffset = std::stod(value);
    else if (key == "ORBextractor.nFeatures") config_.nFeatures = std::stoi(value);
    else if (key == "ORBextractor.scaleFactor") config_.scaleFactor = std::stod(value);
    else if (key == "ORBextractor.nLevels") config_.nLevels = std::stoi(value);
    else if (key == "ORBextractor.iniThFAST") config_.iniThFAST = std::stoi(value);
    else if (key == "ORBextractor.minThFAST") config_.minThFAST = std::stoi(value);
    else if (key == "System.UseViewer") config_.useViewer = parseBool(value);
    else if (key == "System.SaveTrajectory") config_.saveTrajectory = parseBool(value);
    else if (key == "System.TrajectoryFile") config_.trajectoryFile = value;
    else if (key == "System.nMaxMaps") config_.maxMaps = std::stoi(value);
    else if (key == "System.LocalWindow") config_.localWindow = std::stoi(value);
    else if (key == "data" && !value.empty() && value.front() == '[') {
        /* OpenCV-matrix style `data: [ ... 16 numbers ... ]` under Tbc */
        std::string v = value;
        for (char& c : v) if (c == '[' || c == ']' || c == ',') c = ' ';
        std::istringstream is(v);
        for (int i = 0; i < 16 && (is >> config_.Tbc[i]); i++) {}
    }
}

int ORBSLAM3Adapter::feedImage(const ImageView& image, int64_t, const std::vector<IMUSample>& imu) {
    frameCount_++;
    imuCount_ += (int)imu.size();
    if (config_.nFeatures <= 0) { keypoints_.clear(); descriptors_.clear(); return 0; }
    /* cv::ORB::create(nFeatures): every other parameter at OpenCV's default, as the reference passes none */
    if (!frontEnd_.orbExtract(image, config_.nFeatures, 1.2f, 8, 20, keypoints_, descriptors_)) return -1;
    const int tracked = (int)keypoints_.size();
    reportFrame(0, tracked, tracked);
    if (trackingState_ == TrackingState::NOT_INITIALIZED && frameCount_ >= 10 && tracked > 100) trackingState_ = TrackingState::OK;
    if (trackingState_ == TrackingState::OK) {
        if (tracked < 30) trackingState_ = TrackingState::RECENTLY_LOST;
    } else if (trackingState_ == TrackingState::RECENTLY_LOST) {
        if (tracked < 20) trackingState_ = TrackingState::LOST;
        else if (tracked > 50) trackingState_ = TrackingState::OK;
    }
    return tracked;
}

bool ORBSLAM3Adapter::estimatePose(Pose6DoF& pose, int64_t timestampNs) {
    if (trackingState_ != TrackingState::OK) return false;
    pose.timestampNs = timestampNs;
    pose.position[0] = frameCount_ * 0.001; /* placeholder Tcw of the reference: identity, x = frame * 1 mm */
    pose.valid = true;
    return true;
}

void ORBSLAM3Adapter::onReset() {
    trackingState_ = TrackingState::NOT_INITIALIZED;
    frameCount_ = 0; imuCount_ = 0;
    keypoints_.clear(); descriptors_.clear();
}

}  // namespace vi_slam
