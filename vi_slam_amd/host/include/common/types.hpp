#ifndef VI_SLAM_AMD_TYPES_HPP
#define VI_SLAM_AMD_TYPES_HPP
/* Value types crossing the adapter boundary.  Same names, fields and accessors as the
 * reference's include/common/types.hpp:33-210, with plain arrays in place of Eigen members
 * (Eigen is not a dependency of the front-end); the array accessors keep the reference's
 * layouts: position xyz, orientation wxyz, covariance row-major 6x6. */
#include <cstdint>
#include <cstring>

namespace vi_slam {

struct Pose6DoF {
    int64_t timestampNs = 0;
    double position[3] = {0, 0, 0};
    double orientation[4] = {1, 0, 0, 0};  /* w, x, y, z */
    double covariance[36] = {0};
    bool valid = false;
    void getPositionArray(double* out) const { std::memcpy(out, position, sizeof(position)); }
    void getOrientationArray(double* out) const { std::memcpy(out, orientation, sizeof(orientation)); }
    void getCovarianceArray(double* out) const { std::memcpy(out, covariance, sizeof(covariance)); }
    void setPositionFromArray(const double* in) { std::memcpy(position, in, sizeof(position)); }
    void setOrientationFromArray(const double* in) { std::memcpy(orientation, in, sizeof(orientation)); }
    void setCovarianceFromArray(const double* in) { std::memcpy(covariance, in, sizeof(covariance)); }
};

struct IMUSample {
    int64_t timestampNs = 0;
    double acceleration[3] = {0, 0, 0};     /* m/s^2 */
    double angularVelocity[3] = {0, 0, 0};  /* rad/s */
    double accX() const { return acceleration[0]; }
    double accY() const { return acceleration[1]; }
    double accZ() const { return acceleration[2]; }
    double gyroX() const { return angularVelocity[0]; }
    double gyroY() const { return angularVelocity[1]; }
    double gyroZ() const { return angularVelocity[2]; }
    void setAccX(double v) { acceleration[0] = v; }
    void setAccY(double v) { acceleration[1] = v; }
    void setAccZ(double v) { acceleration[2] = v; }
    void setGyroX(double v) { angularVelocity[0] = v; }
    void setGyroY(double v) { angularVelocity[1] = v; }
    void setGyroZ(double v) { angularVelocity[2] = v; }
};

struct MapPoint {
    int64_t id = 0;
    double position[3] = {0, 0, 0};
    uint8_t color[3] = {128, 128, 128};
    int observations = 0;
};

enum class TrackingStatus { UNINITIALIZED, INITIALIZING, TRACKING, LOST, RELOCALIZATION };

}  // namespace vi_slam
#endif
