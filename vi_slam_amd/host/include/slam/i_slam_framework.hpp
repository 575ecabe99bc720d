#ifndef VI_SLAM_AMD_I_SLAM_FRAMEWORK_HPP
#define VI_SLAM_AMD_I_SLAM_FRAMEWORK_HPP
/* The adapter contract, method for method as in the reference
 * (include/slam/i_slam_framework.hpp:30-110); only the image parameter type differs
 * (ImageView instead of cv::Mat, see image_view.hpp).  The nine pure virtuals are the reference's; what this build
 * adds has default bodies, so an adapter written against the reference's interface compiles unchanged. */
#include "common/types.hpp"
#include "slam/image_view.hpp"
#include "slam/output/frontend_stats.hpp"
#include <string>
#include <vector>

namespace vi_slam {

class ISLAMFramework {
public:
    virtual ~ISLAMFramework() = default;
    virtual bool initialize(const std::string& configPath) = 0;
    virtual bool loadCalibration(const std::string& calibPath) = 0;
    virtual void processImage(const ImageView& image, int64_t timestampNs) = 0;
    virtual void processIMU(const IMUSample& imu) = 0;
    virtual bool getPose(Pose6DoF& pose) const = 0;
    virtual TrackingStatus getStatus() const = 0;
    virtual std::vector<MapPoint> getMapPoints() const = 0;
    virtual void reset() = 0;
    virtual void shutdown() = 0;
    /* additions shared by all adapters of this build (reference: per-adapter
     * getTrackedFeatureCount, include/slam/adapters/openvins_adapter.hpp:105) */
    virtual int getTrackedFeatureCount() const { return 0; }
    /* last front-end (GPU) error message, empty if none: process calls never throw */
    virtual std::string getFrontEndError() const { return std::string(); }
    /* what the front-end did with the last frame that reached it (false before the first one) */
    virtual bool getLastFrameStats(FrontEndFrameStats& out) const { (void)out; return false; }
    /* per-stage GPU times in the frame statistics of this adapter's own front-end context */
    virtual void enableStageTiming(bool on) { (void)on; }
};

}  // namespace vi_slam
#endif
