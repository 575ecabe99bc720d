// An adapter written against the reference's interface -- the nine methods of
// include/slam/i_slam_framework.hpp:30-110 and nothing else -- must compile and work against this build's
// ISLAMFramework: what the build adds to the interface has default bodies.  No GPU needed.
#include "slam/i_slam_framework.hpp"
#include <cstdio>
#include <memory>

using namespace vi_slam;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

namespace {
class ThirdPartyAdapter : public ISLAMFramework {
public:
    bool initialize(const std::string& configPath) override { init_ = !configPath.empty(); return init_; }
    bool loadCalibration(const std::string&) override { return true; }
    void processImage(const ImageView& image, int64_t timestampNs) override { if (!image.empty()) { frames_++; lastTs_ = timestampNs; } }
    void processIMU(const IMUSample&) override { imu_++; }
    bool getPose(Pose6DoF& pose) const override { pose.timestampNs = lastTs_; return frames_ > 0; }
    TrackingStatus getStatus() const override { return frames_ ? TrackingStatus::TRACKING : TrackingStatus::INITIALIZING; }
    std::vector<MapPoint> getMapPoints() const override { return {}; }
    void reset() override { frames_ = 0; }
    void shutdown() override { init_ = false; }
    int frames_ = 0, imu_ = 0;
    int64_t lastTs_ = 0;
    bool init_ = false;
};
}  // namespace

int main() {
    std::unique_ptr<ISLAMFramework> f = std::make_unique<ThirdPartyAdapter>();
    CHECK(f->initialize("cfg.yaml"));
    CHECK(f->loadCalibration("calib.yaml"));
    unsigned char px[64 * 48] = {0};
    f->processIMU(IMUSample());
    f->processImage(ImageView(px, 48, 64, 64, 1), 42);
    Pose6DoF pose;
    CHECK(f->getPose(pose) && pose.timestampNs == 42);
    CHECK(f->getStatus() == TrackingStatus::TRACKING);
    CHECK(f->getMapPoints().empty());
    /* the build's additions answer with their defaults */
    CHECK(f->getTrackedFeatureCount() == 0);
    CHECK(f->getFrontEndError().empty());
    FrontEndFrameStats st;
    CHECK(!f->getLastFrameStats(st));
    f->enableStageTiming(true);
    f->reset();
    CHECK(f->getStatus() == TrackingStatus::INITIALIZING);
    f->shutdown();
    std::printf("reference-shaped adapter: all checks passed\n");
    return 0;
}
