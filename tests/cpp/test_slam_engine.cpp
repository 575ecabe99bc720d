// SLAMEngine facade test in the shape of the reference's tests/test_slam_engine.cpp: framework
// selection, initialisation from files in /tmp, callbacks, reset / shutdown.  Image processing is
// exercised only up to the adapters' gates, so no GPU is needed.
#include "slam/slam_engine.hpp"
#include <cstdio>
#include <fstream>
#include <vector>

using namespace vi_slam;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main() {
    { std::ofstream f("/tmp/vsa_engine_cfg.yaml"); f << "num_pts: 200\nmax_cnt: 150\nmin_dist: 30\n"; }
    { std::ofstream f("/tmp/vsa_engine_calib.yaml"); f << "cam0: {}\n"; }
    SLAMEngine engine;
    int statusCalls = 0, poseCalls = 0;
    TrackingStatus last = TrackingStatus::UNINITIALIZED;
    engine.setStatusCallback([&](TrackingStatus s) { statusCalls++; last = s; });
    engine.setPoseCallback([&](const Pose6DoF&) { poseCalls++; });
    CHECK(engine.getStatus() == TrackingStatus::UNINITIALIZED);
    std::vector<uint8_t> frame(640 * 480, 128);
    engine.processImage(ImageView(frame.data(), 480, 640, 640, 1), 1); /* no framework yet: ignored */
    /* every framework type constructs (the reference returns nullptr for three of them) */
    const SLAMFrameworkType types[] = {SLAMFrameworkType::VINS_MONO, SLAMFrameworkType::OPENVINS,
                                       SLAMFrameworkType::ORB_SLAM3, SLAMFrameworkType::BASALT};
    for (SLAMFrameworkType t : types) {
        CHECK(engine.selectFramework(t));
        CHECK(engine.getCurrentFramework() == t);
        CHECK(!engine.initialize("/tmp/no_such_config.yaml"));
        CHECK(engine.initialize("/tmp/vsa_engine_cfg.yaml"));
        CHECK(engine.loadCalibration("/tmp/vsa_engine_calib.yaml"));
        IMUSample imu;
        imu.timestampNs = 1000000000;
        engine.processIMU(imu);
        if (t != SLAMFrameworkType::VINS_MONO) {
            const int before = statusCalls;
            engine.processImage(ImageView(frame.data(), 480, 640, 640, 1), 1000000000); /* stopped by the IMU gate */
            CHECK(statusCalls == before + 1);
            CHECK(engine.getTrackedFeatureCount() == 0);
        }
        Pose6DoF pose;
        CHECK(!engine.getPose(pose));
        CHECK(engine.getMapPoints().empty());
        engine.reset();
        CHECK(last == TrackingStatus::INITIALIZING);
    }
    engine.shutdown();
    CHECK(last == TrackingStatus::UNINITIALIZED);
    CHECK(engine.getStatus() == TrackingStatus::UNINITIALIZED);
    CHECK(poseCalls == 0);
    std::printf("All SLAMEngine tests passed\n");
    return 0;
}
