// Lifecycle tests in the shape of the reference's adapter tests
// (tests/test_openvins_adapter.cpp:9-98, tests/test_orbslam3_adapter.cpp, tests/test_basalt_adapter.cpp,
// tests/test_vins_mono_adapter.cpp): dummy config / calibration files in /tmp, one IMU sample and one
// constant-128 640x480 frame at t = 1 s.  As in the reference, the IMU initialisation gate returns
// before any pixel is touched, so these run without a GPU.
#include "slam/adapters/basalt_adapter.hpp"
#include "slam/adapters/openvins_adapter.hpp"
#include "slam/adapters/orbslam3_adapter.hpp"
#include "slam/adapters/vins_mono_adapter.hpp"
#include <cstdio>
#include <fstream>
#include <vector>

using namespace vi_slam;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static void writeFile(const char* path, const char* text) { std::ofstream f(path); f << text; }

template <class A>
static int lifecycle(const char* name, const char* cfgText, bool statusTrackingAfterCalib) {
    std::printf("== %s\n", name);
    const std::string cfg = std::string("/tmp/vsa_") + name + "_cfg.yaml", cal = std::string("/tmp/vsa_") + name + "_calib.yaml";
    writeFile(cfg.c_str(), cfgText);
    writeFile(cal.c_str(), "cam0:\n  intrinsics: [458.654, 457.296, 367.215, 248.375]\n");
    A a;
    CHECK(a.getStatus() == TrackingStatus::UNINITIALIZED);
    CHECK(!a.initialize("/tmp/definitely_missing_config.yaml"));
    CHECK(a.initialize(cfg));
    CHECK(a.getStatus() == TrackingStatus::INITIALIZING);
    CHECK(!a.loadCalibration("/tmp/definitely_missing_calib.yaml"));
    CHECK(a.loadCalibration(cal));
    CHECK(a.getStatus() == (statusTrackingAfterCalib ? TrackingStatus::TRACKING : TrackingStatus::INITIALIZING));
    IMUSample imu;
    imu.timestampNs = 1000000000;
    imu.setAccZ(9.81);
    a.processIMU(imu);
    IMUSample bad = imu;
    bad.setGyroX(1e9); /* rejected by the sanity check where the framework has one */
    a.processIMU(bad);
    std::vector<uint8_t> frame(640 * 480, 128);
    if (!statusTrackingAfterCalib) {
        /* gate: less than imuInitWindow seconds of IMU -> the frame is not processed */
        a.processImage(ImageView(frame.data(), 480, 640, 640, 1), 1000000000);
        CHECK(a.getTrackedFeatureCount() == 0);
        CHECK(a.getFrontEndError().empty());
    }
    a.processImage(ImageView(), 1000000000); /* empty image: ignored */
    Pose6DoF pose;
    CHECK(!a.getPose(pose));
    CHECK(a.getMapPoints().empty());
    a.reset();
    CHECK(a.getStatus() == TrackingStatus::INITIALIZING);
    a.shutdown();
    CHECK(a.getStatus() == TrackingStatus::UNINITIALIZED);
    a.processImage(ImageView(frame.data(), 480, 640, 640, 1), 2000000000); /* after shutdown: ignored */
    return 0;
}

int main() {
    {
        OpenVINSAdapter a;
        writeFile("/tmp/vsa_ov.yaml", "# test\nnum_pts: 321\nuse_klt: false\nfast_threshold: 15\ngrid_x: 8\ngrid_y: 6\nmin_px_dist: 12\nimu_init_window: 0.5\nnum_klt_levels: 4\n");
        CHECK(a.initialize("/tmp/vsa_ov.yaml"));
        const OpenVINSConfig c = a.getConfig();
        CHECK(c.numFeatures == 321 && !c.useKLT && c.fastThreshold == 15 && c.gridX == 8 && c.gridY == 6 && c.minPxDist == 12);
        CHECK(c.imuInitWindow == 0.5 && c.numKltLevels == 4 && c.maxClones == 11);
    }
    {
        BasaltAdapter a;
        writeFile("/tmp/vsa_ba.yaml", "vio_mode: \"mono\"\noptical_flow:\n  max_points: 180\n  quality_threshold: 0.5\n  pyramid_levels: 4\n  min_distance: 8\ninitialization:\n  imu_window: 2.0\nimu:\n  gravity: 9.8\n");
        CHECK(a.initialize("/tmp/vsa_ba.yaml"));
        const BasaltConfig c = a.getConfig();
        CHECK(c.maxFlowPoints == 180 && c.flowQualityThreshold == 0.5 && c.pyramidLevels == 4 && c.minDistance == 8);
        CHECK(c.imuInitWindow == 2.0 && c.gravity == 9.8 && c.patchSize == 21 && c.vioMode == "mono");
    }
    {
        ORBSLAM3Adapter a;
        writeFile("/tmp/vsa_orb.yaml", "%YAML:1.0\nCamera.fx: 500.5\nCamera.width: 640\nORBextractor.nFeatures: 1500\nORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: 8\nORBextractor.iniThFAST: 20\nORBextractor.minThFAST: 7\n");
        CHECK(a.initialize("/tmp/vsa_orb.yaml"));
        const ORBSLAM3Config c = a.getConfig();
        CHECK(c.nFeatures == 1500 && c.nLevels == 8 && c.iniThFAST == 20 && c.minThFAST == 7 && c.fx == 500.5 && c.imageWidth == 640);
        CHECK(a.getTrackingState() == ORBSLAM3Adapter::TrackingState::NOT_INITIALIZED);
    }
    {
        VINSMonoAdapter a;
        writeFile("/tmp/vsa_vm.yaml", "image_width: 752\nimage_height: 480\nmax_cnt: 150\nmin_dist: 30\nequalize: 1\nF_threshold: 1.0\n");
        CHECK(a.initialize("/tmp/vsa_vm.yaml"));
        const VINSMonoConfig c = a.getConfig();
        CHECK(c.maxCnt == 150 && c.minDist == 30 && c.equalize && c.imageWidth == 752 && c.imageHeight == 480);
    }
    if (lifecycle<OpenVINSAdapter>("openvins", "num_pts: 200\n", false)) return 1;
    if (lifecycle<BasaltAdapter>("basalt", "optical_flow:\n  max_points: 200\n", false)) return 1;
    if (lifecycle<ORBSLAM3Adapter>("orbslam3", "ORBextractor.nFeatures: 1200\n", false)) return 1;
    std::printf("All adapter lifecycle tests passed\n");
    return 0;
}
