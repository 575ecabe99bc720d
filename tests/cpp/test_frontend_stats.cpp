/* FrontEndStats: the latency window with the arithmetic of the reference's pose publisher
 * (src/slam/output/zmq_publisher.cpp:140-176: mean of the window, P99 = sorted[size * 0.99], oldest sample out first),
 * counts, JSON line; and the engine's observer hook on a machine without a GPU (frames are recorded as failed).
 * Plain main(), exit code 1 on failure, like the reference's own tests. */
#include "slam/slam_engine.hpp"
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

using namespace vi_slam;

#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) {                                                    \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

static FrontEndFrameStats frame(uint64_t i, double us, int tracked, int detected, bool ok = true) {
    FrontEndFrameStats f;
    f.frameIndex = i; f.timestampNs = 1'000'000'000 + (int64_t)i * 33'333'333; f.ok = ok; f.tracked = tracked; f.detected = detected;
    f.total = tracked + detected; f.latencyUs = us; f.status = TrackingStatus::TRACKING;
    return f;
}

int main() {
    {   /* empty */
        FrontEndStats s;
        FrontEndFrameStats f;
        CHECK(s.getAverageLatencyUs() == 0.0 && s.getP99LatencyUs() == 0.0 && s.frames() == 0 && !s.last(f));
    }
    {   /* 200 samples 1..200 us: mean 100.5, P99 = sorted[198] = 199 */
        FrontEndStats s;
        for (int i = 1; i <= 200; i++) s.record(frame(i, (double)((i * 37) % 200 + 1), 900, 50));
        CHECK(std::fabs(s.getAverageLatencyUs() - 100.5) < 1e-9);
        CHECK(s.getP99LatencyUs() == 199.0);
        CHECK(s.frames() == 200 && s.failedFrames() == 0);
        CHECK(std::fabs(s.averageTracked() - 900.0) < 1e-9 && std::fabs(s.averageDetected() - 50.0) < 1e-9);
        s.resetLatencyStats();
        CHECK(s.getAverageLatencyUs() == 0.0 && s.frames() == 200); /* the window is cleared, the frame count is not */
    }
    {   /* window of 4: the oldest sample leaves first; one sample: P99 is that sample */
        FrontEndStats s(4);
        s.record(frame(1, 1000.0, 0, 10));
        CHECK(s.getP99LatencyUs() == 1000.0);
        for (int i = 2; i <= 6; i++) s.record(frame(i, (double)i, 5, 1, i != 6));
        CHECK(std::fabs(s.getAverageLatencyUs() - 4.5) < 1e-9); /* 3, 4, 5, 6 */
        CHECK(s.getP99LatencyUs() == 6.0);                       /* index 4 * 0.99 = 3 */
        CHECK(s.failedFrames() == 1);
        FrontEndFrameStats f;
        CHECK(s.last(f) && f.frameIndex == 6 && !f.ok);
        const std::string j = s.toJSON();
        CHECK(j.find("\"frontend\":{\"frame\":6,\"ok\":false,\"tracked\":5,\"detected\":1,\"total\":6") != std::string::npos);
        CHECK(j.find("\"status\":\"TRACKING\"") != std::string::npos && j.find("\"timestamp\":1.199999998") != std::string::npos);
        CHECK(j.find("stages_us") == std::string::npos);
        f.haveStages = true; f.stageUs[0] = 12.5; f.stageUs[3] = 80.25;
        s.record(f);
        const std::string k = s.toJSON();
        CHECK(k.find("\"stages_us\":{\"k_pyr_down\":12.500,\"k_klt\":80.250}") != std::string::npos);
    }
    {   /* engine hook: VINS-Mono has no IMU gate, so every image reaches the front-end; without a GPU each is recorded as
         * failed, with one, as ok -- either way one record and one callback per frame, none for an empty image */
        std::ofstream("/tmp/vi_slam_amd_stats_cfg.yaml") << "max_cnt: 100\nmin_dist: 20\n";
        SLAMEngine engine;
        CHECK(engine.selectFramework(SLAMFrameworkType::VINS_MONO));
        CHECK(engine.initialize("/tmp/vi_slam_amd_stats_cfg.yaml"));
        CHECK(engine.loadCalibration("/tmp/vi_slam_amd_stats_cfg.yaml"));
        int calls = 0;
        uint64_t lastIndex = 0;
        engine.setFrontEndStatsCallback([&](const FrontEndFrameStats& f) { calls++; lastIndex = f.frameIndex; });
        std::vector<uint8_t> img(640 * 480, 128);
        for (int i = 0; i < 3; i++) engine.processImage(ImageView(img.data(), 480, 640, 640, 1), 2'000'000'000 + i * 33'000'000LL);
        engine.processImage(ImageView(nullptr, 0, 0, 0, 1), 2'200'000'000); /* turned away before the front-end */
        CHECK(calls == 3 && lastIndex == 3);
        CHECK(engine.frontEndStats().frames() == 3);
        FrontEndFrameStats f;
        CHECK(engine.frontEndStats().last(f) && f.timestampNs == 2'066'000'000 && f.latencyUs >= 0.0);
        if (f.ok) CHECK(f.total == 0 && engine.frontEndStats().failedFrames() == 0); /* constant frame: nothing to find */
        else CHECK(engine.frontEndStats().failedFrames() == 3 && !engine.getFrontEndError().empty());
        engine.shutdown();
    }
    std::printf("frontend stats ok\n");
    return 0;
}
