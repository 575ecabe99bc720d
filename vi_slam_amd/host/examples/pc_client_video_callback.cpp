/* The body the reference's PC client leaves open in its receiver callbacks (pc_client/src/main.cpp:182-205: the video
 * callback counts frames and prints, the IMU callback counts samples), written out: frames and IMU packets go through
 * StreamIngest into a SLAMEngine, and the front-end statistics come back out through the engine's observer.
 *
 * The reference's WebRTCReceiver is itself a placeholder that fabricates 640x480 frames at 30 fps and IMU samples
 * at 200 Hz (pc_client/src/webrtc_receiver.cpp:92-155); FakeReceiver below does the same with the phone app's wire
 * formats (56-byte IMU packets, frame-metadata JSON), so this file builds and runs without WebRTC, OpenCV or a UI.
 *
 *   pc_client_video_callback [config.yaml [calibration.yaml [frames]]]
 * Without a GPU the engine reports the front-end error for every frame and the program still exits 0. */
#include "slam/slam_engine.hpp"
#include "streaming/ingest.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <vector>

using namespace vi_slam;

namespace {
/* stand-in for pc_client's WebRTCReceiver: same callback surface, deterministic content */
class FakeReceiver {
public:
    using VideoCallback = std::function<void(const ImageView& frame, const std::string& metadataJson)>;
    using ImuCallback = std::function<void(const uint8_t* datagram, std::size_t size)>;
    void setVideoCallback(VideoCallback cb) { video_ = std::move(cb); }
    void setIMUCallback(ImuCallback cb) { imu_ = std::move(cb); }
    void run(int frames, int w, int h) {
        std::vector<uint8_t> img((size_t)w * h);
        int64_t t = 1'000'000'000;           /* ns */
        const int64_t imuDt = 5'000'000;     /* 200 Hz */
        int64_t nextFrame = t + 1'200'000'000; /* the adapters want a second of IMU before the first frame */
        for (int f = 0; f < frames;) {
            IMUSample s;
            s.timestampNs = t;
            s.acceleration[0] = 0.0; s.acceleration[1] = 0.0; s.acceleration[2] = 9.81;
            s.angularVelocity[0] = s.angularVelocity[1] = 0.0; s.angularVelocity[2] = 0.01;
            uint8_t pkt[kImuPacketSize];
            encodeImuPacket(s, pkt);
            if (imu_) imu_(pkt, sizeof(pkt));
            if (t >= nextFrame) {
                /* a drifting checkerboard with a gradient: corners to find and to track */
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++)
                        img[(size_t)y * w + x] = (uint8_t)((((x + 2 * f) / 24 + (y + f) / 24) & 1 ? 200 : 60) + (x * 31 + y * 17) % 23);
                FrameMetadata m;
                m.sequenceNumber = f; m.timestampNs = t; m.width = w; m.height = h; m.exposureTimeNs = 8'000'000; m.iso = 200;
                if (video_) video_(ImageView(img.data(), h, w, (size_t)w, 1), encodeFrameMetadata(m));
                nextFrame += 33'333'333;
                f++;
            }
            t += imuDt;
        }
    }

private:
    VideoCallback video_;
    ImuCallback imu_;
};

std::string writeDefault(const char* path, const char* text) {
    std::ofstream(path) << text;
    return path;
}
}  // namespace

int main(int argc, char** argv) {
    const std::string config = argc > 1 ? argv[1] : writeDefault("/tmp/vi_slam_amd_example_config.yaml",
                                                                 "max_cnt: 150\nmin_dist: 30\nequalize: 0\nF_threshold: 1.0\n");
    const std::string calib = argc > 2 ? argv[2] : writeDefault("/tmp/vi_slam_amd_example_calib.yaml", "camera.fx: 460\n");
    const int frames = argc > 3 ? std::atoi(argv[3]) : 60;

    SLAMEngine engine;
    if (!engine.selectFramework(SLAMFrameworkType::VINS_MONO) || !engine.initialize(config) || !engine.loadCalibration(calib)) {
        std::fprintf(stderr, "engine set-up failed\n");
        return 1;
    }
    long poses = 0;
    engine.setPoseCallback([&poses](const Pose6DoF&) { poses++; });
    engine.setFrontEndStatsCallback([](const FrontEndFrameStats& f) {
        if (f.frameIndex % 30 == 0)  /* the cadence of the reference's own print-out */
            std::printf("frame %llu: %s, %d tracked + %d new = %d features, %.0f us\n", (unsigned long long)f.frameIndex,
                        f.ok ? "ok" : "front-end error", f.tracked, f.detected, f.total, f.latencyUs);
    });

    StreamIngest ingest(engine);
    FakeReceiver receiver;
    /* === the two callback bodies of pc_client/src/main.cpp:182-205 ================================================ */
    receiver.setVideoCallback([&ingest](const ImageView& frame, const std::string& metadataJson) { ingest.onFrame(frame, metadataJson); });
    receiver.setIMUCallback([&ingest](const uint8_t* datagram, std::size_t size) { ingest.onImuDatagram(datagram, size); });
    /* =============================================================================================================== */
    receiver.run(frames, 640, 480);

    const StreamIngest::Counters& c = ingest.counters();
    const FrontEndStats& st = engine.frontEndStats();
    std::printf("fed %lld IMU samples and %lld frames (%lld frames dropped before the engine)\n", (long long)c.imuSamples,
                (long long)c.frames, (long long)(c.framesBadMetadata + c.framesSizeMismatch + c.framesOutOfOrder));
    std::printf("front-end: %llu frames (%llu failed), latency avg %.0f us, P99 %.0f us, %.1f tracked / %.1f new per frame\n",
                (unsigned long long)st.frames(), (unsigned long long)st.failedFrames(), st.getAverageLatencyUs(), st.getP99LatencyUs(),
                st.averageTracked(), st.averageDetected());
    std::printf("%s\n", st.toJSON().c_str());
    if (st.failedFrames()) std::printf("front-end error: %s\n", engine.getFrontEndError().c_str());
    engine.shutdown();
    return 0;
}
