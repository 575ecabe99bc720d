#ifndef VI_SLAM_AMD_FRONTEND_STATS_HPP
#define VI_SLAM_AMD_FRONTEND_STATS_HPP
/* Egress side of the front-end: what happened to each frame, for observers.
 *
 * The reference exposes a single number per adapter (getTrackedFeatureCount,
 * include/slam/adapters/openvins_adapter.hpp:105) and keeps latency statistics only in its ZMQ pose publisher
 * (getAverageLatencyUs / getP99LatencyUs / resetLatencyStats over a window of the most recent samples,
 * include/slam/output/zmq_publisher.hpp:85-118, src/slam/output/zmq_publisher.cpp:140-176).  FrontEndStats keeps
 * the same three calls with the same arithmetic (mean of the window; P99 = sorted[size * 0.99], clipped) for the
 * per-frame front-end latency, next to per-frame feature counts and, on request, the time each GPU stage took.
 * No ZMQ or ROS dependency: toJSON() gives the line a publisher would send. */
#include "common/types.hpp"
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

namespace vi_slam {

constexpr int kFrontEndStages = 12; /* the kernel-timing slots of the GPU library (vfe_prof_kernel_name) */

struct FrontEndFrameStats {
    int64_t timestampNs = 0;   /* sensor timestamp handed to processImage */
    uint64_t frameIndex = 0;   /* frames that reached the front-end since the last reset, this one included */
    bool ok = false;           /* false: the front-end call failed (see getFrontEndError) or never ran */
    int tracked = 0;           /* features carried over from the previous frame */
    int detected = 0;          /* features found new in this frame */
    int total = 0;             /* features alive after this frame */
    double latencyUs = 0.0;    /* wall time of the front-end call (upload, kernels, read-back) */
    TrackingStatus status = TrackingStatus::UNINITIALIZED;
    /* GPU time per stage in microseconds, only when stage timing is on (SLAMEngine::enableStageTiming) */
    bool haveStages = false;
    double stageUs[kFrontEndStages] = {0};
};

using FrontEndStatsCallback = std::function<void(const FrontEndFrameStats&)>;

class FrontEndStats {
public:
    explicit FrontEndStats(std::size_t maxSamples = 1000) : maxSamples_(maxSamples ? maxSamples : 1) {}

    void record(const FrontEndFrameStats& f);
    /* the reference's three calls, same arithmetic */
    double getAverageLatencyUs() const;
    double getP99LatencyUs() const;
    void resetLatencyStats();

    uint64_t frames() const;
    uint64_t failedFrames() const;
    bool last(FrontEndFrameStats& out) const; /* false before the first frame */
    double averageTracked() const;            /* over the same window as the latencies */
    double averageDetected() const;
    /* {"timestamp": seconds, "frontend": {"frame": n, "ok": true, "tracked": .., "detected": .., "total": ..,
     *  "latency_us": .., "avg_latency_us": .., "p99_latency_us": .., "status": "TRACKING", "stages_us": {...}}} */
    std::string toJSON() const;
    static const char* stageName(int i);

private:
    mutable std::mutex mu_;
    std::size_t maxSamples_;
    std::vector<double> latencySamples_;
    std::vector<int> trackedSamples_, detectedSamples_;
    FrontEndFrameStats last_;
    uint64_t frames_ = 0, failed_ = 0;
};

}  // namespace vi_slam
#endif
