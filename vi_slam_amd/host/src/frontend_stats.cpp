#include "slam/output/frontend_stats.hpp"
#include "vfe.h"
#include <algorithm>
#include <cstdio>
#include <numeric>

namespace vi_slam {

static_assert(kFrontEndStages == VFE_K_COUNT, "stage table out of step with the GPU library's timing slots");

void FrontEndStats::record(const FrontEndFrameStats& f) {
    std::lock_guard<std::mutex> lock(mu_);
    if (latencySamples_.size() >= maxSamples_) { /* window of the most recent samples, oldest out first */
        latencySamples_.erase(latencySamples_.begin());
        trackedSamples_.erase(trackedSamples_.begin());
        detectedSamples_.erase(detectedSamples_.begin());
    }
    latencySamples_.push_back(f.latencyUs);
    trackedSamples_.push_back(f.tracked);
    detectedSamples_.push_back(f.detected);
    last_ = f;
    frames_++;
    if (!f.ok) failed_++;
}

double FrontEndStats::getAverageLatencyUs() const {
    std::lock_guard<std::mutex> lock(mu_);
    if (latencySamples_.empty()) return 0.0;
    return std::accumulate(latencySamples_.begin(), latencySamples_.end(), 0.0) / (double)latencySamples_.size();
}

double FrontEndStats::getP99LatencyUs() const {
    std::lock_guard<std::mutex> lock(mu_);
    if (latencySamples_.empty()) return 0.0;
    std::vector<double> sorted = latencySamples_;
    std::sort(sorted.begin(), sorted.end());
    std::size_t i = (std::size_t)((double)sorted.size() * 0.99);
    if (i >= sorted.size()) i = sorted.size() - 1;
    return sorted[i];
}

void FrontEndStats::resetLatencyStats() {
    std::lock_guard<std::mutex> lock(mu_);
    latencySamples_.clear(); trackedSamples_.clear(); detectedSamples_.clear();
}

uint64_t FrontEndStats::frames() const { std::lock_guard<std::mutex> lock(mu_); return frames_; }
uint64_t FrontEndStats::failedFrames() const { std::lock_guard<std::mutex> lock(mu_); return failed_; }

bool FrontEndStats::last(FrontEndFrameStats& out) const {
    std::lock_guard<std::mutex> lock(mu_);
    if (frames_ == 0) return false;
    out = last_;
    return true;
}

static double mean_of(const std::vector<int>& v) {
    if (v.empty()) return 0.0;
    return (double)std::accumulate(v.begin(), v.end(), (long long)0) / (double)v.size();
}
double FrontEndStats::averageTracked() const { std::lock_guard<std::mutex> lock(mu_); return mean_of(trackedSamples_); }
double FrontEndStats::averageDetected() const { std::lock_guard<std::mutex> lock(mu_); return mean_of(detectedSamples_); }

const char* FrontEndStats::stageName(int i) { return vfe_prof_kernel_name(i); }

static const char* status_name(TrackingStatus s) {
    switch (s) {
        case TrackingStatus::UNINITIALIZED: return "UNINITIALIZED";
        case TrackingStatus::INITIALIZING: return "INITIALIZING";
        case TrackingStatus::TRACKING: return "TRACKING";
        case TrackingStatus::LOST: return "LOST";
        case TrackingStatus::RELOCALIZATION: return "RELOCALIZATION";
    }
    return "UNKNOWN";
}

std::string FrontEndStats::toJSON() const {
    const double avg = getAverageLatencyUs(), p99 = getP99LatencyUs();
    std::lock_guard<std::mutex> lock(mu_);
    char buf[512];
    std::snprintf(buf, sizeof(buf),
                  "{\"timestamp\":%.9f,\"frontend\":{\"frame\":%llu,\"ok\":%s,\"tracked\":%d,\"detected\":%d,\"total\":%d,"
                  "\"latency_us\":%.3f,\"avg_latency_us\":%.3f,\"p99_latency_us\":%.3f,\"status\":\"%s\"",
                  (double)last_.timestampNs * 1e-9, (unsigned long long)last_.frameIndex, last_.ok ? "true" : "false", last_.tracked,
                  last_.detected, last_.total, last_.latencyUs, avg, p99, status_name(last_.status));
    std::string s = buf;
    if (last_.haveStages) {
        s += ",\"stages_us\":{";
        bool first = true;
        for (int i = 0; i < kFrontEndStages; i++) {
            if (last_.stageUs[i] <= 0.0) continue;
            std::snprintf(buf, sizeof(buf), "%s\"%s\":%.3f", first ? "" : ",", stageName(i), last_.stageUs[i]);
            s += buf;
            first = false;
        }
        s += "}";
    }
    s += "}}";
    return s;
}

}  // namespace vi_slam
