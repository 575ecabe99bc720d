/* Flat C view of SLAMEngine for non-C++ harnesses (the Python streaming harness and tests). */
#include "slam/slam_engine.hpp"
#include "slam/batch_frontend.hpp"
#include "slam/sharded_matcher.hpp"
#include "slam/adapters/openvins_adapter.hpp"
#include "slam/adapters/basalt_adapter.hpp"
#include "slam/adapters/orbslam3_adapter.hpp"
#include "slam/adapters/vins_mono_adapter.hpp"
#include <cstring>

using namespace vi_slam;

namespace {
struct Handle {
    SLAMEngine engine;
    long poseCalls = 0, statusCalls = 0, statsCalls = 0;
    std::string json;
    int lastStatus = 0;
    std::string err;
};
}  // namespace

extern "C" {

void* vis_engine_create() {
    Handle* h = new Handle;
    h->engine.setPoseCallback([h](const Pose6DoF&) { h->poseCalls++; });
    h->engine.setStatusCallback([h](TrackingStatus s) { h->statusCalls++; h->lastStatus = (int)s; });
    h->engine.setFrontEndStatsCallback([h](const FrontEndFrameStats&) { h->statsCalls++; });
    return h;
}
void vis_engine_destroy(void* p) { delete static_cast<Handle*>(p); }
int vis_engine_select(void* p, int type) {
    if (type < 0 || type > 3) return 0;
    return static_cast<Handle*>(p)->engine.selectFramework((SLAMFrameworkType)type) ? 1 : 0;
}
int vis_engine_current(void* p) { return (int)static_cast<Handle*>(p)->engine.getCurrentFramework(); }
int vis_engine_initialize(void* p, const char* cfg) { return static_cast<Handle*>(p)->engine.initialize(cfg) ? 1 : 0; }
int vis_engine_load_calibration(void* p, const char* calib) { return static_cast<Handle*>(p)->engine.loadCalibration(calib) ? 1 : 0; }
void vis_engine_process_image(void* p, const uint8_t* data, int rows, int cols, size_t step, int channels, int64_t ts) {
    static_cast<Handle*>(p)->engine.processImage(ImageView(data, rows, cols, step, channels), ts);
}
void vis_engine_process_imu(void* p, int64_t ts, const double* acc, const double* gyro) {
    IMUSample s;
    s.timestampNs = ts;
    std::memcpy(s.acceleration, acc, sizeof(s.acceleration));
    std::memcpy(s.angularVelocity, gyro, sizeof(s.angularVelocity));
    static_cast<Handle*>(p)->engine.processIMU(s);
}
int vis_engine_status(void* p) { return (int)static_cast<Handle*>(p)->engine.getStatus(); }
/* out: timestamp (as double), position xyz, orientation wxyz; returns 1 if a valid pose exists */
int vis_engine_pose(void* p, double* out8) {
    Pose6DoF pose;
    const bool ok = static_cast<Handle*>(p)->engine.getPose(pose);
    out8[0] = (double)pose.timestampNs;
    std::memcpy(out8 + 1, pose.position, 3 * sizeof(double));
    std::memcpy(out8 + 4, pose.orientation, 4 * sizeof(double));
    return ok ? 1 : 0;
}
long vis_engine_stats_callbacks(void* p) { return static_cast<Handle*>(p)->statsCalls; }
void vis_engine_enable_stage_timing(void* p, int on) { static_cast<Handle*>(p)->engine.enableStageTiming(on != 0); }
/* out9: frames, failed frames, average latency us, P99 latency us, last frame's ok / tracked / detected / total / latency us;
 * returns 1 once a frame has been recorded */
int vis_engine_stats(void* p, double* out9) {
    const FrontEndStats& st = static_cast<Handle*>(p)->engine.frontEndStats();
    FrontEndFrameStats f;
    const bool have = st.last(f);
    out9[0] = (double)st.frames(); out9[1] = (double)st.failedFrames();
    out9[2] = st.getAverageLatencyUs(); out9[3] = st.getP99LatencyUs();
    out9[4] = f.ok ? 1.0 : 0.0; out9[5] = f.tracked; out9[6] = f.detected; out9[7] = f.total; out9[8] = f.latencyUs;
    return have ? 1 : 0;
}
/* GPU time per stage of the last frame in microseconds (`out`: room for 64 slots, one per name of vfe_prof_kernel_name; zeros unless stage timing is on); returns 1 if present */
int vis_engine_stage_us(void* p, double* out) {
    FrontEndFrameStats f;
    const bool have = static_cast<Handle*>(p)->engine.frontEndStats().last(f);
    for (int i = 0; i < kFrontEndStages; i++) out[i] = have ? f.stageUs[i] : 0.0;
    return have && f.haveStages ? 1 : 0;
}
const char* vis_engine_stats_json(void* p) {
    Handle* h = static_cast<Handle*>(p);
    h->json = h->engine.frontEndStats().toJSON();
    return h->json.c_str();
}
void vis_engine_reset_latency_stats(void* p) { static_cast<Handle*>(p)->engine.frontEndStats().resetLatencyStats(); }
int vis_engine_tracked_count(void* p) { return static_cast<Handle*>(p)->engine.getTrackedFeatureCount(); }
long vis_engine_pose_callbacks(void* p) { return static_cast<Handle*>(p)->poseCalls; }
long vis_engine_status_callbacks(void* p) { return static_cast<Handle*>(p)->statusCalls; }
int vis_engine_last_status(void* p) { return static_cast<Handle*>(p)->lastStatus; }
void vis_engine_reset(void* p) { static_cast<Handle*>(p)->engine.reset(); }
void vis_engine_shutdown(void* p) { static_cast<Handle*>(p)->engine.shutdown(); }
const char* vis_engine_frontend_error(void* p) {
    Handle* h = static_cast<Handle*>(p);
    h->err = h->engine.getFrontEndError();
    return h->err.c_str();
}
/* feature positions of the last frame (x, y pairs): corners / tracked points / ORB keypoints */
int vis_engine_points(void* p, float* xy, int cap) {
    Handle* h = static_cast<Handle*>(p);
    ISLAMFramework* f = h->engine.framework();
    std::vector<float> v;
    if (auto* a = dynamic_cast<OpenVINSAdapter*>(f)) v = a->getCorners();
    else if (auto* b = dynamic_cast<BasaltAdapter*>(f)) v = b->getCorners();
    else if (auto* c = dynamic_cast<VINSMonoAdapter*>(f)) v = c->getTrackedPoints();
    else if (auto* d = dynamic_cast<ORBSLAM3Adapter*>(f))
        for (const auto& k : d->getKeypoints()) { v.push_back(k.x); v.push_back(k.y); }
    int n = (int)(v.size() / 2);
    if (n > cap) n = cap;
    std::memcpy(xy, v.data(), (size_t)n * 2 * sizeof(float));
    return n;
}
/* selected configuration values, for parser tests: returns 1 if `key` is known */
int vis_engine_config_value(void* p, const char* key, double* out) {
    ISLAMFramework* f = static_cast<Handle*>(p)->engine.framework();
    const std::string k = key;
    if (auto* a = dynamic_cast<OpenVINSAdapter*>(f)) {
        const OpenVINSConfig c = a->getConfig();
        if (k == "numFeatures") *out = c.numFeatures; else if (k == "minPxDist") *out = c.minPxDist;
        else if (k == "fastThreshold") *out = c.fastThreshold; else if (k == "gridX") *out = c.gridX;
        else if (k == "gridY") *out = c.gridY; else if (k == "useKLT") *out = c.useKLT;
        else if (k == "imuInitWindow") *out = c.imuInitWindow; else if (k == "numKltLevels") *out = c.numKltLevels;
        else if (k == "trackFrontEnd") *out = c.trackFrontEnd;
        else return 0;
        return 1;
    }
    if (auto* b = dynamic_cast<BasaltAdapter*>(f)) {
        const BasaltConfig c = b->getConfig();
        if (k == "maxFlowPoints") *out = c.maxFlowPoints; else if (k == "flowQualityThreshold") *out = c.flowQualityThreshold;
        else if (k == "pyramidLevels") *out = c.pyramidLevels; else if (k == "patchSize") *out = c.patchSize;
        else if (k == "minDistance") *out = c.minDistance; else if (k == "imuInitWindow") *out = c.imuInitWindow;
        else if (k == "gravity") *out = c.gravity; else if (k == "maxIterations") *out = c.maxIterations;
        else return 0;
        return 1;
    }
    if (auto* o = dynamic_cast<ORBSLAM3Adapter*>(f)) {
        const ORBSLAM3Config c = o->getConfig();
        if (k == "nFeatures") *out = c.nFeatures; else if (k == "scaleFactor") *out = c.scaleFactor;
        else if (k == "nLevels") *out = c.nLevels; else if (k == "iniThFAST") *out = c.iniThFAST;
        else if (k == "minThFAST") *out = c.minThFAST; else if (k == "fx") *out = c.fx;
        else if (k == "imageWidth") *out = c.imageWidth;
        else return 0;
        return 1;
    }
    if (auto* v = dynamic_cast<VINSMonoAdapter*>(f)) {
        const VINSMonoConfig c = v->getConfig();
        if (k == "maxCnt") *out = c.maxCnt; else if (k == "minDist") *out = c.minDist;
        else if (k == "equalize") *out = c.equalize; else if (k == "imageWidth") *out = c.imageWidth;
        else if (k == "imageHeight") *out = c.imageHeight;
        else return 0;
        return 1;
    }
    return 0;
}

/* ---- BatchFrontEnd: many streams per call ------------------------------------------------------------------------- */
namespace {
struct BatchHandle {
    BatchFrontEnd fe;
    std::vector<BatchStreamResult> res;
    std::vector<ImageView> views;
    std::string err;
};
}  // namespace

/* devices: `ndev` HIP ordinals (one shard of streams each), or ndev = 0 for device 0 */
void* vis_batch_open(int width, int height, int streams, int max_features, int levels, int win, double quality, double min_distance,
                     const int* devices, int ndev, int copy_threads) {
    BatchHandle* h = new BatchHandle;
    BatchFrontEndConfig c;
    c.width = width; c.height = height; c.streams = streams; c.maxFeatures = max_features; c.pyramidLevels = levels; c.winSize = win;
    c.quality = quality; c.minDistance = min_distance; c.copyThreads = copy_threads;
    for (int i = 0; i < ndev; i++) c.devices.push_back(devices[i]);
    (void)h->fe.open(c); /* a failed open is reported by vis_batch_is_open / vis_batch_error */
    return h;
}
void vis_batch_close(void* p) { delete static_cast<BatchHandle*>(p); }
int vis_batch_is_open(void* p) { return static_cast<BatchHandle*>(p)->fe.isOpen() ? 1 : 0; }
int vis_batch_shards(void* p) { return static_cast<BatchHandle*>(p)->fe.shards(); }
const char* vis_batch_error(void* p) {
    BatchHandle* h = static_cast<BatchHandle*>(p);
    h->err = h->fe.lastError();
    return h->err.c_str();
}
/* frames: `count` pointers to 8-bit gray images of the configured size, `step` bytes per row each */
int vis_batch_submit(void* p, const uint8_t* const* frames, int count, size_t step, int64_t ts) {
    BatchHandle* h = static_cast<BatchHandle*>(p);
    const BatchFrontEndConfig& c = h->fe.config();
    h->views.clear();
    for (int i = 0; i < count; i++) h->views.emplace_back(frames[i], c.height, c.width, step, 1);
    return h->fe.submit(h->views.data(), count, ts);
}
uint8_t* vis_batch_staging(void* p, int stream) { return static_cast<BatchHandle*>(p)->fe.staging(stream); }
size_t vis_batch_staging_pitch(void* p) { return static_cast<BatchHandle*>(p)->fe.stagingPitch(); }
int vis_batch_submit_staged(void* p, int64_t ts) { return static_cast<BatchHandle*>(p)->fe.submitStaged(ts); }
/* totals / tracked: one int per stream (may be NULL); returns 1 when the ticket's results are in */
int vis_batch_wait(void* p, int ticket, int* totals, int* tracked) {
    BatchHandle* h = static_cast<BatchHandle*>(p);
    if (!h->fe.wait(ticket, h->res)) return 0;
    for (size_t i = 0; i < h->res.size(); i++) {
        if (totals) totals[i] = h->res[i].total;
        if (tracked) tracked[i] = h->res[i].tracked;
    }
    return 1;
}
/* points of `stream` from the last successful vis_batch_wait: copies up to cap (x, y) pairs, returns the count */
int vis_batch_points(void* p, int stream, float* xy, int cap) {
    BatchHandle* h = static_cast<BatchHandle*>(p);
    if (stream < 0 || (size_t)stream >= h->res.size()) return -1;
    int n = h->res[stream].total;
    if (n > cap) n = cap;
    std::memcpy(xy, h->res[stream].xy, (size_t)n * 2 * sizeof(float));
    return n;
}
int vis_batch_reset(void* p) { return static_cast<BatchHandle*>(p)->fe.reset() ? 1 : 0; }

/* ---- ShardedMatcher: map split over the devices, RCCL all-gather of the per-shard top-2 ------------------------------ */
namespace {
struct MatcherHandle { ShardedMatcher m; std::string err, coll; };
}  // namespace

void* vis_matcher_open(const int* devices, int ndev) {
    MatcherHandle* h = new MatcherHandle;
    std::vector<int> d(devices, devices + (ndev > 0 ? ndev : 0));
    (void)h->m.open(d);
    return h;
}
void vis_matcher_close(void* p) { delete static_cast<MatcherHandle*>(p); }
int vis_matcher_is_open(void* p) { return static_cast<MatcherHandle*>(p)->m.isOpen() ? 1 : 0; }
int vis_matcher_world(void* p) { return static_cast<MatcherHandle*>(p)->m.world(); }
const char* vis_matcher_error(void* p) {
    MatcherHandle* h = static_cast<MatcherHandle*>(p);
    h->err = h->m.lastError();
    return h->err.c_str();
}
const char* vis_matcher_collective(void* p) {
    MatcherHandle* h = static_cast<MatcherHandle*>(p);
    h->coll = h->m.collective();
    return h->coll.c_str();
}
int vis_matcher_set_map(void* p, const uint8_t* desc, long long n) { return static_cast<MatcherHandle*>(p)->m.setMap(desc, n) ? 1 : 0; }
int vis_matcher_match(void* p, const uint8_t* q, int nq, int* idx, int* dist, float ratio, uint8_t* good) {
    return static_cast<MatcherHandle*>(p)->m.match(q, nq, idx, dist, ratio, good) ? 1 : 0;
}
double vis_matcher_last_ms(void* p) { return static_cast<MatcherHandle*>(p)->m.lastMatchMs(); }

}  // extern "C"
