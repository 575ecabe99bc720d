#include "slam/batch_frontend.hpp"
#include "common/logging.hpp"
#include "vfe.h"
#include <hip/hip_runtime_api.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

namespace vi_slam {

namespace {

constexpr int kSlots = 3;      /* staging / frame / result slots per stream: set i + 1 uploads while set i is stepped */
constexpr int kMaxLevels = 8;

size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* VSA_BATCH_TRACE=1 in the environment: one line per stage on stderr (where does a stalled pipeline stand) */
bool traceOn() {
    static const bool on = [] { const char* e = std::getenv("VSA_BATCH_TRACE"); return e && *e && *e != '0'; }();
    return on;
}
#define BF_TRACE(...)                                                     \
    do {                                                                  \
        if (traceOn()) { std::fprintf(stderr, "[batch] " __VA_ARGS__); std::fputc('\n', stderr); std::fflush(stderr); } \
    } while (0)

/* n persistent threads; run(count, fn) calls fn(i) for every i in [0, count) on them and returns when all are done */
class CopyPool {
public:
    explicit CopyPool(int n) {
        for (int i = 0; i < n; i++) threads_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> l(mu_); quit_ = true; }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void run(int count, const std::function<void(int)>& fn) {
        if (count <= 0) return;
        if (threads_.empty()) { for (int i = 0; i < count; i++) fn(i); return; }
        std::unique_lock<std::mutex> l(mu_);
        fn_ = &fn; next_ = 0; count_ = count; pending_ = count;
        cv_.notify_all();
        done_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr; count_ = 0; next_ = 0;
    }

private:
    void loop() {
        std::unique_lock<std::mutex> l(mu_);
        for (;;) {
            cv_.wait(l, [this] { return quit_ || next_ < count_; });
            if (quit_) return;
            const int i = next_++;
            const std::function<void(int)>* fn = fn_; /* stays valid until pending_ reaches 0 */
            l.unlock();
            (*fn)(i);
            l.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int next_ = 0, count_ = 0, pending_ = 0;
    bool quit_ = false;
};

/* One block of streams on one device: a host thread that makes every HIP call of the shard, its buffers and streams */
struct Shard {
    int dev = 0, s0 = 0, s1 = 0, nb = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::packaged_task<bool()>> jobs;
    bool quit = false;
    std::unique_ptr<CopyPool> pool;
    hipStream_t compute = nullptr, copy = nullptr;
    hipEvent_t upDone[kSlots] = {nullptr, nullptr, nullptr}, stepDone[kSlots] = {nullptr, nullptr, nullptr},
               resDone[kSlots] = {nullptr, nullptr, nullptr};
    uint8_t* frame[kSlots] = {nullptr, nullptr, nullptr};
    uint8_t* lev[2][kMaxLevels] = {{nullptr}, {nullptr}};
    float* pts = nullptr;
    int* counts = nullptr;      /* n_pts [nb] then n_tracked [nb] */
    void* ws = nullptr;
    size_t wsBytes = 0;
    uint8_t* stage[kSlots] = {nullptr, nullptr, nullptr};
    float* resPts[kSlots] = {nullptr, nullptr, nullptr};
    int* resN[kSlots] = {nullptr, nullptr, nullptr};
    std::string err;

    std::future<bool> post(std::function<bool()> fn) {
        std::packaged_task<bool()> task(std::move(fn));
        std::future<bool> f = task.get_future();
        { std::lock_guard<std::mutex> l(mu); jobs.push_back(std::move(task)); }
        cv.notify_one();
        return f;
    }
    void loop() {
        for (;;) {
            std::packaged_task<bool()> task;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [this] { return quit || !jobs.empty(); });
                if (jobs.empty()) return;
                task = std::move(jobs.front());
                jobs.pop_front();
            }
            task();
        }
    }
};

bool hipFail(Shard& s, hipError_t e, const char* what) {
    s.err = std::string(what) + " failed on device " + std::to_string(s.dev) + ": " + hipGetErrorString(e);
    return false;
}
bool vfeFail(Shard& s, int rc, const char* what) {
    s.err = std::string(what) + ": " + vfe_status_string(rc) + ": " + vfe_last_error();
    return false;
}
#define BF_HIP(s, call)                                         \
    do {                                                        \
        const hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return hipFail(s, e_, #call);     \
    } while (0)

}  // namespace

struct BatchFrontEnd::Impl {
    BatchFrontEndConfig cfg;
    std::vector<std::unique_ptr<Shard>> shards;
    std::vector<int> shardOf;   /* stream -> shard */
    int nlevels = 1;
    int lw[kMaxLevels], lh[kMaxLevels], lpitch[kMaxLevels];
    size_t lstride[kMaxLevels];
    size_t pitch0 = 0, frameBytes = 0;
    int nextTicket = 0;
    bool havePrev = false;
    bool opened = false;
    mutable std::mutex errMu;
    std::string error;

    void setError(const std::string& e) {
        std::lock_guard<std::mutex> l(errMu);
        error = e;
        LOG_ERROR("BatchFrontEnd", "%s", e.c_str());
    }
    vfe_dimg level(Shard& s, int slotFrame, int slotLev, int l) const {
        vfe_dimg d;
        d.width = lw[l]; d.height = lh[l]; d.batch = s.nb;
        if (l == 0) {
            d.data = s.frame[slotFrame]; d.pitch = (int)pitch0; d.frame_stride = frameBytes; d.border = 0;
        } else {
            d.pitch = lpitch[l]; d.frame_stride = lstride[l]; d.border = VFE_LK_BORDER;
            d.data = s.lev[slotLev][l] + (size_t)VFE_LK_BORDER * d.pitch + VFE_LK_BORDER;
        }
        return d;
    }

    bool shardOpen(Shard& s) {
        BF_HIP(s, hipSetDevice(s.dev));
        BF_HIP(s, hipStreamCreateWithFlags(&s.compute, hipStreamNonBlocking));
        BF_HIP(s, hipStreamCreateWithFlags(&s.copy, hipStreamNonBlocking));
        for (int i = 0; i < kSlots; i++) {
            BF_HIP(s, hipEventCreateWithFlags(&s.upDone[i], hipEventDisableTiming));
            BF_HIP(s, hipEventCreateWithFlags(&s.stepDone[i], hipEventDisableTiming));
            BF_HIP(s, hipEventCreateWithFlags(&s.resDone[i], hipEventDisableTiming));
            BF_HIP(s, hipMalloc((void**)&s.frame[i], frameBytes * s.nb));
            BF_HIP(s, hipHostMalloc((void**)&s.stage[i], frameBytes * s.nb, hipHostMallocDefault));
            BF_HIP(s, hipHostMalloc((void**)&s.resPts[i], (size_t)s.nb * cfg.maxFeatures * 2 * sizeof(float), hipHostMallocDefault));
            BF_HIP(s, hipHostMalloc((void**)&s.resN[i], (size_t)2 * s.nb * sizeof(int), hipHostMallocDefault));
        }
        for (int k = 0; k < 2; k++)
            for (int l = 1; l < nlevels; l++) BF_HIP(s, hipMalloc((void**)&s.lev[k][l], lstride[l] * s.nb));
        BF_HIP(s, hipMalloc((void**)&s.pts, (size_t)s.nb * cfg.maxFeatures * 2 * sizeof(float)));
        BF_HIP(s, hipMalloc((void**)&s.counts, (size_t)2 * s.nb * sizeof(int)));
        BF_HIP(s, hipMemsetAsync(s.counts, 0, (size_t)2 * s.nb * sizeof(int), s.compute));
        s.wsBytes = vfe_frontend_workspace(cfg.width, cfg.height, s.nb, cfg.maxFeatures);
        const size_t gf = vfe_good_features_workspace(cfg.width, cfg.height, s.nb, cfg.maxFeatures, cfg.maxFeatures);
        if (gf > s.wsBytes) s.wsBytes = gf;
        BF_HIP(s, hipMalloc(&s.ws, s.wsBytes));
        BF_HIP(s, hipStreamSynchronize(s.compute));
        return true;
    }

    bool shardClose(Shard& s) {
        (void)hipSetDevice(s.dev);
        if (s.compute) (void)hipStreamSynchronize(s.compute);
        if (s.copy) (void)hipStreamSynchronize(s.copy);
        for (int i = 0; i < kSlots; i++) {
            if (s.upDone[i]) (void)hipEventDestroy(s.upDone[i]);
            if (s.stepDone[i]) (void)hipEventDestroy(s.stepDone[i]);
            if (s.resDone[i]) (void)hipEventDestroy(s.resDone[i]);
            (void)hipFree(s.frame[i]);
            (void)hipHostFree(s.stage[i]); (void)hipHostFree(s.resPts[i]); (void)hipHostFree(s.resN[i]);
        }
        for (int k = 0; k < 2; k++)
            for (int l = 1; l < kMaxLevels; l++) (void)hipFree(s.lev[k][l]);
        (void)hipFree(s.pts); (void)hipFree(s.counts); (void)hipFree(s.ws);
        if (s.compute) (void)hipStreamDestroy(s.compute);
        if (s.copy) (void)hipStreamDestroy(s.copy);
        return true;
    }

    /* repack this shard's frames (caller memory, any step) into the staging slot */
    void shardRepack(Shard& s, int slot, const ImageView* frames) {
        const int rowsPerPiece = 135; /* 1080 rows = 8 pieces: enough pieces for the copy threads at any stream count */
        const int pieces = (cfg.height + rowsPerPiece - 1) / rowsPerPiece;
        const std::function<void(int)> fn = [&](int item) {
            const int b = item / pieces, r0 = (item - b * pieces) * rowsPerPiece;
            const int r1 = r0 + rowsPerPiece < cfg.height ? r0 + rowsPerPiece : cfg.height;
            const ImageView& im = frames[s.s0 + b];
            uint8_t* dst = s.stage[slot] + (size_t)b * frameBytes;
            if (im.step == pitch0) std::memcpy(dst + (size_t)r0 * pitch0, im.data + (size_t)r0 * im.step, (size_t)(r1 - r0) * pitch0);
            else for (int y = r0; y < r1; y++) std::memcpy(dst + (size_t)y * pitch0, im.data + (size_t)y * im.step, (size_t)cfg.width);
        };
        BF_TRACE("shard %d-%d: repack into slot %d", s.s0, s.s1, slot);
        s.pool->run(s.nb * pieces, fn);
        BF_TRACE("shard %d-%d: repack done", s.s0, s.s1);
    }

    /* upload of the staging slot + the step on it; everything is only enqueued */
    bool shardSubmit(Shard& s, int ticket, bool first) {
        const int slot = ticket % kSlots;
        BF_TRACE("shard %d-%d: submit ticket %d (first %d)", s.s0, s.s1, ticket, (int)first);
        BF_HIP(s, hipSetDevice(s.dev));
        /* the device frame of this slot was the previous image of the step two tickets ago */
        if (ticket >= 2) BF_HIP(s, hipStreamWaitEvent(s.copy, s.stepDone[(ticket - 2) % kSlots], 0));
        BF_HIP(s, hipMemcpyAsync(s.frame[slot], s.stage[slot], frameBytes * s.nb, hipMemcpyHostToDevice, s.copy));
        BF_HIP(s, hipEventRecord(s.upDone[slot], s.copy));
        BF_HIP(s, hipStreamWaitEvent(s.compute, s.upDone[slot], 0));
        vfe_dimg next[kMaxLevels], prev[kMaxLevels];
        for (int l = 0; l < nlevels; l++) next[l] = level(s, slot, ticket & 1, l);
        int* nPts = s.counts;
        int* nTracked = s.counts + s.nb;
        int rc;
        if (first) {
            if ((rc = vfe_dev_build_pyramid(next, nlevels, s.compute))) return vfeFail(s, rc, "vfe_dev_build_pyramid");
            if ((rc = vfe_dev_good_features(&next[0], cfg.maxFeatures, cfg.quality, cfg.minDistance, nullptr, nullptr, 0, s.pts,
                                            nullptr, nPts, s.ws, s.wsBytes, s.compute))) return vfeFail(s, rc, "vfe_dev_good_features");
            BF_HIP(s, hipMemsetAsync(nTracked, 0, (size_t)s.nb * sizeof(int), s.compute));
        } else {
            const int pslot = (ticket + kSlots - 1) % kSlots;
            for (int l = 0; l < nlevels; l++) prev[l] = level(s, pslot, (ticket & 1) ^ 1, l);
            if ((rc = vfe_dev_frontend_step(prev, next, nlevels, s.pts, nPts, nTracked, cfg.maxFeatures, cfg.quality, cfg.minDistance,
                                            cfg.winSize, 30, 0.01, 1e-4, s.ws, s.wsBytes, s.compute))) return vfeFail(s, rc, "vfe_dev_frontend_step");
        }
        BF_HIP(s, hipEventRecord(s.stepDone[slot], s.compute));
        BF_HIP(s, hipMemcpyAsync(s.resPts[slot], s.pts, (size_t)s.nb * cfg.maxFeatures * 2 * sizeof(float), hipMemcpyDeviceToHost, s.compute));
        BF_HIP(s, hipMemcpyAsync(s.resN[slot], s.counts, (size_t)2 * s.nb * sizeof(int), hipMemcpyDeviceToHost, s.compute));
        BF_HIP(s, hipEventRecord(s.resDone[slot], s.compute));
        BF_TRACE("shard %d-%d: ticket %d enqueued", s.s0, s.s1, ticket);
        return true;
    }

    /* run fn on every shard's thread at once; false (and the first shard's message) if any failed */
    bool onAll(const std::function<bool(Shard&)>& fn) {
        std::vector<std::future<bool>> f;
        for (auto& sp : shards) {
            Shard* s = sp.get();
            f.push_back(s->post([fn, s] { return fn(*s); }));
        }
        bool ok = true;
        for (size_t i = 0; i < f.size(); i++)
            if (!f[i].get()) {
                if (ok) setError(shards[i]->err);
                ok = false;
            }
        return ok;
    }
};

BatchFrontEnd::BatchFrontEnd() : impl_(new Impl) {}
BatchFrontEnd::~BatchFrontEnd() { close(); }

void BatchFrontEnd::shardRange(int n, int rank, int world, int& begin, int& end) {
    const int base = n / world, rem = n % world;
    begin = rank * base + (rank < rem ? rank : rem);
    end = begin + base + (rank < rem ? 1 : 0);
}

bool BatchFrontEnd::isOpen() const { return impl_->opened; }
const BatchFrontEndConfig& BatchFrontEnd::config() const { return impl_->cfg; }
int BatchFrontEnd::shards() const { return (int)impl_->shards.size(); }
size_t BatchFrontEnd::stagingPitch() const { return impl_->pitch0; }
std::string BatchFrontEnd::lastError() const {
    std::lock_guard<std::mutex> l(impl_->errMu);
    return impl_->error;
}

bool BatchFrontEnd::open(const BatchFrontEndConfig& cfg) {
    close();
    Impl& m = *impl_;
    m.cfg = cfg;
    if (m.cfg.devices.empty()) m.cfg.devices.push_back(0);
    if (cfg.width < 8 || cfg.height < 8 || cfg.width >= 32768 || cfg.height >= 32768 || cfg.streams <= 0 || cfg.maxFeatures <= 0 ||
        cfg.pyramidLevels < 1 || cfg.pyramidLevels > kMaxLevels || cfg.winSize < 3 || !(cfg.winSize & 1)) {
        m.setError("bad configuration (frame size, streams, maxFeatures, pyramidLevels 1..8, odd winSize)");
        return false;
    }
    int ndev = 0;
    if (vfe_device_count(&ndev) != VFE_OK || ndev <= 0) {
        m.setError(std::string("no usable GPU: ") + vfe_last_error() + " (the front-end has no CPU fallback)");
        return false;
    }
    for (int d : m.cfg.devices)
        if (d < 0 || d >= ndev) { m.setError("device ordinal " + std::to_string(d) + " out of range (" + std::to_string(ndev) + " visible)"); return false; }
    /* level geometry: level 0 rows packed to 16 bytes (only pixels cross the link), bordered levels on 256-byte rows */
    m.nlevels = vfe_lk_max_level(cfg.width, cfg.height, cfg.winSize, cfg.pyramidLevels - 1) + 1;
    m.pitch0 = alignUp((size_t)cfg.width, 16);
    m.frameBytes = m.pitch0 * cfg.height;
    int w = cfg.width, h = cfg.height;
    for (int l = 0; l < m.nlevels; l++) {
        m.lw[l] = w; m.lh[l] = h;
        m.lpitch[l] = (int)alignUp((size_t)w + 2 * VFE_LK_BORDER, 256);
        m.lstride[l] = (size_t)m.lpitch[l] * (h + 2 * VFE_LK_BORDER);
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    const int world = (int)m.cfg.devices.size();
    m.shardOf.assign(cfg.streams, 0);
    for (int r = 0; r < world; r++) {
        int b, e;
        shardRange(cfg.streams, r, world, b, e);
        if (e <= b) continue; /* more shards than streams */
        auto s = std::make_unique<Shard>();
        s->dev = m.cfg.devices[r]; s->s0 = b; s->s1 = e; s->nb = e - b;
        s->pool.reset(new CopyPool(cfg.copyThreads > 0 ? cfg.copyThreads : 0));
        for (int i = b; i < e; i++) m.shardOf[i] = (int)m.shards.size();
        Shard* raw = s.get();
        s->th = std::thread([raw] { raw->loop(); });
        m.shards.push_back(std::move(s));
    }
    m.opened = true;
    if (!m.onAll([&m](Shard& s) { return m.shardOpen(s); })) { close(); return false; }
    m.nextTicket = 0;
    m.havePrev = false;
    return true;
}

void BatchFrontEnd::close() {
    Impl& m = *impl_;
    if (!m.opened) return;
    (void)m.onAll([&m](Shard& s) { return m.shardClose(s); });
    for (auto& s : m.shards) {
        { std::lock_guard<std::mutex> l(s->mu); s->quit = true; }
        s->cv.notify_all();
        if (s->th.joinable()) s->th.join();
    }
    m.shards.clear();
    m.opened = false;
}

uint8_t* BatchFrontEnd::staging(int stream) {
    Impl& m = *impl_;
    if (!m.opened || stream < 0 || stream >= m.cfg.streams) return nullptr;
    Shard& s = *m.shards[m.shardOf[stream]];
    return s.stage[m.nextTicket % kSlots] + (size_t)(stream - s.s0) * m.frameBytes;
}

int BatchFrontEnd::submitStaged(int64_t timestampNs) {
    (void)timestampNs;
    Impl& m = *impl_;
    if (!m.opened) { m.setError("not open"); return -1; }
    const int ticket = m.nextTicket;
    const bool first = !m.havePrev;
    if (!m.onAll([&m, ticket, first](Shard& s) { return m.shardSubmit(s, ticket, first); })) return -1;
    m.nextTicket++;
    m.havePrev = true;
    /* the slot handed out by staging() next must not be overwritten while its last upload is still reading it */
    const int nslot = m.nextTicket % kSlots;
    if (m.nextTicket >= kSlots)
        (void)m.onAll([nslot](Shard& s) {
            const hipError_t e = hipEventSynchronize(s.upDone[nslot]);
            return e == hipSuccess ? true : hipFail(s, e, "hipEventSynchronize(upload)");
        });
    return ticket;
}

int BatchFrontEnd::submit(const ImageView* frames, int count, int64_t timestampNs) {
    Impl& m = *impl_;
    if (!m.opened) { m.setError("not open"); return -1; }
    if (!frames || count != m.cfg.streams) { m.setError("submit needs one frame per stream"); return -1; }
    for (int i = 0; i < count; i++)
        if (frames[i].empty() || frames[i].channels() != 1 || frames[i].cols != m.cfg.width || frames[i].rows != m.cfg.height ||
            frames[i].step < (size_t)m.cfg.width) {
            m.setError("frame " + std::to_string(i) + ": every stream delivers 8-bit gray frames of the configured size");
            return -1;
        }
    const int slot = m.nextTicket % kSlots;
    (void)m.onAll([&m, slot, frames](Shard& s) { m.shardRepack(s, slot, frames); return true; });
    return submitStaged(timestampNs);
}

bool BatchFrontEnd::wait(int ticket, std::vector<BatchStreamResult>& out) {
    Impl& m = *impl_;
    out.clear();
    if (!m.opened) { m.setError("not open"); return false; }
    if (ticket < 0 || ticket >= m.nextTicket || ticket < m.nextTicket - kSlots) {
        m.setError("ticket " + std::to_string(ticket) + " is not outstanding (results live for three submissions)");
        return false;
    }
    const int slot = ticket % kSlots;
    if (!m.onAll([slot, ticket](Shard& s) {
            BF_TRACE("shard %d-%d: wait for ticket %d", s.s0, s.s1, ticket);
            const hipError_t e = hipEventSynchronize(s.resDone[slot]);
            BF_TRACE("shard %d-%d: ticket %d done (%d)", s.s0, s.s1, ticket, (int)e);
            return e == hipSuccess ? true : hipFail(s, e, "hipEventSynchronize(results)");
        })) return false;
    out.resize(m.cfg.streams);
    for (int i = 0; i < m.cfg.streams; i++) {
        const Shard& s = *m.shards[m.shardOf[i]];
        const int b = i - s.s0;
        BatchStreamResult& r = out[i];
        r.total = s.resN[slot][b];
        r.tracked = s.resN[slot][s.nb + b];
        r.detected = r.total - r.tracked;
        r.xy = s.resPts[slot] + (size_t)b * m.cfg.maxFeatures * 2;
    }
    return true;
}

bool BatchFrontEnd::reset() {
    Impl& m = *impl_;
    if (!m.opened) return false;
    const bool ok = m.onAll([](Shard& s) {
        BF_HIP(s, hipSetDevice(s.dev));
        BF_HIP(s, hipStreamSynchronize(s.copy));
        BF_HIP(s, hipMemsetAsync(s.counts, 0, (size_t)2 * s.nb * sizeof(int), s.compute));
        BF_HIP(s, hipStreamSynchronize(s.compute));
        return true;
    });
    m.havePrev = false;
    return ok;
}

}  // namespace vi_slam
