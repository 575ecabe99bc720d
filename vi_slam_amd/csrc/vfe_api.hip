/* Status / error plumbing of libvfe. */
#include "vfe_common.h"
#include <map>
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void vfe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vfe_last_error(void) { return g_err; }

extern "C" const char* vfe_status_string(int s) {
    switch (s) {
        case VFE_OK: return "ok";
        case VFE_E_INVALID: return "invalid argument";
        case VFE_E_HIP: return "HIP runtime error";
        case VFE_E_NOMEM: return "out of memory / workspace too small";
        case VFE_E_CAPACITY: return "buffer capacity exceeded";
        case VFE_E_NODEVICE: return "no usable gfx950 device";
        default: return "unknown status";
    }
}

extern "C" const char* vfe_version(void) { return "vfe 0.1 (gfx950)"; }

extern "C" int vfe_device_count(int* count) {
    VFE_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        vfe_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return VFE_E_NODEVICE;
    }
    *count = n;
    return VFE_OK;
}

/* ---- kernel timing --------------------------------------------------------------------- */
std::atomic<int> g_vfe_prof_on{0};
thread_local const VfeCallOpts* t_vfe_call = nullptr;
namespace {
struct ProfRec { int id, dev; hipEvent_t a, b; const void* owner; }; /* owner: the context that asked for its own timing, or null */
std::mutex g_prof_mu;                        /* guards the finished list, the event pool and the totals */
/* begin recorded, end pending: scopes nest and close in order within a host thread, so the open list is per thread and
 * threads (or devices driven by different threads) cannot close each other's records */
thread_local std::vector<ProfRec> g_prof_open;
std::vector<ProfRec> g_prof_done;
std::vector<std::vector<hipEvent_t>> g_prof_pool; /* spare events per device: an event belongs to the device it was created on */
struct ProfTotals { double ms[VFE_K_COUNT]; long long n[VFE_K_COUNT]; };
std::map<const void*, ProfTotals> g_prof_totals; /* key null: the process-wide totals */
const size_t PROF_MAX_PENDING = 1u << 16;    /* nobody reading: stop recording rather than grow without bound */
hipEvent_t prof_event(int dev) {
    if ((size_t)dev < g_prof_pool.size() && !g_prof_pool[dev].empty()) {
        hipEvent_t e = g_prof_pool[dev].back(); g_prof_pool[dev].pop_back(); return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void prof_recycle(int dev, hipEvent_t e) {
    if (!e || dev < 0) return;
    if ((size_t)dev >= g_prof_pool.size()) g_prof_pool.resize(dev + 1);
    g_prof_pool[dev].push_back(e);
}
}  // namespace

bool vfe_prof_begin(int id, hipStream_t st) {
    ProfRec r; r.id = id; r.dev = 0;
    r.owner = (t_vfe_call && t_vfe_call->prof > 0) ? t_vfe_call->owner : nullptr;
    if (hipGetDevice(&r.dev) != hipSuccess) return false;
    {
        std::lock_guard<std::mutex> g(g_prof_mu);
        if (g_prof_done.size() >= PROF_MAX_PENDING) return false;
        r.a = prof_event(r.dev); r.b = prof_event(r.dev);
    }
    if (!r.a || !r.b || hipEventRecord(r.a, st) != hipSuccess) {
        std::lock_guard<std::mutex> g(g_prof_mu);
        prof_recycle(r.dev, r.a); prof_recycle(r.dev, r.b);
        return false;
    }
    g_prof_open.push_back(r);
    return true;
}
void vfe_prof_end(int id, hipStream_t st) {
    for (size_t i = g_prof_open.size(); i-- > 0;)
        if (g_prof_open[i].id == id) {
            ProfRec r = g_prof_open[i];
            g_prof_open.erase(g_prof_open.begin() + i);
            const bool ok = hipEventRecord(r.b, st) == hipSuccess;
            std::lock_guard<std::mutex> g(g_prof_mu);
            if (ok) g_prof_done.push_back(r);
            else { prof_recycle(r.dev, r.a); prof_recycle(r.dev, r.b); }
            return;
        }
}

extern "C" int vfe_prof_enable(int enable) { g_vfe_prof_on.store(enable ? 1 : 0); return VFE_OK; }

/* finished records of `owner` into its totals (the events' device must be current for nothing here: elapsed-time queries
 * work from any device), then the totals out */
int vfe_prof_read_owner(const void* owner, double* ms, long long* launches, int reset) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    ProfTotals& tot = g_prof_totals[owner];
    size_t kept = 0;
    for (auto& r : g_prof_done) {
        if (r.owner != owner) { g_prof_done[kept++] = r; continue; }
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess &&
            r.id >= 0 && r.id < VFE_K_COUNT) {
            tot.ms[r.id] += t; tot.n[r.id] += 1;
        }
        prof_recycle(r.dev, r.a); prof_recycle(r.dev, r.b);
    }
    g_prof_done.resize(kept);
    for (int i = 0; i < VFE_K_COUNT; i++) {
        if (ms) ms[i] = tot.ms[i];
        if (launches) launches[i] = tot.n[i];
        if (reset) { tot.ms[i] = 0; tot.n[i] = 0; }
    }
    return VFE_OK;
}

void vfe_prof_forget_owner(const void* owner) {
    (void)vfe_prof_read_owner(owner, nullptr, nullptr, 1);
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof_totals.erase(owner);
}

extern "C" int vfe_prof_read(double* ms, long long* launches, int reset) { return vfe_prof_read_owner(nullptr, ms, launches, reset); }

extern "C" const char* vfe_prof_kernel_name(int id) {
    static const char* names[VFE_K_COUNT] = {"k_pyr_down", "k_gftt_eig", "k_gftt_select", "k_klt", "k_compact",
                                             "k_append", "k_fast", "k_orb_resize", "k_orb_select",
                                             "k_orb_desc", "k_match", "misc"};
    return (id >= 0 && id < VFE_K_COUNT) ? names[id] : "?";
}
