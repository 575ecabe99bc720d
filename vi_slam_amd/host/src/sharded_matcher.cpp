#include "slam/sharded_matcher.hpp"
#include "common/logging.hpp"
#include "vfe.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <cstring>

namespace vi_slam {

namespace {

/* the RCCL entry points this file uses, resolved from the library at run time */
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;

    bool load(std::string& err) {
        if (handle) return true;
        /* private to this library and bound to its own symbols first: a process may already hold another RCCL build (a Python
         * process with torch brings its own), and the two must not resolve into each other */
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND))) break;
        if (!handle) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(handle, n); if (!p) { err = std::string("RCCL lacks ") + n; ok = false; } return p; };
        GetVersion = reinterpret_cast<decltype(GetVersion)>(sym("ncclGetVersion"));
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return ok;
    }
};

struct Rank {
    int dev = 0;
    long long row0 = 0, rows = 0;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    uint8_t* map = nullptr;      /* this rank's rows */
    uint8_t* query = nullptr;    /* replicated query set */
    int *idx = nullptr, *dist = nullptr;         /* this shard's top-2 [nq][2] */
    int *allIdx = nullptr, *allDist = nullptr;   /* gathered [world][nq][2] */
    int *outIdx = nullptr, *outDist = nullptr;   /* merged [nq][2] */
    uint8_t* good = nullptr;
    void* ws = nullptr;
    size_t wsBytes = 0;
    int qcap = 0;
};

size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct ShardedMatcher::Impl {
    Rccl rccl;
    std::vector<Rank> ranks;
    long long n = 0;
    bool opened = false;
    int version = 0;
    std::string error;
    double lastMs = 0;
    /* pinned host buffers for queries and results */
    uint8_t* hq = nullptr; int* hres = nullptr; uint8_t* hgood = nullptr; int hcap = 0;

    bool fail(const std::string& e) { error = e; LOG_ERROR("ShardedMatcher", "%s", e.c_str()); return false; }
    bool hipOk(hipError_t e, const char* what) { return e == hipSuccess ? true : fail(std::string(what) + " failed: " + hipGetErrorString(e)); }
    bool ncclOk(ncclResult_t r, const char* what) { return r == ncclSuccess ? true : fail(std::string(what) + " failed: " + rccl.GetErrorString(r)); }
    bool vfeOk(int rc, const char* what) { return rc == VFE_OK ? true : fail(std::string(what) + ": " + vfe_status_string(rc) + ": " + vfe_last_error()); }
#define SM_HIP(call) do { if (!hipOk((call), #call)) return false; } while (0)
#define SMM_HIP(call) do { if (!m.hipOk((call), #call)) return false; } while (0)

    void freeRank(Rank& r) {
        (void)hipSetDevice(r.dev);
        if (r.stream) (void)hipStreamSynchronize(r.stream);
        void* bufs[] = {r.map, r.query, r.idx, r.dist, r.allIdx, r.allDist, r.outIdx, r.outDist, r.good, r.ws};
        for (void* p : bufs) (void)hipFree(p);
        r.map = r.query = r.good = nullptr; r.idx = r.dist = r.allIdx = r.allDist = r.outIdx = r.outDist = nullptr; r.ws = nullptr;
        r.qcap = 0; r.wsBytes = 0;
    }

    /* per-rank buffers for nq queries (grown on demand) */
    bool reserve(int nq) {
        const int world = (int)ranks.size();
        for (Rank& r : ranks) {
            SM_HIP(hipSetDevice(r.dev));
            const size_t need = vfe_match_workspace(nq, r.rows);
            if (nq > r.qcap) {
                void* bufs[] = {r.query, r.idx, r.dist, r.allIdx, r.allDist, r.outIdx, r.outDist, r.good};
                SM_HIP(hipStreamSynchronize(r.stream));
                for (void* p : bufs) (void)hipFree(p);
                const size_t l = (size_t)nq * 2 * sizeof(int);
                SM_HIP(hipMalloc((void**)&r.query, (size_t)nq * 32));
                SM_HIP(hipMalloc((void**)&r.idx, l)); SM_HIP(hipMalloc((void**)&r.dist, l));
                SM_HIP(hipMalloc((void**)&r.allIdx, l * world)); SM_HIP(hipMalloc((void**)&r.allDist, l * world));
                SM_HIP(hipMalloc((void**)&r.outIdx, l)); SM_HIP(hipMalloc((void**)&r.outDist, l));
                SM_HIP(hipMalloc((void**)&r.good, (size_t)nq));
                r.qcap = nq;
            }
            if (need > r.wsBytes) {
                SM_HIP(hipStreamSynchronize(r.stream));
                (void)hipFree(r.ws); r.ws = nullptr; r.wsBytes = 0;
                SM_HIP(hipMalloc(&r.ws, alignUp(need, 256)));
                r.wsBytes = alignUp(need, 256);
            }
        }
        if (nq > hcap) {
            (void)hipHostFree(hq); (void)hipHostFree(hres); (void)hipHostFree(hgood);
            hq = nullptr; hres = nullptr; hgood = nullptr; hcap = 0;
            SM_HIP(hipHostMalloc((void**)&hq, (size_t)nq * 32, hipHostMallocDefault));
            SM_HIP(hipHostMalloc((void**)&hres, (size_t)nq * 4 * sizeof(int), hipHostMallocDefault));
            SM_HIP(hipHostMalloc((void**)&hgood, (size_t)nq, hipHostMallocDefault));
            hcap = nq;
        }
        return true;
    }
};

ShardedMatcher::ShardedMatcher() : impl_(new Impl) {}
ShardedMatcher::~ShardedMatcher() { close(); }
bool ShardedMatcher::isOpen() const { return impl_->opened; }
int ShardedMatcher::world() const { return (int)impl_->ranks.size(); }
long long ShardedMatcher::mapRows() const { return impl_->n; }
std::string ShardedMatcher::lastError() const { return impl_->error; }
double ShardedMatcher::lastMatchMs() const { return impl_->lastMs; }

std::string ShardedMatcher::collective() const {
    if (!impl_->opened) return "none";
    const int v = impl_->version;
    return "ncclAllGather x2 grouped (RCCL " + std::to_string(v / 10000) + "." + std::to_string(v / 100 % 100) + "." + std::to_string(v % 100) +
           ", " + std::to_string(impl_->ranks.size()) + (impl_->ranks.size() == 1 ? " rank)" : " ranks)");
}

bool ShardedMatcher::open(const std::vector<int>& devicesIn) {
    close();
    Impl& m = *impl_;
    std::vector<int> devices = devicesIn;
    if (devices.empty()) devices.push_back(0);
    int ndev = 0;
    if (vfe_device_count(&ndev) != VFE_OK || ndev <= 0) return m.fail(std::string("no usable GPU: ") + vfe_last_error() + " (no CPU fallback)");
    for (size_t i = 0; i < devices.size(); i++) {
        if (devices[i] < 0 || devices[i] >= ndev) return m.fail("device ordinal " + std::to_string(devices[i]) + " out of range");
        for (size_t j = 0; j < i; j++)
            if (devices[j] == devices[i]) return m.fail("a device can hold only one communicator rank: ordinal " + std::to_string(devices[i]) + " listed twice");
    }
    std::string err;
    if (!m.rccl.load(err)) return m.fail(err);
    if (!m.ncclOk(m.rccl.GetVersion(&m.version), "ncclGetVersion")) return false;
    std::vector<ncclComm_t> comms(devices.size(), nullptr);
    if (!m.ncclOk(m.rccl.CommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll")) return false;
    m.ranks.resize(devices.size());
    m.opened = true;
    for (size_t i = 0; i < devices.size(); i++) {
        m.ranks[i].dev = devices[i];
        m.ranks[i].comm = comms[i];
        if (!m.hipOk(hipSetDevice(devices[i]), "hipSetDevice") ||
            !m.hipOk(hipStreamCreateWithFlags(&m.ranks[i].stream, hipStreamNonBlocking), "hipStreamCreateWithFlags")) { close(); return false; }
    }
    return true;
}

void ShardedMatcher::close() {
    Impl& m = *impl_;
    if (!m.opened) return;
    for (Rank& r : m.ranks) {
        m.freeRank(r);
        if (r.comm) (void)m.rccl.CommDestroy(r.comm);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    m.ranks.clear();
    (void)hipHostFree(m.hq); (void)hipHostFree(m.hres); (void)hipHostFree(m.hgood);
    m.hq = nullptr; m.hres = nullptr; m.hgood = nullptr; m.hcap = 0;
    m.n = 0;
    m.opened = false;
}

bool ShardedMatcher::setMap(const uint8_t* descriptors, long long n) {
    Impl& m = *impl_;
    if (!m.opened) return m.fail("not open");
    if (n < 0 || (n > 0 && !descriptors) || n > 0x7fffffffLL) return m.fail("bad map");
    const int world = (int)m.ranks.size();
    for (int r = 0; r < world; r++) {
        Rank& k = m.ranks[r];
        const long long base = n / world, rem = n % world;
        k.row0 = r * base + (r < rem ? r : rem);
        k.rows = base + (r < rem ? 1 : 0);
        if (!m.hipOk(hipSetDevice(k.dev), "hipSetDevice")) return false;
        (void)hipStreamSynchronize(k.stream);
        (void)hipFree(k.map); k.map = nullptr;
        if (k.rows > 0) {
            if (!m.hipOk(hipMalloc((void**)&k.map, (size_t)k.rows * 32), "hipMalloc(map shard)")) return false;
            if (!m.hipOk(hipMemcpyAsync(k.map, descriptors + (size_t)k.row0 * 32, (size_t)k.rows * 32, hipMemcpyHostToDevice, k.stream), "hipMemcpyAsync(map shard)")) return false;
        }
        k.wsBytes = 0; (void)hipFree(k.ws); k.ws = nullptr;
    }
    for (Rank& k : m.ranks) {
        if (!m.hipOk(hipSetDevice(k.dev), "hipSetDevice")) return false;
        if (!m.hipOk(hipStreamSynchronize(k.stream), "hipStreamSynchronize")) return false;
    }
    m.n = n;
    return true;
}

bool ShardedMatcher::match(const uint8_t* queries, int nq, int* idx, int* dist, float ratio, uint8_t* good) {
    Impl& m = *impl_;
    if (!m.opened) return m.fail("not open");
    if (!queries || nq <= 0 || !idx || !dist) return m.fail("bad arguments");
    if (!m.reserve(nq)) return false;
    const int world = (int)m.ranks.size();
    std::memcpy(m.hq, queries, (size_t)nq * 32);
    const auto t0 = std::chrono::steady_clock::now();
    /* 1. every rank: queries in, scan of its shard (global row numbers) */
    for (Rank& k : m.ranks) {
        SMM_HIP(hipSetDevice(k.dev));
        SMM_HIP(hipMemcpyAsync(k.query, m.hq, (size_t)nq * 32, hipMemcpyHostToDevice, k.stream));
        if (!m.vfeOk(vfe_dev_match_knn2(k.query, nq, k.map, k.rows, (int)k.row0, k.idx, k.dist, k.ws, k.wsBytes, k.stream), "vfe_dev_match_knn2")) return false;
    }
    /* 2. the exchange: both lists of every rank to every rank, as one fused group */
    if (!m.ncclOk(m.rccl.GroupStart(), "ncclGroupStart")) return false;
    for (Rank& k : m.ranks) {
        const ncclResult_t a = m.rccl.AllGather(k.idx, k.allIdx, (size_t)nq * 2, ncclInt32, k.comm, k.stream);
        const ncclResult_t b = m.rccl.AllGather(k.dist, k.allDist, (size_t)nq * 2, ncclInt32, k.comm, k.stream);
        if (a != ncclSuccess || b != ncclSuccess) { (void)m.rccl.GroupEnd(); return m.ncclOk(a != ncclSuccess ? a : b, "ncclAllGather"); }
    }
    if (!m.ncclOk(m.rccl.GroupEnd(), "ncclGroupEnd")) return false;
    /* 3. every rank merges by (distance, global row); rank 0's copy goes back to the caller */
    for (Rank& k : m.ranks) {
        SMM_HIP(hipSetDevice(k.dev));
        if (!m.vfeOk(vfe_dev_merge_knn2(k.allIdx, k.allDist, world, nq, k.outIdx, k.outDist, k.stream), "vfe_dev_merge_knn2")) return false;
        if (good && !m.vfeOk(vfe_dev_ratio_test(k.outIdx, k.outDist, nq, ratio, k.good, k.stream), "vfe_dev_ratio_test")) return false;
    }
    Rank& z = m.ranks[0];
    SMM_HIP(hipSetDevice(z.dev));
    SMM_HIP(hipMemcpyAsync(m.hres, z.outIdx, (size_t)nq * 2 * sizeof(int), hipMemcpyDeviceToHost, z.stream));
    SMM_HIP(hipMemcpyAsync(m.hres + (size_t)nq * 2, z.outDist, (size_t)nq * 2 * sizeof(int), hipMemcpyDeviceToHost, z.stream));
    if (good) SMM_HIP(hipMemcpyAsync(m.hgood, z.good, (size_t)nq, hipMemcpyDeviceToHost, z.stream));
    for (Rank& k : m.ranks) {
        SMM_HIP(hipSetDevice(k.dev));
        SMM_HIP(hipStreamSynchronize(k.stream));
    }
    m.lastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::memcpy(idx, m.hres, (size_t)nq * 2 * sizeof(int));
    std::memcpy(dist, m.hres + (size_t)nq * 2, (size_t)nq * 2 * sizeof(int));
    if (good) std::memcpy(good, m.hgood, (size_t)nq);
    return true;
}

}  // namespace vi_slam
