/* Brute-force Hamming kNN (k = 2) on 256-bit descriptors (cv::BFMatcher(NORM_HAMMING).knnMatch
 * semantics), ratio test and shard merge, for gfx950.
 *
 * Integer-VALU bound (8 xor + 8 popcount-accumulate per pair), not HBM bound: each lane keeps
 * one query descriptor in 8 VGPRs; the train descriptors of a chunk are wave-uniform, so they
 * are fetched once per wave through the scalar data path and broadcast to all 64 lanes as
 * SGPR operands.  A workgroup covers 256 queries x one train chunk; per-chunk top-2 lists are
 * merged by (distance, index), which reproduces the order of one ascending scan with strict
 * '<' (ties keep the lower train index).
 */
#include "vfe_common.h"
#include <mutex>

typedef unsigned int u32;

#define MT_THREADS 256
#define NO_IDX (-1)
#define NO_DIST 0x7fffffff

__device__ static inline void top2_insert(int d, int i, int& d0, int& i0, int& d1, int& i1) {
    /* lexicographic (dist, idx) order; equals strict-'<' insertion when indices arrive ascending */
    if (d < d0 || (d == d0 && i < i0)) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
    else if (d < d1 || (d == d1 && i < i1)) { d1 = d; i1 = i; }
}

/* Top-2 of the train rows [0, n) at `tr` (n <= 65536, wave-uniform address: scalar loads) for the query in a[]: the
 * running pair is kept as two keys dist << 16 | row, so that the lexicographic (dist, row) order -- the order of one
 * ascending scan with strict '<' -- is the integer order and an update is a median and a minimum. */
#define KEY_NONE 0x7fffffffu
#define SCAN_MAX_ROWS 65536
struct Row8 { u32 w[8]; };
/* d + popcount(x): the accumulating form, which the compiler trades for separate counts and three-operand adds */
__device__ static inline u32 popc_acc(u32 x, u32 d) {
    u32 r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(d));
    return r;
}
__device__ static inline void top2_row(const u32 (&a)[8], const Row8& r, u32 t, u32& k0, u32& k1) {
    u32 d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d = popc_acc(a[k] ^ r.w[k], d);
    const u32 key = (d << 16) | t;
    u32 m;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k0), "v"(k1), "v"(key)); /* second smallest of the three */
    k1 = m;
    k0 = min(k0, key);
}
__device__ static inline void scan_top2(const u32 (&a)[8], const u32* __restrict__ tr, int n, u32& k0, u32& k1) {
    k0 = k1 = KEY_NONE;
    if (n <= 0) return;
    /* the row after the one being compared is already on its way (the last row is requested twice) */
    const Row8* rows = reinterpret_cast<const Row8*>(tr);
    Row8 cur = rows[0];
    int t = 0;
    for (; t + 2 <= n; t += 2) {
        const Row8 nx = rows[t + 1];
        top2_row(a, cur, (u32)t, k0, k1);
        cur = rows[min(t + 2, n - 1)];
        top2_row(a, nx, (u32)t + 1u, k0, k1);
    }
    if (t < n) top2_row(a, cur, (u32)t, k0, k1);
}
__device__ static inline void key_decode(u32 key, int row0, int& d, int& i) {
    d = key == KEY_NONE ? NO_DIST : (int)(key >> 16);
    i = key == KEY_NONE ? NO_IDX : row0 + (int)(key & 0xFFFFu);
}

__global__ __launch_bounds__(MT_THREADS) void k_match(const u32* __restrict__ query, int nq,
                                                     const u32* __restrict__ train, long long nt, int chunk,
                                                     int nchunks, int base_index,
                                                     int* __restrict__ part_idx, int* __restrict__ part_dist) {
    const int q = blockIdx.x * MT_THREADS + threadIdx.x;
    const int c = blockIdx.y;
    u32 a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = (q < nq) ? query[(size_t)q * 8 + k] : 0u;
    const long long t0 = (long long)c * chunk;
    const long long t1 = (t0 + chunk < nt) ? t0 + chunk : nt;
    u32 k0, k1;
    scan_top2(a, train + (size_t)t0 * 8, (int)(t1 - t0), k0, k1); /* chunk <= SCAN_MAX_ROWS */
    int d0, d1, i0, i1;
    key_decode(k0, base_index + (int)t0, d0, i0);
    key_decode(k1, base_index + (int)t0, d1, i1);
    if (q < nq) {
        const size_t o = ((size_t)q * nchunks + c) * 2;
        part_idx[o] = i0; part_idx[o + 1] = i1;
        part_dist[o] = d0; part_dist[o + 1] = d1;
    }
}

/* merge `nlists` top-2 lists per query; list l of query q at [(q*q_stride + l*l_stride) * 2] */
__global__ __launch_bounds__(256) void k_merge(const int* __restrict__ in_idx, const int* __restrict__ in_dist,
                                               int nlists, size_t q_stride, size_t l_stride, int nq,
                                               int* __restrict__ out_idx, int* __restrict__ out_dist) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    int d0 = NO_DIST, d1 = NO_DIST, i0 = NO_IDX, i1 = NO_IDX;
    for (int l = 0; l < nlists; l++) {
        const size_t o = ((size_t)q * q_stride + (size_t)l * l_stride) * 2;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = in_idx[o + k], d = in_dist[o + k];
            if (i >= 0) top2_insert(d, i, d0, i0, d1, i1);
        }
    }
    out_idx[2 * q] = i0; out_idx[2 * q + 1] = i1;
    out_dist[2 * q] = d0; out_dist[2 * q + 1] = d1;
}

/* The same merge with one wavefront per query, for many lists per query (the per-chunk partial lists of a large train
 * set): the lanes take the lists round-robin -- consecutive lists of a query are consecutive in memory -- and their
 * top-2 lists are merged by a butterfly.  The two smallest (dist, idx) pairs of a set do not depend on the order in
 * which its members are visited, so the result is the one of the serial merge. */
__global__ __launch_bounds__(256) void k_merge_wave(const int* __restrict__ in_idx, const int* __restrict__ in_dist,
                                                    int nlists, int nq, int* __restrict__ out_idx, int* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (q >= nq) return;
    int d0 = NO_DIST, d1 = NO_DIST, i0 = NO_IDX, i1 = NO_IDX;
    const int2* pi = reinterpret_cast<const int2*>(in_idx) + (size_t)q * nlists;
    const int2* pd = reinterpret_cast<const int2*>(in_dist) + (size_t)q * nlists;
    for (int l = lane; l < nlists; l += 64) {
        const int2 i = pi[l], d = pd[l];
        if (i.x >= 0) top2_insert(d.x, i.x, d0, i0, d1, i1);
        if (i.y >= 0) top2_insert(d.y, i.y, d0, i0, d1, i1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int pd0 = __shfl_xor(d0, o), pi0 = __shfl_xor(i0, o), pd1 = __shfl_xor(d1, o), pi1 = __shfl_xor(i1, o);
        if (pi0 >= 0) top2_insert(pd0, pi0, d0, i0, d1, i1);
        if (pi1 >= 0) top2_insert(pd1, pi1, d0, i0, d1, i1);
    }
    if (lane == 0) {
        out_idx[2 * q] = i0; out_idx[2 * q + 1] = i1;
        out_dist[2 * q] = d0; out_dist[2 * q + 1] = d1;
    }
}

__global__ __launch_bounds__(256) void k_ratio(const int* __restrict__ idx, const int* __restrict__ dist, int nq,
                                               float ratio, uint8_t* __restrict__ good) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    good[q] = (idx[2 * q] >= 0 && idx[2 * q + 1] >= 0 && (float)dist[2 * q] < ratio * (float)dist[2 * q + 1]) ? 1 : 0;
}

/* Batched frame-to-frame matching: workgroup (x, b) serves 64 queries of frame b; its MB_WAVES waves scan
 * an equal share of that frame's train rows each and merge their top-2 lists through LDS (eight waves: 16 frame pairs of
 * 2000 descriptors are then 4096 waves, four per SIMD, instead of two).  Set sizes come
 * from device memory, so extraction and matching can be queued back to back. */
#define MB_WAVES 8
__global__ __launch_bounds__(64 * MB_WAVES) void k_match_batch(const u32* __restrict__ query, const int* __restrict__ nq_dev, int q_cap,
                                                     const u32* __restrict__ train, const int* __restrict__ nt_dev, int t_cap,
                                                     int* __restrict__ out_idx, int* __restrict__ out_dist, float ratio,
                                                     uint8_t* __restrict__ good) {
    __shared__ int s_i[MB_WAVES][64][2], s_d[MB_WAVES][64][2];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nq = min(max(nq_dev[b], 0), q_cap), nt = min(max(nt_dev[b], 0), t_cap);
    const int q = blockIdx.x * 64 + lane;
    const u32* qp = query + ((size_t)b * q_cap + min(q, q_cap - 1)) * 8;
    u32 a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = qp[k];
    /* workgroups past the last query only write the 'no neighbour' records */
    const int per = (nt + MB_WAVES - 1) / MB_WAVES, t0 = min(wv * per, nt), t1 = (int)blockIdx.x * 64 < nq ? min(t0 + per, nt) : t0;
    const u32* tb = train + (size_t)b * t_cap * 8;
    int d0 = NO_DIST, d1 = NO_DIST, i0 = NO_IDX, i1 = NO_IDX;
    for (int ts = t0; ts < t1; ts += SCAN_MAX_ROWS) { /* one round unless the set is huge */
        u32 k0, k1;
        scan_top2(a, tb + (size_t)ts * 8, min(t1 - ts, SCAN_MAX_ROWS), k0, k1);
        int d, i;
        key_decode(k0, ts, d, i);
        if (i >= 0) top2_insert(d, i, d0, i0, d1, i1);
        key_decode(k1, ts, d, i);
        if (i >= 0) top2_insert(d, i, d0, i0, d1, i1);
    }
    s_i[wv][lane][0] = i0; s_i[wv][lane][1] = i1; s_d[wv][lane][0] = d0; s_d[wv][lane][1] = d1;
    __syncthreads();
    if (wv == 0 && q < q_cap) {
        d0 = d1 = NO_DIST; i0 = i1 = NO_IDX;
#pragma unroll
        for (int w = 0; w < MB_WAVES; w++)
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (s_i[w][lane][k] >= 0) top2_insert(s_d[w][lane][k], s_i[w][lane][k], d0, i0, d1, i1);
        if (q >= nq) { d0 = d1 = NO_DIST; i0 = i1 = NO_IDX; }
        const size_t o = ((size_t)b * q_cap + q) * 2;
        out_idx[o] = i0; out_idx[o + 1] = i1;
        out_dist[o] = d0; out_dist[o + 1] = d1;
        if (good) good[(size_t)b * q_cap + q] = (i0 >= 0 && i1 >= 0 && (float)d0 < ratio * (float)d1) ? 1 : 0;
    }
}

static int match_chunk(int nq, long long nt) {
    /* enough (query block x chunk) workgroups to fill 256 CUs several times, chunks >= 64 rows */
    const long long qblocks = (nq + MT_THREADS - 1) / MT_THREADS;
    long long want = 4096 / (qblocks > 0 ? qblocks : 1);
    if (want < 1) want = 1;
    long long chunk = (nt + want - 1) / want;
    if (chunk < 64) chunk = 64;
    if (chunk > 65536) chunk = 65536;
    return (int)chunk;
}

extern "C" size_t vfe_match_workspace(int nq, long long nt) {
    if (nq <= 0 || nt < 0) return 0;
    const int chunk = match_chunk(nq, nt);
    const long long nchunks = nt > 0 ? (nt + chunk - 1) / chunk : 1;
    return align_up((size_t)nq * nchunks * 2 * sizeof(int), 256) * 2;
}

/* k_match over all (query block, chunk) pairs into the partial lists at `workspace` (vfe_match_workspace bytes), then
 * the merge of every query's lists */
static int match_launch(const uint8_t* query, int nq, const uint8_t* train, long long nt, int base_index,
                        int* out_idx, int* out_dist, void* workspace, size_t need, int chunk, long long nchunks, hipStream_t st) {
    int* part_idx = (int*)workspace;
    int* part_dist = (int*)((char*)workspace + need / 2);
    {
        ProfScope prof(VFE_K_MATCH, st);
        hipLaunchKernelGGL(k_match, dim3((nq + MT_THREADS - 1) / MT_THREADS, (unsigned)nchunks), dim3(MT_THREADS), 0, st,
                           (const u32*)query, nq, (const u32*)train, nt, chunk, (int)nchunks, base_index, part_idx, part_dist);
    }
    VFE_LAUNCH_CHECK();
    ProfScope prof(VFE_K_MISC, st);
    if (nchunks >= 16) /* a thread per query would walk its lists alone: a wave per query instead */
        hipLaunchKernelGGL(k_merge_wave, dim3((nq + 3) / 4), dim3(256), 0, st, part_idx, part_dist, (int)nchunks, nq, out_idx, out_dist);
    else
        hipLaunchKernelGGL(k_merge, dim3((nq + 255) / 256), dim3(256), 0, st, part_idx, part_dist, (int)nchunks,
                           (size_t)nchunks, (size_t)1, nq, out_idx, out_dist);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

static int match_check_args(const uint8_t* query, int nq, const uint8_t* train, long long nt, int base_index,
                            const int* out_idx, const int* out_dist, int* chunk, long long* nchunks) {
    VFE_REQUIRE(query && out_idx && out_dist && nq > 0, "NULL buffer or nq <= 0");
    VFE_REQUIRE(nt >= 0 && (nt == 0 || train), "bad train set");
    VFE_REQUIRE((long long)base_index + nt <= 0x7fffffffLL, "train indices overflow int");
    VFE_REQUIRE((uintptr_t)query % 4 == 0 && (uintptr_t)train % 4 == 0, "descriptors must be 4-byte aligned");
    *chunk = match_chunk(nq, nt);
    *nchunks = nt > 0 ? (nt + *chunk - 1) / *chunk : 1;
    VFE_REQUIRE(*nchunks <= 65535, "train set too large for one call");
    return VFE_OK;
}

extern "C" int vfe_dev_match_knn2(const uint8_t* query, int nq, const uint8_t* train, long long nt,
                                  int base_index, int* out_idx, int* out_dist,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    int chunk; long long nchunks;
    const int rc = match_check_args(query, nq, train, nt, base_index, out_idx, out_dist, &chunk, &nchunks);
    if (rc != VFE_OK) return rc;
    const size_t need = vfe_match_workspace(nq, nt);
    if (!workspace || workspace_bytes < need) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
        return VFE_E_NOMEM;
    }
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    return match_launch(query, nq, train, nt, base_index, out_idx, out_dist, workspace, need, chunk, nchunks, (hipStream_t)stream);
}

/* ---- the same search for callers that cannot allocate --------------------------------------------------------------
 * The per-chunk partial lists live in a file-scope device array instead of a caller workspace.  Every GPU the library
 * runs on holds its own copy of the array at its own address for the life of the process; the host-side name below is
 * only a handle for hipGetSymbolAddress, never a device address, so every use goes through the per-device pointer
 * resolved once (vfe_match_scratch_init, called by vfe_create and by the entry point's first call on a device).
 * k_match writes every (query, chunk) slot the merge reads, so nothing has to be cleared between calls; calls that
 * overlap on one GPU share the array and race. */
static __device__ __align__(256) unsigned char scratch[VFE_MATCH_NOALLOC_BYTES];
#define MATCH_MAX_DEVICES 64
static void* g_scratch_ptr[MATCH_MAX_DEVICES];
static std::mutex g_scratch_mu;

int vfe_match_scratch_init(void** out) {
    int dev = -1;
    VFE_HIP(hipGetDevice(&dev));
    VFE_REQUIRE(dev >= 0 && dev < MATCH_MAX_DEVICES, "device ordinal beyond the scratch table");
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        VFE_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        VFE_REQUIRE(p && (uintptr_t)p % 256 == 0, "scratch symbol did not resolve to an aligned device address");
        g_scratch_ptr[dev] = p;
    }
    if (out) *out = g_scratch_ptr[dev];
    return VFE_OK;
}

extern "C" int vfe_dev_match_knn2_noalloc(const uint8_t* query, int nq, const uint8_t* train, long long nt,
                                          int base_index, int* out_idx, int* out_dist, void* stream) {
    int chunk; long long nchunks;
    int rc = match_check_args(query, nq, train, nt, base_index, out_idx, out_dist, &chunk, &nchunks);
    if (rc != VFE_OK) return rc;
    const size_t need = vfe_match_workspace(nq, nt);
    if (need > (size_t)VFE_MATCH_NOALLOC_BYTES) {
        vfe_set_error("%d queries x %lld chunks need %zu bytes of partial lists, the built-in scratch holds %zu: "
                      "use vfe_dev_match_knn2 with a workspace", nq, nchunks, need, (size_t)VFE_MATCH_NOALLOC_BYTES);
        return VFE_E_CAPACITY;
    }
    void* ws = nullptr;
    rc = vfe_match_scratch_init(&ws);
    if (rc != VFE_OK) return rc;
    return match_launch(query, nq, train, nt, base_index, out_idx, out_dist, ws, need, chunk, nchunks, (hipStream_t)stream);
}

extern "C" int vfe_dev_merge_knn2(const int* shard_idx, const int* shard_dist, int nshards, int nq,
                                  int* out_idx, int* out_dist, void* stream) {
    VFE_REQUIRE(shard_idx && shard_dist && out_idx && out_dist && nshards > 0 && nq > 0, "NULL buffer or bad sizes");
    ProfScope prof(VFE_K_MISC, (hipStream_t)stream);
    /* input layout [nshards][nq][2] */
    hipLaunchKernelGGL(k_merge, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, shard_idx, shard_dist, nshards,
                       (size_t)1, (size_t)nq, nq, out_idx, out_dist);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_ratio_test(const int* idx, const int* dist, int nq, float ratio, uint8_t* good, void* stream) {
    VFE_REQUIRE(idx && dist && good && nq > 0, "NULL buffer or nq <= 0");
    hipLaunchKernelGGL(k_ratio, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, dist, nq, ratio, good);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_match_knn2_batch(const uint8_t* query, const int* nq, int q_cap,
                                        const uint8_t* train, const int* nt, int t_cap, int batch,
                                        int* out_idx, int* out_dist, float ratio, uint8_t* good, void* stream) {
    VFE_REQUIRE(query && train && nq && nt && out_idx && out_dist, "NULL buffer");
    VFE_REQUIRE(q_cap > 0 && t_cap > 0 && batch > 0 && batch <= 65535, "bad sizes");
    VFE_REQUIRE((uintptr_t)query % 4 == 0 && (uintptr_t)train % 4 == 0, "descriptors must be 4-byte aligned");
    ProfScope prof(VFE_K_MATCH, (hipStream_t)stream);
    hipLaunchKernelGGL(k_match_batch, dim3((q_cap + 63) / 64, batch), dim3(64 * MB_WAVES), 0, (hipStream_t)stream,
                       (const u32*)query, nq, q_cap, (const u32*)train, nt, t_cap, out_idx, out_dist, ratio, good);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
