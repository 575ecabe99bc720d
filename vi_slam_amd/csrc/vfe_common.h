/* Shared host/device helpers of libvfe (gfx950 only). */
#ifndef VFE_COMMON_H
#define VFE_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "vfe.h"

void vfe_set_error(const char* fmt, ...);

#define VFE_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            vfe_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? VFE_E_NOMEM                                   \
                   : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? VFE_E_NODEVICE : VFE_E_HIP; \
        }                                                                                      \
    } while (0)

#define VFE_REQUIRE(cond, msg)                               \
    do {                                                     \
        if (!(cond)) {                                       \
            vfe_set_error("invalid argument: %s", msg);      \
            return VFE_E_INVALID;                            \
        }                                                    \
    } while (0)

#define VFE_LAUNCH_CHECK()                                                                 \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) {                                                            \
            vfe_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return VFE_E_HIP;                                                              \
        }                                                                                  \
    } while (0)

/* kernel timing: when enabled (vfe_prof_enable), every launch site brackets its kernel with
 * a pair of HIP events on the launch stream */
#include <atomic>
extern std::atomic<int> g_vfe_prof_on;
/* Settings of the context whose entry point the calling thread is inside (null outside one): a context's own overlap /
 * timing choice then wins over the process-wide switches, and its timing records are kept apart under `owner`.
 * overlap / prof: -1 = follow the process-wide switch, 0 = off, 1 = on. */
struct VfeCallOpts { const void* owner; int overlap, prof; };
extern thread_local const VfeCallOpts* t_vfe_call;
struct VfeCallScope {
    const VfeCallOpts* saved;
    explicit VfeCallScope(const VfeCallOpts* o) : saved(t_vfe_call) { t_vfe_call = o; }
    ~VfeCallScope() { t_vfe_call = saved; }
};
bool vfe_prof_begin(int id, hipStream_t st); /* true when a record was opened (the matching end must then be called) */
void vfe_prof_end(int id, hipStream_t st);
int vfe_prof_read_owner(const void* owner, double* ms, long long* launches, int reset);
void vfe_prof_forget_owner(const void* owner);
struct ProfScope {
    int id; hipStream_t st; bool on;
    ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), on(false) {
        const VfeCallOpts* o = t_vfe_call;
        const bool want = (o && o->prof >= 0) ? o->prof != 0 : g_vfe_prof_on.load(std::memory_order_relaxed) != 0;
        if (want) on = vfe_prof_begin(id, st);
    }
    ~ProfScope() { if (on) vfe_prof_end(id, st); }
};

/* A helper stream of the library (one per host thread and device) for independent kernels that run beside the caller's
 * stream, forked from it and joined back into it with the two events.  Once forked, every way out of the entry point --
 * error returns included -- must leave the caller's stream waiting for the helper: the caller may free or reuse its
 * buffers as soon as it has the status code.  ForkGuard does that on scope exit unless the join has been made. */
struct AuxStream { hipStream_t s; hipEvent_t fork, join; };
AuxStream* vfe_aux_stream(hipStream_t caller);
/* vfe_pyramid.hip: will vfe_dev_build_pyramid form levels 1 to 3 of this pyramid in one launch? */
bool vfe_pyramid_three_levels_per_launch(const vfe_dimg* levels, int nlevels);
bool vfe_overlap_on();
struct ForkGuard {
    AuxStream* a; hipStream_t st; bool joined;
    ~ForkGuard() {
        if (a && !joined && hipEventRecord(a->join, a->s) == hipSuccess) (void)hipStreamWaitEvent(st, a->join, 0);
    }
};

/* plain-data view of one batched image, passed to kernels by value */
struct DImg {
    uint8_t* data;
    int w, h, pitch, batch;
    size_t fstride;
    int border; /* reflect-101 padding around every frame, in pixels */
    /* optional second level: each frame is cut into sub_n equal windows (sub_gx per row,
     * sub_dx bytes apart horizontally, sub_dy bytes apart vertically) and `batch` counts
     * windows; sub_n = 1 means whole frames */
    int sub_n, sub_gx, sub_dx;
    size_t sub_dy;
    __host__ __device__ size_t offset(int b) const {
        if (sub_n == 1) return (size_t)b * fstride;
        const int f = b / sub_n, c = b - f * sub_n, cy = c / sub_gx, cx = c - cy * sub_gx;
        return (size_t)f * fstride + (size_t)cy * sub_dy + (size_t)cx * sub_dx;
    }
    __host__ __device__ const uint8_t* frame(int b) const { return data + offset(b); }
    __host__ __device__ uint8_t* frame(int b) { return data + offset(b); }
};

static inline DImg to_dimg(const vfe_dimg* v) {
    DImg d; d.data = v->data; d.w = v->width; d.h = v->height; d.pitch = v->pitch; d.batch = v->batch;
    d.fstride = v->frame_stride; d.border = v->border; d.sub_n = 1; d.sub_gx = 1; d.sub_dx = 0; d.sub_dy = 0; return d;
}

static inline int vfe_check_dimg(const vfe_dimg* v, const char* name) {
    if (!v || !v->data || v->width <= 0 || v->height <= 0 || v->batch <= 0 || v->pitch < v->width ||
        v->border < 0 || v->border > 1024 || v->pitch < v->width + 2 * v->border ||
        v->frame_stride < (size_t)v->pitch * (size_t)(v->height - 1 + (v->batch > 1 ? 2 * v->border : 0)) + (size_t)v->width) {
        vfe_set_error("invalid argument: bad image descriptor %s", name);
        return VFE_E_INVALID;
    }
    return VFE_OK;
}

/* BORDER_REFLECT_101 */
__host__ __device__ static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - 2 - p;
    }
    return p;
}

/* Neighbour lanes of a wave by DPP wave shifts (one VALU move each, no trip through the LDS crossbar).  With bound_ctrl
 * the lane shifted in from outside the wave (lane 0 going up, lane 63 going down) reads 0 and the move needs no prior
 * value in its destination; the streaming kernels keep those two lanes as halo, so nothing reads that 0. */
__device__ static inline uint32_t lane_up_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true); }   /* value of lane - 1 (wave_shr:1) */
__device__ static inline uint32_t lane_down_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true); } /* value of lane + 1 (wave_shl:1) */

/* Workgroups are dealt round-robin over the 8 XCDs of the chip, each with its own L2.  A launch whose workgroups are
 * numbered frame by frame (`per_frame` consecutive workgroups each) spreads every frame over all eight caches, and
 * whatever its workgroups share -- halo rows and columns, cache lines straddled by neighbouring strips, overlapping
 * keypoint windows -- is fetched from memory up to eight times.  This renumbering keeps frame f on XCD f % 8: workgroup
 * `lin` works on (frame, within).  Any renumbering is correct as long as it is a bijection; the frames past the last
 * whole group of eight keep their order. */
__device__ static inline void xcd_frame_of(unsigned lin, unsigned per_frame, unsigned batch, unsigned& frame, unsigned& within) {
    const unsigned grouped = (batch & ~7u) * per_frame;
    if (lin < grouped) {
        const unsigned xcd = lin & 7u, seq = lin >> 3, g = seq / per_frame;
        within = seq - g * per_frame;
        frame = g * 8u + xcd;
    } else {
        frame = lin / per_frame;
        within = lin - frame * per_frame;
    }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int div_up(int a, int b) { return (a + b - 1) / b; }

#endif
