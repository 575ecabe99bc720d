/* ORB keypoints + steered rBRIEF descriptors (cv::ORB::detectAndCompute semantics), batched.
 *
 * Per call: exact-bilinear pyramid (fixed-point INTER_LINEAR_EXACT weights computed on the GPU in
 * IEEE double), FAST-9 + NMS per level (vfe_fast.hip), then per (frame, level):
 *   k_orb_hist    256-bin histogram of the FAST scores,
 *   k_orb_harris  retainBest(2*quota) on the FAST score (smallest kept score from the histogram, ties
 *                 kept), Harris response of the survivors (7x7 block of integer Sobel products, one
 *                 wavefront per survivor, lanes = block positions),
 *   k_orb_thresh  retainBest(quota) on the response via a 3-pass radix select over the survivors'
 *                 order-preserving keys (ties kept),
 *   k_orb_emit    stable compaction into the frame's output list at the level's base offset, so
 *                 keypoints come out in (octave, y, x) order,
 * then a 7x7 sigma-2 fixed-point Gaussian of every level through LDS tiles, and
 *   k_orb_desc    one wavefront per keypoint: intensity-centroid angle (wave reduction), rotation
 *                 by a deterministic double-precision sin/cos, 4 tap pairs per lane, and four
 *                 ballots that are the descriptor's four 64-bit words.
 */
#include "vfe_common.h"
#include "vfe_fast.h"
#include <float.h>
#include <math.h>
#include <string.h>

typedef unsigned int u32;
typedef unsigned long long u64;

#define ORB_MAX_LEVELS 12
#define ORB_EDGE 31
#define ORB_PATCH 31
#define ORB_HALF 15

__constant__ signed char c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
/* half-width of the circular patch at row |v| */
__host__ __device__ constexpr int orb_umax(int v) {
    constexpr int t[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    return t[v];
}

/* ---- level geometry (host; same expressions as OpenCV's ORB) ------------------------------ */
static float orb_scale(int level, float scale_factor) { return (float)pow((double)scale_factor, (double)level); }

extern "C" int vfe_orb_level_sizes(int width, int height, float scale_factor, int nlevels, int* wh, float* scales) {
    VFE_REQUIRE(wh && nlevels >= 1 && nlevels <= ORB_MAX_LEVELS && scale_factor > 1.f, "bad ORB level parameters");
    for (int l = 0; l < nlevels; l++) {
        const float scale = orb_scale(l, scale_factor);
        const float inv = 1.0f / scale;
        wh[2 * l] = (int)lrintf(width * inv);
        wh[2 * l + 1] = (int)lrintf(height * inv);
        if (scales) scales[l] = scale;
    }
    return VFE_OK;
}

extern "C" int vfe_orb_quotas(int nfeatures, float scale_factor, int nlevels, int* quota) {
    VFE_REQUIRE(quota && nlevels >= 1 && nlevels <= ORB_MAX_LEVELS && scale_factor > 1.f && nfeatures > 0, "bad ORB quota parameters");
    const float factor = (float)(1.0 / scale_factor);
    float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { quota[l] = (int)lrintf(nd); sum += quota[l]; nd *= factor; }
    quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    return VFE_OK;
}

/* ---- INTER_LINEAR_EXACT ------------------------------------------------------------------- */
/* tab[v] = source offset << 9 | weight of the right/lower neighbour (0..256); blockIdx.y picks the table */
struct OrbTabs {
    int n;
    int ssize[2 * ORB_MAX_LEVELS], dsize[2 * ORB_MAX_LEVELS];
    u32* tab[2 * ORB_MAX_LEVELS];
};
__global__ void k_orb_tables(OrbTabs T) {
    const int e = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int ssize = T.ssize[e], dsize = T.dsize[e];
    if (v >= dsize) return;
    const double inv_scale = (double)dsize / (double)ssize;
    const double scale = 1.0 / inv_scale;
    const double f = scale * ((double)v + 0.5) - 0.5;
    const int i = (int)floor(f);
    u32 ofs = 0, w1 = 0;
    if (i >= 0 && ssize > 1) {
        if (i < ssize - 1) { ofs = (u32)i; w1 = (u32)(int)rint((f - (double)i) * 256.0); }
        else ofs = (u32)(ssize - 1);
    }
    T.tab[e][v] = (ofs << 9) | w1;
}

/* RS_PX horizontally adjacent output pixels per thread, so that 4 * RS_PX independent source loads are in flight: a
 * thread makes one round trip to memory in its life, and the kernel's rate is the bytes in flight per thread times the
 * threads a CU holds, divided by that round trip.  xtab holds at least RS_PX * wn entries (those past dst.w are not read
 * as offsets). */
#define RS_PX 8
__global__ __launch_bounds__(256) void k_orb_resize(DImg src, DImg dst, const u32* __restrict__ xtab, const u32* __restrict__ ytab,
                                                    int wn, unsigned per_frame) {
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    const int b = blockIdx.y;
    if (id >= per_frame) return;
    const int y = id / (unsigned)wn, x = RS_PX * (int)(id - (unsigned)y * (unsigned)wn);
    u32 tx[RS_PX];
#pragma unroll
    for (int k = 0; k < RS_PX; k += 4) {
        const uint4 t4 = *reinterpret_cast<const uint4*>(xtab + x + k);
        tx[k] = t4.x; tx[k + 1] = t4.y; tx[k + 2] = t4.z; tx[k + 3] = t4.w;
    }
#pragma unroll
    for (int k = 1; k < RS_PX; k++)
        if (x + k >= dst.w) tx[k] = 0;
    const u32 ty = ytab[y];
    const int yo = ty >> 9;
    const u32 wy1 = ty & 511, wy0 = 256 - wy1;
    const uint8_t* r0 = src.frame(b) + (size_t)yo * src.pitch;
    const uint8_t* r1 = src.frame(b) + (size_t)min(yo + 1, src.h - 1) * src.pitch;
    u32 p[RS_PX][4];
#pragma unroll
    for (int k = 0; k < RS_PX; k++) {
        const int xo = tx[k] >> 9, x1 = min(xo + 1, src.w - 1);
        p[k][0] = r0[xo]; p[k][1] = r0[x1]; p[k][2] = r1[xo]; p[k][3] = r1[x1];
    }
    u32 o[RS_PX];
#pragma unroll
    for (int k = 0; k < RS_PX; k++) {
        const u32 wx1 = tx[k] & 511, wx0 = 256 - wx1;
        const u32 h0 = p[k][0] * wx0 + p[k][1] * wx1;
        const u32 h1 = p[k][2] * wx0 + p[k][3] * wx1;
        o[k] = ((h0 * wy0 + h1 * wy1) + (1u << 15)) >> 16;
    }
    uint8_t* op = dst.frame(b) + (size_t)y * dst.pitch + x;
    if (x + RS_PX - 1 < dst.w && ((dst.pitch | dst.fstride | (size_t)(uintptr_t)dst.data) & 3) == 0) {
#pragma unroll
        for (int k = 0; k < RS_PX; k += 4)
            *reinterpret_cast<u32*>(op + k) = o[k] | (o[k + 1] << 8) | (o[k + 2] << 16) | (o[k + 3] << 24);
    } else {
#pragma unroll
        for (int k = 0; k < RS_PX; k++)
            if (x + k < dst.w) op[k] = (uint8_t)o[k];
    }
}

/* The same resize for 4-byte aligned sources and a scale factor of at most 2, separable and streaming.  (The byte
 * gathers of the kernel above go through the texture-address path at a handful of lanes per clock, which bounds it at
 * about 1 TB/s.)  A wave owns RS_SEG output columns and a band of output rows, and a lane RS_PX columns of them
 * for the whole band, so its source offsets and horizontal weights are loop constants.  Source rows are visited once
 * each, in ascending order: the wave copies the row's span into its own LDS area with coalesced dword loads, every lane
 * gathers its 2 * RS_PX taps from there and keeps the row's horizontal blend (<= 255 * 256, two pixels per register).  An output
 * row is the vertical blend of the two newest such rows: one v_dot2 per pixel on (upper, lower) pairs. */
#define RS_SEG (64 * RS_PX)             /* output columns per wave */
#define RS_SPAN_DW (2 * RS_SEG / 4 + 4) /* dwords per staged source row span */
#define RS_BAND 16                      /* output rows per wave */
#define RS_AHEAD 1                      /* source rows requested ahead (deeper: more registers, wasted rows at the end of a band, slower) */
typedef unsigned short rs_u16x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_orb_resize_rows(DImg src, DImg dst, const u32* __restrict__ xtab, const u32* __restrict__ ytab, int gx, int gy) {
    __shared__ u32 s_rows[4][RS_SPAN_DW];
    const int tid = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    /* all workgroups of a frame on one XCD: neighbouring segments and bands share source rows and cache lines */
    unsigned fb, within;
    xcd_frame_of(blockIdx.x, (unsigned)(gx * gy), (unsigned)dst.batch, fb, within);
    const int b = (int)fb, by = (int)within / gx, bx = (int)within - by * gx;
    u32* s_row = s_rows[wv];
    const int y0 = by * RS_BAND, y1 = min(y0 + RS_BAND, dst.h);
    const int X0 = (bx * 4 + wv) * RS_SEG; /* first output pixel of the wave */
    if (X0 >= dst.w) return;                       /* whole waves only: no workgroup barrier below */
    const int xe = min(X0 + RS_SEG, dst.w) - 1;
    const int s_lo = (int)(xtab[X0] >> 9) & ~3, s_hi = min((int)(xtab[xe] >> 9) + 1, src.w - 1);
    const int ndw = min((s_hi - s_lo) / 4 + 1, RS_SPAN_DW);
    const int x = X0 + RS_PX * tid;
    const bool live = x < dst.w;
    /* per thread: byte offsets of the two taps of each pixel within the staged span, weights as 16-bit pairs */
    u32 a01[RS_PX], W0[RS_PX / 2], W1[RS_PX / 2];
    {
        u32 tx[RS_PX];
#pragma unroll
        for (int k = 0; k < RS_PX; k += 4) {
            const uint4 t4 = live ? *reinterpret_cast<const uint4*>(xtab + x + k) : make_uint4(0, 0, 0, 0);
            tx[k] = t4.x; tx[k + 1] = t4.y; tx[k + 2] = t4.z; tx[k + 3] = t4.w;
        }
#pragma unroll
        for (int k = 1; k < RS_PX; k++)
            if (x + k >= dst.w) tx[k] = tx[0];
        if (!live) {
#pragma unroll
            for (int k = 0; k < RS_PX; k++) tx[k] = (u32)s_lo << 9;
        }
#pragma unroll
        for (int k = 0; k < RS_PX; k++) {
            const int xo = (int)(tx[k] >> 9);
            a01[k] = (u32)(xo - s_lo) | ((u32)(min(xo + 1, src.w - 1) - s_lo) << 16);
        }
#pragma unroll
        for (int k = 0; k < RS_PX; k += 2) {
            const u32 w1a = tx[k] & 511, w1b = tx[k + 1] & 511;
            W1[k / 2] = w1a | (w1b << 16);
            W0[k / 2] = (256 - w1a) | ((256 - w1b) << 16);
        }
    }
    const uint8_t* sp = src.frame(b) + s_lo;
    /* horizontal blend of source row r: pixels (k, k+1) of the thread as the 16-bit halves of H[k / 2] */
    constexpr int N_DW = (RS_SPAN_DW + 63) / 64;
    /* this lane's dwords of source rows prow .. prow + RS_AHEAD - 1, requested ahead: a row is a few hundred cycles of
     * work, a trip to memory thousands */
    u32 pv[RS_AHEAD][N_DW];
    int prow = -1 - RS_AHEAD;
    auto request = [&](int r, u32 (&v)[N_DW]) {
        const u32* row = reinterpret_cast<const u32*>(sp + (size_t)min(r, src.h - 1) * src.pitch);
#pragma unroll
        for (int i = 0; i < N_DW; i++) v[i] = tid + 64 * i < ndw ? row[tid + 64 * i] : 0u;
    };
    auto hrow = [&](int r, u32 (&H)[RS_PX / 2]) {
        if (r != prow) { /* uniform; rows are visited in ascending order, one by one within a band */
#pragma unroll
            for (int d = 0; d < RS_AHEAD; d++) request(r + d, pv[d]);
            prow = r;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* the previous row's gathers are done */
#pragma unroll
        for (int i = 0; i < N_DW; i++)
            if (tid + 64 * i < ndw) s_row[tid + 64 * i] = pv[0][i];
#pragma unroll
        for (int d = 0; d + 1 < RS_AHEAD; d++)
#pragma unroll
            for (int i = 0; i < N_DW; i++) pv[d][i] = pv[d + 1][i];
        request(r + RS_AHEAD, pv[RS_AHEAD - 1]);
        prow = r + 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint8_t* l = reinterpret_cast<const uint8_t*>(s_row);
#pragma unroll
        for (int k = 0; k < RS_PX; k += 2) {
            const u32 p0 = (u32)l[a01[k] & 0xFFFFu] | ((u32)l[a01[k + 1] & 0xFFFFu] << 16);
            const u32 p1 = (u32)l[a01[k] >> 16] | ((u32)l[a01[k + 1] >> 16] << 16);
            const rs_u16x2 h = __builtin_bit_cast(rs_u16x2, p0) * __builtin_bit_cast(rs_u16x2, W0[k / 2]) +
                               __builtin_bit_cast(rs_u16x2, p1) * __builtin_bit_cast(rs_u16x2, W1[k / 2]);
            H[k / 2] = __builtin_bit_cast(u32, h);
        }
    };
    u32 Ha[RS_PX / 2], Hb[RS_PX / 2]; /* rows ra and min(ra + 1, src.h - 1) */
    int ra = -2;
    const bool store4 = ((dst.pitch | dst.fstride | (size_t)(uintptr_t)dst.data) & 3) == 0;
    for (int y = y0; y < y1; y++) { /* all branches on y are uniform over the wave */
        const u32 ty = ytab[y];
        const int yo = ty >> 9;
        const u32 wy1 = ty & 511, wy0 = 256 - wy1;
        if (yo != ra) {
            if (yo == ra + 1) {
#pragma unroll
                for (int k = 0; k < RS_PX / 2; k++) Ha[k] = Hb[k];
            } else hrow(yo, Ha);
            if (yo + 1 <= src.h - 1) hrow(yo + 1, Hb);
            else {
#pragma unroll
                for (int k = 0; k < RS_PX / 2; k++) Hb[k] = Ha[k];
            }
            ra = yo;
        }
        if (!live) continue;
        const u32 wy = wy0 | (wy1 << 16);
        u32 o[RS_PX];
#pragma unroll
        for (int k = 0; k < RS_PX; k += 2) {
            /* (upper, lower) of pixel k and of pixel k + 1 */
            const u32 e = __builtin_amdgcn_perm(Hb[k / 2], Ha[k / 2], 0x05040100u), d = __builtin_amdgcn_perm(Hb[k / 2], Ha[k / 2], 0x07060302u);
            o[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, e), __builtin_bit_cast(rs_u16x2, wy), 1u << 15, false) >> 16;
            o[k + 1] = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, d), __builtin_bit_cast(rs_u16x2, wy), 1u << 15, false) >> 16;
        }
        uint8_t* op = dst.frame(b) + (size_t)y * dst.pitch + x;
        if (x + RS_PX - 1 < dst.w && store4) {
#pragma unroll
            for (int k = 0; k < RS_PX; k += 4)
                *reinterpret_cast<u32*>(op + k) = o[k] | (o[k + 1] << 8) | (o[k + 2] << 16) | (o[k + 3] << 24);
        } else {
#pragma unroll
            for (int k = 0; k < RS_PX; k++)
                if (x + k < dst.w) op[k] = (uint8_t)o[k];
        }
    }
}


/* ---- keypoint selection --------------------------------------------------------------------- */
struct OrbLevels {
    DImg raw[ORB_MAX_LEVELS];
    float scale[ORB_MAX_LEVELS];
    int quota[ORB_MAX_LEVELS];
    int cap[ORB_MAX_LEVELS];          /* FAST keypoint capacity of the level */
    const int* kps[ORB_MAX_LEVELS];   /* [batch][cap][3] */
    const int* cnt[ORB_MAX_LEVELS];   /* [batch] */
    u32* key[ORB_MAX_LEVELS];         /* [batch][cap] order-preserving Harris keys by FAST index, 0 = dropped */
    u32* skey[ORB_MAX_LEVELS];        /* [batch][cap] keys of the survivors of the first cut, any order */
    u32* hist;                        /* [batch][ORB_MAX_LEVELS][256] FAST score histogram (zeroed per call) */
    u32* scnt;                        /* [batch][ORB_MAX_LEVELS] survivors of the first cut (zeroed per call) */
    u32* thr;                         /* [batch][ORB_MAX_LEVELS] smallest kept key; 0 keeps nothing */
    int nlevels;
};

__device__ static inline u32 float_key(float f) {
    const u32 u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ static inline float key_float(u32 k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

#define SEL_CHUNKS 32 /* workgroups per (level, frame) in the per-keypoint passes */

__global__ __launch_bounds__(256) void k_orb_hist(OrbLevels L) {
    __shared__ u32 s_hist[256];
    const int l = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int cap = L.cap[l];
    const int n = min(L.cnt[l][b], cap);
    if (n <= 2 * L.quota[l] || (int)(blockIdx.x * 256u) >= n) return; /* no first cut: histogram unused */
    const int* kps = L.kps[l] + (size_t)b * cap * 3;
    s_hist[tid] = 0;
    __syncthreads();
    for (int i = blockIdx.x * 256 + tid; i < n; i += SEL_CHUNKS * 256) atomicAdd(&s_hist[kps[3 * i + 2] & 255], 1u);
    __syncthreads();
    if (s_hist[tid]) atomicAdd(&L.hist[((size_t)b * ORB_MAX_LEVELS + l) * 256 + tid], s_hist[tid]);
}

/* sum over each row of 16 lanes, result in every lane of the row */
__device__ static inline int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);  /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);  /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); /* row_ror 4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); /* row_ror 8 */
    return v;
}

#define HARRIS_KP 64      /* keypoints a wave looks at per pass: a lane each for the first cut; the few that survive it
                           * (2 x quota of a level's thousands) then fill the four 16-lane rows of the response pass */
#define HARRIS_CHUNKS 32  /* single-wave workgroups per (level, frame) */

__global__ __launch_bounds__(64) void k_orb_harris(OrbLevels L, int nlevels, int batch) {
    __shared__ u32 s_xy[HARRIS_KP], s_key[HARRIS_KP];
    __shared__ u32 s_win[4][28];
    /* a frame's workgroups (all chunks of all levels) on one XCD: the 9 x 9 windows of neighbouring candidates overlap */
    unsigned fb, within;
    xcd_frame_of(blockIdx.x, (unsigned)(HARRIS_CHUNKS * nlevels), (unsigned)batch, fb, within);
    const int b = (int)fb, l = (int)within / HARRIS_CHUNKS, chunk = (int)within - l * HARRIS_CHUNKS, lane = threadIdx.x;
    const int cap = L.cap[l];
    const int n = min(L.cnt[l][b], cap);
    if (chunk * HARRIS_KP >= n) return;
    const int* kps = L.kps[l] + (size_t)b * cap * 3;
    u32* key = L.key[l] + (size_t)b * cap;
    u32* skey = L.skey[l] + (size_t)b * cap;
    const int quota = L.quota[l];
    const DImg& I = L.raw[l];
    const uint8_t* ip = I.frame(b);
    const int pitch = I.pitch;

    /* retainBest(2*quota) on the FAST score: smallest kept score = largest s >= 1 with at least 2*quota
     * scores >= s (0 if there is none).  Lane j holds bins 4j .. 4j+3. */
    int smin = 0;
    if (n > 2 * quota) {
        if (2 * quota == 0) smin = 256;
        else {
            const uint4 h = reinterpret_cast<const uint4*>(L.hist + ((size_t)b * ORB_MAX_LEVELS + l) * 256)[lane];
            u32 S = h.x + h.y + h.z + h.w; /* becomes the number of scores >= 4 * lane */
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 t = (u32)__shfl_down((int)S, o);
                if (lane + o < 64) S += t;
            }
            const u32 need = (u32)(2 * quota);
            const unsigned long long ok = __ballot(S >= need);
            if (ok) {
                const int jh = 63 - __clzll((long long)ok);
                int d = 4 * lane;
                if (S - h.x >= need) d = 4 * lane + 1;
                if (S - h.x - h.y >= need) d = 4 * lane + 2;
                if (S - h.x - h.y - h.z >= need) d = 4 * lane + 3;
                smin = __shfl(d, jh);
            }
        }
    }
    /* a row of 16 lanes works on one survivor; lane `sub` of the row takes positions sub, sub + 16, ... of the
     * 7x7 block */
    const int row = lane >> 4, sub = lane & 15;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int i0 = chunk * HARRIS_KP; i0 < n; i0 += HARRIS_CHUNKS * HARRIS_KP) { /* wave-uniform */
        const int i = i0 + lane;
        const bool mine = lane < HARRIS_KP && i < n;
        int kx = 0, ky = 0;
        bool surv = false;
        if (mine) { kx = kps[3 * i]; ky = kps[3 * i + 1]; surv = kps[3 * i + 2] >= smin; }
        const unsigned long long m = __ballot(surv);
        const int cnt = __popcll(m);
        const int slot = surv ? __popcll(m & lt_mask) : -1;
        if (surv) s_xy[slot] = (u32)kx | ((u32)ky << 16);
        __syncthreads();
        for (int s0 = 0; s0 < cnt; s0 += 4) { /* wave-uniform */
            const int sv = s0 + row;
            int a = 0, bb = 0, c = 0;
            /* the survivor's 9 x 9 window (rows y-4 .. y+4, columns x-4 .. x+4, inside the image: keypoints keep 31 pixels
             * from the rim) goes through LDS as 9 rows of three unaligned dwords, two dwords per lane */
            u32* win = s_win[row];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (sv < cnt) {
                const u32 xy = s_xy[sv];
                const uint8_t* base = ip + (size_t)((int)(xy >> 16) - 4) * pitch + ((int)(xy & 0xFFFFu) - 4);
                u32 w0, w1 = 0;
                __builtin_memcpy(&w0, base + (size_t)(sub / 3) * pitch + 4 * (sub % 3), 4);
                if (sub + 16 < 27) __builtin_memcpy(&w1, base + (size_t)((sub + 16) / 3) * pitch + 4 * ((sub + 16) % 3), 4);
                win[sub] = w0;
                if (sub + 16 < 27) win[sub + 16] = w1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (sv < cnt) {
                const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int p = sub + 16 * r;
                    if (p < 49) {
                        const int py = p / 7, px = p - 7 * py;
                        const uint8_t* r0 = wb + py * 12 + px;
                        const uint8_t* r1 = r0 + 12;
                        const uint8_t* r2 = r1 + 12;
                        const int Ix = (r1[2] - r1[0]) * 2 + (r0[2] - r0[0]) + (r2[2] - r2[0]);
                        const int Iy = (r2[1] - r0[1]) * 2 + (r2[0] - r0[0]) + (r2[2] - r0[2]);
                        a += Ix * Ix; bb += Iy * Iy; c += Ix * Iy;
                    }
                }
            }
            a = row16_sum(a); bb = row16_sum(bb); c = row16_sum(c);
            const float scale = 1.f / ((float)((1 << 2) * 7) * 255.f);
            const float scale4 = scale * scale * scale * scale;
            const float fa = (float)a, fb = (float)bb, fc = (float)c;
            const float sm = fa + fb;
            const float resp = ((fa * fb - fc * fc) - (0.04f * sm) * sm) * scale4;
            if (sv < cnt && sub == 0) s_key[sv] = float_key(resp);
        }
        __syncthreads();
        if (mine) key[i] = slot >= 0 ? s_key[slot] : 0u;
        if (cnt) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&L.scnt[b * ORB_MAX_LEVELS + l], (u32)cnt);
            base = (u32)__shfl((int)base, 0);
            if (lane < cnt) skey[base + lane] = s_key[lane];
        }
        __syncthreads();
    }
}

/* retainBest(quota) on the response: key of the quota-th largest among the survivors, ties kept */
__global__ __launch_bounds__(1024) void k_orb_thresh(OrbLevels L, int* __restrict__ lvl_count) {
    __shared__ u32 s_hist[2048 + 1];
    __shared__ u32 s_scan[1024];
    __shared__ u32 s_val;
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cap = L.cap[l];
    const int quota = L.quota[l];
    const u32 n1 = min(L.scnt[b * ORB_MAX_LEVELS + l], (u32)cap);
    const u32* skey = L.skey[l] + (size_t)b * cap;
    u32 T = 1; /* keep every survivor (keys are >= 1) */
    if (quota == 0) T = 0;
    else if (n1 > (u32)quota) {
        u32 prefix = 0, need = (u32)quota;
        int hi = 32;
        while (hi > 0) {
            const int bits = hi >= 11 ? 11 : hi, sh = hi - bits;
            for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0;
            __syncthreads();
            for (u32 i = tid; i < n1; i += 1024) {
                const u32 k = skey[i];
                if (hi == 32 || (k >> hi) == prefix) atomicAdd(&s_hist[(k >> sh) & ((1u << bits) - 1)], 1u);
            }
            __syncthreads();
            {   /* digit = largest d >= 1 with at least `need` keys of digit >= d (0 if none); thread t owns bins 2t, 2t+1 */
                const u32 h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
                if (tid == 0) s_val = 0;
                s_scan[tid] = h0 + h1;
                __syncthreads();
                u32 S = h0 + h1; /* becomes the number of keys of digit >= 2 * tid */
                for (int o = 1; o < 1024; o <<= 1) {
                    const u32 t = tid + o < 1024 ? s_scan[tid + o] : 0u;
                    __syncthreads();
                    S += t;
                    s_scan[tid] = S;
                    __syncthreads();
                }
                if (S - h0 >= need) atomicMax(&s_val, (u32)(2 * tid + 1));
                else if (S >= need && tid > 0) atomicMax(&s_val, (u32)(2 * tid));
                __syncthreads();
                const u32 d = s_val;
                if ((d >> 1) == (u32)tid) s_hist[2048] = (d & 1) ? S - h0 - h1 : S - h0; /* keys of a larger digit */
                __syncthreads();
            }
            const u32 d = s_val;
            need -= s_hist[2048];
            prefix = (hi == 32 ? 0u : (prefix << bits)) | d;
            hi = sh;
            __syncthreads();
        }
        T = prefix;
    }
    u32 cnt = 0;
    if (T != 0)
        for (u32 i = tid; i < n1; i += 1024) cnt += skey[i] >= T ? 1u : 0u;
    s_scan[tid] = cnt;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (tid < o) s_scan[tid] += s_scan[tid + o]; __syncthreads(); }
    if (tid == 0) { lvl_count[b * ORB_MAX_LEVELS + l] = (int)s_scan[0]; L.thr[b * ORB_MAX_LEVELS + l] = T; }
}

__global__ __launch_bounds__(1024) void k_orb_emit(OrbLevels L, const int* __restrict__ lvl_count,
                                                   vfe_orb_kp* __restrict__ out, int out_cap, int* __restrict__ counts) {
    __shared__ u32 s_scan[2][1024];
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cap = L.cap[l];
    const int n = min(L.cnt[l][b], cap);
    const int* kps = L.kps[l] + (size_t)b * cap * 3;
    const u32* key = L.key[l] + (size_t)b * cap;
    const u32 T = L.thr[b * ORB_MAX_LEVELS + l];
    int base = 0;
    for (int k = 0; k < l; k++) base += lvl_count[b * ORB_MAX_LEVELS + k];
    const float sf = L.scale[l];
    /* stable compaction: thread t owns a contiguous slice */
    const int per = (n + 1023) / 1024;
    const int i0 = min(tid * per, n), i1 = min(i0 + per, n);
    u32 mine = 0;
    for (int i = i0; i < i1; i++) mine += (T != 0 && key[i] >= T) ? 1u : 0u;
    int cur = 0;
    s_scan[0][tid] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        u32 x = s_scan[cur][tid];
        if (tid >= o) x += s_scan[cur][tid - o];
        s_scan[cur ^ 1][tid] = x;
        cur ^= 1;
        __syncthreads();
    }
    int pos = base + (int)(s_scan[cur][tid] - mine);
    for (int i = i0; i < i1; i++) {
        if (T == 0 || key[i] < T) continue;
        if (pos < out_cap) {
            vfe_orb_kp kp;
            kp.lx = kps[3 * i]; kp.ly = kps[3 * i + 1];
            kp.x = (float)kp.lx * sf; kp.y = (float)kp.ly * sf;
            kp.size = (float)ORB_PATCH * sf;
            kp.angle = -1.f;
            kp.response = key_float(key[i]);
            kp.octave = l;
            out[(size_t)b * out_cap + pos] = kp;
        }
        pos++;
    }
    if (l == L.nlevels - 1 && tid == 1023) counts[b] = min(base + (int)s_scan[cur][1023], out_cap);
}

/* ---- orientation + descriptor ---------------------------------------------------------------- */
__device__ static inline float fast_atan2f_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI),
                p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON); c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON); c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* same operation sequence as the CPU checker's deterministic sin/cos (double, one float rounding) */
__device__ static inline void sincos_det(float xf, float* s_out, float* c_out) {
    const double x = (double)xf;
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    const double kd = rint(x * two_over_pi);
    const int k = (int)kd;
    double r = x - kd * pio2_hi;
    r = r - kd * pio2_lo;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S6;
    ps = ps * z + S5; ps = ps * z + S4; ps = ps * z + S3; ps = ps * z + S2; ps = ps * z + S1;
    const double sn = r + (r * z) * ps;
    double pc = C6;
    pc = pc * z + C5; pc = pc * z + C4; pc = pc * z + C3; pc = pc * z + C2; pc = pc * z + C1;
    const double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s; *c_out = (float)c;
}

#define DP_R 18                /* largest |coordinate| of a rotated pattern point */
#define DP_ROWS (2 * DP_R + 7) /* rows of horizontal sums a keypoint needs */
#define DP_W 40                /* columns of horizontal sums kept: 2 * DP_R + 1, rounded up to whole lanes of four */
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ static inline u16x2 as_h2(u32 v) { return __builtin_bit_cast(u16x2, v); }
__device__ static inline u32 as_u(u16x2 v) { return __builtin_bit_cast(u32, v); }
/* (byte i of the 8 bytes {hi:lo}) | (byte j) << 16 */
__device__ static inline u32 byte_pair(u32 lo, u32 hi, int i, int j) {
    return __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (u32)i | ((u32)j << 16));
}

__global__ __launch_bounds__(256) void k_orb_desc(OrbLevels L, vfe_orb_kp* __restrict__ kps, uint8_t* __restrict__ desc,
                                                  const int* __restrict__ counts, int out_cap, int batch) {
    __shared__ __align__(16) unsigned short s_h[4][DP_ROWS][DP_W]; /* horizontal blur sums of the keypoint's window */
    __shared__ __align__(16) u32 s_raw[4][DP_ROWS][DP_W / 4];       /* the window itself */
    const int lane = threadIdx.x & 63;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    /* items are workgroups (four consecutive keypoints of a frame), a frame's items all on one XCD: keypoints come in
     * (octave, y, x) order and the 43 x 48 windows of neighbours overlap; the grid is a multiple of 8, so a workgroup
     * stays on its XCD from item to item */
    const unsigned per_frame = ((unsigned)out_cap + 3u) / 4u, total = (unsigned)batch * per_frame;
    unsigned short* hs = s_h[threadIdx.x >> 6][0];
    u32* raw = s_raw[threadIdx.x >> 6][0];
    for (unsigned item = blockIdx.x; item < total; item += gridDim.x) {
        unsigned fb, within;
        xcd_frame_of(item, per_frame, (unsigned)batch, fb, within);
        const int b = (int)fb, i = (int)(within * 4u + wv);
        if (i >= out_cap || i >= counts[b]) continue;
        vfe_orb_kp* kp = kps + (size_t)b * out_cap + i;
        const int l = kp->octave, px = kp->lx, py = kp->ly;
        const DImg& I = L.raw[l];
        const uint8_t* ip = I.frame(b);
        /* the centre of the sampling pattern is recomputed from the scaled position (equal to (px, py) up to rounding) */
        const float sf = L.scale[l];
        const float inv = 1.f / sf;
        const int cx = __float2int_rn(kp->x * inv), cy = __float2int_rn(kp->y * inv);
        /* Everything a keypoint reads lies in a 43-row x 48-column window of its level around the centre, inside the
         * image because keypoints keep 31 pixels from the rim: the 31 x 31 patch of the intensity centroid, and the
         * pixels under the 7x7 Gaussian (cv::GaussianBlur, sigma 2: 8.8 fixed-point taps 18 34 48 56 48 34 18, horizontal
         * sums kept in 16 bits, one rounding after the vertical pass) at the rotated pattern points, which stay within
         * 18 pixels of the centre (|(-13, -13)| = 18.4).  The wave loads the window once (lane -> four columns of a row
         * as three unaligned dwords), keeps it and its horizontal blur sums in LDS, and finishes the blur only at the
         * 512 sampled points. */
        {
            const uint8_t* rp = ip + (size_t)(cy - DP_R - 3) * I.pitch + (cx - DP_R);
            constexpr int N_IT = (DP_ROWS * (DP_W / 4) + 63) / 64;
            u32 lw[N_IT], cw[N_IT], rw[N_IT];
#pragma unroll
            for (int it = 0; it < N_IT; it++) { /* all loads first */
                const int item = min(it * 64 + lane, DP_ROWS * (DP_W / 4) - 1);
                const int r = item / (DP_W / 4), g = item - r * (DP_W / 4);
                const uint8_t* p = rp + (size_t)r * I.pitch + 4 * g;
                __builtin_memcpy(&lw[it], p - 4, 4); __builtin_memcpy(&cw[it], p, 4); __builtin_memcpy(&rw[it], p + 4, 4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* the previous keypoint's LDS reads are done */
#pragma unroll
            for (int it = 0; it < N_IT; it++) {
                const int item = it * 64 + lane;
                /* bytes of columns x-3 .. x+6 are l1 l2 l3 c0 c1 c2 c3 r0 r1 r2 =: q0 .. q9; pixel k taps q[k] .. q[k+6] */
                const u32 l_ = lw[it], c_ = cw[it], r_ = rw[it];
                u16x2 P[9];
                P[0] = as_h2(byte_pair(l_, l_, 1, 2)); P[1] = as_h2(byte_pair(l_, l_, 2, 3)); P[2] = as_h2(byte_pair(l_, c_, 3, 4));
                P[3] = as_h2(byte_pair(c_, c_, 0, 1)); P[4] = as_h2(byte_pair(c_, c_, 1, 2)); P[5] = as_h2(byte_pair(c_, c_, 2, 3));
                P[6] = as_h2(byte_pair(c_, r_, 3, 4)); P[7] = as_h2(byte_pair(r_, r_, 0, 1)); P[8] = as_h2(byte_pair(r_, r_, 1, 2));
                uint2 h;
                /* 18 (q0+q6) + 34 (q1+q5) + 48 (q2+q4) + 56 q3 <= 65280: fits 16 bits */
                h.x = as_u((P[0] + P[6]) * (unsigned short)18 + (P[1] + P[5]) * (unsigned short)34 +
                           (P[2] + P[4]) * (unsigned short)48 + P[3] * (unsigned short)56);
                h.y = as_u((P[2] + P[8]) * (unsigned short)18 + (P[3] + P[7]) * (unsigned short)34 +
                           (P[4] + P[6]) * (unsigned short)48 + P[5] * (unsigned short)56);
                if (item < DP_ROWS * (DP_W / 4)) { /* item = r * (DP_W / 4) + g: both arrays are row-major in it */
                    *reinterpret_cast<uint2*>(hs + 4 * item) = h;
                    raw[item] = c_;
                }
                __builtin_amdgcn_sched_barrier(0); /* one item at a time: interleaving them multiplies the registers in flight */
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        /* intensity centroid over the circular patch: lane & 31 <-> column u; lanes 0..31 take rows -15 .. 0,
         * lanes 32..63 rows 1 .. 15 (row 16 - j of the lower half pairs with row j - 15 of the upper: same
         * half-width) */
        int m10 = 0, m01 = 0;
        {
            const int u = (lane & 31) - ORB_HALF;
            const bool low = lane >= 32;
            /* pixel (px + u, py + v) is byte (px - cx + u + DP_R) of window row (py - cy + v + DP_R + 3) */
            const uint8_t* col = reinterpret_cast<const uint8_t*>(raw) + (py - cy + DP_R + 3) * DP_W + (px - cx + DP_R) + min(u, ORB_HALF);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int d = orb_umax(15 - j);              /* |v| = 15 - j in both halves ... */
                const int v = low ? 15 - j : j - 15;         /* ... v = -(15 - j) above, 15 - j below */
                const bool on = (lane & 31) < ORB_PATCH && u >= -d && u <= d && !(low && j == 15);
                const int val = on ? (int)col[v * DP_W] : 0;
                m10 += u * val; m01 += v * val;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
        const float angle = fast_atan2f_deg((float)m01, (float)m10);
        float sa, ca;
        sincos_det(angle * (float)(M_PI / 180.f), &sa, &ca);
        /* smoothed value at (dx, dy) from the centre: rows dy-3 .. dy+3 of the level are rows dy + DP_R .. + 6 of hs */
        auto smooth = [&](int dx, int dy) -> int {
            const unsigned short* c = hs + (dy + DP_R) * DP_W + (dx + DP_R);
            const u32 v = 18u * ((u32)c[0] + c[6 * DP_W]) + 34u * ((u32)c[DP_W] + c[5 * DP_W]) +
                          48u * ((u32)c[2 * DP_W] + c[4 * DP_W]) + 56u * (u32)c[3 * DP_W];
            return (int)((v + (1u << 15)) >> 16);
        };
        u64 words[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const signed char* pt = &c_pattern[(k * 64 + lane) * 4];
            const float x0 = (float)pt[0] * ca - (float)pt[1] * sa, y0 = (float)pt[0] * sa + (float)pt[1] * ca;
            const float x1 = (float)pt[2] * ca - (float)pt[3] * sa, y1 = (float)pt[2] * sa + (float)pt[3] * ca;
            const int t0 = smooth(__float2int_rn(x0), __float2int_rn(y0));
            const int t1 = smooth(__float2int_rn(x1), __float2int_rn(y1));
            words[k] = __ballot(t0 < t1);
        }
        if (lane == 0) {
            kp->angle = angle;
            u64* d = reinterpret_cast<u64*>(desc + ((size_t)b * out_cap + i) * 32);
            d[0] = words[0]; d[1] = words[1]; d[2] = words[2]; d[3] = words[3];
        }
    }
}

/* ---- host orchestration --------------------------------------------------------------------- */
struct OrbWs {
    int nlevels;
    int w[ORB_MAX_LEVELS], h[ORB_MAX_LEVELS], pitch[ORB_MAX_LEVELS], cap[ORB_MAX_LEVELS];
    size_t raw_off[ORB_MAX_LEVELS], xtab_off[ORB_MAX_LEVELS], ytab_off[ORB_MAX_LEVELS];
    size_t kps_off[ORB_MAX_LEVELS], cnt_off[ORB_MAX_LEVELS], key_off[ORB_MAX_LEVELS], skey_off[ORB_MAX_LEVELS];
    size_t lvlcnt_off, hist_off, hist_bytes, thr_off, fast_off, fast_bytes, total;
};

static int orb_layout(int width, int height, int batch, float scale_factor, int nlevels, OrbWs* L) {
    int wh[2 * ORB_MAX_LEVELS];
    int rc = vfe_orb_level_sizes(width, height, scale_factor, nlevels, wh, nullptr);
    if (rc) return rc;
    L->nlevels = nlevels;
    size_t o = 0;
    for (int l = 0; l < nlevels; l++) {
        L->w[l] = wh[2 * l]; L->h[l] = wh[2 * l + 1];
        if (L->w[l] < 1 || L->h[l] < 1) { vfe_set_error("ORB level %d is empty", l); return VFE_E_INVALID; }
        L->pitch[l] = (int)align_up((size_t)L->w[l], 64);
        L->cap[l] = (L->w[l] * L->h[l]) / 4 + 64;
        const size_t img = align_up((size_t)batch * L->h[l] * L->pitch[l], 256);
        L->raw_off[l] = o; if (l > 0) o += img;   /* level 0 is the caller's image */
        L->xtab_off[l] = o; o += align_up((size_t)L->w[l] * sizeof(u32), 256);
        L->ytab_off[l] = o; o += align_up((size_t)L->h[l] * sizeof(u32), 256);
        L->kps_off[l] = o; o += align_up((size_t)batch * L->cap[l] * 3 * sizeof(int), 256);
        L->cnt_off[l] = o; o += align_up((size_t)batch * sizeof(int), 256);
        L->key_off[l] = o; o += align_up((size_t)batch * L->cap[l] * sizeof(u32), 256);
        L->skey_off[l] = o; o += align_up((size_t)batch * L->cap[l] * sizeof(u32), 256);
    }
    L->lvlcnt_off = o; o += align_up((size_t)batch * ORB_MAX_LEVELS * sizeof(int), 256);
    /* score histograms and survivor counters, cleared together at the start of every call */
    L->hist_bytes = align_up((size_t)batch * ORB_MAX_LEVELS * 257 * sizeof(u32), 256);
    L->hist_off = o; o += L->hist_bytes;
    L->thr_off = o; o += align_up((size_t)batch * ORB_MAX_LEVELS * sizeof(u32), 256);
    {   /* FAST runs on all levels at once (levels too small for the 31-pixel border yield nothing) */
        int fw[ORB_MAX_LEVELS], fh[ORB_MAX_LEVELS], nf = 0;
        for (int l = 0; l < nlevels; l++)
            if (L->w[l] > 2 * ORB_EDGE && L->h[l] > 2 * ORB_EDGE) { fw[nf] = L->w[l]; fh[nf] = L->h[l]; nf++; }
        L->fast_bytes = nf ? fast_set_layout(nf, fw, fh, batch).total : 0;
    }
    L->fast_off = o; o += align_up(L->fast_bytes, 256);
    L->total = o;
    return VFE_OK;
}

extern "C" size_t vfe_orb_workspace(int width, int height, int batch, float scale_factor, int nlevels) {
    if (width <= 0 || height <= 0 || batch <= 0 || nlevels < 1 || nlevels > ORB_MAX_LEVELS || !(scale_factor > 1.f)) return 0;
    OrbWs L;
    if (orb_layout(width, height, batch, scale_factor, nlevels, &L)) return 0;
    return L.total;
}

extern "C" int vfe_dev_orb_extract(const vfe_dimg* img, int nfeatures, float scale_factor, int nlevels,
                                   int fast_threshold, vfe_orb_kp* kps, uint8_t* desc, int cap, int* counts,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    int rc = vfe_check_dimg(img, "img");
    if (rc) return rc;
    VFE_REQUIRE(nfeatures > 0 && nlevels >= 1 && nlevels <= ORB_MAX_LEVELS && scale_factor > 1.f, "bad ORB parameters");
    VFE_REQUIRE(fast_threshold >= 1 && fast_threshold <= 254, "fast_threshold must be 1..254");
    VFE_REQUIRE(kps && desc && counts && workspace && cap > 0, "NULL output or workspace");
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0 && (uintptr_t)desc % 8 == 0, "workspace/desc alignment");
    VFE_REQUIRE(img->batch <= 65535, "batch too large");
    VFE_REQUIRE(img->width <= 65535 && img->height <= 65535, "image too large");
    OrbWs W;
    if ((rc = orb_layout(img->width, img->height, img->batch, scale_factor, nlevels, &W))) return rc;
    if (workspace_bytes < W.total) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", W.total, workspace_bytes);
        return VFE_E_NOMEM;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int batch = img->batch;
    OrbLevels L;
    memset(&L, 0, sizeof(L));
    L.nlevels = nlevels;
    float scales[ORB_MAX_LEVELS]; int wh[2 * ORB_MAX_LEVELS];
    vfe_orb_level_sizes(img->width, img->height, scale_factor, nlevels, wh, scales);
    vfe_orb_quotas(nfeatures, scale_factor, nlevels, L.quota);
    for (int l = 0; l < nlevels; l++) {
        DImg d; d.border = 0; d.w = W.w[l]; d.h = W.h[l]; d.pitch = W.pitch[l]; d.batch = batch; d.fstride = (size_t)W.h[l] * W.pitch[l];
        d.sub_n = 1; d.sub_gx = 1; d.sub_dx = 0; d.sub_dy = 0;
        if (l == 0) L.raw[0] = to_dimg(img);
        else { d.data = (uint8_t*)(ws + W.raw_off[l]); L.raw[l] = d; }
        L.scale[l] = scales[l];
        L.cap[l] = W.cap[l];
        L.kps[l] = (const int*)(ws + W.kps_off[l]);
        L.cnt[l] = (const int*)(ws + W.cnt_off[l]);
        L.key[l] = (u32*)(ws + W.key_off[l]);
        L.skey[l] = (u32*)(ws + W.skey_off[l]);
    }
    int* lvl_count = (int*)(ws + W.lvlcnt_off);
    L.hist = (u32*)(ws + W.hist_off);
    L.scnt = L.hist + (size_t)batch * ORB_MAX_LEVELS * 256;
    L.thr = (u32*)(ws + W.thr_off);
    /* pyramid */
    if (nlevels > 1) {
        ProfScope prof(VFE_K_ORB_RESIZE, st);
        OrbTabs T;
        T.n = 0;
        int maxd = 1;
        for (int l = 1; l < nlevels; l++) {
            T.ssize[T.n] = W.w[l - 1]; T.dsize[T.n] = W.w[l]; T.tab[T.n] = (u32*)(ws + W.xtab_off[l]); T.n++;
            T.ssize[T.n] = W.h[l - 1]; T.dsize[T.n] = W.h[l]; T.tab[T.n] = (u32*)(ws + W.ytab_off[l]); T.n++;
            maxd = W.w[l] > maxd ? W.w[l] : maxd; maxd = W.h[l] > maxd ? W.h[l] : maxd;
        }
        hipLaunchKernelGGL(k_orb_tables, dim3(div_up(maxd, 256), T.n), dim3(256), 0, st, T);
        VFE_LAUNCH_CHECK();
        for (int l = 1; l < nlevels; l++) {
            const u32* xtab = (const u32*)(ws + W.xtab_off[l]); const u32* ytab = (const u32*)(ws + W.ytab_off[l]);
            const int wn = div_up(W.w[l], RS_PX);
            const unsigned per_frame = (unsigned)wn * (unsigned)W.h[l];
            const DImg& sl = L.raw[l - 1];
            const bool rows_ok = (((uintptr_t)sl.data | (uintptr_t)sl.pitch | (uintptr_t)sl.fstride) & 3) == 0 && sl.sub_n == 1 &&
                                 W.w[l - 1] <= 2 * W.w[l] && W.h[l] <= 65535 && batch <= 65535;
            if (rows_ok) {
                const int gx = div_up(W.w[l], 4 * RS_SEG), gy = div_up(W.h[l], RS_BAND);
                hipLaunchKernelGGL(k_orb_resize_rows, dim3((unsigned)gx * gy * batch), dim3(256), 0, st, sl, L.raw[l], xtab, ytab, gx, gy);
            } else
                hipLaunchKernelGGL(k_orb_resize, dim3((per_frame + 255) / 256, batch), dim3(256), 0, st, sl, L.raw[l], xtab, ytab, wn, per_frame);
            VFE_LAUNCH_CHECK();
        }
    }
    {   /* FAST on all levels at once (levels too small for the 31-pixel border yield nothing) */
        ProfScope prof(VFE_K_FAST, st);
        FastSet F;
        F.n = 0;
        for (int l = 0; l < nlevels; l++) {
            if (W.w[l] > 2 * ORB_EDGE && W.h[l] > 2 * ORB_EDGE) {
                F.img[F.n] = fast_img(L.raw[l]); F.kps[F.n] = (int*)(ws + W.kps_off[l]); F.counts[F.n] = (int*)(ws + W.cnt_off[l]);
                F.cap[F.n] = W.cap[l]; F.n++;
            } else {
                VFE_HIP(hipMemsetAsync(ws + W.cnt_off[l], 0, (size_t)batch * sizeof(int), st));
            }
        }
        if (F.n && (rc = fast_detect_set(F, batch, fast_threshold, ORB_EDGE, ws + W.fast_off, st))) return rc;
    }
    {
        ProfScope prof(VFE_K_ORB_SELECT, st);
        VFE_HIP(hipMemsetAsync(ws + W.hist_off, 0, W.hist_bytes, st));
        hipLaunchKernelGGL(k_orb_hist, dim3(SEL_CHUNKS, nlevels, batch), dim3(256), 0, st, L);
        VFE_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_orb_harris, dim3((unsigned)HARRIS_CHUNKS * nlevels * batch), dim3(64), 0, st, L, nlevels, batch);
        VFE_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_orb_thresh, dim3(nlevels, batch), dim3(1024), 0, st, L, lvl_count);
        VFE_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_orb_emit, dim3(nlevels, batch), dim3(1024), 0, st, L, lvl_count, kps, cap, counts);
        VFE_LAUNCH_CHECK();
    }
    {
        ProfScope prof(VFE_K_ORB_DESC, st);
        const unsigned blocks = (unsigned)batch * (((unsigned)cap + 3u) / 4u);
        hipLaunchKernelGGL(k_orb_desc, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, st, L, kps, desc, counts, cap, batch);
        VFE_LAUNCH_CHECK();
    }
    return VFE_OK;
}
