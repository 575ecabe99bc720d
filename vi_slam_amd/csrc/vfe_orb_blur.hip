/* 7x7 Gaussian blur, sigma 2, as cv::GaussianBlur computes it for 8-bit images (8.8 fixed-point
 * taps 18 34 48 56 48 34 18, horizontal sums kept in 16 bits, one rounding after the vertical pass,
 * reflect-101 borders) -- the smoothing cv::ORB applies before sampling the rBRIEF pairs
 * (reference call site: src/slam/adapters/orbslam3_adapter.cpp:76-79).
 *
 * Streaming, no LDS: a wave owns a strip of 62 x 4 output columns (lanes 0 and 63 are halo) and walks
 * down a band of rows.  Per source row a lane loads one dword, gets the three bytes it needs on either
 * side from its neighbours by shuffles, forms its four horizontal sums as two packed 16-bit pairs and
 * keeps the last seven rows of sums in a register ring; from the seventh row on every step emits one
 * output row.  The loop is unrolled seven times so that the ring slots are compile-time constants.  One launch
 * serves a set of images of different sizes (the levels of an ORB pyramid). */
#include "vfe_common.h"
#include <type_traits>

typedef unsigned int u32;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <int N> using ic = std::integral_constant<int, N>;

#define GB_OUT 248 /* output columns per strip */

__device__ static inline u16x2 as_h2(u32 v) { return __builtin_bit_cast(u16x2, v); }
__device__ static inline u32 as_u(u16x2 v) { return __builtin_bit_cast(u32, v); }
/* (byte i of the 8 bytes {hi:lo}) | (byte j) << 16 */
__device__ static inline u32 byte_pair(u32 lo, u32 hi, int i, int j) {
    return __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (u32)i | ((u32)j << 16));
}

#define GB_MAX_LEVELS 12
struct BlurSet {
    int n, band_h;
    DImg src[GB_MAX_LEVELS], dst[GB_MAX_LEVELS];
    int strips_x[GB_MAX_LEVELS], bands_y[GB_MAX_LEVELS], aligned[GB_MAX_LEVELS];
    unsigned item_begin[GB_MAX_LEVELS + 1]; /* first (frame, band, strip) work item of an image */
};

__global__ __launch_bounds__(256) void k_orb_blur(BlurSet S) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4u; /* wave-uniform */
    const unsigned total_work = S.item_begin[S.n];
    const int band_h = S.band_h;
    for (unsigned work = wave; work < total_work; work += nwaves) {
        int lv = 0;
        while (lv + 1 < S.n && work >= S.item_begin[lv + 1]) lv++;
        const DImg& src = S.src[lv];
        DImg& dst = S.dst[lv];
        const bool ALIGNED = S.aligned[lv] != 0;
        const int strips_x = S.strips_x[lv];
        const int W = src.w, H = src.h;
        const unsigned per_frame = (unsigned)(strips_x * S.bands_y[lv]);
        const unsigned item = work - S.item_begin[lv];
        const int b = item / per_frame, t_item = item - (unsigned)b * per_frame;
        const int sxi = t_item % strips_x, byi = t_item / strips_x;
        const int X0 = sxi * GB_OUT + 4 * (lane - 1); /* image column of this lane's pixel 0 */
        const int y0 = byi * band_h, y1 = min(y0 + band_h, H);
        const uint8_t* ip = src.frame(b);
        uint8_t* op = dst.frame(b);
        const bool fast_load = ALIGNED && X0 >= 0 && X0 + 3 < W;
        const bool dead_load = X0 + 3 < -3 || X0 > W + 2;
        const bool out_lane = lane >= 1 && lane < 63 && X0 < W;
        const bool fast_store = ALIGNED && X0 + 3 < W;

        u32 HR[7][2]; /* horizontal sums of the last seven rows: pixel pairs (0,1) and (2,3), 16 bits each */
#pragma unroll
        for (int i = 0; i < 7; i++) HR[i][0] = HR[i][1] = 0;

        /* the lane's four pixels of source row yy (reflect-101) */
        auto load_row = [&](int yy) -> u32 {
            const uint8_t* row = ip + (size_t)reflect101(yy, H) * src.pitch;
            if (fast_load) return *reinterpret_cast<const uint32_t*>(row + X0);
            if (dead_load) return 0u;
            return (u32)row[reflect101(X0, W)] | ((u32)row[reflect101(X0 + 1, W)] << 8) |
                   ((u32)row[reflect101(X0 + 2, W)] << 16) | ((u32)row[reflect101(X0 + 3, W)] << 24);
        };
        u32 pre = load_row(y0 - 3);

        /* stc: 1 = fill the ring only, 2 = also emit output row yy - 3 */
        auto step = [&](auto phc, auto stc, int t) {
            constexpr int ph = decltype(phc)::value, ST = decltype(stc)::value;
            const int yy = y0 - 3 + t;
            {
                /* the row was requested a step ago; request the next one now (always a valid address) */
                const u32 c = pre;
                pre = load_row(yy + 1);
                const u32 l = lane_up_u(c), r = lane_down_u(c);
                /* bytes of columns X0-3 .. X0+6 are l1 l2 l3 c0 c1 c2 c3 r0 r1 r2 =: q0 .. q9; pixel k taps
                 * q[k] .. q[k+6].  Pair P[j] = (q[j], q[j+1]) serves pixels (0,1) as tap j and, as P[j+2],
                 * pixels (2,3) */
                u16x2 P[9];
                P[0] = as_h2(byte_pair(l, l, 1, 2)); P[1] = as_h2(byte_pair(l, l, 2, 3)); P[2] = as_h2(byte_pair(l, c, 3, 4));
                P[3] = as_h2(byte_pair(c, c, 0, 1)); P[4] = as_h2(byte_pair(c, c, 1, 2)); P[5] = as_h2(byte_pair(c, c, 2, 3));
                P[6] = as_h2(byte_pair(c, r, 3, 4)); P[7] = as_h2(byte_pair(r, r, 0, 1)); P[8] = as_h2(byte_pair(r, r, 1, 2));
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const u16x2* Q = P + 2 * p;
                    /* 18 (q0+q6) + 34 (q1+q5) + 48 (q2+q4) + 56 q3 <= 65280: fits 16 bits */
                    HR[ph][p] = as_u((Q[0] + Q[6]) * (unsigned short)18 + (Q[1] + Q[5]) * (unsigned short)34 +
                                     (Q[2] + Q[4]) * (unsigned short)48 + Q[3] * (unsigned short)56);
                }
            }
            if constexpr (ST < 2) return;
            const int N = yy - 3;
            u32 o[4];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                /* rows N-3 .. N+3 sit in slots ph+1 .. ph+7 (mod 7) */
                const u32 r0 = HR[(ph + 1) % 7][p], r1 = HR[(ph + 2) % 7][p], r2 = HR[(ph + 3) % 7][p], r3 = HR[(ph + 4) % 7][p];
                const u32 r4 = HR[(ph + 5) % 7][p], r5 = HR[(ph + 6) % 7][p], r6 = HR[ph][p];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int sh = 16 * e;
                    const u32 v = 18u * (((r0 >> sh) & 0xFFFFu) + ((r6 >> sh) & 0xFFFFu)) + 34u * (((r1 >> sh) & 0xFFFFu) + ((r5 >> sh) & 0xFFFFu)) +
                                  48u * (((r2 >> sh) & 0xFFFFu) + ((r4 >> sh) & 0xFFFFu)) + 56u * ((r3 >> sh) & 0xFFFFu);
                    o[2 * p + e] = (v + (1u << 15)) >> 16;
                }
            }
            if (out_lane) {
                uint8_t* orow = op + (size_t)N * dst.pitch + X0;
                if (fast_store) *reinterpret_cast<uint32_t*>(orow) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
                else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (X0 + k < W) orow[k] = (uint8_t)o[k];
                }
            }
        };

        /* rows y0-3 .. y1+2: six steps fill the ring, then every step emits one output row */
        step(ic<0>{}, ic<1>{}, 0); step(ic<1>{}, ic<1>{}, 1); step(ic<2>{}, ic<1>{}, 2);
        step(ic<3>{}, ic<1>{}, 3); step(ic<4>{}, ic<1>{}, 4); step(ic<5>{}, ic<1>{}, 5);
        const int T = 6 + (y1 - y0);
        int t0 = 6;
        for (; t0 + 6 < T; t0 += 7) {
            step(ic<6>{}, ic<2>{}, t0); step(ic<0>{}, ic<2>{}, t0 + 1); step(ic<1>{}, ic<2>{}, t0 + 2); step(ic<2>{}, ic<2>{}, t0 + 3);
            step(ic<3>{}, ic<2>{}, t0 + 4); step(ic<4>{}, ic<2>{}, t0 + 5); step(ic<5>{}, ic<2>{}, t0 + 6);
        }
        if (t0 < T) step(ic<6>{}, ic<2>{}, t0);
        if (t0 + 1 < T) step(ic<0>{}, ic<2>{}, t0 + 1);
        if (t0 + 2 < T) step(ic<1>{}, ic<2>{}, t0 + 2);
        if (t0 + 3 < T) step(ic<2>{}, ic<2>{}, t0 + 3);
        if (t0 + 4 < T) step(ic<3>{}, ic<2>{}, t0 + 4);
        if (t0 + 5 < T) step(ic<4>{}, ic<2>{}, t0 + 5);
    }
}

/* blurs n images of different sizes (same number of frames each) with one launch */
int orb_blur_launch(const DImg* src, const DImg* dst, int n, hipStream_t st) {
    if (n < 1 || n > GB_MAX_LEVELS) { vfe_set_error("blur: 1..%d images per launch", GB_MAX_LEVELS); return VFE_E_INVALID; }
    BlurSet S;
    S.n = n;
    int band_h = 64;
    for (; band_h > 16; band_h /= 2) {
        long long items = 0;
        for (int l = 0; l < n; l++) items += (long long)div_up(src[l].w, GB_OUT) * div_up(src[l].h, band_h) * src[l].batch;
        if (items >= 16384) break;
    }
    S.band_h = band_h;
    unsigned long long items = 0;
    for (int l = 0; l < n; l++) {
        S.src[l] = src[l]; S.dst[l] = dst[l];
        S.strips_x[l] = div_up(src[l].w, GB_OUT); S.bands_y[l] = div_up(src[l].h, band_h);
        S.aligned[l] = ((uintptr_t)src[l].data % 4 == 0) && (src[l].pitch % 4 == 0) && (src[l].fstride % 4 == 0) &&
                       ((uintptr_t)dst[l].data % 4 == 0) && (dst[l].pitch % 4 == 0) && (dst[l].fstride % 4 == 0) && src[l].sub_n == 1;
        S.item_begin[l] = (unsigned)items;
        items += (unsigned long long)S.strips_x[l] * S.bands_y[l] * src[l].batch;
    }
    if (items >= (1ull << 31)) { vfe_set_error("blur: image set too large"); return VFE_E_INVALID; }
    S.item_begin[n] = (unsigned)items;
    const unsigned nblk = ((unsigned)items + 3) / 4;
    hipLaunchKernelGGL(k_orb_blur, dim3(nblk < (1u << 20) ? nblk : (1u << 20)), dim3(256), 0, st, S);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
