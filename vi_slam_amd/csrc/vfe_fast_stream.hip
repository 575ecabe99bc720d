/* FAST-9/16 score + 3x3 non-max suppression for whole frames, streaming (cv::FAST semantics, same
 * scores and raster order as the tiled kernel in vfe_fast.hip).
 *
 * One wavefront owns a strip of 62 x 4 columns (lanes 0 and 63 are halo) and walks down a band of rows.
 * A lane keeps the last seven image rows as three dwords each (its own four bytes and the neighbours'),
 * so every pixel of the 16-pixel circle is a byte of a register.  Two horizontally adjacent pixels are
 * scored together in packed 16-bit arithmetic:
 *   score = max(t, max over 9-arcs of min(d), -min over 9-arcs of max(d)) - 1,   d = centre - circle,
 * with the sliding minima / maxima built by doubling (2, 4, 8, 9 wide); a pixel is a corner iff the
 * bracket exceeds t.  Before that, a necessary condition on the four compass pixels (each opposite
 * pair must hold a pixel that differs from the centre by more than t) is evaluated for the whole wave:
 * if no lane passes, the pair is skipped.  Scores of three rows stay in registers for the 3x3
 * suppression; keypoints of a row segment are numbered in raster order with ballots and written to
 * the segment's slot together with the segment's count. */
#include "vfe_common.h"
#include "vfe_fast.h"

typedef unsigned int u32;
typedef short s2v __attribute__((ext_vector_type(2)));
typedef unsigned short u2v __attribute__((ext_vector_type(2)));

__device__ static inline s2v as_s(u32 v) { return __builtin_bit_cast(s2v, v); }
__device__ static inline u32 as_u(s2v v) { return __builtin_bit_cast(u32, v); }
__device__ static inline u2v as_uh(u32 v) { return __builtin_bit_cast(u2v, v); }
__device__ static inline s2v pmin(s2v a, s2v b) { return __builtin_elementwise_min(a, b); }
__device__ static inline s2v pmax(s2v a, s2v b) { return __builtin_elementwise_max(a, b); }
__device__ static inline s2v splat(int v) { return s2v{(short)v, (short)v}; }

/* bytes i and i+1 (1 <= i <= 9) of the 12-byte window {l, c, r} as a pair of 16-bit values */
__device__ static inline s2v win_pair(u32 l, u32 c, u32 r, int i) {
    if (i + 1 <= 7) return as_s(__builtin_amdgcn_perm(c, l, 0x0c000c00u | (u32)i | ((u32)(i + 1) << 16)));
    return as_s(__builtin_amdgcn_perm(r, c, 0x0c000c00u | (u32)(i - 4) | ((u32)(i - 3) << 16)));
}

/* circle position k -> (dx, dy), dy positive = down (the order of cv::FAST's offsets) */
#define FS_DX(k) ((k) == 0 ? 0 : (k) == 1 ? 1 : (k) == 2 ? 2 : (k) == 3 ? 3 : (k) == 4 ? 3 : (k) == 5 ? 3 : (k) == 6 ? 2 : (k) == 7 ? 1 : \
                  (k) == 8 ? 0 : (k) == 9 ? -1 : (k) == 10 ? -2 : (k) == 11 ? -3 : (k) == 12 ? -3 : (k) == 13 ? -3 : (k) == 14 ? -2 : -1)
#define FS_DY(k) ((k) == 0 ? 3 : (k) == 1 ? 3 : (k) == 2 ? 2 : (k) == 3 ? 1 : (k) == 4 ? 0 : (k) == 5 ? -1 : (k) == 6 ? -2 : (k) == 7 ? -3 : \
                  (k) == 8 ? -3 : (k) == 9 ? -3 : (k) == 10 ? -2 : (k) == 11 ? -1 : (k) == 12 ? 0 : (k) == 13 ? 1 : (k) == 14 ? 2 : 3)

/* scores of the pixel pair (px, px+1) of the centre row R[3]; px = 0 or 2 within the lane's four pixels */
template <int PX>
__device__ static inline u32 fast_score_pair(const u32 (&R)[7][3], int t) {
    const s2v v = win_pair(R[3][0], R[3][1], R[3][2], PX + 4);
    /* compass pixels first: a 9-arc holds one pixel of each opposite pair */
    s2v p[16];
#pragma unroll
    for (int k = 0; k < 16; k += 4) p[k] = win_pair(R[3 + FS_DY(k)][0], R[3 + FS_DY(k)][1], R[3 + FS_DY(k)][2], PX + FS_DX(k) + 4);
    bool pass;
    {
        const s2v a0 = pmax(v - p[0], p[0] - v), a8 = pmax(v - p[8], p[8] - v);
        const s2v a4 = pmax(v - p[4], p[4] - v), a12 = pmax(v - p[12], p[12] - v);
        const s2v m = pmin(pmax(a0, a8), pmax(a4, a12));
        pass = m.x > (short)t || m.y > (short)t;
    }
    if (__ballot(pass) == 0) return 0u;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k & 3) p[k] = win_pair(R[3 + FS_DY(k)][0], R[3 + FS_DY(k)][1], R[3 + FS_DY(k)][2], PX + FS_DX(k) + 4);
    s2v d[16], mn[16], mx[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - p[k];
    s2v mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = pmin(d[k], d[(k + 1) & 15]); mx2[k] = pmax(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = pmin(mn2[k], mn2[(k + 2) & 15]); mx4[k] = pmax(mx2[k], mx2[(k + 2) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn[k] = pmin(pmin(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        mx[k] = pmax(pmax(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
    }
    s2v a = mn[0], b = mx[0];
#pragma unroll
    for (int k = 1; k < 16; k++) { a = pmax(a, mn[k]); b = pmin(b, mx[k]); }
    const s2v m = pmax(pmax(a, splat(t)), splat(0) - pmin(b, splat(-t))); /* >= t; > t iff corner */
    const s2v e = m - splat(t);
    /* score = m - 1 for corners, 0 otherwise */
    return as_u(pmin(e, splat(1)) * (e + splat(t - 1)));
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_fast_stream(DImg img, int threshold, int border, int strips_x, int band_h, int bands_y,
                                                     unsigned total_work, uint8_t* __restrict__ rowcnt, u32* __restrict__ tmp) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    const int W = img.w, H = img.h;
    const unsigned per_frame = (unsigned)(strips_x * bands_y);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (unsigned work = wave; work < total_work; work += nwaves) {
        const int b = work / per_frame, t_item = work - (unsigned)b * per_frame;
        const int sxi = t_item % strips_x, byi = t_item / strips_x;
        const int X0 = sxi * FS_OUT + 4 * (lane - 1); /* image column of this lane's pixel 0 */
        const int y0 = byi * band_h, y1 = min(y0 + band_h, H);
        const uint8_t* ip = img.frame(b);
        const bool fast_load = ALIGNED && X0 >= 0 && X0 + 3 < W;
        const bool out_lane = lane >= 1 && lane < 63;
        /* per pixel: may be scored (circle inside the image), may be reported */
        u32 score_mask = 0; /* 0xFFFF per scorable pixel, as two pair masks */
        u32 smask[2] = {0, 0};
        bool rep[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = X0 + k;
            if (x >= 3 && x < W - 3) smask[k >> 1] |= 0xFFFFu << (16 * (k & 1));
            rep[k] = out_lane && x >= border && x < W - border && x < W;
        }
        (void)score_mask;

        u32 R[7][3];  /* image rows yy-6 .. yy: left neighbour's, own, right neighbour's dword */
        u32 S[3][3];  /* score rows N-2 .. N: pixel pairs (0,1), (2,3) as 16-bit values, and s[-1] | s[4] << 16 */
#pragma unroll
        for (int i = 0; i < 7; i++) R[i][0] = R[i][1] = R[i][2] = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) S[i][0] = S[i][1] = S[i][2] = 0;

        const int T = (y1 - y0) + 8; /* image rows y0-4 .. y1+3 */
        for (int t = 0; t < T; t++) {
            const int yy = y0 - 4 + t;
#pragma unroll
            for (int i = 0; i < 6; i++) { R[i][0] = R[i + 1][0]; R[i][1] = R[i + 1][1]; R[i][2] = R[i + 1][2]; }
            {
                /* rows and columns outside the image are never part of a scored circle: clamp */
                const uint8_t* row = ip + (size_t)min(max(yy, 0), H - 1) * img.pitch;
                u32 c;
                if (fast_load) c = *reinterpret_cast<const uint32_t*>(row + X0);
                else c = (u32)row[min(max(X0, 0), W - 1)] | ((u32)row[min(max(X0 + 1, 0), W - 1)] << 8) |
                         ((u32)row[min(max(X0 + 2, 0), W - 1)] << 16) | ((u32)row[min(max(X0 + 3, 0), W - 1)] << 24);
                R[6][1] = c;
                R[6][0] = (u32)__shfl_up((int)c, 1);
                R[6][2] = (u32)__shfl_down((int)c, 1);
            }
            if (t < 6) continue;
            /* ---- scores of row N = yy - 3 ---- */
            const int N = yy - 3;
#pragma unroll
            for (int i = 0; i < 2; i++) { S[i][0] = S[i + 1][0]; S[i][1] = S[i + 1][1]; S[i][2] = S[i + 1][2]; }
            {
                u32 s01 = 0, s23 = 0;
                if (N >= 3 && N < H - 3) { /* wave-uniform */
                    s01 = fast_score_pair<0>(R, threshold) & smask[0];
                    s23 = fast_score_pair<2>(R, threshold) & smask[1];
                }
                S[2][0] = s01; S[2][1] = s23;
                const u32 sl = (u32)__shfl_up((int)s23, 1) >> 16, sr = (u32)__shfl_down((int)s01, 1) & 0xFFFFu;
                S[2][2] = sl | (sr << 16);
            }
            if (t < 8) continue;
            /* ---- suppression and output of row M = yy - 4 (score rows M-1, M, M+1 = S[0], S[1], S[2]) ---- */
            const int M = yy - 4;
            const size_t seg = ((size_t)b * H + M) * strips_x + sxi;
            bool kp[4] = {false, false, false, false};
            if (__ballot((S[1][0] | S[1][1]) != 0) != 0) {
                u2v nb01, nb23; /* largest of the eight neighbours */
                {
                    u2v h01[3], h23[3];
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const u32 p01 = S[i][0], p23 = S[i][1], e = S[i][2];
                        const u2v L01 = as_uh(__builtin_amdgcn_perm(p01, e, 0x05040100u));    /* (s[-1], s0) */
                        const u2v M12 = as_uh(__builtin_amdgcn_perm(p23, p01, 0x05040302u));  /* (s1, s2) */
                        const u2v R23 = as_uh(__builtin_amdgcn_perm(e, p23, 0x07060302u));    /* (s3, s4) */
                        if (i == 1) {
                            h01[i] = __builtin_elementwise_max(L01, M12);
                            h23[i] = __builtin_elementwise_max(M12, R23);
                        } else {
                            h01[i] = __builtin_elementwise_max(__builtin_elementwise_max(L01, M12), as_uh(p01));
                            h23[i] = __builtin_elementwise_max(__builtin_elementwise_max(M12, R23), as_uh(p23));
                        }
                    }
                    nb01 = __builtin_elementwise_max(__builtin_elementwise_max(h01[0], h01[1]), h01[2]);
                    nb23 = __builtin_elementwise_max(__builtin_elementwise_max(h23[0], h23[1]), h23[2]);
                }
                const u2v c01 = as_uh(S[1][0]), c23 = as_uh(S[1][1]);
                const bool rowok = M >= border && M < H - border;
                kp[0] = rowok && rep[0] && c01.x > nb01.x;
                kp[1] = rowok && rep[1] && c01.y > nb01.y;
                kp[2] = rowok && rep[2] && c23.x > nb23.x;
                kp[3] = rowok && rep[3] && c23.y > nb23.y;
            }
            const unsigned long long m0 = __ballot(kp[0]), m1 = __ballot(kp[1]), m2 = __ballot(kp[2]), m3 = __ballot(kp[3]);
            const int total = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
            if (total) {
                int pos = __popcll(m0 & lt_mask) + __popcll(m1 & lt_mask) + __popcll(m2 & lt_mask) + __popcll(m3 & lt_mask);
                u32* dst = tmp + seg * FS_ROWCAP;
                const u32 sc[4] = {S[1][0] & 0xFFFFu, S[1][0] >> 16, S[1][1] & 0xFFFFu, S[1][1] >> 16};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (kp[k]) { dst[pos] = (u32)(4 * (lane - 1) + k) | (sc[k] << 8); pos++; }
            }
            if (lane == 0) rowcnt[seg] = (uint8_t)total;
        }
    }
}

/* one thread per (frame, row, strip) segment copies the segment's keypoints to their final offset */
__global__ __launch_bounds__(256) void k_fast_scatter_stream(int H, int strips_x, unsigned n_seg, const uint8_t* __restrict__ rowcnt,
                                                             const u32* __restrict__ rowoff, const u32* __restrict__ tmp,
                                                             int* __restrict__ kps, int cap) {
    const unsigned seg = blockIdx.x * 256u + threadIdx.x;
    if (seg >= n_seg) return;
    const int n = rowcnt[seg];
    if (n == 0) return;
    const unsigned per_frame = (unsigned)(H * strips_x);
    const int b = seg / per_frame;
    const unsigned r = seg - (unsigned)b * per_frame;
    const int y = r / strips_x, sxi = r - y * strips_x;
    const u32 off0 = rowoff[seg];
    for (int j = 0; j < n; j++) {
        const u32 e = tmp[(size_t)seg * FS_ROWCAP + j];
        const u32 off = off0 + j;
        if (off < (u32)cap) {
            int* o = kps + ((size_t)b * cap + off) * 3;
            o[0] = sxi * FS_OUT + (int)(e & 255u); o[1] = y; o[2] = (int)(e >> 8);
        }
    }
}

int fast_stream_launch(const DImg& d, int threshold, int border, uint8_t* rowcnt, u32* tmp, hipStream_t st) {
    const int strips_x = div_up(d.w, FS_OUT);
    int band_h = 64;
    while (band_h > 16 && (long long)strips_x * div_up(d.h, band_h) * d.batch < 16384) band_h /= 2;
    const int bands_y = div_up(d.h, band_h);
    const unsigned total = (unsigned)(strips_x * bands_y * d.batch), nblk = (total + 3) / 4;
    const bool aligned = ((uintptr_t)d.data % 4 == 0) && (d.pitch % 4 == 0) && (d.fstride % 4 == 0);
    const dim3 grid(nblk < (1u << 20) ? nblk : (1u << 20));
    if (aligned) hipLaunchKernelGGL(k_fast_stream<true>, grid, dim3(256), 0, st, d, threshold, border, strips_x, band_h, bands_y, total, rowcnt, tmp);
    else hipLaunchKernelGGL(k_fast_stream<false>, grid, dim3(256), 0, st, d, threshold, border, strips_x, band_h, bands_y, total, rowcnt, tmp);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

int fast_scatter_stream_launch(const DImg& d, const uint8_t* rowcnt, const u32* rowoff, const u32* tmp, int* kps, int cap, hipStream_t st) {
    const int strips_x = div_up(d.w, FS_OUT);
    const unsigned n_seg = (unsigned)d.batch * (unsigned)d.h * (unsigned)strips_x;
    hipLaunchKernelGGL(k_fast_scatter_stream, dim3((n_seg + 255) / 256), dim3(256), 0, st, d.h, strips_x, n_seg, rowcnt, rowoff, tmp, kps, cap);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
