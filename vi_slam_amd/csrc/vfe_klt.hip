/* Pyramidal Lucas-Kanade tracking (cv::calcOpticalFlowPyrLK semantics) for gfx950, batched.
 *
 * One 64-lane wavefront tracks one point through all pyramid levels.  Per level the wave
 * stages the (win+3)^2 source patch of the previous image in LDS (reflect-101 resolved on
 * load), derives the Scharr gradients of the patch in LDS instead of reading a derivative
 * image from HBM, builds the bilinear template (14-bit integer weights, as OpenCV) and the
 * 2x2 Gram matrix, then iterates: stage the (win+1)^2 patch of the next image, accumulate the
 * mismatch vector, solve, move.  All window sums are exact integers (per-lane int32 partials,
 * cross-lane butterfly in int64), so the float part of the algorithm sees the same inputs as
 * the CPU oracle and positions agree bit for bit.
 *
 * HBM traffic is only the patches actually touched; the pyramids themselves are shared with
 * the detector.
 */
#include "vfe_common.h"
#include <float.h>

#define KLT_MAX_LEVELS 8

struct KltParams {
    DImg prev[KLT_MAX_LEVELS], next[KLT_MAX_LEVELS];
    int nlevels;
    const float* prev_pts;
    const int* n_pts;
    int cap;
    float* next_pts;
    uint8_t* status;
    float* err;
    int win, max_iter;
    double epsilon; /* squared */
    double min_eig_thr;
    int use_initial;
    unsigned total_work;
};

__device__ static inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

__device__ static inline long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ static inline void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
    w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << 14));
    w01 = __float2int_rn(a * (1.f - b) * (float)(1 << 14));
    w10 = __float2int_rn((1.f - a) * b * (float)(1 << 14));
    w11 = (1 << 14) - w00 - w01 - w10;
}

__global__ __launch_bounds__(64) void k_klt(KltParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x;
    const int win = p.win, PW = win + 3, DW = win + 1;
    uint8_t* s_patch = smem;                                                    /* PW*PW bytes */
    short2* s_deriv = reinterpret_cast<short2*>(smem + ((PW * PW + 15) & ~15)); /* DW*DW */
    short* s_I = reinterpret_cast<short*>(s_deriv + DW * DW);                   /* win*win */
    short2* s_dI = reinterpret_cast<short2*>(s_I + ((win * win + 1) & ~1));     /* win*win */

    /* persistent single-wave workgroups striding over all (frame, point) items */
    for (unsigned work = blockIdx.x; work < p.total_work; work += gridDim.x) {
    const int b = work / (unsigned)p.cap, pt = work - (unsigned)b * (unsigned)p.cap;
    const size_t o = (size_t)b * p.cap + pt;
    const int n = p.n_pts ? min(p.n_pts[b], p.cap) : p.cap;
    if (pt >= n) {
        if (lane == 0) { p.status[o] = 0; if (p.err) p.err[o] = 0.f; }
        continue;
    }
    const float ppx = p.prev_pts[2 * o], ppy = p.prev_pts[2 * o + 1];
    float outx = p.use_initial ? p.next_pts[2 * o] : 0.f, outy = p.use_initial ? p.next_pts[2 * o + 1] : 0.f;
    int status = 1;
    float errv = 0.f;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int max_level = p.nlevels - 1;

    for (int level = max_level; level >= 0; level--) {
        const DImg& I = p.prev[level];
        const DImg& J = p.next[level];
        const uint8_t* ip = I.frame(b);
        const uint8_t* jp = J.frame(b);
        const float sc = (float)(1. / (double)(1 << level));
        float px = ppx * sc, py = ppy * sc, nx, ny;
        if (level == max_level) {
            if (p.use_initial) { nx = outx * sc; ny = outy * sc; }
            else { nx = px; ny = py; }
        } else { nx = outx * 2.f; ny = outy * 2.f; }
        outx = nx; outy = ny;

        px -= half; py -= half;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
            if (level == 0) { status = 0; errv = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        bilinear_weights(px - (float)ipx, py - (float)ipy, iw00, iw01, iw10, iw11);

        __syncthreads();
        for (int i = lane; i < PW * PW; i += 64) {
            const int r = i / PW, c = i - r * PW;
            s_patch[i] = ip[(size_t)reflect101(ipy - 1 + r, I.h) * I.pitch + reflect101(ipx - 1 + c, I.w)];
        }
        __syncthreads();
        /* Scharr gradient at the (win+1)^2 positions the bilinear template touches; zero outside the image */
        for (int i = lane; i < DW * DW; i += 64) {
            const int r = i / DW, c = i - r * DW;
            const int gx = ipx + c, gy = ipy + r;
            short2 d = make_short2(0, 0);
            if (gx >= 0 && gy >= 0 && gx < I.w && gy < I.h) {
                const uint8_t* q = s_patch + r * PW + c;
                const int a00 = q[0], a01 = q[1], a02 = q[2];
                const int a10 = q[PW], a12 = q[PW + 2];
                const int a20 = q[2 * PW], a21 = q[2 * PW + 1], a22 = q[2 * PW + 2];
                d.x = (short)(3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20));
                d.y = (short)(3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02));
            }
            s_deriv[i] = d;
        }
        __syncthreads();
        int a11 = 0, a12 = 0, a22 = 0;
        for (int i = lane; i < win * win; i += 64) {
            const int y = i / win, x = i - y * win;
            const uint8_t* q = s_patch + (y + 1) * PW + (x + 1);
            const int ival = descale(q[0] * iw00 + q[1] * iw01 + q[PW] * iw10 + q[PW + 1] * iw11, 14 - 5);
            const short2 d00 = s_deriv[y * DW + x], d01 = s_deriv[y * DW + x + 1];
            const short2 d10 = s_deriv[(y + 1) * DW + x], d11 = s_deriv[(y + 1) * DW + x + 1];
            const int ixval = descale(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
            const int iyval = descale(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
            s_I[i] = (short)ival;
            s_dI[i] = make_short2((short)ixval, (short)iyval);
            a11 += ixval * ixval; a12 += ixval * iyval; a22 += iyval * iyval;
        }
        const long long iA11 = wave_sum_i64(a11), iA12 = wave_sum_i64(a12), iA22 = wave_sum_i64(a22);
        const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
        if ((double)minEig < p.min_eig_thr || D < FLT_EPSILON) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < p.max_iter; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            bilinear_weights(nx - (float)inx, ny - (float)iny, iw00, iw01, iw10, iw11);
            __syncthreads();
            for (int i = lane; i < DW * DW; i += 64) {
                const int r = i / DW, c = i - r * DW;
                s_patch[i] = jp[(size_t)reflect101(iny + r, J.h) * J.pitch + reflect101(inx + c, J.w)];
            }
            __syncthreads();
            int b1 = 0, b2 = 0;
            for (int i = lane; i < win * win; i += 64) {
                const int y = i / win, x = i - y * win;
                const uint8_t* q = s_patch + y * DW + x;
                const int diff = descale(q[0] * iw00 + q[1] * iw01 + q[DW] * iw10 + q[DW + 1] * iw11, 14 - 5) - s_I[i];
                const short2 d = s_dI[i];
                b1 += diff * d.x; b2 += diff * d.y;
            }
            const long long ib1 = wave_sum_i64(b1), ib2 = wave_sum_i64(b2);
            const float fb1 = (float)ib1 * FLT_SCALE, fb2 = (float)ib2 * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= p.epsilon) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {
            const float fx = outx - half, fy = outy - half;
            const int inx = (int)floorf(fx), iny = (int)floorf(fy);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { status = 0; continue; }
            bilinear_weights(fx - (float)inx, fy - (float)iny, iw00, iw01, iw10, iw11);
            __syncthreads();
            for (int i = lane; i < DW * DW; i += 64) {
                const int r = i / DW, c = i - r * DW;
                s_patch[i] = jp[(size_t)reflect101(iny + r, J.h) * J.pitch + reflect101(inx + c, J.w)];
            }
            __syncthreads();
            int es = 0;
            for (int i = lane; i < win * win; i += 64) {
                const int y = i / win, x = i - y * win;
                const uint8_t* q = s_patch + y * DW + x;
                const int diff = descale(q[0] * iw00 + q[1] * iw01 + q[DW] * iw10 + q[DW + 1] * iw11, 14 - 5) - s_I[i];
                es += abs(diff);
            }
            const long long esum = wave_sum_i64(es);
            errv = (float)esum * 1.f / (float)(32 * win * win);
        }
    }
    if (lane == 0) {
        p.next_pts[2 * o] = outx; p.next_pts[2 * o + 1] = outy;
        p.status[o] = (uint8_t)status;
        if (p.err) p.err[o] = errv;
    }
    }
}

/* ---- 21x21 window, register-resident ---------------------------------------------------------
 * The common window size gets its own kernel: lane l < 63 owns 7 consecutive pixels of window
 * row l/3 (21 rows x 3 segments = 63 lanes), keeps its share of the template (I, Ix, Iy) in
 * registers for the whole level and reads the image rows it needs straight from global memory
 * with unaligned dword loads (the patches are L1/L2 resident after the first touch) -- no LDS,
 * no barriers.  Window sums are reduced with DPP row operations in two 16-bit halves so that
 * they stay exact. */
__device__ static inline int dpp_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);  /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);  /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); /* row_ror 4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); /* row_ror 8 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); /* row_bcast 15 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); /* row_bcast 31 */
    return __builtin_amdgcn_readlane(v, 63);
}
/* exact sum over the wave of per-lane values with |v| < 2^31 */
__device__ static inline long long wave_sum_exact(int v) {
    const int lo = dpp_sum(v & 0xFFFF), hi = dpp_sum(v >> 16);
    return ((long long)hi << 16) + (long long)lo;
}

/* 10 (or 8) consecutive bytes of an image row starting at column x0, as ints */
template <int N>
__device__ static inline void load_row_fast(const uint8_t* row, int x0, int* out) {
    uint32_t w[(N + 3) / 4];
#pragma unroll
    for (int k = 0; k < (N + 3) / 4; k++) __builtin_memcpy(&w[k], row + x0 + 4 * k, 4);
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = (int)((w[k >> 2] >> (8 * (k & 3))) & 255u);
}
template <int N>
__device__ static inline void load_row_reflect(const uint8_t* img, int pitch, int W, int H, int y, int x0, int* out) {
    const uint8_t* row = img + (size_t)reflect101(y, H) * pitch;
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = row[reflect101(x0 + k, W)];
}

__global__ __launch_bounds__(256) void k_klt21(KltParams p) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    const int r = lane / 3, x0 = (lane - 3 * r) * 7; /* lane 63: r = 21, contributes nothing */
    const bool active = lane < 63;
    const int win = 21;
    const float half = 10.f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int max_level = p.nlevels - 1;

    for (unsigned work = wave; work < p.total_work; work += nwaves) {
        const int b = work / (unsigned)p.cap, pt = work - (unsigned)b * (unsigned)p.cap;
        const size_t o = (size_t)b * p.cap + pt;
        const int n = p.n_pts ? min(p.n_pts[b], p.cap) : p.cap;
        if (pt >= n) {
            if (lane == 0) { p.status[o] = 0; if (p.err) p.err[o] = 0.f; }
            continue;
        }
        const float ppx = p.prev_pts[2 * o], ppy = p.prev_pts[2 * o + 1];
        float outx = p.use_initial ? p.next_pts[2 * o] : 0.f, outy = p.use_initial ? p.next_pts[2 * o + 1] : 0.f;
        int status = 1;
        float errv = 0.f;

        for (int level = max_level; level >= 0; level--) {
            const DImg& I = p.prev[level];
            const DImg& J = p.next[level];
            const uint8_t* ip = I.frame(b);
            const uint8_t* jp = J.frame(b);
            const float sc = (float)(1. / (double)(1 << level));
            float px = ppx * sc, py = ppy * sc, nx, ny;
            if (level == max_level) {
                if (p.use_initial) { nx = outx * sc; ny = outy * sc; }
                else { nx = px; ny = py; }
            } else { nx = outx * 2.f; ny = outy * 2.f; }
            outx = nx; outy = ny;

            px -= half; py -= half;
            const int ipx = (int)floorf(px), ipy = (int)floorf(py);
            if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
                if (level == 0) { status = 0; errv = 0.f; }
                continue;
            }
            int iw00, iw01, iw10, iw11;
            bilinear_weights(px - (float)ipx, py - (float)ipy, iw00, iw01, iw10, iw11);

            /* rows ipy+r-1 .. ipy+r+2, columns ipx+x0-1 .. ipx+x0+8 of the previous image */
            int R[4][10];
            const bool interior = ipx >= 1 && ipy >= 1 && ipx + win + 4 < I.w && ipy + win + 2 < I.h;
            if (active) {
                if (interior) {
#pragma unroll
                    for (int i = 0; i < 4; i++) load_row_fast<10>(ip + (size_t)(ipy + r - 1 + i) * I.pitch, ipx + x0 - 1, R[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) load_row_reflect<10>(ip, I.pitch, I.w, I.h, ipy + r - 1 + i, ipx + x0 - 1, R[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 10; j++) R[i][j] = 0;
            }
            /* Scharr at rows i = 1, 2 and columns j = 1..8 of the 4x10 block */
            int tI[7], tIx[7], tIy[7];
            int a11 = 0, a12 = 0, a22 = 0;
            {
                int gx[2][8], gy[2][8];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    int S[10], D[10];
#pragma unroll
                    for (int j = 0; j < 10; j++) { S[j] = 3 * (R[i][j] + R[i + 2][j]) + 10 * R[i + 1][j]; D[j] = R[i + 2][j] - R[i][j]; }
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        int vx = S[j + 2] - S[j];
                        int vy = 3 * (D[j] + D[j + 2]) + 10 * D[j + 1];
                        if (!interior) { /* derivative image is zero outside the image */
                            const int cx = ipx + x0 + j, cy = ipy + r + i;
                            if (cx < 0 || cy < 0 || cx >= I.w || cy >= I.h) { vx = 0; vy = 0; }
                        }
                        gx[i][j] = vx; gy[i][j] = vy;
                    }
                }
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const int ival = descale(R[1][k + 1] * iw00 + R[1][k + 2] * iw01 + R[2][k + 1] * iw10 + R[2][k + 2] * iw11, 14 - 5);
                    const int ixval = descale(gx[0][k] * iw00 + gx[0][k + 1] * iw01 + gx[1][k] * iw10 + gx[1][k + 1] * iw11, 14);
                    const int iyval = descale(gy[0][k] * iw00 + gy[0][k + 1] * iw01 + gy[1][k] * iw10 + gy[1][k + 1] * iw11, 14);
                    tI[k] = active ? ival : 0; tIx[k] = active ? ixval : 0; tIy[k] = active ? iyval : 0;
                    a11 += tIx[k] * tIx[k]; a12 += tIx[k] * tIy[k]; a22 += tIy[k] * tIy[k];
                }
            }
            const long long iA11 = wave_sum_exact(a11), iA12 = wave_sum_exact(a12), iA22 = wave_sum_exact(a22);
            const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
            if ((double)minEig < p.min_eig_thr || D < FLT_EPSILON) {
                if (level == 0) status = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j <= p.max_iter; j++) {
                /* j == max_iter is the pass that only evaluates the residual at the final position */
                float qx = nx, qy = ny;
                const bool err_pass = (j == p.max_iter);
                if (err_pass) {
                    if (!(status && level == 0)) break;
                    qx = outx - half; qy = outy - half;
                }
                const int inx = (int)floorf(qx), iny = (int)floorf(qy);
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                    if (level == 0) status = 0;
                    break;
                }
                bilinear_weights(qx - (float)inx, qy - (float)iny, iw00, iw01, iw10, iw11);
                int Q[2][8];
                const bool jin = inx >= 0 && iny >= 0 && inx + win + 1 < J.w && iny + win + 1 < J.h;
                if (active) {
                    if (jin) {
                        load_row_fast<8>(jp + (size_t)(iny + r) * J.pitch, inx + x0, Q[0]);
                        load_row_fast<8>(jp + (size_t)(iny + r + 1) * J.pitch, inx + x0, Q[1]);
                    } else {
                        load_row_reflect<8>(jp, J.pitch, J.w, J.h, iny + r, inx + x0, Q[0]);
                        load_row_reflect<8>(jp, J.pitch, J.w, J.h, iny + r + 1, inx + x0, Q[1]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) { Q[0][k] = 0; Q[1][k] = 0; }
                }
                int b1 = 0, b2 = 0, es = 0;
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const int diff = descale(Q[0][k] * iw00 + Q[0][k + 1] * iw01 + Q[1][k] * iw10 + Q[1][k + 1] * iw11, 14 - 5) - tI[k];
                    b1 += diff * tIx[k]; b2 += diff * tIy[k];
                    es += active ? abs(diff) : 0;
                }
                if (err_pass) {
                    const long long esum = wave_sum_exact(es);
                    errv = (float)esum * 1.f / (float)(32 * win * win);
                    break;
                }
                const long long ib1 = wave_sum_exact(b1), ib2 = wave_sum_exact(b2);
                const float fb1 = (float)ib1 * FLT_SCALE, fb2 = (float)ib2 * FLT_SCALE;
                const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
                nx += dx; ny += dy;
                outx = nx + half; outy = ny + half;
                bool done = ((double)dx * (double)dx + (double)dy * (double)dy <= p.epsilon);
                if (!done && j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                    outx -= dx * 0.5f; outy -= dy * 0.5f;
                    done = true;
                }
                pdx = dx; pdy = dy;
                if (done || j == p.max_iter - 1) j = p.max_iter - 1; /* next pass is the residual pass */
            }
        }
        if (lane == 0) {
            p.next_pts[2 * o] = outx; p.next_pts[2 * o + 1] = outy;
            p.status[o] = (uint8_t)status;
            if (p.err) p.err[o] = errv;
        }
    }
}

extern "C" int vfe_dev_klt_track(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                 const float* prev_pts, const int* n_pts, int pts_cap,
                                 float* next_pts, uint8_t* status, float* err,
                                 int win, int max_iter, double eps, double min_eig_thr, int use_initial_flow,
                                 void* stream) {
    VFE_REQUIRE(prev_levels && next_levels && nlevels >= 1 && nlevels <= KLT_MAX_LEVELS, "nlevels must be 1..8");
    VFE_REQUIRE(prev_pts && next_pts && status, "NULL point buffers");
    VFE_REQUIRE(pts_cap > 0, "pts_cap must be > 0");
    VFE_REQUIRE(win >= 3 && win <= 41 && (win & 1), "win must be odd, 3..41");
    VFE_REQUIRE(max_iter >= 1 && max_iter <= 1000, "max_iter out of range");
    KltParams p;
    for (int l = 0; l < nlevels; l++) {
        int rc;
        if ((rc = vfe_check_dimg(&prev_levels[l], "prev level")) || (rc = vfe_check_dimg(&next_levels[l], "next level"))) return rc;
        VFE_REQUIRE(prev_levels[l].width == next_levels[l].width && prev_levels[l].height == next_levels[l].height &&
                        prev_levels[l].batch == next_levels[l].batch && prev_levels[l].batch == prev_levels[0].batch,
                    "prev/next pyramids differ in shape");
        if (l > 0)
            VFE_REQUIRE(prev_levels[l].width == (prev_levels[l - 1].width + 1) / 2 &&
                            prev_levels[l].height == (prev_levels[l - 1].height + 1) / 2,
                        "level sizes must halve (rounded up)");
        p.prev[l] = to_dimg(&prev_levels[l]);
        p.next[l] = to_dimg(&next_levels[l]);
    }
    VFE_REQUIRE(prev_levels[0].batch <= 65535, "batch too large");
    p.nlevels = nlevels;
    p.prev_pts = prev_pts; p.n_pts = n_pts; p.cap = pts_cap;
    p.next_pts = next_pts; p.status = status; p.err = err;
    p.win = win; p.max_iter = max_iter; p.epsilon = eps * eps; p.min_eig_thr = min_eig_thr;
    p.use_initial = use_initial_flow ? 1 : 0;
    const int PW = win + 3, DW = win + 1;
    const size_t smem = ((PW * PW + 15) & ~15) + (size_t)DW * DW * 4 + (size_t)((win * win + 1) & ~1) * 2 + (size_t)win * win * 4;
    ProfScope prof(VFE_K_KLT, (hipStream_t)stream);
    VFE_REQUIRE((long long)pts_cap * prev_levels[0].batch < (1ll << 31), "too many points");
    p.total_work = (unsigned)pts_cap * (unsigned)prev_levels[0].batch;
    if (win == 21) {
        const unsigned blocks = (p.total_work + 3) / 4;
        hipLaunchKernelGGL(k_klt21, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL(k_klt, dim3(p.total_work < 8192u ? p.total_work : 8192u), dim3(64), smem, (hipStream_t)stream, p);
    }
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
