/* Lucas-Kanade tracking with the common 21x21 window, register-resident (cv::calcOpticalFlowPyrLK
 * semantics, same arithmetic as the generic kernel in vfe_klt.hip).
 *
 * One wavefront tracks one point.  Lane l < 63 owns 7 consecutive pixels of window row l/3
 * (21 rows x 3 segments = 63 lanes), keeps its share of the template in registers for the whole
 * level and reads the image rows it needs straight from global memory with unaligned dword loads
 * (the patches are cache resident after the first touch) -- no LDS, no barriers.
 *
 * The kernel is bound by vector-ALU issue, not by memory, so the arithmetic is arranged for the
 * packed 16-bit and dot-product instructions of gfx950:
 *   - two vertically adjacent pixels travel as one dword (lo = upper row, hi = lower row), made from
 *     the loaded dwords with one v_perm each;
 *   - Scharr row sums and differences are packed 16-bit operations on those pairs (all values fit:
 *     |3a+10b+3c| <= 4080);
 *   - a bilinear tap is two v_dot2_i32_i16 with the weight pairs (w00,w10) and (w01,w11); rounding
 *     constant and template value ride in the accumulator operand.
 * Window sums are exact integers: partial sums stay in int32 only as long as their bound allows
 * (16 lanes for the Gram matrix, 8 lanes for the mismatch vector), the rest is added in 64-bit
 * scalar arithmetic. */
#include "vfe_klt.h"
#include <float.h>

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ static inline s16x2 as_s2(unsigned v) { return __builtin_bit_cast(s16x2, v); }
__device__ static inline unsigned as_u(s16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ static inline int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ static inline unsigned ldu32(const uint8_t* p) { unsigned v; __builtin_memcpy(&v, p, 4); return v; }
/* (byte k of a) | (byte k of b) << 16 */
__device__ static inline unsigned pair_bytes(int k, unsigned a, unsigned b) {
    return __builtin_amdgcn_perm(b, a, 0x0c040c00u + (unsigned)k * 0x00010001u);
}
__device__ static inline int dot2(unsigned a, unsigned w, int c) { return __builtin_amdgcn_sdot2(as_s2(a), as_s2(w), c, false); }

__device__ static inline int dpp_quad(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false); /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false); /* quad_perm [2,3,0,1] */
    return v;
}
/* exact wave total of per-lane values whose sum over any 16 lanes fits int32 */
__device__ static inline long long wave_sum_r16(int v) {
    v = dpp_quad(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); /* row_ror 4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); /* row_ror 8 */
    return (long long)__builtin_amdgcn_readlane(v, 0) + (long long)__builtin_amdgcn_readlane(v, 16) +
           (long long)__builtin_amdgcn_readlane(v, 32) + (long long)__builtin_amdgcn_readlane(v, 48);
}
/* exact wave total of per-lane values whose sum over any 8 lanes fits int32: after the rotate by
 * four every lane holds its own quad plus one neighbouring quad of its row, so lanes 0 and 8 of a
 * row cover the row's four quads exactly once */
__device__ static inline long long wave_sum_r8(int v) {
    v = dpp_quad(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); /* row_ror 4 */
    long long s = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 8) s += (long long)__builtin_amdgcn_readlane(v, l);
    return s;
}

template <int N>
__device__ static inline void load_row_reflect(const uint8_t* img, int pitch, int W, int H, int y, int x0, int* out) {
    const uint8_t* row = img + (size_t)reflect101(y, H) * pitch;
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = row[reflect101(x0 + k, W)];
}

__global__ __launch_bounds__(64) void k_klt21(KltParams p) {
    const int lane = threadIdx.x;
    /* one wave per workgroup: points differ in iteration count, and single-wave workgroups let the
     * dispatcher refill a SIMD slot as soon as its point is done */
    const unsigned wave = blockIdx.x, nwaves = gridDim.x;
    const bool active = lane < 63;
    /* lane 63 has no pixels: it shadows lane 0 and is left out of every window sum */
    const int r = active ? lane / 3 : 0, x0 = active ? (lane - 3 * (lane / 3)) * 7 : 0;
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
            const int ipx = rfl((int)floorf(px)), ipy = rfl((int)floorf(py));
            if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
                if (level == 0) { status = 0; errv = 0.f; }
                continue;
            }
            int iw00, iw01, iw10, iw11;
            bilinear_weights(px - (float)ipx, py - (float)ipy, iw00, iw01, iw10, iw11);
            unsigned W0 = (unsigned)rfl((int)((unsigned)iw00 | ((unsigned)iw10 << 16))), W1 = (unsigned)rfl((int)((unsigned)iw01 | ((unsigned)iw11 << 16)));

            /* rows ipy+r-1 .. ipy+r+2, columns ipx+x0-1 .. ipx+x0+8 of the previous image as vertical
             * pairs: P01[j] = (row 0, row 1), P12[j] = (row 1, row 2), P23[j] = (row 2, row 3) */
            unsigned P01[10], P12[10], P23[10];
            const bool interior = ipx >= 1 && ipy >= 1 && ipx + win + 4 < I.w && ipy + win + 2 < I.h;
            if (interior) {
                const uint8_t* base = ip + (size_t)(ipy - 1) * I.pitch + (ipx - 1);
                const unsigned off = (unsigned)r * (unsigned)I.pitch + (unsigned)x0;
                unsigned w[4][3];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int k = 0; k < 3; k++) w[i][k] = ldu32(base + off + (unsigned)i * (unsigned)I.pitch + 4u * k);
#pragma unroll
                for (int j = 0; j < 10; j++) {
                    P01[j] = pair_bytes(j & 3, w[0][j >> 2], w[1][j >> 2]);
                    P12[j] = pair_bytes(j & 3, w[1][j >> 2], w[2][j >> 2]);
                    P23[j] = pair_bytes(j & 3, w[2][j >> 2], w[3][j >> 2]);
                }
            } else {
                int R[4][10];
#pragma unroll
                for (int i = 0; i < 4; i++) load_row_reflect<10>(ip, I.pitch, I.w, I.h, ipy + r - 1 + i, ipx + x0 - 1, R[i]);
#pragma unroll
                for (int j = 0; j < 10; j++) {
                    P01[j] = (unsigned)R[0][j] | ((unsigned)R[1][j] << 16);
                    P12[j] = (unsigned)R[1][j] | ((unsigned)R[2][j] << 16);
                    P23[j] = (unsigned)R[2][j] | ((unsigned)R[3][j] << 16);
                }
            }
            /* Scharr at window rows (r, r+1) packed, columns x0 .. x0+7 */
            unsigned vx[8], vy[8];
            {
                s16x2 S[10], D[10];
#pragma unroll
                for (int j = 0; j < 10; j++) {
                    S[j] = (as_s2(P01[j]) + as_s2(P23[j])) * (short)3 + as_s2(P12[j]) * (short)10;
                    D[j] = as_s2(P23[j]) - as_s2(P01[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    vx[j] = as_u(S[j + 2] - S[j]);
                    vy[j] = as_u((D[j] + D[j + 2]) * (short)3 + D[j + 1] * (short)10);
                }
                if (!interior) { /* the derivative image is zero outside the image */
                    const int cy = ipy + r;
                    const unsigned mrow = (cy >= 0 && cy < I.h ? 0xFFFFu : 0u) | (cy + 1 >= 0 && cy + 1 < I.h ? 0xFFFF0000u : 0u);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int cx = ipx + x0 + j;
                        const unsigned m = (cx >= 0 && cx < I.w) ? mrow : 0u;
                        vx[j] &= m; vy[j] &= m;
                    }
                }
            }
            /* template: tC = rounding constant minus the template intensity in accumulator scale,
             * so that dot2(dot2(tC)) >> 9 is directly the intensity difference */
            int tC[7], tIx[7], tIy[7];
            int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int ival = dot2(P12[k + 1], W0, dot2(P12[k + 2], W1, 1 << 8)) >> 9;
                const int ixval = dot2(vx[k], W0, dot2(vx[k + 1], W1, 1 << 13)) >> 14;
                const int iyval = dot2(vy[k], W0, dot2(vy[k + 1], W1, 1 << 13)) >> 14;
                tC[k] = (1 << 8) - (ival << 9); tIx[k] = ixval; tIy[k] = iyval;
                a11 += ixval * ixval; a12 += ixval * iyval; a22 += iyval * iyval;
            }
            if (!active) { a11 = 0; a12 = 0; a22 = 0; }
            const long long iA11 = wave_sum_r16(a11), iA12 = wave_sum_r16(a12), iA22 = wave_sum_r16(a22);
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
            const unsigned joff = (unsigned)r * (unsigned)J.pitch + (unsigned)x0;
            for (int j = 0; j <= p.max_iter; j++) {
                /* j == max_iter is the pass that only evaluates the residual at the final position */
                float qx = nx, qy = ny;
                const bool err_pass = (j == p.max_iter);
                if (err_pass) {
                    if (!(status && level == 0)) break;
                    qx = outx - half; qy = outy - half;
                }
                const int inx = rfl((int)floorf(qx)), iny = rfl((int)floorf(qy));
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                    if (level == 0) status = 0;
                    break;
                }
                bilinear_weights(qx - (float)inx, qy - (float)iny, iw00, iw01, iw10, iw11);
                W0 = (unsigned)rfl((int)((unsigned)iw00 | ((unsigned)iw10 << 16))); W1 = (unsigned)rfl((int)((unsigned)iw01 | ((unsigned)iw11 << 16)));
                /* rows iny+r, iny+r+1, columns inx+x0 .. inx+x0+7 as vertical pairs */
                unsigned Q[8];
                const bool jin = inx >= 0 && iny >= 0 && inx + win + 1 < J.w && iny + win + 1 < J.h;
                if (jin) {
                    const uint8_t* base = jp + (size_t)iny * J.pitch + inx;
                    const unsigned w0a = ldu32(base + joff), w0b = ldu32(base + joff + 4u);
                    const unsigned w1a = ldu32(base + joff + (unsigned)J.pitch), w1b = ldu32(base + joff + (unsigned)J.pitch + 4u);
#pragma unroll
                    for (int k = 0; k < 8; k++) Q[k] = pair_bytes(k & 3, k < 4 ? w0a : w0b, k < 4 ? w1a : w1b);
                } else {
                    int q0[8], q1[8];
                    load_row_reflect<8>(jp, J.pitch, J.w, J.h, iny + r, inx + x0, q0);
                    load_row_reflect<8>(jp, J.pitch, J.w, J.h, iny + r + 1, inx + x0, q1);
#pragma unroll
                    for (int k = 0; k < 8; k++) Q[k] = (unsigned)q0[k] | ((unsigned)q1[k] << 16);
                }
                int diff[7];
#pragma unroll
                for (int k = 0; k < 7; k++) diff[k] = dot2(Q[k], W0, dot2(Q[k + 1], W1, tC[k])) >> 9;
                if (err_pass) {
                    int es = 0;
#pragma unroll
                    for (int k = 0; k < 7; k++) es += abs(diff[k]);
                    if (!active) es = 0;
                    const long long esum = wave_sum_r16(es);
                    errv = (float)esum * 1.f / (float)(32 * win * win);
                    break;
                }
                int b1 = 0, b2 = 0;
#pragma unroll
                for (int k = 0; k < 7; k++) { b1 += diff[k] * tIx[k]; b2 += diff[k] * tIy[k]; }
                if (!active) { b1 = 0; b2 = 0; }
                const long long ib1 = wave_sum_r8(b1), ib2 = wave_sum_r8(b2);
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

void klt21_launch(const KltParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(k_klt21, dim3(p.total_work < (1u << 20) ? p.total_work : (1u << 20)), dim3(64), 0, stream, p);
}
