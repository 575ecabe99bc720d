/* Lucas-Kanade tracking with the common 21x21 window (cv::calcOpticalFlowPyrLK semantics, same
 * arithmetic as the generic kernel in vfe_klt.hip).
 *
 * One wavefront tracks FOUR points, one per row of 16 lanes.  The 441 window pixels of a point
 * are split over its 16 lanes as two horizontal spans per lane: lane l owns the whole window
 * row l (21 pixels) and, for l < 15, a piece of one of the rows 16..20 (columns 0-7, 8-15 or
 * 16-20).  A lane keeps its share of the template in registers for the whole level.  All
 * per-point scalar work (positions, weights, the 2x2 solve, convergence tests) is ordinary
 * per-lane arithmetic, replicated over the 16 lanes of the point, so four points share every
 * instruction issue; control flow is uniform within a row of 16 lanes, which is what the
 * row-wise DPP reductions need.
 *
 * Memory: the 16 lanes of a point fetch its 24x24-byte patch with consecutive unaligned dword
 * loads (a few cache lines per instruction instead of one per lane) and hand it over through a
 * private 768-byte LDS area, from which every lane reads the rows of its spans.  The areas of
 * the four points are independent, so the four rows of the wave may diverge freely; within a
 * wave LDS operations execute in order, only compiler reordering has to be fenced.
 *
 * Arithmetic is arranged for the packed 16-bit and dot-product instructions of gfx950:
 *   - two vertically adjacent pixels travel as one dword (lo = upper row, hi = lower row), made from
 *     the loaded dwords with one v_perm each;
 *   - Scharr row sums and differences are packed 16-bit operations on those pairs (all values fit:
 *     |3a+10b+3c| <= 4080);
 *   - a bilinear tap is two v_dot2_i32_i16 with the weight pairs (w00,w10) and (w01,w11); rounding
 *     constant and template value ride in the accumulator operand;
 *   - the mismatch vector is accumulated with v_dot2 over pairs of pixels.
 * Window sums are exact integers: per-lane partial sums fit int32, the 16-lane totals are formed
 * separately for the low and high halves and recombined in double before the single rounding to
 * float that the CPU code performs on its 64-bit sum. */
#include "vfe_klt.h"
#include <float.h>

typedef short s16x2 __attribute__((ext_vector_type(2)));

#define KLT_NEED_BORDER 24 /* reflect-101 margin a level needs for direct reads of any accepted window */
#define PATCH_ROW_DW 8     /* LDS row stride in dwords (24 bytes used) */
#define PATCH_DW (24 * PATCH_ROW_DW + 16) /* per point; the tail staggers the banks of the four areas */

__device__ static inline s16x2 as_s2(unsigned v) { return __builtin_bit_cast(s16x2, v); }
__device__ static inline unsigned as_u(s16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ static inline unsigned ldu32(const uint8_t* p) { unsigned v; __builtin_memcpy(&v, p, 4); return v; }
/* (byte k of a) | (byte k of b) << 16 */
__device__ static inline unsigned pair_bytes(int k, unsigned a, unsigned b) {
    return __builtin_amdgcn_perm(b, a, 0x0c040c00u + (unsigned)k * 0x00010001u);
}
/* (lo & 0xffff) | (hi << 16) */
__device__ static inline unsigned pack16(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u); }
__device__ static inline int dot2(unsigned a, unsigned w, int c) { return __builtin_amdgcn_sdot2(as_s2(a), as_s2(w), c, false); }

/* sum over the row of 16 lanes, result in every lane of the row; the row total must fit int32 */
__device__ static inline int row_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);  /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);  /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); /* row_ror 4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); /* row_ror 8 */
    return v;
}
/* float(sum over the row of 16 lanes), rounded once from the exact integer total */
__device__ static inline float row_sum_to_float(int v) {
    const int lo = row_sum(v & 0xFFFF), hi = row_sum(v >> 16);
    return (float)__builtin_fma((double)hi, 65536.0, (double)lo);
}

/* The 16 lanes of a point copy NROWS rows x 24 bytes, starting at image position (x0, y0), into the
 * point's LDS area.  Levels that carry a reflect-101 border are read directly; otherwise rows are
 * reflected by index and columns byte by byte where a dword overhangs the image. */
template <int NROWS>
__device__ static inline void stage_patch(const DImg& im, const uint8_t* fp, int x0, int y0, int sub, unsigned* area) {
    const bool padded = im.border >= KLT_NEED_BORDER;
    const bool cols_in = padded || (x0 >= 0 && x0 + 24 <= im.w);
    unsigned v[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int d = sub + 16 * i, row = d / 6, k = d - 6 * row;
        v[i] = 0;
        if (NROWS * 6 >= 16 * (i + 1) || row < NROWS) {
            const int y = padded ? y0 + row : reflect101(y0 + row, im.h);
            const uint8_t* src = fp + (ptrdiff_t)y * im.pitch;
            const int x = x0 + 4 * k;
            if (cols_in) v[i] = ldu32(src + x);
            else v[i] = (unsigned)src[reflect101(x, im.w)] | ((unsigned)src[reflect101(x + 1, im.w)] << 8) |
                        ((unsigned)src[reflect101(x + 2, im.w)] << 16) | ((unsigned)src[reflect101(x + 3, im.w)] << 24);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* earlier reads of the area stay ahead of these writes */
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int d = sub + 16 * i, row = d / 6, k = d - 6 * row;
        if (NROWS * 6 >= 16 * (i + 1) || row < NROWS) area[row * PATCH_ROW_DW + k] = v[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
/* NROWS consecutive patch rows starting at `row`, NDW dwords each starting at dword `dw` (NDW = 6: the
 * whole row; NDW = 3: an 8-byte aligned piece) */
template <int NROWS, int NDW>
__device__ static inline void read_rows(const unsigned* area, int row, int dw, unsigned (&w)[NROWS][NDW]) {
#pragma unroll
    for (int i = 0; i < NROWS; i++) {
        const unsigned* r = area + (row + i) * PATCH_ROW_DW + dw;
        if (NDW == 6) {
            const uint4 a = *reinterpret_cast<const uint4*>(r);
            const uint2 c = *reinterpret_cast<const uint2*>(r + 4);
            w[i][0] = a.x; w[i][1] = a.y; w[i][2] = a.z; w[i][3] = a.w; w[i][4] = c.x; w[i][5] = c.y;
        } else {
            const uint2 a = *reinterpret_cast<const uint2*>(r);
            w[i][0] = a.x; w[i][1] = a.y; w[i][2] = r[2];
        }
    }
}

/* Template of a span of NPX pixels in one window row.  w[i] holds image rows y-1 .. y+2, bytes
 * x-1 .. x+NPX+1 (x, y = image position of the span's first pixel, integer part); only the first
 * `nvalid` pixels belong to the window (checked for pixels >= 5 only).  Outputs: tC = rounding
 * constant minus template intensity in accumulator scale, Tx/Ty = gradients packed in pairs of
 * pixels (zero for pixels outside the window); the Gram sums are accumulated. */
template <int NPX>
__device__ static inline void span_template(const unsigned (&w)[4][(NPX + 6) / 4], unsigned W0, unsigned W1,
                                            bool interior, int x, int y, int IW, int IH, int nvalid,
                                            int* tC, unsigned* Tx, unsigned* Ty, int& a11, int& a12, int& a22) {
    constexpr int NC = NPX + 3;
    unsigned P12[NC];
    s16x2 S[NC], D[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) {
        const unsigned p01 = pair_bytes(j & 3, w[0][j >> 2], w[1][j >> 2]);
        P12[j] = pair_bytes(j & 3, w[1][j >> 2], w[2][j >> 2]);
        const unsigned p23 = pair_bytes(j & 3, w[2][j >> 2], w[3][j >> 2]);
        S[j] = (as_s2(p01) + as_s2(p23)) * (short)3 + as_s2(P12[j]) * (short)10;
        D[j] = as_s2(p23) - as_s2(p01);
    }
    /* Scharr at window rows (y, y+1) packed, columns x .. x+NPX */
    unsigned vx[NPX + 1], vy[NPX + 1];
#pragma unroll
    for (int j = 0; j < NPX + 1; j++) {
        vx[j] = as_u(S[j + 2] - S[j]);
        vy[j] = as_u((D[j] + D[j + 2]) * (short)3 + D[j + 1] * (short)10);
    }
    if (!interior) { /* the derivative image is zero outside the image */
        const unsigned mrow = (y >= 0 && y < IH ? 0xFFFFu : 0u) | (y + 1 >= 0 && y + 1 < IH ? 0xFFFF0000u : 0u);
#pragma unroll
        for (int j = 0; j < NPX + 1; j++) {
            const unsigned m = (x + j >= 0 && x + j < IW) ? mrow : 0u;
            vx[j] &= m; vy[j] &= m;
        }
    }
    int ix[NPX + 1], iy[NPX + 1];
#pragma unroll
    for (int k = 0; k < NPX; k++) {
        const int ival = dot2(P12[k + 1], W0, dot2(P12[k + 2], W1, 1 << 8)) >> 9;
        ix[k] = dot2(vx[k], W0, dot2(vx[k + 1], W1, 1 << 13)) >> 14;
        iy[k] = dot2(vy[k], W0, dot2(vy[k + 1], W1, 1 << 13)) >> 14;
        if (NPX == 8 && k >= 5 && k >= nvalid) { ix[k] = 0; iy[k] = 0; }
        tC[k] = (1 << 8) - (ival << 9);
        a11 += ix[k] * ix[k]; a12 += ix[k] * iy[k]; a22 += iy[k] * iy[k];
    }
    ix[NPX] = 0; iy[NPX] = 0;
#pragma unroll
    for (int k = 0; k < (NPX + 1) / 2; k++) { Tx[k] = pack16(ix[2 * k], ix[2 * k + 1]); Ty[k] = pack16(iy[2 * k], iy[2 * k + 1]); }
}

/* Intensity differences of a span against its template.  w[i] holds image rows y, y+1, bytes
 * x .. x+NPX. */
template <int NPX>
__device__ static inline void span_diff(const unsigned (&w)[2][(NPX + 6) / 4], unsigned W0, unsigned W1, const int* tC, int* diff) {
    unsigned Q[NPX + 1];
#pragma unroll
    for (int k = 0; k < NPX + 1; k++) Q[k] = pair_bytes(k & 3, w[0][k >> 2], w[1][k >> 2]);
#pragma unroll
    for (int k = 0; k < NPX; k++) diff[k] = dot2(Q[k], W0, dot2(Q[k + 1], W1, tC[k])) >> 9;
}
template <int NPX>
__device__ static inline void span_mismatch(const int* diff, const unsigned* Tx, const unsigned* Ty, int& b1, int& b2) {
#pragma unroll
    for (int k = 0; k < (NPX + 1) / 2; k++) {
        const unsigned d = pack16(diff[2 * k], 2 * k + 1 < NPX ? diff[2 * k + 1] : 0);
        b1 = dot2(d, Tx[k], b1); b2 = dot2(d, Ty[k], b2);
    }
}

__global__ __launch_bounds__(64) void k_klt21(KltParams p) {
    __shared__ __align__(16) unsigned s_patch[4][PATCH_DW];
    const int lane = threadIdx.x, sub = lane & 15, q = lane >> 4;
    unsigned* area = s_patch[q];
    /* span A: window row sub, columns 0..20.  span B: a piece of one of the rows 16..20; lane 15
     * has none, shadows lane 14 and is left out of the sums */
    const bool hasB = sub < 15;
    const int sb = hasB ? sub : 14;
    const int rB = 16 + sb / 3, pB = sb - 3 * (sb / 3), cB = 8 * pB, nB = pB == 2 ? 5 : 8;
    const int win = 21;
    const float half = 10.f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int max_level = p.nlevels - 1;
    const unsigned n_items = (p.total_work + 3u) / 4u;

    for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x) {
        const unsigned idx = item * 4u + (unsigned)q;
        if (idx >= p.total_work) continue;
        const int b = (int)(idx / (unsigned)p.cap), pt = (int)(idx - (unsigned)b * (unsigned)p.cap);
        const size_t o = idx;
        const int n = p.n_pts ? min(p.n_pts[b], p.cap) : p.cap;
        if (pt >= n) {
            if (sub == 0) { p.status[o] = 0; if (p.err) p.err[o] = 0.f; }
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
            unsigned W0 = (unsigned)iw00 | ((unsigned)iw10 << 16), W1 = (unsigned)iw01 | ((unsigned)iw11 << 16);

            int tCA[21], tCB[8];
            unsigned TxA[11], TyA[11], TxB[4], TyB[4];
            int a11 = 0, a12 = 0, a22 = 0;
            {
                /* patch rows ipy-1 .. ipy+22, columns ipx-1 .. ipx+22; gradients are needed at columns
                 * ipx .. ipx+21, rows ipy .. ipy+21 */
                stage_patch<24>(I, ip, ipx - 1, ipy - 1, sub, area);
                const bool interior = ipx >= 0 && ipy >= 0 && ipx + win < I.w && ipy + win < I.h;
                int b11 = 0, b12 = 0, b22 = 0;
                {
                    unsigned w[4][6];
                    read_rows<4, 6>(area, sub, 0, w);
                    span_template<21>(w, W0, W1, interior, ipx, ipy + sub, I.w, I.h, 21, tCA, TxA, TyA, a11, a12, a22);
                }
                {
                    unsigned w[4][3];
                    read_rows<4, 3>(area, rB, cB / 4, w);
                    span_template<8>(w, W0, W1, interior, ipx + cB, ipy + rB, I.w, I.h, nB, tCB, TxB, TyB, b11, b12, b22);
                }
                if (hasB) { a11 += b11; a12 += b12; a22 += b22; }
            }
            const float A11 = row_sum_to_float(a11) * FLT_SCALE, A12 = row_sum_to_float(a12) * FLT_SCALE,
                        A22 = row_sum_to_float(a22) * FLT_SCALE;
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
                W0 = (unsigned)iw00 | ((unsigned)iw10 << 16); W1 = (unsigned)iw01 | ((unsigned)iw11 << 16);
                /* patch rows iny .. iny+21, columns inx .. inx+23 (22 used) */
                stage_patch<22>(J, jp, inx, iny, sub, area);
                int dA[21], dB[8];
                {
                    unsigned w[2][6];
                    read_rows<2, 6>(area, sub, 0, w);
                    span_diff<21>(w, W0, W1, tCA, dA);
                }
                {
                    unsigned w[2][3];
                    read_rows<2, 3>(area, rB, cB / 4, w);
                    span_diff<8>(w, W0, W1, tCB, dB);
                }
                if (err_pass) {
                    int es = 0, eb = 0;
#pragma unroll
                    for (int k = 0; k < 21; k++) es += abs(dA[k]);
#pragma unroll
                    for (int k = 0; k < 8; k++) eb += (k >= 5 && k >= nB) ? 0 : abs(dB[k]);
                    if (hasB) es += eb;
                    errv = (float)row_sum(es) * 1.f / (float)(32 * win * win);
                    break;
                }
                int b1 = 0, b2 = 0, c1 = 0, c2 = 0;
                span_mismatch<21>(dA, TxA, TyA, b1, b2);
                span_mismatch<8>(dB, TxB, TyB, c1, c2);
                if (hasB) { b1 += c1; b2 += c2; }
                const float fb1 = row_sum_to_float(b1) * FLT_SCALE, fb2 = row_sum_to_float(b2) * FLT_SCALE;
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
        if (sub == 0) {
            p.next_pts[2 * o] = outx; p.next_pts[2 * o + 1] = outy;
            p.status[o] = (uint8_t)status;
            if (p.err) p.err[o] = errv;
        }
    }
}

void klt21_launch(const KltParams& p, hipStream_t stream) {
    const unsigned items = (p.total_work + 3u) / 4u;
    hipLaunchKernelGGL(k_klt21, dim3(items < (1u << 20) ? items : (1u << 20)), dim3(64), 0, stream, p);
}
