/* Lucas-Kanade tracking with the common 21x21 window (cv::calcOpticalFlowPyrLK semantics, same
 * arithmetic as the generic kernel in vfe_klt.hip).
 *
 * One wavefront tracks FOUR points, one per row of 16 lanes.  The 441 window pixels of a point
 * are split over its 16 lanes in pieces of at most 8 pixels of one window row: lane l owns the
 * whole window row l (pieces at columns 0, 8 and 16) and, for l < 15, one piece of the rows
 * 16..20.  A lane keeps the gradients of its pixels in registers for the whole level.  All
 * per-point scalar work (positions, weights, the 2x2 solve, convergence tests) is ordinary
 * per-lane arithmetic, replicated over the 16 lanes of the point, so four points share every
 * instruction issue; control flow is uniform within a row of 16 lanes, which is what the
 * row-wise DPP reductions need.
 *
 * Memory: the lanes of a point fetch a 24-byte wide patch with consecutive unaligned 8-byte
 * loads (a few cache lines per instruction instead of one per lane) and hand it over through a
 * private 768-byte LDS area, from which every lane reads the rows of its pieces.  The areas of
 * the four points are independent, so the four rows of the wave may diverge freely; within a
 * wave LDS operations execute in order, only compiler reordering has to be fenced.  The patch of
 * the next image for the coming pass is requested as soon as its position is known, i.e. before
 * the template is computed and before the bookkeeping that ends a pass.
 *
 * Arithmetic is arranged for the packed 16-bit and dot-product instructions of gfx950:
 *   - two vertically adjacent pixels travel as one dword (lo = upper row, hi = lower row), made from
 *     the loaded dwords with one v_perm each;
 *   - Scharr row sums and differences are packed 16-bit operations on those pairs (all values fit:
 *     |3a+10b+3c| <= 4080);
 *   - a bilinear tap is two v_dot2_i32_i16 with the weight pairs (w00,w10) and (w01,w11);
 *   - the mismatch vector sum((J - I) * g) is accumulated as sum(J * g) with v_dot2 over pairs of
 *     pixels, minus sum(I * g), which is a constant of the level; the template intensities I
 *     themselves are only needed for the final residual and are recomputed there.
 * Window sums are exact integers: per-lane partial sums fit int32, the 16-lane totals are formed
 * separately for the low and high halves and recombined in double before the single rounding to
 * float that the CPU code performs on its 64-bit sum. */
#include "vfe_klt.h"
#include <float.h>
#include <stdlib.h>

typedef short s16x2 __attribute__((ext_vector_type(2)));

#define KLT_NEED_BORDER 24 /* reflect-101 margin a level needs for direct reads of any accepted window */
#define PATCH_ROW_DW 8     /* LDS row stride in dwords (24 bytes used) */
#define PATCH_DW (24 * PATCH_ROW_DW + 16) /* per point; the tail staggers the banks of the four areas */

__device__ static inline s16x2 as_s2(unsigned v) { return __builtin_bit_cast(s16x2, v); }
__device__ static inline unsigned as_u(s16x2 v) { return __builtin_bit_cast(unsigned, v); }
/* (byte k of a) | (byte k of b) << 16 */
__device__ static inline unsigned pair_bytes(int k, unsigned a, unsigned b) {
    return __builtin_amdgcn_perm(b, a, 0x0c040c00u + (unsigned)k * 0x00010001u);
}
/* (lo >> 16) | (hi & 0xffff0000) */
__device__ static inline unsigned pack_hi16(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x07060302u); }
/* (lo >> 9) & 0xffff | (hi >> 9) << 16 for non-negative values below 2^25: the second shift writes its 16-bit result
 * into the upper half of the destination and keeps the lower half (sub-dword addressing), which saves the permute that
 * would join two shifted values.  A result written to half a register must not be read by the very next vector
 * instruction; the compiler does not see into the statement, so the wait state is part of it. */
__device__ static inline unsigned pack_shr9(int lo, int hi) {
    unsigned d = (unsigned)lo >> 9;
    asm("v_lshrrev_b32_sdwa %0, 9, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\ts_nop 0"
        : "+v"(d) : "v"(hi));
    return d;
}
/* a.lo*w.lo + a.hi*w.hi + c on signed 16-bit halves.  Written as the three-address form: the
 * compiler's choice, the accumulating two-address form, costs a register copy per use whenever the
 * addend has to survive. */
__device__ static inline int dot2(unsigned a, unsigned w, int c) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(w), "v"(c));
    return d;
}

/* sum over the row of 16 lanes, result in every lane of the row; the row total must fit int32 */
__device__ static inline int row_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);  /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);  /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); /* row_ror 4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); /* row_ror 8 */
    return v;
}
/* float(sum over the row of 16 lanes), rounded once from the exact integer total: both halves are exact as floats
 * (|hi| < 2^20, lo < 2^20) and the fused multiply-add rounds hi * 65536 + lo once */
__device__ static inline float row_sum_to_float(int v) {
    const int lo = row_sum(v & 0xFFFF), hi = row_sum(v >> 16);
    return __builtin_fmaf((float)hi, 65536.f, (float)lo);
}

/* Twelve of the 16 lanes of a point request NROWS rows x 24 bytes starting at image position
 * (x0, y0): lane (pr, pk) = (sub / 3, sub % 3) takes bytes 8*pk .. 8*pk+7 of the rows pr, pr+4, ...
 * Levels that carry a reflect-101 border are read directly; otherwise rows are reflected by index
 * and columns byte by byte where the patch overhangs the image. */
struct PatchLane { int pr, pk; bool on; };
__device__ static inline uint2 ldu64(const uint8_t* p) { uint2 v; __builtin_memcpy(&v, p, 8); return v; }
template <int NROWS>
__device__ static inline void fetch_patch(const DImg& im, const uint8_t* fp, int x0, int y0, const PatchLane& P, uint2 (&v)[6]) {
    const bool padded = im.border >= KLT_NEED_BORDER;
    const bool cols_in = padded || (x0 >= 0 && x0 + 24 <= im.w);
    const int x = x0 + 8 * P.pk;
    /* Patches that lie inside the image (or its border) for every point of the wave -- nearly all -- take one address
     * computation and a pitch step per row; the general form reflects row by row (wave-uniform choice). */
    const bool inside = cols_in && (padded || (y0 >= 0 && y0 + NROWS <= im.h));
    if (__builtin_amdgcn_ballot_w64(P.on && !inside) == 0) {
        const uint8_t* src = fp + (ptrdiff_t)(y0 + P.pr) * im.pitch + x;
        const ptrdiff_t step = (ptrdiff_t)4 * im.pitch;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            v[i] = make_uint2(0u, 0u);
            if (P.on && (4 * i + 4 <= NROWS || 4 * i + P.pr < NROWS)) v[i] = ldu64(src);
            src += step;
        }
        return;
    }
    /* this path is rare: its row numbers are formed here, behind a statement the optimiser cannot look through -- hoisted
     * out of the level loop they would sit in registers (or in scratch) for the whole kernel */
    int pr = P.pr;
    asm volatile("" : "+v"(pr));
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int row = 4 * i + pr;
        v[i] = make_uint2(0u, 0u);
        if (P.on && (4 * i + 4 <= NROWS || row < NROWS)) {
            const int y = padded ? y0 + row : reflect101(y0 + row, im.h);
            const uint8_t* src = fp + (ptrdiff_t)y * im.pitch;
            if (cols_in) v[i] = ldu64(src + x);
            else {
                unsigned lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    lo |= (unsigned)src[reflect101(x + k, im.w)] << (8 * k);
                    hi |= (unsigned)src[reflect101(x + 4 + k, im.w)] << (8 * k);
                }
                v[i] = make_uint2(lo, hi);
            }
        }
    }
}
/* ... and publish them in the point's LDS area */
template <int NROWS>
__device__ static inline void store_patch(const uint2 (&v)[6], const PatchLane& P, unsigned* area) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* earlier reads of the area stay ahead of these writes */
    uint2* dst = reinterpret_cast<uint2*>(area + P.pr * PATCH_ROW_DW + 2 * P.pk);
#pragma unroll
    for (int i = 0; i < 6; i++)
        if (P.on && (4 * i + 4 <= NROWS || 4 * i + P.pr < NROWS)) dst[i * 4 * PATCH_ROW_DW / 2] = v[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
/* NROWS consecutive patch rows starting at `row`, three dwords each starting at dword `dw` (even) */
template <int NROWS>
__device__ static inline void read_piece(const unsigned* area, int row, int dw, unsigned (&w)[NROWS][3]) {
#pragma unroll
    for (int i = 0; i < NROWS; i++) {
        const unsigned* r = area + (row + i) * PATCH_ROW_DW + dw;
        const uint2 a = *reinterpret_cast<const uint2*>(r);
        w[i][0] = a.x; w[i][1] = a.y; w[i][2] = r[2];
    }
}

/* Gradient template of a piece of NPX <= 8 pixels of one window row.  w[i] holds image rows
 * y-1 .. y+2, bytes x-1 .. x+10 (x, y = image position of the piece's first pixel, integer
 * part); only the first `nvalid` pixels belong to the window (checked for pixels >= 5 only).
 * Tx/Ty receive the bilinear-interpolated Scharr gradients packed in pairs of pixels (zero
 * outside the window); the Gram sums and the sums I*gx, I*gy are accumulated.  MASKED is the
 * variant for windows that overhang the image. */
/* what a piece hands to the next piece of the same window row, which starts 8 columns on: the three columns they share
 * (row sums and differences, the middle row pairs of the last two) and the Scharr values of the first shared column */
struct PieceCarry { s16x2 S[3], D[3]; unsigned P[2], vx, vy; };
template <int NPX, bool MASKED, bool CARRY_IN, bool CARRY_OUT>
__device__ static inline void piece_template(const unsigned (&w)[4][3], unsigned W0, unsigned W1,
                                             bool interior, int x, int y, int IW, int IH, int nvalid,
                                             unsigned* Tx, unsigned* Ty, int& a11, int& a12, int& a22, int& c1, int& c2,
                                             PieceCarry& carry) {
    constexpr int NC = NPX + 3;
    unsigned P12[NC];
    s16x2 S[NC], D[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) {
        if (CARRY_IN && j < 3) {
            S[j] = carry.S[j]; D[j] = carry.D[j];
            P12[j] = j ? carry.P[j - 1] : 0u;
            continue;
        }
        const unsigned p01 = pair_bytes(j & 3, w[0][j >> 2], w[1][j >> 2]);
        P12[j] = pair_bytes(j & 3, w[1][j >> 2], w[2][j >> 2]);
        const unsigned p23 = pair_bytes(j & 3, w[2][j >> 2], w[3][j >> 2]);
        /* the Scharr row sums and, below, the column sums carry a factor 4 (12 / 40 for 3 / 10; everything still fits
         * 16 bits: 4 * 4080), so that the interpolated gradient's rounding shift becomes 16 and the result is the
         * product's high half, which the pair-packing permute picks up directly */
        S[j] = (as_s2(p01) + as_s2(p23)) * (short)12 + as_s2(P12[j]) * (short)40;
        D[j] = as_s2(p23) - as_s2(p01);
    }
    /* Scharr at window rows (y, y+1) packed, columns x .. x+NPX; the derivative image is zero
     * outside the image (rows masked by halves, columns as a whole) */
    const unsigned mrow = interior ? 0xFFFFFFFFu
                                   : ((y >= 0 && y < IH ? 0xFFFFu : 0u) | (y + 1 >= 0 && y + 1 < IH ? 0xFFFF0000u : 0u));
    unsigned vx[NPX + 1], vy[NPX + 1];
#pragma unroll
    for (int j = 0; j < NPX + 1; j++) {
        if (CARRY_IN && j == 0) { vx[0] = carry.vx; vy[0] = carry.vy; continue; }
        const unsigned m = !MASKED || (unsigned)(x + j) < (unsigned)IW ? mrow : 0u;
        vx[j] = as_u(S[j + 2] - S[j]);
        vy[j] = as_u((D[j] + D[j + 2]) * (short)12 + D[j + 1] * (short)40);
        if (MASKED) { vx[j] &= m; vy[j] &= m; }
    }
    if (CARRY_OUT) {
        static_assert(!CARRY_OUT || NPX == 8, "the next piece starts 8 columns on");
#pragma unroll
        for (int j = 0; j < 3; j++) { carry.S[j] = S[NPX + j]; carry.D[j] = D[NPX + j]; }
        carry.P[0] = P12[NPX + 1]; carry.P[1] = P12[NPX + 2];
        carry.vx = vx[NPX]; carry.vy = vy[NPX];
    }
    int ix[NPX + 1], iy[NPX + 1], iv[NPX + 1];
#pragma unroll
    for (int k = 0; k < NPX; k++) {
        iv[k] = dot2(P12[k + 1], W0, dot2(P12[k + 2], W1, 1 << 8)); /* intensity x 32 in bits 9.. */
        ix[k] = dot2(vx[k], W0, dot2(vx[k + 1], W1, 1 << 15)); /* gradient in bits 16..31 */
        iy[k] = dot2(vy[k], W0, dot2(vy[k + 1], W1, 1 << 15));
        if (k >= 5 && k >= nvalid) { ix[k] = 0; iy[k] = 0; }
    }
    ix[NPX] = 0; iy[NPX] = 0; iv[NPX] = 0;
    /* every value fits 16 bits (intensities <= 255 * 32, gradients <= 16 * 255): the Gram sums and the sums I*gx, I*gy
     * take two pixels per instruction as dot products of the packed pairs */
#pragma unroll
    for (int k = 0; k < (NPX + 1) / 2; k++) {
        const unsigned tx = pack_hi16(ix[2 * k], ix[2 * k + 1]), ty = pack_hi16(iy[2 * k], iy[2 * k + 1]);
        const unsigned tv = pack_shr9(iv[2 * k], iv[2 * k + 1]);
        Tx[k] = tx; Ty[k] = ty;
        a11 = dot2(tx, tx, a11); a12 = dot2(tx, ty, a12); a22 = dot2(ty, ty, a22);
        c1 = dot2(tv, tx, c1); c2 = dot2(tv, ty, c2);
    }
}

/* Bilinear-interpolated intensities of a piece before the rounding shift (intensity times 32 = val >> 9; the rounding
 * constant is already added).  w[i] holds image rows y, y+1, bytes x .. x+11. */
template <int NPX>
__device__ static inline void piece_values(const unsigned (&w)[2][3], unsigned W0, unsigned W1, int* val) {
    unsigned Q[NPX + 1];
#pragma unroll
    for (int k = 0; k < NPX + 1; k++) Q[k] = pair_bytes(k & 3, w[0][k >> 2], w[1][k >> 2]);
#pragma unroll
    for (int k = 0; k < NPX; k++) val[k] = dot2(Q[k], W0, dot2(Q[k + 1], W1, 1 << 8));
}
template <int NPX>
__device__ static inline void piece_mismatch(const int* val, const unsigned* Tx, const unsigned* Ty, int& b1, int& b2) {
#pragma unroll
    for (int k = 0; k < (NPX + 1) / 2; k++) {
        const unsigned d = 2 * k + 1 < NPX ? pack_shr9(val[2 * k], val[2 * k + 1]) : (unsigned)val[2 * k] >> 9;
        b1 = dot2(d, Tx[k], b1); b2 = dot2(d, Ty[k], b2);
    }
}

/* the four pieces of a lane: window row `sub` at columns 0, 8, 16 (8, 8 and 5 pixels) and the
 * piece (rB, cB) with nB pixels */
struct LanePieces { int sub, rB, cB, nB; bool hasB; };

/* One pass over the pieces of the lane, patch of the next image staged in `area`: accumulates
 * sum(J*gx), sum(J*gy) (rows 16..20 piece separately, since lane 15 has none).  The scheduling
 * barriers keep the pieces from being interleaved, which would multiply the registers in flight. */
template <int NPX>
__device__ static inline void piece_pass(const unsigned* area, int row, int dw, unsigned W0, unsigned W1,
                                         const unsigned* Tx, const unsigned* Ty, int& b1, int& b2) {
    unsigned w[2][3];
    int v[NPX];
    read_piece<2>(area, row, dw, w);
    piece_values<NPX>(w, W0, W1, v);
    piece_mismatch<NPX>(v, Tx, Ty, b1, b2);
    __builtin_amdgcn_sched_barrier(0);
}
/* the same with the piece's rows already in registers (read while the piece before it was being computed) */
template <int NPX>
__device__ static inline void piece_pass_rows(const unsigned (&w)[2][3], unsigned W0, unsigned W1,
                                              const unsigned* Tx, const unsigned* Ty, int& b1, int& b2) {
    int v[NPX];
    piece_values<NPX>(w, W0, W1, v);
    piece_mismatch<NPX>(v, Tx, Ty, b1, b2);
    __builtin_amdgcn_sched_barrier(0);
}
/* sum |J - I| over the first nvalid pixels of a piece; both patches staged (22-row layout) */
template <int NPX>
__device__ static inline int piece_residual(const unsigned* areaJ, const unsigned* areaI, int row, int dw, int nvalid,
                                            unsigned W0, unsigned W1, unsigned Wt0, unsigned Wt1) {
    unsigned w[2][3];
    int vj[NPX], vi[NPX];
    read_piece<2>(areaJ, row, dw, w);
    piece_values<NPX>(w, W0, W1, vj);
    read_piece<2>(areaI, row, dw, w);
    piece_values<NPX>(w, Wt0, Wt1, vi);
    int es = 0;
#pragma unroll
    for (int k = 0; k < NPX; k++) es += (k >= 5 && k >= nvalid) ? 0 : abs((vj[k] >> 9) - (vi[k] >> 9));
    __builtin_amdgcn_sched_barrier(0);
    return es;
}

/* three waves per SIMD: 168 registers, which the kernel fits but for a handful of spills */
#ifndef KLT21_WPE
#define KLT21_WPE 3
#endif
template <bool ONE_ITEM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KLT21_WPE, KLT21_WPE))) void k_klt21(KltParams p) {
    __shared__ __align__(16) unsigned s_patch[4][PATCH_DW];
    __shared__ __align__(16) unsigned s_patch2[4][PATCH_DW]; /* template patch for the residual pass */
    const int lane = threadIdx.x, sub = lane & 15, q = lane >> 4;
    unsigned* area = s_patch[q];
    unsigned* area2 = s_patch2[q];
    PatchLane P;
    P.pr = sub / 3; P.pk = sub - 3 * P.pr; P.on = sub < 12;
    /* lane 15 has no piece in rows 16..20: it shadows lane 14 and is left out of the sums */
    LanePieces L;
    L.sub = sub; L.hasB = sub < 15;
    {
        const int sb = L.hasB ? sub : 14, pB = sb - 3 * (sb / 3);
        L.rB = 16 + sb / 3; L.cB = 8 * pB; L.nB = pB == 2 ? 5 : 8;
    }
    const int win = 21;
    const float half = 10.f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int max_level = p.nlevels - 1;
    const unsigned n_items = (p.total_work + 3u) / 4u;

    /* Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Items are renumbered so
     * that all points of a frame run on one XCD (frames f, f+8, ... per XCD): the coarse pyramid levels
     * of a frame, which all its points read, then stay in that one L2 instead of being fetched by
     * eight.  Any renumbering is correct as long as it is a bijection; the tail that does not fill a
     * group of 8 frames keeps its order. */
    const unsigned per_frame = (unsigned)max(1, p.cap / 4);
    const unsigned grouped = n_items / (8u * per_frame) * (8u * per_frame);
    /* ONE_ITEM: the grid has a workgroup per item (every launch below 2^20 items), so nothing is invariant across a loop
     * and nothing is kept in registers -- or spilled -- for a next round */
    for (unsigned wg = blockIdx.x; wg < n_items; wg += ONE_ITEM ? n_items : gridDim.x) {
        unsigned item = wg;
        if (wg < grouped) {
            const unsigned xcd = wg & 7u, slot = wg >> 3, grp = slot / per_frame, within = slot - grp * per_frame;
            item = (grp * 8u + xcd) * per_frame + within;
        }
        const unsigned idx = item * 4u + (unsigned)q;
        if (idx >= p.total_work) continue;
        const int b = (int)(idx / (unsigned)p.cap), slot = (int)(idx - (unsigned)b * (unsigned)p.cap);
        const int n = p.n_pts ? min(p.n_pts[b], p.cap) : p.cap;
        if (slot >= n) {
            if (sub == 0) { p.status[idx] = 0; if (p.err) p.err[idx] = 0.f; }
            continue;
        }
        const int pt = p.order ? (int)p.order[idx] : slot;
        const size_t o = (size_t)b * p.cap + pt;
        const float ppx = p.prev_pts[2 * o], ppy = p.prev_pts[2 * o + 1];
        float outx = p.use_initial ? p.next_pts[2 * o] : 0.f, outy = p.use_initial ? p.next_pts[2 * o + 1] : 0.f;
        int status = 1;
        float errv = 0.f;

        for (int level = max_level; level >= 0; level--) {
            const DImg& I = p.prev[level];
            const DImg& J = p.next[level];
            const uint8_t* ip = I.frame(b);
            const uint8_t* jp = J.frame(b);
            const float sc = __uint_as_float((unsigned)(127 - level) << 23); /* 2^-level */
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
            const unsigned Wt0 = (unsigned)iw00 | ((unsigned)iw10 << 16), Wt1 = (unsigned)iw01 | ((unsigned)iw11 << 16);

            /* request the template patch (rows ipy-1 .. ipy+22, columns ipx-1 .. ipx+22) and, right
             * behind it, the patch of the next image for the first pass */
            uint2 vI[6], vJ[6];
            fetch_patch<24>(I, ip, ipx - 1, ipy - 1, P, vI);
            nx -= half; ny -= half;
            float qx = nx, qy = ny;
            int inx = (int)floorf(qx), iny = (int)floorf(qy);
            bool inr = !(inx < -win || inx >= J.w || iny < -win || iny >= J.h);
            if (inr) fetch_patch<22>(J, jp, inx, iny, P, vJ);

            unsigned TxA[11], TyA[11], TxB[4], TyB[4];
            int a11 = 0, a12 = 0, a22 = 0, c1 = 0, c2 = 0;
            {
                store_patch<24>(vI, P, area);
                /* gradients are needed at columns ipx .. ipx+21, rows ipy .. ipy+21 */
                const bool interior = ipx >= 0 && ipy >= 0 && ipx + win < I.w && ipy + win < I.h;
                int b11 = 0, b12 = 0, b22 = 0, d1 = 0, d2 = 0;
                unsigned w[4][3], w2[4][3]; /* a piece's rows are read while the piece before it is computed */
                PieceCarry carry;
                /* wave-uniform choice: the masked variant only when some point of the wave needs it */
#define KLT_TEMPLATE_PIECES(MASKED)                                                                                        \
    read_piece<4>(area, sub, 0, w);                                                                                        \
    read_piece<4>(area, sub, 2, w2);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    piece_template<8, MASKED, false, true>(w, Wt0, Wt1, interior, ipx, ipy + sub, I.w, I.h, 8, TxA, TyA, a11, a12, a22, c1, c2, carry); \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    read_piece<4>(area, sub, 4, w);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    piece_template<8, MASKED, true, true>(w2, Wt0, Wt1, interior, ipx + 8, ipy + sub, I.w, I.h, 8, TxA + 4, TyA + 4, a11, a12, a22, c1, c2, carry); \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    read_piece<4>(area, L.rB, L.cB / 4, w2);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    piece_template<5, MASKED, true, false>(w, Wt0, Wt1, interior, ipx + 16, ipy + sub, I.w, I.h, 5, TxA + 8, TyA + 8, a11, a12, a22, c1, c2, carry); \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    piece_template<8, MASKED, false, false>(w2, Wt0, Wt1, interior, ipx + L.cB, ipy + L.rB, I.w, I.h, L.nB, TxB, TyB, b11, b12, b22, d1, d2, carry); \
    __builtin_amdgcn_sched_barrier(0);
                if (__builtin_amdgcn_ballot_w64(!interior) != 0) { KLT_TEMPLATE_PIECES(true) } else { KLT_TEMPLATE_PIECES(false) }
#undef KLT_TEMPLATE_PIECES
                if (L.hasB) { a11 += b11; a12 += b12; a22 += b22; c1 += d1; c2 += d2; }
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
            float pdx = 0.f, pdy = 0.f;
            bool err_pass = false;
            bool fresh = true; /* vJ holds a patch that is not in the LDS area yet */
            for (int j = 0;;) {
                if (!inr) {
                    if (level == 0) status = 0;
                    break;
                }
                bilinear_weights(qx - (float)inx, qy - (float)iny, iw00, iw01, iw10, iw11);
                const unsigned W0 = (unsigned)iw00 | ((unsigned)iw10 << 16), W1 = (unsigned)iw01 | ((unsigned)iw11 << 16);
                /* small steps keep the integer origin: the patch in the area is still the right one, only the
                 * interpolation weights change */
                if (fresh) store_patch<22>(vJ, P, area);
                if (err_pass) {
                    /* residual: the template intensities are the same interpolation of the previous
                     * image at the window origin */
                    uint2 vT[6];
                    fetch_patch<22>(I, ip, ipx, ipy, P, vT);
                    store_patch<22>(vT, P, area2);
                    int es = piece_residual<8>(area, area2, sub, 0, 8, W0, W1, Wt0, Wt1);
                    es += piece_residual<8>(area, area2, sub, 2, 8, W0, W1, Wt0, Wt1);
                    es += piece_residual<5>(area, area2, sub, 4, 5, W0, W1, Wt0, Wt1);
                    const int eb = piece_residual<8>(area, area2, L.rB, L.cB / 4, L.nB, W0, W1, Wt0, Wt1);
                    if (L.hasB) es += eb;
                    errv = (float)row_sum(es) * 1.f / (float)(32 * win * win);
                    break;
                }
                int b1 = 0, b2 = 0, e1 = 0, e2 = 0;
                {
                    /* a piece's rows are read from LDS while the piece before it is computed */
                    unsigned wa[2][3], wb[2][3];
                    read_piece<2>(area, sub, 0, wa);
                    read_piece<2>(area, sub, 2, wb);
                    __builtin_amdgcn_sched_barrier(0);
                    piece_pass_rows<8>(wa, W0, W1, TxA, TyA, b1, b2);
                    read_piece<2>(area, sub, 4, wa);
                    __builtin_amdgcn_sched_barrier(0);
                    piece_pass_rows<8>(wb, W0, W1, TxA + 4, TyA + 4, b1, b2);
                    read_piece<2>(area, L.rB, L.cB / 4, wb);
                    __builtin_amdgcn_sched_barrier(0);
                    piece_pass_rows<5>(wa, W0, W1, TxA + 8, TyA + 8, b1, b2);
                    piece_pass_rows<8>(wb, W0, W1, TxB, TyB, e1, e2);
                }
                if (L.hasB) { b1 += e1; b2 += e2; }
                const float fb1 = row_sum_to_float(b1 - c1) * FLT_SCALE, fb2 = row_sum_to_float(b2 - c2) * FLT_SCALE;
                const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
                nx += dx; ny += dy;
                outx = nx + half; outy = ny + half;
                bool done = ((double)dx * (double)dx + (double)dy * (double)dy <= p.epsilon);
                /* (double)|x| < 0.01  <=>  |x| <= the float below 0.01, which is (float)0.01 */
                const float kSmall = 0.01f;
                static_assert((double)0.01f < 0.01, "the float nearest 0.01 lies below it");
                if (!done && j > 0 && fabsf(dx + pdx) <= kSmall && fabsf(dy + pdy) <= kSmall) {
                    outx -= dx * 0.5f; outy -= dy * 0.5f;
                    done = true;
                }
                pdx = dx; pdy = dy;
                j++;
                if (done || j == p.max_iter) {
                    /* last pass: only the residual at the final position, and only where it is reported */
                    if (!(status && level == 0 && p.err)) break;
                    err_pass = true;
                    qx = outx - half; qy = outy - half;
                } else { qx = nx; qy = ny; }
                const int ninx = (int)floorf(qx), niny = (int)floorf(qy);
                fresh = ninx != inx || niny != iny;
                inx = ninx; iny = niny;
                inr = !(inx < -win || inx >= J.w || iny < -win || iny >= J.h);
                if (inr && fresh) fetch_patch<22>(J, jp, inx, iny, P, vJ);
            }
        }
        if (sub == 0) {
            p.next_pts[2 * o] = outx; p.next_pts[2 * o + 1] = outy;
            p.status[o] = (uint8_t)status;
            if (p.err) p.err[o] = errv;
        }
    }
}


/* ---- one wavefront per point --------------------------------------------------------------------------------------
 * For launches with so few points that four per wave leave most of the chip's SIMDs without a wave, the time of the
 * launch is the time of one wave, and that is the number of instructions on a point's path.  Here the 441 pixels of a
 * point are spread over 63 lanes instead of 16: lane l owns the single piece (window row l / 3, columns 8 * (l % 3) ..)
 * of 8, 8 or 5 pixels; lane 63 shadows lane 62 and is left out of the sums.  Same arithmetic, same exact integer
 * window sums, same order of the floating-point operations on them: results are those of k_klt21 bit for bit. */
__device__ static inline int wave_sum(int v) {
    v = row_sum(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ static inline float wave_sum_to_float(int v) {
    const int lo = wave_sum(v & 0xFFFF), hi = wave_sum(v >> 16);
    return __builtin_fmaf((float)hi, 65536.f, (float)lo);
}
/* lane (row, pk) requests bytes 8*pk .. 8*pk+7 of patch row `row` (0..20) and, the first lanes, of the rows from 21 on */
struct WideLane { int row, pk; bool on; };
template <int NROWS>
__device__ static inline void fetch_patch_wide(const DImg& im, const uint8_t* fp, int x0, int y0, const WideLane& P, uint2 (&v)[2]) {
    const bool padded = im.border >= KLT_NEED_BORDER;
    const bool cols_in = padded || (x0 >= 0 && x0 + 24 <= im.w);
    const bool inside = cols_in && (padded || (y0 >= 0 && y0 + NROWS <= im.h)); /* the same for every lane */
    const int x = x0 + 8 * P.pk;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = P.row + 21 * i;
        v[i] = make_uint2(0u, 0u);
        if (!P.on || r >= NROWS) continue;
        if (inside) { v[i] = ldu64(fp + (ptrdiff_t)(y0 + r) * im.pitch + x); continue; }
        const int y = padded ? y0 + r : reflect101(y0 + r, im.h);
        const uint8_t* src = fp + (ptrdiff_t)y * im.pitch;
        if (cols_in) v[i] = ldu64(src + x);
        else {
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                lo |= (unsigned)src[reflect101(x + k, im.w)] << (8 * k);
                hi |= (unsigned)src[reflect101(x + 4 + k, im.w)] << (8 * k);
            }
            v[i] = make_uint2(lo, hi);
        }
    }
}
template <int NROWS>
__device__ static inline void store_patch_wide(const uint2 (&v)[2], const WideLane& P, unsigned* area) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = P.row + 21 * i;
        if (P.on && r < NROWS) *reinterpret_cast<uint2*>(area + r * PATCH_ROW_DW + 2 * P.pk) = v[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__global__ __launch_bounds__(64) void k_klt21_wide(KltParams p) {
    __shared__ __align__(16) unsigned s_patch[PATCH_DW];
    __shared__ __align__(16) unsigned s_patch2[PATCH_DW];
    unsigned* area = s_patch;
    unsigned* area2 = s_patch2;
    const int lane = threadIdx.x;
    const bool counted = lane < 63;
    WideLane P;
    P.on = counted;
    {
        const int l = counted ? lane : 62;
        P.row = l / 3; P.pk = l - 3 * P.row;
    }
    const int row = P.row, dw = 2 * P.pk, nvalid = P.pk == 2 ? 5 : 8;
    const int win = 21;
    const float half = 10.f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int max_level = p.nlevels - 1;

    /* all points of a frame on one XCD, as in k_klt21 (one workgroup per point here) */
    const unsigned per_frame = (unsigned)max(1, p.cap);
    const unsigned grouped = p.total_work / (8u * per_frame) * (8u * per_frame);
    unsigned idx = blockIdx.x;
    if (idx < grouped) {
        const unsigned xcd = idx & 7u, slot = idx >> 3, grp = slot / per_frame, within = slot - grp * per_frame;
        idx = (grp * 8u + xcd) * per_frame + within;
    }
    if (idx >= p.total_work) return;
    const int b = (int)(idx / (unsigned)p.cap), slot = (int)(idx - (unsigned)b * (unsigned)p.cap);
    const int n = p.n_pts ? min(p.n_pts[b], p.cap) : p.cap;
    if (slot >= n) {
        if (lane == 0) { p.status[idx] = 0; if (p.err) p.err[idx] = 0.f; }
        return;
    }
    const int pt = p.order ? (int)p.order[idx] : slot;
    const size_t o = (size_t)b * p.cap + pt;
    const float ppx = p.prev_pts[2 * o], ppy = p.prev_pts[2 * o + 1];
    float outx = p.use_initial ? p.next_pts[2 * o] : 0.f, outy = p.use_initial ? p.next_pts[2 * o + 1] : 0.f;
    int status = 1;
    float errv = 0.f;

    for (int level = max_level; level >= 0; level--) {
        const DImg& I = p.prev[level];
        const DImg& J = p.next[level];
        const uint8_t* ip = I.frame(b);
        const uint8_t* jp = J.frame(b);
        const float sc = __uint_as_float((unsigned)(127 - level) << 23); /* 2^-level */
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
        const unsigned Wt0 = (unsigned)iw00 | ((unsigned)iw10 << 16), Wt1 = (unsigned)iw01 | ((unsigned)iw11 << 16);

        uint2 vI[2], vJ[2];
        fetch_patch_wide<24>(I, ip, ipx - 1, ipy - 1, P, vI);
        nx -= half; ny -= half;
        float qx = nx, qy = ny;
        int inx = (int)floorf(qx), iny = (int)floorf(qy);
        bool inr = !(inx < -win || inx >= J.w || iny < -win || iny >= J.h);
        if (inr) fetch_patch_wide<22>(J, jp, inx, iny, P, vJ);

        unsigned Tx[4], Ty[4];
        int a11 = 0, a12 = 0, a22 = 0, c1 = 0, c2 = 0;
        {
            store_patch_wide<24>(vI, P, area);
            const bool interior = ipx >= 0 && ipy >= 0 && ipx + win < I.w && ipy + win < I.h;
            unsigned w[4][3];
            PieceCarry carry;
            read_piece<4>(area, row, dw, w);
            if (interior)
                piece_template<8, false, false, false>(w, Wt0, Wt1, true, ipx + 4 * dw, ipy + row, I.w, I.h, nvalid, Tx, Ty, a11, a12, a22, c1, c2, carry);
            else
                piece_template<8, true, false, false>(w, Wt0, Wt1, false, ipx + 4 * dw, ipy + row, I.w, I.h, nvalid, Tx, Ty, a11, a12, a22, c1, c2, carry);
            if (!counted) { a11 = 0; a12 = 0; a22 = 0; c1 = 0; c2 = 0; }
        }
        const float A11 = wave_sum_to_float(a11) * FLT_SCALE, A12 = wave_sum_to_float(a12) * FLT_SCALE,
                    A22 = wave_sum_to_float(a22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
        if ((double)minEig < p.min_eig_thr || D < FLT_EPSILON) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        float pdx = 0.f, pdy = 0.f;
        bool err_pass = false;
        bool fresh = true;
        for (int j = 0;;) {
            if (!inr) {
                if (level == 0) status = 0;
                break;
            }
            bilinear_weights(qx - (float)inx, qy - (float)iny, iw00, iw01, iw10, iw11);
            const unsigned W0 = (unsigned)iw00 | ((unsigned)iw10 << 16), W1 = (unsigned)iw01 | ((unsigned)iw11 << 16);
            if (fresh) store_patch_wide<22>(vJ, P, area);
            if (err_pass) {
                uint2 vT[2];
                fetch_patch_wide<22>(I, ip, ipx, ipy, P, vT);
                store_patch_wide<22>(vT, P, area2);
                int es = piece_residual<8>(area, area2, row, dw, nvalid, W0, W1, Wt0, Wt1);
                if (!counted) es = 0;
                errv = (float)wave_sum(es) * 1.f / (float)(32 * win * win);
                break;
            }
            int b1 = 0, b2 = 0;
            piece_pass<8>(area, row, dw, W0, W1, Tx, Ty, b1, b2);
            if (!counted) { b1 = 0; b2 = 0; }
            const float fb1 = wave_sum_to_float(b1 - c1) * FLT_SCALE, fb2 = wave_sum_to_float(b2 - c2) * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            bool done = ((double)dx * (double)dx + (double)dy * (double)dy <= p.epsilon);
            const float kSmall = 0.01f; /* see k_klt21 */
            if (!done && j > 0 && fabsf(dx + pdx) <= kSmall && fabsf(dy + pdy) <= kSmall) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                done = true;
            }
            pdx = dx; pdy = dy;
            j++;
            if (done || j == p.max_iter) {
                if (!(status && level == 0 && p.err)) break;
                err_pass = true;
                qx = outx - half; qy = outy - half;
            } else { qx = nx; qy = ny; }
            const int ninx = (int)floorf(qx), niny = (int)floorf(qy);
            fresh = ninx != inx || niny != iny;
            inx = ninx; iny = niny;
            inr = !(inx < -win || inx >= J.w || iny < -win || iny >= J.h);
            if (inr && fresh) fetch_patch_wide<22>(J, jp, inx, iny, P, vJ);
        }
    }
    if (lane == 0) {
        p.next_pts[2 * o] = outx; p.next_pts[2 * o + 1] = outy;
        p.status[o] = (uint8_t)status;
        if (p.err) p.err[o] = errv;
    }
}

/* Launches of at most this many (frame, point) slots take one wave per point.  Measured on 1080p pairs with 1000 points
 * per frame, tracker alone, four points per wave against one: 1 frame 47 -> 32 us, 2 frames 58 -> 35, 4 frames 65 -> 52,
 * 8 frames 69 -> 82 (from there on the chip is full either way and the quarter-filled lanes of the scalar work cost more
 * than the shorter path saves).  VFE_KLT_WIDE_MAX overrides the limit for measurements; it is read at every call. */
#define KLT21_WIDE_MAX 5000u
bool klt21_one_wave_per_point(unsigned total_work) {
    const char* e = getenv("VFE_KLT_WIDE_MAX");
    return total_work <= (e ? (unsigned)strtoul(e, nullptr, 10) : KLT21_WIDE_MAX);
}

void klt21_launch(const KltParams& p, hipStream_t stream) {
    if (klt21_one_wave_per_point(p.total_work)) {
        hipLaunchKernelGGL(k_klt21_wide, dim3(p.total_work), dim3(64), 0, stream, p);
        return;
    }
    const unsigned items = (p.total_work + 3u) / 4u;
    if (items <= (1u << 20)) hipLaunchKernelGGL(k_klt21<true>, dim3(items), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL(k_klt21<false>, dim3(1u << 20), dim3(64), 0, stream, p);
}
