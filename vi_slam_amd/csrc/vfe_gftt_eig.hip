/* Shi-Tomasi eigenvalue pass (cv::cornerMinEigenVal with blockSize 3, Sobel 3, followed by the
 * 3x3 local-maximum test of cv::goodFeaturesToTrack), register-marching.
 *
 * One wavefront owns a strip of 64 lanes x 4 columns and walks down a band of rows; per image row a
 * lane issues one coalesced dword load.  Everything else stays in registers as rings of three rows:
 * the horizontal Sobel row filters (so each image row is converted and filtered once, not once per
 * row it contributes to), the horizontal sums of the gradient products, the eigenvalues and their
 * horizontal maxima.  The loop is unrolled three times so that the ring slots are compile-time
 * constants and rows never have to be moved; the pipeline fill is unrolled too, so that a slot
 * written in a step is written unconditionally and the slot it replaces is dead.  Values of the
 * neighbouring columns come from the adjacent lanes by cross-lane shuffles; lanes 0, 1, 62, 63 are
 * halo (240 output columns per strip).  No LDS tiles, no barriers.
 *
 * The kernel is bound by vector-ALU issue, so a lane's four pixels are kept as two float pairs and
 * the element-wise stages run as packed fp32 operations (two pixels per instruction).  The
 * arithmetic (operation order, roundings, sign of the gradients where the product plane is a
 * reflection) is that of the CPU code, so results stay bit-exact: (float)(d - a) is formed as
 * float(d) - float(a), which is the same number, and the square root is refined to the correctly
 * rounded float.
 *
 * Local maxima above the running quality threshold are staged per wave in LDS and flushed with
 * one reservation atomic per batch of keys. */
#include "vfe_gftt.h"
#include "vfe_sqrt_rn.h"
#include <type_traits>

#define EV_STAGE 256

template <int N> using ic = std::integral_constant<int, N>;

__device__ static inline float lane_up(float v) { return __uint_as_float(lane_up_u(__float_as_uint(v))); }     /* value of lane - 1 */
__device__ static inline float lane_down(float v) { return __uint_as_float(lane_down_u(__float_as_uint(v))); } /* value of lane + 1 */

/* maximum of three floats as one instruction.  None of the values compared here can be a NaN, which the compiler
 * cannot know: through fmaxf it quiets every operand first (v_max x, x) and rarely forms the three-operand maximum. */
__device__ static inline float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

/* a + b as a plain (unpacked) add whatever the compiler would like to pair it with */
__device__ static inline float add1(float a, float b) {
    float d;
    asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ static inline float mul1(float a, float b) { /* likewise a plain multiply */
    float d;
    asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ static inline float min2(float a, float b) { /* no NaNs here: one instruction, no quieting */
    float d;
    asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

#ifndef EV_WPB
#define EV_WPB 4 /* waves per workgroup */
#endif
#ifndef EV_WPE
#define EV_WPE 4 /* waves per SIMD the register budget is cut for */
#endif
template <bool WRITE_MAP, bool ALIGNED>
__global__ __launch_bounds__(64 * EV_WPB) __attribute__((amdgpu_waves_per_eu(EV_WPE, EV_WPE))) void k_gftt_eig(DImg img, int strips_x, int band_h, int bands_y, unsigned total_work,
                                                  float* __restrict__ eigmap, u32* __restrict__ frame_max,
                                                  u32* __restrict__ cand_count, u64* __restrict__ cand, u32 seg_cap,
                                                  double quality, int band_mod, int band_sel) {
    __shared__ u64 s_stage[EV_WPB][EV_STAGE];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* wave-uniform: row counters and addresses stay scalar */
    const unsigned wave = blockIdx.x * (unsigned)EV_WPB + wv, nwaves = gridDim.x * (unsigned)EV_WPB;
    const int W = img.w, H = img.h;
    const double scale = 1.0 / ((1 << 2) * 3 * 255.0);
    const float k_o = (float)scale, k_c = (float)(2.0 * scale);
    u64* stage = s_stage[wv];
    const u64 lt_mask = (1ull << lane) - 1ull;
    const bool out_lane = lane >= 2 && lane < 62;
    const u64 inner_lanes = 0x7FFFFFFFFFFFFFFEull; /* lanes 1..62 */

    for (unsigned work = wave; work < total_work; work += nwaves) {
        const int b = work % img.batch, t_item = work / img.batch;
        const int sxi = t_item % strips_x, byi = t_item / strips_x;
        /* two-pass launches: this pass takes every band_mod-th band (band_sel != 0) or all the others */
        if (band_mod > 0 && ((byi % band_mod == band_mod / 2) != (band_sel != 0))) continue;
        const int X0 = sxi * EV_OUT + 4 * (lane - 2); /* image column of this lane's pixel 0 */
        const int y0 = byi * band_h, y1 = min(y0 + band_h, H);
        const uint8_t* ip = img.frame(b);
        /* 0x00020000: untyped 32-bit data; every access of a frame (plus the row past its end) lies inside the range */
        const __amdgpu_buffer_rsrc_t frame_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ip, 0, ALIGNED ? (H + 1) * img.pitch : 0, 0x00020000);
        u32 seen = 0;
        float prune = 0.f, wmax = 0.f;
        float prune_twice = -1.f; /* rows whose every min(sxx, syy) = 2 min(a, c) is <= this are skipped (negative: none) */
        if (!WRITE_MAP) {
            seen = __hip_atomic_load(&frame_max[b * GF_CSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wmax = __uint_as_float(seen);
            prune = (float)((double)wmax * quality);
            prune_twice = 2.f * (prune * 0.5f - 2e-7f);
        }
        int nbuf = 0;
        const int seg = t_item % GF_NSEG;
        u64* gdst = cand + ((size_t)b * GF_NSEG + seg) * seg_cap;
        u32* gcnt = cand_count + (size_t)(b * GF_NSEG + seg) * GF_CSTRIDE;
        /* column flags.  xflip: the product plane at x = -1 and x = W is the reflection of the plane,
         * i.e. the horizontal gradient there changes sign.  Only strips that touch the left or
         * right rim of the image have any lane with a flag set. */
        const bool edge_strip = sxi == 0 || sxi * EV_OUT + 4 * 62 > W;
        bool xin[4], xflip[4], cok[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = X0 + k;
            xin[k] = x >= 0 && x < W; xflip[k] = (x == -1 || x == W);
            cok[k] = out_lane && x >= 1 && x <= W - 2;
        }
        const bool fast_load = ALIGNED && X0 >= 0 && X0 + 3 < W;
        const bool dead_load = X0 + 3 < -4 || X0 > W + 4;
        /* Rim strips with dword-aligned rows: every lane loads one in-row dword (its own where that lies inside the row) and
         * the columns outside the image -- only x = -2, -1 and W, W + 1 feed anything that is kept -- are filled in from the
         * lane's own and its two neighbours' dwords by two byte permutes with per-lane selectors, fixed for the item: the
         * reflected source column 2 W - 2 - x or -x is at most one lane away.  No byte loads, no per-row reflection. */
        const int X0c = min(max(X0, 0), (W - 1) & ~3);
        u32 sel_prev = 0x07060504u, sel_next = 0x03020100u; /* identity: the lane's own dword */
        if (ALIGNED && edge_strip) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = X0 + k;
                int s = k; /* byte of the own dword; columns nobody reads keep whatever the lane loaded */
                if (x < 0 || x >= W) s = reflect101(x, W) - X0;
                if (s >= -4 && s < 0) sel_prev = (sel_prev & ~(0xFFu << (8 * k))) | ((u32)(s + 4) << (8 * k));
                else if (s >= 0 && s < 4) sel_prev = (sel_prev & ~(0xFFu << (8 * k))) | ((u32)(4 + s) << (8 * k));
                else if (s >= 4 && s < 8) sel_next = (sel_next & ~(0xFFu << (8 * k))) | ((u32)s << (8 * k));
            }
        }

        /* the item body, specialised (wave-uniformly) for strips away from the rim: all loads are plain
         * dword loads and no column needs a flag */
        auto item = [&](auto edgec) {
        constexpr bool EDGE = decltype(edgec)::value;
        /* three-row rings of pixel pairs (pixels 0,1 and 2,3); the slot of the row handled in step t
         * is t mod 3 */
        f2 RX[3][2], RY[3][2];   /* Sobel row filters of image rows yy-2, yy-1, yy */
        f2 HS[3][3][2];          /* [slot][plane][pair]: horizontal sums of product rows P-2, P-1, P */
        f2 EG[3][2], HM[3][2];   /* eigenvalue rows E-2, E-1, E and their horizontal 3-maxima */
        bool dead_row[3] = {false, false, false}; /* wave-uniform: the slot's eigenvalue row was skipped (see below) */
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                RX[i][p] = RY[i][p] = f2{0.f, 0.f}; EG[i][p] = HM[i][p] = f2{-1.f, -1.f};
                HS[i][0][p] = HS[i][1][p] = HS[i][2][p] = f2{0.f, 0.f};
            }

        /* the lane's four pixels of image row yy (reflect-101) */
        auto load_row = [&](int yy) -> u32 {
            const uint8_t* row = ip + (size_t)reflect101(yy, H) * img.pitch;
            /* away from the rim: the frame as a raw buffer, the row as the scalar offset, the lane's column as the vector
             * offset -- no vector address arithmetic at all */
            if (ALIGNED && !EDGE) return (u32)__builtin_amdgcn_raw_buffer_load_b32(frame_rsrc, X0, reflect101(yy, H) * img.pitch, 0);
            if (ALIGNED) return (u32)__builtin_amdgcn_raw_buffer_load_b32(frame_rsrc, X0c, reflect101(yy, H) * img.pitch, 0);
            if (fast_load) return *reinterpret_cast<const uint32_t*>(row + X0);
            if (dead_load) return 0u;
            return (u32)row[reflect101(X0, W)] | ((u32)row[reflect101(X0 + 1, W)] << 8) |
                   ((u32)row[reflect101(X0 + 2, W)] << 16) | ((u32)row[reflect101(X0 + 3, W)] << 24);
        };
        u32 pre = load_row(y0 - 3);

        /* stc = how far the pipeline is filled: 1 row filters only, 2 + products, 3 + eigenvalues,
         * 4 + output */
        auto step = [&](auto phc, auto stc, auto ybc, int t) {
            constexpr int ST = decltype(stc)::value;
            constexpr bool YB = decltype(ybc)::value; /* the product or eigenvalue row of this step may lie outside the image */
            constexpr int s0 = decltype(phc)::value, s1 = (s0 + 2) % 3, s2 = (s0 + 1) % 3; /* newest, previous, oldest */
            const int yy = y0 - 3 + t;
            /* ---- image row yy (reflect-101): columns X0-1 .. X0+4 as floats, row filters ---- */
            {
                /* the row was requested a step ago; request the next one now (the row after the band's last is a
                 * valid address too and never used) */
                u32 rC = pre;
                pre = load_row(yy + 1);
                if (ALIGNED && EDGE) { /* columns outside the image from the neighbours' (raw) dwords */
                    const u32 t = __builtin_amdgcn_perm(rC, lane_up_u(rC), sel_prev);
                    rC = __builtin_amdgcn_perm(lane_down_u(rC), t, sel_next);
                }
                const u32 l = lane_up_u(rC), r = lane_down_u(rC);
                const float fl = (float)(l >> 24), fr = (float)(r & 255u);
                const float b0 = (float)(rC & 255u), b1 = (float)((rC >> 8) & 255u);
                const float b2 = (float)((rC >> 16) & 255u), b3 = (float)(rC >> 24);
                /* pixel k sees columns k-1, k, k+1.  The six values sit in three register pairs, which serve as the left and
                 * right neighbours of both pixel pairs; the centre pairs (b0, b1), (b2, b3) straddle them, so their products
                 * are plain multiplies (same issue cost per element, no moves to assemble the pairs) */
                const f2 A = {fl, b0}, C = {b1, b2}, C2 = {b3, fr};
                RX[s0][0] = C - A; RX[s0][1] = C2 - C;
                RY[s0][0] = f2{mul1(k_c, b0), mul1(k_c, b1)} + k_o * (A + C);
                RY[s0][1] = f2{mul1(k_c, b2), mul1(k_c, b3)} + k_o * (C + C2);
            }
            if constexpr (ST < 2) return;
            /* ---- gradient products at row P = yy - 1 and their horizontal 3-sums ---- */
            {
                const int P = yy - 1;
                const bool yflip = YB && (P == -1 || P == H);
                f2 cv[3][2];
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    f2 dx = k_c * RX[s1][p] + k_o * (RX[s2][p] + RX[s0][p]);
                    f2 dy = RY[s0][p] - RY[s2][p];
                    if (EDGE) {
                        if (xflip[2 * p]) dx.x = -dx.x;
                        if (xflip[2 * p + 1]) dx.y = -dx.y;
                    }
                    if (yflip) dy = -dy;
                    cv[0][p] = dx * dx; cv[1][p] = dx * dy; cv[2][p] = dy * dy;
                }
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    /* plain adds: a packed add costs as much as two plain ones, the pairs (l, c0), (c1, c2), (c3, r) would
                     * have to be assembled with moves first, and a plain add takes the neighbour lane's value directly
                     * (DPP operand) */
                    const f2 c01 = cv[pl][0], c23 = cv[pl][1];
                    const float l = lane_up(c23.y), r = lane_down(c01.x);
                    const float m01 = add1(c01.x, c01.y), m12 = add1(c01.y, c23.x), m23 = add1(c23.x, c23.y);
                    HS[s0][pl][0] = f2{add1(l + c01.x, c01.y), add1(m01, c23.x)};
                    HS[s0][pl][1] = f2{add1(m12, c23.y), m23 + r};
                }
            }
            if constexpr (ST < 3) return;
            /* ---- eigenvalue at row E = yy - 2 (sum rows E-1, E, E+1), and its horizontal 3-max ---- */
            {
                const int E = yy - 2;
                const bool yin = !YB || (E >= 0 && E < H);
                /* a = sxx / 2, c = syy / 2 are never formed: halving is exact, so a + c = (sxx + syy) / 2 and
                 * (a - c)^2 = (sxx - syy)^2 / 4 round to the same floats, and min(a, c) <= x is min(sxx, syy) <= 2 x */
                f2 sxx_[2], syy_[2], xy_[2], s_[2];
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    sxx_[p] = (HS[s2][0][p] + HS[s1][0][p]) + HS[s0][0][p];
                    xy_[p] = (HS[s2][1][p] + HS[s1][1][p]) + HS[s0][1][p];
                    syy_[p] = (HS[s2][2][p] + HS[s1][2][p]) + HS[s0][2][p];
                    s_[p] = sxx_[p] + syy_[p]; /* twice a + c */
                }
                /* A row in which no pixel of the wave can pass the quality threshold needs neither the square roots nor a
                 * place in the maximum or the candidate test.  The smaller eigenvalue is at most 2 min(a, c); as computed
                 * here (five roundings) at most 2 min(a, c) + 5 * 2^-24 (a + c), and a + c <= 1 for 8-bit pixels (|Sobel| <=
                 * 1020 / 3060 per component, nine products per sum, halved).  Rows with min(a, c) <= prune / 2 - 2e-7 for
                 * every pixel of lanes 1..62 (lanes 0 and 63 hold incomplete sums and feed nothing) are given -1 instead.
                 * prune <= the frame maximum seen so far, so such a row can neither raise the maximum nor hold a candidate,
                 * and as a neighbour in the 3x3 test any value <= prune loses against a candidate (> prune) exactly as the
                 * true one would.  Steady-state rows of the candidate pass only. */
                bool dead = false;
                if constexpr (!WRITE_MAP && !YB) {
                    const float tm = max3(max3(min2(sxx_[0].x, syy_[0].x), min2(sxx_[0].y, syy_[0].y), min2(sxx_[1].x, syy_[1].x)),
                                          min2(sxx_[1].y, syy_[1].y), 0.f);
                    dead = (__ballot(tm > prune_twice) & inner_lanes) == 0;
                }
                dead_row[s0] = dead;
                if (dead) {
                    EG[s0][0] = EG[s0][1] = HM[s0][0] = HM[s0][1] = f2{-1.f, -1.f};
                } else {
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const f2 d = sxx_[p] - syy_[p];
                        const f2 dd = d * d, bb = xy_[p] * xy_[p];
                        const f2 q = {__builtin_fmaf(dd.x, 0.25f, bb.x), __builtin_fmaf(dd.y, 0.25f, bb.y)}; /* the quarter is exact: one rounding, that of the sum */
                        const f2 rt = sqrt_rn2(q);
                        f2 e = __builtin_elementwise_fma(s_[p], f2{0.5f, 0.5f}, -rt); /* (a + c) - root: the halving is exact, one rounding */
                        if (!yin) e = f2{-1.f, -1.f};
                        if (EDGE) {
                            if (!xin[2 * p]) e.x = -1.f;
                            if (!xin[2 * p + 1]) e.y = -1.f;
                        }
                        EG[s0][p] = e;
                    }
                    const f2 e01 = EG[s0][0], e23 = EG[s0][1];
                    const float l = lane_up(e23.y), r = lane_down(e01.x);
                    HM[s0][0] = f2{max3(l, e01.x, e01.y), max3(e01.x, e01.y, e23.x)};
                    HM[s0][1] = f2{max3(e01.y, e23.x, e23.y), max3(e23.x, e23.y, r)};
                }
            }
            if constexpr (ST < 4) return;
            /* ---- output row N = yy - 3: value EG[s1], neighbourhood maxima HM[s2], HM[s1], HM[s0] ---- */
            const int N = yy - 3;
            const float v[4] = {EG[s1][0].x, EG[s1][0].y, EG[s1][1].x, EG[s1][1].y};
            if (WRITE_MAP) {
                if (out_lane) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (X0 + k < W) eigmap[((size_t)b * H + N) * W + X0 + k] = v[k];
                }
                return;
            }
            if (dead_row[s1]) return; /* nothing above prune in this row: no maximum to raise, no candidate */
            wmax = max3(wmax, HM[s1][0].y, HM[s1][1].x); /* the row's horizontal maxima over pixels 0-2 and 1-3 are in the lane already */
            if (N < 1 || N > H - 2) return;
            const float m[4] = {max3(HM[s2][0].x, HM[s1][0].x, HM[s0][0].x), max3(HM[s2][0].y, HM[s1][0].y, HM[s0][0].y),
                                max3(HM[s2][1].x, HM[s1][1].x, HM[s0][1].x), max3(HM[s2][1].y, HM[s1][1].y, HM[s0][1].y)};
            bool isc[4];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                /* prune >= 0, so v > prune also rejects non-positive responses */
                isc[k] = cok[k] && v[k] > prune && v[k] >= m[k];
                any |= isc[k];
            }
            if (__ballot(any) == 0) return;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u64 bm = __ballot(isc[k]);
                if (bm) {
                    const int cnt = __popcll(bm);
                    if (nbuf + cnt > EV_STAGE) { /* flush the staged keys */
                        u32 base = 0;
                        if (lane == 0) base = atomicAdd(gcnt, (u32)nbuf);
                        base = (u32)__shfl((int)base, 0);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        for (int i = lane; i < nbuf; i += 64)
                            if (base + i < seg_cap) gdst[base + i] = stage[i];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        nbuf = 0;
                    }
                    if (isc[k]) stage[nbuf + __popcll(bm & lt_mask)] = ((u64)__float_as_uint(v[k]) << 32) | ((u32)N << 16) | (u32)(X0 + k);
                    nbuf += cnt;
                }
            }
        };

        /* rows y0-3 .. y1+2: six steps fill the pipeline, then every step emits one output row.  Product and
         * eigenvalue rows outside the image occur in the fill steps of the top band and in the last two steps of the
         * bottom band only, so the steady-state loop carries no tests for them. */
        constexpr std::true_type yb{};
        constexpr std::false_type nyb{};
        step(ic<0>{}, ic<1>{}, yb, 0); step(ic<1>{}, ic<1>{}, yb, 1); step(ic<2>{}, ic<2>{}, yb, 2);
        step(ic<0>{}, ic<2>{}, yb, 3); step(ic<1>{}, ic<3>{}, yb, 4); step(ic<2>{}, ic<3>{}, yb, 5);
        const int T = 6 + (y1 - y0), T_in = min(T, H + 4 - y0); /* steps t < T_in have rows P = yy-1 and E = yy-2 inside */
        int t0 = 6;
        for (; t0 + 2 < T_in; t0 += 3) { /* whole triples only: a conditional step would keep the slot it replaces alive */
            step(ic<0>{}, ic<4>{}, nyb, t0);
            step(ic<1>{}, ic<4>{}, nyb, t0 + 1);
            step(ic<2>{}, ic<4>{}, nyb, t0 + 2);
            if constexpr (!WRITE_MAP) {
                /* Every 24 rows the wave shares what it knows about the frame maximum: all waves of a launch that fits on
                 * the chip at once start blind (maximum 0, nothing pruned, no row skipped), and the maximum only reaches
                 * memory when a wave ends.  Publishing the running maximum and reading the frame's back gives every wave
                 * a threshold close to the final one after its first rows.  Any lower bound of the final maximum is a
                 * valid pruning value: the selection applies the exact threshold afterwards. */
                if ((t0 - 6) % 24 == 21) {
                    float wm = out_lane ? wmax : 0.f;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
                    if (lane == 0 && __float_as_uint(wm) > seen) atomicMax(&frame_max[b * GF_CSTRIDE], __float_as_uint(wm));
                    const u32 now = __hip_atomic_load(&frame_max[b * GF_CSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    seen = max(max(seen, now), __float_as_uint(wm)); /* positive floats order like their bit patterns */
                    wmax = fmaxf(wmax, __uint_as_float(seen));
                    prune = (float)((double)__uint_as_float(seen) * quality);
                    prune_twice = 2.f * (prune * 0.5f - 2e-7f);
                }
            }
        }
        /* at most two steps left of the loop's range and two beyond it */
        if (t0 < T) step(ic<0>{}, ic<4>{}, yb, t0);
        if (t0 + 1 < T) step(ic<1>{}, ic<4>{}, yb, t0 + 1);
        if (t0 + 2 < T) step(ic<2>{}, ic<4>{}, yb, t0 + 2);
        if (t0 + 3 < T) step(ic<0>{}, ic<4>{}, yb, t0 + 3);
        };
        if (edge_strip) item(std::true_type{}); else item(std::false_type{});

        if (!WRITE_MAP) {
            if (nbuf) {
                u32 base = 0;
                if (lane == 0) base = atomicAdd(gcnt, (u32)nbuf);
                base = (u32)__shfl((int)base, 0);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (int i = lane; i < nbuf; i += 64)
                    if (base + i < seg_cap) gdst[base + i] = stage[i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            /* halo lanes hold eigenvalues built from missing neighbours: leave them out of the maximum */
            if (!out_lane) wmax = 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
            if (lane == 0 && wmax > 0.f && __float_as_uint(wmax) > seen) atomicMax(&frame_max[b * GF_CSTRIDE], __float_as_uint(wmax));
        }
    }
}

void gftt_eig_launch(const DImg& d, float* eigmap, u32* frame_max, u32* cand_count, u64* cand, u32 seg_cap,
                     double quality, hipStream_t st, int wgs_per_cu) {
    const EigGeom g = eig_geom(d.w, d.h, d.batch);
    const unsigned nblk = (g.total + EV_WPB - 1) / EV_WPB;
    /* one item per wave: items are tens of microseconds long and few per wave, so handing them
     * out through the dispatcher balances better than persistent waves with a fixed stride */
    dim3 grid(nblk < (1u << 22) ? nblk : (1u << 22));
    /* the dword-per-lane form addresses a frame as a raw buffer with 32-bit offsets: rows on dword boundaries, and a frame
     * (plus one row) below 2 GB */
    const bool aligned = ((uintptr_t)d.data % 4 == 0) && (d.pitch % 4 == 0) && (d.fstride % 4 == 0) &&
                         (size_t)(d.h + 1) * (size_t)d.pitch < ((size_t)1 << 31);
    if (eigmap) {
        hipLaunchKernelGGL((k_gftt_eig<true, false>), grid, dim3(64 * EV_WPB), 0, st, d, g.strips_x, g.band_h, g.bands_y, g.total,
                           eigmap, (u32*)nullptr, (u32*)nullptr, (u64*)nullptr, 0u, 0.0, 0, 0);
        return;
    }
    /* Candidates are pruned against the frame maximum seen so far.  In a launch of which a large part fits on the chip at
     * once those waves start with a maximum of 0 and all their local maxima become candidates (tens of thousands on a
     * 1080p frame, of which a tenth survives the final threshold), which the selection kernel then pays for.  Such
     * launches run in two passes: an eighth of the bands first, the others with the maximum those found.  Measured on
     * 1080p front-end steps: the selection 86 -> 52 us at 8 streams, the step +1 % at 8, +5 % at 12, +2.5 % at 64 streams
     * and +4 % at 32; nothing at 128, -1 % at 256 (their waves walk 64 rows and exchange the maximum on the way). */
    const bool two_pass = eig_strip_rows(d.w, d.h, d.batch) < 800000 && g.bands_y >= 8;
    /* at most wgs_per_cu workgroups per CU: each asks for so much LDS that one more does not fit, with half a share left
     * for the kernels it runs beside (160 KB per CU) */
    unsigned extra_lds = 0;
    if (wgs_per_cu > 0 && wgs_per_cu < 4) {
        const unsigned share = 163840u * 2u / (2u * (unsigned)wgs_per_cu + 1u), own = (unsigned)sizeof(u64) * EV_WPB * EV_STAGE + 512u;
        extra_lds = (share - own) / 256u * 256u;
    }
    for (int pass = two_pass ? 1 : 0; pass >= 0; pass--) {
        const int mod = two_pass ? 8 : 0;
        if (aligned)
            hipLaunchKernelGGL((k_gftt_eig<false, true>), grid, dim3(64 * EV_WPB), extra_lds, st, d, g.strips_x, g.band_h, g.bands_y, g.total,
                               (float*)nullptr, frame_max, cand_count, cand, seg_cap, quality, mod, pass);
        else
            hipLaunchKernelGGL((k_gftt_eig<false, false>), grid, dim3(64 * EV_WPB), 0, st, d, g.strips_x, g.band_h, g.bands_y, g.total,
                               (float*)nullptr, frame_max, cand_count, cand, seg_cap, quality, mod, pass);
    }
}
