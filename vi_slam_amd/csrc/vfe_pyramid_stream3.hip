/* cv::buildOpticalFlowPyramid levels 1, 2 and 3 from level 0 in ONE launch, streaming (no re-read of any level).
 *
 * Level 0 -> 1 is the walk of vfe_pyramid_stream.hip: a wavefront owns a strip of 62 x 8 level-1 columns (lanes 0 and 63
 * are halo), steps down a band of rows, one coalesced 16-byte load per lane and source row, requested six rows ahead;
 * horizontal [1 4 6 4 1] sums as v_dot4, vertical sums accumulated on arrival.  What is new:
 *
 *  - every level-1 row a wave completes (8 bytes per lane, still in registers) feeds a second instance of the same
 *    pipeline at half width -- 4 level-2 outputs per lane, the two halo bytes on either side from the neighbour lanes by
 *    DPP wave shifts -- and every level-2 row a third one (2 level-3 outputs per lane).  The strip's halo lanes hold enough
 *    valid level-1 columns for level 2's halo, and enough level-2 columns for level 3's, so one strip geometry serves all
 *    three levels.
 *  - a workgroup is one strip of one frame, its (up to eight) waves the bands of that strip.  A band yields the level-2
 *    and level-3 rows whose five source rows lie inside it; the rows at band seams (two of level 2, three of level 3 per
 *    seam) and at the image's top and bottom (reflection) need rows of two bands: every wave leaves its first and last
 *    few level-1 and level-2 rows in LDS, the workgroup meets ONCE when the walks are done, and each wave then forms the
 *    rows of the seam below it (wave 0 also the rows at the top).  No band reads another band's source rows, no
 *    workgroup waits for another.
 *  - all three levels get their reflect-101 border in the same pass.
 *
 * Geometry the kernel is built for (the launcher checks it; anything else takes the per-level kernels): level-0 width a
 * multiple of 16 and height a multiple of 8 (all lanes' 8 / 4 / 2 outputs whole, all levels exact halves), 16-byte aligned
 * source rows, destination levels with a 32-pixel border, level 3 at least 48 x 34. */
#include "vfe_common.h"
#include <type_traits>

typedef unsigned int u32;
template <int N> using ic = std::integral_constant<int, N>;

#define P3_OUT_LANES 62
#define P3_AHEAD 6
#define P3_BORDER 32
#define P3_MAX_WAVES 8
#define P3_RND 0x00800080u /* + 128 in both 16-bit halves: the rounding of (sum + 128) >> 8 */

struct Pyr3 {
    DImg src, d1, d2, d3;
    int strips_x, band_h, bands; /* band_h in level-1 rows, a multiple of 4 */
    int read_once;
};

/* source bytes IX0 .. IX0+15 of row iy (reflected); chunks outside the image are never read (their bytes that matter are
 * reflections the neighbour already holds) */
__device__ static inline uint4 p3_load_row(const DImg& src, const uint8_t* sp, int iy, int IX0, bool once) {
    if (IX0 < 0 || IX0 + 15 >= src.w) return make_uint4(0, 0, 0, 0);
    const uint8_t* row = sp + (size_t)reflect101(iy, src.h) * src.pitch;
    if (!once) return *reinterpret_cast<const uint4*>(row + IX0);
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(row + IX0));
    return make_uint4(v.x, v.y, v.z, v.w);
}

#define P3_WT 0x04060401u /* byte 0 (first source byte) x1, then x4, x6, x4 */
/* horizontal [1 4 6 4 1] sums of 16 / 8 / 4 source bytes -> 8 / 4 / 2 outputs packed in pairs; lw = the left
 * neighbour's last dword (source bytes -4 .. -1), rb = the source byte behind the last */
__device__ static inline void p3_hsums16(const uint4& c, u32 lw, u32 rb, u32 (&hp)[4]) {
    const u32 d[4] = {c.x, c.y, c.z, c.w};
    u32 r[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 prev = j == 0 ? lw : d[j - 1];
        const u32 nextb = j == 3 ? rb : (d[j + 1] & 0xFFu);
        r[2 * j] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d[j], prev, 2), P3_WT, (d[j] >> 16) & 0xFFu, false);
        r[2 * j + 1] = __builtin_amdgcn_udot4(d[j], P3_WT, nextb, false);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) hp[i] = r[2 * i] | (r[2 * i + 1] << 16);
}
__device__ static inline void p3_hsums8(u32 c0, u32 c1, u32 lw, u32 rb, u32 (&hp)[2]) {
    const u32 r0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c0, lw, 2), P3_WT, (c0 >> 16) & 0xFFu, false);
    const u32 r1 = __builtin_amdgcn_udot4(c0, P3_WT, c1 & 0xFFu, false);
    const u32 r2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c1, c0, 2), P3_WT, (c1 >> 16) & 0xFFu, false);
    const u32 r3 = __builtin_amdgcn_udot4(c1, P3_WT, rb, false);
    hp[0] = r0 | (r1 << 16); hp[1] = r2 | (r3 << 16);
}
__device__ static inline u32 p3_hsums4(u32 c, u32 lw, u32 rb) {
    const u32 r0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, lw, 2), P3_WT, (c >> 16) & 0xFFu, false);
    const u32 r1 = __builtin_amdgcn_udot4(c, P3_WT, rb, false);
    return r0 | (r1 << 16);
}
/* the result bytes of a pair of packed sums: (sum + 128) >> 8 of each half = bytes 1 and 3 */
__device__ static inline u32 p3_bytes4(u32 a0, u32 a1) { return __builtin_amdgcn_perm(a1, a0, 0x07050301u); }
__device__ static inline u32 p3_bytes2(u32 a0) { return __builtin_amdgcn_perm(a0, a0, 0x0c0c0301u); }

/* One row of a level: NB = 8 / 4 / 2 bytes per lane at column X of row oy, plus its share of the level's reflect-101
 * border: rows near the top and bottom once more at their mirrored position, the first / last lanes of a row reversed
 * into the left / right margin (column -c mirrors c, column DW-1+c mirrors DW-1-c, c = 1 .. 32). */
template <int NB>
__device__ static inline void p3_store_at(uint8_t* dp, int pitch, int DW, int oy, int X, u32 lo, u32 hi, bool out_lane, bool rim_strip) {
    uint8_t* r0 = dp + (ptrdiff_t)oy * pitch;
    if (out_lane) {
        if (NB == 8) *reinterpret_cast<uint2*>(r0 + X) = make_uint2(lo, hi);
        else if (NB == 4) *reinterpret_cast<u32*>(r0 + X) = lo;
        else *reinterpret_cast<unsigned short*>(r0 + X) = (unsigned short)lo;
    }
    if (rim_strip && out_lane) {
        const bool pl = X <= 32 - NB, pl1 = X == 32, pr = X >= DW - 32, pr1 = X == DW - 32 - NB;
        if (NB == 8) {
            const uint2 v = make_uint2(__builtin_amdgcn_perm(hi, hi, 0x00010203u), __builtin_amdgcn_perm(lo, lo, 0x00010203u));
            if (pl) __builtin_memcpy(r0 - X - 7, &v, 8);
            if (pr) __builtin_memcpy(r0 + 2 * (DW - 1) - X - 7, &v, 8);
            if (pl1) r0[-32] = (uint8_t)lo;
            if (pr1) r0[DW + 31] = (uint8_t)(hi >> 24);
        } else if (NB == 4) {
            const u32 v = __builtin_amdgcn_perm(lo, lo, 0x00010203u);
            if (pl) __builtin_memcpy(r0 - X - 3, &v, 4);
            if (pr) __builtin_memcpy(r0 + 2 * (DW - 1) - X - 3, &v, 4);
            if (pl1) r0[-32] = (uint8_t)lo;
            if (pr1) r0[DW + 31] = (uint8_t)(lo >> 24);
        } else {
            const unsigned short v = (unsigned short)(((lo & 0xFFu) << 8) | ((lo >> 8) & 0xFFu));
            if (pl) __builtin_memcpy(r0 - X - 1, &v, 2);
            if (pr) __builtin_memcpy(r0 + 2 * (DW - 1) - X - 1, &v, 2);
            if (pl1) r0[-32] = (uint8_t)lo;
            if (pr1) r0[DW + 31] = (uint8_t)(lo >> 8);
        }
    }
}
template <int NB>
__device__ static inline void p3_store(uint8_t* dp, int pitch, int DW, int DH, int oy, int X, u32 lo, u32 hi, bool out_lane, bool rim_strip) {
    p3_store_at<NB>(dp, pitch, DW, oy, X, lo, hi, out_lane, rim_strip);
    if (oy >= 1 && oy <= P3_BORDER) p3_store_at<NB>(dp, pitch, DW, -oy, X, lo, hi, out_lane, rim_strip);
    if (oy <= DH - 2 && oy >= DH - 1 - P3_BORDER) p3_store_at<NB>(dp, pitch, DW, 2 * (DH - 1) - oy, X, lo, hi, out_lane, rim_strip);
}

__global__ __launch_bounds__(64 * P3_MAX_WAVES) void k_pyr3_stream(Pyr3 p) {
    /* first / last rows of every band, all 64 lanes' bytes (halo lanes included: their valid columns are the halo) */
    __shared__ uint2 s_f1[P3_MAX_WAVES][3][64], s_e1[P3_MAX_WAVES][4][64]; /* level 1: first 3, last 4 rows */
    __shared__ u32 s_f2[P3_MAX_WAVES][4][64], s_e2[P3_MAX_WAVES][3][64];   /* level 2: first 4, last 3 rows a band forms itself */
    const bool once = p.read_once != 0;
    const int lane = threadIdx.x & 63;
    const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* band */
    unsigned fb, sx;
    xcd_frame_of(blockIdx.x, (unsigned)p.strips_x, (unsigned)p.src.batch, fb, sx); /* a frame's strips on one XCD */
    const int b = (int)fb, sxi = (int)sx;
    const DImg &src = p.src, &d1 = p.d1, &d2 = p.d2, &d3 = p.d3;
    const int W1 = d1.w, H1 = d1.h, W2 = d2.w, H2 = d2.h, W3 = d3.w, H3 = d3.h;
    const int X0 = (sxi * P3_OUT_LANES + lane - 1) * 8, X2 = X0 >> 1, X3 = X0 >> 2; /* the lane's first column per level */
    const int IX0 = 2 * X0;
    const uint8_t* sp = src.frame(b);
    uint8_t *dp1 = d1.data + d1.offset(b), *dp2 = d2.data + d2.offset(b), *dp3 = d3.data + d3.offset(b);
    const bool out_lane = lane >= 1 && lane <= P3_OUT_LANES && X0 >= 0 && X0 < W1;
    const bool fix_l = X0 == 0, fix_r = X0 + 8 == W1; /* the lane at the image's left / right rim, on every level */
    /* strips whose lanes may own margin columns of some level: left margin = columns <= 32 of a level (strip 0 only), right
     * margin = level-1 columns from W1 - 136 on (level 3's W3 - 34, the widest reach, in level-1 columns) */
    const bool rim_strip = sxi == 0 || (sxi + 1) * P3_OUT_LANES * 8 + 136 >= W1;
    const int oy0 = k * p.band_h, oy1 = min(oy0 + p.band_h, H1); /* level-1 rows of this band */
    const int r_lo = oy0 / 2 + 1, r_hi = oy1 / 2 - 2;              /* level-2 rows whose five source rows lie in the band */
    const int i_lo = oy0 / 4 + 2, i_hi = oy1 / 4 - 2;              /* level-3 rows likewise */

    u32 A2P[2] = {0, 0}, A2C[2] = {0, 0}, A2N[2] = {0, 0};
    u32 A3P = 0, A3C = 0, A3N = 0;

    /* a level-2 row of this lane (4 bytes) arrives: store, keep for the seams, feed level 3 */
    auto level2_row = [&](int r, u32 d) {
        p3_store<4>(dp2, d2.pitch, W2, H2, r, X2, d, 0u, out_lane, rim_strip);
        if (r - r_lo < 4) s_f2[k][r - r_lo][lane] = d;
        if (r >= r_hi - 2) s_e2[k][r - (r_hi - 2)][lane] = d;
        u32 lw = lane_up_u(d), rb = lane_down_u(d) & 0xFFu;
        if (fix_l) lw = __builtin_amdgcn_perm(d, d, 0x01020c0cu);
        if (fix_r) rb = (d >> 16) & 0xFFu;
        const u32 g = p3_hsums4(d, lw, rb);
        if (!(r & 1)) {
            A3P += g; A3C += 6u * g; A3N = g + P3_RND;
            const int i = r / 2 - 1;
            if (i >= i_lo && i <= i_hi) p3_store<2>(dp3, d3.pitch, W3, H3, i, X3, p3_bytes2(A3P), 0u, out_lane, rim_strip);
        } else {
            A3C += 4u * g; A3N += 4u * g;
            A3P = A3C; A3C = A3N;
        }
    };
    /* a level-1 row of this lane (8 bytes) arrives */
    auto level1_row = [&](int oy, u32 o0, u32 o1) {
        p3_store<8>(dp1, d1.pitch, W1, H1, oy, X0, o0, o1, out_lane, rim_strip);
        if (oy - oy0 < 3) s_f1[k][oy - oy0][lane] = make_uint2(o0, o1);
        if (oy >= oy1 - 4) s_e1[k][oy - (oy1 - 4)][lane] = make_uint2(o0, o1);
        u32 lw = lane_up_u(o1), rb = lane_down_u(o0) & 0xFFu;
        if (fix_l) lw = __builtin_amdgcn_perm(o0, o0, 0x01020c0cu);
        if (fix_r) rb = (o1 >> 16) & 0xFFu;
        u32 hs[2];
        p3_hsums8(o0, o1, lw, rb, hs);
        if (!(oy & 1)) {
            A2P[0] += hs[0]; A2P[1] += hs[1];
            A2C[0] += 6u * hs[0]; A2C[1] += 6u * hs[1];
            A2N[0] = hs[0] + P3_RND; A2N[1] = hs[1] + P3_RND;
            const int r = oy / 2 - 1;
            if (r >= r_lo && r <= r_hi) level2_row(r, p3_bytes4(A2P[0], A2P[1]));
        } else {
            A2C[0] += 4u * hs[0]; A2C[1] += 4u * hs[1];
            A2N[0] += 4u * hs[0]; A2N[1] += 4u * hs[1];
            A2P[0] = A2C[0]; A2P[1] = A2C[1];
            A2C[0] = A2N[0]; A2C[1] = A2N[1];
        }
    };

    /* ---- the walk: level 0 -> 1 as in k_pyr_down_stream, every completed row handed to level1_row ---- */
    {
        const int iy0 = 2 * oy0 - 2, iy1 = 2 * oy1; /* source rows iy0 .. iy1 */
        const bool src_l = IX0 == 0, src_r = IX0 + 16 == src.w;
        uint4 pre[P3_AHEAD];
#pragma unroll
        for (int q = 0; q < P3_AHEAD; q++) pre[q] = (iy0 + q <= iy1) ? p3_load_row(src, sp, iy0 + q, IX0, once) : make_uint4(0, 0, 0, 0);
        u32 A[3][4];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int q = 0; q < 4; q++) A[i][q] = 0;
        auto pair_step = [&](auto phc, int m) {
            constexpr int ph = decltype(phc)::value;
            constexpr int P = ph % 3, C = (ph + 1) % 3, N = (ph + 2) % 3;
            u32 hp[4];
            {   /* even row 2m */
                const uint4 cur = pre[2 * ph];
                if (2 * m + P3_AHEAD <= iy1) pre[2 * ph] = p3_load_row(src, sp, 2 * m + P3_AHEAD, IX0, once);
                u32 lw = lane_up_u(cur.w), rb = lane_down_u(cur.x) & 0xFFu;
                if (src_l) lw = __builtin_amdgcn_perm(cur.x, cur.x, 0x01020c0cu);
                if (src_r) rb = (cur.w >> 16) & 0xFFu;
                p3_hsums16(cur, lw, rb, hp);
#pragma unroll
                for (int q = 0; q < 4; q++) { A[P][q] += hp[q]; A[C][q] += 6u * hp[q]; A[N][q] = hp[q] + P3_RND; }
                const int oy = m - 1;
                if (oy >= oy0) level1_row(oy, p3_bytes4(A[P][0], A[P][1]), p3_bytes4(A[P][2], A[P][3]));
            }
            {   /* odd row 2m+1 */
                const uint4 cur = pre[2 * ph + 1];
                if (2 * m + 1 + P3_AHEAD <= iy1) pre[2 * ph + 1] = p3_load_row(src, sp, 2 * m + 1 + P3_AHEAD, IX0, once);
                u32 lw = lane_up_u(cur.w), rb = lane_down_u(cur.x) & 0xFFu;
                if (src_l) lw = __builtin_amdgcn_perm(cur.x, cur.x, 0x01020c0cu);
                if (src_r) rb = (cur.w >> 16) & 0xFFu;
                p3_hsums16(cur, lw, rb, hp);
#pragma unroll
                for (int q = 0; q < 4; q++) { A[C][q] += 4u * hp[q]; A[N][q] += 4u * hp[q]; }
            }
        };
        const int m_end = oy1 + 1;
        int m = oy0 - 1;
        for (; m + 2 < m_end; m += 3) {
            pair_step(ic<0>{}, m);
            pair_step(ic<1>{}, m + 1);
            pair_step(ic<2>{}, m + 2);
        }
        if (m < m_end) pair_step(ic<0>{}, m);
        if (m + 1 < m_end) pair_step(ic<1>{}, m + 1);
    }
    __syncthreads(); /* every band's first and last rows are in LDS */

    /* ---- the rows that need two bands: seam s lies above level-1 row Y = s * band_h (s = 0: the image's top, s = bands: its
     * bottom, where the missing rows are reflections).  Wave k forms the rows of seam k + 1, wave 0 also those of seam 0. ---- */
    auto seam = [&](int s) {
        const bool top = s == 0, bottom = s == p.bands;
        const int Y = min(s * p.band_h, H1), u = s - 1, l = s; /* upper / lower band */
        /* horizontal sums of level-1 rows Y-4 .. Y+2 (index j = row - (Y - 4)) */
        u32 hs[7][2];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int rel = j - 4;
            uint2 v;
            if (top) v = s_f1[0][rel < -2 ? 0 : rel < 0 ? -rel : rel][lane]; /* rows -1, -2 mirror rows 1, 2 (rows -3, -4: unused) */
            else if (rel < 0) v = s_e1[u][rel + 4][lane];
            else if (bottom) v = s_e1[u][2 - rel][lane];                 /* rows H1, H1+1, H1+2 mirror H1-2, H1-3, H1-4 */
            else v = s_f1[l][rel][lane];
            u32 lw = lane_up_u(v.y), rb = lane_down_u(v.x) & 0xFFu;
            if (fix_l) lw = __builtin_amdgcn_perm(v.x, v.x, 0x01020c0cu);
            if (fix_r) rb = (v.y >> 16) & 0xFFu;
            p3_hsums8(v.x, v.y, lw, rb, hs[j]);
        }
        u32 da = 0, db = 0; /* level-2 rows Y/2 - 1 and Y/2 */
        if (!top) {
            da = p3_bytes4(hs[0][0] + 4u * hs[1][0] + 6u * hs[2][0] + 4u * hs[3][0] + hs[4][0] + P3_RND,
                           hs[0][1] + 4u * hs[1][1] + 6u * hs[2][1] + 4u * hs[3][1] + hs[4][1] + P3_RND);
            p3_store<4>(dp2, d2.pitch, W2, H2, Y / 2 - 1, X2, da, 0u, out_lane, rim_strip);
        }
        if (!bottom) {
            db = p3_bytes4(hs[2][0] + 4u * hs[3][0] + 6u * hs[4][0] + 4u * hs[5][0] + hs[6][0] + P3_RND,
                           hs[2][1] + 4u * hs[3][1] + 6u * hs[4][1] + 4u * hs[5][1] + hs[6][1] + P3_RND);
            p3_store<4>(dp2, d2.pitch, W2, H2, Y / 2, X2, db, 0u, out_lane, rim_strip);
        }
        /* horizontal sums of level-2 rows Y/2-4 .. Y/2+4 (index j = row - (Y/2 - 4)) */
        u32 g[9];
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const int rel = j - 4;
            u32 d;
            if (rel == -1) d = top ? s_f2[0][0][lane] : da;                       /* top: row -1 mirrors row 1 */
            else if (rel == 0) d = bottom ? s_e2[u][2][lane] : db;                /* bottom: row H2 mirrors H2-2 */
            else if (rel < 0) d = top ? (rel == -2 ? s_f2[0][1][lane] : 0u) : s_e2[u][rel + 4][lane]; /* top: row -2 mirrors row 2 */
            else d = bottom ? 0u : s_f2[l][rel - 1][lane];
            u32 lw = lane_up_u(d), rb = lane_down_u(d) & 0xFFu;
            if (fix_l) lw = __builtin_amdgcn_perm(d, d, 0x01020c0cu);
            if (fix_r) rb = (d >> 16) & 0xFFu;
            g[j] = p3_hsums4(d, lw, rb);
        }
        if (!top) p3_store<2>(dp3, d3.pitch, W3, H3, Y / 4 - 1, X3, p3_bytes2(g[0] + 4u * g[1] + 6u * g[2] + 4u * g[3] + g[4] + P3_RND), 0u, out_lane, rim_strip);
        if (!bottom) {
            p3_store<2>(dp3, d3.pitch, W3, H3, Y / 4, X3, p3_bytes2(g[2] + 4u * g[3] + 6u * g[4] + 4u * g[5] + g[6] + P3_RND), 0u, out_lane, rim_strip);
            p3_store<2>(dp3, d3.pitch, W3, H3, Y / 4 + 1, X3, p3_bytes2(g[4] + 4u * g[5] + 6u * g[6] + 4u * g[7] + g[8] + P3_RND), 0u, out_lane, rim_strip);
        }
    };
    if (k == 0) seam(0);
    seam(k + 1);
}

/* band height (level-1 rows, a multiple of 4, at least 16) and band count (at most P3_MAX_WAVES) for a level 1 of h1 rows;
 * false if no such split exists */
static bool pyr3_bands(int h1, int* band_h, int* bands) {
    for (int bh = (div_up(h1, P3_MAX_WAVES) + 3) / 4 * 4; bh <= h1; bh += 4) {
        if (bh < 16) continue;
        const int nb = div_up(h1, bh), last = h1 - (nb - 1) * bh;
        if (nb <= P3_MAX_WAVES && last >= 16) { *band_h = bh; *bands = nb; return true; }
    }
    return false;
}

static bool p3_aligned(const DImg& d, int a) {
    return d.sub_n == 1 && ((uintptr_t)d.data % a == 0) && (d.pitch % a == 0) && (d.fstride % a == 0);
}

bool pyr3_stream_ok(const DImg& s, const DImg& d1, const DImg& d2, const DImg& d3) {
    int bh, nb;
    return p3_aligned(s, 16) && p3_aligned(d1, 8) && p3_aligned(d2, 4) && p3_aligned(d3, 2) &&
           s.w % 16 == 0 && s.h % 8 == 0 &&
           d1.w == s.w / 2 && d1.h == s.h / 2 && d2.w == s.w / 4 && d2.h == s.h / 4 && d3.w == s.w / 8 && d3.h == s.h / 8 &&
           d1.border == P3_BORDER && d2.border == P3_BORDER && d3.border == P3_BORDER &&
           d1.batch == s.batch && d2.batch == s.batch && d3.batch == s.batch &&
           d3.w >= 48 && d3.h >= P3_BORDER + 2 && pyr3_bands(d1.h, &bh, &nb);
}

#define P3_ONCE_BYTES (64ll << 20)

int pyr3_stream_launch(const DImg& s, const DImg& d1, const DImg& d2, const DImg& d3, hipStream_t st) {
    Pyr3 p;
    p.src = s; p.d1 = d1; p.d2 = d2; p.d3 = d3;
    if (!pyr3_bands(d1.h, &p.band_h, &p.bands)) { vfe_set_error("pyramid: no band split for %d rows", d1.h); return VFE_E_INVALID; }
    p.strips_x = div_up(d1.w / 8, P3_OUT_LANES);
    p.read_once = (long long)s.w * s.h * s.batch >= P3_ONCE_BYTES;
    const unsigned long long blocks = (unsigned long long)p.strips_x * (unsigned)s.batch;
    if (blocks >= (1ull << 31)) { vfe_set_error("pyramid: too many strips"); return VFE_E_INVALID; }
    hipLaunchKernelGGL(k_pyr3_stream, dim3((unsigned)blocks), dim3(64 * p.bands), 0, st, p);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
