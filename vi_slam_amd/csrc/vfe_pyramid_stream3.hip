/* cv::buildOpticalFlowPyramid levels 1, 2 and 3 from level 0 in ONE launch, streaming (no re-read of any level).
 *
 * Level 0 -> 1 is the walk of vfe_pyramid_stream.hip: a wavefront owns a strip of up to 62 x 8 level-1 columns (lanes 0
 * and 63 are halo), steps down a band of rows, one coalesced 16-byte load per lane and source row, requested six rows
 * ahead; horizontal [1 4 6 4 1] sums as v_dot4, vertical sums accumulated on arrival.  What is new:
 *
 *  - every level-1 row a wave completes (8 bytes per lane, still in registers) feeds a second instance of the same
 *    pipeline at half width -- 4 level-2 outputs per lane, the two halo bytes on either side from the neighbour lanes by
 *    DPP wave shifts -- and every level-2 row a third one (2 level-3 outputs per lane).  The strip's halo lanes hold enough
 *    valid level-1 columns for level 2's halo, and enough level-2 columns for level 3's, so one strip geometry serves all
 *    three levels.
 *  - a workgroup is one strip of one frame, its (up to eight) waves the bands of that strip.  A band yields the level-2
 *    and level-3 rows whose five source rows lie inside it; the rows at band seams (two of level 2, three of level 3 per
 *    seam) and at the image's top and bottom (reflection) need rows of two bands: every wave leaves its first and last
 *    few level-1 and level-2 rows in LDS, the workgroup meets when the walks are done, and each wave then forms the
 *    rows of the seam below it (wave 0 also the rows at the top).  No band reads another band's source rows, no
 *    workgroup waits for another.
 *  - the reflect-101 border of all three levels goes out with the rows, from registers: a row near the top or bottom is
 *    stored once more at its mirrored position, and the lanes next to the image's left / right rim store their bytes
 *    reversed into the margin (shifted by one column with the neighbour lane's byte the horizontal filter fetched
 *    anyway, so that whole lanes cover the 32 margin columns).  Re-reading what was just written instead costs a
 *    seventh of the launch's HBM traffic: the lines have left L2 by the time a workgroup is done.  Strips are of equal
 *    width.
 *
 * The loop is kept lean (32-bit running store offsets against uniform bases, unconditional stores into LDS rings for the
 * rows the seams need, loads clamped into the image instead of masked) and every load goes into the registers of the row
 * just consumed: loaded into fresh registers, the compiler copies them back at the end of every trip round the loop and
 * waits for all six loads in flight to do so.
 *
 * Geometry the kernel is built for (the launcher checks it; anything else takes the per-level kernels): level-0 width a
 * multiple of 16 and height a multiple of 8 (all lanes' 8 / 4 / 2 outputs whole, all levels exact halves), 16-byte aligned
 * source rows, destination levels with a 32-pixel border and a padded frame below 4 GB, level 3 at least 48 x 34. */
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
    int strips_x, lanes;  /* strips per frame, output lanes per strip (the last strip may use fewer) */
    int band_h, bands;    /* band_h in level-1 rows, a multiple of 4 */
};

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

/* A level as the kernel addresses it: `base` = the frame's padded origin (row -32, column -32), offsets from it 32-bit. */
struct P3Level {
    uint8_t* base;
    u32 pitch;
    int w, h;
    __device__ u32 at(int row, int col) const { return (u32)(row + P3_BORDER) * pitch + (u32)(col + P3_BORDER); }
};
/* What a lane needs to store its share of a level's rows: its first column X, whether it stores at all, and for the strips
 * at the image's rims which lanes own margin columns and how far from the lane's own bytes they lie. */
template <int NB>
struct P3Lane {
    bool out, rim_l, rim_r; /* rim_l / rim_r: uniform over the workgroup */
    bool pl, pr;            /* this lane stores into the left / right margin */
    u32 dl, dr;             /* margin bytes at offset - dl / offset + dr */
    __device__ P3Lane(const P3Level& L, int X, bool out_lane, bool rl, bool rr)
        : out(out_lane), rim_l(rl), rim_r(rr),
          pl(out_lane && X <= P3_BORDER - NB), pr(out_lane && X >= L.w - P3_BORDER),
          dl((u32)(2 * X + NB)), dr((u32)(2 * L.w - 2 * X - NB)) {}
};
/* One row of a level at offset `off` (the lane's NB bytes, in lo / hi) and its share of the border:
 *  - rows 1 .. 32 and h-33 .. h-2 once more at their mirrored rows;
 *  - left margin: columns -32 .. -1 mirror columns 32 .. 1.  The lanes with X <= 32 - NB store the reverse of columns
 *    X+1 .. X+NB (their own bytes shifted by one, the last from the right neighbour) at columns -(X+NB) .. -(X+1);
 *  - right margin: columns w .. w+31 mirror columns w-2 .. w-33.  The lanes with X >= w - 32 store the reverse of columns
 *    X-1 .. X+NB-2 (the first from the left neighbour) at columns 2w-X-NB .. 2w-X-1.
 * Called by all 64 lanes (the neighbour bytes come by DPP). */
template <int NB>
__device__ static inline void p3_emit(const P3Level& L, const P3Lane<NB>& ln, u32 off, int row, u32 lo, u32 hi) {
    u32 l0 = 0, l1 = 0, r0 = 0, r1 = 0;
    if (ln.rim_l) {
        const u32 nb = lane_down_u(lo); /* byte 0 = column X+NB */
        /* v_perm: bytes 0-3 of the pair are the second operand's, 4-7 the first's */
        if (NB == 8) { l0 = __builtin_amdgcn_perm(nb, hi, 0x01020304u); l1 = __builtin_amdgcn_perm(hi, lo, 0x01020304u); }
        else if (NB == 4) l0 = __builtin_amdgcn_perm(nb, lo, 0x01020304u);
        else l0 = __builtin_amdgcn_perm(nb, lo, 0x0c0c0104u);
    }
    if (ln.rim_r) {
        const u32 pb = lane_up_u(NB == 8 ? hi : lo); /* its last byte = column X-1 */
        if (NB == 8) { r0 = __builtin_amdgcn_perm(hi, lo, 0x03040506u); r1 = __builtin_amdgcn_perm(lo, pb, 0x03040506u); }
        else if (NB == 4) r0 = __builtin_amdgcn_perm(lo, pb, 0x03040506u);
        else r0 = __builtin_amdgcn_perm(lo, pb, 0x0c0c0104u);
    }
    auto put = [&](u32 o) {
        if (ln.out) {
            if (NB == 8) *reinterpret_cast<uint2*>(L.base + o) = make_uint2(lo, hi);
            else if (NB == 4) *reinterpret_cast<u32*>(L.base + o) = lo;
            else *reinterpret_cast<unsigned short*>(L.base + o) = (unsigned short)lo;
        }
        if (ln.rim_l && ln.pl) {
            if (NB == 8) { const uint2 v = make_uint2(l0, l1); __builtin_memcpy(L.base + (o - ln.dl), &v, 8); }
            else if (NB == 4) __builtin_memcpy(L.base + (o - ln.dl), &l0, 4);
            else { const unsigned short v = (unsigned short)l0; __builtin_memcpy(L.base + (o - ln.dl), &v, 2); }
        }
        if (ln.rim_r && ln.pr) {
            if (NB == 8) { const uint2 v = make_uint2(r0, r1); __builtin_memcpy(L.base + (o + ln.dr), &v, 8); }
            else if (NB == 4) __builtin_memcpy(L.base + (o + ln.dr), &r0, 4);
            else { const unsigned short v = (unsigned short)r0; __builtin_memcpy(L.base + (o + ln.dr), &v, 2); }
        }
    };
    put(off);
    if ((unsigned)(row - 1) < (unsigned)P3_BORDER) put(off - 2u * (u32)row * L.pitch);
    if ((unsigned)(L.h - 2 - row) < (unsigned)P3_BORDER) put(off + 2u * (u32)(L.h - 1 - row) * L.pitch);
}

template <bool ONCE> /* ONCE: level 0 is too large to stay in cache until its next reader: non-temporal loads */
__global__ __launch_bounds__(64 * P3_MAX_WAVES) void k_pyr3_stream(Pyr3 p) {
    /* rows of every band the seams need, all 64 lanes' bytes (halo lanes included: their valid columns are the halo):
     * the first three level-1 rows and a ring of the last four; the first four level-2 rows a band forms itself and a ring
     * of its last four */
    __shared__ uint2 s_f1[P3_MAX_WAVES][3][64], s_e1[P3_MAX_WAVES][4][64];
    __shared__ u32 s_f2[P3_MAX_WAVES][4][64], s_e2[P3_MAX_WAVES][4][64];
    const int lane = threadIdx.x & 63;
    const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* band */
    unsigned fb, sx;
    xcd_frame_of(blockIdx.x, (unsigned)p.strips_x, (unsigned)p.src.batch, fb, sx); /* a frame's strips on one XCD */
    const int b = (int)fb, sxi = (int)sx;
    const DImg& src = p.src;
    const P3Level L1 = {p.d1.data + p.d1.offset(b) - (ptrdiff_t)P3_BORDER * p.d1.pitch - P3_BORDER, (u32)p.d1.pitch, p.d1.w, p.d1.h};
    const P3Level L2 = {p.d2.data + p.d2.offset(b) - (ptrdiff_t)P3_BORDER * p.d2.pitch - P3_BORDER, (u32)p.d2.pitch, p.d2.w, p.d2.h};
    const P3Level L3 = {p.d3.data + p.d3.offset(b) - (ptrdiff_t)P3_BORDER * p.d3.pitch - P3_BORDER, (u32)p.d3.pitch, p.d3.w, p.d3.h};
    const int X0 = (sxi * p.lanes + lane - 1) * 8, X2 = X0 >> 1, X3 = X0 >> 2; /* the lane's first column per level */
    const bool out_lane = lane >= 1 && lane <= p.lanes && X0 < L1.w;
    const bool rim_l = sxi == 0, rim_r = sxi == p.strips_x - 1;
    const P3Lane<8> ln1(L1, X0, out_lane, rim_l, rim_r);
    const P3Lane<4> ln2(L2, X2, out_lane, rim_l, rim_r);
    const P3Lane<2> ln3(L3, X3, out_lane, rim_l, rim_r);
    const bool fix_l = X0 == 0, fix_r = X0 + 8 == L1.w; /* the lane at the image's left / right rim, on every level */
    const int oy0 = k * p.band_h, oy1 = min(oy0 + p.band_h, L1.h); /* level-1 rows of this band */
    const int r_lo = oy0 / 2 + 1;                                    /* level-2 rows oy0/2+1 .. oy1/2-2 have their five source rows in the band */
    const int i_lo = oy0 / 4 + 2;                                    /* level-3 rows oy0/4+2 .. oy1/4-2 likewise */

    u32 A2P[2] = {0, 0}, A2C[2] = {0, 0}, A2N[2] = {0, 0};
    u32 A3P = 0, A3C = 0, A3N = 0;
    u32 off1 = L1.at(oy0, X0), off2 = L2.at(r_lo, X2), off3 = L3.at(i_lo, X3); /* where the lane's next row of each level goes */

    /* a level-2 row of this lane (4 bytes) arrives: store, keep for the seams, feed level 3 */
    auto level2_row = [&](int r, u32 d) {
        p3_emit<4>(L2, ln2, off2, r, d, 0u);
        off2 += L2.pitch;
        if (r < r_lo + 4) s_f2[k][r - r_lo][lane] = d;
        s_e2[k][r & 3][lane] = d;
        u32 lw = lane_up_u(d), rb = lane_down_u(d) & 0xFFu;
        if (fix_l) lw = __builtin_amdgcn_perm(d, d, 0x01020c0cu);
        if (fix_r) rb = (d >> 16) & 0xFFu;
        const u32 g = p3_hsums4(d, lw, rb);
        if (!(r & 1)) {
            A3P += g; A3C += 6u * g; A3N = g + P3_RND;
            if (r / 2 - 1 >= i_lo) { /* the last one is row oy1/4 - 2 */
                p3_emit<2>(L3, ln3, off3, r / 2 - 1, p3_bytes2(A3P), 0u);
                off3 += L3.pitch;
            }
        } else {
            A3C += 4u * g; A3N += 4u * g;
            A3P = A3C; A3C = A3N;
        }
    };
    /* a level-1 row of this lane (8 bytes) arrives */
    auto level1_row = [&](int oy, u32 o0, u32 o1) {
        p3_emit<8>(L1, ln1, off1, oy, o0, o1);
        off1 += L1.pitch;
        if (oy < oy0 + 3) s_f1[k][oy - oy0][lane] = make_uint2(o0, o1);
        s_e1[k][oy & 3][lane] = make_uint2(o0, o1);
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
            if (r >= r_lo) level2_row(r, p3_bytes4(A2P[0], A2P[1])); /* the last one is row oy1/2 - 2 */
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
        /* halo lanes left or right of the image load a chunk inside it instead: what they hold is never used (the lanes at
         * the rim take their halo from their own bytes) */
        const u32 ixc = (u32)min(max(2 * X0, 0), src.w - 16);
        const uint8_t* sp = src.frame(b);
        auto load_row = [&](int iy) -> uint4 { /* source row iy, reflected at the image's top and bottom, never past iy1 */
            iy = min(iy, iy1);
            iy = iy < 0 ? -iy : iy >= src.h ? 2 * (src.h - 1) - iy : iy;
            const uint8_t* q = sp + (size_t)iy * src.pitch + ixc;
            if (!ONCE) return *reinterpret_cast<const uint4*>(q);
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(q));
            return make_uint4(v.x, v.y, v.z, v.w);
        };
        uint4 pre[P3_AHEAD];
#pragma unroll
        for (int q = 0; q < P3_AHEAD; q++) pre[q] = load_row(iy0 + q);
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
                u32 lw = lane_up_u(cur.w), rb = lane_down_u(cur.x) & 0xFFu;
                if (fix_l) lw = __builtin_amdgcn_perm(cur.x, cur.x, 0x01020c0cu);
                if (fix_r) rb = (cur.w >> 16) & 0xFFu;
                p3_hsums16(cur, lw, rb, hp);
                /* the row's registers are free now: the load for six rows on goes into them (issued before the row is
                 * consumed it would need registers of its own, and copies back at the end of every trip round the loop,
                 * each waiting for its load) */
                __builtin_amdgcn_sched_barrier(0);
                pre[2 * ph] = load_row(2 * m + P3_AHEAD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; q++) { A[P][q] += hp[q]; A[C][q] += 6u * hp[q]; A[N][q] = hp[q] + P3_RND; }
                const int oy = m - 1;
                if (oy >= oy0) level1_row(oy, p3_bytes4(A[P][0], A[P][1]), p3_bytes4(A[P][2], A[P][3]));
            }
            {   /* odd row 2m+1 */
                const uint4 cur = pre[2 * ph + 1];
                u32 lw = lane_up_u(cur.w), rb = lane_down_u(cur.x) & 0xFFu;
                if (fix_l) lw = __builtin_amdgcn_perm(cur.x, cur.x, 0x01020c0cu);
                if (fix_r) rb = (cur.w >> 16) & 0xFFu;
                p3_hsums16(cur, lw, rb, hp);
                __builtin_amdgcn_sched_barrier(0);
                pre[2 * ph + 1] = load_row(2 * m + 1 + P3_AHEAD);
                __builtin_amdgcn_sched_barrier(0);
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
        const int Y = min(s * p.band_h, L1.h), u = s - 1, l = s; /* upper / lower band */
        /* horizontal sums of level-1 rows Y-4 .. Y+2 (index j = row - (Y - 4)); Y is a multiple of 4 */
        u32 hs[7][2];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int rel = j - 4;
            uint2 v;
            if (top) v = s_f1[0][rel < -2 ? 0 : rel < 0 ? -rel : rel][lane]; /* rows -1, -2 mirror rows 1, 2 (rows -3, -4: unused) */
            else if (rel < 0) v = s_e1[u][rel + 4][lane];
            else if (bottom) v = s_e1[u][2 - rel][lane];                     /* rows H1, H1+1, H1+2 mirror H1-2, H1-3, H1-4 */
            else v = s_f1[l][rel][lane];
            u32 lw = lane_up_u(v.y), rb = lane_down_u(v.x) & 0xFFu;
            if (fix_l) lw = __builtin_amdgcn_perm(v.x, v.x, 0x01020c0cu);
            if (fix_r) rb = (v.y >> 16) & 0xFFu;
            p3_hsums8(v.x, v.y, lw, rb, hs[j]);
        }
        const int R = Y / 2; /* level-2 rows R-1 and R */
        u32 da = 0, db = 0;
        if (!top) {
            da = p3_bytes4(hs[0][0] + 4u * hs[1][0] + 6u * hs[2][0] + 4u * hs[3][0] + hs[4][0] + P3_RND,
                           hs[0][1] + 4u * hs[1][1] + 6u * hs[2][1] + 4u * hs[3][1] + hs[4][1] + P3_RND);
            p3_emit<4>(L2, ln2, L2.at(R - 1, X2), R - 1, da, 0u);
        }
        if (!bottom) {
            db = p3_bytes4(hs[2][0] + 4u * hs[3][0] + 6u * hs[4][0] + 4u * hs[5][0] + hs[6][0] + P3_RND,
                           hs[2][1] + 4u * hs[3][1] + 6u * hs[4][1] + 4u * hs[5][1] + hs[6][1] + P3_RND);
            p3_emit<4>(L2, ln2, L2.at(R, X2), R, db, 0u);
        }
        /* horizontal sums of level-2 rows R-4 .. R+4 (index j = row - (R - 4)) */
        u32 g[9];
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const int rel = j - 4;
            u32 d;
            if (rel == -1) d = top ? s_f2[0][0][lane] : da;                                /* top: row -1 mirrors row 1 */
            else if (rel == 0) d = bottom ? s_e2[u][(R - 2) & 3][lane] : db;               /* bottom: row H2 mirrors H2-2 */
            else if (rel < 0) d = top ? s_f2[0][1][lane] : s_e2[u][(R + rel) & 3][lane];   /* top: row -2 mirrors row 2 (-3, -4: unused) */
            else d = bottom ? 0u : s_f2[l][rel - 1][lane];
            u32 lw = lane_up_u(d), rb = lane_down_u(d) & 0xFFu;
            if (fix_l) lw = __builtin_amdgcn_perm(d, d, 0x01020c0cu);
            if (fix_r) rb = (d >> 16) & 0xFFu;
            g[j] = p3_hsums4(d, lw, rb);
        }
        const int I = Y / 4; /* level-3 rows I-1, I, I+1 */
        if (!top) p3_emit<2>(L3, ln3, L3.at(I - 1, X3), I - 1, p3_bytes2(g[0] + 4u * g[1] + 6u * g[2] + 4u * g[3] + g[4] + P3_RND), 0u);
        if (!bottom) {
            p3_emit<2>(L3, ln3, L3.at(I, X3), I, p3_bytes2(g[2] + 4u * g[3] + 6u * g[4] + 4u * g[5] + g[6] + P3_RND), 0u);
            p3_emit<2>(L3, ln3, L3.at(I + 1, X3), I + 1, p3_bytes2(g[4] + 4u * g[5] + 6u * g[6] + 4u * g[7] + g[8] + P3_RND), 0u);
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

/* strips of equal width (in lanes of 8 level-1 columns); false if the last strip would be narrower than the 17 lanes
 * (33 level-3 columns) a margin mirrors -- only frames tens of thousands of pixels wide */
static bool pyr3_strips(int w1, int* strips, int* lanes) {
    const int total = w1 / 8;
    *strips = div_up(total, P3_OUT_LANES);
    *lanes = div_up(total, *strips);
    return total - (*strips - 1) * *lanes >= 17;
}

static bool p3_aligned(const DImg& d, int a) {
    return d.sub_n == 1 && ((uintptr_t)d.data % a == 0) && (d.pitch % a == 0) && (d.fstride % a == 0);
}
static bool p3_small(const DImg& d) { return (long long)(d.h + 2 * P3_BORDER + 1) * d.pitch < (1ll << 32); }

bool pyr3_stream_ok(const DImg& s, const DImg& d1, const DImg& d2, const DImg& d3) {
    int bh, nb, sx, ln;
    return p3_aligned(s, 16) && p3_aligned(d1, 8) && p3_aligned(d2, 4) && p3_aligned(d3, 2) &&
           s.w % 16 == 0 && s.h % 8 == 0 &&
           d1.w == s.w / 2 && d1.h == s.h / 2 && d2.w == s.w / 4 && d2.h == s.h / 4 && d3.w == s.w / 8 && d3.h == s.h / 8 &&
           d1.border == P3_BORDER && d2.border == P3_BORDER && d3.border == P3_BORDER &&
           d1.batch == s.batch && d2.batch == s.batch && d3.batch == s.batch &&
           p3_small(d1) && p3_small(d2) && p3_small(d3) &&
           d3.w >= 48 && d3.h >= P3_BORDER + 2 && pyr3_bands(d1.h, &bh, &nb) && pyr3_strips(d1.w, &sx, &ln);
}

/* level 0 of a launch this large is read with non-temporal loads: it will not be in cache for its next reader anyway, and
 * kept out of L2 it leaves the lines being written there until they are whole (256 x 1080p: 154 us against 183 us with plain
 * loads; non-temporal stores on top: 183 us) */
#define P3_ONCE_BYTES (64ll << 20)

int pyr3_stream_launch(const DImg& s, const DImg& d1, const DImg& d2, const DImg& d3, hipStream_t st) {
    Pyr3 p;
    p.src = s; p.d1 = d1; p.d2 = d2; p.d3 = d3;
    if (!pyr3_bands(d1.h, &p.band_h, &p.bands)) { vfe_set_error("pyramid: no band split for %d rows", d1.h); return VFE_E_INVALID; }
    if (!pyr3_strips(d1.w, &p.strips_x, &p.lanes)) { vfe_set_error("pyramid: no strip split for %d columns", d1.w); return VFE_E_INVALID; }
    const unsigned long long blocks = (unsigned long long)p.strips_x * (unsigned)s.batch;
    if (blocks >= (1ull << 31)) { vfe_set_error("pyramid: too many strips"); return VFE_E_INVALID; }
    if ((long long)s.w * s.h * s.batch >= P3_ONCE_BYTES) hipLaunchKernelGGL(k_pyr3_stream<true>, dim3((unsigned)blocks), dim3(64 * p.bands), 0, st, p);
    else hipLaunchKernelGGL(k_pyr3_stream<false>, dim3((unsigned)blocks), dim3(64 * p.bands), 0, st, p);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
