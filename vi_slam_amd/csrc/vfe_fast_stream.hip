/* FAST-9/16 score + 3x3 non-max suppression for whole frames, streaming (cv::FAST semantics, same
 * scores and raster order as the tiled kernel in vfe_fast.hip).  One launch serves a set of images of
 * different sizes (the levels of an ORB pyramid), so that small levels fill the gaps large ones leave.
 *
 * One wavefront owns a strip of 62 x 4 columns (lanes 0 and 63 are halo) and walks down a band of rows.
 * A lane keeps the last image rows as three dwords each (its own four bytes and the neighbours'),
 * so every pixel of the 16-pixel circle is a byte of a register.  Two horizontally adjacent pixels are
 * handled together in packed 16-bit arithmetic.
 *
 * Per row every lane evaluates only a necessary condition on the four compass pixels (each opposite pair
 * must hold a pixel that differs from the centre by more than t).  It passes for a few percent of the
 * pixel pairs (3 % on a full-resolution level of a typical frame, a third on the coarsest ORB level), yet
 * nearly every 248-pixel row holds some, so a wave-wide skip of the 9-arc evaluation would almost never
 * fire.  Instead the pairs that pass are queued (row, lane, pair) in LDS; whenever 64 are waiting, every
 * lane takes one, reads its 7 x 12-byte window from a ring of the wave's last image rows in LDS and scores
 * it:
 *   score = max(t, max over 9-arcs of min(d), -min over 9-arcs of max(d)) - 1,   d = centre - circle,
 * a pixel is a corner iff the bracket exceeds t.  The sixteen 9-arc minima come from suffix minima of
 * d[k..7], d[k..15] and prefix minima of d[0..k], d[8..k]: arc k..k+8 is one suffix plus the prefix
 * eight places on (28 + 16 operations instead of 64 by doubling).  The two scores go to a ring of score
 * rows in LDS.  Suppression and output of a row follow FQ_K rows behind its compass test, reading the
 * score ring; a candidate older than that is scored at once even if fewer than 64 are waiting.  Scores of
 * three rows stay in registers for the 3x3 suppression; keypoints of a row segment are numbered in raster
 * order with ballots and written to the segment's slot together with the segment's count.
 *
 * The row loop is unrolled nine times so that the register slots of the image ring (a row is kept for six steps)
 * and of the three suppression rows are compile-time constants; a band is 9k - 8 rows high so that every wave runs
 * whole turns. */
#include "vfe_common.h"
#include "vfe_fast.h"
#include <type_traits>

typedef unsigned int u32;
typedef unsigned long long u64;
typedef short s2v __attribute__((ext_vector_type(2)));
typedef unsigned short u2v __attribute__((ext_vector_type(2)));
template <int N> using ic = std::integral_constant<int, N>;

__device__ static inline s2v as_s(u32 v) { return __builtin_bit_cast(s2v, v); }
__device__ static inline u32 as_u(s2v v) { return __builtin_bit_cast(u32, v); }
__device__ static inline u2v as_uh(u32 v) { return __builtin_bit_cast(u2v, v); }
__device__ static inline s2v pmin(s2v a, s2v b) { return __builtin_elementwise_min(a, b); }
__device__ static inline s2v pmax(s2v a, s2v b) { return __builtin_elementwise_max(a, b); }
__device__ static inline s2v splat(int v) { return s2v{(short)v, (short)v}; }

/* bytes i and i+1 (1 <= i <= 9) of the 12-byte window {l, c, r} as a pair of 16-bit values */
__device__ static inline s2v win_pair(const u32 (&row)[3], int i) {
    if (i + 1 <= 7) return as_s(__builtin_amdgcn_perm(row[1], row[0], 0x0c000c00u | (u32)i | ((u32)(i + 1) << 16)));
    return as_s(__builtin_amdgcn_perm(row[2], row[1], 0x0c000c00u | (u32)(i - 4) | ((u32)(i - 3) << 16)));
}

/* circle position k -> (dx, dy), dy positive = down (the order of cv::FAST's offsets) */
#define FS_DX(k) ((k) == 0 ? 0 : (k) == 1 ? 1 : (k) == 2 ? 2 : (k) == 3 ? 3 : (k) == 4 ? 3 : (k) == 5 ? 3 : (k) == 6 ? 2 : (k) == 7 ? 1 : \
                  (k) == 8 ? 0 : (k) == 9 ? -1 : (k) == 10 ? -2 : (k) == 11 ? -3 : (k) == 12 ? -3 : (k) == 13 ? -3 : (k) == 14 ? -2 : -1)
#define FS_DY(k) ((k) == 0 ? 3 : (k) == 1 ? 3 : (k) == 2 ? 2 : (k) == 3 ? 1 : (k) == 4 ? 0 : (k) == 5 ? -1 : (k) == 6 ? -2 : (k) == 7 ? -3 : \
                  (k) == 8 ? -3 : (k) == 9 ? -3 : (k) == 10 ? -2 : (k) == 11 ? -1 : (k) == 12 ? 0 : (k) == 13 ? 1 : (k) == 14 ? 2 : 3)

#define FQ_K 8               /* rows between the compass test of a row and its suppression */
#define FQ_IR 16             /* image ring (>= FQ_K + 7: rows c-3 of the oldest queued centre c .. the row being loaded); a power of two */
#define FQ_SR 16             /* score ring (>= FQ_K + 2: from a row's compass test to its last use as a neighbour); a power of two */
#define FQ_PERIOD 9          /* steps per turn of the unrolled row loop: the register ring of image rows must hold a row for six
                              * steps, the three suppression rows rotate with period three */
#define FQ_AHEAD FQ_PERIOD   /* image rows requested ahead: a row step is a few hundred cycles, a trip to HBM thousands */
#define FQ_QCAP 256          /* queue entries (63 left over + one row's 128 at most) */
static_assert(FQ_IR >= FQ_K + 7 && FQ_SR >= FQ_K + 2 && (FQ_IR & (FQ_IR - 1)) == 0 && (FQ_SR & (FQ_SR - 1)) == 0 && FQ_IR <= 32 && FQ_SR <= 16,
              "ring sizes: powers of two that cover the delay, and ring slots must fit the 5 + 4 bits of a queue entry");
static_assert(FQ_AHEAD == FQ_PERIOD && FQ_PERIOD % 3 == 0 && FQ_PERIOD >= 7, "the prefetch ring and the suppression rows share the period of the unrolled row loop");

/* necessary condition on the compass pixels of the pair (PX, PX+1): true if either pixel can be a corner.  The ring
 * slot PH holds the newest row (centre + 3), so centre + dy sits in slot (PH - 3 + dy) mod FQ_PERIOD; only the rows
 * centre - 3, centre, centre + 3 are read. */
template <int PX, int PH>
__device__ static inline bool fast_compass_pair(const u32 (&R)[FQ_PERIOD][3], int t, u32 sm) {
#define FS_ROW(dy) R[(PH - 3 + (dy) + 2 * FQ_PERIOD) % FQ_PERIOD]
    const s2v v = win_pair(FS_ROW(0), PX + 4);
    const s2v d0 = v - win_pair(FS_ROW(FS_DY(0)), PX + FS_DX(0) + 4), d4 = v - win_pair(FS_ROW(FS_DY(4)), PX + FS_DX(4) + 4);
    const s2v d8 = v - win_pair(FS_ROW(FS_DY(8)), PX + FS_DX(8) + 4), d12 = v - win_pair(FS_ROW(FS_DY(12)), PX + FS_DX(12) + 4);
#undef FS_ROW
    const s2v a0 = pmax(d0, splat(0) - d0), a8 = pmax(d8, splat(0) - d8);
    const s2v a4 = pmax(d4, splat(0) - d4), a12 = pmax(d12, splat(0) - d12);
    const s2v m = pmin(pmax(a0, a8), pmax(a4, a12));
    return (m.x > (short)t && (sm & 0xFFFFu)) || (m.y > (short)t && (sm >> 16));
}

/* scores of the pixel pair at bytes 4, 5 of the 12-byte windows R[0..6] (centre row R[3]); no shortcut */
__device__ static inline u32 fast_score_window(const u32 (&R)[7][3], int t) {
    const s2v v = win_pair(R[3], 4);
    s2v d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - win_pair(R[3 + FS_DY(k)], FS_DX(k) + 4);
    s2v a = splat(-32768), b = splat(32767);
#pragma unroll
    for (int h = 0; h < 16; h += 8) {
        const int o = h ^ 8;
        s2v pn[8], px[8];
        pn[0] = px[0] = d[o];
#pragma unroll
        for (int j = 1; j < 8; j++) { pn[j] = pmin(d[o + j], pn[j - 1]); px[j] = pmax(d[o + j], px[j - 1]); }
        s2v rn = d[h + 7], rx = d[h + 7];
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            if (j < 7) { rn = pmin(rn, d[h + j]); rx = pmax(rx, d[h + j]); }
            a = pmax(a, pmin(rn, pn[j]));
            b = pmin(b, pmax(rx, px[j]));
        }
    }
    const s2v m = pmax(pmax(a, splat(t)), splat(0) - pmin(b, splat(-t)));
    const s2v e = m - splat(t);
    return as_u(pmin(e, splat(1)) * (e + splat(t - 1)));
}

/* One wave per workgroup: items differ in length with their share of candidates, and a workgroup's LDS is held until
 * its slowest wave ends. */
/* ALIGNED: every image of the set has dword-aligned rows (one load instruction per row, in straight-line code) */
template <bool ALIGNED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fast_queue(FastSet S, int threshold, int border, uint8_t* __restrict__ rowcnt, u32* __restrict__ head, u32* __restrict__ tmp) {
    __shared__ u32 img_ring[FQ_IR][64];
    __shared__ u32 sc_ring[FQ_SR][64];
    __shared__ unsigned short queue[FQ_QCAP];
    const int lane = threadIdx.x;
    const unsigned wave = blockIdx.x, nwaves = gridDim.x; /* wave-uniform */
    const unsigned total_work = S.item_begin[S.n];
    const u64 lt_mask = (1ull << lane) - 1ull;
    for (unsigned work = wave; work < total_work; work += nwaves) {
        int lv = 0;
        while (lv + 1 < S.n && work >= S.item_begin[lv + 1]) lv++;
        const FastImg img = S.img[lv];
        const int strips_x = S.strips_x[lv], band_h = S.band_h;
        const int W = img.w, H = img.h;
        const unsigned per_frame = (unsigned)(strips_x * S.bands_y[lv]);
        /* all strips and bands of a frame's level on one XCD: neighbouring strips straddle cache lines, bands share their
         * halo rows (a level's item count is a multiple of 8 whenever the batch is, so the global work number and the
         * level's own agree about the XCD; other batch sizes keep the plain order) */
        unsigned fb, t_item;
        xcd_frame_of(work - S.item_begin[lv], per_frame, (S.batch & 7) ? 0u : (unsigned)S.batch, fb, t_item);
        const int b = (int)fb;
        const int sxi = t_item % strips_x, byi = t_item / strips_x;
        const int X0 = sxi * FS_OUT + 4 * (lane - 1); /* image column of this lane's pixel 0 */
        const int y0 = byi * band_h;
        const uint8_t* ip = img.data + (size_t)b * img.fstride;
        /* with dword-aligned rows every lane loads one dword: bytes left of column 0 or right of column W - 1 feed no
         * scored pixel, and a dword that starts inside the row ends inside its pitch */
        const int X0c = min(max(X0, 0), (W - 1) & ~3);
        const bool out_lane = lane >= 1 && lane < 63;
        const size_t seg0 = (size_t)S.seg_begin[lv] + (size_t)b * H * strips_x + sxi;
        /* per pixel: may be scored (circle inside the image), may be reported */
        u32 smask[2] = {0, 0};
        bool rep[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = X0 + k;
            /* the halo lanes score too: pixel 3 of lane 0 and pixel 0 of lane 63 are neighbours in the suppression */
            if (x >= 3 && x < W - 3) smask[k >> 1] |= 0xFFFFu << (16 * (k & 1));
            rep[k] = out_lane && x >= border && x < W - border;
        }

        u32 R[FQ_PERIOD][3]; /* image rows: left neighbour's, own, right neighbour's dword; the row of step t in slot t mod FQ_PERIOD */
        /* the three newest score rows, the row fetched in step t in slot t mod 3: pixel pairs (0,1), (2,3) as 16-bit values,
         * and per pair the largest horizontal neighbour without (H2) and with (H3) the pixel itself */
        u32 SP[3][2], H2[3][2], H3[3][2];
#pragma unroll
        for (int i = 0; i < FQ_PERIOD; i++) R[i][0] = R[i][1] = R[i][2] = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) SP[i][0] = SP[i][1] = H2[i][0] = H2[i][1] = H3[i][0] = H3[i][1] = 0;
        int qn = 0, q_head = 0, first_c = 0; /* queue state (wave-uniform); first_c = centre index of the oldest entry */

        auto load_row = [&](int yy) -> u32 {
            const uint8_t* row = ip + (size_t)min(max(yy, 0), H - 1) * img.pitch;
            if (ALIGNED || img.aligned) return *reinterpret_cast<const uint32_t*>(row + X0c); /* wave-uniform */
            return (u32)row[min(max(X0, 0), W - 1)] | ((u32)row[min(max(X0 + 1, 0), W - 1)] << 8) |
                   ((u32)row[min(max(X0 + 2, 0), W - 1)] << 16) | ((u32)row[min(max(X0 + 3, 0), W - 1)] << 24);
        };
        u32 pre[FQ_AHEAD]; /* rows y0 - 4 + t .. + FQ_AHEAD - 1 on their way; the row of step t sits in slot t mod FQ_PERIOD */
#pragma unroll
        for (int i = 0; i < FQ_AHEAD; i++) pre[i] = load_row(y0 - 4 + i);

        /* Rows are counted in steps: step t loads image row y0 - 4 + t; index c names the row loaded in step c. */
        /* score up to 64 queued pairs: lane i takes entry q_head + i */
        auto drain = [&](int tc_now) {
            const int n = min(qn, 64);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const u32 e = queue[(q_head + lane) & (FQ_QCAP - 1)];
            const bool act = lane < n;
            const int pair = (int)(e & 1u), L = act ? (int)((e >> 1) & 63u) : 1, srow = (int)((e >> 7) & 15u);
            const int cslot = act ? (int)(e >> 11) : 3;
            u32 Rq[7][3];
            const u32 sh = (u32)pair << 4; /* the pair at bytes 6, 7 of its lane's window moves to bytes 4, 5 */
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const u32* p = img_ring[(cslot + i - 3) & (FQ_IR - 1)];
                const u32 wl = p[max(L - 1, 0)], wc = p[L], wr = p[min(L + 1, 63)]; /* lanes 0 and 63: their outer pixels feed nothing */
                Rq[i][0] = __builtin_amdgcn_alignbit(wc, wl, sh);
                Rq[i][1] = __builtin_amdgcn_alignbit(wr, wc, sh);
                Rq[i][2] = wr >> sh;
            }
            u32 s = fast_score_window(Rq, threshold);
            /* pixels whose circle leaves the image score 0 */
            const int x = sxi * FS_OUT + 4 * (L - 1) + 2 * pair;
            if (!(x >= 3 && x < W - 3)) s &= 0xFFFF0000u;
            if (!(x + 1 >= 3 && x + 1 < W - 3)) s &= 0x0000FFFFu;
            if (act) reinterpret_cast<unsigned short*>(&sc_ring[srow][0])[2 * L + pair] = (unsigned short)((s & 0xFFu) | ((s >> 8) & 0xFF00u));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            q_head = (q_head + n) & (FQ_QCAP - 1);
            qn -= n;
            if (qn > 0) { /* centre index of the new oldest entry, from its ring slot */
                const int hs = __builtin_amdgcn_readfirstlane((int)queue[q_head] >> 11);
                first_c = tc_now - ((tc_now - hs) & (FQ_IR - 1));
            }
        };
        /* everything older than FQ_K rows, and whole batches of 64 */
        auto drain_due = [&](int t) {
            while (qn >= 64 || (qn > 0 && first_c <= t - 3 - FQ_K)) drain(t - 3);
        };
        /* suppression and output of row index m = t - 4 - FQ_K (image row M); scores of rows <= m + 1 are final.  s3c = t mod 3. */
        auto suppress = [&](auto s3c, int t, int n_rows) {
            constexpr int SN = decltype(s3c)::value, SCN = (SN + 2) % 3, SO = (SN + 1) % 3; /* slots of rows m + 1, m, m - 1 */
            const int cn = t - 3 - FQ_K; /* newest score row needed */
            {
                u32 p01 = 0, p23 = 0;
                if (cn >= 3 && cn < n_rows) { /* rows that went through the compass test (wave-uniform) */
                    const u32 w4 = sc_ring[cn & (FQ_SR - 1)][lane];
                    p01 = __builtin_amdgcn_perm(0u, w4, 0x0c010c00u);
                    p23 = __builtin_amdgcn_perm(0u, w4, 0x0c030c02u);
                }
                const u32 e = (lane_up_u(p23) >> 16) | (lane_down_u(p01) << 16); /* s[-1] | s[4] << 16 */
                const u2v L01 = as_uh(__builtin_amdgcn_perm(p01, e, 0x05040100u));    /* (s[-1], s0) */
                const u2v M12 = as_uh(__builtin_amdgcn_perm(p23, p01, 0x05040302u));  /* (s1, s2) */
                const u2v R23 = as_uh(__builtin_amdgcn_perm(e, p23, 0x07060302u));    /* (s3, s4) */
                const u2v h01 = __builtin_elementwise_max(L01, M12), h23 = __builtin_elementwise_max(M12, R23);
                SP[SN][0] = p01; SP[SN][1] = p23;
                H2[SN][0] = __builtin_bit_cast(u32, h01); H2[SN][1] = __builtin_bit_cast(u32, h23);
                H3[SN][0] = __builtin_bit_cast(u32, __builtin_elementwise_max(h01, as_uh(p01)));
                H3[SN][1] = __builtin_bit_cast(u32, __builtin_elementwise_max(h23, as_uh(p23)));
            }
            const int m = cn - 1, M = y0 - 4 + m;
            if (m < 4 || M >= H || M >= y0 + band_h) return; /* wave-uniform */
            int total = 0;
            if (__ballot((SP[SCN][0] | SP[SCN][1]) != 0) != 0 && M >= border && M < H - border) {
                /* largest of the eight neighbours */
                const u2v nb01 = __builtin_elementwise_max(__builtin_elementwise_max(as_uh(H3[SO][0]), as_uh(H2[SCN][0])), as_uh(H3[SN][0]));
                const u2v nb23 = __builtin_elementwise_max(__builtin_elementwise_max(as_uh(H3[SO][1]), as_uh(H2[SCN][1])), as_uh(H3[SN][1]));
                const u2v c01 = as_uh(SP[SCN][0]), c23 = as_uh(SP[SCN][1]);
                const bool kp[4] = {rep[0] && c01.x > nb01.x, rep[1] && c01.y > nb01.y, rep[2] && c23.x > nb23.x, rep[3] && c23.y > nb23.y};
                const unsigned long long m0 = __ballot(kp[0]), m1 = __ballot(kp[1]), m2 = __ballot(kp[2]), m3 = __ballot(kp[3]);
                total = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
                if (total) {
                    /* keypoints of lower lanes come first, then this lane's in pixel order */
                    u32 pos = 0;
                    pos = __builtin_amdgcn_mbcnt_hi((u32)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m0, pos));
                    pos = __builtin_amdgcn_mbcnt_hi((u32)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m1, pos));
                    pos = __builtin_amdgcn_mbcnt_hi((u32)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m2, pos));
                    pos = __builtin_amdgcn_mbcnt_hi((u32)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m3, pos));
                    /* the first FS_HEAD keypoints of a segment go to a compact array (16 bytes per segment), the rest to the
                     * segment's full-size slot: with one keypoint per segment on average, a 512-byte slot per segment
                     * made the scatter pass fetch a cache line per keypoint */
                    const size_t sg = seg0 + (size_t)M * strips_x;
                    u32* dst_h = head + sg * FS_HEAD;
                    u32* dst = tmp + sg * FS_ROWCAP;
                    const u32 sc[4] = {SP[SCN][0] & 0xFFFFu, SP[SCN][0] >> 16, SP[SCN][1] & 0xFFFFu, SP[SCN][1] >> 16};
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (kp[k]) { (pos < FS_HEAD ? dst_h : dst)[pos] = (u32)(4 * (lane - 1) + k) | (sc[k] << 8); pos++; }
                }
            }
            if (lane == 0) rowcnt[seg0 + (size_t)M * strips_x] = (uint8_t)total;
        };

        const int T = band_h + 8; /* load steps; rows with a compass test have indices 3 .. T - 4 */
        /* stc: 0 = load only, 1 = also the compass test of row index t - 3, 2 = also drain what is due and suppress */
        auto step = [&](auto phc, auto stc, int t) {
            constexpr int PH = decltype(phc)::value, ST = decltype(stc)::value;
            const int yy = y0 - 4 + t;
            {
                /* The row moves from its prefetch register into the ring by an instruction of its own, and only then is the
                 * load for nine rows on issued: into the same prefetch register.  (Left to the compiler the row stays
                 * where it was loaded, the new load gets a fresh register, and at the end of every turn of the loop nine
                 * copies put the prefetch registers back -- behind a wait for all nine loads in flight.) */
                u32 c;
                asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(pre[PH]));
                __builtin_amdgcn_sched_barrier(0);
                pre[PH] = load_row(yy + FQ_AHEAD); /* clamped: always a valid address */
                __builtin_amdgcn_sched_barrier(0);
                R[PH][1] = c;
                R[PH][0] = lane_up_u(c);
                R[PH][2] = lane_down_u(c);
                img_ring[t & (FQ_IR - 1)][lane] = c;
            }
            if constexpr (ST < 1) return;
            const int tc = t - 3, N = yy - 3;
            sc_ring[tc & (FQ_SR - 1)][lane] = 0u;
            if (N >= 3 && N < H - 3) { /* wave-uniform */
                const bool p0 = fast_compass_pair<0, PH>(R, threshold, smask[0]);
                const bool p1 = fast_compass_pair<2, PH>(R, threshold, smask[1]);
                const u64 m0 = __ballot(p0), m1 = __ballot(p1);
                const int n0 = __popcll(m0), n1 = __popcll(m1);
                if (n0 + n1) {
                    const u32 base = ((u32)(tc & (FQ_IR - 1)) << 11) | ((u32)(tc & (FQ_SR - 1)) << 7) | ((u32)lane << 1);
                    const int tail = q_head + qn;
                    if (p0) queue[(tail + __popcll(m0 & lt_mask)) & (FQ_QCAP - 1)] = (unsigned short)base;
                    if (p1) queue[(tail + n0 + __popcll(m1 & lt_mask)) & (FQ_QCAP - 1)] = (unsigned short)(base | 1u);
                    if (qn == 0) first_c = tc;
                    qn += n0 + n1;
                }
            }
            if constexpr (ST < 2) return;
            drain_due(t);
            suppress(ic<PH % 3>{}, t, T - 3);
        };

        step(ic<0>{}, ic<0>{}, 0); step(ic<1>{}, ic<0>{}, 1); step(ic<2>{}, ic<0>{}, 2); step(ic<3>{}, ic<0>{}, 3);
        step(ic<4>{}, ic<0>{}, 4); step(ic<5>{}, ic<0>{}, 5); step(ic<6>{}, ic<2>{}, 6); step(ic<7>{}, ic<2>{}, 7); step(ic<8>{}, ic<2>{}, 8);
        int t0 = FQ_PERIOD;
        for (; t0 < T; t0 += FQ_PERIOD) {
            if (y0 - 7 + t0 >= H - 3) break; /* no row left that could be scored (wave-uniform) */
            step(ic<0>{}, ic<2>{}, t0); step(ic<1>{}, ic<2>{}, t0 + 1); step(ic<2>{}, ic<2>{}, t0 + 2);
            step(ic<3>{}, ic<2>{}, t0 + 3); step(ic<4>{}, ic<2>{}, t0 + 4); step(ic<5>{}, ic<2>{}, t0 + 5);
            step(ic<6>{}, ic<2>{}, t0 + 6); step(ic<7>{}, ic<2>{}, t0 + 7); step(ic<8>{}, ic<2>{}, t0 + 8);
        }
        /* the rows still on their way: no more loads, only what is queued and the suppression behind it.  Score rows of
         * indices >= t0 - 3 never had a compass test: suppress() reads them as 0.  t0 is a multiple of three. */
        const int t_end = min(band_h, H - y0) + 8 + FQ_K;
        int t = t0;
        for (; t + 2 < t_end; t += 3) {
            drain_due(t); suppress(ic<0>{}, t, t0 - 3);
            drain_due(t + 1); suppress(ic<1>{}, t + 1, t0 - 3);
            drain_due(t + 2); suppress(ic<2>{}, t + 2, t0 - 3);
        }
        if (t < t_end) { drain_due(t); suppress(ic<0>{}, t, t0 - 3); }
        if (t + 1 < t_end) { drain_due(t + 1); suppress(ic<1>{}, t + 1, t0 - 3); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* the next item reuses the rings */
    }
}

/* one thread per (frame, row, strip) segment copies the segment's keypoints to their final offset */
__global__ __launch_bounds__(256) void k_fast_scatter_stream(FastSet S, const uint8_t* __restrict__ rowcnt, const u32* __restrict__ rowoff,
                                                             const u32* __restrict__ head, const u32* __restrict__ tmp) {
    const unsigned seg = blockIdx.x * 256u + threadIdx.x;
    if (seg >= S.seg_begin[S.n]) return;
    const int n = rowcnt[seg];
    if (n == 0) return;
    int lv = 0;
    while (lv + 1 < S.n && seg >= S.seg_begin[lv + 1]) lv++;
    const int H = S.img[lv].h, strips_x = S.strips_x[lv], cap = S.cap[lv];
    const unsigned per_frame = (unsigned)(H * strips_x), ls = seg - S.seg_begin[lv];
    const int b = ls / per_frame;
    const unsigned r = ls - (unsigned)b * per_frame;
    const int y = r / strips_x, sxi = r - y * strips_x;
    const u32 off0 = rowoff[seg];
    int* kps = S.kps[lv];
    for (int j = 0; j < n; j++) {
        const u32 e = j < FS_HEAD ? head[(size_t)seg * FS_HEAD + j] : tmp[(size_t)seg * FS_ROWCAP + j];
        const u32 off = off0 + j;
        if (off < (u32)cap) {
            int* o = kps + ((size_t)b * cap + off) * 3;
            o[0] = sxi * FS_OUT + (int)(e & 255u); o[1] = y; o[2] = (int)(e >> 8);
        }
    }
}

/* exclusive prefix of rowcnt over the (row, strip) segments of one frame of one level, in raster order;
 * one workgroup per (frame, level) */
__global__ __launch_bounds__(1024) void k_fast_scan_stream(FastSet S, const uint8_t* __restrict__ rowcnt, u32* __restrict__ rowoff) {
    __shared__ u32 s_buf[2][1024];
    const int b = blockIdx.x, lv = blockIdx.y, tid = threadIdx.x;
    const int n_per_frame = S.img[lv].h * S.strips_x[lv];
    const uint8_t* rc = rowcnt + S.seg_begin[lv] + (size_t)b * n_per_frame;
    u32* ro = rowoff + S.seg_begin[lv] + (size_t)b * n_per_frame;
    const int per = (n_per_frame + 1023) / 1024;
    const int i0 = min(tid * per, n_per_frame), i1 = min(i0 + per, n_per_frame);
    u32 sum = 0;
    for (int i = i0; i < i1; i++) sum += rc[i];
    int cur = 0;
    s_buf[0][tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        u32 x = s_buf[cur][tid];
        if (tid >= o) x += s_buf[cur][tid - o];
        s_buf[cur ^ 1][tid] = x;
        cur ^= 1;
        __syncthreads();
    }
    u32 run = s_buf[cur][tid] - sum;
    for (int i = i0; i < i1; i++) { ro[i] = run; run += rc[i]; }
    if (tid == 1023) S.counts[lv][b] = (int)min(s_buf[cur][1023], (u32)S.cap[lv]);
}

static int fast_set_geometry(FastSet& S, int batch) {
    /* band height 9k - 8 (whole turns of the unrolled row loop); the tallest that still gives every SIMD several items */
    static const int heights[5] = {64, 37, 28, 19, 10};
    int band_h = heights[4]; /* small sets: the shortest band that is still a whole number of turns */
    for (int i = 0; i < 5; i++) {
        long long items = 0;
        for (int l = 0; l < S.n; l++) items += (long long)div_up(S.img[l].w, FS_OUT) * div_up(S.img[l].h, heights[i]) * batch;
        if (items >= 12288) { band_h = heights[i]; break; }
    }
    if ((band_h + 8) % FQ_PERIOD != 0) { vfe_set_error("FAST: band height %d is not a whole number of row-loop turns", band_h); return VFE_E_INVALID; }
    S.band_h = band_h;
    S.batch = batch;
    unsigned long long items = 0, segs = 0;
    for (int l = 0; l < S.n; l++) {
        S.strips_x[l] = div_up(S.img[l].w, FS_OUT);
        S.bands_y[l] = div_up(S.img[l].h, band_h);
        S.item_begin[l] = (unsigned)items; S.seg_begin[l] = (unsigned)segs;
        items += (unsigned long long)S.strips_x[l] * S.bands_y[l] * batch;
        segs += (unsigned long long)S.strips_x[l] * S.img[l].h * batch;
    }
    if (items >= (1ull << 31) || segs >= (1ull << 31)) { vfe_set_error("FAST: image set too large"); return VFE_E_INVALID; }
    S.item_begin[S.n] = (unsigned)items; S.seg_begin[S.n] = (unsigned)segs;
    return VFE_OK;
}

FastSetWs fast_set_layout(int n, const int* w, const int* h, int batch) {
    FastSetWs L;
    size_t segs = 0;
    for (int l = 0; l < n; l++) segs += (size_t)div_up(w[l], FS_OUT) * h[l] * batch;
    size_t o = 0;
    L.rowcnt_off = o; o += align_up(segs, 256);
    L.rowoff_off = o; o += align_up(segs * sizeof(u32), 256);
    L.head_off = o; o += align_up(segs * FS_HEAD * sizeof(u32), 256);
    L.tmp_off = o; o += align_up(segs * FS_ROWCAP * sizeof(u32), 256);
    L.total = o;
    return L;
}

int fast_detect_set(FastSet& S, int batch, int threshold, int border, char* ws, hipStream_t st) {
    if (S.n < 1 || S.n > FAST_MAX_LEVELS) { vfe_set_error("FAST: 1..%d images per set", FAST_MAX_LEVELS); return VFE_E_INVALID; }
    int w[FAST_MAX_LEVELS], h[FAST_MAX_LEVELS];
    for (int l = 0; l < S.n; l++) {
        w[l] = S.img[l].w; h[l] = S.img[l].h;
        if (w[l] < 16 || h[l] < 16) { vfe_set_error("FAST (streaming): image must be at least 16x16"); return VFE_E_INVALID; }
    }
    int rc = fast_set_geometry(S, batch);
    if (rc) return rc;
    const FastSetWs L = fast_set_layout(S.n, w, h, batch);
    uint8_t* rowcnt = (uint8_t*)(ws + L.rowcnt_off);
    u32* rowoff = (u32*)(ws + L.rowoff_off);
    u32* tmp = (u32*)(ws + L.tmp_off);
    u32* head = (u32*)(ws + L.head_off);
    const unsigned total = S.item_begin[S.n], n_seg = S.seg_begin[S.n];
    bool all_aligned = true;
    for (int l = 0; l < S.n; l++) all_aligned = all_aligned && S.img[l].aligned;
    if (all_aligned) hipLaunchKernelGGL(k_fast_queue<true>, dim3(total < (1u << 22) ? total : (1u << 22)), dim3(64), 0, st, S, threshold, border, rowcnt, head, tmp);
    else hipLaunchKernelGGL(k_fast_queue<false>, dim3(total < (1u << 22) ? total : (1u << 22)), dim3(64), 0, st, S, threshold, border, rowcnt, head, tmp);
    VFE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fast_scan_stream, dim3(batch, S.n), dim3(1024), 0, st, S, rowcnt, rowoff);
    VFE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fast_scatter_stream, dim3((n_seg + 255) / 256), dim3(256), 0, st, S, rowcnt, rowoff, head, tmp);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

FastImg fast_img(const DImg& d) {
    FastImg f;
    f.data = d.data; f.w = d.w; f.h = d.h; f.pitch = d.pitch; f.fstride = d.fstride;
    f.aligned = ((uintptr_t)d.data % 4 == 0) && (d.pitch % 4 == 0) && (d.fstride % 4 == 0);
    return f;
}
