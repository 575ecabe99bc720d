/* cv::pyrDown for 16-byte aligned images, streaming (no LDS, no barriers).
 *
 * One wavefront owns a strip of 62 x 8 output columns (lanes 0 and 63 are halo) and walks down a band
 * of output rows.  Per source row each lane issues one coalesced 16-byte load (1 KiB per wave),
 * requested six rows ahead, and takes the three halo bytes it needs from its neighbours by shuffles.
 *
 * Horizontal pass: output k reads source bytes 2k-2 .. 2k+2.  Four of the five taps are one
 * v_dot4_u32_u8 with the weights (1, 4, 6, 4) on a dword that is either a loaded dword as it is (odd
 * outputs) or two of them funnel-shifted by two bytes (even outputs); the fifth tap rides in the
 * accumulator operand.  The eight sums of a lane are packed two per dword (16 * 255 fits 16 bits).
 *
 * Vertical pass: instead of keeping five rows of sums, every row is added on arrival into the three
 * output rows it contributes to (weights 1, 4, 6, 4, 1 over rows 2m-2 .. 2m+2): one accumulator is
 * completed, one is in the middle, one is started per pair of source rows.  The accumulators rotate
 * roles with period three, so the loop is unrolled three pairs deep and nothing is ever moved.  The
 * rounding constant is the accumulator's initial value; the two result bytes of a packed pair are
 * its bytes 1 and 3, picked with one v_perm per four outputs.
 *
 * Levels that carry a reflect-101 border (vfe_dimg.border) get it in the same pass: rows near the top
 * and bottom are stored a second time at their mirrored position, and the first / last lanes of a
 * row store their bytes reversed into the left / right margin. */
#include "vfe_common.h"
#include <type_traits>

typedef unsigned int u32;
template <int N> using ic = std::integral_constant<int, N>;

#define PS_OUT_LANES 62
#define PS_AHEAD 6
#define PS_BORDER 32 /* the only border width the fused path writes */

/* Source bytes IX0 .. IX0+15 of row iy.  Chunks entirely left of the image, or entirely right of
 * it when the width is a multiple of 16, are never read: the only bytes a neighbour would take
 * from them are reflections that the neighbour already holds (fixed up after the shuffle).
 * `once`: the launch reads far more than the caches hold, so its loads carry the non-temporal hint
 * (a 4 : 1 read / write stream moves ~10 % more bytes per second that way; the stores stay plain,
 * the next level reads them back from cache). */
template <bool W16, bool once>
__device__ static inline uint4 pyr_load_row(const DImg& src, const uint8_t* sp, int iy, int IX0) {
    const uint8_t* row = sp + (size_t)reflect101(iy, src.h) * src.pitch;
    /* W16: every lane loads unconditionally, lanes outside the image a chunk inside it (never used) -- a load under a
     * lane condition merges with the zeros of the other branch, which is a copy again */
    const int IXc = W16 ? min(max(IX0, 0), src.w - 16) : IX0;
    if (W16 || (IX0 >= 0 && IX0 + 15 < src.w)) {
        if (!once) return *reinterpret_cast<const uint4*>(row + IXc);
        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
        const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(row + IXc));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    uint32_t w[4] = {0, 0, 0, 0};
    if (!W16 && IX0 + 15 >= src.w && IX0 <= src.w + 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) w[k >> 2] |= (uint32_t)row[reflect101(IX0 + k, src.w)] << (8 * (k & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

/* the eight horizontal [1 4 6 4 1] sums of a lane, packed in pairs; lw = left neighbour's last dword
 * (source bytes -4 .. -1), rb = source byte 16 */
__device__ static inline void pyr_hsums(const uint4& c, u32 lw, u32 rb, u32 (&hp)[4]) {
    const u32 WT = 0x04060401u; /* byte 0 (first source byte) x1, then x4, x6, x4 */
    const u32 d[4] = {c.x, c.y, c.z, c.w};
    u32 r[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 prev = j == 0 ? lw : d[j - 1];
        const u32 nextb = j == 3 ? rb : (d[j + 1] & 0xFFu);
        /* even output 2j: bytes 4j-2 .. 4j+1 by weight, byte 4j+2 once */
        r[2 * j] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d[j], prev, 2), WT, (d[j] >> 16) & 0xFFu, false);
        /* odd output 2j+1: bytes 4j .. 4j+3 by weight, byte 4j+4 once */
        r[2 * j + 1] = __builtin_amdgcn_udot4(d[j], WT, nextb, false);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) hp[i] = r[2 * i] | (r[2 * i + 1] << 16);
}

/* W16: the source width is a multiple of 16 (no lane straddles the right rim); ONCE: non-temporal loads.  Both are
 * template parameters so that a row's load is ONE instruction in straight-line code: a load goes out after the row it
 * replaces has been consumed and lands in that row's registers.  (Issued earlier it gets registers of its own, and the
 * compiler copies them back at the end of every trip round the loop behind an `s_waitcnt vmcnt(0)` -- all six loads in
 * flight waited for, once per six rows; a run-time choice between two load instructions defeats the wait counting too.) */
template <bool PAD, bool W16, bool ONCE>
__global__ __launch_bounds__(256) void k_pyr_down_stream(DImg src, DImg dst, int strips_x, int band_h, int bands_y, unsigned total_work) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    const unsigned per_frame = (unsigned)(strips_x * bands_y);
    const int DW = dst.w, DH = dst.h;
    for (unsigned work = wave; work < total_work; work += nwaves) {
        const int b = work / per_frame, t = work - b * per_frame;
        const int sxi = t % strips_x, byi = t / strips_x;
        const int X0 = (sxi * PS_OUT_LANES + lane - 1) * 8; /* first output column of this lane */
        const int IX0 = 2 * X0;                             /* source column of byte 0 of its 16-byte load */
        const int oy0 = byi * band_h, oy1 = min(oy0 + band_h, DH);
        const uint8_t* sp = src.frame(b);
        uint8_t* dp = dst.frame(b);
        const bool out_lane = lane >= 1 && lane <= PS_OUT_LANES && X0 < DW;
        const bool full_lane = out_lane && X0 + 7 < DW;
        const bool fix_l = (IX0 == 0), fix_r = W16 && (IX0 + 16 == src.w);
        /* lanes that also write a margin (PAD only; DW is a multiple of 8 there) */
        const bool rim_strip = PAD && (sxi == 0 || (sxi + 1) * PS_OUT_LANES * 8 + 8 > DW - 40);
        const bool padl8 = out_lane && X0 <= 24, padl1 = out_lane && X0 == 32;
        const bool padr8 = out_lane && X0 >= DW - 32, padr1 = out_lane && X0 == DW - 40;

        const int iy0 = 2 * oy0 - 2, iy1 = 2 * oy1; /* source rows iy0 .. iy1 */
        uint4 pre[PS_AHEAD];
#pragma unroll
        for (int k = 0; k < PS_AHEAD; k++) pre[k] = pyr_load_row<W16, ONCE>(src, sp, min(iy0 + k, iy1), IX0);
        u32 A[3][4]; /* accumulators of three consecutive output rows, two 16-bit sums per dword */
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) A[i][k] = 0;

        auto store_row = [&](int oy, u32 o0, u32 o1, u32 rv0, u32 rv1) {
            uint8_t* drow = dp + (ptrdiff_t)oy * dst.pitch + X0;
            if (full_lane) *reinterpret_cast<uint2*>(drow) = make_uint2(o0, o1); /* plain stores: the next level reads them back from cache (non-temporal ones cost 15 % of the three launches) */
            else if (out_lane) {
                const u32 oo[2] = {o0, o1};
                for (int k = 0; k < 8 && X0 + k < DW; k++) drow[k] = (uint8_t)(oo[k >> 2] >> (8 * (k & 3)));
            }
            if (PAD && rim_strip) {
                /* margins: column -c mirrors column c, column DW-1+c mirrors DW-1-c; the reversed eight
                 * bytes of a lane land on -(X0+7) .. -X0 and 2(DW-1)-(X0+7) .. 2(DW-1)-X0 */
                uint8_t* r0 = dp + (ptrdiff_t)oy * dst.pitch;
                if (padl8) { uint2 v = make_uint2(rv1, rv0); __builtin_memcpy(r0 - X0 - 7, &v, 8); }
                if (padl1) r0[-32] = (uint8_t)o0;
                if (padr8) { uint2 v = make_uint2(rv1, rv0); __builtin_memcpy(r0 + 2 * (DW - 1) - X0 - 7, &v, 8); }
                if (padr1) r0[DW + 31] = (uint8_t)(o1 >> 24);
            }
        };

        /* one pair of source rows 2m, 2m+1; phc rotates the accumulator roles */
        auto pair_step = [&](auto phc, int m) {
            constexpr int ph = decltype(phc)::value;
            constexpr int P = ph % 3, C = (ph + 1) % 3, N = (ph + 2) % 3; /* completed by row 2m, in the middle, started */
            u32 hp[4];
            {   /* even row 2m */
                const uint4 cur = pre[2 * ph];
                u32 lw = (u32)__shfl_up((int)cur.w, 1), rb = (u32)__shfl_down((int)cur.x, 1) & 0xFFu;
                /* at the image rim the halo bytes are reflections this lane already holds:
                 * bytes -2, -1 = bytes 2, 1; byte 16 = byte 14 */
                if (fix_l) lw = __builtin_amdgcn_perm(cur.x, cur.x, 0x01020c0cu);
                if (fix_r) rb = (cur.w >> 16) & 0xFFu;
                pyr_hsums(cur, lw, rb, hp);
                __builtin_amdgcn_sched_barrier(0);
                pre[2 * ph] = pyr_load_row<W16, ONCE>(src, sp, min(2 * m + PS_AHEAD, iy1), IX0); /* rows past the band: the last one again, unused */
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; k++) { A[P][k] += hp[k]; A[C][k] += 6u * hp[k]; A[N][k] = hp[k] + 0x00800080u; }
                const int oy = m - 1;
                if (oy >= oy0) {
                    /* (sum + 128) >> 8 of each half = bytes 1 and 3 of the accumulator dwords */
                    const u32 o0 = __builtin_amdgcn_perm(A[P][1], A[P][0], 0x07050301u), o1 = __builtin_amdgcn_perm(A[P][3], A[P][2], 0x07050301u);
                    u32 rv0 = 0, rv1 = 0;
                    if (PAD) { rv0 = __builtin_amdgcn_perm(o0, o0, 0x00010203u); rv1 = __builtin_amdgcn_perm(o1, o1, 0x00010203u); }
                    store_row(oy, o0, o1, rv0, rv1);
                    if (PAD) { /* rows above and below the frame mirror rows 1.. and DH-2.. */
                        if (oy >= 1 && oy <= PS_BORDER) store_row(-oy, o0, o1, rv0, rv1);
                        if (oy <= DH - 2 && oy >= DH - 1 - PS_BORDER) store_row(2 * (DH - 1) - oy, o0, o1, rv0, rv1);
                    }
                }
            }
            {   /* odd row 2m+1 (not needed after the last output row; harmless then) */
                const uint4 cur = pre[2 * ph + 1];
                u32 lw = (u32)__shfl_up((int)cur.w, 1), rb = (u32)__shfl_down((int)cur.x, 1) & 0xFFu;
                if (fix_l) lw = __builtin_amdgcn_perm(cur.x, cur.x, 0x01020c0cu);
                if (fix_r) rb = (cur.w >> 16) & 0xFFu;
                pyr_hsums(cur, lw, rb, hp);
                __builtin_amdgcn_sched_barrier(0);
                pre[2 * ph + 1] = pyr_load_row<W16, ONCE>(src, sp, min(2 * m + 1 + PS_AHEAD, iy1), IX0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; k++) { A[C][k] += 4u * hp[k]; A[N][k] += 4u * hp[k]; }
            }
        };

        /* pairs m = oy0-1 .. oy1: the first two only fill, the last only completes row oy1-1 */
        const int m_end = oy1 + 1;
        int m = oy0 - 1;
        for (; m + 2 < m_end; m += 3) { /* whole triples only: a conditional step would keep replaced registers alive */
            pair_step(ic<0>{}, m);
            pair_step(ic<1>{}, m + 1);
            pair_step(ic<2>{}, m + 2);
        }
        if (m < m_end) pair_step(ic<0>{}, m);
        if (m + 1 < m_end) pair_step(ic<1>{}, m + 1);
    }
}

/* geometry check: the streaming kernel needs 16-byte aligned sources and 8-byte aligned destinations */
static bool vec_ok(const DImg& d, int align) {
    return d.sub_n == 1 && ((uintptr_t)d.data % align == 0) && (d.pitch % align == 0) && (d.fstride % align == 0);
}
bool pyr_down_stream_ok(const DImg& s, const DImg& d) { return vec_ok(s, 16) && vec_ok(d, 8); }

typedef void (*pyr_kernel_t)(DImg, DImg, int, int, int, unsigned);

#define PS_ONCE_BYTES (64ll << 20) /* a source this large is gone from the caches before anything reads it again */

/* waves of `kern` that fit one SIMD (512 VGPRs each, at most 8 waves) */
static int waves_per_simd(pyr_kernel_t kern) {
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kern)) != hipSuccess || a.numRegs <= 0) return 4;
    const int regs = (a.numRegs + 7) / 8 * 8;
    const int w = 512 / regs;
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}

/* *padded = the kernel also wrote dst's border */
int pyr_down_stream_launch(const DImg& s, const DImg& d, hipStream_t st, bool* padded) {
    const bool pad = d.border == PS_BORDER && d.w % 8 == 0 && d.w >= 48 && d.h >= PS_BORDER + 2;
    const bool w16 = s.w % 16 == 0;
    const bool once = (long long)s.w * s.h * s.batch >= PS_ONCE_BYTES;
    static const pyr_kernel_t kerns[8] = {k_pyr_down_stream<false, false, false>, k_pyr_down_stream<false, false, true>,
                                          k_pyr_down_stream<false, true, false>,  k_pyr_down_stream<false, true, true>,
                                          k_pyr_down_stream<true, false, false>,  k_pyr_down_stream<true, false, true>,
                                          k_pyr_down_stream<true, true, false>,   k_pyr_down_stream<true, true, true>};
    const int ki = (pad ? 4 : 0) + (w16 ? 2 : 0) + (once ? 1 : 0);
    const pyr_kernel_t kern = kerns[ki];
    static int occ[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int& o = occ[ki];
    if (o == 0) o = waves_per_simd(kern);
    /* Band height: every wave is an item (strip x band x frame) and all waves of a round run side by
     * side, so the launch takes ceil(items / resident waves) rounds of (2 band + 9) source rows (two
     * output rows of halo plus the prefetch ramp).  Pick the band height with the cheapest total. */
    const int strips_x = div_up(div_up(d.w, 8), PS_OUT_LANES);
    const long long slots = 256LL * 4 * o;
    int band_h = 8;
    long long best = -1;
    for (int bh = 8; bh <= 128; bh += 2) {
        const long long items = (long long)strips_x * div_up(d.h, bh) * d.batch;
        const long long cost = ((items + slots - 1) / slots) * (2 * bh + 9);
        if (best < 0 || cost < best) { best = cost; band_h = bh; }
    }
    const int bands_y = div_up(d.h, band_h);
    const unsigned total = (unsigned)(strips_x * bands_y * d.batch), nblk = (total + 3) / 4;
    const dim3 grid(nblk < (1u << 20) ? nblk : (1u << 20));
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, s, d, strips_x, band_h, bands_y, total);
    VFE_LAUNCH_CHECK();
    *padded = pad;
    return VFE_OK;
}
