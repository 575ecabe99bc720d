/* pyrDown / LK pyramid construction for gfx950.
 *
 * HBM-bound stencil: each 256-thread workgroup produces a 128x16 tile of the half-resolution
 * level.  The 35 x 264-byte source tile is read once from HBM with coalesced dword loads
 * (reflect-101 resolved per byte only on the image rim), staged in LDS, filtered
 * horizontally into 16-bit row sums in LDS and then vertically; both passes use the
 * [1 4 6 4 1] taps in integers with a single rounding, (sum + 128) >> 8, exactly like
 * cv::pyrDown.  Algorithmic traffic per output pixel: 4 source bytes + 1 written byte.
 */
#include "vfe_common.h"

#define PD_TW 128
#define PD_TH 16
#define PD_IN_ROWS (2 * PD_TH + 3)
#define PD_IN_DW (2 * PD_TW / 4 + 2)

template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_pyr_down(DImg src, DImg dst, int tiles_x, unsigned tiles_per_frame, unsigned total_work) {
    __shared__ __align__(16) uint32_t s_in[PD_IN_ROWS][PD_IN_DW];
    __shared__ __align__(16) uint16_t s_h[PD_IN_ROWS][PD_TW];
    /* persistent workgroups striding over (frame, tile) items: dispatching one workgroup per
     * tile costs more than the tile's work */
    for (unsigned work = blockIdx.x; work < total_work; work += gridDim.x) {
    const int b = work / tiles_per_frame, tile = work - b * tiles_per_frame;
    const uint8_t* sp = src.frame(b);
    uint8_t* dp = dst.frame(b);
    const int ox = (tile % tiles_x) * PD_TW, oy = (tile / tiles_x) * PD_TH;
    const int gx0 = 2 * ox - 4, gy0 = 2 * oy - 2;

    for (int i = threadIdx.x; i < PD_IN_ROWS * PD_IN_DW; i += 256) {
        const int r = i / PD_IN_DW, k = i - r * PD_IN_DW;
        const int sy = reflect101(gy0 + r, src.h);
        const int c = gx0 + 4 * k;
        const uint8_t* row = sp + (size_t)sy * src.pitch;
        uint32_t v;
        if (ALIGNED && c >= 0 && c + 3 < src.w) {
            v = *reinterpret_cast<const uint32_t*>(row + c);
        } else if (c - 4 >= 2 * src.w) {
            v = 0; /* columns no output of this tile can reach */
        } else {
            v = (uint32_t)row[reflect101(c, src.w)] | ((uint32_t)row[reflect101(c + 1, src.w)] << 8) |
                ((uint32_t)row[reflect101(c + 2, src.w)] << 16) | ((uint32_t)row[reflect101(c + 3, src.w)] << 24);
        }
        s_in[r][k] = v;
    }
    __syncthreads();

    /* horizontal pass: 4 outputs per item from 16 staged bytes; tile byte index of output x,
     * tap k is 2x + 2 + k */
    for (int i = threadIdx.x; i < PD_IN_ROWS * (PD_TW / 4); i += 256) {
        const int r = i / (PD_TW / 4), j = i % (PD_TW / 4);
        const uint2 lo = *reinterpret_cast<const uint2*>(&s_in[r][2 * j]);
        const uint2 hi = *reinterpret_cast<const uint2*>(&s_in[r][2 * j + 2]);
        const uint32_t b2 = (lo.x >> 16) & 255, b3 = lo.x >> 24;
        const uint32_t b4 = lo.y & 255, b5 = (lo.y >> 8) & 255, b6 = (lo.y >> 16) & 255, b7 = lo.y >> 24;
        const uint32_t b8 = hi.x & 255, b9 = (hi.x >> 8) & 255, b10 = (hi.x >> 16) & 255, b11 = hi.x >> 24;
        const uint32_t b12 = hi.y & 255;
        const uint32_t o0 = b2 + 4 * b3 + 6 * b4 + 4 * b5 + b6;
        const uint32_t o1 = b4 + 4 * b5 + 6 * b6 + 4 * b7 + b8;
        const uint32_t o2 = b6 + 4 * b7 + 6 * b8 + 4 * b9 + b10;
        const uint32_t o3 = b8 + 4 * b9 + 6 * b10 + 4 * b11 + b12;
        *reinterpret_cast<uint2*>(&s_h[r][4 * j]) = make_uint2(o0 | (o1 << 16), o2 | (o3 << 16));
    }
    __syncthreads();

    /* vertical pass on packed pairs of 16-bit sums: 16*4080 + 128 < 65536, so the two halves
     * of a dword never carry into each other */
    for (int i = threadIdx.x; i < PD_TH * (PD_TW / 4); i += 256) {
        const int y = i / (PD_TW / 4), j = i % (PD_TW / 4);
        const int dy = oy + y, dx = ox + 4 * j;
        if (dy >= dst.h || dx >= dst.w) continue;
        const uint2 r0 = *reinterpret_cast<const uint2*>(&s_h[2 * y][4 * j]);
        const uint2 r1 = *reinterpret_cast<const uint2*>(&s_h[2 * y + 1][4 * j]);
        const uint2 r2 = *reinterpret_cast<const uint2*>(&s_h[2 * y + 2][4 * j]);
        const uint2 r3 = *reinterpret_cast<const uint2*>(&s_h[2 * y + 3][4 * j]);
        const uint2 r4 = *reinterpret_cast<const uint2*>(&s_h[2 * y + 4][4 * j]);
        const uint32_t t0 = r0.x + 4 * r1.x + 6 * r2.x + 4 * r3.x + r4.x + 0x00800080u;
        const uint32_t t1 = r0.y + 4 * r1.y + 6 * r2.y + 4 * r3.y + r4.y + 0x00800080u;
        const uint32_t p0 = (t0 >> 8) & 255, p1 = t0 >> 24, p2 = (t1 >> 8) & 255, p3 = t1 >> 24;
        uint8_t* drow = dp + (size_t)dy * dst.pitch + dx;
        if (ALIGNED && dx + 3 < dst.w) {
            *reinterpret_cast<uint32_t*>(drow) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
        } else {
            drow[0] = (uint8_t)p0;
            if (dx + 1 < dst.w) drow[1] = (uint8_t)p1;
            if (dx + 2 < dst.w) drow[2] = (uint8_t)p2;
            if (dx + 3 < dst.w) drow[3] = (uint8_t)p3;
        }
    }
    __syncthreads();
    }
}

/* ---- streaming variant (16-byte aligned images) -------------------------------------------------
 * One wavefront owns a strip of 62 x 8 output columns and walks down a band of output rows.  Per
 * source row each lane issues one coalesced 16-byte load (1 KiB per wave), takes the three halo
 * bytes it needs from its neighbours by shuffles, forms 8 horizontal [1 4 6 4 1] sums as packed
 * pairs of 16-bit values and keeps the last five rows of sums in registers; every second source row
 * it emits one output row (8 bytes per lane).  Source rows are requested several rows ahead so
 * that HBM latency overlaps the arithmetic.  No LDS, no barriers. */
#define PS_OUT_LANES 62
#define PS_AHEAD 4

/* Source bytes IX0 .. IX0+15 of row iy.  Chunks entirely left of the image, or entirely right of
 * it when the width is a multiple of 16, are never read: the only bytes a neighbour would take
 * from them are reflections that the neighbour already holds (fixed up after the shuffle). */
__device__ static inline uint4 pyr_load_row(const DImg& src, const uint8_t* sp, int iy, int IX0, bool w16) {
    const uint8_t* row = sp + (size_t)reflect101(iy, src.h) * src.pitch;
    if (IX0 >= 0 && IX0 + 15 < src.w) return *reinterpret_cast<const uint4*>(row + IX0);
    uint32_t w[4] = {0, 0, 0, 0};
    if (!w16 && IX0 + 15 >= src.w && IX0 <= src.w + 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) w[k >> 2] |= (uint32_t)row[reflect101(IX0 + k, src.w)] << (8 * (k & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void k_pyr_down2(DImg src, DImg dst, int strips_x, int band_h, int bands_y, unsigned total_work) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    const unsigned per_frame = (unsigned)(strips_x * bands_y);
    for (unsigned work = wave; work < total_work; work += nwaves) {
        const int b = work / per_frame, t = work - b * per_frame;
        const int sxi = t % strips_x, byi = t / strips_x;
        const int X0 = (sxi * PS_OUT_LANES + lane - 1) * 8; /* first output column of this lane */
        const int IX0 = 2 * X0;                             /* source column of byte 0 of its 16-byte load */
        const int oy0 = byi * band_h, oy1 = min(oy0 + band_h, dst.h);
        const uint8_t* sp = src.frame(b);
        uint8_t* dp = dst.frame(b);
        const bool out_lane = lane >= 1 && lane <= PS_OUT_LANES && X0 < dst.w;
        const int iy0 = 2 * oy0 - 2, iy1 = 2 * (oy1 - 1) + 2;
        const bool w16 = (src.w & 15) == 0;
        const bool fix_l = (IX0 == 0), fix_r = w16 && (IX0 + 16 == src.w);

        uint4 pre[PS_AHEAD];
#pragma unroll
        for (int k = 0; k < PS_AHEAD; k++) pre[k] = pyr_load_row(src, sp, iy0 + k, IX0, w16);
        uint32_t h[5][4]; /* five rows of 8 sums, two 16-bit sums per dword */
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) h[i][k] = 0;

        for (int iy = iy0; iy <= iy1; iy++) {
            const uint4 cur = pre[0];
#pragma unroll
            for (int k = 0; k < PS_AHEAD - 1; k++) pre[k] = pre[k + 1];
            if (iy + PS_AHEAD <= iy1) pre[PS_AHEAD - 1] = pyr_load_row(src, sp, iy + PS_AHEAD, IX0, w16);
            /* halo bytes: source columns IX0-2, IX0-1 (left lane's bytes 14, 15) and IX0+16 (right lane's byte 0) */
            const uint32_t lw = (uint32_t)__shfl_up((int)cur.w, 1), rx = (uint32_t)__shfl_down((int)cur.x, 1);
            const uint32_t d[4] = {cur.x, cur.y, cur.z, cur.w};
            uint32_t E[4], O[4]; /* even / odd source bytes as packed pairs: E[i] = in[4i] | in[4i+2] << 16 */
#pragma unroll
            for (int i = 0; i < 4; i++) { E[i] = d[i] & 0x00FF00FFu; O[i] = (d[i] >> 8) & 0x00FF00FFu; }
            /* in[-2], in[-1], in[16]; at the image rim they are reflections this lane already holds */
            const uint32_t El = fix_l ? ((cur.x >> 16) & 0xFFu) : ((lw >> 16) & 0xFFu);
            const uint32_t Ol = fix_l ? ((cur.x >> 8) & 0xFFu) : (lw >> 24);
            const uint32_t Er = fix_r ? ((cur.w >> 16) & 0xFFu) : (rx & 0xFFu);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (i < 4) {
#pragma unroll
                    for (int k = 0; k < 4; k++) h[i][k] = h[i + 1][k];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                /* outputs 2i, 2i+1 of this lane: sum_k = E[k-1] + 6 E[k] + E[k+1] + 4 (O[k-1] + O[k]) with E, O indexed by output */
                const uint32_t Ep = (i == 0) ? El : (E[i - 1] >> 16);
                const uint32_t Em1 = (E[i] << 16) | Ep;                                   /* E[k-1] pair */
                const uint32_t En = (i == 3) ? Er : (E[i + 1] & 0xFFFFu);
                const uint32_t Ep1 = (E[i] >> 16) | (En << 16);                           /* E[k+1] pair */
                const uint32_t Op = (i == 0) ? Ol : (O[i - 1] >> 16);
                const uint32_t Om1 = (O[i] << 16) | Op;                                   /* O[k-1] pair */
                h[4][i] = Em1 + 6u * E[i] + Ep1 + 4u * (Om1 + O[i]);
            }
            const int done = iy - iy0;
            if (done >= 4 && ((done & 1) == 0)) {
                const int oy = (iy - 2) >> 1;
                uint32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t s = h[0][k] + 4u * h[1][k] + 6u * h[2][k] + 4u * h[3][k] + h[4][k] + 0x00800080u;
                    v[k] = (s >> 8) & 0x00FF00FFu; /* two result bytes in bits 0..7 and 16..23 */
                }
                const uint32_t o0 = (v[0] & 0xFFu) | ((v[0] >> 8) & 0xFF00u) | ((v[1] & 0xFFu) << 16) | ((v[1] & 0xFF0000u) << 8);
                const uint32_t o1 = (v[2] & 0xFFu) | ((v[2] >> 8) & 0xFF00u) | ((v[3] & 0xFFu) << 16) | ((v[3] & 0xFF0000u) << 8);
                if (out_lane) {
                    uint8_t* drow = dp + (size_t)oy * dst.pitch + X0;
                    if (X0 + 7 < dst.w) {
                        *reinterpret_cast<uint2*>(drow) = make_uint2(o0, o1);
                    } else {
                        const uint32_t oo[2] = {o0, o1};
                        for (int k = 0; k < 8 && X0 + k < dst.w; k++) drow[k] = (uint8_t)(oo[k >> 2] >> (8 * (k & 3)));
                    }
                }
            }
        }
    }
}

static bool dword_ok(const DImg& d) {
    return ((uintptr_t)d.data % 4 == 0) && (d.pitch % 4 == 0) && (d.fstride % 4 == 0);
}
static bool vec_ok(const DImg& d, int align) {
    return d.sub_n == 1 && ((uintptr_t)d.data % align == 0) && (d.pitch % align == 0) && (d.fstride % align == 0);
}

static int launch_pyr_down(const DImg& s, const DImg& d, hipStream_t st) {
    if (vec_ok(s, 16) && vec_ok(d, 8)) {
        const int strips_x = div_up(div_up(d.w, 8), PS_OUT_LANES);
        int band_h = 64;
        while (band_h > 8 && (long long)strips_x * div_up(d.h, band_h) * d.batch < 8192) band_h /= 2;
        const int bands_y = div_up(d.h, band_h);
        const unsigned total = (unsigned)(strips_x * bands_y * d.batch), nblk = (total + 3) / 4;
        ProfScope prof(VFE_K_PYR_DOWN, st);
        hipLaunchKernelGGL(k_pyr_down2, dim3(nblk < 4096u ? nblk : 4096u), dim3(256), 0, st, s, d, strips_x, band_h, bands_y, total);
        VFE_LAUNCH_CHECK();
        return VFE_OK;
    }
    const int tiles_x = div_up(d.w, PD_TW), tiles_y = div_up(d.h, PD_TH);
    const unsigned tpf = (unsigned)(tiles_x * tiles_y), total = tpf * (unsigned)d.batch;
    dim3 grid(total < 2048u ? total : 2048u);
    ProfScope prof(VFE_K_PYR_DOWN, st);
    if (dword_ok(s) && dword_ok(d))
        hipLaunchKernelGGL(k_pyr_down<true>, grid, dim3(256), 0, st, s, d, tiles_x, tpf, total);
    else
        hipLaunchKernelGGL(k_pyr_down<false>, grid, dim3(256), 0, st, s, d, tiles_x, tpf, total);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

/* ---- reflect-101 border of a level ----------------------------------------------------------------
 * Fills the `border` pixels around every frame from the frame's interior, four bytes per thread:
 * first the left and right margins of the interior rows, then the full-width rows above and below
 * (corners included; every byte is fetched from the interior, so the two parts are independent). */
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_pad_border(DImg d, unsigned side_items, unsigned per_frame, unsigned total) {
    const int B = d.border, W = d.w, H = d.h;
    const int side_g = B / 2;            /* groups of 4 bytes per interior row: 2B / 4 */
    const int row_g = (W + 2 * B + 3) / 4; /* groups per full-width border row */
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int b = i / per_frame;
        unsigned t = i - (unsigned)b * per_frame;
        int x, y;
        if (t < side_items) {
            y = (int)(t / (unsigned)side_g);
            const int k = 4 * (int)(t - (unsigned)y * (unsigned)side_g);
            x = k < B ? k - B : W + (k - B);
        } else {
            t -= side_items;
            const int ry = (int)(t / (unsigned)row_g);
            x = 4 * (int)(t - (unsigned)ry * (unsigned)row_g) - B;
            y = ry < B ? ry - B : H + (ry - B);
        }
        uint8_t* fp = d.frame(b);
        const uint8_t* srow = fp + (ptrdiff_t)reflect101(y, H) * d.pitch;
        uint32_t v;
        if (ALIGNED && x >= 0 && x + 3 < W) v = *reinterpret_cast<const uint32_t*>(srow + x);
        else v = (uint32_t)srow[reflect101(x, W)] | ((uint32_t)srow[reflect101(x + 1, W)] << 8) |
                 ((uint32_t)srow[reflect101(x + 2, W)] << 16) | ((uint32_t)srow[reflect101(x + 3, W)] << 24);
        uint8_t* dp = fp + (ptrdiff_t)y * d.pitch + x;
        if (ALIGNED) *reinterpret_cast<uint32_t*>(dp) = v;
        else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (x + j < W + B) dp[j] = (uint8_t)(v >> (8 * j));
        }
    }
}

static int launch_pad_border(const DImg& d, hipStream_t st) {
    if (d.border == 0) return VFE_OK;
    VFE_REQUIRE(d.border % 4 == 0, "border must be a multiple of 4");
    const unsigned side_items = (unsigned)d.h * (unsigned)(d.border / 2);
    const unsigned per_frame = side_items + (unsigned)(2 * d.border) * (unsigned)((d.w + 2 * d.border + 3) / 4);
    const unsigned total = per_frame * (unsigned)d.batch;
    const unsigned nblk = (total + 255u) / 256u;
    const bool aligned = d.w % 4 == 0 && (uintptr_t)d.data % 4 == 0 && d.pitch % 4 == 0 && d.fstride % 4 == 0;
    if (aligned) hipLaunchKernelGGL(k_pad_border<true>, dim3(nblk < 8192u ? nblk : 8192u), dim3(256), 0, st, d, side_items, per_frame, total);
    else hipLaunchKernelGGL(k_pad_border<false>, dim3(nblk < 8192u ? nblk : 8192u), dim3(256), 0, st, d, side_items, per_frame, total);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_pyr_down(const vfe_dimg* src, const vfe_dimg* dst, void* stream) {
    int rc;
    if ((rc = vfe_check_dimg(src, "src")) || (rc = vfe_check_dimg(dst, "dst"))) return rc;
    VFE_REQUIRE(dst->width == (src->width + 1) / 2 && dst->height == (src->height + 1) / 2, "dst must be ((w+1)/2, (h+1)/2)");
    VFE_REQUIRE(dst->batch == src->batch, "batch mismatch");
    VFE_REQUIRE(src->batch <= 65535, "batch too large");
    if ((rc = launch_pyr_down(to_dimg(src), to_dimg(dst), (hipStream_t)stream))) return rc;
    return launch_pad_border(to_dimg(dst), (hipStream_t)stream);
}

extern "C" int vfe_dev_build_pyramid(const vfe_dimg* levels, int nlevels, void* stream) {
    VFE_REQUIRE(levels && nlevels >= 1 && nlevels <= VFE_MAX_LEVELS, "nlevels out of range");
    for (int l = 1; l < nlevels; l++) {
        int rc = vfe_dev_pyr_down(&levels[l - 1], &levels[l], stream);
        if (rc) return rc;
    }
    return VFE_OK;
}

extern "C" int vfe_lk_max_level(int w, int h, int win, int max_level) {
    int level = 0;
    for (; level < max_level; level++) {
        int nw = (w + 1) / 2, nh = (h + 1) / 2;
        if (nw <= win || nh <= win) break;
        w = nw; h = nh;
    }
    return level;
}
