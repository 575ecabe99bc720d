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
#include <stdlib.h>

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

/* streaming kernel for 16-byte aligned images (vfe_pyramid_stream.hip) */
bool pyr_down_stream_ok(const DImg& s, const DImg& d);
int pyr_down_stream_launch(const DImg& s, const DImg& d, hipStream_t st, bool* padded);

static bool dword_ok(const DImg& d) {
    return ((uintptr_t)d.data % 4 == 0) && (d.pitch % 4 == 0) && (d.fstride % 4 == 0);
}
static int launch_pyr_down(const DImg& s, const DImg& d, hipStream_t st, bool* padded) {
    *padded = false;
    if (pyr_down_stream_ok(s, d)) {
        ProfScope prof(VFE_K_PYR_DOWN, st);
        return pyr_down_stream_launch(s, d, st, padded);
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
    bool padded = false;
    if ((rc = launch_pyr_down(to_dimg(src), to_dimg(dst), (hipStream_t)stream, &padded))) return rc;
    return padded ? VFE_OK : launch_pad_border(to_dimg(dst), (hipStream_t)stream);
}

/* vfe_pyramid_fused.hip */
bool pyr_fused2_ok(const DImg& s1, const DImg& d2, const DImg& d3);
int pyr_fused2_launch(const DImg& s1, const DImg& d2, const DImg& d3, hipStream_t st);
/* Two levels per launch save a dependent launch; the streaming kernel moves more bytes per instruction.  Measured on 1080p
 * streams (levels 2 and 3 from level 1): +9 % of a front-end step at 4 streams, +7 % at 8, +2 % at 16, +1 % at 32, -2.5 %
 * at 64, -6 % at 256.  So: source levels of at most a megapixel, and only launches of at most this many 32 x 8 blocks. */
#define PYR_FUSE_MAX_PIXELS (1 << 20)
#define PYR_FUSE_MAX_TILES 4500

/* vfe_pyramid_stream3.hip: three levels per launch, one workgroup of up to eight waves per strip of a frame.  A lone
 * workgroup needs ~55 us for a 1080-row strip whatever the launch holds, so the launch must bring enough strips to fill
 * the chip.  Measured on 1080p (two strips per frame), this path against one launch per level: 16 frames 55 / 32 us,
 * 32 frames 56 / 50, 64 frames 65 / 64, 128 frames 84 / 113, 256 frames 154 / 206-217. */
bool pyr3_stream_ok(const DImg& s, const DImg& d1, const DImg& d2, const DImg& d3);
int pyr3_stream_launch(const DImg& s, const DImg& d1, const DImg& d2, const DImg& d3, hipStream_t st);
#define PYR3_MIN_BLOCKS 160
static long long pyr3_min_blocks() { /* VFE_PYR3_MIN_BLOCKS: for tests and measurements, read at every call */
    const char* e = getenv("VFE_PYR3_MIN_BLOCKS");
    return e && *e ? atoll(e) : (long long)PYR3_MIN_BLOCKS;
}

/* levels l, l+1, l+2 from level l-1 in one launch? */
static bool pyr3_applies(const vfe_dimg* levels, int nlevels, int l) {
    return l + 2 < nlevels && !vfe_check_dimg(&levels[l - 1], "src") && !vfe_check_dimg(&levels[l], "dst") &&
           !vfe_check_dimg(&levels[l + 1], "dst") && !vfe_check_dimg(&levels[l + 2], "dst") &&
           (long long)((levels[l].width / 8 + 61) / 62) * levels[l - 1].batch >= pyr3_min_blocks() &&
           pyr3_stream_ok(to_dimg(&levels[l - 1]), to_dimg(&levels[l]), to_dimg(&levels[l + 1]), to_dimg(&levels[l + 2]));
}
bool vfe_pyramid_three_levels_per_launch(const vfe_dimg* levels, int nlevels) { return levels && pyr3_applies(levels, nlevels, 1); }

extern "C" int vfe_dev_build_pyramid(const vfe_dimg* levels, int nlevels, void* stream) {
    VFE_REQUIRE(levels && nlevels >= 1 && nlevels <= VFE_MAX_LEVELS, "nlevels out of range");
    for (int l = 1; l < nlevels;) {
        int rc;
        if (pyr3_applies(levels, nlevels, l)) {
            ProfScope prof(VFE_K_PYR_DOWN, (hipStream_t)stream);
            if ((rc = pyr3_stream_launch(to_dimg(&levels[l - 1]), to_dimg(&levels[l]), to_dimg(&levels[l + 1]), to_dimg(&levels[l + 2]), (hipStream_t)stream))) return rc;
            l += 3;
            continue;
        }
        /* two small levels in one launch where the source level carries a border (i.e. from level 1 on) */
        if (l + 1 < nlevels && (long long)levels[l - 1].width * levels[l - 1].height <= PYR_FUSE_MAX_PIXELS &&
            (long long)((levels[l + 1].width + 31) / 32) * ((levels[l + 1].height + 7) / 8) * levels[l + 1].batch <= PYR_FUSE_MAX_TILES &&
            !vfe_check_dimg(&levels[l - 1], "src") && !vfe_check_dimg(&levels[l], "dst") && !vfe_check_dimg(&levels[l + 1], "dst") &&
            levels[l - 1].batch <= 65535 &&
            pyr_fused2_ok(to_dimg(&levels[l - 1]), to_dimg(&levels[l]), to_dimg(&levels[l + 1]))) {
            ProfScope prof(VFE_K_PYR_DOWN, (hipStream_t)stream);
            if ((rc = pyr_fused2_launch(to_dimg(&levels[l - 1]), to_dimg(&levels[l]), to_dimg(&levels[l + 1]), (hipStream_t)stream))) return rc;
            l += 2;
            continue;
        }
        if ((rc = vfe_dev_pyr_down(&levels[l - 1], &levels[l], stream))) return rc;
        l++;
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
