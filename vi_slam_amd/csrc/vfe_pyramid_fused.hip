/* The two coarsest levels of an optical-flow pyramid in one launch: level n+1 and n+2 from level n (cv::pyrDown twice,
 * 5x5 binomial kernel, reflect-101, one rounding per level: (sum + 128) >> 8).
 *
 * The small levels are short, latency-bound launches that depend on each other; beside a long kernel on another stream
 * each of them waits for slots again.  Here a workgroup owns a 32 x 8 block of the coarsest level: it loads the part of
 * level n that block depends on (137 x 41 pixels, as aligned dwords) into LDS, forms the 67 x 19 pixels of level n+1 in
 * between (horizontal pass to 16-bit sums, vertical pass with the rounding), writes the 64 x 16 of them it owns, and
 * computes its block of level n+2 from the LDS copy.  Level n must carry a reflect-101 border (>= 8 pixels: every tap of
 * a valid level-n+1 pixel then lies in memory); level n+1 is indexed with the reflection where level n+2 reaches over
 * its rim, exactly as a separate pyrDown would read it.  A pixel within `border` of a rim is also written to its mirror
 * positions, so both destination borders are complete when the launch ends. */
#include "vfe_common.h"

#define PF_T3W 32
#define PF_T3H 8
#define PF_R2W (2 * PF_T3W + 3)  /* 67 columns of level n+1 */
#define PF_R2H (2 * PF_T3H + 3)  /* 19 rows */
#define PF_R1DW 35               /* dwords per row of level n: columns 4 x3 - 8 .. 4 x3 + 131 */
#define PF_R1H (2 * PF_R2H + 3)  /* 41 rows */

/* v at (x, y) of frame b and at its mirror images in the border */
__device__ static inline void store_mirrored(const DImg& d, uint8_t* fp, int x, int y, uint8_t v) {
    const int B = d.border, W = d.w, H = d.h;
    int xs[3], ys[3], nx = 1, ny = 1;
    xs[0] = x; ys[0] = y;
    if (x >= 1 && x <= B) xs[nx++] = -x;
    if (x <= W - 2 && x >= W - 1 - B) xs[nx++] = 2 * (W - 1) - x;
    if (y >= 1 && y <= B) ys[ny++] = -y;
    if (y <= H - 2 && y >= H - 1 - B) ys[ny++] = 2 * (H - 1) - y;
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) fp[(ptrdiff_t)ys[j] * d.pitch + xs[i]] = v;
}

#define PF_PAIRS 34 /* pixel pairs of level n+1 per row held (68 columns, the last one spare) */

__global__ __launch_bounds__(256) void k_pyr_fused2(DImg s1, DImg d2, DImg d3, int tiles_x, int tiles_y) {
    __shared__ uint32_t s_l1[PF_R1H][PF_R1DW + 1];      /* level n, 140 bytes per row + a spare dword */
    __shared__ uint32_t s_h1[PF_R1H][PF_PAIRS];          /* horizontal sums, two 16-bit columns per dword */
    __shared__ uint8_t s_l2[PF_R2H][2 * PF_PAIRS];       /* level n+1 */
    const int tid = threadIdx.x;
    const unsigned per_frame = (unsigned)(tiles_x * tiles_y);
    const int b = blockIdx.x / per_frame, t = blockIdx.x - (unsigned)b * per_frame;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int x3_0 = tx * PF_T3W, y3_0 = ty * PF_T3H;
    const int x2_0 = 2 * x3_0 - 2, y2_0 = 2 * y3_0 - 2; /* first column / row of level n+1 held */
    const int x1_0 = 4 * x3_0 - 8, y1_0 = 2 * y2_0 - 2;  /* first column (a multiple of four) / row of level n held */
    const uint8_t* p1 = s1.frame(b);
    /* level n: dwords, clamped into the frame and its border (what the clamp changes only feeds pixels past the rim of
     * level n+1, which nothing reads) */
    {
        const int xmax = (s1.w + s1.border - 4) & ~3, ymax = s1.h + s1.border - 1;
        for (int i = tid; i < PF_R1H * (PF_R1DW + 1); i += 256) {
            const int r = i / (PF_R1DW + 1), c = i - r * (PF_R1DW + 1);
            const int X = min(max(x1_0 + 4 * c, -s1.border), xmax), Y = min(max(y1_0 + r, -s1.border), ymax);
            s_l1[r][c] = *reinterpret_cast<const uint32_t*>(p1 + (ptrdiff_t)Y * s1.pitch + X);
        }
    }
    __syncthreads();
    /* horizontal pass, two columns per item: column x2_0 + c is centred on byte 2 c + 4 of the row held, i.e. columns
     * 2 m and 2 m + 1 take bytes 4 m + 2 .. 4 m + 6 and 4 m + 4 .. 4 m + 8: four taps by one dot product each (weights
     * 1 4 6 4), the fifth (weight 1) in its accumulator */
    for (int i = tid; i < PF_R1H * PF_PAIRS; i += 256) {
        const int r = i / PF_PAIRS, m = i - r * PF_PAIRS;
        const uint32_t a = s_l1[r][m], c = s_l1[r][m + 1], e = s_l1[r][m + 2];
        const uint32_t lo = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbit(c, a, 16), 0x04060401u, (c >> 16) & 0xFFu, false);
        const uint32_t hi = __builtin_amdgcn_udot4(c, 0x04060401u, e & 0xFFu, false);
        s_h1[r][m] = lo | (hi << 16);
    }
    __syncthreads();
    /* vertical pass on the packed pairs (a column's sum stays below 2^16, so the halves never carry): row y2_0 + r is
     * centred on row 2 r + 2 of the rows held */
    uint8_t* q2 = d2.frame(b);
    for (int i = tid; i < PF_R2H * PF_PAIRS; i += 256) {
        const int r = i / PF_PAIRS, m = i - r * PF_PAIRS;
        const uint32_t sum = s_h1[2 * r][m] + s_h1[2 * r + 4][m] + ((s_h1[2 * r + 1][m] + s_h1[2 * r + 3][m]) << 2) + 6u * s_h1[2 * r + 2][m] +
                             0x00800080u;
        const uint8_t v0 = (uint8_t)(sum >> 8), v1 = (uint8_t)(sum >> 24);
        s_l2[r][2 * m] = v0; s_l2[r][2 * m + 1] = v1;
        const int X = x2_0 + 2 * m, Y = y2_0 + r; /* the block owned: columns 2 .. 65, rows 2 .. 17 of what is held */
        if (m >= 1 && m <= PF_T3W && r >= 2 && r < 2 + 2 * PF_T3H && Y < d2.h) {
            if (X < d2.w) store_mirrored(d2, q2, X, Y, v0);
            if (X + 1 < d2.w) store_mirrored(d2, q2, X + 1, Y, v1);
        }
    }
    __syncthreads();
    /* level n+2: one pixel per thread, level n+1 read with the reflection at its rim */
    {
        const int X3 = x3_0 + (tid & (PF_T3W - 1)), Y3 = y3_0 + tid / PF_T3W;
        if (X3 < d3.w && Y3 < d3.h) {
            int cx[5], cy[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                cx[k] = reflect101(2 * X3 + k - 2, d2.w) - x2_0;
                cy[k] = reflect101(2 * Y3 + k - 2, d2.h) - y2_0;
            }
            int sum = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint8_t* row = s_l2[cy[j]];
                const int h = row[cx[0]] + row[cx[4]] + 4 * (row[cx[1]] + row[cx[3]]) + 6 * row[cx[2]];
                sum += (j == 0 || j == 4) ? h : (j == 2 ? 6 * h : 4 * h);
            }
            store_mirrored(d3, d3.frame(b), X3, Y3, (uint8_t)((sum + 128) >> 8));
        }
    }
}

/* whether the launch applies: level n readable as dwords with at least 8 pixels of border, both destinations larger than
 * their border (the mirror images of a pixel are then single reflections), sizes halving as cv::pyrDown's do */
bool pyr_fused2_ok(const DImg& s1, const DImg& d2, const DImg& d3) {
    auto dw = [](const DImg& d) { return d.sub_n == 1 && (uintptr_t)d.data % 4 == 0 && d.pitch % 4 == 0 && d.fstride % 4 == 0; };
    if (!dw(s1) || s1.border < 8 || s1.border % 4 != 0) return false;
    if (d2.sub_n != 1 || d3.sub_n != 1) return false;
    if (d2.w != (s1.w + 1) / 2 || d2.h != (s1.h + 1) / 2 || d3.w != (d2.w + 1) / 2 || d3.h != (d2.h + 1) / 2) return false;
    if (d2.batch != s1.batch || d3.batch != s1.batch) return false;
    if (d2.w <= d2.border + 1 || d2.h <= d2.border + 1 || d3.w <= d3.border + 1 || d3.h <= d3.border + 1) return false;
    if (d3.w < 3 || d3.h < 3) return false;
    const long long tiles = (long long)div_up(d3.w, PF_T3W) * div_up(d3.h, PF_T3H) * s1.batch;
    return tiles < (1ll << 31);
}

int pyr_fused2_launch(const DImg& s1, const DImg& d2, const DImg& d3, hipStream_t st) {
    const int tiles_x = div_up(d3.w, PF_T3W), tiles_y = div_up(d3.h, PF_T3H);
    const unsigned total = (unsigned)(tiles_x * tiles_y) * (unsigned)s1.batch;
    hipLaunchKernelGGL(k_pyr_fused2, dim3(total), dim3(256), 0, st, s1, d2, d3, tiles_x, tiles_y);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
