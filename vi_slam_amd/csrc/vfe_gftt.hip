/* Shi-Tomasi corner detection (cv::goodFeaturesToTrack semantics) for gfx950, batched.
 *
 * Two kernels per call:
 *  k_gftt_eig    one pass over the image: Sobel -> gradient products -> 3x3 box sums ->
 *                smaller eigenvalue, all staged through LDS tiles with halos; fused with the
 *                frame maximum (atomic) and 3x3 non-max suppression, so the float map never
 *                goes to HBM.  Local maxima are appended to a per-frame candidate list as
 *                64-bit keys (score bits << 32 | linear index).  Candidates that cannot pass
 *                the final quality threshold are dropped early against the running maximum
 *                (a lower bound of the final one), which only removes keys the exact filter
 *                would remove anyway.
 *  k_gftt_select one workgroup per frame: exact threshold, then candidates are consumed in
 *                descending key order in batches (radix select of the next <= 8192 keys, bitonic
 *                sort in LDS) by a wave-parallel version of OpenCV's greedy min-distance walk:
 *                64 candidates are tested against the accepted set at once, conflicts inside
 *                the 64 are resolved in rank order with ballots.  The result equals the serial
 *                walk because a candidate's fate depends only on higher-ranked ones.
 *
 * Float arithmetic follows the CPU oracle operation by operation (no contraction), so corner
 * sets are bit-exact.
 */
#include "vfe_common.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define GF_TW 64
#define GF_TH 16
#define GF_IWB 80            /* staged tile row: 20 dwords; tile byte t <-> image x = ox - 4 + t */
#define GF_IH (GF_TH + 6)
#define GF_CW (GF_TW + 4)
#define GF_CH (GF_TH + 4)
#define GF_HW (GF_TW + 2)
#define GF_EH (GF_TH + 2)
#define GF_MAX_WGS 2048u      /* persistent workgroups: 256 CUs x 8 */
#define GF_CSTRIDE 64         /* u32 stride between counters: one 256-B line each, so atomics on
                                different frames / segments go to different L2 channels */
#define GF_NSEG 8            /* candidate lists per frame (spreads the reservation atomics) */

#define SEL_THREADS 1024
#define SEL_M 8192

/* Workgroup id -> (tile, frame) with the frame index fastest, so that consecutive workgroups
 * update different frames' counters and maxima. */
template <bool WRITE_MAP, bool ALIGNED>
__global__ __launch_bounds__(256) void k_gftt_eig(DImg img, int tiles_x, unsigned total_work, float* __restrict__ eigmap,
                                                  u32* __restrict__ frame_max, u32* __restrict__ cand_count,
                                                  u64* __restrict__ cand, u32 seg_cap, double quality) {
    __shared__ __align__(16) uint32_t s_imgw[GF_IH][GF_IWB / 4];
    __shared__ float s_cov[3][GF_CH][GF_CW];
    __shared__ float s_h[3][GF_CH][GF_HW];
    __shared__ float s_eig[GF_EH][GF_HW];
    __shared__ float s_red[4];
    __shared__ float s_prune;
    __shared__ u32 s_ncand, s_base, s_seen;
    const uint8_t* s_img = reinterpret_cast<const uint8_t*>(&s_imgw[0][0]);

    const int tid = threadIdx.x;
    const int W = img.w, H = img.h;
    /* persistent workgroups: a few thousand workgroups stride over all (tile, frame) items,
     * which costs far less than dispatching one workgroup per tile */
    for (unsigned work = blockIdx.x; work < total_work; work += gridDim.x) {
    const int b = work % img.batch, tile = work / img.batch;
    const uint8_t* ip = img.frame(b);
    const int ox = (tile % tiles_x) * GF_TW, oy = (tile / tiles_x) * GF_TH;

    if (tid == 0) {
        s_ncand = 0;
        if (!WRITE_MAP) {
            const u32 rm = __hip_atomic_load(&frame_max[b * GF_CSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_seen = rm;
            s_prune = (float)((double)__uint_as_float(rm) * quality);
        }
    }
    for (int i = tid; i < GF_IH * (GF_IWB / 4); i += 256) {
        const int ty = i / (GF_IWB / 4), k = i - ty * (GF_IWB / 4);
        const int gx = ox - 4 + 4 * k, gy = oy - 3 + ty;
        uint32_t v = 0;
        if (gx <= W + 2 && gy <= H + 2) {
            const uint8_t* row = ip + (size_t)reflect101(gy, H) * img.pitch;
            if (ALIGNED && gx >= 0 && gx + 3 < W)
                v = *reinterpret_cast<const uint32_t*>(row + gx);
            else
                v = (uint32_t)row[reflect101(gx, W)] | ((uint32_t)row[reflect101(gx + 1, W)] << 8) |
                    ((uint32_t)row[reflect101(gx + 2, W)] << 16) | ((uint32_t)row[reflect101(gx + 3, W)] << 24);
        }
        s_imgw[ty][k] = v;
    }
    __syncthreads();

    const double scale = 1.0 / ((1 << 2) * 3 * 255.0);
    const float k_o = (float)scale, k_c = (float)(2.0 * scale);
    for (int i = tid; i < GF_CH * GF_CW; i += 256) {
        const int cy = i / GF_CW, cx = i - cy * GF_CW;
        const int px = ox - 2 + cx, py = oy - 2 + cy;
        float xx = 0.f, xy = 0.f, yy = 0.f;
        if (px >= -1 && px <= W && py >= -1 && py <= H) {
            /* products live on the image grid; positions outside are the reflection of the
             * product plane, i.e. the gradient taken at the reflected centre */
            const int tx = reflect101(px, W) - (ox - 4), ty = reflect101(py, H) - (oy - 3);
            float rx[3], ry[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint8_t* r = s_img + (ty - 1 + k) * GF_IWB + tx;
                const int a = r[-1], m = r[0], c = r[1];
                rx[k] = (float)(c - a);
                ry[k] = k_c * (float)m + k_o * (float)(a + c);
            }
            const float dx = k_c * rx[1] + k_o * (rx[0] + rx[2]);
            const float dy = ry[2] - ry[0];
            xx = dx * dx; xy = dx * dy; yy = dy * dy;
        }
        s_cov[0][cy][cx] = xx; s_cov[1][cy][cx] = xy; s_cov[2][cy][cx] = yy;
    }
    __syncthreads();

    for (int i = tid; i < GF_CH * GF_HW; i += 256) {
        const int hy = i / GF_HW, hx = i - hy * GF_HW;
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
            s_h[pl][hy][hx] = (s_cov[pl][hy][hx] + s_cov[pl][hy][hx + 1]) + s_cov[pl][hy][hx + 2];
    }
    __syncthreads();

    for (int i = tid; i < GF_EH * GF_HW; i += 256) {
        const int ey = i / GF_HW, ex = i - ey * GF_HW;
        const int px = ox - 1 + ex, py = oy - 1 + ey;
        float e = -1.f; /* outside the image: never a maximum, never a neighbour that wins */
        if (px >= 0 && px < W && py >= 0 && py < H) {
            const float sxx = (s_h[0][ey][ex] + s_h[0][ey + 1][ex]) + s_h[0][ey + 2][ex];
            const float sxy = (s_h[1][ey][ex] + s_h[1][ey + 1][ex]) + s_h[1][ey + 2][ex];
            const float syy = (s_h[2][ey][ex] + s_h[2][ey + 1][ex]) + s_h[2][ey + 2][ex];
            const float a = sxx * 0.5f, bb = sxy, cc = syy * 0.5f;
            const float d = a - cc;
            e = (a + cc) - sqrtf(d * d + bb * bb);
        }
        s_eig[ey][ex] = e;
    }
    __syncthreads();

    float bmax = 0.f;
    const float prune = WRITE_MAP ? 0.f : s_prune;
    u64 mykeys[GF_TW * GF_TH / 256];
    int nmine = 0;
#pragma unroll
    for (int k = 0; k < GF_TW * GF_TH / 256; k++) {
        const int i = tid + k * 256;
        const int y = i / GF_TW, x = i - y * GF_TW;
        const int px = ox + x, py = oy + y;
        mykeys[k] = 0;
        if (px >= W || py >= H) continue;
        const float v = s_eig[y + 1][x + 1];
        if (WRITE_MAP) {
            eigmap[((size_t)b * H + py) * W + px] = v;
        } else {
            bmax = fmaxf(bmax, v);
            if (px >= 1 && px <= W - 2 && py >= 1 && py <= H - 2 && v > 0.f && v > prune) {
                float m = fmaxf(fmaxf(s_eig[y][x], s_eig[y][x + 1]), s_eig[y][x + 2]);
                m = fmaxf(m, fmaxf(s_eig[y + 1][x], s_eig[y + 1][x + 2]));
                m = fmaxf(m, fmaxf(fmaxf(s_eig[y + 2][x], s_eig[y + 2][x + 1]), s_eig[y + 2][x + 2]));
                if (v >= m) {
                    mykeys[k] = ((u64)__float_as_uint(v) << 32) | (u32)(py * W + px);
                    nmine++;
                }
            }
        }
    }
    if (!WRITE_MAP) {
        /* one reservation per workgroup: LDS counter for the local offsets, one global atomic */
        u32 lpos = 0;
        if (nmine) lpos = atomicAdd(&s_ncand, (u32)nmine);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, o));
        if ((tid & 63) == 0) s_red[tid >> 6] = bmax;
        __syncthreads();
        const int seg = tile % GF_NSEG;
        if (tid == 0) {
            const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
            /* positive floats order as uints; most tiles cannot raise the maximum already seen */
            if (m > 0.f && __float_as_uint(m) > s_seen) atomicMax(&frame_max[b * GF_CSTRIDE], __float_as_uint(m));
            s_base = s_ncand ? atomicAdd(&cand_count[(b * GF_NSEG + seg) * GF_CSTRIDE], s_ncand) : 0u;
        }
        __syncthreads();
        if (nmine) {
            u64* dst = cand + ((size_t)b * GF_NSEG + seg) * seg_cap;
            u32 pos = s_base + lpos;
#pragma unroll
            for (int k = 0; k < GF_TW * GF_TH / 256; k++)
                if (mykeys[k]) { if (pos < seg_cap) dst[pos] = mykeys[k]; pos++; }
        }
    }
    __syncthreads(); /* LDS tiles are reused by the next work item */
    }
}

/* ---- register-marching eigenvalue kernel ------------------------------------------------------
 * One wavefront owns a strip of 64 lanes x 4 columns and walks down a band of rows.  Each lane
 * keeps the last three rows of everything it needs in registers (image bytes, horizontal sums of
 * the gradient products, eigenvalues, horizontal eigenvalue maxima); values of the neighbouring
 * columns come from the adjacent lanes by cross-lane shuffles.  No LDS tiles, no barriers; one
 * coalesced dword load per lane per image row.  Lanes 0,1 and 62,63 are halo: 240 output columns
 * per strip.  The arithmetic (operation order, roundings, reflect-101 of the product planes) is
 * that of the tiled kernel above and of the CPU checker, so results stay bit-exact.
 * Candidates are staged per wave in LDS and flushed with one reservation atomic per 192+ keys. */
#define EV_OUT 240
#define EV_STAGE 256

__device__ static inline float shl1(float v) { return __shfl_down(v, 1); } /* value of lane + 1 */
__device__ static inline float shr1(float v) { return __shfl_up(v, 1); }   /* value of lane - 1 */

template <bool WRITE_MAP, bool ALIGNED>
__global__ __launch_bounds__(256) void k_gftt_eig2(DImg img, int strips_x, int band_h, int bands_y, unsigned total_work,
                                                   float* __restrict__ eigmap, u32* __restrict__ frame_max,
                                                   u32* __restrict__ cand_count, u64* __restrict__ cand, u32 seg_cap,
                                                   double quality) {
    __shared__ u64 s_stage[4][EV_STAGE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned wave = blockIdx.x * 4u + wv, nwaves = gridDim.x * 4u;
    const int W = img.w, H = img.h;
    const double scale = 1.0 / ((1 << 2) * 3 * 255.0);
    const float k_o = (float)scale, k_c = (float)(2.0 * scale);
    u64* stage = s_stage[wv];
    const u64 lt_mask = (1ull << lane) - 1ull;

    for (unsigned work = wave; work < total_work; work += nwaves) {
        const int b = work % img.batch, t = work / img.batch;
        const int sxi = t % strips_x, byi = t / strips_x;
        const int X0 = sxi * EV_OUT + 4 * (lane - 2); /* image column of this lane's pixel 0 */
        const int y0 = byi * band_h, y1 = min(y0 + band_h, H);
        const uint8_t* ip = img.frame(b);
        const bool out_lane = lane >= 2 && lane < 62;
        u32 seen = 0;
        float prune = 0.f, wmax = 0.f;
        if (!WRITE_MAP) {
            seen = __hip_atomic_load(&frame_max[b * GF_CSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wmax = __uint_as_float(seen);
            prune = (float)((double)wmax * quality);
        }
        int nbuf = 0;
        const int seg = t % GF_NSEG;
        u64* gdst = cand + ((size_t)b * GF_NSEG + seg) * seg_cap;
        u32* gcnt = cand_count + (size_t)(b * GF_NSEG + seg) * GF_CSTRIDE;
        /* column flags: gradient sign flips where the product plane is a reflection */
        bool xin[4], xflip[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int x = X0 + k; xin[k] = x >= 0 && x < W; xflip[k] = (x == -1 || x == W); }

        u32 rA = 0, rB = 0, rC = 0;           /* image rows yy-2, yy-1, yy: this lane's 4 bytes */
        /* three-row rings, index 2 = newest row; shifted with plain moves so that every index is
         * a compile-time constant (runtime-indexed register arrays would spill to scratch) */
        float hs[3][3][4];                    /* [row][plane][pixel]: rows P-2, P-1, P */
        float eg[3][4], hm[3][4];             /* rows E-2, E-1, E */
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) { eg[i][k] = -1.f; hm[i][k] = -1.f; hs[i][0][k] = hs[i][1][k] = hs[i][2][k] = 0.f; }

        for (int yy = y0 - 3; yy <= y1 + 2; yy++) {
            /* ---- load image row yy (reflect-101) ---- */
            rA = rB; rB = rC;
            {
                const uint8_t* row = ip + (size_t)reflect101(yy, H) * img.pitch;
                if (ALIGNED && X0 >= 0 && X0 + 3 < W) rC = *reinterpret_cast<const uint32_t*>(row + X0);
                else if (X0 + 3 < -4 || X0 > W + 4) rC = 0;
                else rC = (u32)row[reflect101(X0, W)] | ((u32)row[reflect101(X0 + 1, W)] << 8) |
                          ((u32)row[reflect101(X0 + 2, W)] << 16) | ((u32)row[reflect101(X0 + 3, W)] << 24);
            }
            if (yy < y0 - 1) continue;
            /* ---- gradient products at row P = yy - 1 ---- */
            const int P = yy - 1;
            float cv[3][4];
            {
                /* 6 columns X0-1 .. X0+4 of the three rows */
                int c[3][6];
                const u32 rows[3] = {rA, rB, rC};
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const u32 l = (u32)__shfl_up((int)rows[i], 1), r = (u32)__shfl_down((int)rows[i], 1);
                    c[i][0] = (int)(l >> 24);
                    c[i][1] = (int)(rows[i] & 255); c[i][2] = (int)((rows[i] >> 8) & 255);
                    c[i][3] = (int)((rows[i] >> 16) & 255); c[i][4] = (int)(rows[i] >> 24);
                    c[i][5] = (int)(r & 255);
                }
                const bool yflip = (P == -1 || P == H);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float rx[3], ry[3];
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const int a = c[i][k], m = c[i][k + 1], d = c[i][k + 2];
                        rx[i] = (float)(d - a);
                        ry[i] = k_c * (float)m + k_o * (float)(a + d);
                    }
                    float dx = k_c * rx[1] + k_o * (rx[0] + rx[2]);
                    float dy = ry[2] - ry[0];
                    if (xflip[k]) dx = -dx;
                    if (yflip) dy = -dy;
                    cv[0][k] = dx * dx; cv[1][k] = dx * dy; cv[2][k] = dy * dy;
                }
            }
            /* ---- horizontal 3-sums at row P ---- */
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                const float l = shr1(cv[pl][3]), r = shl1(cv[pl][0]);
#pragma unroll
                for (int k = 0; k < 4; k++) { hs[0][pl][k] = hs[1][pl][k]; hs[1][pl][k] = hs[2][pl][k]; }
                hs[2][pl][0] = (l + cv[pl][0]) + cv[pl][1];
                hs[2][pl][1] = (cv[pl][0] + cv[pl][1]) + cv[pl][2];
                hs[2][pl][2] = (cv[pl][1] + cv[pl][2]) + cv[pl][3];
                hs[2][pl][3] = (cv[pl][2] + cv[pl][3]) + r;
            }
            if (yy < y0 + 1) continue;
            /* ---- eigenvalue at row E = yy - 2 (rows E-1, E, E+1 of the sums), and its 3-max ---- */
            const int E = yy - 2;
            {
                const bool yin = E >= 0 && E < H;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    eg[0][k] = eg[1][k]; eg[1][k] = eg[2][k];
                    hm[0][k] = hm[1][k]; hm[1][k] = hm[2][k];
                    float e = -1.f;
                    if (yin && xin[k]) {
                        const float sxx = (hs[0][0][k] + hs[1][0][k]) + hs[2][0][k];
                        const float sxy = (hs[0][1][k] + hs[1][1][k]) + hs[2][1][k];
                        const float syy = (hs[0][2][k] + hs[1][2][k]) + hs[2][2][k];
                        const float a = sxx * 0.5f, bb = sxy, cc = syy * 0.5f;
                        const float d = a - cc;
                        e = (a + cc) - sqrtf(d * d + bb * bb);
                    }
                    eg[2][k] = e;
                }
                const float l = shr1(eg[2][3]), r = shl1(eg[2][0]);
                hm[2][0] = fmaxf(fmaxf(l, eg[2][0]), eg[2][1]);
                hm[2][1] = fmaxf(fmaxf(eg[2][0], eg[2][1]), eg[2][2]);
                hm[2][2] = fmaxf(fmaxf(eg[2][1], eg[2][2]), eg[2][3]);
                hm[2][3] = fmaxf(fmaxf(eg[2][2], eg[2][3]), r);
            }
            if (yy < y0 + 3) continue;
            /* ---- output row N = yy - 3: value eg[1], neighbours through hm[0..2] ---- */
            const int N = yy - 3;
            if (N >= y1) continue;
            if (WRITE_MAP) {
                if (out_lane) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (X0 + k < W) eigmap[((size_t)b * H + N) * W + X0 + k] = eg[1][k];
                }
            } else {
                const bool yok = N >= 1 && N <= H - 2;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float v = eg[1][k];
                    const int x = X0 + k;
                    bool isc = false;
                    if (out_lane && x < W) {
                        wmax = fmaxf(wmax, v);
                        if (yok && x >= 1 && x <= W - 2 && v > 0.f && v > prune)
                            isc = v >= fmaxf(fmaxf(hm[0][k], hm[1][k]), hm[2][k]);
                    }
                    const u64 m = __ballot(isc);
                    if (m) {
                        const int cnt = __popcll(m);
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
                        if (isc) stage[nbuf + __popcll(m & lt_mask)] = ((u64)__float_as_uint(v) << 32) | (u32)(N * W + x);
                        nbuf += cnt;
                    }
                }
            }
        }
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
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
            if (lane == 0 && wmax > 0.f && __float_as_uint(wmax) > seen) atomicMax(&frame_max[b * GF_CSTRIDE], __float_as_uint(wmax));
        }
    }
}

struct EigGeom { int strips_x, band_h, bands_y; unsigned total; };
static EigGeom eig_geom(int w, int h, int batch) {
    EigGeom g;
    g.strips_x = div_up(w, EV_OUT);
    g.band_h = 64;
    while (g.band_h > 16 && (long long)g.strips_x * div_up(h, g.band_h) * batch < 16384) g.band_h /= 2;
    g.bands_y = div_up(h, g.band_h);
    g.total = (unsigned)(g.strips_x * g.bands_y * batch);
    return g;
}

struct SelParams {
    int W, H, max_corners;
    double quality;
    int use_grid, cell, gw, gh;
    u32 md2i;
    const u32* frame_max;
    const u32* cand_count;
    const u64* cand;
    u32 seg_cap;     /* keys per candidate segment; GF_NSEG segments per frame */
    u32* heads;      /* [batch][gw*gh], 0 = empty, else node index + 1 */
    u32* node_xy;    /* [batch][node_cap]  x | y << 16 */
    u32* node_next;  /* [batch][node_cap] */
    int node_cap;
    const float* seeds;
    const int* n_seeds;
    int seed_cap;
    float* corners;
    float* scores;
    int* counts;
};

__device__ static inline u32 ld_agent(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline void st_agent(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* inclusive suffix scan of 1024 values held one per thread; returns sum over threads >= tid */
__device__ static inline u32 block_suffix_scan(u32 v, u32 (*buf)[SEL_THREADS]) {
    const int tid = threadIdx.x;
    int cur = 0;
    buf[0][tid] = v;
    __syncthreads();
    for (int o = 1; o < SEL_THREADS; o <<= 1) {
        u32 x = buf[cur][tid];
        if (tid + o < SEL_THREADS) x += buf[cur][tid + o];
        buf[cur ^ 1][tid] = x;
        cur ^= 1;
        __syncthreads();
    }
    const u32 r = buf[cur][tid];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(SEL_THREADS) void k_gftt_select(SelParams p) {
    __shared__ u64 s_keys[SEL_M];
    __shared__ u32 s_hist[2048 + 1];
    __shared__ u32 s_scan[2][SEL_THREADS];
    __shared__ u32 s_cnt;
    __shared__ u32 s_dmin;
    __shared__ int s_nacc;

    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ u32 s_segn[GF_NSEG];
    if (tid < GF_NSEG) s_segn[tid] = min(p.cand_count[(b * GF_NSEG + tid) * GF_CSTRIDE], p.seg_cap);
    const u64* keys = p.cand + (size_t)b * GF_NSEG * p.seg_cap;
/* visit every candidate key of this frame */
#define FOR_EACH_KEY(k)                                              \
    for (int sg_ = 0; sg_ < GF_NSEG; sg_++) {                         \
        const u32 cn_ = s_segn[sg_];                                 \
        const u64* kp_ = keys + (size_t)sg_ * p.seg_cap;             \
        for (u32 i_ = tid; i_ < cn_; i_ += SEL_THREADS) {            \
            const u64 k = kp_[i_];
#define END_EACH_KEY }}
    const float maxv = __uint_as_float(p.frame_max[b * GF_CSTRIDE]);
    const float thr = (float)((double)maxv * p.quality);
    const u64 lowkey = ((u64)__float_as_uint(thr) << 32) | 0xFFFFFFFFull; /* valid keys are > lowkey */
    u32* heads = p.heads + (size_t)b * p.gw * p.gh;
    u32* node_xy = p.node_xy + (size_t)b * p.node_cap;
    u32* node_next = p.node_next + (size_t)b * p.node_cap;

    if (tid == 0) { s_cnt = 0; s_nacc = 0; }
    /* refill budget: tracked points count towards max_corners */
    const int ns = (p.seeds && p.n_seeds) ? max(0, min(p.n_seeds[b], p.seed_cap)) : 0;
    const int limit = max(0, p.max_corners - ns);
    /* tracked points occupy the grid first */
    if (p.use_grid && ns > 0) {
        const float* sd = p.seeds + (size_t)b * p.seed_cap * 2;
        for (int s = tid; s < ns; s += SEL_THREADS) {
            const int sx = __float2int_rn(sd[2 * s]), sy = __float2int_rn(sd[2 * s + 1]);
            if (sx < 0 || sy < 0 || sx >= p.W || sy >= p.H) continue;
            st_agent(&node_xy[s], (u32)sx | ((u32)sy << 16));
            const u32 old = atomicExch(&heads[(sy / p.cell) * p.gw + sx / p.cell], (u32)s + 1);
            st_agent(&node_next[s], old);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    /* how many candidates pass the exact threshold */
    u32 mine = 0;
    FOR_EACH_KEY(k) mine += (k > lowkey) ? 1u : 0u; END_EACH_KEY
    const u32 total_valid = block_suffix_scan(mine, s_scan) /* value at tid 0 */;
    __shared__ u32 s_total;
    if (tid == 0) s_total = total_valid;
    __syncthreads();
    const u32 total = s_total;

    u64 U = ~0ull; /* exclusive upper bound of keys not yet consumed */
    u32 consumed = 0;
    int n_acc = 0;
    while (consumed < total && n_acc < limit) {
        u64 T;
        if (total - consumed <= SEL_M) {
            T = lowkey + 1;
        } else {
            /* MSD radix select: largest top range [T, U) holding <= SEL_M keys */
            u64 prefix = 0;
            int hi_shift = 64; /* keys are compared on bits [hi_shift, 64) against prefix */
            T = 0;
            for (;;) {
                const int bits = hi_shift >= 11 ? 11 : hi_shift;
                const int shift = hi_shift - bits;
                const u32 nb = 1u << bits;
                for (u32 i = tid; i <= 2048; i += SEL_THREADS) s_hist[i] = 0;
                __syncthreads();
                FOR_EACH_KEY(k)
                    if (k > lowkey && k < U && (hi_shift == 64 || (k >> hi_shift) == prefix))
                        atomicAdd(&s_hist[(u32)(k >> shift) & (nb - 1)], 1u);
                END_EACH_KEY
                __syncthreads();
                const u32 h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
                const u32 ss = block_suffix_scan(h0 + h1, s_scan);
                /* S(d) = number of keys in bins >= d */
                s_hist[2 * tid] = ss;
                s_hist[2 * tid + 1] = ss - h0;
                if (tid == 0) s_hist[2048] = 0;
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const u32 d = 2 * tid + q;
                    if (s_hist[d] <= SEL_M && (d == 0 || s_hist[d - 1] > SEL_M)) s_dmin = d;
                }
                if (tid == 0 && s_hist[2047] > SEL_M) s_dmin = 2048;
                __syncthreads();
                const u32 dmin = s_dmin;
                const u32 taken = s_hist[dmin];
                __syncthreads();
                if (taken > 0 || dmin == 0) {
                    T = (hi_shift == 64 ? 0ull : (prefix << hi_shift)) | ((u64)dmin << shift);
                    break;
                }
                /* bin dmin-1 alone holds more than SEL_M keys: refine inside it */
                prefix = (hi_shift == 64 ? 0ull : (prefix << bits)) | (u64)(dmin - 1);
                hi_shift = shift;
            }
            if (T <= lowkey) T = lowkey + 1;
        }
        /* collect [T, U) */
        FOR_EACH_KEY(k)
            if (k >= T && k < U) {
                const u32 pos = atomicAdd(&s_cnt, 1u);
                if (pos < SEL_M) s_keys[pos] = k;
            }
        END_EACH_KEY
        __syncthreads();
        const u32 taken = min(s_cnt, (u32)SEL_M);
        u32 n2 = 64;
        while (n2 < taken) n2 <<= 1;
        for (u32 i = taken + tid; i < n2; i += SEL_THREADS) s_keys[i] = 0;
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        /* bitonic sort, descending */
        for (u32 k = 2; k <= n2; k <<= 1)
            for (u32 j = k >> 1; j > 0; j >>= 1) {
                for (u32 i = tid; i < n2; i += SEL_THREADS) {
                    const u32 l = i ^ j;
                    if (l > i) {
                        const u64 a = s_keys[i], c = s_keys[l];
                        const bool desc = (i & k) == 0;
                        if (desc ? (a < c) : (a > c)) { s_keys[i] = c; s_keys[l] = a; }
                    }
                }
                __syncthreads();
            }

        /* wave-parallel greedy walk over the sorted batch */
        if (tid < 64) {
            const int lane = tid;
            int acc = s_nacc;
            for (u32 base = 0; base < taken && acc < limit; base += 64) {
                const u32 i = base + lane;
                const bool valid = i < taken;
                const u64 key = valid ? s_keys[i] : 0ull;
                const u32 idx = (u32)key;
                const int y = (int)(idx / (u32)p.W), x = (int)(idx - (u32)y * (u32)p.W);
                bool alive = valid;
                int xc = 0, yc = 0;
                if (p.use_grid) {
                    xc = x / p.cell; yc = y / p.cell;
                    if (alive) {
                        const int x1 = max(0, xc - 1), x2 = min(p.gw - 1, xc + 1);
                        const int y1 = max(0, yc - 1), y2 = min(p.gh - 1, yc + 1);
                        for (int yy = y1; yy <= y2 && alive; yy++)
                            for (int xx = x1; xx <= x2 && alive; xx++) {
                                u32 n = ld_agent(&heads[yy * p.gw + xx]);
                                while (n) {
                                    const u32 q = ld_agent(&node_xy[n - 1]);
                                    const int dx = x - (int)(q & 0xFFFF), dy = y - (int)(q >> 16);
                                    if ((u32)(dx * dx + dy * dy) < p.md2i) { alive = false; break; }
                                    n = ld_agent(&node_next[n - 1]);
                                }
                            }
                    }
                }
                u64 mask = __ballot(alive);
                while (mask && acc < limit) {
                    const int l = __ffsll((long long)mask) - 1;
                    const int ax = __shfl(x, l), ay = __shfl(y, l);
                    if (lane == l) {
                        float* c = p.corners + ((size_t)b * p.max_corners + acc) * 2;
                        c[0] = (float)x; c[1] = (float)y;
                        if (p.scores) p.scores[(size_t)b * p.max_corners + acc] = __uint_as_float((u32)(key >> 32));
                        if (p.use_grid) {
                            const u32 n = (u32)(p.seed_cap + acc);
                            u32* hd = &heads[yc * p.gw + xc];
                            st_agent(&node_xy[n], (u32)x | ((u32)y << 16));
                            st_agent(&node_next[n], ld_agent(hd));
                            st_agent(hd, n + 1);
                        }
                    }
                    acc++;
                    if (p.use_grid) {
                        const int dx = x - ax, dy = y - ay;
                        if ((u32)(dx * dx + dy * dy) < p.md2i) alive = false;
                    } else if (lane == l) {
                        alive = false;
                    }
                    mask = __ballot(alive) & ~((2ull << l) - 1ull);
                }
                /* this chunk's insertions must be in L2 before the next chunk looks them up */
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (lane == 0) s_nacc = acc;
        }
        __syncthreads();
        n_acc = s_nacc;
        consumed += taken;
        U = T;
    }
    if (tid == 0) p.counts[b] = n_acc;
}

/* workspace layout (all offsets 256-byte aligned) */
struct GfWs {
    size_t hdr_off, heads_off, cand_off, nxy_off, nnext_off, total;
    size_t zero_bytes; /* [hdr_off, hdr_off + zero_bytes) is cleared every call */
    int cell, gw, gh, use_grid, node_cap;
    u32 seg_cap;
};

static GfWs gf_layout(int w, int h, int batch, int max_corners, int seed_cap, double min_distance) {
    GfWs L;
    L.use_grid = min_distance >= 1.0;
    L.cell = L.use_grid ? (int)lrint(min_distance) : 1;
    if (L.cell < 1) L.cell = 1;
    L.gw = L.use_grid ? (w + L.cell - 1) / L.cell : 1;
    L.gh = L.use_grid ? (h + L.cell - 1) / L.cell : 1;
    L.node_cap = seed_cap + max_corners;
    size_t o = 0;
    L.hdr_off = o; o += align_up((size_t)batch * (1 + GF_NSEG) * GF_CSTRIDE * sizeof(u32), 256);
    L.heads_off = o; o += align_up((size_t)batch * L.gw * L.gh * sizeof(u32), 256);
    L.zero_bytes = o;
    {   /* a segment receives every GF_NSEG-th (strip, band) item; each can emit one key per pixel */
        const EigGeom g = eig_geom(w, h, batch);
        L.seg_cap = (u32)div_up(g.strips_x * g.bands_y, GF_NSEG) * (u32)(EV_OUT * g.band_h);
    }
    L.cand_off = o; o += align_up((size_t)batch * GF_NSEG * L.seg_cap * sizeof(u64), 256);
    L.nxy_off = o; o += align_up((size_t)batch * L.node_cap * sizeof(u32), 256);
    L.nnext_off = o; o += align_up((size_t)batch * L.node_cap * sizeof(u32), 256);
    L.total = o;
    return L;
}

extern "C" size_t vfe_good_features_workspace(int width, int height, int batch, int max_corners, int seed_cap) {
    if (width <= 0 || height <= 0 || batch <= 0 || max_corners <= 0 || seed_cap < 0) return 0;
    /* the grid is largest for min_distance = 1 (one cell per pixel) */
    return gf_layout(width, height, batch, max_corners, seed_cap, 1.0).total;
}

extern "C" int vfe_dev_good_features(const vfe_dimg* img, int max_corners, double quality, double min_distance,
                                     const float* seeds_xy, const int* n_seeds, int seed_cap,
                                     float* corners_xy, float* scores, int* counts,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    int rc = vfe_check_dimg(img, "img");
    if (rc) return rc;
    VFE_REQUIRE(img->width >= 8 && img->height >= 8, "image must be at least 8x8");
    VFE_REQUIRE((size_t)img->width * img->height < (1u << 31), "image too large");
    VFE_REQUIRE(img->width < 65536 && img->height < 65536, "image too large");
    VFE_REQUIRE(max_corners > 0, "max_corners must be > 0");
    VFE_REQUIRE(quality > 0 && quality < 1e6, "quality out of range");
    VFE_REQUIRE(min_distance >= 0 && min_distance < 32768, "min_distance out of range");
    VFE_REQUIRE(corners_xy && counts && workspace, "NULL output or workspace");
    VFE_REQUIRE(seed_cap >= 0 && (seed_cap == 0 || !seeds_xy == !n_seeds), "seeds_xy and n_seeds go together");
    VFE_REQUIRE(img->batch <= 65535, "batch too large");
    if (!seeds_xy) seed_cap = 0;
    const GfWs L = gf_layout(img->width, img->height, img->batch, max_corners, seed_cap, min_distance);
    if (workspace_bytes < L.total) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
        return VFE_E_NOMEM;
    }
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    VFE_HIP(hipMemsetAsync(ws + L.hdr_off, 0, L.zero_bytes, st));
    u32* frame_max = (u32*)(ws + L.hdr_off);
    u32* cand_count = frame_max + (size_t)img->batch * GF_CSTRIDE;
    DImg d = to_dimg(img);
    const EigGeom g = eig_geom(d.w, d.h, d.batch);
    const unsigned nblk = (g.total + 3) / 4;
    dim3 grid(nblk < GF_MAX_WGS ? nblk : GF_MAX_WGS);
    const bool aligned = ((uintptr_t)d.data % 4 == 0) && (d.pitch % 4 == 0) && (d.fstride % 4 == 0);
    {
        ProfScope prof(VFE_K_GFTT_EIG, st);
        if (aligned)
            hipLaunchKernelGGL((k_gftt_eig2<false, true>), grid, dim3(256), 0, st, d, g.strips_x, g.band_h, g.bands_y, g.total,
                               (float*)nullptr, frame_max, cand_count, (u64*)(ws + L.cand_off), L.seg_cap, quality);
        else
            hipLaunchKernelGGL((k_gftt_eig2<false, false>), grid, dim3(256), 0, st, d, g.strips_x, g.band_h, g.bands_y, g.total,
                               (float*)nullptr, frame_max, cand_count, (u64*)(ws + L.cand_off), L.seg_cap, quality);
    }
    VFE_LAUNCH_CHECK();
    SelParams p;
    p.W = d.w; p.H = d.h; p.max_corners = max_corners; p.quality = quality;
    p.use_grid = L.use_grid; p.cell = L.cell; p.gw = L.gw; p.gh = L.gh;
    p.md2i = (u32)ceil(min_distance * min_distance);
    p.frame_max = frame_max; p.cand_count = cand_count; p.cand = (const u64*)(ws + L.cand_off); p.seg_cap = L.seg_cap;
    p.heads = (u32*)(ws + L.heads_off); p.node_xy = (u32*)(ws + L.nxy_off); p.node_next = (u32*)(ws + L.nnext_off);
    p.node_cap = L.node_cap;
    p.seeds = seeds_xy; p.n_seeds = n_seeds; p.seed_cap = seed_cap;
    p.corners = corners_xy; p.scores = scores; p.counts = counts;
    {
        ProfScope prof(VFE_K_GFTT_SELECT, st);
        hipLaunchKernelGGL(k_gftt_select, dim3(d.batch), dim3(SEL_THREADS), 0, st, p);
    }
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_min_eigen_map(const vfe_dimg* img, float* eig, void* stream) {
    int rc = vfe_check_dimg(img, "img");
    if (rc) return rc;
    VFE_REQUIRE(img->width >= 8 && img->height >= 8, "image must be at least 8x8");
    VFE_REQUIRE(eig, "eig is NULL");
    DImg d = to_dimg(img);
    const EigGeom g = eig_geom(d.w, d.h, d.batch);
    const unsigned nblk = (g.total + 3) / 4;
    dim3 grid(nblk < GF_MAX_WGS ? nblk : GF_MAX_WGS);
    ProfScope prof(VFE_K_MISC, (hipStream_t)stream);
    hipLaunchKernelGGL((k_gftt_eig2<true, false>), grid, dim3(256), 0, (hipStream_t)stream, d, g.strips_x, g.band_h, g.bands_y,
                       g.total, eig, (u32*)nullptr, (u32*)nullptr, (u64*)nullptr, 0u, 0.0);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
