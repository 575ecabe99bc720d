/* Pyramidal Lucas-Kanade tracking (cv::calcOpticalFlowPyrLK semantics) for gfx950, batched.
 *
 * One 64-lane wavefront tracks one point through all pyramid levels.  Per level the wave
 * stages the (win+3)^2 source patch of the previous image in LDS (reflect-101 resolved on
 * load), derives the Scharr gradients of the patch in LDS instead of reading a derivative
 * image from HBM, builds the bilinear template (14-bit integer weights, as OpenCV) and the
 * 2x2 Gram matrix, then iterates: stage the (win+1)^2 patch of the next image, accumulate the
 * mismatch vector, solve, move.  All window sums are exact integers (per-lane int32 partials,
 * cross-lane butterfly in int64), so the float part of the algorithm sees the same inputs as
 * the CPU oracle and positions agree bit for bit.
 *
 * HBM traffic is only the patches actually touched; the pyramids themselves are shared with
 * the detector.
 */
#include "vfe_klt.h"
#include <float.h>

__device__ static inline long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(64) void k_klt(KltParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x;
    const int win = p.win, PW = win + 3, DW = win + 1;
    uint8_t* s_patch = smem;                                                    /* PW*PW bytes */
    short2* s_deriv = reinterpret_cast<short2*>(smem + ((PW * PW + 15) & ~15)); /* DW*DW */
    short* s_I = reinterpret_cast<short*>(s_deriv + DW * DW);                   /* win*win */
    short2* s_dI = reinterpret_cast<short2*>(s_I + ((win * win + 1) & ~1));     /* win*win */

    /* persistent single-wave workgroups striding over all (frame, point) items */
    for (unsigned work = blockIdx.x; work < p.total_work; work += gridDim.x) {
    const int b = work / (unsigned)p.cap, pt = work - (unsigned)b * (unsigned)p.cap;
    const size_t o = (size_t)b * p.cap + pt;
    const int n = p.n_pts ? min(p.n_pts[b], p.cap) : p.cap;
    if (pt >= n) {
        if (lane == 0) { p.status[o] = 0; if (p.err) p.err[o] = 0.f; }
        continue;
    }
    const float ppx = p.prev_pts[2 * o], ppy = p.prev_pts[2 * o + 1];
    float outx = p.use_initial ? p.next_pts[2 * o] : 0.f, outy = p.use_initial ? p.next_pts[2 * o + 1] : 0.f;
    int status = 1;
    float errv = 0.f;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int max_level = p.nlevels - 1;

    for (int level = max_level; level >= 0; level--) {
        const DImg& I = p.prev[level];
        const DImg& J = p.next[level];
        const uint8_t* ip = I.frame(b);
        const uint8_t* jp = J.frame(b);
        const float sc = (float)(1. / (double)(1 << level));
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

        __syncthreads();
        for (int i = lane; i < PW * PW; i += 64) {
            const int r = i / PW, c = i - r * PW;
            s_patch[i] = ip[(size_t)reflect101(ipy - 1 + r, I.h) * I.pitch + reflect101(ipx - 1 + c, I.w)];
        }
        __syncthreads();
        /* Scharr gradient at the (win+1)^2 positions the bilinear template touches; zero outside the image */
        for (int i = lane; i < DW * DW; i += 64) {
            const int r = i / DW, c = i - r * DW;
            const int gx = ipx + c, gy = ipy + r;
            short2 d = make_short2(0, 0);
            if (gx >= 0 && gy >= 0 && gx < I.w && gy < I.h) {
                const uint8_t* q = s_patch + r * PW + c;
                const int a00 = q[0], a01 = q[1], a02 = q[2];
                const int a10 = q[PW], a12 = q[PW + 2];
                const int a20 = q[2 * PW], a21 = q[2 * PW + 1], a22 = q[2 * PW + 2];
                d.x = (short)(3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20));
                d.y = (short)(3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02));
            }
            s_deriv[i] = d;
        }
        __syncthreads();
        int a11 = 0, a12 = 0, a22 = 0;
        for (int i = lane; i < win * win; i += 64) {
            const int y = i / win, x = i - y * win;
            const uint8_t* q = s_patch + (y + 1) * PW + (x + 1);
            const int ival = descale(q[0] * iw00 + q[1] * iw01 + q[PW] * iw10 + q[PW + 1] * iw11, 14 - 5);
            const short2 d00 = s_deriv[y * DW + x], d01 = s_deriv[y * DW + x + 1];
            const short2 d10 = s_deriv[(y + 1) * DW + x], d11 = s_deriv[(y + 1) * DW + x + 1];
            const int ixval = descale(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
            const int iyval = descale(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
            s_I[i] = (short)ival;
            s_dI[i] = make_short2((short)ixval, (short)iyval);
            a11 += ixval * ixval; a12 += ixval * iyval; a22 += iyval * iyval;
        }
        const long long iA11 = wave_sum_i64(a11), iA12 = wave_sum_i64(a12), iA22 = wave_sum_i64(a22);
        const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
        if ((double)minEig < p.min_eig_thr || D < FLT_EPSILON) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < p.max_iter; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            bilinear_weights(nx - (float)inx, ny - (float)iny, iw00, iw01, iw10, iw11);
            __syncthreads();
            for (int i = lane; i < DW * DW; i += 64) {
                const int r = i / DW, c = i - r * DW;
                s_patch[i] = jp[(size_t)reflect101(iny + r, J.h) * J.pitch + reflect101(inx + c, J.w)];
            }
            __syncthreads();
            int b1 = 0, b2 = 0;
            for (int i = lane; i < win * win; i += 64) {
                const int y = i / win, x = i - y * win;
                const uint8_t* q = s_patch + y * DW + x;
                const int diff = descale(q[0] * iw00 + q[1] * iw01 + q[DW] * iw10 + q[DW + 1] * iw11, 14 - 5) - s_I[i];
                const short2 d = s_dI[i];
                b1 += diff * d.x; b2 += diff * d.y;
            }
            const long long ib1 = wave_sum_i64(b1), ib2 = wave_sum_i64(b2);
            const float fb1 = (float)ib1 * FLT_SCALE, fb2 = (float)ib2 * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= p.epsilon) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {
            const float fx = outx - half, fy = outy - half;
            const int inx = (int)floorf(fx), iny = (int)floorf(fy);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { status = 0; continue; }
            bilinear_weights(fx - (float)inx, fy - (float)iny, iw00, iw01, iw10, iw11);
            __syncthreads();
            for (int i = lane; i < DW * DW; i += 64) {
                const int r = i / DW, c = i - r * DW;
                s_patch[i] = jp[(size_t)reflect101(iny + r, J.h) * J.pitch + reflect101(inx + c, J.w)];
            }
            __syncthreads();
            int es = 0;
            for (int i = lane; i < win * win; i += 64) {
                const int y = i / win, x = i - y * win;
                const uint8_t* q = s_patch + y * DW + x;
                const int diff = descale(q[0] * iw00 + q[1] * iw01 + q[DW] * iw10 + q[DW + 1] * iw11, 14 - 5) - s_I[i];
                es += abs(diff);
            }
            const long long esum = wave_sum_i64(es);
            errv = (float)esum * 1.f / (float)(32 * win * win);
        }
    }
    if (lane == 0) {
        p.next_pts[2 * o] = outx; p.next_pts[2 * o + 1] = outy;
        p.status[o] = (uint8_t)status;
        if (p.err) p.err[o] = errv;
    }
    }
}

#define ORD_CELLS 4096
__global__ __launch_bounds__(256) void k_klt_order(const float* __restrict__ pts, const int* __restrict__ n_pts, int cap, int w, int h,
                                                   int shift, int gx, int tiles, int nlevels, unsigned short* __restrict__ order) {
    __shared__ unsigned s_cnt[ORD_CELLS];
    __shared__ unsigned s_scan[2][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = n_pts ? max(0, min(n_pts[b], cap)) : cap;
    const float* p = pts + (size_t)b * cap * 2;
    unsigned short* o = order + (size_t)b * cap;
    const int cells = tiles * (nlevels + 1);
    for (int c = tid; c < cells; c += 256) s_cnt[c] = 0;
    __syncthreads();
    /* Sort key: first the coarsest pyramid level at which the point's 21 x 21 window still overhangs the level (a window
     * lies inside level l when the point keeps 11 * 2^l pixels from the rim; class nlevels = inside everywhere), then the
     * tile.  A wave of four points takes the tracker's costlier masked template at a level as soon as one of its windows
     * overhangs there; with the rim points of a frame next to each other only their waves do. */
    auto cell_of_point = [&](int i) {
        const int x = min(max((int)p[2 * i], 0), w - 1), y = min(max((int)p[2 * i + 1], 0), h - 1);
        const int rim = min(min(x, w - 1 - x), min(y, h - 1 - y));
        int cls = 0;
        while (cls < nlevels && rim >= (11 << cls)) cls++;
        return cls * tiles + (y >> shift) * gx + (x >> shift);
    };
    for (int i = tid; i < n; i += 256) atomicAdd(&s_cnt[cell_of_point(i)], 1u);
    __syncthreads();
    /* exclusive prefix over the tiles: thread t owns tiles 16t .. 16t+15 */
    unsigned loc[ORD_CELLS / 256], sum = 0;
#pragma unroll
    for (int k = 0; k < ORD_CELLS / 256; k++) {
        const int c = tid * (ORD_CELLS / 256) + k;
        loc[k] = c < cells ? s_cnt[c] : 0u;
        sum += loc[k];
    }
    int cur = 0;
    s_scan[0][tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned v = s_scan[cur][tid];
        if (tid >= off) v += s_scan[cur][tid - off];
        s_scan[cur ^ 1][tid] = v;
        cur ^= 1;
        __syncthreads();
    }
    unsigned run = s_scan[cur][tid] - sum;
#pragma unroll
    for (int k = 0; k < ORD_CELLS / 256; k++) {
        const int c = tid * (ORD_CELLS / 256) + k;
        if (c < cells) s_cnt[c] = run;
        run += loc[k];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) o[atomicAdd(&s_cnt[cell_of_point(i)], 1u)] = (unsigned short)i;
}

int klt_order_launch(const float* pts, const int* n_pts, int cap, int batch, int w, int h, int nlevels, unsigned short* order, hipStream_t stream) {
    VFE_REQUIRE(pts && order && cap > 0 && cap <= 65536 && batch > 0 && w > 0 && h > 0 && nlevels >= 1 && nlevels <= KLT_MAX_LEVELS,
                "bad arguments for the point order");
    int shift = 6; /* 64-pixel tiles, coarser for very large frames */
    while ((long long)((w >> shift) + 1) * ((h >> shift) + 1) * (nlevels + 1) > ORD_CELLS) shift++;
    const int gx = (w >> shift) + 1, tiles = gx * ((h >> shift) + 1);
    ProfScope prof(VFE_K_MISC, stream);
    hipLaunchKernelGGL(k_klt_order, dim3(batch), dim3(256), 0, stream, pts, n_pts, cap, w, h, shift, gx, tiles, nlevels, order);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_klt_track(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                 const float* prev_pts, const int* n_pts, int pts_cap,
                                 float* next_pts, uint8_t* status, float* err,
                                 int win, int max_iter, double eps, double min_eig_thr, int use_initial_flow,
                                 void* stream) {
    return klt_track_ordered(prev_levels, next_levels, nlevels, prev_pts, n_pts, pts_cap, next_pts, status, err, win, max_iter, eps,
                             min_eig_thr, use_initial_flow, nullptr, (hipStream_t)stream);
}

extern "C" size_t vfe_klt_track_workspace(int pts_cap, int batch) {
    if (pts_cap <= 0 || batch <= 0) return 0;
    return align_up((size_t)pts_cap * (size_t)batch * sizeof(unsigned short), 256);
}

extern "C" int vfe_dev_klt_track_ws(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                                    const float* prev_pts, const int* n_pts, int pts_cap,
                                    float* next_pts, uint8_t* status, float* err,
                                    int win, int max_iter, double eps, double min_eig_thr, int use_initial_flow,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    VFE_REQUIRE(prev_levels && next_levels && nlevels >= 1 && nlevels <= KLT_MAX_LEVELS, "nlevels must be 1..8");
    VFE_REQUIRE(prev_pts && pts_cap > 0, "NULL point buffers");
    const unsigned short* order = nullptr;
    if (win == 21 && pts_cap <= 65536 && workspace) {
        VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
        const int batch = prev_levels[0].batch;
        VFE_REQUIRE(batch > 0 && workspace_bytes >= vfe_klt_track_workspace(pts_cap, batch), "workspace too small");
        /* (the kernel for small launches has a workgroup per point: no order to choose) */
        if (!klt21_one_wave_per_point((unsigned)((size_t)pts_cap * (size_t)batch > 0xFFFFFFFFu ? 0xFFFFFFFFu : (size_t)pts_cap * (size_t)batch))) {
            int rc = klt_order_launch(prev_pts, n_pts, pts_cap, batch, prev_levels[0].width, prev_levels[0].height, nlevels,
                                      (unsigned short*)workspace, (hipStream_t)stream);
            if (rc) return rc;
            order = (const unsigned short*)workspace;
        }
    }
    return klt_track_ordered(prev_levels, next_levels, nlevels, prev_pts, n_pts, pts_cap, next_pts, status, err, win, max_iter, eps,
                             min_eig_thr, use_initial_flow, order, (hipStream_t)stream);
}

int klt_track_ordered(const vfe_dimg* prev_levels, const vfe_dimg* next_levels, int nlevels,
                      const float* prev_pts, const int* n_pts, int pts_cap, float* next_pts, uint8_t* status, float* err,
                      int win, int max_iter, double eps, double min_eig_thr, int use_initial_flow,
                      const unsigned short* order, hipStream_t stream) {
    VFE_REQUIRE(prev_levels && next_levels && nlevels >= 1 && nlevels <= KLT_MAX_LEVELS, "nlevels must be 1..8");
    VFE_REQUIRE(prev_pts && next_pts && status, "NULL point buffers");
    VFE_REQUIRE(pts_cap > 0, "pts_cap must be > 0");
    VFE_REQUIRE(win >= 3 && win <= 41 && (win & 1), "win must be odd, 3..41");
    VFE_REQUIRE(max_iter >= 1 && max_iter <= 1000, "max_iter out of range");
    KltParams p;
    for (int l = 0; l < nlevels; l++) {
        int rc;
        if ((rc = vfe_check_dimg(&prev_levels[l], "prev level")) || (rc = vfe_check_dimg(&next_levels[l], "next level"))) return rc;
        VFE_REQUIRE(prev_levels[l].width == next_levels[l].width && prev_levels[l].height == next_levels[l].height &&
                        prev_levels[l].batch == next_levels[l].batch && prev_levels[l].batch == prev_levels[0].batch,
                    "prev/next pyramids differ in shape");
        if (l > 0)
            VFE_REQUIRE(prev_levels[l].width == (prev_levels[l - 1].width + 1) / 2 &&
                            prev_levels[l].height == (prev_levels[l - 1].height + 1) / 2,
                        "level sizes must halve (rounded up)");
        p.prev[l] = to_dimg(&prev_levels[l]);
        p.next[l] = to_dimg(&next_levels[l]);
    }
    VFE_REQUIRE(prev_levels[0].batch <= 65535, "batch too large");
    p.nlevels = nlevels;
    p.prev_pts = prev_pts; p.n_pts = n_pts; p.cap = pts_cap;
    p.next_pts = next_pts; p.status = status; p.err = err;
    p.win = win; p.max_iter = max_iter; p.epsilon = eps * eps; p.min_eig_thr = min_eig_thr;
    p.use_initial = use_initial_flow ? 1 : 0;
    p.order = win == 21 ? order : nullptr; /* the generic kernel keeps the given order */
    const int PW = win + 3, DW = win + 1;
    const size_t smem = ((PW * PW + 15) & ~15) + (size_t)DW * DW * 4 + (size_t)((win * win + 1) & ~1) * 2 + (size_t)win * win * 4;
    ProfScope prof(VFE_K_KLT, (hipStream_t)stream);
    VFE_REQUIRE((long long)pts_cap * prev_levels[0].batch < (1ll << 31), "too many points");
    p.total_work = (unsigned)pts_cap * (unsigned)prev_levels[0].batch;
    if (win == 21) {
        klt21_launch(p, (hipStream_t)stream);
    } else {
        hipLaunchKernelGGL(k_klt, dim3(p.total_work < 8192u ? p.total_work : 8192u), dim3(64), smem, (hipStream_t)stream, p);
    }
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
