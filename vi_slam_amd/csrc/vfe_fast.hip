/* FAST-9/16 corner detection with 3x3 non-max suppression (cv::FAST semantics), batched,
 * raster-ordered output without atomics.
 *
 *  k_fast_tiles   64x16 tiles: the 72x24 source tile (radius-3 circle + 1 px for the NMS
 *                 neighbours) is staged in LDS; scores of the 66x18 positions are computed
 *                 once into LDS (contiguous-arc test on 16-bit brighter/darker masks, score
 *                 by a doubling min/max over the 16 arcs) and suppressed in place.  Each wave
 *                 owns one image row of the tile per step, so a ballot gives the row's
 *                 keypoints already in raster order; they go to a fixed slot of a per-tile
 *                 scratch area (at most 32 per 64-pixel row) together with the row count.
 *  k_fast_scan    one workgroup per image: exclusive prefix sum of the row counts in raster
 *                 order (row-major over image rows, then tiles left to right).
 *  k_fast_scatter copies each tile row to its final offset.
 * The same three kernels serve every ORB pyramid level.
 */
#include "vfe_common.h"
#include "vfe_fast.h"

#define FT_TW 64
#define FT_TH 16
#define FT_IW (FT_TW + 8)
#define FT_IH (FT_TH + 8)
#define FT_SW (FT_TW + 2)
#define FT_SH (FT_TH + 2)
#define FT_ROWCAP 32

__device__ static inline bool arc9(uint32_t m16) {
    uint32_t m = m16 | (m16 << 16);
    m &= m >> 1;          /* 2 contiguous */
    m &= m >> 2;          /* 4 */
    m &= m >> 4;          /* 8 */
    m &= (m16 | (m16 << 16)) >> 8; /* 9 */
    return (m & 0xFFFFu) != 0;
}

/* score of the pixel at s (LDS tile pointer, row stride FT_IW); 0 if not a corner */
__device__ static inline int fast_score(const uint8_t* s, int t) {
    const int v = s[0];
    int d[16];
    d[0] = v - s[3 * FT_IW];       d[1] = v - s[3 * FT_IW + 1];   d[2] = v - s[2 * FT_IW + 2];   d[3] = v - s[FT_IW + 3];
    d[4] = v - s[3];               d[5] = v - s[-FT_IW + 3];      d[6] = v - s[-2 * FT_IW + 2];  d[7] = v - s[-3 * FT_IW + 1];
    d[8] = v - s[-3 * FT_IW];      d[9] = v - s[-3 * FT_IW - 1];  d[10] = v - s[-2 * FT_IW - 2]; d[11] = v - s[-FT_IW - 3];
    d[12] = v - s[-3];             d[13] = v - s[FT_IW - 3];      d[14] = v - s[2 * FT_IW - 2];  d[15] = v - s[3 * FT_IW - 1];
    uint32_t dark = 0, bright = 0; /* circle darker than centre: d > t ; brighter: d < -t */
#pragma unroll
    for (int k = 0; k < 16; k++) {
        dark |= (uint32_t)(d[k] > t) << k;
        bright |= (uint32_t)(d[k] < -t) << k;
    }
    if (!arc9(dark) && !arc9(bright)) return 0;
    /* cornerScore: max(t, max over 9-arcs of min d, max over 9-arcs of min -d) - 1 */
    int mn2[16], mx2[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
    int mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
    int a0 = t, b0 = -t;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        a0 = max(a0, mn9);
        b0 = min(b0, mx9);
    }
    return max(a0, -b0) - 1;
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_fast_tiles(DImg img, int threshold, int border, int tiles_x,
                                                    unsigned tiles_per_frame, unsigned total_work,
                                                    uint8_t* __restrict__ rowcnt, uint32_t* __restrict__ tmp) {
    __shared__ __align__(16) uint32_t s_imgw[FT_IH][FT_IW / 4];
    __shared__ uint8_t s_sc[FT_SH][FT_SW + 2];
    __shared__ unsigned short s_queue[FT_SH * FT_SW];
    __shared__ unsigned s_nq;
    const uint8_t* s_img = reinterpret_cast<const uint8_t*>(&s_imgw[0][0]);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = img.w, H = img.h;
    for (unsigned work = blockIdx.x; work < total_work; work += gridDim.x) {
        const int b = work / tiles_per_frame, tile = work - b * tiles_per_frame;
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int ox = tx * FT_TW, oy = ty * FT_TH;
        const uint8_t* ip = img.frame(b);
        /* tile byte (r, c) <-> image (oy - 4 + r, ox - 4 + c); pixels outside the image are never
         * used by a scored position, any in-bounds value will do */
        for (int i = tid; i < FT_IH * (FT_IW / 4); i += 256) {
            const int r = i / (FT_IW / 4), k = i - r * (FT_IW / 4);
            const int gx = ox - 4 + 4 * k, gy = min(max(oy - 4 + r, 0), H - 1);
            const uint8_t* row = ip + (size_t)gy * img.pitch;
            uint32_t v;
            if (ALIGNED && gx >= 0 && gx + 3 < W)
                v = *reinterpret_cast<const uint32_t*>(row + gx);
            else
                v = (uint32_t)row[min(max(gx, 0), W - 1)] | ((uint32_t)row[min(max(gx + 1, 0), W - 1)] << 8) |
                    ((uint32_t)row[min(max(gx + 2, 0), W - 1)] << 16) | ((uint32_t)row[min(max(gx + 3, 0), W - 1)] << 24);
            s_imgw[r][k] = v;
        }
        __syncthreads();
        /* Pass 1, every position: the cheap necessary condition.  A 9-arc of the 16-pixel circle
         * contains at least two of the four compass pixels, so a corner has two of them brighter
         * than centre + t or two darker than centre - t.  Positions that pass are queued in LDS
         * (one counter update per wave); the rest score 0.
         * Pass 2, queued positions only, one per thread: the full arc test and score.  Few positions
         * survive pass 1, so the expensive code runs densely packed instead of once per pixel. */
        if (tid == 0) s_nq = 0;
        __syncthreads();
        for (int i0 = 0; i0 < FT_SH * FT_SW; i0 += 256) {
            const int i = i0 + tid;
            bool pass = false;
            if (i < FT_SH * FT_SW) {
                const int sy = i / FT_SW, sx = i - sy * FT_SW;
                const int px = ox - 1 + sx, py = oy - 1 + sy;
                if (px >= 3 && px < W - 3 && py >= 3 && py < H - 3) {
                    const uint8_t* c = s_img + (sy + 3) * FT_IW + (sx + 3);
                    const int v = c[0], hi = v + threshold, lo = v - threshold;
                    const int p0 = c[3 * FT_IW], p4 = c[3], p8 = c[-3 * FT_IW], p12 = c[-3];
                    const int nb = (p0 > hi) + (p4 > hi) + (p8 > hi) + (p12 > hi);
                    const int nd = (p0 < lo) + (p4 < lo) + (p8 < lo) + (p12 < lo);
                    pass = nb >= 2 || nd >= 2;
                }
                if (!pass) s_sc[sy][sx] = 0;
            }
            const unsigned long long m = __ballot(pass);
            if (m) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_nq, (unsigned)__popcll(m));
                base = (unsigned)__shfl((int)base, 0);
                if (pass) s_queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
            }
        }
        __syncthreads();
        const int nq = (int)s_nq;
        for (int j = tid; j < nq; j += 256) {
            const int i = s_queue[j];
            const int sy = i / FT_SW, sx = i - sy * FT_SW;
            s_sc[sy][sx] = (uint8_t)fast_score(s_img + (sy + 3) * FT_IW + (sx + 3), threshold);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FT_TH / 4; k++) {
            const int y = 4 * k + wv, x = lane;
            const int px = ox + x, py = oy + y;
            const int s = s_sc[y + 1][x + 1];
            bool kp = false;
            if (s > 0 && px >= border && px < W - border && py >= border && py < H - border) {
                kp = s > s_sc[y][x] && s > s_sc[y][x + 1] && s > s_sc[y][x + 2] && s > s_sc[y + 1][x] &&
                     s > s_sc[y + 1][x + 2] && s > s_sc[y + 2][x] && s > s_sc[y + 2][x + 1] && s > s_sc[y + 2][x + 2];
            }
            const unsigned long long m = __ballot(kp);
            if (py < H) {
                if (kp) {
                    const int pos = __popcll(m & ((1ull << lane) - 1ull));
                    tmp[((size_t)work * FT_TH + y) * FT_ROWCAP + pos] = (uint32_t)x | ((uint32_t)s << 8);
                }
                if (lane == 0) rowcnt[((size_t)b * H + py) * tiles_x + tx] = (uint8_t)__popcll(m);
            }
        }
        __syncthreads();
    }
}

/* exclusive prefix of rowcnt over (row, tile) in raster order; one workgroup per image */
__global__ __launch_bounds__(1024) void k_fast_scan(const uint8_t* __restrict__ rowcnt, uint32_t* __restrict__ rowoff,
                                                    int n_per_frame, int* __restrict__ counts, int cap) {
    __shared__ uint32_t s_buf[2][1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t* rc = rowcnt + (size_t)b * n_per_frame;
    uint32_t* ro = rowoff + (size_t)b * n_per_frame;
    const int per = (n_per_frame + 1023) / 1024;
    const int i0 = tid * per, i1 = min(i0 + per, n_per_frame);
    uint32_t sum = 0;
    for (int i = i0; i < i1; i++) sum += rc[i];
    int cur = 0;
    s_buf[0][tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        uint32_t x = s_buf[cur][tid];
        if (tid >= o) x += s_buf[cur][tid - o];
        s_buf[cur ^ 1][tid] = x;
        cur ^= 1;
        __syncthreads();
    }
    uint32_t run = s_buf[cur][tid] - sum;
    for (int i = i0; i < i1; i++) { ro[i] = run; run += rc[i]; }
    if (tid == 1023 && counts) counts[b] = (int)min(s_buf[cur][1023], (uint32_t)cap);
}

__global__ __launch_bounds__(256) void k_fast_scatter(int W, int H, int tiles_x, unsigned tiles_per_frame, unsigned total_work,
                                                      const uint8_t* __restrict__ rowcnt, const uint32_t* __restrict__ rowoff,
                                                      const uint32_t* __restrict__ tmp, int* __restrict__ kps, int cap) {
    const int tid = threadIdx.x;
    for (unsigned work = blockIdx.x; work < total_work; work += gridDim.x) {
        const int b = work / tiles_per_frame, tile = work - b * tiles_per_frame;
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        /* 16 rows x up to 32 entries = 512 slots for 256 threads */
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int slot = tid + k * 256;
            const int y = slot / FT_ROWCAP, j = slot - y * FT_ROWCAP;
            const int py = ty * FT_TH + y;
            if (py >= H) continue;
            const size_t ri = ((size_t)b * H + py) * tiles_x + tx;
            if (j < rowcnt[ri]) {
                const uint32_t e = tmp[((size_t)work * FT_TH + y) * FT_ROWCAP + j];
                const uint32_t off = rowoff[ri] + j;
                if (off < (uint32_t)cap) {
                    int* o = kps + ((size_t)b * cap + off) * 3;
                    o[0] = tx * FT_TW + (int)(e & 255); o[1] = py; o[2] = (int)(e >> 8);
                }
            }
        }
    }
}

FastWs fast_layout(int w, int h, int batch) {
    FastWs L;
    L.tiles_x = div_up(w, FT_TW); L.tiles_y = div_up(h, FT_TH);
    size_t o = 0;
    L.rowcnt_off = o; o += align_up((size_t)batch * h * L.tiles_x, 256);
    L.rowoff_off = o; o += align_up((size_t)batch * h * L.tiles_x * sizeof(uint32_t), 256);
    L.tmp_off = o; o += align_up((size_t)batch * L.tiles_x * L.tiles_y * FT_TH * FT_ROWCAP * sizeof(uint32_t), 256);
    {   /* whole frames go through the streaming kernels, which lay the workspace out themselves */
        const size_t stream = fast_set_layout(1, &w, &h, batch).total;
        if (stream > o) o = stream;
    }
    L.total = o;
    return L;
}

int fast_detect(const DImg& d, int threshold, int border, int* kps, int cap, int* counts, char* ws, hipStream_t st) {
    const FastWs L = fast_layout(d.w, d.h, d.batch);
    const unsigned tpf = (unsigned)(L.tiles_x * L.tiles_y), total = tpf * (unsigned)d.batch;
    uint8_t* rowcnt = (uint8_t*)(ws + L.rowcnt_off);
    uint32_t* rowoff = (uint32_t*)(ws + L.rowoff_off);
    uint32_t* tmp = (uint32_t*)(ws + L.tmp_off);
    const bool aligned = ((uintptr_t)d.data % 4 == 0) && (d.pitch % 4 == 0) && (d.fstride % 4 == 0) &&
                         (d.sub_n == 1 || (d.sub_dx % 4 == 0 && d.sub_dy % 4 == 0));
    dim3 grid(total < 2048u ? total : 2048u);
    ProfScope prof(VFE_K_FAST, st);
    if (d.sub_n == 1 && d.w >= 16 && d.h >= 16) { /* whole frames: streaming kernels */
        FastSet S;
        S.n = 1; S.img[0] = fast_img(d); S.kps[0] = kps; S.counts[0] = counts; S.cap[0] = cap;
        return fast_detect_set(S, d.batch, threshold, border, ws, st);
    }
    if (aligned)
        hipLaunchKernelGGL(k_fast_tiles<true>, grid, dim3(256), 0, st, d, threshold, border, L.tiles_x, tpf, total, rowcnt, tmp);
    else
        hipLaunchKernelGGL(k_fast_tiles<false>, grid, dim3(256), 0, st, d, threshold, border, L.tiles_x, tpf, total, rowcnt, tmp);
    VFE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fast_scan, dim3(d.batch), dim3(1024), 0, st, rowcnt, rowoff, d.h * L.tiles_x, counts, cap);
    VFE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fast_scatter, grid, dim3(256), 0, st, d.w, d.h, L.tiles_x, tpf, total, rowcnt, rowoff, tmp, kps, cap);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" size_t vfe_fast_workspace(int width, int height, int batch) {
    if (width < 7 || height < 7 || batch <= 0) return 0;
    return fast_layout(width, height, batch).total;
}

extern "C" int vfe_dev_fast9(const vfe_dimg* img, int threshold, int* kps, int cap, int* counts,
                             void* workspace, size_t workspace_bytes, void* stream) {
    int rc = vfe_check_dimg(img, "img");
    if (rc) return rc;
    VFE_REQUIRE(img->width >= 7 && img->height >= 7, "image must be at least 7x7");
    VFE_REQUIRE(threshold >= 1 && threshold <= 254, "threshold must be 1..254");
    VFE_REQUIRE(kps && counts && workspace && cap > 0, "NULL output or workspace");
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    const FastWs L = fast_layout(img->width, img->height, img->batch);
    if (workspace_bytes < L.total) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
        return VFE_E_NOMEM;
    }
    return fast_detect(to_dimg(img), threshold, 0, kps, cap, counts, (char*)workspace, (hipStream_t)stream);
}


/* ---- grid-bucketed FAST (OpenVINS-style) --------------------------------------------------
 * FAST runs on every grid cell as an image of its own (3-pixel rim and NMS per cell), the best
 * `per_cell` of each cell by score (ties: raster order) are offered, cell after cell, to an
 * occupancy grid of min_px_dist pixels that existing tracks have marked first. */
__global__ __launch_bounds__(256) void k_fast_grid_select(const int* __restrict__ kps, const int* __restrict__ counts, int ccap,
                                                          int ncells, int grid_x, int size_x, int size_y, int num_features,
                                                          int min_px_dist, int ow, int oh, uint8_t* __restrict__ occ_all,
                                                          const float* __restrict__ seeds, const int* __restrict__ n_seeds,
                                                          int seed_cap, float* __restrict__ pts, int* __restrict__ scores,
                                                          int cap, int* __restrict__ out_counts) {
    __shared__ unsigned long long s_red[4];
    __shared__ int s_n;
    const int b = blockIdx.x, tid = threadIdx.x;
    uint8_t* occ = occ_all + (size_t)b * ow * oh;
    const int ns = (seeds && n_seeds) ? max(0, min(n_seeds[b], seed_cap)) : 0;
    const int needed = num_features - ns;
    if (tid == 0) s_n = 0;
    if (needed <= 0) { if (tid == 0) out_counts[b] = 0; return; }
    const int per_cell = (int)((double)needed / (double)ncells) + 1;
    for (int s = tid; s < ns; s += 256) {
        const int gx = (int)(seeds[((size_t)b * seed_cap + s) * 2] / (float)min_px_dist);
        const int gy = (int)(seeds[((size_t)b * seed_cap + s) * 2 + 1] / (float)min_px_dist);
        if (gx >= 0 && gy >= 0 && gx < ow && gy < oh) occ[gy * ow + gx] = 1;
    }
    __syncthreads();
    for (int r = 0; r < ncells; r++) {
        const int cb = b * ncells + r;
        const int n = min(counts[cb], ccap);
        const int* k = kps + (size_t)cb * ccap * 3;
        const int x0 = (r % grid_x) * size_x, y0 = (r / grid_x) * size_y;
        unsigned long long last = ~0ull;
        for (int it = 0; it < per_cell && it < n; it++) {
            /* next key below `last`: score descending, then list (raster) order */
            unsigned long long best = 0;
            for (int i = tid; i < n; i += 256) {
                const unsigned long long key = ((unsigned long long)(unsigned)k[3 * i + 2] << 32) | (unsigned)(0x7fffffff - i);
                if (key < last && key > best) best = key;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(best, o); best = v > best ? v : best; }
            if ((tid & 63) == 0) s_red[tid >> 6] = best;
            __syncthreads();
            best = s_red[0];
#pragma unroll
            for (int q = 1; q < 4; q++) best = s_red[q] > best ? s_red[q] : best;
            if (tid == 0 && best) {
                const int i = 0x7fffffff - (int)(unsigned)(best & 0xffffffffu);
                const float x = (float)(k[3 * i] + x0), y = (float)(k[3 * i + 1] + y0);
                const int gx = (int)(x / (float)min_px_dist), gy = (int)(y / (float)min_px_dist);
                if (gx >= 0 && gy >= 0 && gx < ow && gy < oh && !occ[gy * ow + gx]) {
                    occ[gy * ow + gx] = 1;
                    const int o = s_n;
                    if (o < cap) {
                        pts[((size_t)b * cap + o) * 2] = x; pts[((size_t)b * cap + o) * 2 + 1] = y;
                        scores[(size_t)b * cap + o] = k[3 * i + 2];
                    }
                    s_n = o + 1;
                }
            }
            last = best;
            __syncthreads();
            if (!best) break;
        }
    }
    if (tid == 0) out_counts[b] = min(s_n, cap);
}

struct FastGridWs { int size_x, size_y, ccap, ow, oh; size_t kps_off, cnt_off, occ_off, fast_off, total; };

static FastGridWs fast_grid_layout(int w, int h, int batch, int grid_x, int grid_y, int min_px_dist) {
    FastGridWs L;
    L.size_x = w / grid_x; L.size_y = h / grid_y;
    L.ccap = (L.size_x * L.size_y) / 4 + 64;
    L.ow = (int)((float)w / (float)min_px_dist); L.oh = (int)((float)h / (float)min_px_dist);
    const int nc = grid_x * grid_y;
    size_t o = 0;
    L.kps_off = o; o += align_up((size_t)batch * nc * L.ccap * 3 * sizeof(int), 256);
    L.cnt_off = o; o += align_up((size_t)batch * nc * sizeof(int), 256);
    L.occ_off = o; o += align_up((size_t)batch * (size_t)(L.ow > 0 ? L.ow : 1) * (size_t)(L.oh > 0 ? L.oh : 1), 256);
    L.fast_off = o; o += (L.size_x >= 7 && L.size_y >= 7) ? fast_layout(L.size_x, L.size_y, batch * nc).total : 0;
    L.total = o;
    return L;
}

extern "C" size_t vfe_fast_grid_workspace(int width, int height, int batch, int grid_x, int grid_y, int min_px_dist) {
    if (width <= 0 || height <= 0 || batch <= 0 || grid_x <= 0 || grid_y <= 0 || min_px_dist <= 0) return 0;
    return fast_grid_layout(width, height, batch, grid_x, grid_y, min_px_dist).total;
}

extern "C" int vfe_dev_fast_grid(const vfe_dimg* img, int threshold, int grid_x, int grid_y, int num_features,
                                 int min_px_dist, const float* seeds_xy, const int* n_seeds, int seed_cap,
                                 float* pts_xy, int* scores, int cap, int* counts,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    int rc = vfe_check_dimg(img, "img");
    if (rc) return rc;
    VFE_REQUIRE(threshold >= 1 && threshold <= 254, "threshold must be 1..254");
    VFE_REQUIRE(grid_x > 0 && grid_y > 0 && grid_x * grid_y <= 4096 && num_features > 0 && min_px_dist > 0, "bad grid parameters");
    VFE_REQUIRE(pts_xy && scores && counts && workspace && cap > 0, "NULL output or workspace");
    VFE_REQUIRE(!seeds_xy == !n_seeds, "seeds_xy and n_seeds go together");
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const FastGridWs L = fast_grid_layout(img->width, img->height, img->batch, grid_x, grid_y, min_px_dist);
    if (workspace_bytes < L.total) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
        return VFE_E_NOMEM;
    }
    if (L.size_x < 7 || L.size_y < 7 || L.ow <= 0 || L.oh <= 0) {
        VFE_HIP(hipMemsetAsync(counts, 0, (size_t)img->batch * sizeof(int), st));
        return VFE_OK;
    }
    char* ws = (char*)workspace;
    const int nc = grid_x * grid_y;
    DImg cells = to_dimg(img);
    cells.w = L.size_x; cells.h = L.size_y; cells.batch = img->batch * nc;
    cells.sub_n = nc; cells.sub_gx = grid_x; cells.sub_dx = L.size_x; cells.sub_dy = (size_t)L.size_y * img->pitch;
    int* ckps = (int*)(ws + L.kps_off);
    int* ccnt = (int*)(ws + L.cnt_off);
    if ((rc = fast_detect(cells, threshold, 0, ckps, L.ccap, ccnt, ws + L.fast_off, st))) return rc;
    VFE_HIP(hipMemsetAsync(ws + L.occ_off, 0, (size_t)img->batch * L.ow * L.oh, st));
    ProfScope prof(VFE_K_FAST, st);
    hipLaunchKernelGGL(k_fast_grid_select, dim3(img->batch), dim3(256), 0, st, ckps, ccnt, L.ccap, nc, grid_x, L.size_x, L.size_y,
                       num_features, min_px_dist, L.ow, L.oh, (uint8_t*)(ws + L.occ_off), seeds_xy, n_seeds,
                       seeds_xy ? seed_cap : 0, pts_xy, scores, cap, counts);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
