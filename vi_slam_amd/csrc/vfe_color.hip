/* Colour / intensity pre-processing in front of the detectors. */
#include "vfe_common.h"

/* cv::cvtColor(COLOR_BGR2GRAY), 8-bit: 15-bit fixed-point BT.601 weights, one rounding.
 * Each thread converts 4 pixels (12 source bytes -> one dword store). */
__global__ __launch_bounds__(256) void k_bgr2gray(DImg bgr, DImg gray) {
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, b = blockIdx.z;
    if (x4 >= gray.w) return;
    const uint8_t* s = bgr.frame(b) + (size_t)y * bgr.pitch + 3 * x4;
    uint8_t* d = gray.frame(b) + (size_t)y * gray.pitch + x4;
    const int n = min(4, gray.w - x4);
    uint32_t packed = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t g = (s[3 * i] * 3735u + s[3 * i + 1] * 19235u + s[3 * i + 2] * 9798u + (1u << 14)) >> 15;
        packed |= g << (8 * i);
    }
    if (n == 4 && ((uintptr_t)d & 3) == 0) *reinterpret_cast<uint32_t*>(d) = packed;
    else for (int i = 0; i < n; i++) d[i] = (uint8_t)(packed >> (8 * i));
}

extern "C" int vfe_dev_bgr2gray(const vfe_dimg* bgr, const vfe_dimg* gray, void* stream) {
    VFE_REQUIRE(bgr && gray && bgr->data && gray->data, "NULL image");
    VFE_REQUIRE(bgr->width == gray->width && bgr->height == gray->height && bgr->batch == gray->batch, "shape mismatch");
    VFE_REQUIRE(bgr->width > 0 && bgr->height > 0 && bgr->batch > 0 && bgr->height <= 65535 && bgr->batch <= 65535, "bad sizes");
    VFE_REQUIRE(bgr->pitch >= 3 * bgr->width && gray->pitch >= gray->width, "pitch too small");
    ProfScope prof(VFE_K_MISC, (hipStream_t)stream);
    hipLaunchKernelGGL(k_bgr2gray, dim3(div_up(div_up(gray->width, 4), 256), gray->height, gray->batch), dim3(256), 0,
                       (hipStream_t)stream, to_dimg(bgr), to_dimg(gray));
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

/* ---- cv::equalizeHist, 8-bit (config/vins_mono.yaml:104 `equalize`) ------------------------------------
 * Pass 1: per-frame histogram; every wave counts into its own LDS copy (fewer same-address
 * collisions), the workgroup adds its total to the frame's 256 global bins.
 * Pass 2: every workgroup rebuilds the frame's lookup table from the histogram (a 256-element scan,
 * cheaper than another launch) and maps its rows through it:
 *   lut[v] = saturate(round(cdf'(v) * 255 / (total - hist[first non-empty bin]))), cdf' counted from the
 *   bin after the first non-empty one; a constant image maps to itself. */
#define EQ_ROWS 16 /* image rows per workgroup */

__global__ __launch_bounds__(256) void k_eq_hist(DImg img, unsigned* __restrict__ hist) {
    __shared__ unsigned s_h[4][256];
    const int tid = threadIdx.x, wv = tid >> 6, b = blockIdx.y;
    for (int i = tid; i < 4 * 256; i += 256) (&s_h[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* ip = img.frame(b);
    const int y0 = blockIdx.x * EQ_ROWS, y1 = min(y0 + EQ_ROWS, img.h);
    const bool aligned = ((uintptr_t)ip % 4 == 0) && (img.pitch % 4 == 0);
    const int wq = aligned ? img.w / 4 : 0; /* whole dwords per row */
    for (int y = y0; y < y1; y++) {
        const uint8_t* row = ip + (size_t)y * img.pitch;
        for (int q = tid; q < wq; q += 256) {
            const uint32_t v = reinterpret_cast<const uint32_t*>(row)[q];
            atomicAdd(&s_h[wv][v & 255u], 1u); atomicAdd(&s_h[wv][(v >> 8) & 255u], 1u);
            atomicAdd(&s_h[wv][(v >> 16) & 255u], 1u); atomicAdd(&s_h[wv][v >> 24], 1u);
        }
        for (int x = 4 * wq + tid; x < img.w; x += 256) atomicAdd(&s_h[wv][row[x]], 1u);
    }
    __syncthreads();
    const unsigned n = s_h[0][tid] + s_h[1][tid] + s_h[2][tid] + s_h[3][tid];
    if (n) atomicAdd(&hist[(size_t)b * 256 + tid], n);
}

__global__ __launch_bounds__(256) void k_eq_apply(DImg src, DImg dst, const unsigned* __restrict__ hist) {
    __shared__ unsigned s_scan[2][256];
    __shared__ uint8_t s_lut[256];
    __shared__ int s_first;
    const int tid = threadIdx.x, b = blockIdx.y;
    const unsigned h = hist[(size_t)b * 256 + tid];
    if (tid == 0) s_first = 256;
    __syncthreads();
    if (h) atomicMin(&s_first, tid);
    /* inclusive prefix sum of the bins */
    int cur = 0;
    s_scan[0][tid] = h;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        unsigned v = s_scan[cur][tid];
        if (tid >= o) v += s_scan[cur][tid - o];
        s_scan[cur ^ 1][tid] = v;
        cur ^= 1;
        __syncthreads();
    }
    const int first = s_first;
    const unsigned total = (unsigned)src.w * (unsigned)src.h;
    const unsigned hfirst = s_scan[cur][first] - (first ? s_scan[cur][first - 1] : 0u);
    uint8_t lv;
    if (hfirst == total) lv = (uint8_t)first;
    else if (tid <= first) lv = 0;
    else {
        const float scale = (256 - 1.f) / (float)(int)(total - hfirst);
        const int sum = (int)(s_scan[cur][tid] - s_scan[cur][first]);
        const int v = __float2int_rn((float)sum * scale);
        lv = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    s_lut[tid] = lv;
    __syncthreads();
    const uint8_t* ip = src.frame(b);
    uint8_t* op = dst.frame(b);
    const int y0 = blockIdx.x * EQ_ROWS, y1 = min(y0 + EQ_ROWS, src.h);
    const bool aligned = ((uintptr_t)ip % 4 == 0) && (src.pitch % 4 == 0) && ((uintptr_t)op % 4 == 0) && (dst.pitch % 4 == 0);
    const int wq = aligned ? src.w / 4 : 0;
    for (int y = y0; y < y1; y++) {
        const uint8_t* row = ip + (size_t)y * src.pitch;
        uint8_t* orow = op + (size_t)y * dst.pitch;
        for (int q = tid; q < wq; q += 256) {
            const uint32_t v = reinterpret_cast<const uint32_t*>(row)[q];
            reinterpret_cast<uint32_t*>(orow)[q] = (uint32_t)s_lut[v & 255u] | ((uint32_t)s_lut[(v >> 8) & 255u] << 8) |
                                                   ((uint32_t)s_lut[(v >> 16) & 255u] << 16) | ((uint32_t)s_lut[v >> 24] << 24);
        }
        for (int x = 4 * wq + tid; x < src.w; x += 256) orow[x] = s_lut[row[x]];
    }
}

extern "C" size_t vfe_equalize_workspace(int batch) {
    if (batch <= 0) return 0;
    return align_up((size_t)batch * 256 * sizeof(unsigned), 256);
}

extern "C" int vfe_dev_equalize_hist(const vfe_dimg* src, const vfe_dimg* dst, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    int rc;
    if ((rc = vfe_check_dimg(src, "src")) || (rc = vfe_check_dimg(dst, "dst"))) return rc;
    VFE_REQUIRE(src->width == dst->width && src->height == dst->height && src->batch == dst->batch, "shape mismatch");
    VFE_REQUIRE(src->batch <= 65535, "batch too large");
    VFE_REQUIRE((unsigned long long)src->width * src->height < (1ull << 31), "image too large");
    const size_t need = vfe_equalize_workspace(src->batch);
    if (!workspace || workspace_bytes < need) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
        return VFE_E_NOMEM;
    }
    hipStream_t st = (hipStream_t)stream;
    VFE_HIP(hipMemsetAsync(workspace, 0, need, st));
    ProfScope prof(VFE_K_MISC, st);
    const dim3 grid(div_up(src->height, EQ_ROWS), src->batch);
    hipLaunchKernelGGL(k_eq_hist, grid, dim3(256), 0, st, to_dimg(src), (unsigned*)workspace);
    VFE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_eq_apply, grid, dim3(256), 0, st, to_dimg(src), to_dimg(dst), (const unsigned*)workspace);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

/* ---- CLAHE ------------------------------------------------------------------------------------
 * cv::createCLAHE(clip, Size(tiles_x, tiles_y))->apply on 8-bit gray frames.  Two kernels: one workgroup
 * per (tile, frame) builds the tile's clipped histogram and its LUT; then every pixel blends the LUT
 * entries of the four surrounding tiles in float (same operation order as the CPU restatement). */
struct ClaheGeom {
    int tiles_x, tiles_y, tw, th; /* tile size on the (possibly extended) image */
    int clip_limit;               /* 0 = no clipping */
    float lut_scale, inv_tw, inv_th;
};

__global__ __launch_bounds__(256) void k_clahe_lut(DImg img, ClaheGeom g, uint8_t* __restrict__ lut) {
    __shared__ unsigned s_h[4][256];
    __shared__ unsigned s_scan[2][256];
    const int tid = threadIdx.x, wv = tid >> 6, b = blockIdx.y;
    const int tile = blockIdx.x, ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    for (int i = tid; i < 4 * 256; i += 256) (&s_h[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* ip = img.frame(b);
    /* tile pixels; positions past the image are its reflect-101 extension */
    const int area = g.tw * g.th;
    for (int i = tid; i < area; i += 256) {
        const int yy = i / g.tw, xx = i - yy * g.tw;
        const int y = reflect101(ty * g.th + yy, img.h), x = reflect101(tx * g.tw + xx, img.w);
        atomicAdd(&s_h[wv][ip[(size_t)y * img.pitch + x]], 1u);
    }
    __syncthreads();
    int h = (int)(s_h[0][tid] + s_h[1][tid] + s_h[2][tid] + s_h[3][tid]);
    if (g.clip_limit > 0) {
        const int excess = h > g.clip_limit ? h - g.clip_limit : 0;
        h -= excess;
        /* total excess */
        s_scan[0][tid] = (unsigned)excess;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_scan[0][tid] += s_scan[0][tid + o]; __syncthreads(); }
        const int clipped = (int)s_scan[0][0];
        __syncthreads();
        const int share = clipped / 256, residual = clipped - share * 256;
        h += share;
        if (residual != 0) {
            const int rstep = max(256 / residual, 1);
            if (tid % rstep == 0 && tid / rstep < residual) h++;
        }
    }
    /* inclusive prefix sum -> LUT */
    int cur = 0;
    s_scan[0][tid] = (unsigned)h;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        unsigned v = s_scan[cur][tid];
        if (tid >= o) v += s_scan[cur][tid - o];
        s_scan[cur ^ 1][tid] = v;
        cur ^= 1;
        __syncthreads();
    }
    const int v = __float2int_rn((float)(int)s_scan[cur][tid] * g.lut_scale);
    lut[((size_t)b * g.tiles_x * g.tiles_y + tile) * 256 + tid] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

#define CL_ROWS 16 /* image rows per workgroup of the blending kernel */

__global__ __launch_bounds__(256) void k_clahe_apply(DImg src, DImg dst, ClaheGeom g, const uint8_t* __restrict__ lut, int max_tile_rows) {
    extern __shared__ uint8_t s_lut[]; /* LUTs of the tile rows this band touches: [tile row][tiles_x][256] */
    const int tid = threadIdx.x, b = blockIdx.y;
    const int y0 = blockIdx.x * CL_ROWS, y1 = min(y0 + CL_ROWS, src.h);
    const int t_lo = max((int)floorf((float)y0 * g.inv_th - 0.5f), 0);
    const int t_hi = min((int)floorf((float)(y1 - 1) * g.inv_th - 0.5f) + 1, g.tiles_y - 1);
    const int n_tr = min(t_hi - t_lo + 1, max_tile_rows);
    {
        const uint32_t* gl = reinterpret_cast<const uint32_t*>(lut + ((size_t)b * g.tiles_y + t_lo) * g.tiles_x * 256);
        uint32_t* sl = reinterpret_cast<uint32_t*>(s_lut);
        for (int i = tid; i < n_tr * g.tiles_x * 64; i += 256) sl[i] = gl[i];
    }
    __syncthreads();
    const uint8_t* ip = src.frame(b);
    uint8_t* op = dst.frame(b);
    for (int y = y0; y < y1; y++) {
        const float tyf = (float)y * g.inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        ty1 = max(ty1, 0); ty2 = min(ty2, g.tiles_y - 1);
        const uint8_t* P1 = s_lut + (size_t)(ty1 - t_lo) * g.tiles_x * 256;
        const uint8_t* P2 = s_lut + (size_t)(ty2 - t_lo) * g.tiles_x * 256;
        const uint8_t* row = ip + (size_t)y * src.pitch;
        uint8_t* orow = op + (size_t)y * dst.pitch;
        for (int x = tid; x < src.w; x += 256) {
            const float txf = (float)x * g.inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            tx1 = max(tx1, 0); tx2 = min(tx2, g.tiles_x - 1);
            const int v = row[x];
            const float p11 = (float)P1[tx1 * 256 + v], p12 = (float)P1[tx2 * 256 + v];
            const float p21 = (float)P2[tx1 * 256 + v], p22 = (float)P2[tx2 * 256 + v];
            const float res = (p11 * xa1 + p12 * xa) * ya1 + (p21 * xa1 + p22 * xa) * ya;
            const int r = __float2int_rn(res);
            orow[x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
}

static int clahe_geom(int w, int h, double clip, int tiles_x, int tiles_y, ClaheGeom* g, int* max_tile_rows) {
    VFE_REQUIRE(tiles_x >= 1 && tiles_y >= 1 && tiles_x <= 64 && tiles_y <= 64, "CLAHE: 1..64 tiles per dimension");
    int ew = w, eh = h;
    if (w % tiles_x != 0 || h % tiles_y != 0) { ew = w + tiles_x - (w % tiles_x); eh = h + tiles_y - (h % tiles_y); }
    VFE_REQUIRE(ew - w < w && eh - h < h, "CLAHE: image too small for this tile grid");
    g->tiles_x = tiles_x; g->tiles_y = tiles_y; g->tw = ew / tiles_x; g->th = eh / tiles_y;
    const int area = g->tw * g->th;
    g->clip_limit = 0;
    if (clip > 0.0) { g->clip_limit = (int)(clip * area / 256); if (g->clip_limit < 1) g->clip_limit = 1; }
    g->lut_scale = (float)(256 - 1) / (float)area;
    g->inv_tw = 1.0f / (float)g->tw; g->inv_th = 1.0f / (float)g->th;
    /* tile rows a band of CL_ROWS image rows can touch */
    *max_tile_rows = (CL_ROWS + g->th - 1) / g->th + 2;
    if (*max_tile_rows > tiles_y) *max_tile_rows = tiles_y;
    VFE_REQUIRE((size_t)*max_tile_rows * tiles_x * 256 <= 48 * 1024, "CLAHE: tiles too small for this grid width");
    return VFE_OK;
}

extern "C" size_t vfe_clahe_workspace(int batch, int tiles_x, int tiles_y) {
    if (batch <= 0 || tiles_x < 1 || tiles_y < 1 || tiles_x > 64 || tiles_y > 64) return 0;
    return align_up((size_t)batch * tiles_x * tiles_y * 256, 256);
}

extern "C" int vfe_dev_clahe(const vfe_dimg* src, const vfe_dimg* dst, double clip_limit, int tiles_x, int tiles_y,
                             void* workspace, size_t workspace_bytes, void* stream) {
    int rc;
    if ((rc = vfe_check_dimg(src, "src")) || (rc = vfe_check_dimg(dst, "dst"))) return rc;
    VFE_REQUIRE(src->width == dst->width && src->height == dst->height && src->batch == dst->batch, "shape mismatch");
    VFE_REQUIRE(src->batch <= 65535, "batch too large");
    ClaheGeom g;
    int max_tile_rows = 0;
    if ((rc = clahe_geom(src->width, src->height, clip_limit, tiles_x, tiles_y, &g, &max_tile_rows))) return rc;
    const size_t need = vfe_clahe_workspace(src->batch, tiles_x, tiles_y);
    if (!workspace || workspace_bytes < need) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
        return VFE_E_NOMEM;
    }
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(VFE_K_MISC, st);
    hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y, src->batch), dim3(256), 0, st, to_dimg(src), g, (uint8_t*)workspace);
    VFE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_clahe_apply, dim3(div_up(src->height, CL_ROWS), src->batch), dim3(256),
                       (size_t)max_tile_rows * tiles_x * 256, st, to_dimg(src), to_dimg(dst), g, (const uint8_t*)workspace, max_tile_rows);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
