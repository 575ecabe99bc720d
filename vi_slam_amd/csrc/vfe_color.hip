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
