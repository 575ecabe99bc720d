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
