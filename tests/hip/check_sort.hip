/* sort1024_desc (vfe_sort.h) against std::sort: every size 64 .. 1024, random 64-bit keys, keys that differ in one half
 * only, runs of equal keys and zero padding.  Exit code 1 on a mismatch. */
#include "vfe_sort.h"
#include <algorithm>
#include <cstdio>
#include <functional>
#include <random>
#include <vector>

__global__ __launch_bounds__(1024) void k_sort(u64* data, u32 n2) {
    __shared__ u64 buf[2048];
    u64* d = data + (size_t)blockIdx.x * n2;
    if (threadIdx.x < n2) buf[threadIdx.x] = d[threadIdx.x];
    __syncthreads();
    const u64* sorted = sort1024_desc(buf, n2);
    if (threadIdx.x < n2) d[threadIdx.x] = sorted[threadIdx.x];
}

int main() {
    std::mt19937_64 rng(12345);
    int bad = 0, cases = 0;
    for (u32 n2 = 64; n2 <= 1024; n2 <<= 1) {
        const int blocks = 48;
        std::vector<u64> h((size_t)blocks * n2);
        for (int b = 0; b < blocks; b++)
            for (u32 i = 0; i < n2; i++) {
                u64 v = rng();
                switch (b % 6) {
                case 1: v &= 0xFFFFFFFFull; break;                 /* high halves equal */
                case 2: v &= 0xFFFFFFFF00000000ull; break;         /* low halves equal */
                case 3: v = v % 7; break;                          /* long runs of equal keys */
                case 4: if (i >= n2 / 2 + (u32)b) v = 0; break;    /* zero padding */
                case 5: v = ((u64)(0x3F000000u + (u32)(v % 1000)) << 32) | ((v >> 20) & 0x07FF07FFull); break; /* score | y | x */
                default: break;
                }
                h[(size_t)b * n2 + i] = v;
            }
        u64* d = nullptr;
        if (hipMalloc(&d, h.size() * sizeof(u64)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
        if (hipMemcpy(d, h.data(), h.size() * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess) { printf("copy failed\n"); return 1; }
        hipLaunchKernelGGL(k_sort, dim3(blocks), dim3(1024), 0, 0, d, n2);
        std::vector<u64> out(h.size());
        if (hipMemcpy(out.data(), d, h.size() * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
        (void)hipFree(d);
        for (int b = 0; b < blocks; b++) {
            std::sort(h.begin() + (size_t)b * n2, h.begin() + (size_t)(b + 1) * n2, std::greater<u64>());
            cases++;
            if (!std::equal(h.begin() + (size_t)b * n2, h.begin() + (size_t)(b + 1) * n2, out.begin() + (size_t)b * n2)) {
                if (bad < 5) printf("n2 %u block %d: mismatch\n", n2, b);
                bad++;
            }
        }
    }
    printf("sort1024_desc: %d of %d arrays wrong\n", bad, cases);
    return bad ? 1 : 0;
}
