/* Exhaustive check of the eigenvalue kernel's packed square root (vfe_sqrt_rn.h), and of the neighbour-test form it
 * replaced, against sqrtf (which hipcc rounds correctly by default): 0 and every float in [2^-100, 2^100].
 * Built by `make` into tests/cpp/bin/check_sqrt_rn; exit code 0 = no mismatch. */
#include <hip/hip_runtime.h>
#include "vfe_sqrt_rn.h"
#include <cstdio>
#include <cmath>


__device__ static inline float sqrt_nb(float x) { /* hardware estimate + the two neighbour residuals */
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = r_dn <= 0.f ? s_dn : s;
    s = r_up > 0.f ? s_up : s;
    return s;
}

__global__ void k_check(unsigned lo, unsigned hi, unsigned long long* bad_nb, unsigned long long* bad_gs, unsigned* first_bad) {
    const unsigned long long n = (unsigned long long)hi - lo + 1ull;
    unsigned long long b1 = 0, b2 = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned u = lo + (unsigned)i;
        const float x = i == 0 ? 0.f : __uint_as_float(u);
        const float ref = sqrtf(x);
        const float a = sqrt_nb(x);
        const f2 g = sqrt_rn2(f2{x, i + 1 < n ? __uint_as_float(u + 1u) : x});
        if (__float_as_uint(a) != __float_as_uint(ref)) b1++;
        const float ref_y = i + 1 < n ? sqrtf(__uint_as_float(u + 1u)) : ref; /* the two halves carry different values */
        if (__float_as_uint(g.x) != __float_as_uint(ref) || __float_as_uint(g.y) != __float_as_uint(ref_y)) { b2++; atomicMin(first_bad, u); }
    }
    if (b1) atomicAdd(bad_nb, b1);
    if (b2) atomicAdd(bad_gs, b2);
}

int main() {
    unsigned long long *d_bad, h_bad[2] = {0, 0};
    unsigned* d_first, h_first = 0xffffffffu;
    if (hipMalloc(&d_bad, 16) != hipSuccess || hipMalloc(&d_first, 4) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMemcpy(d_bad, h_bad, 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_first, &h_first, 4, hipMemcpyHostToDevice);
    const unsigned lo = (127u - 100u) << 23, hi = ((127u + 100u) << 23) - 1u; /* element 0 of the range stands for x = 0 */
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, lo, hi, d_bad, d_bad + 1, d_first);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    (void)hipMemcpy(h_bad, d_bad, 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&h_first, d_first, 4, hipMemcpyDeviceToHost);
    printf("values %llu  mismatches: neighbour-test %llu  packed rsq refinement %llu  first %08x\n", (unsigned long long)hi - lo + 1ull, h_bad[0], h_bad[1], h_first);
    return (h_bad[0] || h_bad[1]) ? 1 : 0;
}
