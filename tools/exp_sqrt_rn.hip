/* Exhaustive check of the two correctly rounded square roots used by the eigenvalue kernel against sqrtf (which hipcc
 * rounds correctly by default): every float in [2^-100, 2^100] plus 0.
 *   hipcc --offload-arch=gfx950 -O3 -o tools/_build/exp_sqrt_rn tools/exp_sqrt_rn.hip && tools/_build/exp_sqrt_rn */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ static inline float sqrt_nb(float x) { /* hardware estimate + the two neighbour residuals */
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = r_dn <= 0.f ? s_dn : s;
    s = r_up > 0.f ? s_up : s;
    return s;
}

__device__ static inline f2 sqrt_gs(f2 x) { /* reciprocal square root, one coupled refinement, one residual step */
    const f2 r = {__builtin_amdgcn_rsqf(fmaxf(x.x, 1e-36f)), __builtin_amdgcn_rsqf(fmaxf(x.y, 1e-36f))};
    f2 s = x * r, h = r * 0.5f;
    const f2 e = __builtin_elementwise_fma(-h, s, f2{0.5f, 0.5f});
    h = __builtin_elementwise_fma(h, e, h);
    s = __builtin_elementwise_fma(s, e, s);
    const f2 d = __builtin_elementwise_fma(-s, s, x);
    return __builtin_elementwise_fma(d, h, s);
}

__global__ void k_check(unsigned lo, unsigned hi, unsigned long long* bad_nb, unsigned long long* bad_gs, unsigned* first_bad) {
    const unsigned long long n = (unsigned long long)hi - lo + 1ull;
    unsigned long long b1 = 0, b2 = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned u = lo + (unsigned)i;
        const float x = i == 0 ? 0.f : __uint_as_float(u);
        const float ref = sqrtf(x);
        const float a = sqrt_nb(x);
        const f2 g = sqrt_gs(f2{x, x});
        if (__float_as_uint(a) != __float_as_uint(ref)) b1++;
        if (__float_as_uint(g.x) != __float_as_uint(ref) || __float_as_uint(g.y) != __float_as_uint(ref)) { b2++; atomicMin(first_bad, u); }
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
    printf("values %llu  mismatches: neighbour-test %llu  rsq-refine %llu  first %08x\n", (unsigned long long)hi - lo + 1ull, h_bad[0], h_bad[1], h_first);
    return (h_bad[0] || h_bad[1]) ? 1 : 0;
}
