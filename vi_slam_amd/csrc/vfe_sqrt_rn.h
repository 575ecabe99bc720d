/* Correctly rounded square root of two floats in packed fp32 arithmetic (shared by the eigenvalue kernel and its
 * exhaustive check). */
#pragma once
#include <hip/hip_runtime.h>

typedef float f2 __attribute__((ext_vector_type(2)));

/* sqrtf of two values, correctly rounded, for x = 0 or x well inside the normal range (the sums of squares formed here
 * are 0 or > 1e-29): hardware reciprocal square root, one coupled refinement of root and half-reciprocal, one residual
 * step, all but the estimate as packed operations.  This is the compiler's own correctly rounded expansion for
 * flushed-denormal mode; checked against sqrtf over every float in [2^-100, 2^100] (tests/hip/check_sqrt_rn.hip).  The
 * estimate's argument is kept away from 0 so that x = 0 runs through as 0 rather than 0 * inf: by adding 1e-37, which
 * leaves every x > 1e-29 as it is (a plain add issues in half the time of a maximum). */
__device__ static inline f2 sqrt_rn2(f2 x) {
    const f2 r = {__builtin_amdgcn_rsqf(x.x + 1e-37f), __builtin_amdgcn_rsqf(x.y + 1e-37f)};
    f2 s = x * r, h = r * 0.5f;
    const f2 e = __builtin_elementwise_fma(-h, s, f2{0.5f, 0.5f});
    h = __builtin_elementwise_fma(h, e, h);
    s = __builtin_elementwise_fma(s, e, s);
    const f2 d = __builtin_elementwise_fma(-s, s, x);
    return __builtin_elementwise_fma(d, h, s);
}
