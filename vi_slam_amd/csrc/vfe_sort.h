/* Bitonic sort of up to 1024 64-bit keys, descending, by a workgroup of 1024 threads that holds one key per thread in
 * registers.  Compare-exchange steps whose partner lies within the wavefront (distance 1 .. 32) exchange registers
 * across lanes -- DPP for 1, 2, 4 and 8, the row and half swaps of gfx950 for 16 and 32 -- and touch no memory; only the
 * steps at distance 64 and more go through LDS.  512 keys: 39 register steps and 6 LDS steps instead of 45 LDS steps
 * of two dependent round trips each. */
#pragma once
#include "vfe_gftt.h"
#include <type_traits>

/* the value of lane (l ^ J) of the wavefront, J a power of two below 64 */
template <int J>
__device__ static inline u32 lane_xor(u32 v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "partner within the wavefront");
    if constexpr (J == 1) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); /* quad_perm [1,0,3,2] */
    if constexpr (J == 2) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); /* quad_perm [2,3,0,1] */
    if constexpr (J == 4) {
        /* lanes 0-3 and 8-11 of a row take the lane four up (row_shl 4), the others the lane four down (row_shr 4) */
        const int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xf, 0x5, false);
        return (u32)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xf, 0xA, false);
    }
    if constexpr (J == 8) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); /* row_ror 8 */
    if constexpr (J == 16) {
        /* with both operands v: first result = rows (0, 0, 2, 2) of v, second = rows (1, 1, 3, 3) */
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? r[0] : r[1];
    }
    if constexpr (J == 32) {
        /* first result = lower half of v twice, second = upper half twice */
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
    return v;
}

/* buf[0 .. n2) -> sorted descending; n2 a power of two, 64 <= n2 <= 1024; every thread of the 1024 calls it (uniformly).
 * buf is LDS with room for 2048 keys: the steps through LDS alternate between its two halves (one barrier per step), and
 * the sorted keys end up in the half that is returned (ends with a barrier). */
struct Sort1024 {
    u64 key;
    u64 *cur, *alt; /* the half the next LDS step writes and reads, and the one the step before it used */
    u32 i;
    bool on;
    /* keep the larger (want_max) or the smaller of the own key and the partner's: no branch, the lanes of a wave differ */
    __device__ inline void keep(u64 other, bool want_max) {
        key = (other > key) == want_max ? other : key; /* (equal keys: either) */
    }
    template <int K, int J>
    __device__ inline void reg_step() {
        if constexpr (J >= 1 && J < K) {
            keep((u64)lane_xor<J>((u32)key) | ((u64)lane_xor<J>((u32)(key >> 32)) << 32), ((i & J) == 0) == ((i & K) == 0));
            reg_step<K, J / 2>();
        }
    }
    template <int K, int J>
    __device__ inline void lds_step() {
        if constexpr (J >= 64) {
            if (on) cur[i] = key; /* nobody reads this half any more: its readers passed the barrier of the step before */
            __syncthreads();
            if (on) keep(cur[i ^ J], ((i & J) == 0) == ((i & K) == 0));
            u64* t = cur; cur = alt; alt = t;
            lds_step<K, J / 2>();
        }
    }
    template <int K>
    __device__ inline void stages(u32 n2) {
        if constexpr (K <= 1024) {
            if (K > n2) return;
            lds_step<K, K / 2>();
            if (on) reg_step<K, (K / 2 < 32 ? K / 2 : 32)>();
            stages<K * 2>(n2);
        }
    }
};
__device__ static inline const u64* sort1024_desc(u64* buf, u32 n2) {
    Sort1024 s;
    s.i = threadIdx.x;
    s.on = s.i < n2; /* whole wavefronts: n2 is a multiple of 64 */
    s.key = s.on ? buf[s.i] : 0ull;
    s.cur = buf + 1024; s.alt = buf; /* the first LDS step must not write where keys may still be waiting to be read */
    s.stages<2>(n2);
    if (s.on) s.cur[s.i] = s.key;
    __syncthreads();
    return s.cur;
}
