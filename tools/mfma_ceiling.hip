/* What the matrix pipe of gfx950 sustains on the instruction the matcher uses (v_mfma_scale_f32_32x32x64_f8f6f4 with
 * 4-bit operands), measured: every wave issues long streams of that instruction on CHAINS independent accumulators
 * (1 = every product waits for the one before it, 2 and 4 = two / four chains in turn), with 1, 2 or 4 waves per SIMD, on
 * the whole chip.  Timed with HIP events (what a launch gets) and, inside the kernel, with the shader clock against the
 * constant 100 MHz clock (the frequency the chip actually runs at under this load).  Prints products per second, the share
 * of the nominal peak (one 32-cycle product per SIMD at 2.4 GHz = 10.07 PFLOP/s), cycles per product per SIMD and the
 * measured clock.  The matcher's `roofline` is priced against the nominal peak; this table says how much of that peak any
 * kernel can get.  A second table issues 1 to 8 independent vector instructions between the products: the matrix pipe and
 * the vector ALU share the SIMD's issue port, and a kernel that looks at every product's results needs them.
 *
 * Usage: mfma_ceiling        (plain main(), exit code 1 on a HIP error) */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(call)                                                                                      \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CHAINS, int NV>
__global__ __launch_bounds__(256) void k_mfma(int trips, float* out, unsigned long long* clk) {
    v8i a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = i < 4 ? 0x22222222 : 0; b[i] = i < 4 ? (int)(0x2A2A2A2Au ^ (threadIdx.x * 0x01010101u & 0x88888888u)) : 0; }
    v16f acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    const int scale = 0x7f7f7f7f;
    float vr[8]; /* registers of the vector instructions issued between the products */
#pragma unroll
    for (int i = 0; i < 8; i++) vr[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int u = 0; u < 16 / CHAINS; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, scale, 0, scale);
#pragma unroll
                for (int v = 0; v < NV; v++) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(vr[v & 7]) : "v"(vr[(v + 1) & 7]), "v"(vr[(v + 2) & 7]));
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += vr[i];
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[c][r];
    if (s == 12345.678f) out[0] = s; /* keeps the products alive */
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

/* The matcher's inner loop in isolation: two accumulator sets in turn; the four products of a tile are issued, then the
 * 16 results of the previous tile are reduced to their maximum (MODE 0: one chain of v_max3 as the compiler writes it, 1: two
 * chains, 2: four chains) and compared with a threshold no lane passes (a branch that is never taken). */
template <int MODE>
__global__ __launch_bounds__(512) void k_tiles(int trips, float* out, unsigned long long* clk) {
    v8i a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = i < 4 ? 0x22222222 : 0; b[i] = i < 4 ? (int)(0x2A2A2A2Au ^ (threadIdx.x * 0x01010101u & 0x88888888u)) : 0; }
    v16f acc[2], cin;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[0][r] = acc[1][r] = 0.f; cin[r] = (float)r * 0.001f; }
    const int scale = 0x7f7f7f7f;
    float thr = 1e30f, hits = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            v16f& nw = acc[u & 1];
            const v16f& old = acc[(u & 1) ^ 1];
            nw = cin;
            asm volatile("" : "+v"(a), "+v"(b), "+v"(nw)); /* every tile is computed: nothing here is loop-invariant to the compiler */
#pragma unroll
            for (int q = 0; q < 4; q++) nw = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, nw, 4, 4, 0, scale, 0, scale);
            asm volatile("" : "+v"(nw));
            float m;
            if (MODE == 0) {
                m = old[0];
#pragma unroll
                for (int r = 1; r < 16; r++) m = __builtin_fmaxf(m, old[r]);
            } else if (MODE == 1) {
                float m0 = __builtin_fmaxf(old[0], old[1]), m1 = __builtin_fmaxf(old[8], old[9]);
#pragma unroll
                for (int r = 2; r < 8; r++) { m0 = __builtin_fmaxf(m0, old[r]); m1 = __builtin_fmaxf(m1, old[8 + r]); }
                m = __builtin_fmaxf(m0, m1);
            } else {
                float mm[4];
#pragma unroll
                for (int c = 0; c < 4; c++) mm[c] = __builtin_fmaxf(__builtin_fmaxf(old[4 * c], old[4 * c + 1]), __builtin_fmaxf(old[4 * c + 2], old[4 * c + 3]));
                m = __builtin_fmaxf(__builtin_fmaxf(mm[0], mm[1]), __builtin_fmaxf(mm[2], mm[3]));
            }
            if (__builtin_amdgcn_ballot_w64(m > thr) != 0) { hits += m; thr = m; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = hits;
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[0][r] + acc[1][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE>
static int run_tiles(int waves_per_simd, float* out, unsigned long long* clk) {
    const int trips = 4000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_tiles<MODE>, dim3(blocks), dim3(256), 0, 0, 100, out, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_tiles<MODE>, dim3(blocks), dim3(256), 0, 0, trips, out, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double products = (double)blocks * 4 * trips * 16, flop = products * 32.0 * 32 * 64 * 2;
    printf("tiles of four products + maximum of the previous tile's 16 results (%d chain%s) + test  waves/SIMD %d  %8.3f ms  %7.2f PFLOP/s  %5.1f %% of 10.07\n",
           MODE == 0 ? 1 : MODE == 1 ? 2 : 4, MODE == 0 ? "" : "s", waves_per_simd, ms, flop / (ms * 1e-3) / 1e15, 100.0 * flop / (ms * 1e-3) / 10.066e15);
    return 0;
}

template <int CHAINS, int NV>
static int run(int waves_per_simd, float* out, unsigned long long* clk) {
    const int trips = 4000; /* x 16 products per wave */
    const int blocks = 256 * waves_per_simd; /* 256 CUs, a workgroup of four waves per SIMD round */
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mfma<CHAINS, NV>), dim3(blocks), dim3(256), 0, 0, 100, out, clk); /* warm-up */
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mfma<CHAINS, NV>), dim3(blocks), dim3(256), 0, 0, trips, out, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CHECK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    const double products = (double)blocks * 4 * trips * 16, flop = products * 32.0 * 32 * 64 * 2;
    const double per_simd = products / 1024.0;
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    printf("chains %d  vector instructions per product %d  waves/SIMD %d  %8.3f ms  %7.2f PFLOP/s  %5.1f %% of 10.07  %6.1f cycles per product per SIMD (at the measured %.2f GHz)\n",
           CHAINS, NV, waves_per_simd, ms, flop / (ms * 1e-3) / 1e15, 100.0 * flop / (ms * 1e-3) / 10.066e15,
           ms * 1e-3 * ghz * 1e9 / per_simd, ghz);
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&clk, 64));
    for (int w : {1, 2, 4}) {
        if (run<1, 0>(w, out, clk) || run<2, 0>(w, out, clk) || run<4, 0>(w, out, clk)) return 1;
    }
    /* independent vector instructions (v_max3_f32) issued between the products: how many fit beside one */
    for (int w : {1, 2}) {
        if (run<2, 1>(w, out, clk) || run<2, 2>(w, out, clk) || run<2, 3>(w, out, clk) || run<2, 4>(w, out, clk) || run<2, 5>(w, out, clk) ||
            run<2, 6>(w, out, clk) || run<2, 8>(w, out, clk)) return 1;
    }
    for (int w : {1, 2, 3, 4}) {
        if (run_tiles<0>(w, out, clk) || run_tiles<1>(w, out, clk) || run_tiles<2>(w, out, clk)) return 1;
    }
    {   /* the same two waves per SIMD as ONE workgroup of eight waves per CU (the matcher's shape) */
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_tiles<0>, dim3(256), dim3(512), 0, 0, 100, out, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_tiles<0>, dim3(256), dim3(512), 0, 0, 4000, out, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 256.0 * 8 * 4000 * 16 * 32.0 * 32 * 64 * 2;
        printf("the same, one workgroup of eight waves per CU  %8.3f ms  %7.2f PFLOP/s  %5.1f %% of 10.07\n", ms, flop / (ms * 1e-3) / 1e15, 100.0 * flop / (ms * 1e-3) / 10.066e15);
    }
    return 0;
}
