/* What HBM delivers to plain streaming kernels on this chip, measured: the yardstick for the pyramid's 4:1 read/write
 * stream (a 5x5 decimation reads four bytes for every byte it writes) and for the other memory-shaped launches.
 *
 *   read      every lane sums 16-byte loads over a grid-stride walk (nothing written but one dword per wave)
 *   copy      16-byte load -> 16-byte store                                  (1 : 1)
 *   read4w1   four 16-byte loads (two adjacent rows, two adjacent columns) -> one 16-byte store     (4 : 1), plain and
 *             with the non-temporal hint on loads, stores or both
 *   rows4w1   the same 4 : 1 mix walked the way the pyramid walks a frame: a wave owns 1024 source columns and steps
 *             down the rows of a band, DEPTH rows requested ahead
 *
 * Each is timed with HIP events over REPS launches on 512 MiB of source; prints GB/s of bytes moved (reads + writes).
 *
 * Usage: hbm_stream            (plain main(), exit code 1 on a HIP error)
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(call)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const u4* __restrict__ src, size_t n, uint32_t* __restrict__ out) {
    u4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += __builtin_nontemporal_load(src + i);
    const uint32_t s = acc.x + acc.y + acc.z + acc.w;
    if (s == 0x12345678u) out[blockIdx.x] = s; /* never true for the fill used: keeps the loads alive */
}

__global__ __launch_bounds__(256) void k_copy(const u4* __restrict__ src, u4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

/* src seen as rows of `row` u4; output row r, column c from source rows 2r, 2r+1, columns 2c, 2c+1; NT: bit 0 loads,
 * bit 1 stores carry the non-temporal hint */
template <int NT>
__global__ __launch_bounds__(256) void k_read4w1(const u4* __restrict__ src, u4* __restrict__ dst, size_t row, size_t rows_out) {
    const size_t half = row / 2, n = half * rows_out, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const size_t r = i / half, c = i - r * half;
        const u4* p = src + 2 * r * row + 2 * c;
        u4 v;
        if (NT & 1) v = __builtin_nontemporal_load(p) + __builtin_nontemporal_load(p + 1) + __builtin_nontemporal_load(p + row) + __builtin_nontemporal_load(p + row + 1);
        else v = p[0] + p[1] + p[row] + p[row + 1];
        if (NT & 2) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

/* a wave owns 64 u4 columns (1 KiB) of a frame and a band of `band` source rows; two source rows -> one output row of
 * 32 u4 (the even lanes write) */
template <int DEPTH>
__global__ __launch_bounds__(256) void k_rows4w1(const u4* __restrict__ src, u4* __restrict__ dst, int row, int rows, int band, int frames) {
    const int wave = blockIdx.x * 4 + threadIdx.x / 64, lane = threadIdx.x & 63;
    const int strips = row / 64, bands = rows / band;
    const int f = wave / (strips * bands);
    if (f >= frames) return;
    const int rem = wave - f * strips * bands, bd = rem / strips, sp = rem - bd * strips;
    const u4* p = src + ((size_t)f * rows + (size_t)bd * band) * row + sp * 64 + lane;
    u4* q = dst + ((size_t)f * (rows / 2) + (size_t)bd * (band / 2)) * (row / 2) + sp * 32 + lane / 2;
    u4 ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) ring[d] = p[(size_t)d * row];
    for (int r = 0; r < band; r += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d += 2) {
            const u4 a = ring[d], b = ring[d + 1];
            if (r + DEPTH + d + 1 < band) {
                ring[d] = p[(size_t)(r + DEPTH + d) * row];
                ring[d + 1] = p[(size_t)(r + DEPTH + d + 1) * row];
            }
            u4 s = a + b;
            u4 o;
            o.x = s.x + __shfl_xor((int)s.x, 1);
            o.y = s.y + __shfl_xor((int)s.y, 1);
            o.z = s.z + __shfl_xor((int)s.z, 1);
            o.w = s.w + __shfl_xor((int)s.w, 1);
            if (!(lane & 1)) q[(size_t)((r + d) / 2) * (row / 2)] = o;
        }
    }
}

int main() {
    const size_t src_bytes = 512ull << 20, n = src_bytes / 16;
    const int REPS = 20;
    u4 *src, *dst;
    uint32_t* out;
    CHECK(hipMalloc(&src, src_bytes));
    CHECK(hipMalloc(&dst, src_bytes));
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemset(src, 1, src_bytes));
    CHECK(hipMemset(dst, 0, src_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto report = [&](const char* name, double bytes, float ms) {
        printf("%-28s %8.1f us per launch  %8.1f GB/s\n", name, 1e3 * ms / REPS, bytes * REPS / (ms * 1e-3) / 1e9);
    };
    float ms;
    const int grids[] = {1024, 2048, 4096, 8192, 16384};
#define TIME(NAME, BYTES, ...)                                      \
    do {                                                            \
        __VA_ARGS__; __VA_ARGS__;                                   \
        CHECK(hipEventRecord(e0, 0));                               \
        for (int i = 0; i < REPS; i++) { __VA_ARGS__; }             \
        CHECK(hipEventRecord(e1, 0));                               \
        CHECK(hipEventSynchronize(e1));                             \
        CHECK(hipGetLastError());                                   \
        CHECK(hipEventElapsedTime(&ms, e0, e1));                    \
        report(NAME, BYTES, ms);                                    \
    } while (0)
    char name[64];
    for (int g : grids) {
        snprintf(name, sizeof name, "read     grid %5d", g);
        TIME(name, (double)src_bytes, k_read<<<dim3(g), dim3(256), 0, 0>>>(src, n, out));
    }
    for (int g : grids) {
        snprintf(name, sizeof name, "copy     grid %5d", g);
        TIME(name, 2.0 * src_bytes, k_copy<<<dim3(g), dim3(256), 0, 0>>>(src, dst, n));
    }
    /* 256 frames of 1024 rows x 128 u4 (2 KiB rows): 512 MiB in, 128 MiB out */
    const int row = 128, rows = 1024, frames = 256;
    for (int g : grids) {
        snprintf(name, sizeof name, "read4w1  grid %5d", g);
        TIME(name, 1.25 * src_bytes, k_read4w1<0><<<dim3(g), dim3(256), 0, 0>>>(src, dst, (size_t)row, (size_t)rows * frames / 2));
        snprintf(name, sizeof name, "read4w1  grid %5d nt ld", g);
        TIME(name, 1.25 * src_bytes, k_read4w1<1><<<dim3(g), dim3(256), 0, 0>>>(src, dst, (size_t)row, (size_t)rows * frames / 2));
        snprintf(name, sizeof name, "read4w1  grid %5d nt st", g);
        TIME(name, 1.25 * src_bytes, k_read4w1<2><<<dim3(g), dim3(256), 0, 0>>>(src, dst, (size_t)row, (size_t)rows * frames / 2));
        snprintf(name, sizeof name, "read4w1  grid %5d nt both", g);
        TIME(name, 1.25 * src_bytes, k_read4w1<3><<<dim3(g), dim3(256), 0, 0>>>(src, dst, (size_t)row, (size_t)rows * frames / 2));
    }
    const int bands_list[] = {32, 64, 128, 256};
    for (int band : bands_list) {
        const int waves = (row / 64) * (rows / band) * frames;
        snprintf(name, sizeof name, "rows4w1  band %3d depth 4", band);
        TIME(name, 1.25 * src_bytes, k_rows4w1<4><<<dim3(waves / 4), dim3(256), 0, 0>>>(src, dst, row, rows, band, frames));
        snprintf(name, sizeof name, "rows4w1  band %3d depth 8", band);
        TIME(name, 1.25 * src_bytes, k_rows4w1<8><<<dim3(waves / 4), dim3(256), 0, 0>>>(src, dst, row, rows, band, frames));
        snprintf(name, sizeof name, "rows4w1  band %3d depth 16", band);
        TIME(name, 1.25 * src_bytes, k_rows4w1<16><<<dim3(waves / 4), dim3(256), 0, 0>>>(src, dst, row, rows, band, frames));
    }
    return 0;
}
