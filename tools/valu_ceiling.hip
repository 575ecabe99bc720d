/* VALU issue ceiling of gfx950, measured: long streams of independent instructions of one class, issued by
 * 1, 2, 3, 4, 6 and 8 waves per SIMD, timed with the shader clock inside the kernel (one workgroup, so one CU) and with
 * HIP events over a launch that fills the chip (4 workgroups per CU).  Prints wave-instructions per cycle per SIMD and
 * its inverse, the cycles one wave64 instruction occupies the SIMD's issue port.  The hot kernels of the library are
 * priced against these numbers (VALU-bound kernels: instructions x cycles-per-instruction / SIMD-cycles available).
 *
 * Every class runs UNROLL x 16 instructions per loop trip on 16 independent registers (dependency distance 16), so the
 * stream is issue-limited, not latency-limited, from one wave per SIMD upwards unless the table shows otherwise.
 *
 * Usage: valu_ceiling [--csv]          (plain main(), exit code 1 on a HIP error)
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CHECK(call)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

#define UNROLL 4
#define PER_TRIP (UNROLL * 16)

/* one instruction on register i of the 16-entry file r[] (32-bit classes) */
#define OP32(ASM)                                                                     \
    _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                              \
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(r[i]) : "v"(c0), "v"(c1)); \
    }
/* 64-bit register pairs (packed fp32) */
#define OP64(ASM)                                                                     \
    _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                              \
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(q[i]) : "v"(d0), "v"(d1)); \
    }

enum Cls {
    C_ADD_F32, C_FMA_F32, C_MAX3_F32, C_ADD_U32, C_XOR, C_AND_OR, C_LSHL, C_MIN_U32, C_MED3_U32, C_BCNT, C_CNDMASK,
    C_MAD_U24, C_MUL_LO, C_PERM, C_PK_ADD_I16, C_PK_MAX_I16, C_PK_MUL_LO_U16, C_PK_MAD_I16, C_DOT2_I32_I16, C_DOT4_U32_U8,
    C_PK_FMA_F32, C_PK_MUL_F32, C_PK_ADD_F32, C_MOV_DPP_ROW, C_MOV_DPP_WAVE, C_ADD_DPP_ROW, C_RSQ, C_SQRT, C_RCP, C_CVT_F32_I32,
    C_MOV, C_MUL_F32, C_SUB_F32, C_MAX_F32, C_MIN_F32, C_AND, C_OR, C_LSHL_V, C_LSHR, C_ASHR, C_BFE, C_CVT_UB0, C_CVT_UB3,
    C_CNDMASK_S, C_SUB_U32, C_ADD3, C_LSHL_ADD, C_LSHL_OR, C_MUL_U24, C_ALIGNBIT, C_PK_SUB_I16, C_PK_LSHR_B16, C_SAD_U8, C_FMAC_F32,
    C_MAD_I24, C_MAX_U32, C_CVT_U32_F32, C_RNDNE, C_CVT_I32_F32, C_MAX_I32, C_ADD_I16, C_MAD_U16,
    /* 64-bit integer address arithmetic and double precision (register pairs) */
    C_MAD_U64_U32, C_MAD_I64_I32, C_LSHL_ADD_U64, C_CVT_F64_I32, C_CVT_F32_F64, C_CVT_F64_F32, C_FMA_F64, C_MUL_F64, C_ADD_F64,
    C_PK_MOV, C_MOV_B64,
    /* sub-dword addressing: a shift whose 16-bit result lands in the upper half of the destination, the lower half kept */
    C_LSHR_SDWA_W1, C_ADD_SDWA_W1SRC, C_COUNT
};
static const char* kNames[C_COUNT] = {
    "v_add_f32", "v_fma_f32", "v_max3_f32", "v_add_u32", "v_xor_b32", "v_and_or_b32", "v_lshlrev_b32", "v_min_u32", "v_med3_u32",
    "v_bcnt_u32_b32", "v_cndmask_b32", "v_mad_u32_u24", "v_mul_lo_u32", "v_perm_b32", "v_pk_add_i16", "v_pk_max_i16",
    "v_pk_mul_lo_u16", "v_pk_mad_i16", "v_dot2_i32_i16", "v_dot4_u32_u8", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
    "v_mov_b32 dpp row_shr:1", "v_mov_b32 dpp wave_shr:1", "v_add_u32 dpp row_shr:1", "v_rsq_f32", "v_sqrt_f32", "v_rcp_f32",
    "v_cvt_f32_i32", "v_mov_b32", "v_mul_f32", "v_sub_f32", "v_max_f32", "v_min_f32", "v_and_b32", "v_or_b32", "v_lshlrev_b32 (vgpr shift)",
    "v_lshrrev_b32", "v_ashrrev_i32", "v_bfe_u32", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte3", "v_cndmask_b32 (sgpr mask)", "v_sub_u32",
    "v_add3_u32", "v_lshl_add_u32", "v_lshl_or_b32", "v_mul_u32_u24", "v_alignbit_b32", "v_pk_sub_i16", "v_pk_lshrrev_b16", "v_sad_u8",
    "v_fmac_f32", "v_mad_i32_i24", "v_max_u32", "v_cvt_u32_f32", "v_rndne_f32", "v_cvt_i32_f32", "v_max_i32", "v_add_u16", "v_mad_u16",
    "v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64", "v_cvt_f64_i32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_fma_f64", "v_mul_f64", "v_add_f64",
    "v_pk_mov_b32", "v_mov_b64", "v_lshrrev_b32_sdwa dst_sel:WORD_1 preserve", "v_add_u32_sdwa src1_sel:WORD_1"};

template <int CLS>
__global__ __launch_bounds__(1024) void k_issue(long long* __restrict__ stamps, uint32_t* __restrict__ sink, int trips, uint32_t seed) {
    uint32_t r[16];
    unsigned long long q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        r[i] = seed * (i + 1) + threadIdx.x;
        q[i] = ((unsigned long long)__float_as_uint(1.0f + 1e-6f * i) << 32) | __float_as_uint(1.0f - 1e-6f * i);
    }
    uint32_t c0 = seed | 0x00010001u, c1 = (seed >> 3) | 0x01000100u;
    unsigned long long d0 = ((unsigned long long)__float_as_uint(0.999999f) << 32) | __float_as_uint(1.000001f), d1 = d0;
    const unsigned long long smask = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) == (seed & 1)); /* a wave-uniform SGPR-pair mask */
    __syncthreads();
    const long long t0 = clock64();
    for (int t = 0; t < trips; t++) {
        if (CLS == C_ADD_F32) { OP32("v_add_f32 %0, %0, %1") }
        else if (CLS == C_FMA_F32) { OP32("v_fma_f32 %0, %0, %1, %2") }
        else if (CLS == C_MAX3_F32) { OP32("v_max3_f32 %0, %0, %1, %2") }
        else if (CLS == C_ADD_U32) { OP32("v_add_u32 %0, %0, %1") }
        else if (CLS == C_XOR) { OP32("v_xor_b32 %0, %0, %1") }
        else if (CLS == C_AND_OR) { OP32("v_and_or_b32 %0, %0, %1, %2") }
        else if (CLS == C_LSHL) { OP32("v_lshlrev_b32 %0, 1, %0") }
        else if (CLS == C_MIN_U32) { OP32("v_min_u32 %0, %0, %1") }
        else if (CLS == C_MED3_U32) { OP32("v_med3_u32 %0, %0, %1, %2") }
        else if (CLS == C_BCNT) { OP32("v_bcnt_u32_b32 %0, %1, %0") }
        else if (CLS == C_CNDMASK) { OP32("v_cndmask_b32 %0, %0, %1, vcc") }
        else if (CLS == C_MAD_U24) { OP32("v_mad_u32_u24 %0, %0, %1, %2") }
        else if (CLS == C_MUL_LO) { OP32("v_mul_lo_u32 %0, %0, %1") }
        else if (CLS == C_PERM) { OP32("v_perm_b32 %0, %0, %1, %2") }
        else if (CLS == C_PK_ADD_I16) { OP32("v_pk_add_i16 %0, %0, %1") }
        else if (CLS == C_PK_MAX_I16) { OP32("v_pk_max_i16 %0, %0, %1") }
        else if (CLS == C_PK_MUL_LO_U16) { OP32("v_pk_mul_lo_u16 %0, %0, %1") }
        else if (CLS == C_PK_MAD_I16) { OP32("v_pk_mad_i16 %0, %0, %1, %2") }
        else if (CLS == C_DOT2_I32_I16) { OP32("v_dot2_i32_i16 %0, %1, %2, %0") }
        else if (CLS == C_DOT4_U32_U8) { OP32("v_dot4_u32_u8 %0, %1, %2, %0") }
        else if (CLS == C_PK_FMA_F32) { OP64("v_pk_fma_f32 %0, %0, %1, %2") }
        else if (CLS == C_PK_MUL_F32) { OP64("v_pk_mul_f32 %0, %0, %1") }
        else if (CLS == C_PK_ADD_F32) { OP64("v_pk_add_f32 %0, %0, %1") }
        else if (CLS == C_MOV_DPP_ROW) { OP32("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") }
        else if (CLS == C_MOV_DPP_WAVE) { OP32("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") }
        else if (CLS == C_ADD_DPP_ROW) { OP32("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") }
        else if (CLS == C_RSQ) { OP32("v_rsq_f32 %0, %0") }
        else if (CLS == C_SQRT) { OP32("v_sqrt_f32 %0, %0") }
        else if (CLS == C_RCP) { OP32("v_rcp_f32 %0, %0") }
        else if (CLS == C_CVT_F32_I32) { OP32("v_cvt_f32_i32 %0, %0") }
        else if (CLS == C_MOV) { OP32("v_mov_b32 %0, %1") }
        else if (CLS == C_MUL_F32) { OP32("v_mul_f32 %0, %0, %1") }
        else if (CLS == C_SUB_F32) { OP32("v_sub_f32 %0, %0, %1") }
        else if (CLS == C_MAX_F32) { OP32("v_max_f32 %0, %0, %1") }
        else if (CLS == C_MIN_F32) { OP32("v_min_f32 %0, %0, %1") }
        else if (CLS == C_AND) { OP32("v_and_b32 %0, %0, %1") }
        else if (CLS == C_OR) { OP32("v_or_b32 %0, %0, %1") }
        else if (CLS == C_LSHL_V) { OP32("v_lshlrev_b32 %0, %1, %0") }
        else if (CLS == C_LSHR) { OP32("v_lshrrev_b32 %0, 3, %0") }
        else if (CLS == C_ASHR) { OP32("v_ashrrev_i32 %0, 3, %0") }
        else if (CLS == C_BFE) { OP32("v_bfe_u32 %0, %0, 8, 8") }
        else if (CLS == C_CVT_UB0) { OP32("v_cvt_f32_ubyte0 %0, %0") }
        else if (CLS == C_CVT_UB3) { OP32("v_cvt_f32_ubyte3 %0, %0") }
        else if (CLS == C_CNDMASK_S) {
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {
                _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c0), "s"(smask));
            }
        }
        else if (CLS == C_SUB_U32) { OP32("v_sub_u32 %0, %0, %1") }
        else if (CLS == C_ADD3) { OP32("v_add3_u32 %0, %0, %1, %2") }
        else if (CLS == C_LSHL_ADD) { OP32("v_lshl_add_u32 %0, %0, 2, %1") }
        else if (CLS == C_LSHL_OR) { OP32("v_lshl_or_b32 %0, %0, 2, %1") }
        else if (CLS == C_MUL_U24) { OP32("v_mul_u32_u24 %0, %0, %1") }
        else if (CLS == C_ALIGNBIT) { OP32("v_alignbit_b32 %0, %0, %1, 16") }
        else if (CLS == C_PK_SUB_I16) { OP32("v_pk_sub_i16 %0, %0, %1") }
        else if (CLS == C_PK_LSHR_B16) { OP32("v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]") }
        else if (CLS == C_SAD_U8) { OP32("v_sad_u8 %0, %0, %1, %2") }
        else if (CLS == C_FMAC_F32) { OP32("v_fmac_f32 %0, %1, %2") }
        else if (CLS == C_MAD_I24) { OP32("v_mad_i32_i24 %0, %0, %1, %2") }
        else if (CLS == C_MAX_U32) { OP32("v_max_u32 %0, %0, %1") }
        else if (CLS == C_CVT_U32_F32) { OP32("v_cvt_u32_f32 %0, %0") }
        else if (CLS == C_RNDNE) { OP32("v_rndne_f32 %0, %0") }
        else if (CLS == C_CVT_I32_F32) { OP32("v_cvt_i32_f32 %0, %0") }
        else if (CLS == C_MAX_I32) { OP32("v_max_i32 %0, %0, %1") }
        else if (CLS == C_ADD_I16) { OP32("v_add_u16 %0, %0, %1") }
        else if (CLS == C_MAD_U16) { OP32("v_mad_u16 %0, %0, %1, %2") }
        else if (CLS == C_MAD_U64_U32) {
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {
                _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(c0), "v"(c1) : "vcc");
            }
        }
        else if (CLS == C_MAD_I64_I32) {
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {
                _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(c0), "v"(c1) : "vcc");
            }
        }
        else if (CLS == C_LSHL_ADD_U64) { OP64("v_lshl_add_u64 %0, %0, 0, %1") }
        else if (CLS == C_CVT_F64_I32) {
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {
                _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(q[i]) : "v"(r[i]));
            }
        }
        else if (CLS == C_CVT_F32_F64) {
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {
                _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(q[i]));
            }
        }
        else if (CLS == C_CVT_F64_F32) {
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {
                _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(q[i]) : "v"(r[i]));
            }
        }
        else if (CLS == C_FMA_F64) { OP64("v_fma_f64 %0, %0, %1, %2") }
        else if (CLS == C_MUL_F64) { OP64("v_mul_f64 %0, %0, %1") }
        else if (CLS == C_ADD_F64) { OP64("v_add_f64 %0, %0, %1") }
        else if (CLS == C_PK_MOV) { OP64("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]") }
        else if (CLS == C_MOV_B64) { OP64("v_mov_b64 %0, %1") }
        else if (CLS == C_LSHR_SDWA_W1) { OP32("v_lshrrev_b32_sdwa %0, 9, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD") }
        else if (CLS == C_ADD_SDWA_W1SRC) { OP32("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") }
    }
    const long long t1 = clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= r[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    if (acc == 0x12345u) sink[0] = acc; /* keeps the registers live */
    if ((threadIdx.x & 63) == 0) {
        const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * wave] = t0;
        stamps[2 * wave + 1] = t1;
    }
}

typedef void (*kern_t)(long long*, uint32_t*, int, uint32_t);
template <int C> struct Fill {
    static void go(kern_t* t) { t[C] = k_issue<C>; Fill<C + 1>::go(t); }
};
template <> struct Fill<C_COUNT> { static void go(kern_t*) {} };

int main(int argc, char** argv) {
    const bool csv = argc > 1 && !strcmp(argv[1], "--csv");
    int first_cls = 0; /* --from N: start at class N (the classes of an earlier table need not be measured again) */
    for (int i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], "--from")) first_cls = atoi(argv[i + 1]);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    kern_t table[C_COUNT];
    Fill<0>::go(table);
    const int wps_list[] = {1, 2, 3, 4, 6, 8}; /* waves per SIMD: a workgroup of 256 x 1..4 threads, 1 or 2 of them per CU */
    const int trips = 1500;
    long long* d_st = nullptr;
    uint32_t* d_sink = nullptr;
    const int max_waves = cus * 4 * 16 * 2;
    CHECK(hipMalloc(&d_st, sizeof(long long) * 2 * max_waves));
    CHECK(hipMalloc(&d_sink, 256));
    std::vector<long long> st(2 * (size_t)max_waves);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("# device %s (%s), %d CUs, clockRate %d kHz; %d instructions per wave per launch\n", prop.name, prop.gcnArchName, cus,
           prop.clockRate, trips * PER_TRIP);
    printf(csv ? "class,waves_per_simd,cycles_per_inst_one_cu,inst_per_cycle_per_simd_one_cu,chip_ns_per_inst_per_simd,chip_inst_per_ns_per_simd\n"
               : "%-28s %5s %12s %12s %14s\n", "class", "w/SIMD", "cyc/inst(CU)", "inst/cyc/SIMD", "chip ns/inst/SIMD");
    for (int c = first_cls; c < C_COUNT; c++) {
        for (int wps : wps_list) {
            /* 6 and 8 waves per SIMD come from two co-resident workgroups on the CU: measure them on the full chip only for
             * the wall-clock figure, and on one CU with a single workgroup of up to 1024 threads (4 waves per SIMD) */
            const int wg_threads = 256 * (wps <= 4 ? wps : wps / 2);
            const int wgs_per_cu = wps <= 4 ? 1 : 2;
            double cyc_per_inst = 0.0;
            if (wps <= 4) {
                hipLaunchKernelGGL(table[c], dim3(1), dim3(wg_threads), 0, 0, d_st, d_sink, trips, 0x9E3779B9u);
                CHECK(hipDeviceSynchronize());
                hipLaunchKernelGGL(table[c], dim3(1), dim3(wg_threads), 0, 0, d_st, d_sink, trips, 0x9E3779B9u);
                CHECK(hipDeviceSynchronize());
                const int nw = wg_threads / 64;
                CHECK(hipMemcpy(st.data(), d_st, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost));
                long long lo = st[0], hi = st[1];
                for (int w = 0; w < nw; w++) { lo = std::min(lo, st[2 * w]); hi = std::max(hi, st[2 * w + 1]); }
                cyc_per_inst = (double)(hi - lo) / ((double)trips * PER_TRIP * wps);
            }
            /* whole chip: every CU holds `wps` waves per SIMD, 3 launches, the fastest counts */
            const int grid = cus * wgs_per_cu;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(table[c], dim3(grid), dim3(wg_threads), 0, 0, d_st, d_sink, trips, 0x9E3779B9u);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            const double ns_per_inst = (double)best * 1e6 / ((double)trips * PER_TRIP * wps);
            if (csv)
                printf("%s,%d,%.3f,%.4f,%.4f,%.4f\n", kNames[c], wps, cyc_per_inst, cyc_per_inst > 0 ? 1.0 / cyc_per_inst : 0.0, ns_per_inst,
                       1.0 / ns_per_inst);
            else
                printf("%-28s %5d %12.3f %12.4f %14.4f\n", kNames[c], wps, cyc_per_inst, cyc_per_inst > 0 ? 1.0 / cyc_per_inst : 0.0, ns_per_inst);
        }
    }
    CHECK(hipFree(d_st));
    CHECK(hipFree(d_sink));
    return 0;
}
