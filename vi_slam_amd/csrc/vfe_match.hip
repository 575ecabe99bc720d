/* Brute-force Hamming kNN (k = 2) on 256-bit descriptors (cv::BFMatcher(NORM_HAMMING).knnMatch
 * semantics), ratio test and shard merge, for gfx950.
 *
 * Arithmetic bound, not HBM bound: every train row (32 bytes) meets every query, i.e. thousands of
 * 256-bit comparisons per byte fetched.  The comparisons run on the matrix cores as products of +-1
 * vectors (see below); a workgroup covers up to 2048 queries x one chunk of train rows, and per-chunk
 * top-2 lists are merged by (distance, index), which reproduces the order of one ascending scan with
 * strict '<' (ties keep the lower train index).
 */
#include "vfe_common.h"
#include <mutex>

typedef unsigned int u32;

#define NO_IDX (-1)
#define NO_DIST 0x7fffffff

__device__ static inline void top2_insert(int d, int i, int& d0, int& i0, int& d1, int& i1) {
    /* lexicographic (dist, idx) order; equals strict-'<' insertion when indices arrive ascending */
    if (d < d0 || (d == d0 && i < i0)) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
    else if (d < d1 || (d == d1 && i < i1)) { d1 = d; i1 = i; }
}

/* ---- the search on the matrix cores --------------------------------------------------------------------------------
 * With every bit b written as the number 1 - 2b, the inner product of two descriptors is 256 - 2 * hamming: the
 * distance table of a block of train rows against a block of queries is a matrix product with K = 256, and the nearest
 * neighbour is the largest product.  +1 and -1 are exact in the 4-bit float format, whose block-scaled MFMA
 * (v_mfma_scale_f32_32x32x64_f8f6f4, 32 x 32 results from 64-deep operands in 32 cycles, fp32 accumulation, scales 1)
 * delivers a 32 x 32 tile of distances in four instructions: 128 cycles where the vector ALU needs ~1300.
 *
 * A workgroup of eight waves serves QT x 256 queries and one chunk of train rows.  Every wave keeps the expanded
 * descriptors of its QT x 32 queries in registers for the whole chunk (the B operand: the query is the column of the
 * result, i.e. it sits on the lane, lanes l and l + 32 hold the rows 4h .. 4h + 3 (+8, +16, +24) of a column's
 * results).  The train rows pass through LDS 32 at a time: all threads expand a tile of packed rows (1 KB) into
 * nibbles (4 KB, through a 256-entry table in LDS) while the previous tile is being multiplied; each wave reads the
 * tile as its A operand.  How the 256
 * positions are dealt to instructions and lane halves does not matter as long as both operands use the same deal:
 * lane half h takes descriptor words 4h .. 4h + 3, instruction s the word 4h + s.
 *
 * Top-2 per lane: a result's key is product + (32767 - (16 * step + register)) / 32768, i.e. among equal products the
 * row met first (the lower train index) has the larger key, and the key says which row it was; products are even
 * integers of magnitude <= 256, so the sum is exact in fp32 (9 + 15 bits).  The position fraction is the C input of a
 * tile's first MFMA (16 registers, stepped once per 32 rows), so the accumulators come out as keys; the matrix pipe adds
 * it exactly (checked against the CPU scan on sets with thousands of ties).  The running pair (k1 >= k2) is updated by
 * k2 = med3(k1, k2, key), k1 = max(k1, key).  Most tiles cannot change any lane's pair (after t tiles a tile's chance
 * is ~32 / t per query), and the vector work of a tile has to fit beside its four MFMAs, so a tile first takes the
 * maximum of its 16 keys per lane and the update runs only if some lane's maximum exceeds its k2.  Rows past the end
 * of the chunk get a C input that keeps them below every real key. */
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define MM_WAVES 8
#define MM_ROW_DW 36                 /* LDS row: 32 dwords of nibbles + 4 of padding (ds_read_b128 of 32 rows: no bank conflicts) */
#define MM_TILE_DW (32 * MM_ROW_DW)
#define MM_LDS_DW (3 * MM_TILE_DW + 256) /* three train tiles and the byte -> nibbles table */
#define MM_MAX_ROWS 65536            /* per chunk: 2048 steps x 16 rows per lane fit the 15-bit position of a key */
#define MM_AHEAD 4                  /* train tiles requested ahead of the one being written to LDS: with few query tiles
                                      * per wave a step is a few hundred cycles, a trip to memory thousands */
#define MM_NONE (-1024.f)

/* bits 0..7 of x -> eight 4-bit floats, bit b as 1 - 2b (0x2 = +1.0, 0xA = -1.0) */
__device__ static inline u32 spread8(u32 x) {
    u32 t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return (t << 3) | 0x22222222u;
}
__device__ static inline v4i spread32(u32 w) {
    v4i r;
    r[0] = (int)spread8(w & 0xFFu); r[1] = (int)spread8((w >> 8) & 0xFFu);
    r[2] = (int)spread8((w >> 16) & 0xFFu); r[3] = (int)spread8(w >> 24);
    return r;
}
/* row of a tile that result register `reg` of lane half h holds */
__device__ static inline int mm_reg_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

/* (distance, index) of a key; `row0` = train index of the chunk's first row */
__device__ static inline void mm_decode(float key, int h, int row0, int& d, int& i) {
    const float p = floorf(key);
    if (key < -512.f || ((int)p & 1)) { d = NO_DIST; i = NO_IDX; return; } /* nothing found, or a floor (see mm_scan) */
    const int pos = 32767 - (int)((key - p) * 32768.f);
    d = (256 - (int)p) >> 1;
    i = row0 + (pos >> 4) * 32 + mm_reg_row(pos & 15, h);
}

/* Keys of the two nearest rows among tr[0 .. n_rows) for the workgroup's queries q0 + (wave * QT + j) * 32 + (lane & 31),
 * j < QT, over the rows this lane half sees.  `lds`: MM_LDS_DW dwords.  All 512 threads take part.
 * `floor_key` (may be NULL): per query an odd integer F = (product of a row already known to be the query's second
 * nearest elsewhere) - 1.  The pair starts at (F, F) instead of 'nothing': rows with a smaller product cannot be among the
 * query's two nearest, so they no longer trigger updates; rows with the same product still enter (their key exceeds F)
 * and are ordered by index when the lists are merged.  Real keys have even integer parts, so a floor left in a pair
 * decodes as 'nothing found'. */
template <int QT>
__device__ static inline void mm_scan(const u32* __restrict__ query, int nq, int q0, const u32* __restrict__ tr, int n_rows,
                                      const float* __restrict__ floor_key, u32* lds, float (&k1)[QT], float (&k2)[QT]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    v4i Bf[QT][4];
#pragma unroll
    for (int j = 0; j < QT; j++) {
        const int q = q0 + (wv * QT + j) * 32 + c;
        u32 w[4] = {0u, 0u, 0u, 0u};
        if (q < nq) __builtin_memcpy(w, query + (size_t)q * 8 + 4 * h, 16);
#pragma unroll
        for (int s = 0; s < 4; s++) Bf[j][s] = spread32(w[s]);
        k1[j] = k2[j] = (floor_key && q < nq) ? floor_key[q] : MM_NONE;
    }
    if (n_rows <= 0) return;
    const int n_steps = (n_rows + 31) >> 5;
    /* staging: thread -> (row, descriptor word, half of the word) of the tile */
    const int srow = tid >> 4, sd = (tid >> 1) & 7, shalf = tid & 1;
    const u32 lds_w = (u32)(srow * MM_ROW_DW + 4 * sd + 2 * shalf);
    const u32 lds_r = (u32)(c * MM_ROW_DW + 16 * h);
    auto gload = [&](int step) -> u32 { return tr[(size_t)min(step * 32 + srow, n_rows - 1) * 8 + sd]; };
    /* byte -> eight nibbles through a table behind the tiles: two LDS reads instead of 14 vector instructions per thread
     * and step (the step's vector work competes with the matrix instructions for the SIMD's issue slots) */
    u32* lut = lds + 3 * MM_TILE_DW;
    if (tid < 256) lut[tid] = spread8((u32)tid);
    auto swrite = [&](u32 w, int buf) {
        const u32 x = w >> (16 * shalf);
        uint2 v;
        v.x = lut[x & 0xFFu]; v.y = lut[(x >> 8) & 0xFFu];
        *reinterpret_cast<uint2*>(lds + buf * MM_TILE_DW + lds_w) = v;
    };
    /* the C input of a tile's first MFMA: the position part of the keys of this step's 16 rows (rows past the end of
     * the chunk get a value that keeps them below every real key) */
    v16f cin;
#pragma unroll
    for (int r = 0; r < 16; r++) cin[r] = (float)(32767 + 16 - r) * (1.f / 32768.f);
    auto set_cin = [&](int step) { /* steps in ascending order */
#pragma unroll
        for (int r = 0; r < 16; r++) cin[r] -= 16.f / 32768.f;
        if (step == n_steps - 1) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (step * 32 + mm_reg_row(r, h) >= n_rows) cin[r] = -4096.f;
        }
    };
    const int scale = 0x7f7f7f7f; /* E8M0 127 = 1.0 for every block */
    auto read_tile = [&](int buf, v8i (&F)[4]) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const v4i a = *reinterpret_cast<const v4i*>(lds + buf * MM_TILE_DW + lds_r + 4 * s);
            F[s] = v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0};
        }
    };
    v8i Af[4];
    auto tile = [&](int j) -> v16f {
        v16f acc = cin;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const v8i b = v8i{Bf[j][s][0], Bf[j][s][1], Bf[j][s][2], Bf[j][s][3], 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[s], b, acc, 4, 4, 0, scale, 0, scale);
        }
        return acc;
    };
    const float big = 1e30f; /* med3(a, b, big) = max(a, b) for keys; a maximum proper would first quiet each input */
    /* Three LDS tiles.  At the top of step s: Af holds tile s, tile s + 1 is complete in LDS, the packed rows of the
     * tiles from s + 2 on are on their way from memory, and the products of (tile s, query tile 0) are in flight.  The step
     * writes tile s + 2, requests a tile further on, reads tile s + 1 for the first product of the next step as soon as
     * its own products are issued, and meets the other waves once, at its end.  The products of the next query tile are always issued before the keys of the
     * current one are looked at, so that a tile's vector work runs in the shadow of the next tile's matrix work. */
    u32 pre[MM_AHEAD]; /* packed rows of tiles step + 2 .. step + 1 + MM_AHEAD */
    {
        const u32 g0 = gload(0), g1 = gload(1);
#pragma unroll
        for (int i = 0; i < MM_AHEAD; i++) pre[i] = gload(2 + i);
        __syncthreads(); /* the table is complete */
        swrite(g0, 0);
        swrite(g1, 1);
    }
    __syncthreads();
    read_tile(0, Af);
    set_cin(0);
    v16f acc = tile(0);
    int wbuf = 2, rbuf = 1; /* (step + 2) % 3, (step + 1) % 3 */
    for (int step = 0; step < n_steps; step++) {
        const bool more = step + 1 < n_steps; /* wave-uniform */
        swrite(pre[0], wbuf);
#pragma unroll
        for (int i = 0; i + 1 < MM_AHEAD; i++) pre[i] = pre[i + 1];
        pre[MM_AHEAD - 1] = gload(step + 2 + MM_AHEAD);
#pragma unroll
        for (int j = 0; j < QT; j++) {
            v16f nxt = acc;
            if (j + 1 < QT) nxt = tile(j + 1);
            else if (more) {
                /* tile step + 1 has been complete since the barrier that ended the previous step: its first products are
                 * issued before this step's barrier, which then passes in the shadow of the matrix pipe instead of
                 * leaving it idle for the barrier and the LDS round trip */
                read_tile(rbuf, Af); /* this step's products have all been issued */
                set_cin(step + 1);
                nxt = tile(0);
            }
            /* keeps the order: without it the compiler moves the matrix instructions down to where their results are
             * read, behind this tile's vector work */
            asm volatile("" : "+v"(nxt), "+v"(acc));
            /* compiler-visible arithmetic on the results: it pads the wait an MFMA result needs before a vector
             * instruction may read it, which it does not do for inline assembly */
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; r++) m = __builtin_fmaxf(m, acc[r]);
            if (__builtin_amdgcn_ballot_w64(m > k2[j]) != 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    k2[j] = __builtin_amdgcn_fmed3f(k1[j], k2[j], acc[r]);
                    k1[j] = __builtin_amdgcn_fmed3f(k1[j], acc[r], big);
                }
            }
            acc = nxt;
        }
        __syncthreads(); /* tile step + 2 is complete for the read that ends the next step; tile step + 1's buffer is free from there on */
        wbuf = wbuf == 2 ? 0 : wbuf + 1;
        rbuf = rbuf == 2 ? 0 : rbuf + 1;
    }
}

/* the lane's two keys of query tile j -> the query's two nearest rows (both lane halves), valid in lanes 0..31 */
__device__ static inline void mm_finish(float k1, float k2, int h, int row0, int& d0, int& i0, int& d1, int& i1) {
    int da, ia, db, ib;
    mm_decode(k1, h, row0, da, ia);
    mm_decode(k2, h, row0, db, ib);
    d0 = da; i0 = ia; d1 = db; i1 = ib;
    const int oda = __shfl_xor(da, 32), oia = __shfl_xor(ia, 32), odb = __shfl_xor(db, 32), oib = __shfl_xor(ib, 32);
    if (oia >= 0) top2_insert(oda, oia, d0, i0, d1, i1);
    if (oib >= 0) top2_insert(odb, oib, d0, i0, d1, i1);
}

/* workgroup (c, qb): chunk c of the train rows against queries qb * QT * 256 ..; the top-2 list of (query q, chunk c) goes
 * to list slot q * list_stride + list0 + c */
template <int QT>
__global__ __launch_bounds__(64 * MM_WAVES) void k_match_mm(const u32* __restrict__ query, int nq,
                                                            const u32* __restrict__ train, long long nt, int chunk,
                                                            int list_stride, int list0, int base_index, const float* __restrict__ floor_key,
                                                            int* __restrict__ part_idx, int* __restrict__ part_dist) {
    __shared__ __align__(16) u32 s_tile[MM_LDS_DW];
    const int c = blockIdx.x, q0 = blockIdx.y * (QT * 32 * MM_WAVES);
    const long long t0 = (long long)c * chunk;
    const long long t1 = (t0 + chunk < nt) ? t0 + chunk : nt;
    float k1[QT], k2[QT];
    mm_scan<QT>(query, nq, q0, train + (size_t)t0 * 8, (int)(t1 - t0), floor_key, s_tile, k1, k2);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int j = 0; j < QT; j++) {
        int d0, i0, d1, i1;
        mm_finish(k1[j], k2[j], lane >> 5, base_index + (int)t0, d0, i0, d1, i1);
        const int q = q0 + (wv * QT + j) * 32 + lane;
        if (lane < 32 && q < nq) {
            const size_t o = ((size_t)q * list_stride + list0 + c) * 2;
            part_idx[o] = i0; part_idx[o + 1] = i1;
            part_dist[o] = d0; part_dist[o + 1] = d1;
        }
    }
}

/* merge `nlists` top-2 lists per query; list l of query q at [(q*q_stride + l*l_stride) * 2] */
__global__ __launch_bounds__(256) void k_merge(const int* __restrict__ in_idx, const int* __restrict__ in_dist,
                                               int nlists, size_t q_stride, size_t l_stride, int nq,
                                               int* __restrict__ out_idx, int* __restrict__ out_dist) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    int d0 = NO_DIST, d1 = NO_DIST, i0 = NO_IDX, i1 = NO_IDX;
    for (int l = 0; l < nlists; l++) {
        const size_t o = ((size_t)q * q_stride + (size_t)l * l_stride) * 2;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = in_idx[o + k], d = in_dist[o + k];
            if (i >= 0) top2_insert(d, i, d0, i0, d1, i1);
        }
    }
    out_idx[2 * q] = i0; out_idx[2 * q + 1] = i1;
    out_dist[2 * q] = d0; out_dist[2 * q + 1] = d1;
}

/* The same merge with one wavefront per query, for many lists per query (the per-chunk partial lists of a large train
 * set): the lanes take the lists round-robin -- consecutive lists of a query are consecutive in memory -- and their
 * top-2 lists are merged by a butterfly.  The two smallest (dist, idx) pairs of a set do not depend on the order in
 * which its members are visited, so the result is the one of the serial merge. */
__global__ __launch_bounds__(256) void k_merge_wave(const int* __restrict__ in_idx, const int* __restrict__ in_dist,
                                                    int nlists, int nq, int* __restrict__ out_idx, int* __restrict__ out_dist,
                                                    int out_stride, float* __restrict__ floor_key) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (q >= nq) return;
    int d0 = NO_DIST, d1 = NO_DIST, i0 = NO_IDX, i1 = NO_IDX;
    const int2* pi = reinterpret_cast<const int2*>(in_idx) + (size_t)q * nlists;
    const int2* pd = reinterpret_cast<const int2*>(in_dist) + (size_t)q * nlists;
    for (int l = lane; l < nlists; l += 64) {
        const int2 i = pi[l], d = pd[l];
        if (i.x >= 0) top2_insert(d.x, i.x, d0, i0, d1, i1);
        if (i.y >= 0) top2_insert(d.y, i.y, d0, i0, d1, i1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int pd0 = __shfl_xor(d0, o), pi0 = __shfl_xor(i0, o), pd1 = __shfl_xor(d1, o), pi1 = __shfl_xor(i1, o);
        if (pi0 >= 0) top2_insert(pd0, pi0, d0, i0, d1, i1);
        if (pi1 >= 0) top2_insert(pd1, pi1, d0, i0, d1, i1);
    }
    if (lane == 0) { /* the result is list slot q * out_stride; out_stride = 1: the final [nq][2] layout */
        const size_t o = (size_t)q * out_stride * 2;
        out_idx[o] = i0; out_idx[o + 1] = i1;
        out_dist[o] = d0; out_dist[o + 1] = d1;
        if (floor_key) floor_key[q] = i1 >= 0 ? (float)(256 - 2 * d1 - 1) : MM_NONE; /* see mm_scan */
    }
}

__global__ __launch_bounds__(256) void k_ratio(const int* __restrict__ idx, const int* __restrict__ dist, int nq,
                                               float ratio, uint8_t* __restrict__ good) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    good[q] = (idx[2 * q] >= 0 && idx[2 * q + 1] >= 0 && (float)dist[2 * q] < ratio * (float)dist[2 * q + 1]) ? 1 : 0;
}

/* Batched frame-to-frame matching: workgroup (x, b) serves QT * 256 queries of frame b against all of that frame's train
 * rows, so its lists are final and the ratio test can follow at once.  Set sizes come from device memory, so extraction
 * and matching can be queued back to back. */
template <int QT>
__global__ __launch_bounds__(64 * MM_WAVES) void k_match_mm_batch(const u32* __restrict__ query, const int* __restrict__ nq_dev, int q_cap,
                                                                  const u32* __restrict__ train, const int* __restrict__ nt_dev, int t_cap,
                                                                  int* __restrict__ out_idx, int* __restrict__ out_dist, float ratio,
                                                                  uint8_t* __restrict__ good) {
    __shared__ __align__(16) u32 s_tile[MM_LDS_DW];
    const int b = blockIdx.y, q0 = blockIdx.x * (QT * 32 * MM_WAVES);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nq = min(max(nq_dev[b], 0), q_cap);
    /* workgroups past the last query only write the 'no neighbour' records */
    const int nt = q0 < nq ? min(max(nt_dev[b], 0), t_cap) : 0;
    const u32* qb = query + (size_t)b * q_cap * 8;
    const u32* tb = train + (size_t)b * t_cap * 8;
    int d0[QT], i0[QT], d1[QT], i1[QT];
#pragma unroll
    for (int j = 0; j < QT; j++) { d0[j] = d1[j] = NO_DIST; i0[j] = i1[j] = NO_IDX; }
    for (int ts = 0; ts < nt; ts += MM_MAX_ROWS) { /* one round unless the set is huge */
        float k1[QT], k2[QT];
        mm_scan<QT>(qb, nq, q0, tb + (size_t)ts * 8, min(nt - ts, MM_MAX_ROWS), nullptr, s_tile, k1, k2);
        __syncthreads(); /* the next round reuses the tiles */
#pragma unroll
        for (int j = 0; j < QT; j++) {
            int da, ia, db, ib;
            mm_finish(k1[j], k2[j], lane >> 5, ts, da, ia, db, ib);
            if (ia >= 0) top2_insert(da, ia, d0[j], i0[j], d1[j], i1[j]);
            if (ib >= 0) top2_insert(db, ib, d0[j], i0[j], d1[j], i1[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < QT; j++) {
        const int q = q0 + (wv * QT + j) * 32 + lane;
        if (lane < 32 && q < q_cap) {
            if (q >= nq) { d0[j] = d1[j] = NO_DIST; i0[j] = i1[j] = NO_IDX; }
            const size_t o = ((size_t)b * q_cap + q) * 2;
            out_idx[o] = i0[j]; out_idx[o + 1] = i1[j];
            out_dist[o] = d0[j]; out_dist[o + 1] = d1[j];
            if (good) good[(size_t)b * q_cap + q] = (i0[j] >= 0 && i1[j] >= 0 && (float)d0[j] < ratio * (float)d1[j]) ? 1 : 0;
        }
    }
}

/* Query tiles per wave (a workgroup serves QT * 256 queries) and rows per chunk: one workgroup per CU and round where the
 * train set allows it -- long chunks keep the top-2 update rare -- chunks of whole 32-row tiles, at least 256 rows. */
struct MatchGeom {
    int qt, chunk; long long nchunks; unsigned qblocks;
    /* large train sets: a sample of the first s_rows rows (s_chunks chunks of s_chunk rows) is searched first; its
     * second-nearest distance per query becomes the floor of the main search over the rest (mm_scan) */
    int s_chunk, s_chunks; long long s_rows;
    long long lists; /* top-2 lists per query that the final merge reads */
};
#define MM_SAMPLE_CHUNK 256          /* rows per sample chunk (half as many sample chunks as main chunks: their lists
                                      * are workspace, and the sample is a few per cent of the work) */
#define MM_SAMPLE_MIN_ROWS 500000    /* train sets from this size on are searched with a floor */
static MatchGeom match_geom(int nq, long long nt) {
    MatchGeom g;
    g.qt = nq > 1024 ? 8 : nq > 512 ? 4 : nq > 256 ? 2 : 1;
    for (;;) {
        g.qblocks = (unsigned)((nq + g.qt * 256 - 1) / (g.qt * 256));
        long long want = 256 / (long long)g.qblocks;
        if (want < 1) want = 1;
        g.s_chunk = MM_SAMPLE_CHUNK; g.s_chunks = 0; g.s_rows = 0;
        if (nt >= MM_SAMPLE_MIN_ROWS) { g.s_chunks = (int)((want + 1) / 2); g.s_rows = (long long)g.s_chunks * g.s_chunk; }
        const long long rest = nt - g.s_rows;
        long long chunk = ((rest + want - 1) / want + 31) / 32 * 32;
        if (chunk < 128) chunk = 128;
        if (chunk > MM_MAX_ROWS) chunk = MM_MAX_ROWS;
        g.chunk = (int)chunk;
        g.nchunks = rest > 0 ? (rest + chunk - 1) / chunk : 1;
        g.lists = g.nchunks + (g.s_chunks ? 1 : 0);
        /* a small train set gives few chunks: fewer query tiles per wave, i.e. more query blocks, keep the chip busy */
        if (g.qt == 1 || (long long)g.qblocks * g.nchunks >= 128) break;
        g.qt >>= 1;
    }
    return g;
}

/* workspace: the lists the final merge reads, then (with a sample) the sample's lists and the floors */
struct MatchWs { size_t idx, dist, s_idx, s_dist, floor, total; };
static MatchWs match_ws(int nq, const MatchGeom& g) {
    MatchWs w;
    size_t o = 0;
    const size_t lists = align_up((size_t)nq * g.lists * 2 * sizeof(int), 256);
    const size_t s_lists = align_up((size_t)nq * g.s_chunks * 2 * sizeof(int), 256);
    w.idx = o; o += lists;
    w.dist = o; o += lists;
    w.s_idx = o; o += s_lists;
    w.s_dist = o; o += s_lists;
    w.floor = o; o += g.s_chunks ? align_up((size_t)nq * sizeof(float), 256) : 0;
    w.total = o;
    return w;
}

extern "C" size_t vfe_match_workspace(int nq, long long nt) {
    if (nq <= 0 || nt < 0) return 0;
    return match_ws(nq, match_geom(nq, nt)).total;
}

static void match_merge(const int* in_idx, const int* in_dist, long long nlists, int nq, int* out_idx, int* out_dist,
                        int out_stride, float* floor_key, hipStream_t st) {
    if (nlists >= 16 || out_stride != 1 || floor_key) /* a thread per query would walk its lists alone: a wave per query instead */
        hipLaunchKernelGGL(k_merge_wave, dim3((nq + 3) / 4), dim3(256), 0, st, in_idx, in_dist, (int)nlists, nq, out_idx, out_dist,
                           out_stride, floor_key);
    else
        hipLaunchKernelGGL(k_merge, dim3((nq + 255) / 256), dim3(256), 0, st, in_idx, in_dist, (int)nlists,
                           (size_t)nlists, (size_t)1, nq, out_idx, out_dist);
}

/* k_match_mm over all (chunk, query block) pairs into the lists at `workspace` (vfe_match_workspace bytes), then the merge
 * of every query's lists; with a sample, the same two steps over the sample first */
static int match_launch(const uint8_t* query, int nq, const uint8_t* train, long long nt, int base_index,
                        int* out_idx, int* out_dist, void* workspace, const MatchGeom& g, hipStream_t st) {
    const MatchWs w = match_ws(nq, g);
    char* ws = (char*)workspace;
    int* part_idx = (int*)(ws + w.idx);
    int* part_dist = (int*)(ws + w.dist);
    const dim3 block(64 * MM_WAVES);
#define MM_LAUNCH(QT, GRID, TRAIN, NT, CHUNK, STRIDE, LIST0, BASE, FLOOR, IDX, DIST)                                     \
    hipLaunchKernelGGL(k_match_mm<QT>, GRID, block, 0, st, (const u32*)query, nq, (const u32*)(TRAIN), NT, CHUNK,       \
                       STRIDE, LIST0, BASE, FLOOR, IDX, DIST)
#define MM_LAUNCH_QT(...)                                                                                                \
    switch (g.qt) {                                                                                                      \
        case 8: MM_LAUNCH(8, __VA_ARGS__); break;                                                                        \
        case 4: MM_LAUNCH(4, __VA_ARGS__); break;                                                                        \
        case 2: MM_LAUNCH(2, __VA_ARGS__); break;                                                                        \
        default: MM_LAUNCH(1, __VA_ARGS__); break;                                                                       \
    }
    float* floor_key = nullptr;
    if (g.s_chunks) {
        int* s_idx = (int*)(ws + w.s_idx);
        int* s_dist = (int*)(ws + w.s_dist);
        floor_key = (float*)(ws + w.floor);
        {
            ProfScope prof(VFE_K_MATCH, st);
            const dim3 grid((unsigned)g.s_chunks, g.qblocks);
            MM_LAUNCH_QT(grid, train, g.s_rows, g.s_chunk, g.s_chunks, 0, base_index, (const float*)nullptr, s_idx, s_dist);
        }
        VFE_LAUNCH_CHECK();
        ProfScope prof(VFE_K_MISC, st);
        /* the sample's result is list 0 of every query */
        match_merge(s_idx, s_dist, g.s_chunks, nq, part_idx, part_dist, (int)g.lists, floor_key, st);
        VFE_LAUNCH_CHECK();
    }
    {
        ProfScope prof(VFE_K_MATCH, st);
        const dim3 grid((unsigned)g.nchunks, g.qblocks);
        MM_LAUNCH_QT(grid, train + (size_t)g.s_rows * 32, nt - g.s_rows, g.chunk, (int)g.lists, g.s_chunks ? 1 : 0,
                     base_index + (int)g.s_rows, (const float*)floor_key, part_idx, part_dist);
    }
#undef MM_LAUNCH_QT
#undef MM_LAUNCH
    VFE_LAUNCH_CHECK();
    ProfScope prof(VFE_K_MISC, st);
    match_merge(part_idx, part_dist, g.lists, nq, out_idx, out_dist, 1, nullptr, st);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

static int match_check_args(const uint8_t* query, int nq, const uint8_t* train, long long nt, int base_index,
                            const int* out_idx, const int* out_dist, MatchGeom* g) {
    VFE_REQUIRE(query && out_idx && out_dist && nq > 0, "NULL buffer or nq <= 0");
    VFE_REQUIRE(nt >= 0 && (nt == 0 || train), "bad train set");
    VFE_REQUIRE((long long)base_index + nt <= 0x7fffffffLL, "train indices overflow int");
    VFE_REQUIRE((uintptr_t)query % 4 == 0 && (uintptr_t)train % 4 == 0, "descriptors must be 4-byte aligned");
    *g = match_geom(nq, nt);
    VFE_REQUIRE(g->lists <= 0x7fffffffLL && g->qblocks <= 65535u, "train or query set too large for one call");
    return VFE_OK;
}

extern "C" int vfe_dev_match_knn2(const uint8_t* query, int nq, const uint8_t* train, long long nt,
                                  int base_index, int* out_idx, int* out_dist,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    MatchGeom g;
    const int rc = match_check_args(query, nq, train, nt, base_index, out_idx, out_dist, &g);
    if (rc != VFE_OK) return rc;
    const size_t need = vfe_match_workspace(nq, nt);
    if (!workspace || workspace_bytes < need) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
        return VFE_E_NOMEM;
    }
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    return match_launch(query, nq, train, nt, base_index, out_idx, out_dist, workspace, g, (hipStream_t)stream);
}

/* ---- the same search for callers that cannot allocate --------------------------------------------------------------
 * The per-chunk partial lists live in a file-scope device array instead of a caller workspace.  Every GPU the library
 * runs on holds its own copy of the array at its own address for the life of the process; the host-side name below is
 * only a handle for hipGetSymbolAddress, never a device address, so every use goes through the per-device pointer
 * resolved once (vfe_match_scratch_init, called by vfe_create and by the entry point's first call on a device).
 * The search kernels write every list slot and floor the merges read, so nothing has to be cleared between calls; calls that
 * overlap on one GPU share the array and race. */
static __device__ __align__(256) unsigned char scratch[VFE_MATCH_NOALLOC_BYTES];
#define MATCH_MAX_DEVICES 64
static void* g_scratch_ptr[MATCH_MAX_DEVICES];
static std::mutex g_scratch_mu;

int vfe_match_scratch_init(void** out) {
    int dev = -1;
    VFE_HIP(hipGetDevice(&dev));
    VFE_REQUIRE(dev >= 0 && dev < MATCH_MAX_DEVICES, "device ordinal beyond the scratch table");
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        VFE_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        VFE_REQUIRE(p && (uintptr_t)p % 256 == 0, "scratch symbol did not resolve to an aligned device address");
        g_scratch_ptr[dev] = p;
    }
    if (out) *out = g_scratch_ptr[dev];
    return VFE_OK;
}

extern "C" int vfe_dev_match_knn2_noalloc(const uint8_t* query, int nq, const uint8_t* train, long long nt,
                                          int base_index, int* out_idx, int* out_dist, void* stream) {
    MatchGeom g;
    int rc = match_check_args(query, nq, train, nt, base_index, out_idx, out_dist, &g);
    if (rc != VFE_OK) return rc;
    const size_t need = vfe_match_workspace(nq, nt);
    if (need > (size_t)VFE_MATCH_NOALLOC_BYTES) {
        vfe_set_error("%d queries x %lld chunks need %zu bytes of partial lists, the built-in scratch holds %zu: "
                      "use vfe_dev_match_knn2 with a workspace", nq, g.lists, need, (size_t)VFE_MATCH_NOALLOC_BYTES);
        return VFE_E_CAPACITY;
    }
    void* ws = nullptr;
    rc = vfe_match_scratch_init(&ws);
    if (rc != VFE_OK) return rc;
    return match_launch(query, nq, train, nt, base_index, out_idx, out_dist, ws, g, (hipStream_t)stream);
}

extern "C" int vfe_dev_merge_knn2(const int* shard_idx, const int* shard_dist, int nshards, int nq,
                                  int* out_idx, int* out_dist, void* stream) {
    VFE_REQUIRE(shard_idx && shard_dist && out_idx && out_dist && nshards > 0 && nq > 0, "NULL buffer or bad sizes");
    ProfScope prof(VFE_K_MISC, (hipStream_t)stream);
    /* input layout [nshards][nq][2] */
    hipLaunchKernelGGL(k_merge, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, shard_idx, shard_dist, nshards,
                       (size_t)1, (size_t)nq, nq, out_idx, out_dist);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_ratio_test(const int* idx, const int* dist, int nq, float ratio, uint8_t* good, void* stream) {
    VFE_REQUIRE(idx && dist && good && nq > 0, "NULL buffer or nq <= 0");
    hipLaunchKernelGGL(k_ratio, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, dist, nq, ratio, good);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_match_knn2_batch(const uint8_t* query, const int* nq, int q_cap,
                                        const uint8_t* train, const int* nt, int t_cap, int batch,
                                        int* out_idx, int* out_dist, float ratio, uint8_t* good, void* stream) {
    VFE_REQUIRE(query && train && nq && nt && out_idx && out_dist, "NULL buffer");
    VFE_REQUIRE(q_cap > 0 && t_cap > 0 && batch > 0 && batch <= 65535, "bad sizes");
    VFE_REQUIRE((uintptr_t)query % 4 == 0 && (uintptr_t)train % 4 == 0, "descriptors must be 4-byte aligned");
    ProfScope prof(VFE_K_MATCH, (hipStream_t)stream);
    /* the most query tiles per wave that still give every CU a workgroup */
    int qt = 8; /* (64 pairs of 2000 x 2000: 60 us with 256 workgroups, 86 with 128, 145 with 64) */
    while (qt > 1 && (long long)((q_cap + qt * 256 - 1) / (qt * 256)) * batch < 256) qt >>= 1;
    const dim3 grid((unsigned)((q_cap + qt * 256 - 1) / (qt * 256)), (unsigned)batch), block(64 * MM_WAVES);
#define MM_LAUNCH(QT)                                                                                                  \
    hipLaunchKernelGGL(k_match_mm_batch<QT>, grid, block, 0, (hipStream_t)stream, (const u32*)query, nq, q_cap,        \
                       (const u32*)train, nt, t_cap, out_idx, out_dist, ratio, good)
    switch (qt) {
        case 8: MM_LAUNCH(8); break;
        case 4: MM_LAUNCH(4); break;
        case 2: MM_LAUNCH(2); break;
        default: MM_LAUNCH(1); break;
    }
#undef MM_LAUNCH
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
