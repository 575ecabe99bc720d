/* Shi-Tomasi corner detection (cv::goodFeaturesToTrack semantics) for gfx950, batched.
 *
 * Two kernels per call:
 *  k_gftt_eig    (vfe_gftt_eig.hip) one pass over the image: Sobel -> gradient products -> 3x3 box sums ->
 *                smaller eigenvalue, marching down the rows with everything in registers; fused with the
 *                frame maximum (atomic) and 3x3 non-max suppression, so the float map never
 *                goes to HBM.  Local maxima are appended to a per-frame candidate list as
 *                64-bit keys (score bits << 32 | y << 16 | x: same order as the linear index).  Candidates that cannot pass
 *                the final quality threshold are dropped early against the running maximum
 *                (a lower bound of the final one), which only removes keys the exact filter
 *                would remove anyway.
 *  k_gftt_select one workgroup per frame: exact threshold (the keys that pass are packed into the
 *                threads' registers), candidates next to a tracked point dropped, then the rest is consumed in
 *                descending key order in batches (radix select of the next 512 .. 8192 keys, bitonic
 *                sort -- in registers up to 1024 keys, vfe_sort.h, in LDS beyond) by a wave-parallel
 *                version of OpenCV's greedy min-distance walk:
 *                64 candidates are tested against the accepted set at once, conflicts inside
 *                the 64 are resolved in rank order with ballots.  The result equals the serial
 *                walk because a candidate's fate depends only on higher-ranked ones.
 *
 * Float arithmetic follows the CPU oracle operation by operation (no contraction), so corner
 * sets are bit-exact.
 */
#include "vfe_gftt.h"
#include "vfe_sort.h"

#define SEL_THREADS 1024
#define SEL_M 8192
#define SEED_CELLS 8192 /* LDS grid of tracked points: cell words + nodes alias the 64 KiB sort buffer */
#define SEED_NODES 4096
#define SEL_M_FIRST 2048 /* first batch of the ordered walk: usually enough, and sorts 5x faster than a full one */
#define WALK_CELLS 24576 /* LDS min-distance grid of the corners accepted in this call */
#define WALK_NODES 4096
#define KEY_REGS 9 /* candidate keys a thread of the selection holds in registers */

struct SelParams {
    int W, H, max_corners;
    double quality;
    int use_grid, cell, gw, gh;
    u32 cell_magic; /* ceil(2^32 / cell): v / cell = umulhi(v, cell_magic) for v < 65536, cell > 1 */
    u32 md2i;
    const u32* frame_max;
    const u32* cand_count;
    u64* cand;       /* keys blocked by a tracked point are cleared in place */
    u32 seg_cap;     /* keys per candidate segment; GF_NSEG segments per frame */
    u32* heads;      /* [batch][gw*gh], 0 = empty, else node index + 1 */
    u32* node_xy;    /* [batch][node_cap]  x | y << 16 */
    u32* node_next;  /* [batch][node_cap] */
    int node_cap;
    const float* seeds;
    const int* n_seeds;
    int seed_cap;
    float* corners;
    float* scores;
    int* counts;
    /* append mode (both or neither): the accepted corners also go behind the frame's tracked points, append_pts[b][n + i]
     * with n the clamped seed count, and the count becomes n + accepted -- the seeds' own arrays, [batch][seed_cap] */
    float* append_pts;
    int* append_n;
    /* compaction in front (append mode only, or cmp_src = NULL): the frame's tracked points are first gathered -- stably --
     * from cmp_src[b][0 .. n_seeds[b]) where cmp_status is set and the rounded position lies inside the image less a
     * one-pixel rim, into seeds[b][...]; the count goes to n_seeds[b] and cmp_n_out[b] (may be NULL).  One launch less
     * between the tracker and the selection. */
    const float* cmp_src;
    const uint8_t* cmp_status;
    int* cmp_n_out;
};

__device__ static inline u32 ld_agent(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline void st_agent(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* v / cell for 0 <= v < 65536 */
__device__ static inline int cell_of(int v, int cell, u32 magic) { return cell > 1 ? (int)__umulhi((u32)v, magic) : v; }

/* inclusive suffix scan of 1024 values held one per thread; returns sum over threads >= tid.  Inside a wave by shuffles,
 * across the 16 waves through LDS: three barriers instead of a barrier per doubling step. */
__device__ static inline u32 block_suffix_scan(u32 v, u32 (*buf)[SEL_THREADS]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 x = (u32)__shfl_down((int)v, o);
        if (lane + o < 64) v += x;
    }
    if (lane == 0) buf[0][wv] = v; /* the wave's total */
    __syncthreads();
    u32 above = 0;
#pragma unroll
    for (int w = 1; w < SEL_THREADS / 64; w++)
        if (w > wv) above += buf[0][w];
    __syncthreads();
    return v + above;
}

__global__ __launch_bounds__(SEL_THREADS) void k_gftt_select(SelParams p) {
    __shared__ u64 s_keys[SEL_M];
    __shared__ u32 s_hist[2048 + 1];
    __shared__ u32 s_scan[2][SEL_THREADS];
    __shared__ u32 s_cnt;
    __shared__ u32 s_dmin;
    __shared__ int s_nacc;
    /* first the pool through which the valid keys reach the threads' registers, then the walk's grid */
    __shared__ u64 s_pool[KEY_REGS * SEL_THREADS];
    unsigned short* w_heads = reinterpret_cast<unsigned short*>(s_pool);            /* [WALK_CELLS] 0 = empty, else node + 1 */
    u32* w_xy = reinterpret_cast<u32*>(w_heads + WALK_CELLS);                       /* [WALK_NODES] x | y << 16 */
    unsigned short* w_next = reinterpret_cast<unsigned short*>(w_xy + WALK_NODES);  /* [WALK_NODES] */
    static_assert(WALK_CELLS * 2 + WALK_NODES * 6 <= sizeof(u64) * KEY_REGS * SEL_THREADS, "the walk's grid fits the pool");

    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ u32 s_segn[GF_NSEG];
    if (tid < GF_NSEG) s_segn[tid] = min(p.cand_count[(b * GF_NSEG + tid) * GF_CSTRIDE], p.seg_cap);
    u64* keys = p.cand + (size_t)b * GF_NSEG * p.seg_cap;
    __syncthreads();
    /* Every pass below visits all candidate keys of the frame that pass the exact threshold.  The frame's segments are
     * taken as one list; its first KEY_REGS * SEL_THREADS keys (a 1080p frame has ~9000, half of them above the
     * threshold) are loaded once, and those above the threshold are packed through LDS into as few register slots per
     * thread as they need -- every pass then runs over about five full slots per thread instead of ten half-empty
     * ones.  Only a busier frame's rest stays in memory and is re-read by every pass, four loads in flight.
     * f(key) returns true to clear the key (0 = no key). */
    u32 seg_at[GF_NSEG + 1]; /* where each segment starts in the list */
    seg_at[0] = 0;
#pragma unroll
    for (int sg = 0; sg < GF_NSEG; sg++) seg_at[sg + 1] = seg_at[sg] + s_segn[sg];
    const u32 n_keys = seg_at[GF_NSEG];
    auto key_at = [&](u32 g) -> u64* { /* g < n_keys */
        u32 sg = 0, at = 0;
#pragma unroll
        for (int k = 1; k < GF_NSEG; k++)
            if (g >= seg_at[k]) { sg = (u32)k; at = seg_at[k]; }
        return keys + (size_t)sg * p.seg_cap + (g - at);
    };
    u64 rk[KEY_REGS];
#pragma unroll
    for (int j = 0; j < KEY_REGS; j++) rk[j] = (u32)tid + j * SEL_THREADS < n_keys ? *key_at((u32)tid + j * SEL_THREADS) : 0ull;
    u32 n_pool = 0; /* keys in registers: slot j of thread t holds key t + j * SEL_THREADS of the pool */
    auto for_each_key_in_memory = [&](auto&& f) {
        for (u32 i = tid + KEY_REGS * SEL_THREADS; i < n_keys; i += 4 * SEL_THREADS) {
            u64* kp[4];
            u64 k[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                kp[j] = i + j * SEL_THREADS < n_keys ? key_at(i + j * SEL_THREADS) : nullptr;
                k[j] = kp[j] ? *kp[j] : 0ull;
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (k[j] != 0ull && f(k[j])) *kp[j] = 0ull;
        }
    };
    auto for_each_key = [&](auto&& f) {
#pragma unroll
        for (int j = 0; j < KEY_REGS; j++)
            if ((u32)j * SEL_THREADS < n_pool && rk[j] != 0ull && f(rk[j])) rk[j] = 0ull;
        for_each_key_in_memory(f);
    };
    const float maxv = __uint_as_float(p.frame_max[b * GF_CSTRIDE]);
    const float thr = (float)((double)maxv * p.quality);
    const u64 lowkey = ((u64)__float_as_uint(thr) << 32) | 0xFFFFFFFFull; /* valid keys are > lowkey */
    u32* heads = p.heads + (size_t)b * p.gw * p.gh;
    u32* node_xy = p.node_xy + (size_t)b * p.node_cap;
    u32* node_next = p.node_next + (size_t)b * p.node_cap;

    if (tid == 0) { s_cnt = 0; s_nacc = 0; }
    /* refill budget: tracked points count towards max_corners */
    int ns = (p.seeds && p.n_seeds) ? max(0, min(p.n_seeds[b], p.seed_cap)) : 0;
    if (p.cmp_src) {
        const int n_in = ns, lane = tid & 63, wv = tid >> 6;
        const float* src = p.cmp_src + (size_t)b * p.seed_cap * 2;
        const uint8_t* stt = p.cmp_status + (size_t)b * p.seed_cap;
        float* dst = p.append_pts + (size_t)b * p.seed_cap * 2;
        int base = 0;
        for (int start = 0; start < n_in; start += SEL_THREADS) {
            const int i = start + tid;
            bool keep = false;
            float x = 0.f, y = 0.f;
            if (i < n_in && stt[i]) {
                x = src[2 * i]; y = src[2 * i + 1];
                const int ix = __float2int_rn(x), iy = __float2int_rn(y);
                keep = ix >= 1 && ix < p.W - 1 && iy >= 1 && iy < p.H - 1;
            }
            const u64 m = __ballot(keep);
            const int pre = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_scan[0][wv] = (u32)__popcll(m);
            __syncthreads();
            int off = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < SEL_THREADS / 64; k++) { const int c = (int)s_scan[0][k]; if (k < wv) off += c; tot += c; }
            __syncthreads();
            if (keep) { dst[2 * (base + off + pre)] = x; dst[2 * (base + off + pre) + 1] = y; }
            base += tot;
        }
        ns = base;
        if (tid == 0 && p.cmp_n_out) p.cmp_n_out[b] = ns;
        __syncthreads(); /* the gathered points are read back below by other threads */
    }
    const int limit = max(0, p.max_corners - ns);
    if (limit == 0) { /* nothing to add: the frame keeps its tracked points */
        if (tid == 0) { p.counts[b] = 0; if (p.append_n) p.append_n[b] = ns; }
        return;
    }
    /* Tracked points occupy the min-distance grid first.  A candidate within min_distance of one of
     * them is rejected whatever its rank, so when the grid of tracked points fits in LDS (it aliases
     * the sort buffer, which is not in use yet) all candidates are tested against it in parallel
     * up front and the blocked ones are cleared; the ordered walk below then only has to consider
     * the survivors and the corners it accepts itself.  Otherwise the tracked points go into the
     * global grid and are met during the walk. */
    const int cells = p.gw * p.gh;
    const bool lds_seeds = p.use_grid && ns > 0 && cells <= SEED_CELLS && ns <= SEED_NODES;
    /* The corners accepted below go into a grid in LDS when it fits (the walk is a chain of dependent lookups,
     * each a round trip to L2 otherwise); tracked points that were not filtered up front stay in the global grid
     * and are checked there as well. */
    const bool lds_walk = p.use_grid && cells <= WALK_CELLS && limit <= WALK_NODES;
    const bool global_grid = p.use_grid && (!lds_walk || (ns > 0 && !lds_seeds));
    {
        /* pack the keys above the threshold: a reservation per wave and slot, order does not matter */
        const int lane = tid & 63;
        __syncthreads(); /* s_cnt is zero */
#pragma unroll
        for (int j = 0; j < KEY_REGS; j++) {
            const bool valid = rk[j] > lowkey;
            const u64 bm = __ballot(valid);
            if (bm == 0) continue;
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt, (u32)__popcll(bm));
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            if (valid) s_pool[base + __popcll(bm & ((1ull << lane) - 1ull))] = rk[j];
        }
        __syncthreads();
        n_pool = s_cnt;
#pragma unroll
        for (int j = 0; j < KEY_REGS; j++) rk[j] = (u32)tid + j * SEL_THREADS < n_pool ? s_pool[tid + j * SEL_THREADS] : 0ull;
        __syncthreads();
        if (tid == 0) s_cnt = 0;
    }
    if (lds_walk)
        for (int c = tid; c < cells; c += SEL_THREADS) w_heads[c] = 0;
    /* LDS grid of the tracked points (aliases the sort buffer).  A cell word holds the head of the cell's chain in its low
     * half (node + 1, 0 = empty) and, in its high half, nine bits that say which cells of its 3x3 neighbourhood hold any
     * tracked point (bit 3 (dy + 1) + dx + 1): a candidate reads its own cell's word and then only the occupied
     * neighbours -- none at all for a candidate with no tracked point near it. */
    u32* g_cell = reinterpret_cast<u32*>(s_keys);                  /* [SEED_CELLS] */
    u64* g_node = reinterpret_cast<u64*>(g_cell + SEED_CELLS);     /* [SEED_NODES] x | y << 16 | (u64)next << 32 */
    static_assert(SEED_CELLS * sizeof(u32) + SEED_NODES * sizeof(u64) <= sizeof(u64) * SEL_M, "the grid aliases the sort buffer");
    static_assert(SEED_NODES % SEL_THREADS == 0 && SEED_NODES < 65536, "node numbers fit the low half of a cell word");
    auto build_track_grid = [&]() { /* the caller puts a barrier behind it */
        const float* sd = p.seeds + (size_t)b * p.seed_cap * 2;
        for (int c = tid; c < cells; c += SEL_THREADS) g_cell[c] = 0;
        __syncthreads();
        u32 sq[SEED_NODES / SEL_THREADS]; /* this thread's points, ~0 = none */
#pragma unroll
        for (int j = 0; j < SEED_NODES / SEL_THREADS; j++) {
            const int s = tid + j * SEL_THREADS;
            sq[j] = ~0u;
            if (s >= ns) continue;
            const int sx = __float2int_rn(sd[2 * s]), sy = __float2int_rn(sd[2 * s + 1]);
            if (sx < 0 || sy < 0 || sx >= p.W || sy >= p.H) continue;
            sq[j] = (u32)sx | ((u32)sy << 16);
            const int c = cell_of(sy, p.cell, p.cell_magic) * p.gw + cell_of(sx, p.cell, p.cell_magic);
            g_node[s] = (u64)sq[j] | ((u64)atomicExch(&g_cell[c], (u32)s + 1) << 32); /* the high halves are still zero */
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SEED_NODES / SEL_THREADS; j++) {
            if (sq[j] == ~0u) continue;
            const int xc = cell_of((int)(sq[j] & 0xFFFFu), p.cell, p.cell_magic), yc = cell_of((int)(sq[j] >> 16), p.cell, p.cell_magic);
#pragma unroll
            for (int c = 0; c < 9; c++) {
                const int yy = yc - 1 + c / 3, xx = xc - 1 + c % 3;
                /* seen from the cell (yy, xx) this point's cell is the neighbour 8 - c */
                if (yy >= 0 && yy < p.gh && xx >= 0 && xx < p.gw) atomicOr(&g_cell[yy * p.gw + xx], 1u << (16 + 8 - c));
            }
        }
    };
    if (p.use_grid && ns > 0) {
        const float* sd = p.seeds + (size_t)b * p.seed_cap * 2;
        if (lds_seeds) {
            build_track_grid();
        } else {
            for (int s = tid; s < ns; s += SEL_THREADS) {
                const int sx = __float2int_rn(sd[2 * s]), sy = __float2int_rn(sd[2 * s + 1]);
                if (sx < 0 || sy < 0 || sx >= p.W || sy >= p.H) continue;
                const int c = cell_of(sy, p.cell, p.cell_magic) * p.gw + cell_of(sx, p.cell, p.cell_magic);
                st_agent(&node_xy[s], (u32)sx | ((u32)sy << 16));
                const u32 old = atomicExch(&heads[c], (u32)s + 1);
                st_agent(&node_next[s], old);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();

    /* one candidate against the grid of tracked points: its own cell's word, then the chains of the occupied neighbours */
    auto blocked_by_track = [&](u64 k) {
        const int y = (int)(((u32)k) >> 16), x = (int)(((u32)k) & 0xFFFFu);
        const int xc = cell_of(x, p.cell, p.cell_magic), yc = cell_of(y, p.cell, p.cell_magic);
        const int at = yc * p.gw + xc;
        const u32 own = g_cell[at];
        for (u32 m = own >> 16; m; m &= m - 1) {
            const int c = __builtin_ctz(m);
            u32 n = (c == 4 ? own : g_cell[at + (c / 3 - 1) * p.gw + c % 3 - 1]) & 0xFFFFu;
            while (n) {
                const u64 nd = g_node[n - 1];
                const int dx = x - (int)((u32)nd & 0xFFFFu), dy = y - (int)((u32)nd >> 16);
                if ((u32)(dx * dx + dy * dy) < p.md2i) return true;
                n = (u32)(nd >> 32);
            }
        }
        return false;
    };
    __shared__ u32 s_total;
    auto block_total = [&](u32 v) { /* sum over the workgroup, in every thread */
        const u32 t = block_suffix_scan(v, s_scan); /* the sum, at thread 0 */
        if (tid == 0) s_total = t;
        __syncthreads();
        const u32 r = s_total;
        __syncthreads();
        return r;
    };
    /* clears the keys in (above, below) that a tracked point blocks; returns how many are left there */
    auto test_against_tracks = [&](u64 above, u64 below) {
        u32 left = 0;
        for_each_key([&](u64 k) {
            if (!(k > above && k < below)) return false;
            const bool blocked = blocked_by_track(k);
            if (!blocked) left++;
            return blocked;
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return block_total(left);
    };
    /* MSD radix select: the lowest T such that at most m keys lie in [T, below), among the keys above `above` (T > above) */
    auto radix_select = [&](u64 above, u64 below, u32 m) {
        /* keys are compared on bits [hi_shift, 64) against prefix; valid keys lie between the threshold and the
         * frame maximum, so the bits those two share need no pass */
        const u32 diff = __float_as_uint(thr) ^ __float_as_uint(maxv);
        int hi_shift = diff ? 64 - __clz((int)diff) : 32;
        u64 prefix = hi_shift == 64 ? 0ull : (((u64)__float_as_uint(maxv) << 32) >> hi_shift);
        u64 T = 0;
        for (;;) {
            const int bits = hi_shift >= 11 ? 11 : hi_shift;
            const int shift = hi_shift - bits;
            const u32 nb = 1u << bits;
            for (u32 i = tid; i <= 2048; i += SEL_THREADS) s_hist[i] = 0;
            __syncthreads();
            for_each_key([&](u64 k) {
                if (k > above && k < below && (hi_shift == 64 || (k >> hi_shift) == prefix))
                    atomicAdd(&s_hist[(u32)(k >> shift) & (nb - 1)], 1u);
                return false;
            });
            __syncthreads();
            const u32 h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
            const u32 ss = block_suffix_scan(h0 + h1, s_scan);
            /* S(d) = number of keys in bins >= d */
            s_hist[2 * tid] = ss;
            s_hist[2 * tid + 1] = ss - h0;
            if (tid == 0) s_hist[2048] = 0;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const u32 d = 2 * tid + q;
                if (s_hist[d] <= m && (d == 0 || s_hist[d - 1] > m)) s_dmin = d;
            }
            if (tid == 0 && s_hist[2047] > m) s_dmin = 2048;
            __syncthreads();
            const u32 dmin = s_dmin;
            const u32 taken = s_hist[dmin];
            __syncthreads();
            if (taken > 0 || dmin == 0) {
                T = (hi_shift == 64 ? 0ull : (prefix << hi_shift)) | ((u64)dmin << shift);
                break;
            }
            /* bin dmin-1 alone holds more than m keys: refine inside it */
            prefix = (hi_shift == 64 ? 0ull : (prefix << bits)) | (u64)(dmin - 1);
            hi_shift = shift;
        }
        return T <= above ? above + 1 : T;
    };

    /* how many candidates pass the exact threshold and are not blocked by a tracked point */
    u32 total;
    if (lds_seeds) {
        total = test_against_tracks(lowkey, ~0ull);
    } else {
        u32 mine = 0;
        for_each_key([&](u64 k) { mine += (k > lowkey) ? 1u : 0u; return false; });
        total = block_total(mine);
    }

    u64 U = ~0ull; /* exclusive upper bound of keys not yet consumed */
    u32 consumed = 0;
    int n_acc = 0;
    /* a small refill rarely needs more than its few hundred best candidates: a quarter-size first batch sorts in a
     * third of the time, and a further batch follows whenever the first does not fill the budget */
    u32 m_batch = limit <= 64 ? SEL_M_FIRST / 4 : SEL_M_FIRST;
    while (consumed < total && n_acc < limit) {
        const u64 T = total - consumed <= m_batch ? lowkey + 1 : radix_select(lowkey, U, m_batch);
        /* collect [T, U) */
        for_each_key([&](u64 k) {
            if (k >= T && k < U) {
                const u32 pos = atomicAdd(&s_cnt, 1u);
                if (pos < SEL_M) s_keys[pos] = k;
            }
            return false;
        });
        __syncthreads();
        const u32 taken = min(s_cnt, (u32)SEL_M);
        u32 n2 = 64;
        while (n2 < taken) n2 <<= 1;
        for (u32 i = taken + tid; i < n2; i += SEL_THREADS) s_keys[i] = 0;
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        const u64* sorted = s_keys;
        if (n2 <= SEL_THREADS) {
            sorted = sort1024_desc(s_keys, n2); /* a key per thread, in registers */
        } else {
            /* bitonic sort in LDS, descending: one compare-exchange per thread and step (pair t = elements i and i + j,
             * i = t with a zero inserted at bit log2 j).  The 64 pairs of a wave stay inside one 128-element block as long
             * as j <= 64, and a wave's LDS accesses stay in order: only steps that cross blocks, and the last step of a
             * stage, need the workgroup barrier. */
            for (u32 k = 2; k <= n2; k <<= 1)
                for (u32 j = k >> 1; j > 0; j >>= 1) {
                    for (u32 t = tid; t < n2 / 2; t += SEL_THREADS) {
                        const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                        const u64 a = s_keys[i], c = s_keys[l];
                        const bool desc = (i & k) == 0;
                        if (desc ? (a < c) : (a > c)) { s_keys[i] = c; s_keys[l] = a; }
                    }
                    if (j > 64 || j == 1) __syncthreads();
                }
        }

        /* wave-parallel greedy walk over the sorted batch */
        if (tid < 64) {
            const int lane = tid;
            int acc = s_nacc;
            for (u32 base = 0; base < taken && acc < limit; base += 64) {
                const u32 i = base + lane;
                const bool valid = i < taken;
                const u64 key = valid ? sorted[i] : 0ull;
                const int y = (int)(((u32)key) >> 16), x = (int)(((u32)key) & 0xFFFFu);
                bool alive = valid;
                int xc = 0, yc = 0;
                if (p.use_grid) {
                    xc = cell_of(x, p.cell, p.cell_magic); yc = cell_of(y, p.cell, p.cell_magic);
                    const int x1 = max(0, xc - 1), x2 = min(p.gw - 1, xc + 1);
                    const int y1 = max(0, yc - 1), y2 = min(p.gh - 1, yc + 1);
                    if (alive && lds_walk) {
                        /* all nine cell heads first, then the (short) chains side by side */
                        u32 hd[9];
#pragma unroll
                        for (int c = 0; c < 9; c++) {
                            const int yy = yc - 1 + c / 3, xx = xc - 1 + c % 3;
                            hd[c] = (yy >= y1 && yy <= y2 && xx >= x1 && xx <= x2) ? w_heads[yy * p.gw + xx] : 0u;
                        }
                        bool more = true;
                        while (more && alive) {
                            more = false;
#pragma unroll
                            for (int c = 0; c < 9; c++)
                                if (hd[c]) {
                                    const u32 q = w_xy[hd[c] - 1];
                                    const u32 nx = w_next[hd[c] - 1];
                                    const int dx = x - (int)(q & 0xFFFF), dy = y - (int)(q >> 16);
                                    if ((u32)(dx * dx + dy * dy) < p.md2i) alive = false;
                                    hd[c] = nx;
                                    more = more || nx != 0;
                                }
                        }
                    }
                    if (alive && global_grid) {
                        for (int yy = y1; yy <= y2 && alive; yy++)
                            for (int xx = x1; xx <= x2 && alive; xx++) {
                                u32 n = ld_agent(&heads[yy * p.gw + xx]);
                                while (n) {
                                    const u32 q = ld_agent(&node_xy[n - 1]);
                                    const int dx = x - (int)(q & 0xFFFF), dy = y - (int)(q >> 16);
                                    if ((u32)(dx * dx + dy * dy) < p.md2i) { alive = false; break; }
                                    n = ld_agent(&node_next[n - 1]);
                                }
                            }
                    }
                }
                /* Serial part, scalar masks only: take the alive candidates in rank order; each one taken kills the
                 * lower-ranked ones within min_distance of it. */
                u64 alive_m = __ballot(alive), taken_mask = 0;
                int n_new = 0;
                while (alive_m && acc + n_new < limit) {
                    const int l = __ffsll((long long)alive_m) - 1; /* lanes below l are dead or taken */
                    taken_mask |= 1ull << l;
                    n_new++;
                    if (p.use_grid) {
                        const int ax = __builtin_amdgcn_readlane(x, l), ay = __builtin_amdgcn_readlane(y, l);
                        const int dx = x - ax, dy = y - ay;
                        alive_m &= ~__ballot((u32)(dx * dx + dy * dy) < p.md2i); /* includes lane l itself */
                    } else {
                        alive_m &= ~(1ull << l);
                    }
                }
                /* Parallel part: the taken candidates write their corner and enter the grid. */
                const bool took = (taken_mask >> lane) & 1ull;
                const int my = acc + __popcll(taken_mask & ((1ull << lane) - 1ull));
                if (took) {
                    float* c = p.corners + ((size_t)b * p.max_corners + my) * 2;
                    c[0] = (float)x; c[1] = (float)y;
                    if (p.scores) p.scores[(size_t)b * p.max_corners + my] = __uint_as_float((u32)(key >> 32));
                    if (p.append_pts) {
                        float* a = p.append_pts + ((size_t)b * p.seed_cap + ns + my) * 2;
                        a[0] = (float)x; a[1] = (float)y;
                    }
                }
                if (lds_walk) {
                    /* two taken candidates may share a cell: everyone writes the head, whoever reads its own node back
                     * has won and links to the head it saw before; the others go again */
                    /* (volatile: the read-back must come from LDS, where another lane's write may have landed on top
                     * of this lane's; the compiler would otherwise forward the value just stored) */
                    volatile unsigned short* vh = &w_heads[yc * p.gw + xc];
                    bool pending = took;
                    while (__ballot(pending)) {
                        unsigned short old = 0;
                        if (pending) old = *vh;
                        if (pending) *vh = (unsigned short)(my + 1);
                        if (pending && *vh == (unsigned short)(my + 1)) {
                            w_xy[my] = (u32)x | ((u32)y << 16);
                            w_next[my] = old;
                            pending = false;
                        }
                    }
                } else if (p.use_grid) {
                    u64 ins = taken_mask;
                    while (ins) { /* global grid: one at a time */
                        const int l = __ffsll((long long)ins) - 1;
                        ins &= ins - 1;
                        if (lane == l) {
                            const u32 n = (u32)(p.seed_cap + my);
                            u32* hd = &heads[yc * p.gw + xc];
                            st_agent(&node_xy[n], (u32)x | ((u32)y << 16));
                            st_agent(&node_next[n], ld_agent(hd));
                            st_agent(hd, n + 1);
                        }
                    }
                }
                acc += n_new;
                /* this chunk's insertions must be in L2 before the next chunk looks them up */
                if (global_grid) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (lane == 0) s_nacc = acc;
        }
        __syncthreads();
        n_acc = s_nacc;
        consumed += taken;
        U = T;
        m_batch = SEL_M;
    }
    if (tid == 0) { p.counts[b] = n_acc; if (p.append_n) p.append_n[b] = ns + n_acc; }
}

/* workspace layout (all offsets 256-byte aligned) */
struct GfWs {
    size_t hdr_off, heads_off, cand_off, nxy_off, nnext_off, total;
    size_t zero_bytes; /* [hdr_off, hdr_off + zero_bytes) is cleared every call */
    int cell, gw, gh, use_grid, node_cap;
    u32 seg_cap;
};

static GfWs gf_layout(int w, int h, int batch, int max_corners, int seed_cap, double min_distance) {
    GfWs L;
    L.use_grid = min_distance >= 1.0;
    L.cell = L.use_grid ? (int)lrint(min_distance) : 1;
    if (L.cell < 1) L.cell = 1;
    L.gw = L.use_grid ? (w + L.cell - 1) / L.cell : 1;
    L.gh = L.use_grid ? (h + L.cell - 1) / L.cell : 1;
    L.node_cap = seed_cap + max_corners;
    size_t o = 0;
    L.hdr_off = o; o += align_up((size_t)batch * (1 + GF_NSEG) * GF_CSTRIDE * sizeof(u32), 256);
    L.heads_off = o; o += align_up((size_t)batch * L.gw * L.gh * sizeof(u32), 256);
    L.zero_bytes = o;
    /* the selection keeps both min-distance grids in LDS when they fit, and then never touches the grid heads in memory:
     * only the counters need clearing (the same three conditions as in k_gftt_select, on the capacities) */
    if (!L.use_grid || (L.gw * L.gh <= SEED_CELLS && L.gw * L.gh <= WALK_CELLS && seed_cap <= SEED_NODES && max_corners <= WALK_NODES))
        L.zero_bytes = L.heads_off;
    {   /* a segment receives every GF_NSEG-th (strip, band) item; each can emit one key per pixel */
        const EigGeom g = eig_geom(w, h, batch);
        L.seg_cap = (u32)div_up(g.strips_x * g.bands_y, GF_NSEG) * (u32)(EV_OUT * g.band_h);
    }
    L.cand_off = o; o += align_up((size_t)batch * GF_NSEG * L.seg_cap * sizeof(u64), 256);
    L.nxy_off = o; o += align_up((size_t)batch * L.node_cap * sizeof(u32), 256);
    L.nnext_off = o; o += align_up((size_t)batch * L.node_cap * sizeof(u32), 256);
    L.total = o;
    return L;
}

extern "C" size_t vfe_good_features_workspace(int width, int height, int batch, int max_corners, int seed_cap) {
    if (width <= 0 || height <= 0 || batch <= 0 || max_corners <= 0 || seed_cap < 0) return 0;
    /* the grid is largest for min_distance = 1 (one cell per pixel) */
    return gf_layout(width, height, batch, max_corners, seed_cap, 1.0).total;
}

static int gf_check(const vfe_dimg* img, int max_corners, double quality, double min_distance, const float* seeds_xy,
                    const int* n_seeds, int* seed_cap, const float* corners_xy, const int* counts, const void* workspace,
                    size_t workspace_bytes, GfWs* L) {
    int rc = vfe_check_dimg(img, "img");
    if (rc) return rc;
    VFE_REQUIRE(img->width >= 8 && img->height >= 8, "image must be at least 8x8");
    VFE_REQUIRE((size_t)img->width * img->height < (1u << 31), "image too large");
    VFE_REQUIRE(img->width < 65536 && img->height < 65536, "image too large");
    VFE_REQUIRE(max_corners > 0, "max_corners must be > 0");
    VFE_REQUIRE(quality > 0 && quality < 1e6, "quality out of range");
    VFE_REQUIRE(min_distance >= 0 && min_distance < 32768, "min_distance out of range");
    VFE_REQUIRE(corners_xy && counts && workspace, "NULL output or workspace");
    VFE_REQUIRE(*seed_cap >= 0 && (*seed_cap == 0 || !seeds_xy == !n_seeds), "seeds_xy and n_seeds go together");
    VFE_REQUIRE(img->batch <= 65535, "batch too large");
    if (!seeds_xy) *seed_cap = 0;
    *L = gf_layout(img->width, img->height, img->batch, max_corners, *seed_cap, min_distance);
    if (workspace_bytes < L->total) {
        vfe_set_error("workspace too small: need %zu bytes, got %zu", L->total, workspace_bytes);
        return VFE_E_NOMEM;
    }
    VFE_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    return VFE_OK;
}

/* first half of the detector: everything that depends on the image only (clears the counters and grid heads,
 * eigenvalue map + candidates).  May run on another stream than the second half. */
int gf_eig_phase(const vfe_dimg* img, int max_corners, double quality, double min_distance, int seed_cap, void* workspace,
                 size_t workspace_bytes, hipStream_t st, int wgs_per_cu) {
    GfWs L;
    static const float dummy_c = 0.f; static const int dummy_n = 0;
    int rc = gf_check(img, max_corners, quality, min_distance, seed_cap ? &dummy_c : nullptr, seed_cap ? &dummy_n : nullptr, &seed_cap,
                      &dummy_c, &dummy_n, workspace, workspace_bytes, &L);
    if (rc) return rc;
    char* ws = (char*)workspace;
    VFE_HIP(hipMemsetAsync(ws + L.hdr_off, 0, L.zero_bytes, st));
    u32* frame_max = (u32*)(ws + L.hdr_off);
    u32* cand_count = frame_max + (size_t)img->batch * GF_CSTRIDE;
    {
        ProfScope prof(VFE_K_GFTT_EIG, st);
        gftt_eig_launch(to_dimg(img), nullptr, frame_max, cand_count, (u64*)(ws + L.cand_off), L.seg_cap, quality, st, wgs_per_cu);
    }
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

/* second half: ordered selection against the tracked points */
int gf_select_phase(const vfe_dimg* img, int max_corners, double quality, double min_distance,
                    const float* seeds_xy, const int* n_seeds, int seed_cap,
                    float* corners_xy, float* scores, int* counts, void* workspace, size_t workspace_bytes, hipStream_t st,
                    bool append, const float* cmp_src, const uint8_t* cmp_status, int* cmp_n_out) {
    GfWs L;
    int rc = gf_check(img, max_corners, quality, min_distance, seeds_xy, n_seeds, &seed_cap, corners_xy, counts, workspace,
                      workspace_bytes, &L);
    if (rc) return rc;
    char* ws = (char*)workspace;
    u32* frame_max = (u32*)(ws + L.hdr_off);
    u32* cand_count = frame_max + (size_t)img->batch * GF_CSTRIDE;
    SelParams p;
    p.W = img->width; p.H = img->height; p.max_corners = max_corners; p.quality = quality;
    p.use_grid = L.use_grid; p.cell = L.cell; p.gw = L.gw; p.gh = L.gh;
    p.cell_magic = L.cell > 1 ? (u32)((0x100000000ull + (u64)L.cell - 1) / (u64)L.cell) : 0u;
    p.md2i = (u32)ceil(min_distance * min_distance);
    p.frame_max = frame_max; p.cand_count = cand_count; p.cand = (u64*)(ws + L.cand_off); p.seg_cap = L.seg_cap;
    p.heads = (u32*)(ws + L.heads_off); p.node_xy = (u32*)(ws + L.nxy_off); p.node_next = (u32*)(ws + L.nnext_off);
    p.node_cap = L.node_cap;
    p.seeds = seeds_xy; p.n_seeds = n_seeds; p.seed_cap = seed_cap;
    p.corners = corners_xy; p.scores = scores; p.counts = counts;
    p.append_pts = nullptr; p.append_n = nullptr;
    if (append) {
        /* the accepted corners fit behind the seeds: at most max_corners - n of them are taken */
        VFE_REQUIRE(seeds_xy && n_seeds && seed_cap == max_corners, "append mode needs the seed arrays with max_corners slots per frame");
        p.append_pts = const_cast<float*>(seeds_xy); p.append_n = const_cast<int*>(n_seeds);
    }
    VFE_REQUIRE(!cmp_src || (append && cmp_status && cmp_src != seeds_xy), "compaction in front needs append mode and its own source array");
    p.cmp_src = cmp_src; p.cmp_status = cmp_status; p.cmp_n_out = cmp_n_out;
    {
        ProfScope prof(VFE_K_GFTT_SELECT, st);
        hipLaunchKernelGGL(k_gftt_select, dim3(img->batch), dim3(SEL_THREADS), 0, st, p);
    }
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}

extern "C" int vfe_dev_good_features(const vfe_dimg* img, int max_corners, double quality, double min_distance,
                                     const float* seeds_xy, const int* n_seeds, int seed_cap,
                                     float* corners_xy, float* scores, int* counts,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    GfWs L;
    int sc = seed_cap;
    int rc = gf_check(img, max_corners, quality, min_distance, seeds_xy, n_seeds, &sc, corners_xy, counts, workspace, workspace_bytes, &L);
    if (rc) return rc;
    if ((rc = gf_eig_phase(img, max_corners, quality, min_distance, sc, workspace, workspace_bytes, (hipStream_t)stream))) return rc;
    return gf_select_phase(img, max_corners, quality, min_distance, seeds_xy, n_seeds, sc, corners_xy, scores, counts, workspace,
                           workspace_bytes, (hipStream_t)stream, false, nullptr, nullptr, nullptr);
}

extern "C" int vfe_dev_min_eigen_map(const vfe_dimg* img, float* eig, void* stream) {
    int rc = vfe_check_dimg(img, "img");
    if (rc) return rc;
    VFE_REQUIRE(img->width >= 8 && img->height >= 8, "image must be at least 8x8");
    VFE_REQUIRE(eig, "eig is NULL");
    DImg d = to_dimg(img);
    ProfScope prof(VFE_K_MISC, (hipStream_t)stream);
    gftt_eig_launch(d, eig, nullptr, nullptr, nullptr, 0u, 0.0, (hipStream_t)stream);
    VFE_LAUNCH_CHECK();
    return VFE_OK;
}
