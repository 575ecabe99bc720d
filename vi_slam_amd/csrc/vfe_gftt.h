/* Shared between the Shi-Tomasi eigenvalue kernel and the corner selection. */
#pragma once
#include "vfe_common.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define GF_MAX_WGS 2048u /* persistent workgroups: 256 CUs x 8 */
#define GF_CSTRIDE 64    /* u32 stride between counters: one 256-B line each, so atomics on
                            different frames / segments go to different L2 channels */
#define GF_NSEG 8        /* candidate lists per frame (spreads the reservation atomics) */
#define EV_OUT 240       /* output columns per wave strip: 60 lanes x 4 pixels (4 halo lanes) */

/* decomposition of a batch into (strip, band, frame) work items, one per wave */
struct EigGeom { int strips_x, band_h, bands_y; unsigned total; };
static inline EigGeom eig_geom(int w, int h, int batch) {
    EigGeom g;
    g.strips_x = div_up(w, EV_OUT);
    g.band_h = 64;
    while (g.band_h > 16 && (long long)g.strips_x * div_up(h, g.band_h) * batch < 16384) g.band_h /= 2;
    g.bands_y = div_up(h, g.band_h);
    g.total = (unsigned)(g.strips_x * g.bands_y * batch);
    return g;
}

/* Enqueue the eigenvalue pass.  eigmap != NULL: write the float map only.  Otherwise: frame maxima
 * (as float bits, stride GF_CSTRIDE) and the candidate keys (score bits << 32 | y << 16 | x) of the 3x3
 * local maxima, GF_NSEG lists of seg_cap keys per frame with their counters after the maxima. */
void gftt_eig_launch(const DImg& d, float* eigmap, u32* frame_max, u32* cand_count, u64* cand, u32 seg_cap,
                     double quality, hipStream_t st, int wgs_per_cu = 0);

/* How the eigenvalue pass should share the chip when it runs beside the tracker (front-end step).  Its workgroups hold a
 * quarter of a SIMD's registers per wave; alone in front of the queue it fills every SIMD, and the tracker's waves (and the
 * short pyramid launches in front of them) only start as its waves retire.  A long pass overlaps the whole tracker anyway;
 * a shorter one is better held to 3 or 2 workgroups per CU (requested as spare LDS per workgroup), and a short one is
 * better started behind the pyramid.  Measured on 1080p front-end steps, 8 to 256 streams (8640 strip rows per frame):
 * 256 streams uncapped, 128 three workgroups per CU (+3 %), 64 two (+7 %), 32 and fewer behind the pyramid (+4 .. +14 %);
 * up to four streams the pass is the longest thing in the step and goes first again (1 / 2 / 4 streams: 0.114 / 0.124 /
 * 0.144 ms per step behind the pyramid, 0.099 / 0.105 / 0.124 in front of it; 8 streams 0.185 against 0.193).
 * Where the pyramid is one launch of eight-wave workgroups (`pyramid_one_launch`), an uncapped pass in front of it leaves
 * no CU with room for such a workgroup until its own grid has been handed out: 256 streams 1.70 ms per step uncapped,
 * 1.61 ms with three workgroups per CU (behind the pyramid: 1.66). */
struct EigBeside { int wgs_per_cu; bool after_pyramid; };
/* strip rows of a launch: the measure of its length (the number of waves is not -- the band height adapts to the batch) */
static inline long long eig_strip_rows(int w, int h, int batch) { return (long long)div_up(w, EV_OUT) * h * batch; }
static inline EigBeside eig_beside_policy(int w, int h, int batch, bool pyramid_one_launch) {
    const long long rows = eig_strip_rows(w, h, batch); /* 8640 per 1080p frame */
    if (rows >= 1600000) return {pyramid_one_launch ? 3 : 0, false};
    if (rows >= 800000) return {3, false};
    if (rows >= 400000) return {2, false};
    if (rows >= 43200) return {0, true};
    return {0, false};
}

/* the two halves of vfe_dev_good_features: the first depends on the image only and may be enqueued on another
 * stream, the second needs the tracked points; same workspace, same parameters.  `append` (seed arrays with max_corners
 * slots per frame): the accepted corners are also written behind each frame's seeds and n_seeds grows by their number,
 * i.e. the selection does what vfe_dev_append_points would do next */
int gf_eig_phase(const vfe_dimg* img, int max_corners, double quality, double min_distance, int seed_cap, void* workspace,
                 size_t workspace_bytes, hipStream_t st, int wgs_per_cu = 0);
int gf_select_phase(const vfe_dimg* img, int max_corners, double quality, double min_distance,
                    const float* seeds_xy, const int* n_seeds, int seed_cap,
                    float* corners_xy, float* scores, int* counts, void* workspace, size_t workspace_bytes, hipStream_t st,
                    bool append, const float* cmp_src = nullptr, const uint8_t* cmp_status = nullptr, int* cmp_n_out = nullptr);
/* cmp_src / cmp_status (append mode): the selection first gathers the frame's surviving tracks itself -- those of
 * cmp_src[b][0 .. n_seeds[b]) with status set and a rounded position inside the image less a one-pixel rim, in order -- into
 * seeds_xy, and reports their number in cmp_n_out (may be NULL): what vfe_dev_compact_tracks would do in a launch of its own */
