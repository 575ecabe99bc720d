/* Internal interface of the FAST detector (shared with ORB). */
#ifndef VFE_FAST_H
#define VFE_FAST_H
#include "vfe_common.h"

/* streaming kernels for whole frames (vfe_fast_stream.hip): 248 output columns per wave strip, one slot of
 * FS_ROWCAP keypoints per (row, strip) segment (suppression leaves at most 124).  A set of images of
 * different sizes, all with the same number of frames, is processed by one launch of each kernel. */
#define FS_OUT 248
#define FS_ROWCAP 128
#define FS_HEAD 4 /* a segment's first keypoints are staged in a compact array (most segments hold 0..3): see k_fast_scatter_stream */
#define FAST_MAX_LEVELS 12
struct FastImg {
    const uint8_t* data;
    int w, h, pitch, aligned; /* aligned: base, pitch and frame stride are multiples of 4 */
    size_t fstride;
};
struct FastSet {
    int n;                                   /* images in the set: caller fills n, img, kps, counts, cap */
    FastImg img[FAST_MAX_LEVELS];
    int* kps[FAST_MAX_LEVELS];               /* [batch][cap][3] */
    int* counts[FAST_MAX_LEVELS];            /* [batch] */
    int cap[FAST_MAX_LEVELS];
    int band_h, strips_x[FAST_MAX_LEVELS], bands_y[FAST_MAX_LEVELS];
    int batch;                               /* frames per image (set by the launcher) */
    unsigned item_begin[FAST_MAX_LEVELS + 1]; /* first (frame, band, strip) work item of an image */
    unsigned seg_begin[FAST_MAX_LEVELS + 1];  /* first (frame, row, strip) segment of an image */
};
struct FastSetWs { size_t rowcnt_off, rowoff_off, head_off, tmp_off, total; };
FastImg fast_img(const DImg& d);
FastSetWs fast_set_layout(int n, const int* w, const int* h, int batch);
/* FAST on every image of the set (each at least 16x16); results as fast_detect's */
int fast_detect_set(FastSet& S, int batch, int threshold, int border, char* ws, hipStream_t st);

struct FastWs {
    int tiles_x, tiles_y;
    size_t rowcnt_off, rowoff_off, tmp_off, total;
};
FastWs fast_layout(int w, int h, int batch);
/* keypoints (x, y, score) of every frame in raster order into kps[b][cap][3]; counts[b] (clamped
 * to cap); only keypoints inside [border, dim - border) are kept */
int fast_detect(const DImg& d, int threshold, int border, int* kps, int cap, int* counts, char* ws, hipStream_t st);
#endif
