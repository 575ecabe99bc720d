/* Internal interface of the FAST detector (shared with ORB). */
#ifndef VFE_FAST_H
#define VFE_FAST_H
#include "vfe_common.h"

/* streaming kernel for whole frames (vfe_fast_stream.hip): 248 output columns per wave strip, one slot of
 * FS_ROWCAP keypoints per (row, strip) segment (suppression leaves at most 124) */
#define FS_OUT 248
#define FS_ROWCAP 128
int fast_stream_launch(const DImg& d, int threshold, int border, uint8_t* rowcnt, unsigned* tmp, hipStream_t st);
int fast_scatter_stream_launch(const DImg& d, const uint8_t* rowcnt, const unsigned* rowoff, const unsigned* tmp, int* kps, int cap,
                               hipStream_t st);

struct FastWs {
    int tiles_x, tiles_y;
    size_t rowcnt_off, rowoff_off, tmp_off, total;
};
FastWs fast_layout(int w, int h, int batch);
/* keypoints (x, y, score) of every frame in raster order into kps[b][cap][3]; counts[b] (clamped
 * to cap); only keypoints inside [border, dim - border) are kept */
int fast_detect(const DImg& d, int threshold, int border, int* kps, int cap, int* counts, char* ws, hipStream_t st);
#endif
