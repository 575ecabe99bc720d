/* Internal interface of the FAST detector (shared with ORB). */
#ifndef VFE_FAST_H
#define VFE_FAST_H
#include "vfe_common.h"

struct FastWs {
    int tiles_x, tiles_y;
    size_t rowcnt_off, rowoff_off, tmp_off, total;
};
FastWs fast_layout(int w, int h, int batch);
/* keypoints (x, y, score) of every frame in raster order into kps[b][cap][3]; counts[b] (clamped
 * to cap); only keypoints inside [border, dim - border) are kept */
int fast_detect(const DImg& d, int threshold, int border, int* kps, int cap, int* counts, char* ws, hipStream_t st);
#endif
