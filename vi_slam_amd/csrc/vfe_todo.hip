/* Entry points declared in include/vfe.h whose kernels are not written yet.  Each fails
 * loudly; none computes anything. */
#include "vfe_common.h"
#define TODO(name) do { vfe_set_error(name ": not implemented yet"); return VFE_E_INVALID; } while (0)
extern "C" {
int vfe_dev_fast_grid(const vfe_dimg*, int, int, int, int, int, const float*, const int*, int, float*, int*, int, int*, void*, size_t, void*) { TODO("vfe_dev_fast_grid"); }
int vfe_create(const vfe_config*, vfe_ctx**) { TODO("vfe_create"); }
void vfe_destroy(vfe_ctx*) {}
int vfe_reset(vfe_ctx*) { TODO("vfe_reset"); }
int vfe_detect_shi_tomasi(vfe_ctx*, const uint8_t*, int, int, size_t, int, int, double, double, float*, int*) { TODO("vfe_detect_shi_tomasi"); }
int vfe_klt_track(vfe_ctx*, const uint8_t*, const uint8_t*, int, int, size_t, const float*, int, float*, uint8_t*, float*, int, int, double, double) { TODO("vfe_klt_track"); }
int vfe_orb_extract(vfe_ctx*, const uint8_t*, int, int, size_t, int, int, float, int, int, vfe_orb_kp*, uint8_t*, int, int*) { TODO("vfe_orb_extract"); }
int vfe_match_knn2(vfe_ctx*, const uint8_t*, int, const uint8_t*, int, float, int*, int*, uint8_t*) { TODO("vfe_match_knn2"); }
int vfe_process_frame(vfe_ctx*, const uint8_t*, int, int, size_t, int, int64_t, int, double, double, vfe_frame_stats*) { TODO("vfe_process_frame"); }
int vfe_get_points(vfe_ctx*, float*, int, int*) { TODO("vfe_get_points"); }
}
