// Entry points whose kernels are not written yet: they fail loudly (PVLM_ERR_STATE), never fall
// back to a CPU path.  Each is replaced by its real implementation file as it lands.
#include "pvlm_internal.h"

#define PVLM_TODO(ctx, name)                                   \
  do {                                                         \
    if (ctx) PVLM_SET_ERR(ctx, name " is not implemented yet"); \
    return PVLM_ERR_STATE;                                     \
  } while (0)

extern "C" {
pvlm_status pvlm_scan_upload(pvlm_ctx* ctx, const pvlm_scan_desc*, pvlm_scan**) { PVLM_TODO(ctx, "pvlm_scan_upload"); }
pvlm_status pvlm_scan_destroy(pvlm_ctx* ctx, pvlm_scan*) { PVLM_TODO(ctx, "pvlm_scan_destroy"); }
pvlm_status pvlm_knn(pvlm_ctx* ctx, const pvlm_scan*, int, const float*, int, int, float, int32_t*, float*) { PVLM_TODO(ctx, "pvlm_knn"); }
pvlm_status pvlm_assoc_point2plane(pvlm_ctx* ctx, int, pvlm_scan* const*, pvlm_scan* const*, double, float, pvlm_functor, unsigned, double, pvlm_resset**) { PVLM_TODO(ctx, "pvlm_assoc_point2plane"); }
pvlm_status pvlm_assoc_point2plane_debug(pvlm_ctx* ctx, const pvlm_resset*, int32_t*, int32_t*) { PVLM_TODO(ctx, "pvlm_assoc_point2plane_debug"); }
pvlm_status pvlm_line2line_votes(pvlm_ctx* ctx, const pvlm_scan*, const pvlm_scan*, float, int32_t*) { PVLM_TODO(ctx, "pvlm_line2line_votes"); }
pvlm_status pvlm_cam_to_image_f32(pvlm_ctx* ctx, int, int, int64_t, const float*, float*) { PVLM_TODO(ctx, "pvlm_cam_to_image_f32"); }
pvlm_status pvlm_cam_to_image_f64(pvlm_ctx* ctx, int, int, int64_t, const double*, double*) { PVLM_TODO(ctx, "pvlm_cam_to_image_f64"); }
pvlm_status pvlm_image_to_cam_f32(pvlm_ctx* ctx, int, int, int64_t, const float*, float, float*) { PVLM_TODO(ctx, "pvlm_image_to_cam_f32"); }
pvlm_status pvlm_image_to_cam_f64(pvlm_ctx* ctx, int, int, int64_t, const double*, double, double*) { PVLM_TODO(ctx, "pvlm_image_to_cam_f64"); }
pvlm_status pvlm_cam_lidar_votes(pvlm_ctx* ctx, int, int, const float*, int, const pvlm_scan*, const double*, int32_t*) { PVLM_TODO(ctx, "pvlm_cam_lidar_votes"); }
}
