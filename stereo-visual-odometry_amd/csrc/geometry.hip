// geometry.hip -- placeholder until the triangulation / PnP kernels land (next commit).
#include "svo_ctx.h"
namespace svo {
int geom_workspace_bytes(const svo_config &, int, size_t *bytes) { *bytes = 256; return SVO_OK; }
int stage_triangulate(svo_ctx *ctx, const double *, const double *, const svo_pt2f *, const svo_pt2f *, int,
                      svo_pt3f *, int) { ctx->err = "not implemented"; return SVO_ERR_STATE; }
int stage_pnp_ransac(svo_ctx *ctx, const svo_pt3f *, const svo_pt2f *, int, const double *, int, float, double,
                     svo_pnp_result *, uint8_t *, int) { ctx->err = "not implemented"; return SVO_ERR_STATE; }
int pipeline_add_frame(svo_ctx *ctx, const uint8_t *, const uint8_t *, int, int, svo_step_result *)
{ ctx->err = "not implemented"; return SVO_ERR_STATE; }
int pipeline_track_batch(svo_ctx *ctx, const uint8_t *, const uint8_t *, int, int64_t, int, const double *,
                         svo_step_result *, int) { ctx->err = "not implemented"; return SVO_ERR_STATE; }
}
