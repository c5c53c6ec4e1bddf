// placeholder; replaced below
#include "render_common.hpp"
using namespace vamp;
extern "C" {
int vamp_render_camera_backward(const VampRenderDesc* d, const float* geom, const float* mats,
                                const float* us, const float* vs, const float* ds,
                                const float* mids, const float* beta,
                                const void* density_feature, const void* semantic,
                                const void* rgb, const float* g_rgb, const float* g_seg,
                                const float* g_depth, float* grad_density_feature,
                                float* grad_semantic, float* grad_rgb, float* grad_beta,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return fail(VAMP_EINVAL, "%s: not implemented", __func__);
}
int vamp_render_bev_backward(const VampRenderDesc* d, const float* oxs, const float* oys,
                             const float* ozs, const float* bev_mids, const float* beta,
                             const void* density_feature, const void* semantic,
                             const void* rgb, const void* base, const float* g_bev_rgb,
                             const float* g_bev_seg, const float* g_bev_height,
                             const float* g_voxel_density, const float* g_voxel_output,
                             float* grad_density_feature, float* grad_semantic,
                             float* grad_rgb, float* grad_base, float* grad_beta,
                             void* stream) {
  return fail(VAMP_EINVAL, "%s: not implemented", __func__);
}
}
