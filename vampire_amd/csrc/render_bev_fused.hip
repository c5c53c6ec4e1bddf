// BEV (top-down) branch of the renderer, forward, as ONE kernel: base_vampire2.py:408-418, 442-461.
//
// render_bev.hip's forward is two launches -- bev_density (a thread per column: 625 waves on 1024
// SIMDs, pure latency) and bev_channels (a thread per column and channel pair, which reads
// voxel_density back and redoes the weights per pair).  Here a workgroup owns 64 consecutive
// columns of the flattened (y, x) det lattice and NWV waves:
//
//   density   the waves split the volume planes the heights touch (bilinear (y, x) values -> LDS), then
//             the oZ heights: trilinear density sample, sigma_j -> LDS, voxel_density (and the raw
//             sample for the backward), then the compositing weights w_j of the same heights -> LDS
//   channels  a wave owns every (NWV * groups)-th channel: the x-pair loads of a chunk of volume planes
//             are issued together through buffer descriptors (channel and plane offsets are scalars)
//             one chunk ahead of their use; consecutive heights share a plane; semantic / rgb are
//             composited, base passes through to voxel_output
//
// Every global access is coalesced along x (lanes = consecutive columns).  HBM-bound streaming
// with a short per-column scan: no MFMA.
#include "render_bev_fused_dev.hpp"

namespace vamp {

template <typename T, int NWV>
__global__ void __launch_bounds__(NWV * 64)
bev_fwd_fused_kernel(RenderParams P, int NPA, const float* __restrict__ oxs, const float* __restrict__ oys,
                     const float* __restrict__ ozs, const float* __restrict__ bev_mids,
                     const float* __restrict__ beta_raw, const T* __restrict__ dens,
                     const T* __restrict__ sem, const T* __restrict__ rgb, const T* __restrict__ base,
                     float* __restrict__ bev_rgb, float* __restrict__ bev_seg, float* __restrict__ bev_height,
                     float* __restrict__ voxel_density, float* __restrict__ voxel_output,
                     float* __restrict__ s0_save, float* __restrict__ ss_save) {
  bev_fwd_fused_block<T, NWV>(blockIdx.x, (int) blockIdx.y, (int) blockIdx.z, (int) gridDim.z, P, NPA, oxs, oys, ozs, bev_mids,
                              beta_raw, dens, sem, rgb, base, bev_rgb, bev_seg, bev_height, voxel_density, voxel_output,
                              s0_save, ss_save);
}

#ifdef VAMP_BEVF_STAMPS
extern "C" int vamp_debug_bevf_stamps(long long* host, size_t n) {
  return (int) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bevf_stamps), n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

bool bev_fwd_fused_supported(const VampRenderDesc* d) {
  const size_t V = (size_t) d->Z * d->Y * d->X;
  const size_t es = d->in_dtype == VAMP_F32 ? 4 : 2;
  const size_t cmax = (size_t) (d->K > d->C ? d->K : d->C);
  const size_t CO = (size_t) d->C + (d->cat_seg ? d->K : 0);
  const size_t omax = (CO > (size_t) d->K + 3 ? CO : (size_t) d->K + 3) * d->oZ * d->oY * d->oX * 4;
  return d->oZ <= kFusedMaxOZ && d->X >= 2 && (cmax > 3 ? cmax : 3) * V * es < 0x7fffffffull && omax < 0x7fffffffull &&
         bev_planes_alloc(d) <= kFusedMaxNP;
}

int launch_bev_fwd_fused(const VampRenderDesc* d, const RenderParams& P, const float* oxs, const float* oys,
                         const float* ozs, const float* bev_mids, const float* beta, const void* dens,
                         const void* sem, const void* rgb, const void* base, float* bev_rgb, float* bev_seg,
                         float* bev_height, float* voxel_density, float* voxel_output, float* s0_save,
                         float* ss_save, hipStream_t s) {
  constexpr int NWV = VAMP_BEVF_NWV;
  const int np = bev_planes_alloc(d);
  const long cols = (long) P.oY * P.oX;
  const dim3 grid((unsigned) (((cols + 63) / 64 + 7) / 8 * 8), (unsigned) P.B, (unsigned) (P.C > 0 ? VAMP_BEVF_PARTS : 1));
  const size_t dyn = bev_fused_dyn_bytes(P.oZ, np);
#define VAMP_BEVFU(T)                                                                                     \
  VAMP_TIMED(kProfBevFwdCh, s, (bev_fwd_fused_kernel<T, NWV><<<grid, NWV * 64, dyn, s>>>(                 \
      P, np, oxs, oys, ozs, bev_mids, beta, static_cast<const T*>(dens), static_cast<const T*>(sem), \
      static_cast<const T*>(rgb), static_cast<const T*>(base), bev_rgb, bev_seg, bev_height,              \
      voxel_density, voxel_output, s0_save, ss_save)))
  if (d->in_dtype == VAMP_F32) VAMP_BEVFU(float);
  else VAMP_BEVFU(__hip_bfloat16);
#undef VAMP_BEVFU
  return check_launch("bev_fwd_fused_kernel");
}

}  // namespace vamp
