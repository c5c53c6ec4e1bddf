// Camera branch of the renderer, forward, as ONE kernel that reads the reference-layout
// (channel-first) volumes directly: volume_rendering_from_multiple_views, bv2:396-440.
//
// It replaces three launches of render_fwd.hip -- pack_volume (a 117 MB re-layout), cam_term (the
// density-only pre-pass of early ray termination) and the planned march -- for callers that hand
// over [B,c,Z,Y,X] volumes and want the three camera maps:
//
//   plan     (ray_plan.hpp) which depth indices of the 8 x 8 ray tile can hold inside samples
//   density  the four waves share those depth indices; every inside sample takes its density
//            feature (4 x-pair loads from the 2.5 MB density volume) -> tau_i = sigma_i delta_i
//            into LDS, [depth index][ray]
//   scan     lanes = (ray, quarter of the depth range): exclusive prefix of tau along each ray,
//            compositing weights w_i = (1 - exp(-tau_i)) exp(-prefix) back into LDS, sum of
//            weights / expected depth, and the early-termination index keep[ray] (first sample in
//            front of which the optical depth has reached kTermOpticalDepth)
//   gather   the depth indices below the tile's largest keep are dealt to the four waves again;
//            every kept inside sample gathers its K + 3 composited channels -- per channel four
//            8-byte loads of the x-neighbour pair at (z, y), (z, y+1), (z+1, y), (z+1, y+1) -- and
//            adds w_i * s.  The weights are absolute, so there is no serial dependence between
//            samples and no exp() rescaling at the merge.
//   merge    partial sums of the four waves meet in LDS; coalesced stores of the maps
//
// Why channel-first works here although a lone ray would touch 22 x 8 lines per sample: the 64
// lanes of a wave are the 64 rays of the tile at ONE depth index, so for a fixed channel and
// (z, y) tap the wave's 64 pairs lie in a handful of rows of that channel's plane -- no more
// cache lines per wave instruction than the packed 96-byte rows cost, and the packed copy
// (24 us, 117 MB of traffic at cfg-B) disappears.  HBM/L2-bound gather + a short scan: no MFMA.
#include "render_cam_direct_dev.hpp"

namespace vamp {

template <typename T, int NCH, bool ERT, int NW>
__global__ void __launch_bounds__(NW * 64, ERT ? 4 : 3)   // (termination off: 171 registers would be 2 waves per SIMD; on: 126 with rounds of one index per wave)
cam_fwd_direct_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                      const float* __restrict__ vs, const float* __restrict__ ds,
                      const float* __restrict__ mids, const float* __restrict__ beta_raw,
                      const T* __restrict__ dens, const T* __restrict__ sem, const T* __restrict__ rgb,
                      float* __restrict__ rgb_out, float* __restrict__ seg_out,
                      float* __restrict__ depth_out, int* __restrict__ term_out, float* __restrict__ rows) {
  cam_fwd_direct_tile<T, NCH, ERT, NW>(blockIdx.x, P, mats, us, vs, ds, mids, beta_raw, dens, sem, rgb, rgb_out, seg_out,
                                              depth_out, term_out, rows);
}

#ifdef VAMP_DIRECT_STAMPS
extern "C" int vamp_debug_direct_stamps(long long* host, size_t n) {
  return (int) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_direct_stamps), n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

// Diagnostic export (tests): inside mask, floor taps and continuous tap coordinates of EVERY sample
// exactly as cam_fwd_direct_kernel evaluates them -- same tile decomposition, same wave composition (the
// IEEE-division fallback near a face is a wave-level decision), same code (chain_tap).
__global__ void __launch_bounds__(256)
cam_direct_taps_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                       const float* __restrict__ vs, const float* __restrict__ ds,
                       uint8_t* __restrict__ inside, int16_t* __restrict__ ix0, int16_t* __restrict__ iy0,
                       int16_t* __restrict__ iz0, float* __restrict__ fxyz) {
  const RayId id = decode_tile(P, blockIdx.x);
  const int bn = __builtin_amdgcn_readfirstlane((int) id.bn);
  const int S = P.D - 1;
  const float* m = mats + (long) bn * 48;
  const float u = us[id.w], v = vs[id.h];
  const ChainCtx cc = chain_ctx(P, m, u, v);
  for (int i = id.sub; i < S; i += 4) {
    const VolTap tp = chain_tap(P, m, cc, ds[i]);
    if (!id.live) continue;
    const long idx = (((long) bn * S + i) * P.fH + id.h) * P.fW + id.w;
    inside[idx] = tp.inside ? 1 : 0;
    ix0[idx] = (int16_t) tp.ix0; iy0[idx] = (int16_t) tp.iy0; iz0[idx] = (int16_t) tp.iz0;
    if (fxyz) { fxyz[3 * idx] = tp.fx; fxyz[3 * idx + 1] = tp.fy; fxyz[3 * idx + 2] = tp.fz; }
  }
}

int launch_cam_direct_taps(const RenderParams& P, const float* mats, const float* us, const float* vs,
                           const float* ds, uint8_t* inside, int16_t* ix0, int16_t* iy0, int16_t* iz0,
                           float* fxyz, hipStream_t s) {
  const long tiles = (long) P.B * P.N * ((P.fH + 7) / 8) * ((P.fW + 7) / 8);
  cam_direct_taps_kernel<<<(unsigned) ((tiles + 7) / 8 * 8), 256, 0, s>>>(P, mats, us, vs, ds, inside, ix0, iy0, iz0, fxyz);
  return check_launch("cam_direct_taps_kernel");
}

int launch_cam_fwd_direct(const VampRenderDesc* d, const RenderParams& P, const float* mats, const float* us,
                          const float* vs, const float* ds, const float* mids, const float* beta,
                          const void* dens, const void* sem, const void* rgb, float* rgb_out,
                          float* seg_out, float* depth_out, int* term_out, bool ert, float* rows, hipStream_t s) {
  const int S = P.D - 1, nch = P.K + 3;
  const long tiles = (long) P.B * P.N * ((P.fH + 7) / 8) * ((P.fW + 7) / 8);
  const unsigned grid = (unsigned) ((tiles + 7) / 8 * 8);
#define VAMP_CAMD_L(T, NCH, ERT)                                                                        \
  VAMP_TIMED(kProfCamFwd, s, (cam_fwd_direct_kernel<T, NCH, ERT, VAMP_DIRECT_NW><<<grid, VAMP_DIRECT_NW * 64, dyn, s>>>( \
      P, mats, us, vs, ds, mids, beta, static_cast<const T*>(dens), static_cast<const T*>(sem),         \
      static_cast<const T*>(rgb), rgb_out, seg_out, depth_out, term_out, rows)))
#define VAMP_CAMD(T, NCH)                                                                               \
  do {                                                                                                  \
    const size_t dyn = cam_direct_dyn_bytes(S, NCH);                                                    \
    if (ert) VAMP_CAMD_L(T, NCH, true); else VAMP_CAMD_L(T, NCH, false);                                \
  } while (0)
#define VAMP_CAMD_T(T)                                                                                  \
  do {                                                                                                  \
    if (nch <= 8) VAMP_CAMD(T, 8);                                                                      \
    else if (nch <= 12) VAMP_CAMD(T, 12);                                                               \
    else if (nch == 21) VAMP_CAMD(T, 21);                                                               \
    else if (nch <= 24) VAMP_CAMD(T, 24);                                                               \
    else VAMP_CAMD(T, 32);                                                                              \
  } while (0)
  if (d->in_dtype == VAMP_F32) VAMP_CAMD_T(float);
  else VAMP_CAMD_T(__hip_bfloat16);
#undef VAMP_CAMD_T
#undef VAMP_CAMD
#undef VAMP_CAMD_L
  return check_launch("cam_fwd_direct_kernel");
}

}  // namespace vamp

extern "C" int vamp_render_camera_direct_taps(const VampRenderDesc* d, const float* mats, const float* us,
                                              const float* vs, const float* ds, uint8_t* inside, int16_t* ix0,
                                              int16_t* iy0, int16_t* iz0, float* fxyz, void* stream) {
  if (int e = vamp::validate(d)) return e;
  VAMP_REQUIRE(mats && us && vs && ds && inside && ix0 && iy0 && iz0, "null pointer");
  return vamp::launch_cam_direct_taps(vamp::to_params(d), mats, us, vs, ds, inside, ix0, iy0, iz0, fxyz,
                                      static_cast<hipStream_t>(stream));
}
