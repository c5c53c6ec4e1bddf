// Shared pieces of the camera / BEV render kernels.
#pragma once
#include "common.hpp"

namespace vamp {

struct RenderParams {
  int B, N, D, fH, fW, K, C, Z, Y, X, oZ, oY, oX;
  float lo[3], span[3];
  float d_far, z_step;
  int density_mode;
  float sdf_bias, beta_min;
  int cat_seg;
  int CP;   // packed channels per voxel for the camera branch: 1 + K + 3 rounded up to 12/24/32
};

inline RenderParams to_params(const VampRenderDesc* d) {
  RenderParams p;
  p.B = d->B; p.N = d->N; p.D = d->D; p.fH = d->fH; p.fW = d->fW; p.K = d->K; p.C = d->C;
  p.Z = d->Z; p.Y = d->Y; p.X = d->X; p.oZ = d->oZ; p.oY = d->oY; p.oX = d->oX;
  for (int i = 0; i < 3; ++i) { p.lo[i] = d->lo[i]; p.span[i] = d->span[i]; }
  p.d_far = d->d_far; p.z_step = d->z_step_det; p.density_mode = d->density_mode;
  p.sdf_bias = d->sdf_bias; p.beta_min = d->beta_min; p.cat_seg = d->cat_seg;
  // CP is chosen from {12, 24, 32} so that only three kernel bodies are compiled
  const int need = 1 + d->K + 3;
  p.CP = need <= 12 ? 12 : (need <= 24 ? 24 : 32);
  return p;
}

inline int validate(const VampRenderDesc* d) {
  VAMP_REQUIRE(d != nullptr, "desc is NULL");
  VAMP_REQUIRE(d->B > 0 && d->N > 0, "B, N must be positive");
  VAMP_REQUIRE(d->D > 1 && d->fH > 0 && d->fW > 0, "D > 1, fH, fW > 0");
  VAMP_REQUIRE(d->K > 0 && d->K <= 28, "1 <= K <= 28");
  VAMP_REQUIRE(d->C >= 0 && d->C <= 64, "0 <= C <= 64");
  VAMP_REQUIRE(d->Z > 1 && d->Y > 1 && d->X > 1, "Z, Y, X > 1");
  VAMP_REQUIRE(d->X < 2047 && d->Y < 2047 && d->Z < 511, "volume axes limited to 2046 x 2046 x 510 (packed tap keys: 11 / 11 / 9 bits + sign)");
  VAMP_REQUIRE(d->density_mode == VAMP_DENSITY_SIGMOID ||
               d->density_mode == VAMP_DENSITY_SDF_LAPLACE, "density_mode");
  VAMP_REQUIRE(d->in_dtype == VAMP_F32 || d->in_dtype == VAMP_BF16, "in_dtype");
  return VAMP_OK;
}

// ---------------------------------------------------------------------------
// Ray -> thread mapping of the per-ray kernels.  A 256-thread workgroup owns an
// 8 x (32 / LPR) pixel tile of one camera; within a wave the lanes are laid out
// [sub][ray] so that neighbouring lanes march neighbouring rays at the same depth
// (shared taps -> L1 hits, contiguous per-sample stores).  Workgroups are dealt to
// the 8 XCDs round-robin by the hardware; the tile index is permuted so that each
// XCD works on one contiguous range of tiles (~3/4 of a camera) and its private L2
// keeps that camera's part of the volume.
// ---------------------------------------------------------------------------
struct RayId {
  bool live;     // false: lane only takes part in shuffles
  int w, h, sub;
  long bn;
  int b;
  long tile;     // 8 x 8 ray tile of the ray (decode_ray_wps only): per-sample tables are [tile][depth index][64]
  bool tile_ok;  // the workgroup has a tile of its own (the grid is padded to a multiple of 8)
};

template <int LPR>
__device__ __forceinline__ int ray_tile_h() { return 32 / LPR; }

template <int LPR>
inline unsigned ray_grid(const RenderParams& P) {
  const int th = 32 / LPR;
  const long tiles = (long) P.B * P.N * ((P.fH + th - 1) / th) * ((P.fW + 7) / 8);
  return (unsigned) ((tiles + 7) / 8 * 8);
}

template <int LPR>
__device__ __forceinline__ RayId decode_ray(const RenderParams& P) {
  constexpr int RPW = 64 / LPR;                   // rays per wave
  constexpr int THt = 32 / LPR;                   // tile height (tile width is 8)
  const int tiles_w = (P.fW + 7) / 8, tiles_h = (P.fH + THt - 1) / THt;
  const long tiles = (long) P.B * P.N * tiles_h * tiles_w;
  const long per_xcd = (tiles + 7) / 8;
  const long t = (long) (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = wave * RPW + (lane % RPW);        // ray slot inside the tile
  RayId id;
  id.sub = lane / RPW;
  const long tc = t < tiles ? t : tiles - 1;
  id.bn = tc / (tiles_h * tiles_w);
  const int tt = (int) (tc % (tiles_h * tiles_w));
  id.h = (tt / tiles_w) * THt + r / 8;
  id.w = (tt % tiles_w) * 8 + r % 8;
  id.live = t < tiles && id.h < P.fH && id.w < P.fW;
  if (id.h >= P.fH) id.h = P.fH - 1;
  if (id.w >= P.fW) id.w = P.fW - 1;
  id.b = (int) (id.bn / P.N);
  return id;
}

// Alternative mapping ("wave per depth chunk"): the 64 lanes of a wave are the 64 rays of the
// 8 x 8 tile and the wave index is the depth chunk.  All lanes of a load instruction then sit at
// the same depth on neighbouring rays, so lanes that need the same tap are merged by the texture
// addresser (one fetch), which matters in the near and mid field where many rays cross a voxel.
// The chunks are merged through LDS instead of shuffles.  Needs 4 waves (LPR = 4).
// tile `t` (-1: none) of the wave-per-depth-chunk mapping
__device__ __forceinline__ RayId decode_ray_tile(const RenderParams& P, long t) {
  const int tiles_w = (P.fW + 7) / 8, tiles_h = (P.fH + 7) / 8;
  const long tiles = (long) P.B * P.N * tiles_h * tiles_w;
  const int r = threadIdx.x & 63;
  RayId id;
  id.sub = threadIdx.x >> 6;
  id.tile_ok = t >= 0 && t < tiles;
  const long tc = id.tile_ok ? t : tiles - 1;
  id.tile = tc;
  id.bn = tc / (tiles_h * tiles_w);
  const int tt = (int) (tc % (tiles_h * tiles_w));
  id.h = (tt / tiles_w) * 8 + r / 8;
  id.w = (tt % tiles_w) * 8 + r % 8;
  id.live = id.tile_ok && id.h < P.fH && id.w < P.fW;
  if (id.h >= P.fH) id.h = P.fH - 1;
  if (id.w >= P.fW) id.w = P.fW - 1;
  id.b = (int) (id.bn / P.N);
  return id;
}
__device__ __forceinline__ RayId decode_ray_wps(const RenderParams& P) {
  const int tiles_w = (P.fW + 7) / 8, tiles_h = (P.fH + 7) / 8;
  const long tiles = (long) P.B * P.N * tiles_h * tiles_w;
  const long per_xcd = (tiles + 7) / 8;
  return decode_ray_tile(P, (long) (blockIdx.x % 8) * per_xcd + blockIdx.x / 8);
}

// exclusive prefix sum over the LPR lanes of a ray ([sub][ray] lane layout)
template <int LPR>
__device__ __forceinline__ float ray_excl_prefix(float v, int sub) {
  constexpr int RPW = 64 / LPR;
  float incl = v;
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) {
    const float up = __shfl_up(incl, o * RPW, 64);
    if (sub >= o) incl += up;
  }
  const float ex = __shfl_up(incl, RPW, 64);
  return sub == 0 ? 0.f : ex;
}

// sum over the LPR lanes of a ray, valid in the sub == 0 lane
template <int LPR>
__device__ __forceinline__ float ray_sum(float v) {
  constexpr int RPW = 64 / LPR;
#pragma unroll
  for (int o = LPR >> 1; o > 0; o >>= 1) v += __shfl_down(v, o * RPW, 64);
  return v;
}

// exclusive suffix sum over the LPR lanes of a ray
template <int LPR>
__device__ __forceinline__ float ray_excl_suffix(float v, int sub) {
  constexpr int RPW = 64 / LPR;
  float incl = v;
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) {
    const float dn = __shfl_down(incl, o * RPW, 64);
    if (sub + o < LPR) incl += dn;
  }
  const float ex = __shfl_down(incl, RPW, 64);
  return sub == LPR - 1 ? 0.f : ex;
}

// One frustum point in the ego frame: get_geometry (bv2:328-349) followed by
// nan_to_num(nan=-1e3) (bv2:612).  m = [inv(ida), sensor2ego @ inv(intrin), bda].
__device__ __forceinline__ void frustum_point(const float* __restrict__ m, float u, float v,
                                              float dd, float& x, float& y, float& z) {
  Vec4 p{u, v, dd, 1.0f};
  p = matvec(m, p);
  p.x = p.x * p.z;
  p.y = p.y * p.z;
  p = matvec(m + 16, p);
  p = matvec(m + 32, p);
  x = p.x; y = p.y; z = p.z;
}

__device__ __forceinline__ float nan_to_num_geom(float v) {
  // torch.nan_to_num(geom, -1e3): nan -> -1e3, +-inf -> +-FLT_MAX
  if (v != v) return -1e3f;
  return fminf(fmaxf(v, -3.402823466e+38f), 3.402823466e+38f);
}

// Normalise an ego point by the seg bounds (bv2:397-404), test the inclusive
// inside mask (bv2:405-407) and unnormalise with aten's align_corners=True rule.
struct VolTap {
  bool inside;
  int ix0, iy0, iz0;
  float wx0, wx1, wy0, wy1, wz0, wz1;
  float fx, fy, fz;      // continuous tap coordinates
};

__device__ __forceinline__ VolTap volume_tap(const RenderParams& P, float x, float y, float z) {
  const float gx = ((x - P.lo[0]) / P.span[0]) * 2.0f - 1.0f;
  const float gy = ((y - P.lo[1]) / P.span[1]) * 2.0f - 1.0f;
  const float gz = ((z - P.lo[2]) / P.span[2]) * 2.0f - 1.0f;
  VolTap t;
  t.inside = (gx >= -1.0f) && (gx <= 1.0f) && (gy >= -1.0f) && (gy <= 1.0f) &&
             (gz >= -1.0f) && (gz <= 1.0f);
  const float fx = ((gx + 1.0f) / 2.0f) * (float) (P.X - 1);
  const float fy = ((gy + 1.0f) / 2.0f) * (float) (P.Y - 1);
  const float fz = ((gz + 1.0f) / 2.0f) * (float) (P.Z - 1);
  const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
  t.ix0 = (int) flx; t.iy0 = (int) fly; t.iz0 = (int) flz;
  t.wx1 = fx - flx; t.wx0 = (flx + 1.0f) - fx;
  t.wy1 = fy - fly; t.wy0 = (fly + 1.0f) - fy;
  t.wz1 = fz - flz; t.wz0 = (flz + 1.0f) - fz;
  t.fx = fx; t.fy = fy; t.fz = fz;
  return t;
}

// ---------------------------------------------------------------------------
// Compositing weight w = (1 - e^{-tau}) e^{-cum} (bv2:430-434) for the camera forwards, at v_exp_f32 cost but
// without the two error sources of `(1 - __expf(-tau)) * __expf(-cum)`:
//  * __expf(x) = exp2(fl(x log2 e)): the rounding of the product is an error of |x| 2^-24 relative, 1e-6 at an
//    optical depth of 18 -- the product is carried in two floats instead;
//  * 1 - e^{-tau} cancels for the small tau of every masked sample (density(0) * delta = 2e-4: one ulp of the
//    exponential is 3e-4 of the difference) -- below 1/4 the series of -expm1(-tau) is summed directly.
// (Round 5, when the full-size element check was made absolute.  It turned out NOT to be what that check saw --
// 2.1e-4 m on depth_preds at cfg-A came from the sample coordinates, see VAMP_CAMFWD_EXACT_TAPS -- and costs
// nothing measurable, so it stays as margin.)
__device__ __forceinline__ float exp_acc(float x) {             // e^x to ~1.5 ulp
  const float L = 1.4426950216293335f, Ll = 1.9259629911266175e-8f;      // fl(log2 e), log2 e - fl(log2 e)
  const float hi = x * L;
  const float lo = __builtin_fmaf(x, L, -hi) + x * Ll;
  const float e = __builtin_amdgcn_exp2f(hi);
  // (x = -+inf: the correction is inf - inf; the plain exponential is already exact there -- a density feature of
  // -inf, bv2:421's nan_to_num notwithstanding, saturates the ray with tau = inf)
  return fabsf(hi) <= 3.0e38f ? __builtin_fmaf(e, lo * 0.6931471805599453f, e) : e;
}
__device__ __forceinline__ float alpha_acc(float tau) {         // 1 - e^{-tau}, tau >= 0
  // tau (1 - tau/2 (1 - tau/3 (... (1 - tau/8)))): remainder tau^8 / 9! of the leading term, 4e-11 at 1/4
  float p = __builtin_fmaf(tau, -0.125f, 1.0f);
  p = __builtin_fmaf(tau * p, -1.0f / 7.0f, 1.0f);
  p = __builtin_fmaf(tau * p, -1.0f / 6.0f, 1.0f);
  p = __builtin_fmaf(tau * p, -0.2f, 1.0f);
  p = __builtin_fmaf(tau * p, -0.25f, 1.0f);
  p = __builtin_fmaf(tau * p, -1.0f / 3.0f, 1.0f);
  p = __builtin_fmaf(tau * p, -0.5f, 1.0f);
  return tau < 0.25f ? tau * p : 1.0f - exp_acc(-tau);          // (NaN takes the second form and stays NaN)
}
__device__ __forceinline__ float composite_weight(float tau, float cum) { return alpha_acc(tau) * exp_acc(-cum); }

// Early ray termination.  term[ray] = number of leading samples of the ray that are kept: the
// first index whose transmittance in front of it, exp(-sum_{j<i} sigma_j delta_j), is below
// exp(-kTermOpticalDepth).  Every dropped sample has compositing weight w_i <= T_i < 1.6e-8, and
// all of them together sum to < 1.6e-8 (the weights telescope: sum_{i>=t} w_i = T_t - T_end), so
// the rendered maps move by < 1.6e-8 * max|s| and every gradient term the dropped samples carry
// has the same factor -- three orders below the 1e-4 the outputs are held to.  With the sdf
// density of the reference's default configuration rays saturate a few samples after they enter
// occupied space; on the synthetic workload 83 % of the inside samples lie behind that point.
// ---------------------------------------------------------------------------
constexpr float kTermOpticalDepth = 18.0f;

// bytes of the per-ray table (int per ray), and where it lives in the render workspace: behind
// the base region [packed | max(gradient copy, backward scratch)]
size_t packed_bytes(const VampRenderDesc* d);
size_t cam_bwd_v2_bytes(const VampRenderDesc* d);
inline size_t cam_term_bytes(const VampRenderDesc* d) {
  return align_up((size_t) d->B * d->N * d->fH * d->fW * sizeof(int), 256);
}
inline size_t render_base_bytes(const VampRenderDesc* d) {
  const size_t pb = packed_bytes(d), v2 = cam_bwd_v2_bytes(d);
  return pb + (pb > v2 ? pb : v2);
}
inline int* cam_term_ptr(const VampRenderDesc* d, void* workspace) {
  return reinterpret_cast<int*>(static_cast<char*>(workspace) + render_base_bytes(d));
}
int launch_cam_term(const VampRenderDesc* d, const RenderParams& P, const float* mats, const float* us,
                    const float* vs, const float* ds, const float* beta, const void* density_feature,
                    int* term, hipStream_t s);

// What a forward that draws the camera backward's cell ranks itself needs of the cell lists (round 6: the rank pass
// of the prepare step inside the one-kernel camera forward's gather phase, render_cam_direct_dev.hpp): cell counters
// (zero on entry), the per-sample rank table, the tiles' depths.
struct CamRankRefs {
  int* cnt;             // [cells] nullptr: the forward draws no ranks
  int* rank;            // [tiles][S][64]
  int* tile_se;         // [tiles]
  long ncell_b;         // cells per sample of the batch
  float* zero_word;     // not nullptr: the launch also stores 0.f there (the backward's d loss / d beta accumulator)
};
// `workspace`: the render workspace (vamp_render_workspace_bytes).  render_bwd_ray.hip
CamRankRefs cam_rank_refs(const VampRenderDesc* d, void* workspace);
// zero the cell counters (a forward that draws ranks on a workspace not known clean) / scan them and build the heavy
// list behind such a forward (what vamp_render_camera_prepare does behind its own rank pass)
int launch_cam_counters_zero(const VampRenderDesc* d, void* workspace, hipStream_t s);
// (`also`: another cell list scanned by the same launch -- the lift's, vamp_render_camera_prepare_with_lift)
int launch_cam_prepare_ranked(const VampRenderDesc* d, void* workspace, hipStream_t s, const ScanJob* also = nullptr);

// render_cam_direct.hip: plan + density march + scan + channel gather in one kernel, on the
// channel-first volumes; term_out (may be NULL) receives the per-ray table
int launch_cam_fwd_direct(const VampRenderDesc* d, const RenderParams& P, const float* mats, const float* us,
                          const float* vs, const float* ds, const float* mids, const float* beta,
                          const void* dens, const void* sem, const void* rgb, float* rgb_out,
                          float* seg_out, float* depth_out, int* term_out, bool ert, float* rows, hipStream_t s);

// render_bev_fused.hip: the BEV forward (density, weights, all channels) as one kernel
#ifndef VAMP_BEVF_NWV
#define VAMP_BEVF_NWV 4              // waves per column block
#endif
// Planes a lattice of oZ heights with spacing det_step[2] can touch (+ slack): what a wave's slab is sized for.
inline int bev_planes_alloc(const VampRenderDesc* d) {
  const float per = fabsf(d->det_step[2]) * (float) (d->Z - 1) / d->span[2];       // volume planes per height step
  return (int) ceilf((float) (d->oZ - 1) * per) + 4;
}
// dynamic LDS of a column block: sigma [oZ][64] | weights [oZ][64] | density planes [NPA + 1][64] | per wave [NPA][64]
inline size_t bev_fused_dyn_bytes(int oZ, int npa) {
  return sizeof(float) * 64 * (2 * (size_t) oZ + (size_t) (VAMP_BEVF_NWV + 1) * npa + 1);
}
// Do the heights `ozs_host` (the caller's host copy of the device array) fit the slabs?  The kernel clamps a plane index
// that falls outside [pmin, pmin + NPA) -- silently wrong values -- so the one-kernel forward runs only for arrays that
// pass here: the z taps are evaluated with the kernel's own fp32 formula (bev_axis) and must span at most NPA - 1
// planes (one plane of margin for a floor that rounds differently on the host).  The reference's lattice
// (create_voxel_coords, bv2:273-293) passes by construction of bev_planes_alloc; any other array takes the two-kernel
// path, which handles every height on its own.
inline bool bev_fused_heights_fit(const VampRenderDesc* d, const float* ozs_host) {
  if (!ozs_host || d->oZ < 1) return false;
  int pmin = 0, pmax = 0;
  for (int j = 0; j < d->oZ; ++j) {
    const float pos = ozs_host[j];
    if (!(fabsf(pos) <= 3.0e38f)) return false;
    const float g = ((pos - d->lo[2]) / d->span[2]) * 2.0f - 1.0f;
    const float f = ((g + 1.0f) / 2.0f) * (float) (d->Z - 1);
    if (!(fabsf(f) < 1.0e9f)) return false;
    const int i0 = (int) floorf(f);
    pmin = j ? (i0 < pmin ? i0 : pmin) : i0;
    pmax = j ? (i0 + 1 > pmax ? i0 + 1 : pmax) : i0 + 1;
  }
  return pmax - pmin + 2 <= bev_planes_alloc(d);
}

bool bev_fwd_fused_supported(const VampRenderDesc* d);
int launch_bev_fwd_fused(const VampRenderDesc* d, const RenderParams& P, const float* oxs, const float* oys,
                         const float* ozs, const float* bev_mids, const float* beta, const void* dens,
                         const void* sem, const void* rgb, const void* base, float* bev_rgb, float* bev_seg,
                         float* bev_height, float* voxel_density, float* voxel_output, float* s0_save,
                         float* ss_save, hipStream_t s);

// render_fwd_merged.hip: camera tiles + BEV column blocks in one launch (early ray termination on)
bool render_fwd_merged_supported(const VampRenderDesc* d);
int launch_render_fwd_merged(const VampRenderDesc* d, const RenderParams& P, const float* mats, const float* us,
                             const float* vs, const float* ds, const float* mids, const float* oxs, const float* oys,
                             const float* ozs, const float* bev_mids, const float* beta, const void* dens,
                             const void* sem, const void* rgb, const void* base, float* rgb_out, float* seg_out,
                             float* depth_out, int* term_out, float* rows, float* bev_rgb, float* bev_seg,
                             float* bev_height, float* voxel_density, float* voxel_output, float* s0_save,
                             float* ss_save, const CamRankRefs& rank, hipStream_t s);

// 8-tap trilinear gather of CP4*4 packed channels for an INSIDE sample, branch-free: all
// 8 * CP4 16-byte loads are independent and can be in flight together (a per-tap bounds
// branch would serialise eight memory round trips).  Inside => tap indices >= 0; a "+1" tap
// can only fall outside when its weight is exactly zero, so it is clamped and zero-weighted.
template <int CP4>
__device__ __forceinline__ void gather_taps(const RenderParams& P, const float* __restrict__ vol,
                                            const VolTap& tp, float* __restrict__ s) {
  constexpr int CP = CP4 * 4;
  const int x1 = min(tp.ix0 + 1, P.X - 1), y1 = min(tp.iy0 + 1, P.Y - 1), z1 = min(tp.iz0 + 1, P.Z - 1);
  const float wx1 = (tp.ix0 + 1 < P.X) ? tp.wx1 : 0.f;
  const float wy1 = (tp.iy0 + 1 < P.Y) ? tp.wy1 : 0.f;
  const float wz1 = (tp.iz0 + 1 < P.Z) ? tp.wz1 : 0.f;
  const long rx0 = (long) tp.ix0 * CP, rx1 = (long) x1 * CP;
  const long ry0 = (long) tp.iy0 * P.X * CP, ry1 = (long) y1 * P.X * CP;
  const long rz0 = (long) tp.iz0 * P.Y * P.X * CP, rz1 = (long) z1 * P.Y * P.X * CP;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float wt = ((k & 1) ? wx1 : tp.wx0) * ((k & 2) ? wy1 : tp.wy0) * ((k & 4) ? wz1 : tp.wz0);
    const float4* f4 = reinterpret_cast<const float4*>(
        vol + ((k & 4) ? rz1 : rz0) + ((k & 2) ? ry1 : ry0) + ((k & 1) ? rx1 : rx0));
#pragma unroll
    for (int q = 0; q < CP4; ++q) {
      const float4 f = f4[q];
      s[q * 4 + 0] = __builtin_fmaf(wt, f.x, s[q * 4 + 0]);
      s[q * 4 + 1] = __builtin_fmaf(wt, f.y, s[q * 4 + 1]);
      s[q * 4 + 2] = __builtin_fmaf(wt, f.z, s[q * 4 + 2]);
      s[q * 4 + 3] = __builtin_fmaf(wt, f.w, s[q * 4 + 3]);
    }
  }
}

// what the per-ray pass of the camera backward needs of the cell lists (render_bwd_cell.hip)
struct CamCellRefs {
  const int* tile_order; // [tiles] the order in which the per-ray pass takes the tiles (longest first), or nullptr
  const int* rank;      // [tiles][S][64] rank of a kept inside sample in its cell
  int* slot;            // [tiles][S][64] scratch of the per-ray pass
  const int* off;       // cell start = off[c] + boff[c / kScanTile]
  const int* boff;
  float4* R;            // [samples][2] records in cell order
  long ncell_b;         // cells per sample of the batch
};

// parts of the camera backward a caller may issue separately (VAMP_CAMBWD_PART_*): the per-ray pass
// (with the cell lists if they are not prepared), the heavy cells' per-corner sums, the per-voxel gather
constexpr int kCamPartRay = 1, kCamPartGather = 2, kCamPartHeavy = 4, kCamPartAll = 7;

}  // namespace vamp
