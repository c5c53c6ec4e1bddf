// Shared by the lift kernels: parameters and the bit-exact voxel -> pixel projection.
#pragma once
#include "common.hpp"

namespace vamp {

struct LiftParams {
  int B, N, C, D, fH, fW, Z, Y, X;
  float u_max, v_max, u_div, v_div, d_lo, d_hi, d_span;
  int use_depth;
};

inline LiftParams to_params(const VampLiftDesc* d) {
  LiftParams p;
  p.B = d->B; p.N = d->N; p.C = d->C; p.D = d->D; p.fH = d->fH; p.fW = d->fW;
  p.Z = d->Z; p.Y = d->Y; p.X = d->X;
  p.u_max = d->u_max; p.v_max = d->v_max; p.u_div = d->u_div; p.v_div = d->v_div;
  p.d_lo = d->d_lo; p.d_hi = d->d_hi; p.d_span = d->d_span; p.use_depth = d->use_depth;
  return p;
}

// Result of projecting one voxel centre into one camera.
struct LiftTap {
  bool valid;
  int ix0, iy0, iz0;
  float wx0, wx1, wy0, wy1, wz0, wz1;
  float zz;              // projected depth (camera z), used to bin voxels into depth slabs
  float fx, fy, fz;      // continuous tap coordinates
};

// get_pixel (bv2:365-388) + validity / normalisation (bv2:493-505) + aten's
// grid_sampler_unnormalize for align_corners=False.  Evaluation order is part of
// the contract (bit-exact tap indices): do not reassociate, do not fuse.
// WAVE_CULL (callers in wave-uniform control flow only): when the third row of `ida` is exactly
// (0, 0, 1, 0) -- image-plane augmentations never touch depth -- the projected depth zz equals the
// camera-space z bit for bit (or is NaN), so a camera that no lane of the wave has in front of it
// (z > d_lo, resp. z > 0) is invalid for all 64 voxels and the divisions, the third matrix and
// the normalisation are skipped: the same masks as the full chain, about half of the projections
// (the cameras facing away) at a third of the instructions.
template <bool WAVE_CULL = false>
__device__ __forceinline__ LiftTap lift_project(const LiftParams& P, const float* __restrict__ m,
                                                float x, float y, float z) {
  Vec4 p{x, y, z, 1.0f};
  p = matvec(m, p);        // inv(bda)
  p = matvec(m + 16, p);   // intrin @ inv(sensor2ego)
  if (WAVE_CULL) {
    const bool e3 = m[40] == 0.0f && m[41] == 0.0f && m[42] == 1.0f && m[43] == 0.0f;   // uniform
    if (e3 && !__any(p.z > (P.use_depth ? P.d_lo : 0.0f))) {
      LiftTap t;
      t.valid = false;
      t.ix0 = t.iy0 = t.iz0 = 0;
      t.wx0 = t.wx1 = t.wy0 = t.wy1 = t.wz0 = t.wz1 = 0.f;
      t.zz = p.z;
      t.fx = t.fy = t.fz = 0.f;
      return t;
    }
  }
  float zc = (p.z < 1e-6f) ? 1e-6f : p.z;   // clamp(min=eps); NaN stays NaN
  p.x = p.x / zc;
  p.y = p.y / zc;
  p = matvec(m + 32, p);   // ida
  const float u = p.x, v = p.y, zz = p.z;
  LiftTap t;
  bool ok = (u > -0.5f) && (u < P.u_max) && (v > -0.5f) && (v < P.v_max);
  if (P.use_depth) ok = ok && (zz > P.d_lo) && (zz < P.d_hi);
  else ok = ok && (zz > 0.0f);
  t.valid = ok;
  t.zz = zz;
  float nx = 2.0f * (u / P.u_div) - 1.0f;
  float ny = 2.0f * (v / P.v_div) - 1.0f;
  float nz = P.use_depth ? (2.0f * ((zz - P.d_lo) / P.d_span) - 1.0f) : 0.0f;
  nx = fminf(fmaxf(nx, -2.0f), 2.0f);
  ny = fminf(fmaxf(ny, -2.0f), 2.0f);
  nz = fminf(fmaxf(nz, -2.0f), 2.0f);
  const float fx = ((nx + 1.0f) * (float) P.fW - 1.0f) / 2.0f;
  const float fy = ((ny + 1.0f) * (float) P.fH - 1.0f) / 2.0f;
  const float fz = ((nz + 1.0f) * (float) P.D - 1.0f) / 2.0f;
  const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
  t.ix0 = (int) flx; t.iy0 = (int) fly; t.iz0 = (int) flz;
  t.wx1 = fx - flx; t.wx0 = (flx + 1.0f) - fx;
  t.wy1 = fy - fly; t.wy0 = (fly + 1.0f) - fy;
  t.wz1 = fz - flz; t.wz0 = (flz + 1.0f) - fz;
  t.fx = fx; t.fy = fy; t.fz = fz;
  return t;
}


// lift_bwd_cell.hip
size_t lift_bwd_cell_ws_bytes(const VampLiftDesc* d);
int launch_lift_bwd_cell(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, const float* gout,
                         const uint64_t* hits, float* gdepth, float* gfeat, void* scratch,
                         bool cells_valid, int wpp_force, int half, bool softmax_bwd, hipStream_t s);
int launch_lift_cell_prepare(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, void* scratch, hipStream_t s);

}  // namespace vamp
