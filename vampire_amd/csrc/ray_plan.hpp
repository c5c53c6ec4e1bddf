// Per-tile schedule of the camera ray march (forward; the backward's per-ray pass can use it too).
//
// The 64 lanes of a wave are the 64 rays of an 8 x 8 pixel tile at ONE depth index.  At a fixed
// depth index the frustum points are an affine image of the pixel rectangle (get_geometry,
// bv2:328-349: inv(ida) is affine in (u, v), the depth is a constant of the plane), so the tile's
// sample points form a planar parallelogram whose bounding box in voxel coordinates is spanned by
// the tile's four corner rays.  That gives, for four chain evaluations per depth index instead of
// 64: (1) the depth indices at which NO ray of the tile is inside the volume (64 % of all
// tile-steps at cfg-B) -- they are composited in ~20 instructions without evaluating the frustum
// chain -- and (2) the voxel box of the others.
//
// (The box was built for LDS-staged voxel bricks -- copy the box into LDS once per wave and gather
// the taps from there at ds_read_b128 rates.  Built, correct, and measured slower than the direct
// gather on MI355X: DESIGN.md section 4 "LDS bricks", tools/microbench/brick_dma.hip and
// brick_replay.hip hold the experiment.)
#pragma once
#include "render_common.hpp"

namespace vamp {

// inclusive voxel-index box [lo, hi] that contains the floor and floor + 1 taps of every INSIDE
// lane of the wave; wave-uniform
struct BrickBox {
  int lo[3], hi[3];
};

// One axis of the box from the extremes of the continuous tap coordinate over the tile's corner
// rays: 0.01 voxel of slack for the rounding of the interior lanes' chains; fmaxf(nan, 0) = 0.
__device__ __forceinline__ void brick_axis(float mn, float mx, int dim, int& lo, int& hi) {
  const float top = (float) (dim - 2);
  lo = (int) fminf(fmaxf(floorf(mn - 0.01f), 0.f), top);
  hi = (int) fminf(fmaxf(floorf(mx + 0.01f), 0.f), top) + 1;
}

// ---------------------------------------------------------------------------
// Per-tile schedule, shared by the four waves of a workgroup through LDS: which depth indices of
// the tile's rays can hold an inside sample at all, and the box of each.  One wave-wide evaluation
// of the frustum chain covers the four corner rays of 16 consecutive depth indices (lane = 4 * step
// + corner); wave k plans indices [32 k, 32 k + 32).  Depth indices whose corner box misses the
// volume are composited without evaluating the chain at all (the 64 samples are affine
// combinations of the four corners, so they are all outside), and the active indices are dealt
// evenly to the four waves (a near tile has all of its first 22 indices active and none beyond
// the 50th: fixed chunks leave three waves idle).
// ---------------------------------------------------------------------------
constexpr int kPlanMax = 128;                          // depth indices per ray the plan can hold

// plan[i] = {lo | hi << 16 of the box per axis, active flag}
__device__ __forceinline__ void plan_tile(const RenderParams& P, const float* __restrict__ m,
                                          const float* __restrict__ us, const float* __restrict__ vs,
                                          const float* __restrict__ ds, int w_lo, int w_hi, int h_lo,
                                          int h_hi, int wave, int4* __restrict__ plan) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 3;
  const float u = us[(c & 1) ? w_hi : w_lo], v = vs[(c & 2) ? h_hi : h_lo];
  const int S = P.D - 1;
#pragma unroll 1
  for (int e = 0; e < 2; ++e) {
    const int i = 32 * wave + 16 * e + (lane >> 2);
    float x, y, z;
    frustum_point(m, u, v, ds[min(i, P.D - 1)], x, y, z);
    // a non-finite corner voids the affine argument: keep the depth index
    int fin = (fabsf(x) <= 1e30f && fabsf(y) <= 1e30f && fabsf(z) <= 1e30f) ? 1 : 0;
    const VolTap tp = volume_tap(P, nan_to_num_geom(x), nan_to_num_geom(y), nan_to_num_geom(z));
    float mn[3] = {tp.fx, tp.fy, tp.fz}, mx[3] = {tp.fx, tp.fy, tp.fz};
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
        mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
      }
      fin &= __shfl_xor(fin, o, 64);
    }
    const float slack = 0.05f;                           // voxels; the interior lanes deviate < 1e-3
    const bool hit = mx[0] >= -slack && mn[0] <= (float) (P.X - 1) + slack &&
                     mx[1] >= -slack && mn[1] <= (float) (P.Y - 1) + slack &&
                     mx[2] >= -slack && mn[2] <= (float) (P.Z - 1) + slack;
    int4 ent;
    int lo, hi;
    brick_axis(mn[0], mx[0], P.X, lo, hi); ent.x = lo | (hi << 16);
    brick_axis(mn[1], mx[1], P.Y, lo, hi); ent.y = lo | (hi << 16);
    brick_axis(mn[2], mx[2], P.Z, lo, hi); ent.z = lo | (hi << 16);
    ent.w = (i < S && (hit || !fin)) ? 1 : 0;
    if (c == 0) plan[i] = ent;
  }
}

// box of depth index j; j is wave-uniform (every lane reads the same LDS word: a broadcast)
__device__ __forceinline__ BrickBox plan_box(const int4* __restrict__ plan, int j) {
  const int4 e = plan[j];
  const int px = __builtin_amdgcn_readfirstlane(e.x), py = __builtin_amdgcn_readfirstlane(e.y);
  const int pz = __builtin_amdgcn_readfirstlane(e.z);
  BrickBox b;
  b.lo[0] = px & 0xffff; b.hi[0] = px >> 16;
  b.lo[1] = py & 0xffff; b.hi[1] = py >> 16;
  b.lo[2] = pz & 0xffff; b.hi[2] = pz >> 16;
  return b;
}

// the active depth indices as two 64-bit masks (wave-uniform)
struct PlanMask {
  unsigned long long lo, hi;
};

__device__ __forceinline__ PlanMask plan_mask(const int4* __restrict__ plan) {
  const int lane = threadIdx.x & 63;
  PlanMask mk;
  mk.lo = __ballot(plan[lane].w != 0);
  mk.hi = __ballot(plan[lane + 64].w != 0);
  return mk;
}

// drop the depth indices >= n (early ray termination: the tile is done there)
__device__ __forceinline__ void mask_truncate(PlanMask& mk, int n) {
  if (n < 64) { mk.lo &= (n <= 0) ? 0ull : (~0ull >> (64 - n)); mk.hi = 0ull; }
  else if (n < 128) mk.hi &= (n == 64) ? 0ull : (~0ull >> (128 - n));
}

__device__ __forceinline__ bool mask_test(const PlanMask& mk, int j) {
  return (((j < 64) ? (mk.lo >> j) : (mk.hi >> (j - 64))) & 1ull) != 0;
}

// first active depth index >= j (kPlanMax when none); j wave-uniform
__device__ __forceinline__ int mask_next(const PlanMask& mk, int j) {
  if (j < 64) {
    const unsigned long long t = mk.lo >> j;
    if (t) return j + __builtin_ctzll(t);
    j = 64;
  }
  if (j < 128) {
    const unsigned long long t = mk.hi >> (j - 64);
    if (t) return j + __builtin_ctzll(t);
  }
  return kPlanMax;
}

// depth index of the active index of rank r (0-based; r < number of active indices)
__device__ __forceinline__ int mask_select(const PlanMask& mk, int r) {
  unsigned long long t = mk.lo;
  int base = 0;
  const int nlo = __builtin_popcountll(mk.lo);
  if (r >= nlo) { t = mk.hi; base = 64; r -= nlo; }
  for (int k = 0; k < r; ++k) t &= t - 1;
  return base + __builtin_ctzll(t);
}

// wave `sub`'s contiguous range [j0, j1) of the S depth indices: equal shares of the active ones
__device__ __forceinline__ void plan_share(const PlanMask& mk, int S, int sub, int& j0, int& j1) {
  const int A = __builtin_popcountll(mk.lo) + __builtin_popcountll(mk.hi);
  if (A < 4) {                                           // nothing to balance
    const int L = (S + 3) >> 2;
    j0 = min(S, sub * L); j1 = min(S, j0 + L);
    return;
  }
  j0 = sub == 0 ? 0 : mask_select(mk, (sub * A) >> 2);
  j1 = sub == 3 ? S : mask_select(mk, ((sub + 1) * A) >> 2);
}

}  // namespace vamp
