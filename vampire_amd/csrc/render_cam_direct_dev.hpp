// Device side of the one-kernel camera forward (render_cam_direct.hip holds the description and the launcher;
// render_fwd_merged.hip runs the same tile function beside the BEV forward's column blocks in one launch).
#pragma once
#include "render_common.hpp"
#include "ray_plan.hpp"
#include "pair_gather.hpp"
#include "cell_list.hpp"

namespace vamp {

// depth index of the first active index at or after `from`, skipping `skip` active ones (uniform)
__device__ __forceinline__ int mask_skip(const PlanMask& mk, int from, int skip) {
  int j = mask_next(mk, from);
  for (int k = 0; k < skip && j < kPlanMax; ++k) j = mask_next(mk, j + 1);
  return j;
}

// wave `sub` of NW: contiguous range [j0, j1) of the S depth indices, equal shares of the active ones
template <int NW>
__device__ __forceinline__ void plan_share_n(const PlanMask& mk, int S, int sub, int& j0, int& j1) {
  const int A = __builtin_popcountll(mk.lo) + __builtin_popcountll(mk.hi);
  if (A < NW) {
    const int L = (S + NW - 1) / NW;
    j0 = min(S, sub * L); j1 = min(S, j0 + L);
    return;
  }
  j0 = sub == 0 ? 0 : mask_select(mk, (sub * A) / NW);
  j1 = sub == NW - 1 ? S : mask_select(mk, ((sub + 1) * A) / NW);
}

// Per-ray constants of get_geometry's first product (bv2:334-336): inv(ida) @ (u, v, d, 1) evaluated as
// ((m0 u + m1 v) + m2 d) + m3 -- the first sum does not depend on the depth index, and m3 * 1 is m3.
struct RayBase {
  float c[4];
};
__device__ __forceinline__ RayBase ray_base(const float* __restrict__ m, float u, float v) {
  RayBase r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r.c[k] = m[4 * k] * u + m[4 * k + 1] * v;
  return r;
}
// frustum_point (render_common.hpp) with that sum taken from `rb`: the same roundings, bit for bit
__device__ __forceinline__ void frustum_point_rb(const float* __restrict__ m, const RayBase& rb, float dd,
                                                 float& x, float& y, float& z) {
  Vec4 p;
  p.x = (rb.c[0] + m[2] * dd) + m[3] * 1.0f;
  p.y = (rb.c[1] + m[6] * dd) + m[7] * 1.0f;
  p.z = (rb.c[2] + m[10] * dd) + m[11] * 1.0f;
  p.w = (rb.c[3] + m[14] * dd) + m[15] * 1.0f;
  p.x = p.x * p.z;
  p.y = p.y * p.z;
  p = matvec(m + 16, p);
  p = matvec(m + 32, p);
  x = nan_to_num_geom(p.x); y = nan_to_num_geom(p.y); z = nan_to_num_geom(p.z);
}
// the same with `bda` known to be the identity (uniform; the reference's default: base_exp.py:113-120 has every
// bda augmentation off): ((1 x + 0 y) + 0 z) + 0 w is x for finite y, z, w and NaN otherwise -- x + 0 (y + z + w)
__device__ __forceinline__ void frustum_point_rb_id(const float* __restrict__ m, const RayBase& rb, float dd,
                                                    float& x, float& y, float& z) {
  Vec4 p;
  p.x = (rb.c[0] + m[2] * dd) + m[3] * 1.0f;
  p.y = (rb.c[1] + m[6] * dd) + m[7] * 1.0f;
  p.z = (rb.c[2] + m[10] * dd) + m[11] * 1.0f;
  p.w = (rb.c[3] + m[14] * dd) + m[15] * 1.0f;
  p.x = p.x * p.z;
  p.y = p.y * p.z;
  p = matvec(m + 16, p);
  const float poison = (((p.x + p.y) + p.z) + p.w) * 0.0f;        // +-0, or NaN when a component is not finite
  x = nan_to_num_geom(p.x + poison); y = nan_to_num_geom(p.y + poison); z = nan_to_num_geom(p.z + poison);
}

// volume_tap (render_common.hpp) with each IEEE division by the constant span[k] replaced by the quotient from
// a refined reciprocal + one residual step: q = x r, q += (x - q span) r.  That is the correctly rounded
// quotient in all but a vanishing fraction of cases (where it is one ulp off), at 3 instructions instead of
// ~10.  Used for the DENSITY samples of the one-kernel forward, whose coordinates then carry the reference's own
// roundings (see chain taps below); never for the inside mask on its own (near a face the wave takes the
// reference's chain with its IEEE divisions).
struct SpanRcp { float rx, ry, rz; };
__device__ __forceinline__ float rcp_refined(float d) {
  const float y = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, y, 1.0f), y, y);
}
__device__ __forceinline__ float div_by(float x, float d, float r) {
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-q, d, x), r, q);
}
__device__ __forceinline__ VolTap volume_tap_rcp(const RenderParams& P, const SpanRcp& R, float x, float y, float z,
                                                 bool& near_face) {
  const float gx = div_by(x - P.lo[0], P.span[0], R.rx) * 2.0f - 1.0f;
  const float gy = div_by(y - P.lo[1], P.span[1], R.ry) * 2.0f - 1.0f;
  const float gz = div_by(z - P.lo[2], P.span[2], R.rz) * 2.0f - 1.0f;
  VolTap t;
  t.inside = (gx >= -1.0f) && (gx <= 1.0f) && (gy >= -1.0f) && (gy <= 1.0f) && (gz >= -1.0f) && (gz <= 1.0f);
  const float X1 = (float) (P.X - 1), Y1 = (float) (P.Y - 1), Z1 = (float) (P.Z - 1);
  const float fx = ((gx + 1.0f) * 0.5f) * X1, fy = ((gy + 1.0f) * 0.5f) * Y1, fz = ((gz + 1.0f) * 0.5f) * Z1;
  const float e = fminf(fminf(fminf(fabsf(fx), fabsf(fx - X1)), fminf(fabsf(fy), fabsf(fy - Y1))),
                        fminf(fabsf(fz), fabsf(fz - Z1)));
  near_face = !(e > 1e-3f);                         // (also true for a NaN coordinate)
  const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
  t.ix0 = (int) flx; t.iy0 = (int) fly; t.iz0 = (int) flz;
  t.wx1 = fx - flx; t.wx0 = (flx + 1.0f) - fx;
  t.wy1 = fy - fly; t.wy0 = (fly + 1.0f) - fy;
  t.wz1 = fz - flz; t.wz0 = (flz + 1.0f) - fz;
  t.fx = fx; t.fy = fy; t.fz = fz;
  return t;
}

// sigma(s) with one v_exp_f32: for t = s - bias > 0 the reference's 0.5 + 0.5 sign(t) expm1(-|t| / beta) is
// 0.5 e, for t < 0 it is 1 - 0.5 e (e = exp(-|t| / beta)); neither form cancels, so the fast exponential's
// 1e-6 relative error is all there is (the outputs are held to 1e-4)
__device__ __forceinline__ float density_fast(const DensityParams& dp, float s) {
#ifdef VAMP_DENSITY_FAST_EXPF
  if (dp.mode == VAMP_DENSITY_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-s));
  const float t = s - dp.bias;
  const float e = 0.5f * __expf(-fabsf(t) * dp.ib);
#else
  if (dp.mode == VAMP_DENSITY_SIGMOID) return 1.0f / (1.f + exp_acc(-s));
  const float t = s - dp.bias;
  const float e = 0.5f * exp_acc(-fabsf(t) * dp.ib);
#endif
  return dp.ib * (t > 0.f ? e : (t < 0.f ? 1.0f - e : 0.5f));
}

#ifdef VAMP_DIRECT_STAMPS
// diagnostic build only (tools/debug/cam_stamps.py): per-tile phase stamps of wave 0
static __device__ long long g_direct_stamps[4096 * 8];
#define VAMP_STAMP(k)                                                                 \
  do {                                                                                \
    if (threadIdx.x == 0 && bid < 4096) g_direct_stamps[bid * 8 + (k)] = (long long) __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define VAMP_STAMP(k) do { } while (0)
#endif

#ifndef VAMP_DIRECT_LINE_PLAN
#define VAMP_DIRECT_LINE_PLAN 1
#endif
// depth indices a wave takes per round of the density phase.  With early termination ONE (round 5; was 4): the
// tile stops at the first round after which all 64 rays are saturated, the typical tile keeps 8 leading samples,
// and rounds of 16 indices marched 16 or 32 of them; with rounds of 4 the phase does a quarter of the work on
// most tiles and the kernel drops from 166 to 126 registers (4 waves per SIMD) -- forward pair 118.5 -> 112 us.
// Without early termination every active index is marched anyway and a barrier per 4 indices costs more than
// the registers gain (one-kernel forward forced, termination off: G = 1 / 2 / 4 -> 254 / 238 / 248 us).
#ifndef VAMP_DIRECT_G
#define VAMP_DIRECT_G 1
#endif
#ifndef VAMP_DIRECT_G_NOERT
#define VAMP_DIRECT_G_NOERT 2
#endif
#ifndef VAMP_DIRECT_CB
#define VAMP_DIRECT_CB 4
#endif
constexpr int kDirectCB = VAMP_DIRECT_CB; // channels whose loads are in flight together in the gather
#ifndef VAMP_DIRECT_NW
#define VAMP_DIRECT_NW 4
#endif
// channel count the kernel is instantiated for (K + 3 rounded up; 21 is the reference's K = 18 at compile time)
inline int cam_direct_nch(int nch) { return nch <= 8 ? 8 : (nch <= 12 ? 12 : (nch == 21 ? 21 : (nch <= 24 ? 24 : 32))); }
// dynamic LDS of a tile: tau / weights [S][64], then the merge buffer [NW][NCH][64]
inline size_t cam_direct_dyn_bytes(int S, int NCH) {
  return sizeof(float) * 64 * (size_t) (S > VAMP_DIRECT_NW * NCH ? S : VAMP_DIRECT_NW * NCH);
}

// ---------------------------------------------------------------------------
// The sample points of a ray: EVERY tap -- density samples and composited channels alike -- is evaluated with the
// reference's own fp32 coordinate chain, get_geometry bv2:328-349 + the normalisation of bv2:397-404, its roundings
// bit for bit (frustum_point_rb; the three divisions by the spans through refined reciprocals, and with the IEEE
// divisions of volume_tap whenever a lane of the wave is within 1e-3 voxel of a face of the volume, so that the
// inclusive inside mask, bv2:405-407, is the reference's).
//
// Rounds 4 - 5 set every ray up as a LINE in tap coordinates in fp64 (exact map, three fp64 fma per sample) and took
// the composited channels' taps from it.  The line is the correctly rounded value of the exact map -- and that is the
// problem: the reference's fp32 chain deviates from the exact map by a few ulp of a tap coordinate, 1e-5 voxel at
// x = 255 and 1e-4 at x = 400, which on white-noise volumes moves a semantic logit by 7.7e-5 at cfg-A and by 1.06e-4 at
// cfg-D -- over north_star's 1e-4 (round 6's element check at cfg-D found it; the planned march, which has always
// used the chain, is within 9e-7 there).  With the chain everywhere the one-kernel forward is as close; it costs ~85
// vector instructions more per gathered sample and saves the ~500 issue slots of the fp64 set-up per wave.
// ---------------------------------------------------------------------------
struct ChainCtx {
  RayBase rb;            // inv(ida) (u, v, ., 1) without the depth term
  SpanRcp span_r;        // refined reciprocals of the volume's spans
  bool bda_identity;     // uniform
};
__device__ __forceinline__ ChainCtx chain_ctx(const RenderParams& P, const float* __restrict__ m, float u, float v) {
  ChainCtx c;
  c.rb = ray_base(m, u, v);
  c.bda_identity = true;
#pragma unroll
  for (int k = 0; k < 16; ++k) c.bda_identity = c.bda_identity && m[32 + k] == ((k % 5 == 0) ? 1.0f : 0.0f);
  c.span_r = SpanRcp{rcp_refined(P.span[0]), rcp_refined(P.span[1]), rcp_refined(P.span[2])};
  return c;
}
// the frustum point at depth d in the ego frame (nan_to_num'ed, bv2:612).  `mm`: the matrices, which callers read
// again per use (asm barrier) instead of keeping 40 scalars alive across the kernel
__device__ __forceinline__ void chain_point(const float* __restrict__ mm, const ChainCtx& c, float d, float& x, float& y,
                                            float& z) {
  if (c.bda_identity) frustum_point_rb_id(mm, c.rb, d, x, y, z);
  else frustum_point_rb(mm, c.rb, d, x, y, z);
}
// Tap of the sample at depth d as the kernel uses it.  Shared by the forward kernel and the diagnostic export, so
// that the export shows exactly the taps the forward sampled (the face fallback is a wave-level decision).
__device__ __forceinline__ VolTap chain_tap(const RenderParams& P, const float* __restrict__ mm, const ChainCtx& c, float d) {
  float x, y, z;
  chain_point(mm, c, d, x, y, z);
  bool near_face;
  VolTap tp = volume_tap_rcp(P, c.span_r, x, y, z, near_face);
  if (__any(near_face)) tp = volume_tap(P, x, y, z);             // the inside mask is the reference's bit for bit
  return tp;
}

// The same with the IEEE divisions of volume_tap for every lane: bit for bit what the camera backward's per-ray pass
// evaluates (frustum_point + volume_tap; frustum_point_rb has frustum_point's roundings).  For the samples whose
// cell RANK the forward draws (RANK instances): the per-ray pass looks a sample's record slot up as (its cell, the
// rank drawn here), so the two must agree on the cell of every sample, not in all but one in a million.
__device__ __forceinline__ VolTap chain_tap_exact(const RenderParams& P, const float* __restrict__ mm, const ChainCtx& c, float d) {
  float x, y, z;
  chain_point(mm, c, d, x, y, z);
  return volume_tap(P, x, y, z);
}

// tile -> ray with 32-bit arithmetic (decode_ray_wps of render_common.hpp divides 64-bit values)
__device__ __forceinline__ RayId decode_tile(const RenderParams& P, unsigned bid) {
  const int tiles_w = (P.fW + 7) >> 3, tiles_h = (P.fH + 7) >> 3;
  const int per_cam = tiles_h * tiles_w, tiles = P.B * P.N * per_cam;
  const int per_xcd = (tiles + 7) >> 3;
  const int t = (int) (bid & 7u) * per_xcd + (int) (bid >> 3);
  const int r = threadIdx.x & 63;
  RayId id;
  id.sub = threadIdx.x >> 6;
  const int tc = t < tiles ? t : tiles - 1;
  id.tile = tc;
  const int bn = tc / per_cam, tt = tc - bn * per_cam;
  const int ty = tt / tiles_w, tx = tt - ty * tiles_w;
  id.bn = bn;
  id.h = ty * 8 + (r >> 3);
  id.w = tx * 8 + (r & 7);
  id.live = t < tiles && id.h < P.fH && id.w < P.fW;
  if (id.h >= P.fH) id.h = P.fH - 1;
  if (id.w >= P.fW) id.w = P.fW - 1;
  id.b = bn / P.N;
  return id;
}

// NCH = composited channels (K + 3), rounded up by the launcher; channels >= K + 3 are skipped.
// NW = waves per 8 x 8 ray tile.  LDS: plan (2 KB) + round sums + dyn = max(S, NW * NCH) * 64 floats
// (tau / weights, then the merge buffer).
// rows != nullptr (a backward will follow): every inside sample's raw trilinear values -- the density
// feature from the density phase, the K + 3 composited channels from the gather phase -- are kept at
// rows[((tile * S + i) * P.CP + c) * 64 + ray]: 256 contiguous bytes per (tile, depth index, channel), so
// the backward's per-ray pass (same tiles, same lanes) reads them back coalesced instead of repeating
// the 8-tap gathers, which are all that pass was bound by.
// All samples sit on the reference's own fp32 coordinate chain (chain_tap above).
// `bid`: the workgroup's index among the camera tiles' workgroups (blockIdx.x of cam_fwd_direct_kernel; the merged
// render forward of render_fwd_merged.hip passes its own).
// RANK (training calls of the merged render forward): the tile also does the camera backward's RANK PASS -- every kept
// inside sample is counted into its cell (the voxel-grid cube of its floor taps) and takes its rank there, one
// run-aggregated atomic per run of lanes in a cell, issued in front of the sample's 84 channel loads and collected
// behind them.  The prepare step's first kernel (cam_cells_rank_kernel: 16 - 18 us of chain evaluations and atomics on
// the same samples) disappears, and with it the reason to keep camera forward, BEV forward and prepare pass as three
// launches on two streams: training forwards are one launch + the scan.
template <typename T, int NCH, bool ERT, int NW, bool RANK = false>
__device__ __forceinline__ void
cam_fwd_direct_tile(const unsigned bid, const RenderParams& P, const float* __restrict__ mats, const float* __restrict__ us,
                      const float* __restrict__ vs, const float* __restrict__ ds,
                      const float* __restrict__ mids, const float* __restrict__ beta_raw,
                      const T* __restrict__ dens, const T* __restrict__ sem, const T* __restrict__ rgb,
                      float* __restrict__ rgb_out, float* __restrict__ seg_out,
                      float* __restrict__ depth_out, int* __restrict__ term_out, float* __restrict__ rows,
                      const CamRankRefs rk = CamRankRefs{nullptr, nullptr, nullptr, 0}) {
  extern __shared__ __align__(16) float dyn[];
  __shared__ int4 plan[kPlanMax];
  __shared__ int keep_s[64];
  __shared__ float accd_s[64], tunit_s[64];
  __shared__ float part_s[2][NW][64];
  __shared__ unsigned char act_s[kPlanMax + 8];                  // the active depth indices, in order (then S)
  __shared__ unsigned char actf_s[kPlanMax];                     // 1 = active depth index
  float* wbuf = dyn;                                                // [S][64]
  const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const RayId id = decode_tile(P, bid);
  const int w = id.w, h = id.h;
  const int bn = __builtin_amdgcn_readfirstlane((int) id.bn);
  const int b = __builtin_amdgcn_readfirstlane(id.b);
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const int S = P.D - 1;
  const float* m = mats + (long) bn * 48;
  const float u = us[w], v = vs[h];
  const unsigned V = (unsigned) (P.Z * P.Y * P.X);
  // (NCH == 21 is the exact variant of the launcher: K = 18 at compile time, no scalar selects in the
  // channel -> tensor mapping of the gather)
  const int Kc = NCH == 21 ? 18 : P.K;
  const int nch = Kc + 3;
  VAMP_STAMP(0);
#ifdef VAMP_DIRECT_STAMPS
  const long long wall0 = wall_clock64();                          // 100 MHz, one clock for the whole device
#endif
  // `affine`: the chain is affine in the depth -- inv(ida)'s x and y rows do not depend on it (image-plane
  // augmentations never touch depth) -- so a ray's sample points lie on a line at constant speed
  const bool affine = m[2] == 0.0f && m[6] == 0.0f;               // uniform
  const ChainCtx cc = chain_ctx(P, m, u, v);
  // the ray's first and last frustum point: its ego-space length per unit of depth (bv2:426 along a line) and, for
  // the tile's corner rays, the plan's lines
  float ex0, ey0, ez0, ex1, ey1, ez1;
  chain_point(m, cc, ds[0], ex0, ey0, ez0);
  chain_point(m, cc, ds[S], ex1, ey1, ez1);
  const float inv_dd = 1.0f / (ds[S] - ds[0]);
  const float ray_len = sqrtf((ex1 - ex0) * (ex1 - ex0) + (ey1 - ey0) * (ey1 - ey0) + (ez1 - ez0) * (ez1 - ez0)) * inv_dd;
  // length of bin i (bv2:426: norm of consecutive frustum points): along an affine chain ray_len per unit of depth;
  // otherwise the two points themselves
  auto delta_at = [&](int i) -> float {
    if (affine) return ray_len * (ds[i + 1] - ds[i]);
    const float* mm = m;
    asm volatile("" : "+s"(mm));
    float x0, y0, z0, x1, y1, z1;
    chain_point(mm, cc, ds[i], x0, y0, z0);
    chain_point(mm, cc, ds[i + 1], x1, y1, z1);
    const float dx = x1 - x0, dy = y1 - y0, dz = z1 - z0;
    return sqrtf(dx * dx + dy * dy + dz * dz);
  };

  // ---- plan: depth indices of the tile that can hold inside samples.  The tile's samples at one depth index
  // are an affine image of the pixel rectangle, so the box of its four corner rays bounds all 64 (ray_plan.hpp).
  // Along an affine chain a corner ray's tap coordinates are a line through its first and last point (lanes 0, 7, 56
  // and 63 of a wave hold them): a lane per depth index evaluates the four corners with 12 fma instead of two full
  // chains per lane with their matrix, axis and lattice loads (round 5; the plan phase was a fifth of a tile's time).
  // fp32 lines: their error, a few 1e-4 voxel at x = 400, is far inside the box's 0.05 voxel of slack, which also
  // covers the rounding of the chain that decides the mask.  A non-affine chain takes plan_tile.
  if (VAMP_DIRECT_LINE_PLAN && affine) {
    __shared__ float cl[4][6];
    if (sub == 0 && (lane == 0 || lane == 7 || lane == 56 || lane == 63)) {
      float* c = cl[(lane == 0) ? 0 : (lane == 7 ? 1 : (lane == 56 ? 2 : 3))];
      const float e0[3] = {ex0, ey0, ez0}, e1[3] = {ex1, ey1, ez1};
      const float n1[3] = {(float) (P.X - 1), (float) (P.Y - 1), (float) (P.Z - 1)};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float sc = n1[a] / P.span[a];
        const float f0 = (e0[a] - P.lo[a]) * sc, f1 = (e1[a] - P.lo[a]) * sc;
        const float bb = (f1 - f0) * inv_dd;
        c[2 * a] = f0 - ds[0] * bb; c[2 * a + 1] = bb;
      }
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i < kPlanMax) {
      const float d = ds[min(i, P.D - 1)];
      float mn[3], mx[3];
      bool fin = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float f = __builtin_fmaf(d, cl[k][2 * a + 1], cl[k][2 * a]);
          fin = fin && fabsf(f) <= 1e30f;
          mn[a] = k ? fminf(mn[a], f) : f;
          mx[a] = k ? fmaxf(mx[a], f) : f;
        }
      }
      const float slack = 0.05f;
      const bool hit = mx[0] >= -slack && mn[0] <= (float) (P.X - 1) + slack &&
                       mx[1] >= -slack && mn[1] <= (float) (P.Y - 1) + slack &&
                       mx[2] >= -slack && mn[2] <= (float) (P.Z - 1) + slack;
      int4 ent;
      int lo, hi;
      brick_axis(mn[0], mx[0], P.X, lo, hi); ent.x = lo | (hi << 16);
      brick_axis(mn[1], mx[1], P.Y, lo, hi); ent.y = lo | (hi << 16);
      brick_axis(mn[2], mx[2], P.Z, lo, hi); ent.z = lo | (hi << 16);
      ent.w = (i < S && (hit || !fin)) ? 1 : 0;
      plan[i] = ent;
    }
  } else if (sub < 4) {
    plan_tile(P, m, us, vs, ds, __builtin_amdgcn_readlane(w, 0), __builtin_amdgcn_readlane(w, 63),
              __builtin_amdgcn_readlane(h, 0), __builtin_amdgcn_readlane(h, 63), sub, plan);
  }
  __syncthreads();
  PlanMask mk_all = plan_mask(plan);
  if (!affine) {
    // (never with the reference's image augmentations: inv(ida) then mixes the depth into u, v.)  The
    // skipped bins' optical depth is priced per unit of depth along a LINE -- without one, every depth
    // index is marched and carries its own length
    mk_all.lo = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
    mk_all.hi = S > 64 ? (S >= 128 ? ~0ull : ((1ull << (S - 64)) - 1ull)) : 0ull;
  }
  const int A = __builtin_popcountll(mk_all.lo) + __builtin_popcountll(mk_all.hi);
  // the active indices as a list (rank -> depth index) and as flags, so that the loops below index
  // LDS instead of walking the bit masks (75 scalar instructions per index)
  if (sub == 0) {
    const bool a_lo = (mk_all.lo >> lane) & 1ull, a_hi = (mk_all.hi >> lane) & 1ull;
    const int r_lo = __builtin_popcountll(mk_all.lo & ((1ull << lane) - 1ull));
    const int r_hi = __builtin_popcountll(mk_all.lo) + __builtin_popcountll(mk_all.hi & ((1ull << lane) - 1ull));
    if (a_lo) act_s[r_lo] = (unsigned char) lane;
    if (a_hi) act_s[r_hi] = (unsigned char) (64 + lane);
    if (lane < 8) act_s[A + lane] = (unsigned char) S;           // ranks past the end read S ("none")
    actf_s[lane] = a_lo ? 1 : 0;
    actf_s[64 + lane] = a_hi ? 1 : 0;
    // masked samples carry density(0) (Q6): optical depth per unit of depth of the skipped bins
    tunit_s[lane] = density_fast(dp, 0.f) * ray_len;
  }
  __syncthreads();
  VAMP_STAMP(1);

  // ---- density: tau_i = sigma_i delta_i of the active depth indices into LDS, in rounds of NW * G: a
  // wave takes G consecutive active indices, all its taps in flight together; after each round every
  // wave knows every ray's optical depth so far, and the tile stops once all 64 rays are saturated
  int S_eff = S;                      // depth indices the scan covers (early exit: the tile is saturated there)
  {
    const float tau_unit = tunit_s[lane];
    const __amdgpu_buffer_rsrc_t rs_d = make_rsrc(dens + (long) b * V, (size_t) V * sizeof(T));
    constexpr int G = ERT ? VAMP_DIRECT_G : VAMP_DIRECT_G_NOERT, R = NW * G;
    float carry = 0.f;                // sum of the active indices' tau so far (per ray)
    float d_inact = 0.f;              // sum of the skipped bins' depth extents so far (uniform)
    int cursor = 0;                   // depth index where the current round starts
    for (int r0 = 0, rd = 0; r0 < A; r0 += R, ++rd) {
      int idx[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        idx[g] = __builtin_amdgcn_readfirstlane((int) act_s[min(r0 + sub * G + g, A)]);
      float delta[G];
      bool in[G];
      PairTap pt[G];
      PairRaw raw[G][4];
      const float* mround = m;
      asm volatile("" : "+s"(mround));
#pragma unroll
      for (int g = 0; g < G; ++g) {
        in[g] = false; delta[g] = 0.f;
        if (idx[g] < S) {
          const VolTap tp = chain_tap(P, mround, cc, ds[idx[g]]);
          delta[g] = delta_at(idx[g]);                                                 // bv2:426
          in[g] = tp.inside;
          if (tp.inside) {
            pt[g] = pair_tap<T>(P, tp);
            raw[g][0] = ld_pair<T>(rs_d, pt[g].o00, 0u);
            raw[g][1] = ld_pair<T>(rs_d, pt[g].o01, 0u);
            raw[g][2] = ld_pair<T>(rs_d, pt[g].o10, 0u);
            raw[g][3] = ld_pair<T>(rs_d, pt[g].o11, 0u);
          }
        }
      }
      float psum = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (idx[g] < S) {
          float sv = 0.f;
          if (in[g]) {
            const float raw0 = pair_combine<T>(pt[g], raw[g]);
            if (rows) rows[(((long) id.tile * S + idx[g]) * P.CP) * 64 + lane] = raw0;
            sv = nan_to_num(raw0);
          }
          const float tau = density_fast(dp, sv) * delta[g];
          wbuf[idx[g] * 64 + lane] = tau;
          psum += tau;
        }
      }
      part_s[rd & 1][sub][lane] = psum;
      // where the next round starts, and the skipped bins in front of it
      const int nxt = (r0 + R >= A) ? S : __builtin_amdgcn_readfirstlane((int) act_s[r0 + R]);
      if (ERT) {
        // (ds is the depth-plane lattice: the skipped extent is the whole extent minus the active bins')
        float act_ext = 0.f;
        for (int k = r0; k < min(r0 + R, A); ++k) {
          const int i = __builtin_amdgcn_readfirstlane((int) act_s[k]);
          act_ext += ds[i + 1] - ds[i];
        }
        d_inact += (ds[nxt] - ds[cursor]) - act_ext;
      }
      cursor = nxt;
      __syncthreads();
      if (ERT) {
#pragma unroll
        for (int k = 0; k < NW; ++k) carry += part_s[rd & 1][k][lane];
        // (the margin covers the different summation order of the scan below)
        const bool done = carry + tau_unit * d_inact >= kTermOpticalDepth * 1.001f;
        if (__ballot(!done) == 0ull) { S_eff = nxt; break; }
      }
    }
  }
  __syncthreads();
  VAMP_STAMP(2);
#if defined(VAMP_DIRECT_STOP) && VAMP_DIRECT_STOP == 2
  if (wbuf[lane] != 12345.f) return;
#endif

  // ---- scan: lanes = (ray of this wave's 64 / NW, segment of the depth range); the bins of inactive
  // depth indices are all-masked samples: tau = density(0) * bin length
  {
    constexpr int RPW = 64 / NW, SEG = NW;
    const int r = sub * RPW + (lane % RPW), q = lane / RPW;
    const int L_ = (S_eff + SEG - 1) / SEG;
    const int a0 = min(S_eff, q * L_), a1 = min(S_eff, a0 + L_);
    const float tu = tunit_s[r];
    // (the running sums of the scan in fp64, each used once rounded to fp32: an fp32 running optical depth carries
    // an ulp of ~10 per step, i.e. 1e-6 relative on every later weight, and sum w mid + (1 - sum w) d_far a few ulp
    // of 70 m -- together 4e-5 m of depth that depend on how the samples fall into the four segments, i.e. on where
    // early termination cut the tile.  The reference's fp32 cumsum has noise of the same size around the same value.)
    double part = 0.0;
    for (int i = a0; i < a1; ++i) {
      const float tau = actf_s[i] ? wbuf[i * 64 + r] : tu * (ds[i + 1] - ds[i]);
      wbuf[i * 64 + r] = tau;
      part += (double) tau;
    }
    // exclusive prefix over the segments of the ray (bv2:431-433: exclusive cumsum)
    double incl = part;
#pragma unroll
    for (int o = 1; o < SEG; o <<= 1) {
      const double up = __shfl_up(incl, o * RPW, 64);
      if (q >= o) incl += up;
    }
    double cum = incl - part;
    int keep = S;
    double aw = 0.0, ad = 0.0;
    for (int i = a0; i < a1; ++i) {
      const float tau = wbuf[i * 64 + r];
      const float wgt = composite_weight(tau, (float) cum);        // bv2:430-434
      wbuf[i * 64 + r] = wgt;
      aw += (double) wgt;
      ad = __builtin_fma((double) wgt, (double) mids[i], ad);
      cum += (double) tau;
      // samples 0 .. i are kept; the optical depth in front of sample i + 1 is `cum`
      if (ERT && keep == S && !(cum < (double) kTermOpticalDepth)) keep = i + 1;
    }
#pragma unroll
    for (int o = RPW; o < 64; o <<= 1) {
      aw += __shfl_xor(aw, o, 64);
      ad += __shfl_xor(ad, o, 64);
      keep = min(keep, __shfl_xor(keep, o, 64));
    }
    if (q == 0) { keep_s[r] = min(keep, S_eff < S ? S_eff : S); accd_s[r] = (float) (ad + (1.0 - aw) * (double) P.d_far); }   // bv2:436,440
  }
  __syncthreads();
  VAMP_STAMP(3);
#if defined(VAMP_DIRECT_STOP) && VAMP_DIRECT_STOP == 3
  if (keep_s[lane] != 12345) return;
#endif

  // ---- gather: the kept inside samples' K + 3 composited channels
  const int keep = keep_s[lane];
  int Se = keep;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) Se = max(Se, __shfl_xor(Se, o, 64));
  Se = __builtin_amdgcn_readfirstlane(Se);
  float acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = 0.f;
  {
    // the active indices below Se, dealt to the waves by rank
    PlanMask mk = mk_all;
    mask_truncate(mk, Se);
    const int Ae = __builtin_popcountll(mk.lo) + __builtin_popcountll(mk.hi);
    const int k0 = (sub * Ae) / NW, k1 = ((sub + 1) * Ae) / NW;
    const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(sem + (long) b * Kc * V, (size_t) Kc * V * sizeof(T));
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(rgb + (long) b * 3 * V, (size_t) 3 * V * sizeof(T));
    const unsigned vbytes = V * (unsigned) sizeof(T);               // one channel, bytes (launcher: K * V * es < 2 GB)
    for (int k = k0; k < k1; ++k) {
      const int i = __builtin_amdgcn_readfirstlane((int) act_s[k]);
      const float* mg = m;
      asm volatile("" : "+s"(mg));
      const VolTap tp = RANK ? chain_tap_exact(P, mg, cc, ds[i]) : chain_tap(P, mg, cc, ds[i]);
      // the sample's rank in its cell (wave-uniform control flow: every lane takes part in the run detection)
      LaneRun run{false, 0, 0};
      int rbase = 0;
      bool ranked = false;
      if (RANK) {
        ranked = id.live && tp.inside && i < keep;
        const int key = ranked ? pack_cell_key(tp.ix0, tp.iy0, tp.iz0) : 0;
        const long cell = key_to_cell(key, P.Y, P.X, (unsigned) b, rk.ncell_b);
        run = lane_run(ranked, cell, lane);
        if (run.head) rbase = atomicAdd(rk.cnt + cell, run.len);       // (returns under the channel loads below)
      }
      if (tp.inside && i < keep) {
        const PairTap pt = pair_tap<T>(P, tp);
        const float wgt = wbuf[i * 64 + lane];
        float s[NCH];
        // (the per-channel byte offsets are recomputed here, one s_mul each: hoisted out of the loop they
        // were 21 more live scalars than the register file has, and came back through v_readlane)
        unsigned vb = vbytes;
        asm volatile("" : "+s"(vb));
        // channels in batches of kDirectCB: all 4 * CB pair loads of a batch are issued back to back
#pragma unroll
        for (int c0 = 0; c0 < NCH; c0 += kDirectCB) {
          PairRaw raw[kDirectCB][4];
#pragma unroll
          for (int uu = 0; uu < kDirectCB; ++uu) {
            if (c0 + uu < NCH) {
              const int cc = min(c0 + uu, nch - 1);
              const bool is_sem = cc < Kc;
              const __amdgpu_buffer_rsrc_t rs = is_sem ? rs_s : rs_r;
              const unsigned so = (unsigned) (is_sem ? cc : cc - Kc) * vb;
              raw[uu][0] = ld_pair<T>(rs, pt.o00, so);
              raw[uu][1] = ld_pair<T>(rs, pt.o01, so);
              raw[uu][2] = ld_pair<T>(rs, pt.o10, so);
              raw[uu][3] = ld_pair<T>(rs, pt.o11, so);
            }
          }
#pragma unroll
          for (int uu = 0; uu < kDirectCB; ++uu)
            if (c0 + uu < NCH) s[c0 + uu] = pair_combine<T>(pt, raw[uu]);
        }
        if (rows) {
          float* rr = rows + (((long) id.tile * S + i) * P.CP + 1) * 64 + lane;
#pragma unroll
          for (int c = 0; c < NCH; ++c)
            if (c < nch) rr[c * 64] = s[c];
        }
        // nan_to_num of the sampled features (bv2:421) only where something is not finite:
        // sum_c 0 * s_c is nan exactly then
        float chk = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) chk = __builtin_fmaf(s[c], 0.f, chk);
        if (__builtin_expect(chk != chk, 0)) {
          asm volatile("" ::: "memory");                     // (keeps the rare path a branch: if-converted it is ~95 instructions per sample)
#pragma unroll
          for (int c = 0; c < NCH; ++c) s[c] = nan_to_num(s[c]);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = __builtin_fmaf(wgt, s[c], acc[c]);
      }
      if (RANK) {
        rbase = __shfl(rbase, ranked ? run.start : lane, 64);
        if (ranked) rk.rank[((long) id.tile * S + i) * 64 + lane] = rbase + (lane - run.start);
      }
    }
  }
  if (RANK && threadIdx.x < 64) {
    // the tile's depth (its longest LIVE ray): the per-ray pass takes the deepest tiles first
    int sl = id.live ? keep : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sl = max(sl, __shfl_xor(sl, o, 64));
    if (threadIdx.x == 0 && id.live) rk.tile_se[id.tile] = sl;       // (lane 0 is live exactly when the workgroup has a tile of its own)
  }
  __syncthreads();                                                   // all reads of wbuf are done
  VAMP_STAMP(4);
#ifdef VAMP_DIRECT_STAMPS
  if (threadIdx.x == 0 && bid < 4096)
    g_direct_stamps[bid * 8 + 6] = ((long long) A << 32) | ((long long) S_eff << 16) | (long long) Se;
#endif

  // ---- merge the waves' partial sums; stores
  float* xa = dyn;                                                   // [NW][NCH][64]
#pragma unroll
  for (int c = 0; c < NCH; ++c) xa[(sub * NCH + c) * 64 + lane] = acc[c];
  __syncthreads();
  const int HW = P.fH * P.fW;
  const int pix = h * P.fW + w;
  if (id.live) {
    for (int c = sub; c < nch; c += NW) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) t += xa[(k * NCH + c) * 64 + lane];
      if (c < Kc) seg_out[((long) bn * Kc + c) * HW + pix] = t;
      else rgb_out[((long) bn * 3 + (c - Kc)) * HW + pix] = t;
    }
    if (sub == 0) {
      depth_out[(long) bn * HW + pix] = accd_s[lane];
      if (term_out) term_out[(long) bn * HW + pix] = keep;
    }
  }
  VAMP_STAMP(5);
#ifdef VAMP_DIRECT_STAMPS
  if (threadIdx.x == 0 && bid < 4096)
    g_direct_stamps[bid * 8 + 7] = (long long) (((unsigned long long) (wall0 & 0xffffffffll) << 32) | (unsigned long long) (wall_clock64() & 0xffffffffll));
#endif
}

}  // namespace vamp
