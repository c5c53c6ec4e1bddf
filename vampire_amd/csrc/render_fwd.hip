// RENDER forward kernels for gfx950.  Reference call site:
// volume_rendering_from_multiple_views, base_vampire2.py:391-467.
//
//  pack_volume         [B,c,Z,Y,X] x3  ->  channel-last [B,Z,Y,X,CP] (density, sem, rgb)
//  render_cam_fwd_plan  (geometry from the matrices) a wave = the 64 rays of an 8 x 8 tile x one
//                      depth range; a per-tile plan (ray_plan.hpp) finds the depth indices whose
//                      corner box misses the volume -- they are composited without evaluating the
//                      frustum chain -- and deals the others evenly to the tile's four waves;
//                      each inside sample is one 8-tap gather of CP contiguous floats
//  render_cam_fwd      (caller-supplied geom tensor) every lane gathers its own 8 taps of CP
//                      contiguous floats from global memory
//  (the BEV branch lives in render_bev.hip)
// HBM/L2-bound gathers and a short scan: no MFMA.
#include "render_common.hpp"
#include "ray_plan.hpp"

namespace vamp {

// ---------------------------------------------------------------------------
// pack: thread per voxel on the load side, rows transposed through LDS on the store side
// ---------------------------------------------------------------------------
template <typename T, int CP4>
__global__ void __launch_bounds__(256)
pack_volume_kernel(RenderParams P, const T* __restrict__ dens, const T* __restrict__ sem,
                   const T* __restrict__ rgb, float* __restrict__ packed) {
  // The channel-first volumes are read coalesced along x (a thread per voxel) and the 96-byte
  // channel-last rows leave through LDS, so that a workgroup's 256 rows go out as one contiguous
  // 24 KB run of aligned float4 stores (a thread storing its own row puts 16 bytes into each of
  // 64 lines per instruction).
  constexpr int CP = CP4 * 4, ST = CP + 1;       // odd row stride: conflict-free both ways
  __shared__ float rows[256 * ST];
  const long V = (long) P.Z * P.Y * P.X, total = V * P.B;
  const long gid0 = (long) blockIdx.x * 256;
  const long gid = gid0 + threadIdx.x;
  if (gid < total) {
    const long b = gid / V, vox = gid % V;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      float val = 0.f;
      if (c == 0) val = ldf(dens, b * V + vox);
      else if (c <= P.K) val = ldf(sem, (b * P.K + (c - 1)) * V + vox);
      else if (c <= P.K + 3) val = ldf(rgb, (b * 3 + (c - 1 - P.K)) * V + vox);
      rows[threadIdx.x * ST + c] = val;
    }
  }
  __syncthreads();
  const long nrow = min((long) 256, total - gid0);
  float4* dst = reinterpret_cast<float4*>(packed + gid0 * CP);
#pragma unroll
  for (int i = 0; i < CP4; ++i) {
    const int q = i * 256 + threadIdx.x;         // float4 index inside the workgroup's run
    const int r = q / CP4, c4 = q % CP4;
    if (r < nrow) {
      const float* src = rows + r * ST + c4 * 4;
      dst[q] = make_float4(src[0], src[1], src[2], src[3]);
    }
  }
}

// ---------------------------------------------------------------------------
// camera branch forward
// ---------------------------------------------------------------------------
template <int LPR, int CP4, bool WPS>
__global__ void __launch_bounds__(256)
render_cam_fwd_kernel(RenderParams P, const float* __restrict__ geom, const float* __restrict__ mats,
                      const float* __restrict__ us, const float* __restrict__ vs,
                      const float* __restrict__ ds, const float* __restrict__ mids,
                      const float* __restrict__ beta_raw, const float* __restrict__ packed,
                      float* __restrict__ rgb_out, float* __restrict__ seg_out,
                      float* __restrict__ depth_out) {
  constexpr int CP = CP4 * 4;
  static_assert(!WPS || LPR == 4, "wave-per-chunk mapping uses the 4 waves of the workgroup");
  __shared__ float xmerge[WPS ? 4 * (CP + 2) * 64 : 1];
  const RayId id = WPS ? decode_ray_wps(P) : decode_ray<LPR>(P);
  const bool live = id.live;
  const int w = id.w, h = id.h, sub = id.sub, b = id.b;
  const long bn = id.bn;

  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const int S = P.D - 1;
  const int L = (S + LPR - 1) / LPR;
  const int i0 = sub * L, i1 = min(S, i0 + L);
  const float* m = mats ? mats + bn * 48 : nullptr;
  const float u = us[w], v = vs[h];
  const long V = (long) P.Z * P.Y * P.X;
  const float* vol = packed + (long) b * V * CP;
  const long pstride = (long) P.fH * P.fW * 3;     // geom plane stride
  const float* gp = geom ? geom + ((bn * P.D * P.fH + h) * P.fW + w) * 3 : nullptr;

  float px, py, pz;
  auto point = [&](int i, float& x, float& y, float& z) {
    if (gp) {
      const float* q = gp + (long) i * pstride;
      x = q[0]; y = q[1]; z = q[2];
    } else {
      frustum_point(m, u, v, ds[i], x, y, z);
      x = nan_to_num_geom(x); y = nan_to_num_geom(y); z = nan_to_num_geom(z);
    }
  };
  if (i0 < i1) point(i0, px, py, pz);

  float acc[CP];                         // acc[0] = sum of weights, acc[1..K+3] = sem / rgb sums
#pragma unroll
  for (int c = 0; c < CP; ++c) acc[c] = 0.f;
  float acc_depth = 0.f, cum = 0.f;

  for (int i = i0; i < i1; ++i) {
    float qx, qy, qz;
    point(i + 1, qx, qy, qz);
    const VolTap tp = volume_tap(P, px, py, pz);
    float s[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) s[c] = 0.f;
    if (tp.inside) {
      gather_taps<CP4>(P, vol, tp, s);
      // nan_to_num of the sampled features (bv2:421) only where something is not finite:
      // sum_c 0 * s_c is nan exactly then
      float chk = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) chk = __builtin_fmaf(s[c], 0.f, chk);
      if (chk != chk) {
#pragma unroll
        for (int c = 0; c < CP; ++c) s[c] = nan_to_num(s[c]);
      }
    }
    const float sigma = density_fwd(dp, s[0]);                    // masked sample -> density(0)
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float delta = sqrtf(dx * dx + dy * dy + dz * dz);       // bv2:426
    const float tau = sigma * delta;
    const float wgt = (1.0f - expf(-tau)) * expf(-cum);           // bv2:430-434
    cum += tau;
    acc[0] += wgt;
    acc_depth = __builtin_fmaf(wgt, mids[i], acc_depth);
#pragma unroll
    for (int c = 1; c < CP; ++c) acc[c] = __builtin_fmaf(wgt, s[c], acc[c]);
    px = qx; py = qy; pz = qz;
  }

  if (WPS) {
    // merge the four depth chunks (= waves) of each ray through LDS
    const int lane = threadIdx.x & 63;
    float* xc = xmerge;                                // [4][64] optical depth of each chunk
    float* xa = xmerge + 4 * 64;                       // [4][CP + 1][64] scaled partial sums
    xc[sub * 64 + lane] = cum;
    __syncthreads();
    float excl = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < sub) excl += xc[k * 64 + lane];
    const float scale = expf(-excl);                   // transmittance in front of this chunk
    xa[(sub * (CP + 1) + CP) * 64 + lane] = acc_depth * scale;
#pragma unroll
    for (int c = 0; c < CP; ++c) xa[(sub * (CP + 1) + c) * 64 + lane] = acc[c] * scale;
    __syncthreads();
    if (sub == 0) {
      acc_depth = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc_depth += xa[(k * (CP + 1) + CP) * 64 + lane];
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) t += xa[(k * (CP + 1) + c) * 64 + lane];
        acc[c] = t;
      }
    }
  } else if (LPR > 1) {
    // transmittance of everything in front of this lane's chunk
    const float scale = expf(-ray_excl_prefix<LPR>(cum, sub));
    acc_depth = ray_sum<LPR>(acc_depth * scale);
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = ray_sum<LPR>(acc[c] * scale);
  }
  if (!live || sub != 0) return;
  const long HW = (long) P.fH * P.fW;
  const long pix = (long) h * P.fW + w;
  depth_out[bn * HW + pix] = acc_depth + (1.0f - acc[0]) * P.d_far;   // bv2:436,440
#pragma unroll
  for (int c = 1; c < CP; ++c) {
    if (c <= P.K) seg_out[(bn * P.K + (c - 1)) * HW + pix] = acc[c];
    else if (c <= P.K + 3) rgb_out[(bn * 3 + (c - 1 - P.K)) * HW + pix] = acc[c];
  }
}


// ---------------------------------------------------------------------------
// early ray termination table (render_common.hpp): one wave per 8 x 8 ray tile marches the
// density channel only -- 8 taps of 4 bytes from the channel-first density volume, 2.5 MB at
// cfg-B, instead of 8 x 96 -- front to back and stops as soon as all 64 rays are saturated
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64)
cam_term_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                const float* __restrict__ vs, const float* __restrict__ ds,
                const float* __restrict__ beta_raw, const T* __restrict__ dens, int* __restrict__ term) {
  const RayId id = decode_ray_wps(P);
  const long bn = __builtin_amdgcn_readfirstlane((int) id.bn);
  const int b = __builtin_amdgcn_readfirstlane(id.b);
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const int S = P.D - 1;
  const float* m = mats + bn * 48;
  const float u = us[id.w], v = vs[id.h];
  const long V = (long) P.Z * P.Y * P.X;
  const T* vol = dens + (long) b * V;
  auto point = [&](int i, float& x, float& y, float& z) {
    frustum_point(m, u, v, ds[i], x, y, z);
    x = nan_to_num_geom(x); y = nan_to_num_geom(y); z = nan_to_num_geom(z);
  };
  float px, py, pz, qx, qy, qz;
  point(0, px, py, pz);
  float cum = 0.f;
  int keep = S;
  bool done = false, entered = false;
  const float sigma_out = density_fwd(dp, 0.f);
  for (int i = 0; i < S; ++i) {
    point(i + 1, qx, qy, qz);
    const VolTap tp = volume_tap(P, px, py, pz);
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float delta = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!done && entered && !tp.inside) {
      // the ray has left the (convex) volume: every further sample is masked and adds the same
      // density(0) * delta, so the index at which the ray saturates -- if ever -- is known
      const float tau = sigma_out * delta;
      const float k = (tau > 0.f) ? ceilf((kTermOpticalDepth - cum) / tau) : 3.0e9f;
      keep = (k < (float) (S - i)) ? i + (int) k : S;
      done = true;
    }
    entered = entered || tp.inside;
    float s0 = 0.f;
    if (tp.inside) {
      const int x1 = min(tp.ix0 + 1, P.X - 1), y1 = min(tp.iy0 + 1, P.Y - 1), z1 = min(tp.iz0 + 1, P.Z - 1);
      const float wx1 = (tp.ix0 + 1 < P.X) ? tp.wx1 : 0.f, wy1 = (tp.iy0 + 1 < P.Y) ? tp.wy1 : 0.f;
      const float wz1 = (tp.iz0 + 1 < P.Z) ? tp.wz1 : 0.f;
      const long r00 = ((long) tp.iz0 * P.Y + tp.iy0) * P.X, r01 = ((long) tp.iz0 * P.Y + y1) * P.X;
      const long r10 = ((long) z1 * P.Y + tp.iy0) * P.X, r11 = ((long) z1 * P.Y + y1) * P.X;
      const float a00 = tp.wx0 * ldf(vol, r00 + tp.ix0) + wx1 * ldf(vol, r00 + x1);
      const float a01 = tp.wx0 * ldf(vol, r01 + tp.ix0) + wx1 * ldf(vol, r01 + x1);
      const float a10 = tp.wx0 * ldf(vol, r10 + tp.ix0) + wx1 * ldf(vol, r10 + x1);
      const float a11 = tp.wx0 * ldf(vol, r11 + tp.ix0) + wx1 * ldf(vol, r11 + x1);
      s0 = nan_to_num(tp.wz0 * (tp.wy0 * a00 + wy1 * a01) + wz1 * (tp.wy0 * a10 + wy1 * a11));
    }
    cum += density_fwd(dp, s0) * delta;
    // samples 0 .. i are kept; the optical depth in front of sample i + 1 is `cum`
    if (!done && !(cum < kTermOpticalDepth)) { keep = i + 1; done = true; }
    if (__ballot(!done) == 0ull) break;
    px = qx; py = qy; pz = qz;
  }
  if (id.live) term[(bn * P.fH + id.h) * P.fW + id.w] = keep;
}

int launch_cam_term(const VampRenderDesc* d, const RenderParams& P, const float* mats, const float* us,
                    const float* vs, const float* ds, const float* beta, const void* density_feature,
                    int* term, hipStream_t s) {
  const long tiles = (long) P.B * P.N * ((P.fH + 7) / 8) * ((P.fW + 7) / 8);
  const unsigned grid = (unsigned) ((tiles + 7) / 8 * 8);
  if (d->in_dtype == VAMP_F32)
    VAMP_TIMED(kProfCamTerm, s, (cam_term_kernel<float><<<grid, 64, 0, s>>>(
        P, mats, us, vs, ds, beta, static_cast<const float*>(density_feature), term)));
  else
    VAMP_TIMED(kProfCamTerm, s, (cam_term_kernel<__hip_bfloat16><<<grid, 64, 0, s>>>(
        P, mats, us, vs, ds, beta, static_cast<const __hip_bfloat16*>(density_feature), term)));
  return check_launch("cam_term_kernel");
}

// ---------------------------------------------------------------------------
// camera branch forward with a per-tile plan (geometry evaluated from the matrices)
// ---------------------------------------------------------------------------
// SAVE: also store the gathered values of every inside sample at samples[((tile * S + i) * CP + c) * 64 + ray]
// (the layout of render_cam_direct.hip: 256 contiguous bytes per tile, depth index and channel)
// (CP floats, before the nan_to_num of bv2:421) for the backward's per-ray pass.
template <int CP4, bool SAVE>
__global__ void __launch_bounds__(256, 3)
render_cam_fwd_plan_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                           const float* __restrict__ vs, const float* __restrict__ ds,
                           const float* __restrict__ mids, const float* __restrict__ beta_raw,
                           const float* __restrict__ packed, float* __restrict__ rgb_out,
                           float* __restrict__ seg_out, float* __restrict__ depth_out,
                           float* __restrict__ samples, const int* __restrict__ term) {
  constexpr int CP = CP4 * 4;
  __shared__ float xmerge[4 * (CP + 2) * 64];
  __shared__ int4 plan[kPlanMax];
  const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = depth range
  const int lane = threadIdx.x & 63;
  const RayId id = decode_ray_wps(P);
  const bool live = id.live;
  const int w = id.w, h = id.h;
  const long bn = __builtin_amdgcn_readfirstlane((int) id.bn);
  const int b = __builtin_amdgcn_readfirstlane(id.b);

  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const int S = P.D - 1;                                           // <= kPlanMax (launcher)
  const float* m = mats + bn * 48;
  const float u = us[w], v = vs[h];
  const long V = (long) P.Z * P.Y * P.X;
  const float* vol = packed + (long) b * V * CP;

  // which depth indices of the tile can hold inside samples; then this wave's share of the march
  plan_tile(P, m, us, vs, ds, __builtin_amdgcn_readlane(w, 0), __builtin_amdgcn_readlane(w, 63),
            __builtin_amdgcn_readlane(h, 0), __builtin_amdgcn_readlane(h, 63), sub, plan);
  __syncthreads();
  PlanMask mk = plan_mask(plan);
  // (the skipped indices below are priced per unit of depth along a LINE; when inv(ida) mixes the depth
  // into u, v -- never with the reference's augmentations -- the chain is not affine in the depth and
  // every index is marched with its own bin length)
  if (!(m[2] == 0.0f && m[6] == 0.0f)) mk.lo = mk.hi = ~0ull;
  // early ray termination: this ray's samples from index `keep` on are dropped, and the tile is
  // done at the largest `keep` of its 64 rays
  const int keep = term ? term[(bn * P.fH + h) * P.fW + w] : S;
  int Se = keep;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) Se = max(Se, __shfl_xor(Se, o, 64));
  Se = __builtin_amdgcn_readfirstlane(Se);
  mask_truncate(mk, Se);
  int i0, i1;
  plan_share(mk, Se, sub, i0, i1);

  auto point = [&](int i, float& x, float& y, float& z) {
    frustum_point(m, u, v, ds[i], x, y, z);
    x = nan_to_num_geom(x); y = nan_to_num_geom(y); z = nan_to_num_geom(z);
  };

  float acc[CP];                         // acc[0] = sum of weights, acc[1..K+3] = sem / rgb sums
#pragma unroll
  for (int c = 0; c < CP; ++c) acc[c] = 0.f;
  float acc_depth = 0.f, cum = 0.f;

  if (i0 < i1) {
    // ego-space length of a ray per unit of depth: the bin lengths of the depth indices that are
    // skipped (bv2:426 for samples the mask zeroes anyway; equal to the exact norm up to rounding)
    float px, py, pz, qx, qy, qz;
    point(i0, px, py, pz);
    point(i0 + 1, qx, qy, qz);
    float dl_unit;
    {
      const float dx = qx - px, dy = qy - py, dz = qz - pz;
      dl_unit = sqrtf(dx * dx + dy * dy + dz * dz) / (ds[i0 + 1] - ds[i0]);
    }
    const float sigma_out = density_fwd(dp, 0.f);                  // masked sample -> density(0) (Q6)
    bool have_p = true;                                            // (px, py, pz) is the point of index i0
    int jn = mask_next(mk, i0);
    for (int i = i0; i < i1; ++i) {
      if (i != jn) {
        // all 64 samples of this depth index lie outside the volume: s = 0
        const float tau = sigma_out * (dl_unit * (ds[i + 1] - ds[i]));
        const float wgt = composite_weight(tau, cum);
        cum += tau;
        acc[0] += wgt;
        acc_depth = __builtin_fmaf(wgt, mids[i], acc_depth);
        have_p = false;
        continue;
      }
      jn = mask_next(mk, i + 1);
      if (!have_p) point(i, px, py, pz);
      point(i + 1, qx, qy, qz);
      VolTap tp = volume_tap(P, px, py, pz);
      tp.inside = tp.inside && i < keep;
      float s[CP];
#pragma unroll
      for (int c = 0; c < CP; ++c) s[c] = 0.f;
      if (tp.inside) {
        gather_taps<CP4>(P, vol, tp, s);
        if (SAVE) {
          float* rr = samples + (((long) id.tile * S + i) * CP) * 64 + (threadIdx.x & 63);
#pragma unroll
          for (int c = 0; c < CP; ++c) rr[c * 64] = s[c];
        }
        // nan_to_num of the sampled features (bv2:421) only where something is not finite:
        // sum_c 0 * s_c is nan exactly then (two chains, so that they pack)
        float chk0 = 0.f, chk1 = 0.f;
#pragma unroll
        for (int c = 0; c < CP; c += 2) {
          chk0 = __builtin_fmaf(s[c], 0.f, chk0);
          chk1 = __builtin_fmaf(s[c + 1], 0.f, chk1);
        }
        if (chk0 + chk1 != chk0 + chk1) {
#pragma unroll
          for (int c = 0; c < CP; ++c) s[c] = nan_to_num(s[c]);
        }
      }
      const float sigma = density_fwd(dp, s[0]);                    // masked sample -> density(0)
      const float dx = qx - px, dy = qy - py, dz = qz - pz;
      const float delta = sqrtf(dx * dx + dy * dy + dz * dz);       // bv2:426
      const float tau = sigma * delta;
      const float wgt = composite_weight(tau, cum);       // bv2:430-434
      cum += tau;
      acc[0] += wgt;
      acc_depth = __builtin_fmaf(wgt, mids[i], acc_depth);
#pragma unroll
      for (int c = 1; c < CP; ++c) acc[c] = __builtin_fmaf(wgt, s[c], acc[c]);
      px = qx; py = qy; pz = qz;
      have_p = true;
    }
  }

  // merge the four depth ranges (= waves) of each ray through LDS
  float* xc = xmerge;                                // [4][64] optical depth of each range
  float* xa = xmerge + 4 * 64;                       // [4][CP + 1][64] scaled partial sums
  xc[sub * 64 + lane] = cum;
  __syncthreads();
  float excl = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < sub) excl += xc[k * 64 + lane];
  const float scale = exp_acc(-excl);                // transmittance in front of this range
  xa[(sub * (CP + 1) + CP) * 64 + lane] = acc_depth * scale;
#pragma unroll
  for (int c = 0; c < CP; ++c) xa[(sub * (CP + 1) + c) * 64 + lane] = acc[c] * scale;
  __syncthreads();
  if (sub != 0 || !live) return;
  acc_depth = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc_depth += xa[(k * (CP + 1) + CP) * 64 + lane];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) t += xa[(k * (CP + 1) + c) * 64 + lane];
    acc[c] = t;
  }
  const long HW = (long) P.fH * P.fW;
  const long pix = (long) h * P.fW + w;
  depth_out[bn * HW + pix] = acc_depth + (1.0f - acc[0]) * P.d_far;   // bv2:436,440
#pragma unroll
  for (int c = 1; c < CP; ++c) {
    if (c <= P.K) seg_out[(bn * P.K + (c - 1)) * HW + pix] = acc[c];
    else if (c <= P.K + 3) rgb_out[(bn * 3 + (c - 1 - P.K)) * HW + pix] = acc[c];
  }
}

// ---------------------------------------------------------------------------
// diagnostics + standalone geometry
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
render_indices_kernel(RenderParams P, const float* __restrict__ geom, const float* __restrict__ mats,
                      const float* __restrict__ us, const float* __restrict__ vs,
                      const float* __restrict__ ds, uint8_t* __restrict__ inside,
                      int16_t* __restrict__ ix0, int16_t* __restrict__ iy0,
                      int16_t* __restrict__ iz0) {
  const long HW = (long) P.fH * P.fW;
  const long total = (long) P.B * P.N * (P.D - 1) * HW;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int w = gid % P.fW, h = (gid / P.fW) % P.fH;
  const int i = (gid / HW) % (P.D - 1);
  const long bn = gid / (HW * (P.D - 1));
  float x, y, z;
  if (geom) {
    const float* q = geom + (((bn * P.D + i) * P.fH + h) * P.fW + w) * 3;
    x = q[0]; y = q[1]; z = q[2];
  } else {
    frustum_point(mats + bn * 48, us[w], vs[h], ds[i], x, y, z);
    x = nan_to_num_geom(x); y = nan_to_num_geom(y); z = nan_to_num_geom(z);
  }
  const VolTap t = volume_tap(P, x, y, z);
  inside[gid] = t.inside ? 1 : 0;
  ix0[gid] = t.inside ? (int16_t) t.ix0 : 0;
  iy0[gid] = t.inside ? (int16_t) t.iy0 : 0;
  iz0[gid] = t.inside ? (int16_t) t.iz0 : 0;
}

__global__ void __launch_bounds__(256)
frustum_geometry_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                        const float* __restrict__ vs, const float* __restrict__ ds,
                        float* __restrict__ geom) {
  const long HW = (long) P.fH * P.fW;
  const long total = (long) P.B * P.N * P.D * HW;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int w = gid % P.fW, h = (gid / P.fW) % P.fH;
  const int i = (gid / HW) % P.D;
  const long bn = gid / (HW * P.D);
  float x, y, z;
  frustum_point(mats + bn * 48, us[w], vs[h], ds[i], x, y, z);
  geom[gid * 3 + 0] = nan_to_num_geom(x);
  geom[gid * 3 + 1] = nan_to_num_geom(y);
  geom[gid * 3 + 2] = nan_to_num_geom(z);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int launch_pack(const RenderParams& P, int in_dtype, const void* dens, const void* sem,
                const void* rgb, float* packed, hipStream_t s) {
  const long total = (long) P.B * P.Z * P.Y * P.X;
  const unsigned grid = (unsigned) ((total + 255) / 256);
#define VAMP_PACK(T, CP4)                                                                  \
  VAMP_TIMED(kProfPack, s, (pack_volume_kernel<T, CP4><<<grid, 256, 0, s>>>(               \
      P, static_cast<const T*>(dens), static_cast<const T*>(sem), static_cast<const T*>(rgb), packed)))
  if (in_dtype == VAMP_F32) {
    if (P.CP == 12) VAMP_PACK(float, 3); else if (P.CP == 24) VAMP_PACK(float, 6); else VAMP_PACK(float, 8);
  } else {
    if (P.CP == 12) VAMP_PACK(__hip_bfloat16, 3); else if (P.CP == 24) VAMP_PACK(__hip_bfloat16, 6); else VAMP_PACK(__hip_bfloat16, 8);
  }
#undef VAMP_PACK
  return check_launch("pack_volume_kernel");
}

}  // namespace vamp

using namespace vamp;

namespace vamp {
size_t packed_bytes(const VampRenderDesc* d) {
  const RenderParams P = to_params(d);
  return align_up((size_t) d->B * d->Z * d->Y * d->X * P.CP * sizeof(float), 256);
}
}  // namespace vamp

extern "C" {


size_t vamp_render_workspace_bytes(const VampRenderDesc* d) {
  if (!d) return 0;
  // packed volume + backward scratch (v1: packed gradient volume; v2: per-sample buffers) + the
  // per-ray early-termination table
  return render_base_bytes(d) + cam_term_bytes(d);
}

int vamp_render_camera_terminate(const VampRenderDesc* d, const float* mats, const float* us,
                                 const float* vs, const float* ds, const float* beta,
                                 const void* density_feature, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && us && vs && ds && density_feature, "null pointer");
  VAMP_REQUIRE(beta || d->density_mode == VAMP_DENSITY_SIGMOID, "beta is NULL");
  const size_t need = vamp_render_workspace_bytes(d);
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  return launch_cam_term(d, to_params(d), mats, us, vs, ds, beta, density_feature, cam_term_ptr(d, workspace),
                         static_cast<hipStream_t>(stream));
}

size_t vamp_render_term_offset(const VampRenderDesc* d) { return d ? render_base_bytes(d) : 0; }

size_t vamp_render_samples_bytes(const VampRenderDesc* d) {
  if (!d) return 0;
  const RenderParams P = to_params(d);
  // [8 x 8 ray tile][depth index][channel][ray of the tile] (ragged tiles padded)
  const size_t tiles = (size_t) d->B * d->N * ((d->fH + 7) / 8) * ((d->fW + 7) / 8);
  return align_up(tiles * 64 * (d->D - 1) * P.CP * sizeof(float), 256);
}

int vamp_render_camera_forward(const VampRenderDesc* d, const float* geom, const float* mats,
                               const float* us, const float* vs, const float* ds,
                               const float* mids, const float* beta, const void* density_feature,
                               const void* semantic, const void* rgb, float* rgb_out,
                               float* seg_out, float* depth_out, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return vamp_render_camera_forward_ex(d, geom, mats, us, vs, ds, mids, beta, density_feature, semantic, rgb,
                                       rgb_out, seg_out, depth_out, workspace, workspace_bytes, 0, stream);
}

int vamp_render_camera_forward_ex(const VampRenderDesc* d, const float* geom, const float* mats,
                                  const float* us, const float* vs, const float* ds,
                                  const float* mids, const float* beta, const void* density_feature,
                                  const void* semantic, const void* rgb, float* rgb_out,
                                  float* seg_out, float* depth_out, void* workspace,
                                  size_t workspace_bytes, int flags, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(geom || (mats && us && vs && ds), "need geom or (mats, us, vs, ds)");
  VAMP_REQUIRE(mids && density_feature && semantic && rgb && rgb_out && seg_out && depth_out,
               "null pointer");
  VAMP_REQUIRE(beta || d->density_mode == VAMP_DENSITY_SIGMOID, "beta is NULL");
  const bool planned = !geom && d->D - 1 <= kPlanMax;
  if ((flags & VAMP_CAMFWD_DIRECT) && planned) {
    // one kernel on the channel-first volumes (render_cam_direct.hip): no packed copy, the
    // termination table is a by-product (written when the workspace can hold it); with SAVE_SAMPLES the
    // samples' values stay behind the base region for the backward's per-ray pass
    const bool ert = !(flags & VAMP_CAMFWD_NO_ERT);
    int* term = (workspace && workspace_bytes >= vamp_render_workspace_bytes(d)) ? cam_term_ptr(d, workspace) : nullptr;
    float* rows = nullptr;
    if (flags & VAMP_CAMFWD_SAVE_SAMPLES) {
      const size_t need = vamp_render_workspace_bytes(d) + vamp_render_samples_bytes(d);
      if (!workspace || workspace_bytes < need)
        return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
      rows = reinterpret_cast<float*>(static_cast<char*>(workspace) + vamp_render_workspace_bytes(d));
    }
    const RenderParams P = to_params(d);
    return launch_cam_fwd_direct(d, P, mats, us, vs, ds, mids, beta, density_feature, semantic, rgb,
                                 rgb_out, seg_out, depth_out, term, ert, rows, static_cast<hipStream_t>(stream));
  }
  const bool save = (flags & VAMP_CAMFWD_SAVE_SAMPLES) && planned;
  const bool ert = planned && !(flags & VAMP_CAMFWD_NO_ERT);
  const size_t need = save ? vamp_render_workspace_bytes(d) + vamp_render_samples_bytes(d)
                           : (ert ? vamp_render_workspace_bytes(d) : packed_bytes(d));
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  const RenderParams P = to_params(d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* packed = static_cast<float*>(workspace);
  float* samples = save ? reinterpret_cast<float*>(static_cast<char*>(workspace) + vamp_render_workspace_bytes(d))
                        : nullptr;
  int* term = ert ? cam_term_ptr(d, workspace) : nullptr;
  if (ert && !(flags & VAMP_CAMFWD_TERM_VALID)) {
    if (int e = launch_cam_term(d, P, mats, us, vs, ds, beta, density_feature, term, s)) return e;
  } else if (ert) {
    if (int e = debug_expect_range(term, (size_t) d->B * d->N * d->fH * d->fW, 0, d->D - 1, s,
                                   "VAMP_CAMFWD_TERM_VALID: the workspace holds a termination table")) return e;
  }
  // VAMP_CAMFWD_PACK_ONLY / _PACKED_VALID: the channel-last copy as a call of its own (a caller
  // with two streams packs beside its other work and marches when both are there)
  if (!(flags & VAMP_CAMFWD_PACKED_VALID))
    if (int e = launch_pack(P, d->in_dtype, density_feature, semantic, rgb, packed, s)) return e;
  if (flags & VAMP_CAMFWD_PACK_ONLY) return VAMP_OK;
  constexpr int LPR = 4;
  const unsigned grid = ray_grid<LPR>(P);
  // geometry from the matrices and at most kPlanMax samples per ray: the planned march
  if (planned) {
#define VAMP_CAMP(CP4)                                                                     \
  do {                                                                                     \
    if (samples)                                                                           \
      VAMP_TIMED(kProfCamFwd, s, (render_cam_fwd_plan_kernel<CP4, true><<<grid, 256, 0, s>>>(  \
          P, mats, us, vs, ds, mids, beta, packed, rgb_out, seg_out, depth_out, samples, term)));    \
    else                                                                                   \
      VAMP_TIMED(kProfCamFwd, s, (render_cam_fwd_plan_kernel<CP4, false><<<grid, 256, 0, s>>>( \
          P, mats, us, vs, ds, mids, beta, packed, rgb_out, seg_out, depth_out, nullptr, term)));    \
  } while (0)
    if (P.CP == 12) VAMP_CAMP(3); else if (P.CP == 24) VAMP_CAMP(6); else VAMP_CAMP(8);
#undef VAMP_CAMP
    return check_launch("render_cam_fwd_plan_kernel");
  }
#define VAMP_CAM(CP4)                                                                        \
  VAMP_TIMED(kProfCamFwd, s, (render_cam_fwd_kernel<LPR, CP4, true><<<grid, 256, 0, s>>>(    \
      P, geom, mats, us, vs, ds, mids, beta, packed, rgb_out, seg_out, depth_out)))
  if (P.CP == 12) VAMP_CAM(3); else if (P.CP == 24) VAMP_CAM(6); else VAMP_CAM(8);
#undef VAMP_CAM
  return check_launch("render_cam_fwd_kernel");
}

int vamp_render_indices(const VampRenderDesc* d, const float* geom, const float* mats,
                        const float* us, const float* vs, const float* ds, uint8_t* inside,
                        int16_t* ix0, int16_t* iy0, int16_t* iz0, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(geom || (mats && us && vs && ds), "need geom or (mats, us, vs, ds)");
  VAMP_REQUIRE(inside && ix0 && iy0 && iz0, "null output");
  RenderParams P = to_params(d);
  const long total = (long) d->B * d->N * (d->D - 1) * d->fH * d->fW;
  render_indices_kernel<<<(unsigned) ((total + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      P, geom, mats, us, vs, ds, inside, ix0, iy0, iz0);
  return check_launch("render_indices_kernel");
}

int vamp_frustum_geometry(const VampRenderDesc* d, const float* mats, const float* us,
                          const float* vs, const float* ds, float* geom, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && us && vs && ds && geom, "null pointer");
  RenderParams P = to_params(d);
  const long total = (long) d->B * d->N * d->D * d->fH * d->fW;
  frustum_geometry_kernel<<<(unsigned) ((total + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      P, mats, us, vs, ds, geom);
  return check_launch("frustum_geometry_kernel");
}

}  // extern "C"
