// RENDER forward kernels for gfx950.  Reference call site:
// volume_rendering_from_multiple_views, base_vampire2.py:391-467.
//
//  pack_volume         [B,c,Z,Y,X] x3  ->  channel-last [B,Z,Y,X,CP] (density, sem, rgb)
//  render_cam_fwd_brick  (geometry from the matrices) a wave = the 64 rays of an 8 x 8 tile x one
//                      depth chunk; the taps of a depth index come from a voxel brick staged in
//                      LDS by LDS-DMA (brick.hpp), the next brick is in flight while the current
//                      one is composited; depth indices whose corner box misses the volume are
//                      composited without evaluating the frustum chain
//  render_cam_fwd      (caller-supplied geom tensor) every lane gathers its own 8 taps of CP
//                      contiguous floats from global memory
//  (the BEV branch lives in render_bev.hip)
// HBM/L2-bound gathers and a short scan: no MFMA.
#include "render_common.hpp"
#include "brick.hpp"

namespace vamp {

// ---------------------------------------------------------------------------
// pack: thread per voxel, CP/4 float4 stores (lanes contiguous -> coalesced)
// ---------------------------------------------------------------------------
template <typename T, int CP4>
__global__ void __launch_bounds__(256)
pack_volume_kernel(RenderParams P, const T* __restrict__ dens, const T* __restrict__ sem,
                   const T* __restrict__ rgb, float* __restrict__ packed) {
  const long V = (long) P.Z * P.Y * P.X;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * P.B) return;
  const long b = gid / V, vox = gid % V;
  float v[CP4 * 4];
#pragma unroll
  for (int c = 0; c < CP4 * 4; ++c) {
    float val = 0.f;
    if (c == 0) val = ldf(dens, b * V + vox);
    else if (c <= P.K) val = ldf(sem, (b * P.K + (c - 1)) * V + vox);
    else if (c <= P.K + 3) val = ldf(rgb, (b * 3 + (c - 1 - P.K)) * V + vox);
    v[c] = val;
  }
  float4* dst = reinterpret_cast<float4*>(packed + gid * (CP4 * 4));
#pragma unroll
  for (int q = 0; q < CP4; ++q) dst[q] = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
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
// camera branch forward, LDS bricks (geometry evaluated from the matrices)
// ---------------------------------------------------------------------------
#ifdef VAMP_STAMP
// diagnostic build only (tools/ablate.sh): cycle sums per phase of the march, all waves
__device__ unsigned long long g_stamp[16];
#define STAMP(t) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); } while (0)
#define STAMP_ADD(k, a, b) do { st[k] += (b) - (a); } while (0)
#else
#define STAMP(t) do { } while (0)
#define STAMP_ADD(k, a, b) do { } while (0)
#endif
template <int CP4, int CAP>
__global__ void __launch_bounds__(256, 3)
render_cam_fwd_brick_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                            const float* __restrict__ vs, const float* __restrict__ ds,
                            const float* __restrict__ mids, const float* __restrict__ beta_raw,
                            const float* __restrict__ packed, float* __restrict__ rgb_out,
                            float* __restrict__ seg_out, float* __restrict__ depth_out) {
  constexpr int CP = CP4 * 4;
  constexpr int kBrick4 = CAP * BrickLayout<CP4>::RS4;            // float4 per wave
  constexpr int kMerge4 = (CP + 2) * 64;                           // 4 * (CP + 2) * 64 floats
  __shared__ float4 smem[(4 * kBrick4 > kMerge4) ? 4 * kBrick4 : kMerge4];
  __shared__ int4 plan[kPlanMax];
  static_assert(CAP >= 0, "CAP == 0: every lane gathers its own taps from global memory");
  float* xmerge = reinterpret_cast<float*>(smem);                  // aliases the bricks after the march
  const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = depth chunk
  float4* brick = smem + sub * kBrick4;
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

#ifdef VAMP_STAMP
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long cnt[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, tb = 0, te = 0;
#endif
  STAMP(tb);
  // which depth indices of the tile can hold inside samples, and their boxes; then this wave's
  // share of the march
  plan_tile(P, m, us, vs, ds, __builtin_amdgcn_readlane(w, 0), __builtin_amdgcn_readlane(w, 63),
            __builtin_amdgcn_readlane(h, 0), __builtin_amdgcn_readlane(h, 63), sub, plan);
  __syncthreads();
  const PlanMask mk = plan_mask(plan);
  int i0, i1;
  plan_share(mk, S, sub, i0, i1);
  STAMP(te);
  STAMP_ADD(0, tb, te);                                            // plan

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
#ifndef VAMP_ABL_NOPREFETCH
    if constexpr (CAP > 0)
      if (jn < i1) brick_prefetch<CP4, CAP>(P, vol, plan_box(plan, jn), brick);
#endif
    for (int i = i0; i < i1; ++i) {
      STAMP(t0);
      if (i != jn) {
        // all 64 samples of this depth index lie outside the volume: s = 0
        const float tau = sigma_out * (dl_unit * (ds[i + 1] - ds[i]));
        const float wgt = (1.0f - __expf(-tau)) * __expf(-cum);
        cum += tau;
        acc[0] += wgt;
        acc_depth = __builtin_fmaf(wgt, mids[i], acc_depth);
        have_p = false;
        STAMP(t1);
        STAMP_ADD(1, t0, t1);                                        // skipped index
        continue;
      }
#ifdef VAMP_ABL_ONLYDMA
      VolTap tp;
      tp.inside = true; tp.ix0 = lane & 1; tp.iy0 = 0; tp.iz0 = 0; tp.wx0 = tp.wx1 = tp.wy0 = tp.wy1 = tp.wz0 = tp.wz1 = 0.5f;
      tp.fx = tp.fy = tp.fz = 0.f;
      {
        const BrickBox b0 = plan_box(plan, i);
        tp.ix0 = b0.lo[0]; tp.iy0 = b0.lo[1]; tp.iz0 = b0.lo[2];
      }
      qx = px; qy = py; qz = pz;
#else
      if (!have_p) point(i, px, py, pz);
      point(i + 1, qx, qy, qz);
      const VolTap tp = volume_tap(P, px, py, pz);
#endif
      float s[CP];
#pragma unroll
      for (int c = 0; c < CP; ++c) s[c] = 0.f;
      STAMP(t1);
      STAMP_ADD(2, t0, t1);                                          // points + taps
      if constexpr (CAP == 0) {
        if (tp.inside) gather_taps<CP4>(P, vol, tp, s);
      } else
      if (__ballot(tp.inside) != 0ull)
#ifdef VAMP_ABL_NOPREFETCH
        brick_gather<CP4, CAP>(P, vol, tp, tp.inside, plan_box(plan, i), false, brick, s
#else
        brick_gather<CP4, CAP>(P, vol, tp, tp.inside, plan_box(plan, i), true, brick, s
#endif
#ifdef VAMP_STAMP
                               , cnt
#endif
                               );
      STAMP(t2);
      STAMP_ADD(3, t1, t2);                                          // brick wait + gather
      // the next brick travels while this sample is composited
      jn = mask_next(mk, i + 1);
#ifdef VAMP_ABL_NOPREFETCH
      if (false) {
#else
      if (CAP > 0 && jn < i1) {
#endif
        wait_vmem();
        wait_lds();
        brick_prefetch<CP4, CAP>(P, vol, plan_box(plan, jn), brick);
      }
      STAMP(t3);
      STAMP_ADD(4, t2, t3);                                          // prefetch issue
#ifdef VAMP_ABL_ONLYDMA
      acc[0] += s[0] + s[5];
      continue;
#endif
      if (tp.inside) {
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
      const float wgt = (1.0f - __expf(-tau)) * __expf(-cum);       // bv2:430-434
      cum += tau;
      acc[0] += wgt;
      acc_depth = __builtin_fmaf(wgt, mids[i], acc_depth);
#pragma unroll
      for (int c = 1; c < CP; ++c) acc[c] = __builtin_fmaf(wgt, s[c], acc[c]);
      px = qx; py = qy; pz = qz;
      have_p = true;
      STAMP(t4);
      STAMP_ADD(5, t3, t4);                                          // composite
    }
  }
  wait_vmem();
#ifdef VAMP_STAMP
  STAMP(t4);
  st[6] = t4 - tb;                                                 // whole wave up to the merge
  if (lane == 0) {
    for (int k = 0; k < 7; ++k) atomicAdd(&g_stamp[k], st[k]);
    atomicAdd(&g_stamp[7], 1ull);
    for (int k = 0; k < 6; ++k) atomicAdd(&g_stamp[8 + k], cnt[k]);
  }
#endif

  // merge the four depth chunks (= waves) of each ray through LDS
  float* xc = xmerge;                                // [4][64] optical depth of each chunk
  float* xa = xmerge + 4 * 64;                       // [4][CP + 1][64] scaled partial sums
  __syncthreads();                                   // every wave is done with its brick
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
size_t cam_bwd_v2_bytes(const VampRenderDesc* d);
size_t packed_bytes(const VampRenderDesc* d) {
  const RenderParams P = to_params(d);
  return align_up((size_t) d->B * d->Z * d->Y * d->X * P.CP * sizeof(float), 256);
}
}  // namespace vamp

extern "C" {

#ifdef VAMP_DUMP_BOXES
int vamp_debug_boxes(int* out, unsigned* n, int reset) {
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(vamp::g_nbox), sizeof(unsigned)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vamp::g_boxes), 65536 * 8 * sizeof(int)) != hipSuccess) return -1;
  if (reset) { unsigned z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(vamp::g_nbox), &z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
#ifdef VAMP_STAMP
int vamp_debug_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(vamp::g_stamp), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(vamp::g_stamp), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

size_t vamp_render_workspace_bytes(const VampRenderDesc* d) {
  if (!d) return 0;
  // packed volume + backward scratch (v1: packed gradient volume; v2: per-sample buffers)
  const size_t pb = packed_bytes(d), v2 = cam_bwd_v2_bytes(d);
  return pb + (pb > v2 ? pb : v2);
}

int vamp_render_camera_forward(const VampRenderDesc* d, const float* geom, const float* mats,
                               const float* us, const float* vs, const float* ds,
                               const float* mids, const float* beta, const void* density_feature,
                               const void* semantic, const void* rgb, float* rgb_out,
                               float* seg_out, float* depth_out, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(geom || (mats && us && vs && ds), "need geom or (mats, us, vs, ds)");
  VAMP_REQUIRE(mids && density_feature && semantic && rgb && rgb_out && seg_out && depth_out,
               "null pointer");
  VAMP_REQUIRE(beta || d->density_mode == VAMP_DENSITY_SIGMOID, "beta is NULL");
  const size_t need = packed_bytes(d);
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  const RenderParams P = to_params(d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* packed = static_cast<float*>(workspace);
  if (int e = launch_pack(P, d->in_dtype, density_feature, semantic, rgb, packed, s)) return e;
  constexpr int LPR = 4;
  const unsigned grid = ray_grid<LPR>(P);
  // geometry from the matrices and at most 4 x 32 samples per ray: LDS bricks.  Brick capacity per
  // wave: 96 rows of 112 B (CP = 24) = 10.5 KB, 43 KB per workgroup -> three workgroups per CU.
  if (!geom && d->D - 1 <= 128 && (size_t) d->Z * d->Y * d->X * (P.CP / 4) < 0x7fffffffu) {
#define VAMP_CAMB(CP4, CAP)                                                                   \
  VAMP_TIMED(kProfCamFwd, s, (render_cam_fwd_brick_kernel<CP4, CAP><<<grid, 256, 0, s>>>(     \
      P, mats, us, vs, ds, mids, beta, packed, rgb_out, seg_out, depth_out)))
#ifdef VAMP_CAM_DIRECT
    if (P.CP == 12) VAMP_CAMB(3, 0); else if (P.CP == 24) VAMP_CAMB(6, 0); else VAMP_CAMB(8, 0);
#else
    if (P.CP == 12) VAMP_CAMB(3, 128); else if (P.CP == 24) VAMP_CAMB(6, 96); else VAMP_CAMB(8, 64);
#endif
#undef VAMP_CAMB
    return check_launch("render_cam_fwd_brick_kernel");
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
