// Camera-branch render backward, v2: gather formulation -- no atomics on the volume, every
// output element written exactly once, deterministic.
//
//  1. invert_mats      device 4x4 inverses of the three render matrices (ego -> frustum
//                      coordinates), used only to bound candidate boxes
//  2. cam_bwd_ray      LPR lanes per ray re-march the samples once (8-tap gather of the
//                      packed volume), keep (s0, delta, q) per sample in LDS, merge the
//                      chunks with wave shuffles and emit one record per sample: the
//                      compositing weight w_i, dL/ds_i[0], and the continuous tap coordinates
//                      (fx, fy, fz) exactly as the forward computed them (NaN = masked)
//  3. cam_bwd_gather   GL lanes per voxel: bound, per camera, the (depth, h, w) index box of
//                      samples whose trilinear support can contain the voxel, stream the
//                      sample records of the box, and accumulate
//                      w_tap * dL/ds_i[c] for the 1+K+3 channels in registers; a shuffle
//                      reduction and an LDS transpose give coalesced channel-first stores.
// See render_bwd.hip for the compositing algebra.
#include "render_common.hpp"

#include <algorithm>

namespace vamp {

// ---------------------------------------------------------------------------
// 2. per-ray pass
// ---------------------------------------------------------------------------
template <int LPR, int CP4>
__global__ void __launch_bounds__(256)
cam_bwd_ray_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                   const float* __restrict__ vs, const float* __restrict__ ds,
                   const float* __restrict__ mids, const float* __restrict__ beta_raw,
                   const float* __restrict__ packed, const float* __restrict__ g_rgb,
                   const float* __restrict__ g_seg, const float* __restrict__ g_depth,
                   float* __restrict__ Wbuf, float* __restrict__ G0buf, float* __restrict__ FX,
                   float* __restrict__ FY, float* __restrict__ FZ, int* __restrict__ KEY,
                   float* __restrict__ Gcl, float* __restrict__ grad_beta, int L) {
  constexpr int CP = CP4 * 4;
  extern __shared__ float lds[];              // [3][L][256]: s0, delta (sign = no-grad flag), q
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float* l_s0 = lds;
  float* l_dl = lds + (long) L * 256;
  float* l_q = lds + (long) 2 * L * 256;

  // wave-per-depth-chunk mapping (render_common.hpp): lanes = the 64 rays of an 8x8 tile
  static_assert(LPR == 4, "the four waves of the workgroup are the four depth chunks");
  __shared__ float xm[2 * 4 * 64];
  const RayId id = decode_ray_wps(P);
  const bool live = id.live;
  const int w = id.w, h = id.h, sub = id.sub, b = id.b;
  const long bn = id.bn;
  const long ray = (bn * P.fH + h) * P.fW + w;
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const int S = P.D - 1;
  const int i0 = min(S, sub * L), i1 = min(S, i0 + L);
  const float* m = mats + bn * 48;
  const float u = us[w], v = vs[h];
  const long V = (long) P.Z * P.Y * P.X;
  const float* vol = packed + (long) b * V * CP;
  const long HW = (long) P.fH * P.fW;
  const long pix = (long) h * P.fW + w;

  float G[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    float gv = 0.f;
    if (live) {
      if (c >= 1 && c <= P.K) gv = g_seg ? g_seg[(bn * P.K + (c - 1)) * HW + pix] : 0.f;
      else if (c > P.K && c <= P.K + 3) gv = g_rgb ? g_rgb[(bn * 3 + (c - 1 - P.K)) * HW + pix] : 0.f;
    }
    G[c] = gv;
  }
  const float Gd = (live && g_depth) ? g_depth[bn * HW + pix] : 0.f;
  if (live && sub == 0) {
    float4* dst = reinterpret_cast<float4*>(Gcl + ray * CP);
#pragma unroll
    for (int q = 0; q < CP4; ++q) dst[q] = make_float4(G[q * 4], G[q * 4 + 1], G[q * 4 + 2], G[q * 4 + 3]);
  }

  // ---- march the chunk once ----
  float px, py, pz, qx, qy, qz;
  auto point = [&](int i, float& x, float& y, float& z) {
    frustum_point(m, u, v, ds[i], x, y, z);
    x = nan_to_num_geom(x); y = nan_to_num_geom(y); z = nan_to_num_geom(z);
  };
  if (i0 < i1) point(i0, px, py, pz);
  float cum = 0.f, A = 0.f;
  for (int i = i0; i < i1; ++i) {
    point(i + 1, qx, qy, qz);
    const VolTap tp = volume_tap(P, px, py, pz);
    float s[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) s[c] = 0.f;
    if (tp.inside) {
      gather_taps<CP4>(P, vol, tp, s);
    }
    const bool fin = (s[0] == s[0]) && (fabsf(s[0]) <= 3.402823466e+38f);
    const float s0 = nan_to_num(s[0]);
    float qv = Gd * (mids[i] - P.d_far);
#pragma unroll
    for (int c = 1; c < CP; ++c) qv = __builtin_fmaf(G[c], nan_to_num(s[c]), qv);
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float delta = sqrtf(dx * dx + dy * dy + dz * dz);
    const float tau = density_fwd(dp, s0) * delta;
    A = __builtin_fmaf((1.0f - expf(-tau)) * expf(-cum), qv, A);
    cum += tau;
    const int j = i - i0;
    l_s0[j * 256 + tid] = s0;
    // a sample passes gradient to s[0] only if it is inside and finite: flag in the sign
    l_dl[j * 256 + tid] = (tp.inside && fin) ? delta : -delta;
    l_q[j * 256 + tid] = qv;
    if (live) {
      const long sidx = ((bn * S + i) * P.fH + h) * P.fW + w;
      const float nanv = __builtin_nanf("");
      FX[sidx] = tp.inside ? tp.fx : nanv;     // NaN: masked sample, matches no voxel
      FY[sidx] = tp.fy;
      FZ[sidx] = tp.fz;
      // floor taps packed 11/11/10 bits (+1 so that 0 = masked): the gather tests candidates
      // with one load and integer compares
      KEY[sidx] = tp.inside ? ((tp.ix0 + 1) | ((tp.iy0 + 1) << 11) | ((tp.iz0 + 1) << 22)) : 0;
    }
    px = qx; py = qy; pz = qz;
  }

  // ---- merge the LPR chunks of the ray ----
  float scale = 1.f, suffix = 0.f;
  {
    const int lane = tid & 63;
    xm[sub * 64 + lane] = cum;
    __syncthreads();
    float excl = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < sub) excl += xm[k * 64 + lane];
    scale = expf(-excl);
    xm[(4 + sub) * 64 + lane] = scale * A;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k > sub) suffix += xm[(4 + k) * 64 + lane];     // sum_{m > sub} scale_m A_m
  }

  // ---- second loop over the LDS copies: emit w_i and dL/ds_i[0] ----
  float cl = 0.f, prefix = 0.f, dbeta = 0.f;
  for (int i = i0; i < i1; ++i) {
    const int j = i - i0;
    const float s0 = l_s0[j * 256 + tid];
    const float dl = l_dl[j * 256 + tid];
    const float qv = l_q[j * 256 + tid];
    const float delta = fabsf(dl);
    const float tau = density_fwd(dp, s0) * delta;
    const float wloc = (1.0f - expf(-tau)) * expf(-cl);
    const float Tn = scale * expf(-(cl + tau));
    cl += tau;
    prefix = __builtin_fmaf(wloc, qv, prefix);
    const float R = scale * (A - prefix) + suffix;
    const float dtau = qv * Tn - R;
    float dsig_ds, dsig_db;
    density_bwd(dp, s0, dsig_ds, dsig_db);
    dbeta = __builtin_fmaf(dtau * delta, dsig_db, dbeta);
    if (live) {
      const long sidx = ((bn * S + i) * P.fH + h) * P.fW + w;
      Wbuf[sidx] = scale * wloc;
      G0buf[sidx] = (dl > 0.f) ? dtau * delta * dsig_ds : 0.f;
    }
  }
  if (P.density_mode == VAMP_DENSITY_SDF_LAPLACE) {
    float vsum = live ? dbeta : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vsum += __shfl_down(vsum, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = vsum;
    __syncthreads();
    if (tid == 0) {
      const float tot = red[0] + red[1] + red[2] + red[3];
      const float sgn = (beta_raw[0] > 0.f) ? 1.f : ((beta_raw[0] < 0.f) ? -1.f : 0.f);
      atomicAdd(grad_beta, sgn * tot);
    }
  }
}

// ---------------------------------------------------------------------------
// 3. per-voxel gather
// ---------------------------------------------------------------------------
constexpr int GL = 16;              // lanes per voxel
constexpr int VPB = 256 / GL;       // voxels per workgroup: an x-run

// ego -> (u, v, depth) in frustum coordinates (inverse of get_geometry, bv2:328-349)
__device__ __forceinline__ void project_corner(const float* __restrict__ pm, float x, float y,
                                               float z, float& u, float& v, float& dd) {
  // pm holds the inverses in the order of the render matrices: [ida, inv(E), inv(bda)]
  Vec4 p{x, y, z, 1.0f};
  p = matvec(pm + 32, p);
  p = matvec(pm + 16, p);
  dd = p.z;
  const float zc = fmaxf(p.z, 1e-6f);
  p.x = p.x / zc;
  p.y = p.y / zc;
  p = matvec(pm, p);
  u = p.x;
  v = p.y;
}

// weight of tap index `iv` for continuous coordinate f (aten: w0 = floor+1-f, w1 = f-floor)
__device__ __forceinline__ float tap_weight(float f, float iv) {
  const float fl = floorf(f);
  return (fl == iv) ? (fl + 1.0f) - f : ((fl + 1.0f == iv) ? f - fl : 0.f);
}

// One thread per gather workgroup (an x-run of VPB voxels): bit n of the result is set when
// camera n's sample lattice can reach the run's trilinear support (same bound as the
// per-voxel box below, over the whole run).
__global__ void __launch_bounds__(256)
cam_bwd_cull_kernel(RenderParams P, const float* __restrict__ pmats, const float* __restrict__ us,
                    const float* __restrict__ vs, const float* __restrict__ ds,
                    unsigned* __restrict__ cull, int runs_x, const int* __restrict__ total_e,
                    int cap_e) {
  if (total_e && *total_e <= cap_e) return;       // fallback only: the binned lists fit
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long) runs_x * P.Y * P.Z * P.B;
  if (gid >= total) return;
  const int rx = gid % runs_x, iy = (gid / runs_x) % P.Y;
  const int iz = (gid / ((long) runs_x * P.Y)) % P.Z, b = gid / ((long) runs_x * P.Y * P.Z);
  const int S = P.D - 1;
  const float ex = P.span[0] / (float) (P.X - 1), ey = P.span[1] / (float) (P.Y - 1),
              ez = P.span[2] / (float) (P.Z - 1);
  const int x0 = rx * VPB, x1 = min(P.X, x0 + VPB) - 1;
  const float X0 = P.lo[0] + ((float) x0 - 1.01f) * ex, X1 = P.lo[0] + ((float) x1 + 1.01f) * ex;
  const float Y0 = P.lo[1] + ((float) iy - 1.01f) * ey, Y1 = P.lo[1] + ((float) iy + 1.01f) * ey;
  const float Z0 = P.lo[2] + ((float) iz - 1.01f) * ez, Z1 = P.lo[2] + ((float) iz + 1.01f) * ez;
  const float du = (P.fW > 1) ? (us[P.fW - 1] - us[0]) / (float) (P.fW - 1) : 1.f;
  const float dv = (P.fH > 1) ? (vs[P.fH - 1] - vs[0]) / (float) (P.fH - 1) : 1.f;
  const float d0 = ds[0], dstep = (ds[P.D - 1] - d0) / (float) (P.D - 1);
  unsigned mask = 0;
  for (int n = 0; n < P.N; ++n) {
    const float* pm = pmats + ((long) b * P.N + n) * 48;
    float umin = 3e38f, umax = -3e38f, vmin = 3e38f, vmax = -3e38f, zmin = 3e38f, zmax = -3e38f;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float cu, cv, cd;
      project_corner(pm, (k & 1) ? X1 : X0, (k & 2) ? Y1 : Y0, (k & 4) ? Z1 : Z0, cu, cv, cd);
      bad = bad || !(cd > 0.05f) || !(cu == cu) || !(cv == cv);
      umin = fminf(umin, cu); umax = fmaxf(umax, cu);
      vmin = fminf(vmin, cv); vmax = fmaxf(vmax, cv);
      zmin = fminf(zmin, cd); zmax = fmaxf(zmax, cd);
    }
    if (!(zmax >= d0 - dstep)) continue;
    bool seen = true;
    if (!bad) {
      const int i_hi = min(S - 1, (int) ceilf((zmax - d0) / dstep + 0.02f));
      const int i_lo = max(0, (int) floorf((zmin - d0) / dstep - 0.02f));
      const int w_lo = max(0, (int) floorf((umin - us[0]) / du - 0.05f));
      const int w_hi = min(P.fW - 1, (int) ceilf((umax - us[0]) / du + 0.05f));
      const int h_lo = max(0, (int) floorf((vmin - vs[0]) / dv - 0.05f));
      const int h_hi = min(P.fH - 1, (int) ceilf((vmax - vs[0]) / dv + 0.05f));
      seen = i_lo <= i_hi && w_lo <= w_hi && h_lo <= h_hi;
    }
    if (seen) mask |= 1u << n;
  }
  cull[gid] = mask;
}

template <int CP4>
__global__ void __launch_bounds__(256)
cam_bwd_gather_kernel(RenderParams P, const float* __restrict__ pmats, const float* __restrict__ us,
                      const float* __restrict__ vs, const float* __restrict__ ds,
                      const float* __restrict__ FX, const float* __restrict__ FY,
                      const float* __restrict__ FZ, const int* __restrict__ KEY,
                      const float* __restrict__ Wbuf, const float* __restrict__ G0buf,
                      const float* __restrict__ Gcl, const unsigned* __restrict__ cull, float* __restrict__ gdens,
                      float* __restrict__ gsem, float* __restrict__ grgb,
                      const int* __restrict__ total_e, int cap_e) {
  if (total_e && *total_e <= cap_e) return;       // fallback only: the binned lists fit
  constexpr int CP = CP4 * 4;
  __shared__ float outs[CP][VPB + 1];
  const int tid = threadIdx.x;
  const int g = tid / GL, l = tid % GL;
  const int ix = blockIdx.x * VPB + g, iy = blockIdx.y;
  const int iz = blockIdx.z % P.Z, b = blockIdx.z / P.Z;
  const bool vox_ok = ix < P.X;
  const int S = P.D - 1;
  const int nch = 1 + P.K + 3;
  const float ex = P.span[0] / (float) (P.X - 1), ey = P.span[1] / (float) (P.Y - 1),
              ez = P.span[2] / (float) (P.Z - 1);
  // ego-space box of sample positions whose taps can include this voxel: |f - index| < 1
  const float X0 = P.lo[0] + ((float) ix - 1.01f) * ex, X1 = P.lo[0] + ((float) ix + 1.01f) * ex;
  const float Y0 = P.lo[1] + ((float) iy - 1.01f) * ey, Y1 = P.lo[1] + ((float) iy + 1.01f) * ey;
  const float Z0 = P.lo[2] + ((float) iz - 1.01f) * ez, Z1 = P.lo[2] + ((float) iz + 1.01f) * ez;
  const float du = (P.fW > 1) ? (us[P.fW - 1] - us[0]) / (float) (P.fW - 1) : 1.f;
  const float dv = (P.fH > 1) ? (vs[P.fH - 1] - vs[0]) / (float) (P.fH - 1) : 1.f;
  const float d0 = ds[0];
  const float dstep = (ds[P.D - 1] - d0) / (float) (P.D - 1);
  const long HW = (long) P.fH * P.fW;
  const float fix = (float) ix, fiy = (float) iy, fiz = (float) iz;

  float acc[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) acc[c] = 0.f;

  const unsigned vis = cull[((long) blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x];
  for (int n = 0; n < P.N; ++n) {
    if (!((vis >> n) & 1u)) continue;                 // no sample of camera n can reach this x-run
    const long bn = (long) b * P.N + n;
    // lane l projects corner (l & 7); min/max over the 8 corners with xor shuffles
    float cu, cv, cd;
    project_corner(pmats + bn * 48, (l & 1) ? X1 : X0, (l & 2) ? Y1 : Y0, (l & 4) ? Z1 : Z0, cu, cv, cd);
    float bad = (!(cd > 0.05f) || !(cu == cu) || !(cv == cv)) ? 1.f : 0.f;
    float umin = cu, umax = cu, vmin = cv, vmax = cv, zmin = cd, zmax = cd;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      umin = fminf(umin, __shfl_xor(umin, o, GL)); umax = fmaxf(umax, __shfl_xor(umax, o, GL));
      vmin = fminf(vmin, __shfl_xor(vmin, o, GL)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, GL));
      zmin = fminf(zmin, __shfl_xor(zmin, o, GL)); zmax = fmaxf(zmax, __shfl_xor(zmax, o, GL));
      bad = fmaxf(bad, __shfl_xor(bad, o, GL));
    }
    if (!vox_ok || !(zmax >= d0 - dstep)) continue;
    int w_lo = 0, w_hi = P.fW - 1, h_lo = 0, h_hi = P.fH - 1, i_lo = 0;
    const int i_hi = min(S - 1, (int) ceilf((zmax - d0) / dstep + 0.02f));
    if (bad == 0.f) {
      i_lo = max(0, (int) floorf((zmin - d0) / dstep - 0.02f));
      w_lo = max(0, (int) floorf((umin - us[0]) / du - 0.05f));
      w_hi = min(P.fW - 1, (int) ceilf((umax - us[0]) / du + 0.05f));
      h_lo = max(0, (int) floorf((vmin - vs[0]) / dv - 0.05f));
      h_hi = min(P.fH - 1, (int) ceilf((vmax - vs[0]) / dv + 0.05f));
    }
    if (i_lo > i_hi || w_lo > w_hi || h_lo > h_hi) continue;
    const int nw = w_hi - w_lo + 1, nh = h_hi - h_lo + 1;
    const int count = nw * nh * (i_hi - i_lo + 1);
    const long sbase = bn * S * HW;
    const float inv_nw = 1.0f / (float) nw, inv_nh = 1.0f / (float) nh;
    // U candidates per lane per round: all coordinate loads of a round are issued before any
    // is tested, so a round costs one memory latency instead of U.
    constexpr int U = 4;
    for (int base = l; base < count; base += U * GL) {
      long sidx[U];
      int pw[U], ph[U], key[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int idx = base + q * GL;
        // idx -> (i, h, w) without integer division (count < 2^20, quotients exact)
        int r = (int) (((float) idx + 0.5f) * inv_nw);
        int w = idx - r * nw;
        if (w < 0) { w += nw; --r; } else if (w >= nw) { w -= nw; ++r; }
        int i = (int) (((float) r + 0.5f) * inv_nh);
        int h = r - i * nh;
        if (h < 0) { h += nh; --i; } else if (h >= nh) { h -= nh; ++i; }
        pw[q] = w_lo + w; ph[q] = h_lo + h;
        sidx[q] = sbase + ((long) (i_lo + i) * P.fH + ph[q]) * P.fW + pw[q];
        const bool in = idx < count;
        key[q] = in ? KEY[sidx[q]] : 0;
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        // key 0 = masked sample / past the end; a sample touches this voxel iff each floor tap
        // is the voxel index or one below it
        const int kx = (key[q] & 2047) - 1, ky = ((key[q] >> 11) & 2047) - 1, kz = (key[q] >> 22) - 1;
        if (key[q] == 0 || (unsigned) (ix - kx) > 1u || (unsigned) (iy - ky) > 1u ||
            (unsigned) (iz - kz) > 1u) continue;
        const float wt = tap_weight(FX[sidx[q]], fix) * tap_weight(FY[sidx[q]], fiy) *
                         tap_weight(FZ[sidx[q]], fiz);
        const float Wv = wt * Wbuf[sidx[q]];
        acc[0] = __builtin_fmaf(wt, G0buf[sidx[q]], acc[0]);
        const float4* g4 =
            reinterpret_cast<const float4*>(Gcl + (bn * HW + (long) ph[q] * P.fW + pw[q]) * CP);
#pragma unroll
        for (int c4 = 0; c4 < CP4; ++c4) {
          const float4 f = g4[c4];
          if (c4 > 0) acc[c4 * 4] = __builtin_fmaf(Wv, f.x, acc[c4 * 4]);
          acc[c4 * 4 + 1] = __builtin_fmaf(Wv, f.y, acc[c4 * 4 + 1]);
          acc[c4 * 4 + 2] = __builtin_fmaf(Wv, f.z, acc[c4 * 4 + 2]);
          acc[c4 * 4 + 3] = __builtin_fmaf(Wv, f.w, acc[c4 * 4 + 3]);
        }
      }
    }
  }
  // reduce over the GL lanes of the voxel, transpose through LDS, store x-runs
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    float v = acc[c];
#pragma unroll
    for (int o = GL >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, GL);
    if (l == 0) outs[c][g] = v;
  }
  __syncthreads();
  const long V = (long) P.Z * P.Y * P.X;
  const long vox0 = ((long) iz * P.Y + iy) * P.X + (long) blockIdx.x * VPB;
  for (int e = tid; e < nch * VPB; e += 256) {
    const int c = e / VPB, gx = e % VPB;
    if (blockIdx.x * VPB + gx >= P.X) continue;
    const float v = outs[c][gx];
    if (c == 0) gdens[(long) b * V + vox0 + gx] = v;
    else if (c <= P.K) gsem[((long) b * P.K + (c - 1)) * V + vox0 + gx] = v;
    else grgb[((long) b * 3 + (c - 1 - P.K)) * V + vox0 + gx] = v;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
size_t cam_bwd_bin_bytes(const VampRenderDesc* d);     // render_bwd_bin.hip
size_t cam_bwd_cell_bytes(const VampRenderDesc* d);    // render_bwd_cell.hip
int launch_cam_bwd_cell(const VampRenderDesc* d, const RenderParams& P, const float* FX,
                        const float* FY, const float* FZ, const int* KEY, const float* Wbuf,
                        const float* G0buf, const float* Gcl, float* gdens, float* gsem,
                        float* grgb, void* scratch, int accumulate, hipEvent_t wait_event,
                        hipStream_t s);

size_t cam_bwd_v2_bytes(const VampRenderDesc* d) {
  const RenderParams P = to_params(d);
  const size_t samples = (size_t) d->B * d->N * (d->D - 1) * d->fH * d->fW;
  const size_t rays = (size_t) d->B * d->N * d->fH * d->fW;
  return 6 * align_up(samples * sizeof(float), 256) + align_up(rays * P.CP * sizeof(float), 256) +
         align_up((size_t) d->B * d->N * 48 * sizeof(float), 256) +
         align_up((size_t) d->B * d->Z * d->Y * ((d->X + VPB - 1) / VPB) * sizeof(unsigned), 256) +
         std::max(cam_bwd_bin_bytes(d), cam_bwd_cell_bytes(d));
}

size_t cam_bwd_bin_bytes(const VampRenderDesc* d);
int launch_cam_bwd_bin(const VampRenderDesc* d, const RenderParams& P, const float* FX,
                       const float* FY, const float* FZ, const float* Wbuf, const float* G0buf,
                       const float* Gcl, float* gdens, float* gsem, float* grgb, void* scratch,
                       const int** total_out, int* cap_out, hipStream_t s);

// scratch = workspace region after the packed volume
int launch_cam_bwd_v2(const VampRenderDesc* d, const RenderParams& P, const float* mats,
                      const float* us, const float* vs, const float* ds, const float* mids,
                      const float* beta, const float* packed, const float* g_rgb,
                      const float* g_seg, const float* g_depth, float* gdens, float* gsem,
                      float* grgb, float* grad_beta, void* scratch, int accumulate,
                      hipEvent_t wait_event, hipStream_t s) {
  const size_t samples = (size_t) d->B * d->N * (d->D - 1) * d->fH * d->fW;
  const size_t rays = (size_t) d->B * d->N * d->fH * d->fW;
  char* p = static_cast<char*>(scratch);
  float* Wbuf = reinterpret_cast<float*>(p); p += align_up(samples * sizeof(float), 256);
  float* G0buf = reinterpret_cast<float*>(p); p += align_up(samples * sizeof(float), 256);
  float* FX = reinterpret_cast<float*>(p); p += align_up(samples * sizeof(float), 256);
  float* FY = reinterpret_cast<float*>(p); p += align_up(samples * sizeof(float), 256);
  float* FZ = reinterpret_cast<float*>(p); p += align_up(samples * sizeof(float), 256);
  int* KEY = reinterpret_cast<int*>(p); p += align_up(samples * sizeof(float), 256);
  float* Gcl = reinterpret_cast<float*>(p); p += align_up(rays * P.CP * sizeof(float), 256);
  float* pmats = reinterpret_cast<float*>(p); p += align_up((size_t) d->B * d->N * 48 * sizeof(float), 256);
  unsigned* cull = reinterpret_cast<unsigned*>(p);
  p += align_up((size_t) d->B * d->Z * d->Y * ((d->X + VPB - 1) / VPB) * sizeof(unsigned), 256);
  void* bin_scratch = p;

  // scatter stage: cell list (default, render_bwd_cell.hip); VAMP_CAM_BWD=gather selects the
  // candidate-box gather below, VAMP_CAM_BWD=bin the brick lists of render_bwd_bin.hip
  const char* force = getenv("VAMP_CAM_BWD");
  const bool use_cell = !(force && (force[0] == 'g' || force[0] == 'b'));
  if ((accumulate || wait_event) && !use_cell)
    return fail(VAMP_EINVAL, "%s: accumulate / wait_event need the cell-list path", __func__);

  constexpr int LPR = 4;
  const int S = d->D - 1;
  const int L = (S + LPR - 1) / LPR;
  const size_t lds = (size_t) 3 * L * 256 * sizeof(float);
  if (lds > 150 * 1024) return fail(VAMP_EINVAL, "%s: too many depth samples for the LDS staging", __func__);
  const unsigned grid = ray_grid<LPR>(P);
  // 1. per-ray pass: one record per sample
#define VAMP_RAY(CP4)                                                                             \
  do {                                                                                            \
    auto kr = cam_bwd_ray_kernel<LPR, CP4>;                                                       \
    if (lds > 64 * 1024 &&                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kr),                                    \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) \
      return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);                           \
    VAMP_TIMED(kProfCamBwd, s, (kr<<<grid, 256, lds, s>>>(P, mats, us, vs, ds, mids, beta, packed, \
                                                           g_rgb, g_seg, g_depth, Wbuf, G0buf, FX, FY, FZ, \
                                                           KEY, Gcl, grad_beta, L)));                  \
  } while (0)
  if (P.CP == 12) VAMP_RAY(3); else if (P.CP == 24) VAMP_RAY(6); else VAMP_RAY(8);
#undef VAMP_RAY
  if (int e = check_launch("cam_bwd_ray_kernel")) return e;
  if (use_cell)
    return launch_cam_bwd_cell(d, P, FX, FY, FZ, KEY, Wbuf, G0buf, Gcl, gdens, gsem, grgb,
                               bin_scratch, accumulate, wait_event, s);

  // 2. candidate-box gather, or the records binned into per-brick lists with one workgroup
  //    owning each brick (render_bwd_bin.hip), in which case the gather runs only if the lists
  //    overflow the workspace (device-side decision).
  const int* total = nullptr;
  int cap = 0;
  const bool gather_only = force[0] != 'b';
  if (!gather_only)
    if (int e = launch_cam_bwd_bin(d, P, FX, FY, FZ, Wbuf, G0buf, Gcl, gdens, gsem, grgb,
                                   bin_scratch, &total, &cap, s))
      return e;
  if (int e = launch_invert_mats(mats, pmats, d->B * d->N * 3, false, s)) return e;
  dim3 bgrid((d->X + VPB - 1) / VPB, d->Y, d->Z * d->B);
  {
    const int runs_x = (d->X + VPB - 1) / VPB;
    const long nrun = (long) runs_x * d->Y * d->Z * d->B;
    VAMP_TIMED(kProfAux, s, (cam_bwd_cull_kernel<<<(unsigned) ((nrun + 255) / 256), 256, 0, s>>>(
        P, pmats, us, vs, ds, cull, runs_x, total, cap)));
    if (int e = check_launch("cam_bwd_cull_kernel")) return e;
  }
#define VAMP_GATHER(CP4)                                                                          \
  VAMP_TIMED(kProfCamBwdBrick, s, (cam_bwd_gather_kernel<CP4><<<bgrid, 256, 0, s>>>(              \
      P, pmats, us, vs, ds, FX, FY, FZ, KEY, Wbuf, G0buf, Gcl, cull, gdens, gsem, grgb, total, cap)))
  if (P.CP == 12) VAMP_GATHER(3); else if (P.CP == 24) VAMP_GATHER(6); else VAMP_GATHER(8);
#undef VAMP_GATHER
  return check_launch("cam_bwd_gather_kernel");
}

}  // namespace vamp
