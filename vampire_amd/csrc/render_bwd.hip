// RENDER backward kernels for gfx950 (autograd of base_vampire2.py:391-467).
//
// Notation per ray (camera branch) / per column (BEV branch), samples i = 0..S-1:
//   tau_i = sigma(s_i[0]) * delta_i,  T_i = exp(-sum_{j<i} tau_j),  w_i = (1 - exp(-tau_i)) T_i
//   out_c = sum_i w_i s_i[c]          (+ the depth / height expectation)
// With upstream gradients G the scalar q_i = sum_c G_c s_i[c] + G_depth (mid_i - d_far) gives
//   dL/ds_i[c]  = w_i G_c                                  (c >= 1)
//   dL/dtau_i   = q_i T_{i+1} - sum_{j>i} w_j q_j
//   dL/ds_i[0]  = dL/dtau_i * delta_i * sigma'(s_i[0])
// and each dL/ds_i[c] is splatted onto the sample's 8 trilinear taps.
//
// v1: one thread per ray / column, two marches (totals, then splat), float atomics.
#include "render_common.hpp"

namespace vamp {

int launch_pack(const RenderParams& P, int in_dtype, const void* dens, const void* sem,
                const void* rgb, float* packed, hipStream_t s);
int launch_cam_bwd_v2(const VampRenderDesc* d, const RenderParams& P, const float* mats,
                      const float* us, const float* vs, const float* ds, const float* mids,
                      const float* beta, const void* dens, const void* sem, const void* rgbv, const float* g_rgb,
                      const float* g_seg, const float* g_depth, float* gdens, float* gsem,
                      float* grgb, float* grad_beta, void* scratch, int accumulate,
                      hipEvent_t wait_event, int cells_valid, const float* samples, const int* term,
                      int parts, hipStream_t s);
int launch_cam_prepare(const VampRenderDesc* d, const RenderParams& P, const float* mats,
                       const float* us, const float* vs, const float* ds, void* scratch,
                       const int* term, int phase, hipStream_t s, bool counters_clean = false,
                       const ScanJob* also = nullptr);

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  // wave reduce then 4-wave LDS reduce; result valid in thread 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wv] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int) (blockDim.x >> 6); ++i) r += red[i];
  return r;
}

// ---------------------------------------------------------------------------
// camera branch backward, v1
// ---------------------------------------------------------------------------
template <int CP4>
__global__ void __launch_bounds__(256)
render_cam_bwd_kernel(RenderParams P, const float* __restrict__ geom, const float* __restrict__ mats,
                      const float* __restrict__ us, const float* __restrict__ vs,
                      const float* __restrict__ ds, const float* __restrict__ mids,
                      const float* __restrict__ beta_raw, const float* __restrict__ packed,
                      const float* __restrict__ g_rgb, const float* __restrict__ g_seg,
                      const float* __restrict__ g_depth, float* __restrict__ gpacked,
                      float* __restrict__ grad_beta) {
  constexpr int CP = CP4 * 4;
  __shared__ float red[4];
  const long nrays = (long) P.B * P.N * P.fH * P.fW;
  long ray = (long) blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = ray < nrays;
  if (!live) ray = nrays - 1;
  const int w = ray % P.fW;
  const int h = (ray / P.fW) % P.fH;
  const long bn = ray / ((long) P.fW * P.fH);
  const int b = bn / P.N;
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const int S = P.D - 1;
  const float* m = mats ? mats + bn * 48 : nullptr;
  const float u = us[w], v = vs[h];
  const long V = (long) P.Z * P.Y * P.X;
  const float* vol = packed + (long) b * V * CP;
  float* gvol = gpacked + (long) b * V * CP;
  const long pstride = (long) P.fH * P.fW * 3;
  const float* gp = geom ? geom + ((bn * P.D * P.fH + h) * P.fW + w) * 3 : nullptr;
  const long HW = (long) P.fH * P.fW;
  const long pix = (long) h * P.fW + w;

  // upstream gradients of this ray, in packed channel order (G[0] unused)
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

  auto point = [&](int i, float& x, float& y, float& z) {
    if (gp) {
      const float* q = gp + (long) i * pstride;
      x = q[0]; y = q[1]; z = q[2];
    } else {
      frustum_point(m, u, v, ds[i], x, y, z);
      x = nan_to_num_geom(x); y = nan_to_num_geom(y); z = nan_to_num_geom(z);
    }
  };
  // gathers the masked sample; returns q-part sum_c G_c s[c] and raw s0 / finiteness
  auto sample = [&](const VolTap& tp, float& s0, bool& s0_finite) -> float {
    float s[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) s[c] = 0.f;
    if (tp.inside) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = tp.iz0 + (k >> 2), iy = tp.iy0 + ((k >> 1) & 1), ix = tp.ix0 + (k & 1);
        if (iz >= P.Z || iy >= P.Y || ix >= P.X) continue;
        const float wt = ((k & 1) ? tp.wx1 : tp.wx0) * ((k & 2) ? tp.wy1 : tp.wy0) *
                         ((k & 4) ? tp.wz1 : tp.wz0);
        const float4* f4 = reinterpret_cast<const float4*>(
            vol + (((long) iz * P.Y + iy) * P.X + ix) * CP);
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
    s0_finite = (s[0] == s[0]) && (fabsf(s[0]) <= 3.402823466e+38f);
    float qv = 0.f;
#pragma unroll
    for (int c = 1; c < CP; ++c) qv = __builtin_fmaf(G[c], nan_to_num(s[c]), qv);
    s0 = nan_to_num(s[0]);
    return qv;
  };

  // ---- march 1: total = sum_i w_i q_i ----
  float px, py, pz, qx, qy, qz;
  float total = 0.f, cum = 0.f;
  point(0, px, py, pz);
  for (int i = 0; i < S; ++i) {
    point(i + 1, qx, qy, qz);
    const VolTap tp = volume_tap(P, px, py, pz);
    float s0; bool fin;
    const float qv = sample(tp, s0, fin) + Gd * (mids[i] - P.d_far);
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float tau = density_fwd(dp, s0) * sqrtf(dx * dx + dy * dy + dz * dz);
    const float wgt = (1.0f - expf(-tau)) * expf(-cum);
    cum += tau;
    total = __builtin_fmaf(wgt, qv, total);
    px = qx; py = qy; pz = qz;
  }

  // ---- march 2: splat ----
  float prefix = 0.f, dbeta = 0.f;
  cum = 0.f;
  point(0, px, py, pz);
  for (int i = 0; i < S; ++i) {
    point(i + 1, qx, qy, qz);
    const VolTap tp = volume_tap(P, px, py, pz);
    float s0; bool fin;
    const float qv = sample(tp, s0, fin) + Gd * (mids[i] - P.d_far);
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float delta = sqrtf(dx * dx + dy * dy + dz * dz);
    const float tau = density_fwd(dp, s0) * delta;
    const float Tn = expf(-(cum + tau));                 // T_{i+1}
    const float wgt = (1.0f - expf(-tau)) * expf(-cum);
    cum += tau;
    prefix = __builtin_fmaf(wgt, qv, prefix);
    const float dtau = qv * Tn - (total - prefix);
    float dsig_ds, dsig_db;
    density_bwd(dp, s0, dsig_ds, dsig_db);
    dbeta = __builtin_fmaf(dtau * delta, dsig_db, dbeta);
    if (live && tp.inside) {
      const float g0 = fin ? dtau * delta * dsig_ds : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = tp.iz0 + (k >> 2), iy = tp.iy0 + ((k >> 1) & 1), ix = tp.ix0 + (k & 1);
        if (iz >= P.Z || iy >= P.Y || ix >= P.X) continue;
        const float wt = ((k & 1) ? tp.wx1 : tp.wx0) * ((k & 2) ? tp.wy1 : tp.wy0) *
                         ((k & 4) ? tp.wz1 : tp.wz0);
        float* dst = gvol + (((long) iz * P.Y + iy) * P.X + ix) * CP;
        atomicAdd(dst, wt * g0);
        const float ww = wt * wgt;
#pragma unroll
        for (int c = 1; c < CP; ++c)
          if (c <= P.K + 3) atomicAdd(dst + c, ww * G[c]);
      }
    }
    px = qx; py = qy; pz = qz;
  }
  if (P.density_mode == VAMP_DENSITY_SDF_LAPLACE) {
    const float tot = block_sum_256(live ? dbeta : 0.f, red);
    if (threadIdx.x == 0) {
      const float sgn = (beta_raw[0] > 0.f) ? 1.f : ((beta_raw[0] < 0.f) ? -1.f : 0.f);
      atomicAdd(grad_beta, sgn * tot);
    }
  }
}

// packed gradient [B,Z,Y,X,CP] -> channel-first grads (overwrite)
template <int CP4>
__global__ void __launch_bounds__(256)
unpack_grad_kernel(RenderParams P, const float* __restrict__ gpacked, float* __restrict__ gdens,
                   float* __restrict__ gsem, float* __restrict__ grgb) {
  const long V = (long) P.Z * P.Y * P.X;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * P.B) return;
  const long b = gid / V, vox = gid % V;
  const float4* src = reinterpret_cast<const float4*>(gpacked + gid * (CP4 * 4));
  float v[CP4 * 4];
#pragma unroll
  for (int q = 0; q < CP4; ++q) {
    const float4 f = src[q];
    v[q * 4] = f.x; v[q * 4 + 1] = f.y; v[q * 4 + 2] = f.z; v[q * 4 + 3] = f.w;
  }
#pragma unroll
  for (int c = 0; c < CP4 * 4; ++c) {
    if (c == 0) gdens[b * V + vox] = v[0];
    else if (c <= P.K) gsem[(b * P.K + (c - 1)) * V + vox] = v[c];
    else if (c <= P.K + 3) grgb[(b * 3 + (c - 1 - P.K)) * V + vox] = v[c];
  }
}

// ---------------------------------------------------------------------------
// BEV branch backward, v1: thread per column, lanes along x (contiguous atomics)
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float sample_cf_b(const RenderParams& P, const T* __restrict__ vol,
                                             long chan_base, const VolTap& t) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int iz = t.iz0 + (k >> 2), iy = t.iy0 + ((k >> 1) & 1), ix = t.ix0 + (k & 1);
    if (iz < 0 || iz >= P.Z || iy < 0 || iy >= P.Y || ix < 0 || ix >= P.X) continue;
    const float wt = ((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0) *
                     ((k & 4) ? t.wz1 : t.wz0);
    s = __builtin_fmaf(wt, ldf(vol, chan_base + ((long) iz * P.Y + iy) * P.X + ix), s);
  }
  return s;
}

__device__ __forceinline__ void splat_cf(const RenderParams& P, float* __restrict__ gvol,
                                         long chan_base, const VolTap& t, float g) {
  if (g == 0.f) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int iz = t.iz0 + (k >> 2), iy = t.iy0 + ((k >> 1) & 1), ix = t.ix0 + (k & 1);
    if (iz < 0 || iz >= P.Z || iy < 0 || iy >= P.Y || ix < 0 || ix >= P.X) continue;
    const float wt = ((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0) *
                     ((k & 4) ? t.wz1 : t.wz0);
    atomicAdd(gvol + chan_base + ((long) iz * P.Y + iy) * P.X + ix, wt * g);
  }
}

template <typename T>
__global__ void __launch_bounds__(64)
render_bev_bwd_kernel(RenderParams P, const float* __restrict__ oxs, const float* __restrict__ oys,
                      const float* __restrict__ ozs, const float* __restrict__ bev_mids,
                      const float* __restrict__ beta_raw, const T* __restrict__ dens,
                      const T* __restrict__ sem, const T* __restrict__ rgb,
                      const float* __restrict__ g_brgb, const float* __restrict__ g_bseg,
                      const float* __restrict__ g_bh, const float* __restrict__ g_vd,
                      const float* __restrict__ g_vo, float* __restrict__ gdens,
                      float* __restrict__ gsem, float* __restrict__ grgb,
                      float* __restrict__ gbase, float* __restrict__ grad_beta) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  const bool live = x < P.oX;
  const int xc = live ? x : P.oX - 1;
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const long V = (long) P.Z * P.Y * P.X;
  const long OYX = (long) P.oY * P.oX;
  const long col = (long) y * P.oX + xc;
  const int CO = P.C + (P.cat_seg ? P.K : 0);
  const float ox = oxs[xc], oy = oys[y];

  const float Gh = g_bh ? g_bh[(long) b * OYX + col] : 0.f;

  // q_j = sum_k Gseg_k s_j[k] + sum_c Grgb_c s_j[c] + Gh * mid_j
  auto q_of = [&](const VolTap& t, int j) -> float {
    float qv = Gh * bev_mids[j];
    if (g_bseg)
      for (int k = 0; k < P.K; ++k)
        qv = __builtin_fmaf(g_bseg[((long) b * P.K + k) * OYX + col],
                            sample_cf_b(P, sem, ((long) b * P.K + k) * V, t), qv);
    if (g_brgb)
      for (int c = 0; c < 3; ++c)
        qv = __builtin_fmaf(g_brgb[((long) b * 3 + c) * OYX + col],
                            sample_cf_b(P, rgb, ((long) b * 3 + c) * V, t), qv);
    return qv;
  };

  float total = 0.f, cum = 0.f;
  for (int j = 0; j < P.oZ; ++j) {
    const VolTap t = volume_tap(P, ox, oy, ozs[P.oZ - 1 - j]);
    const float tau = density_fwd(dp, sample_cf_b(P, dens, (long) b * V, t)) * (1.0f * P.z_step);
    const float wgt = (1.0f - expf(-tau)) * expf(-cum);
    cum += tau;
    total = __builtin_fmaf(wgt, q_of(t, j), total);
  }
  float prefix = 0.f, dbeta = 0.f;
  cum = 0.f;
  for (int j = 0; j < P.oZ; ++j) {
    const VolTap t = volume_tap(P, ox, oy, ozs[P.oZ - 1 - j]);
    const float s0 = sample_cf_b(P, dens, (long) b * V, t);
    const float tau = density_fwd(dp, s0) * (1.0f * P.z_step);
    const float Tn = expf(-(cum + tau));
    const float wgt = (1.0f - expf(-tau)) * expf(-cum);
    cum += tau;
    const float qv = q_of(t, j);
    prefix = __builtin_fmaf(wgt, qv, prefix);
    const float dtau = qv * Tn - (total - prefix);
    float dsig_ds, dsig_db;
    density_bwd(dp, s0, dsig_ds, dsig_db);
    // sigma feeds tau (compositing) and the voxel_density output directly
    const float gvd = g_vd ? g_vd[((long) b * P.oZ + j) * OYX + col] : 0.f;
    const float dsigma = dtau * (1.0f * P.z_step) + gvd;
    dbeta = __builtin_fmaf(dsigma, dsig_db, dbeta);
    if (!live) continue;
    splat_cf(P, gdens, (long) b * V, t, dsigma * dsig_ds);
    for (int k = 0; k < P.K; ++k) {
      float gk = g_bseg ? wgt * g_bseg[((long) b * P.K + k) * OYX + col] : 0.f;
      if (P.cat_seg && g_vo) gk += g_vo[(((long) b * CO + P.C + k) * P.oZ + j) * OYX + col];
      splat_cf(P, gsem, ((long) b * P.K + k) * V, t, gk);
    }
    if (g_brgb)
      for (int c = 0; c < 3; ++c)
        splat_cf(P, grgb, ((long) b * 3 + c) * V, t, wgt * g_brgb[((long) b * 3 + c) * OYX + col]);
    if (g_vo)
      for (int c = 0; c < P.C; ++c)
        splat_cf(P, gbase, ((long) b * P.C + c) * V, t,
                 g_vo[(((long) b * CO + c) * P.oZ + j) * OYX + col]);
  }
  if (P.density_mode == VAMP_DENSITY_SDF_LAPLACE) {
    float tot = live ? dbeta : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
    if (threadIdx.x == 0) {
      const float sgn = (beta_raw[0] > 0.f) ? 1.f : ((beta_raw[0] < 0.f) ? -1.f : 0.f);
      atomicAdd(grad_beta, sgn * tot);
    }
  }
}

}  // namespace vamp

using namespace vamp;

extern "C" {

int vamp_render_camera_prepare(const VampRenderDesc* d, const float* mats, const float* us,
                               const float* vs, const float* ds, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return vamp_render_camera_prepare_ex(d, mats, us, vs, ds, workspace, workspace_bytes, 0, stream);
}

static int camera_prepare(const VampRenderDesc* d, const float* mats, const float* us, const float* vs,
                          const float* ds, void* workspace, size_t workspace_bytes, int flags, const ScanJob* also,
                          void* stream) {
  if (int e = validate(d)) return e;
  const size_t need = vamp_render_workspace_bytes(d);
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  // the ranks are drawn (vamp_render_forward_merged with VAMP_RENDERFWD_RANK): the cell scan only
  if (flags & VAMP_CAMPREP_RANKED) return launch_cam_prepare_ranked(d, workspace, static_cast<hipStream_t>(stream), also);
  VAMP_REQUIRE(mats && us && vs && ds, "null pointer");
  const RenderParams P = to_params(d);
  return launch_cam_prepare(d, P, mats, us, vs, ds, static_cast<char*>(workspace) + packed_bytes(d),
                            (flags & VAMP_CAMPREP_TERM_VALID) ? cam_term_ptr(d, workspace) : nullptr,
                            /*phase=*/0, static_cast<hipStream_t>(stream),
                            (flags & VAMP_CAMPREP_COUNTERS_CLEAN) != 0, also);
}

int vamp_render_camera_prepare_ex(const VampRenderDesc* d, const float* mats, const float* us,
                                  const float* vs, const float* ds, void* workspace,
                                  size_t workspace_bytes, int flags, void* stream) {
  return camera_prepare(d, mats, us, vs, ds, workspace, workspace_bytes, flags, nullptr, stream);
}

// lift.hip
int lift_scan_job(const VampLiftDesc* d, void* workspace, size_t workspace_bytes, ScanJob* job);

int vamp_render_camera_prepare_with_lift(const VampRenderDesc* d, const float* mats, const float* us,
                                         const float* vs, const float* ds, void* workspace, size_t workspace_bytes,
                                         int flags, const VampLiftDesc* lift_desc, void* lift_workspace,
                                         size_t lift_workspace_bytes, void* stream) {
  VAMP_REQUIRE(lift_desc && lift_workspace, "null pointer");
  ScanJob lift;
  if (int e = lift_scan_job(lift_desc, lift_workspace, lift_workspace_bytes, &lift)) return e;
  return camera_prepare(d, mats, us, vs, ds, workspace, workspace_bytes, flags, &lift, stream);
}

int vamp_render_camera_backward(const VampRenderDesc* d, const float* geom, const float* mats,
                                const float* us, const float* vs, const float* ds,
                                const float* mids, const float* beta, const void* density_feature,
                                const void* semantic, const void* rgb, const float* g_rgb,
                                const float* g_seg, const float* g_depth,
                                float* grad_density_feature, float* grad_semantic, float* grad_rgb,
                                float* grad_beta, void* workspace, size_t workspace_bytes,
                                void* stream) {
  return vamp_render_camera_backward_acc(d, geom, mats, us, vs, ds, mids, beta, density_feature,
                                         semantic, rgb, g_rgb, g_seg, g_depth, grad_density_feature,
                                         grad_semantic, grad_rgb, grad_beta, workspace,
                                         workspace_bytes, 0, nullptr, stream);
}

int vamp_render_camera_backward_acc(const VampRenderDesc* d, const float* geom, const float* mats,
                                    const float* us, const float* vs, const float* ds,
                                    const float* mids, const float* beta,
                                    const void* density_feature, const void* semantic,
                                    const void* rgb, const float* g_rgb, const float* g_seg,
                                    const float* g_depth, float* grad_density_feature,
                                    float* grad_semantic, float* grad_rgb, float* grad_beta,
                                    void* workspace, size_t workspace_bytes, int flags,
                                    void* wait_event, void* stream) {
  const int accumulate = (flags & VAMP_CAMBWD_ACCUMULATE) ? 1 : 0;
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(geom || (mats && us && vs && ds), "need geom or (mats, us, vs, ds)");
  VAMP_REQUIRE(mids && density_feature && semantic && rgb, "null input");
  VAMP_REQUIRE(grad_density_feature && grad_semantic && grad_rgb, "null output");
  VAMP_REQUIRE((beta && grad_beta) || d->density_mode == VAMP_DENSITY_SIGMOID, "beta / grad_beta is NULL");
  const size_t pb = packed_bytes(d);
  const size_t need = vamp_render_workspace_bytes(d);
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  const RenderParams P = to_params(d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* packed = static_cast<float*>(workspace);
  float* gpacked = reinterpret_cast<float*>(static_cast<char*>(workspace) + pb);
  const bool part_ray = !(flags & (VAMP_CAMBWD_PART_GATHER | VAMP_CAMBWD_PART_HEAVY)) || (flags & VAMP_CAMBWD_PART_RAY);
  // (the default path reads the volumes as they are; only the v1 splat below gathers from a channel-last
  // copy, which the forward may have left at the head of the workspace)
  const bool splat = geom || !mats || (flags & VAMP_CAMBWD_SPLAT);
  if (splat && !(flags & VAMP_CAMBWD_PACKED_VALID) && part_ray)
    if (int e = launch_pack(P, d->in_dtype, density_feature, semantic, rgb, packed, s)) return e;
  // Default: per-ray pass + cell-list gather (render_bwd_ray.hip, render_bwd_cell.hip), which
  // evaluates the frustum points itself from the matrices.  A caller-supplied geom tensor, or
  // VAMP_CAMBWD_SPLAT, takes the v1 float-atomic splat below (the independent cross-check).
  int parts = ((flags & VAMP_CAMBWD_PART_RAY) ? kCamPartRay : 0) | ((flags & VAMP_CAMBWD_PART_GATHER) ? kCamPartGather : 0) |
              ((flags & VAMP_CAMBWD_PART_HEAVY) ? kCamPartHeavy : 0);
  if (parts == 0) parts = kCamPartAll;
  if (!geom && mats && !(flags & VAMP_CAMBWD_SPLAT)) {
    const float* samples = nullptr;
    if (flags & VAMP_CAMBWD_SAMPLES_VALID) {
      if (workspace_bytes < need + vamp_render_samples_bytes(d))
        return fail(VAMP_ENOSPC, "%s: workspace %ld has no room for the sample rows", __func__, (long) workspace_bytes);
      samples = reinterpret_cast<const float*>(static_cast<char*>(workspace) + need);
    }
    // early ray termination: the table of the forward (TERM_VALID), or computed here; the cell
    // lists must have been prepared with the same table
    int* term = nullptr;
    if (!(flags & VAMP_CAMBWD_NO_ERT)) {
      term = cam_term_ptr(d, workspace);
      if (!(flags & VAMP_CAMBWD_TERM_VALID) && part_ray) {
        VAMP_REQUIRE(!(flags & VAMP_CAMBWD_CELLS_VALID), "CELLS_VALID with early termination needs TERM_VALID");
        if (int e = launch_cam_term(d, P, mats, us, vs, ds, beta, density_feature, term, s)) return e;
      } else if (part_ray) {
        if (int e = debug_expect_range(term, (size_t) d->B * d->N * d->fH * d->fW, 0, d->D - 1, s,
                                       "VAMP_CAMBWD_TERM_VALID: the workspace holds a termination table")) return e;
      }
    }
    return launch_cam_bwd_v2(d, P, mats, us, vs, ds, mids, beta, density_feature, semantic, rgb, g_rgb, g_seg, g_depth,
                             grad_density_feature, grad_semantic, grad_rgb, grad_beta, gpacked,
                             accumulate, static_cast<hipEvent_t>(wait_event),
                             (flags & VAMP_CAMBWD_CELLS_VALID) ? 1 : 0,
                             samples, term, parts, s);
  }
  VAMP_REQUIRE(!accumulate && !wait_event, "accumulate / wait_event need the cell-list path");
  if (int ze = launch_zero(gpacked, pb, s)) return ze;
  const long nrays = (long) d->B * d->N * d->fH * d->fW;
  const unsigned grid = (unsigned) ((nrays + 255) / 256);
  const long nvox = (long) d->B * d->Z * d->Y * d->X;
  const unsigned ugrid = (unsigned) ((nvox + 255) / 256);
#define VAMP_CAMB(CP4)                                                                           \
  do {                                                                                           \
    VAMP_TIMED(kProfCamBwdV1, s, (render_cam_bwd_kernel<CP4><<<grid, 256, 0, s>>>(                \
        P, geom, mats, us, vs, ds, mids, beta, packed, g_rgb, g_seg, g_depth, gpacked, grad_beta))); \
    if (int e = check_launch("render_cam_bwd_kernel")) return e;                                 \
    VAMP_TIMED(kProfUnpack, s, (unpack_grad_kernel<CP4><<<ugrid, 256, 0, s>>>(                   \
        P, gpacked, grad_density_feature, grad_semantic, grad_rgb)));                            \
  } while (0)
  if (P.CP == 12) VAMP_CAMB(3); else if (P.CP == 24) VAMP_CAMB(6); else VAMP_CAMB(8);
#undef VAMP_CAMB
  return check_launch("unpack_grad_kernel");
}

}  // extern "C"

namespace vamp {
// v1 BEV backward (column threads + float atomics); kept as an independent cross-check
int launch_bev_bwd_v1(const VampRenderDesc* d, const float* oxs, const float* oys,
                      const float* ozs, const float* bev_mids, const float* beta,
                      const void* density_feature, const void* semantic, const void* rgb,
                      const void* base, const float* g_bev_rgb, const float* g_bev_seg,
                      const float* g_bev_height, const float* g_voxel_density,
                      const float* g_voxel_output, float* grad_density_feature,
                      float* grad_semantic, float* grad_rgb, float* grad_base, float* grad_beta,
                      void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(d->oZ > 0 && d->oY > 0 && d->oX > 0, "det grid must be non-empty");
  VAMP_REQUIRE(oxs && oys && ozs && bev_mids && density_feature && semantic && rgb, "null input");
  VAMP_REQUIRE(grad_density_feature && grad_semantic && grad_rgb, "null output");
  VAMP_REQUIRE(grad_base || d->C == 0 || !g_voxel_output, "grad_base is NULL");
  VAMP_REQUIRE((beta && grad_beta) || d->density_mode == VAMP_DENSITY_SIGMOID, "beta / grad_beta is NULL");
  const RenderParams P = to_params(d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((d->oX + 63) / 64, d->oY, d->B);
  if (d->in_dtype == VAMP_F32)
    VAMP_TIMED(kProfBevBwd, s, (render_bev_bwd_kernel<float><<<grid, 64, 0, s>>>(
        P, oxs, oys, ozs, bev_mids, beta, (const float*) density_feature, (const float*) semantic,
        (const float*) rgb, g_bev_rgb, g_bev_seg, g_bev_height, g_voxel_density, g_voxel_output,
        grad_density_feature, grad_semantic, grad_rgb, grad_base, grad_beta)));
  else
    VAMP_TIMED(kProfBevBwd, s, (render_bev_bwd_kernel<__hip_bfloat16><<<grid, 64, 0, s>>>(
        P, oxs, oys, ozs, bev_mids, beta, (const __hip_bfloat16*) density_feature,
        (const __hip_bfloat16*) semantic, (const __hip_bfloat16*) rgb, g_bev_rgb, g_bev_seg,
        g_bev_height, g_voxel_density, g_voxel_output, grad_density_feature, grad_semantic,
        grad_rgb, grad_base, grad_beta)));
  return check_launch("render_bev_bwd_kernel");
}
}  // namespace vamp
