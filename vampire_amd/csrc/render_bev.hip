// BEV (top-down) branch of the renderer for gfx950: base_vampire2.py:408-418, 442-461.
//
// The det-grid sample lattice is regular, so every access pattern here is coalesced
// (lanes along x) and the work splits per channel:
//
//  forward   bev_density   thread per column: density samples -> sigma_j (= voxel_density)
//                          and the height expectation
//            bev_channels  thread per (channel, column): trilinear samples of one channel at
//                          the oZ heights; composite (sem / rgb) with weights rebuilt from
//                          voxel_density, or pass through (base -> voxel_output)
//  backward  bev_q         thread per (channel-lane, column): q_j = sum_c G_c s_j[c]
//            bev_scan      thread per column: weights, dL/dtau_j, dL/ds_j[0], d beta
//            bev_gather    thread per (voxel column, 4 channels): walks the lattice heights
//                          upwards, one (y, x)-weighted plane sum per height, the two voxel
//                          planes it feeds kept in registers -- no atomics; overwrites (the
//                          camera gather then adds) or adds onto what the buffers hold
#include "render_common.hpp"
#include "pair_gather.hpp"

namespace vamp {

int launch_bev_bwd_v1(const VampRenderDesc* d, const float* oxs, const float* oys,
                      const float* ozs, const float* bev_mids, const float* beta,
                      const void* density_feature, const void* semantic, const void* rgb,
                      const void* base, const float* g_bev_rgb, const float* g_bev_seg,
                      const float* g_bev_height, const float* g_voxel_density,
                      const float* g_voxel_output, float* grad_density_feature,
                      float* grad_semantic, float* grad_rgb, float* grad_base, float* grad_beta,
                      void* stream);

// x/y part of a column's taps (shared by all heights) and the z part per height
struct AxisTap {
  int i0;
  float w0, w1;
};

__device__ __forceinline__ AxisTap axis_tap(float pos, float lo, float span, int n) {
  const float g = ((pos - lo) / span) * 2.0f - 1.0f;
  const float f = ((g + 1.0f) / 2.0f) * (float) (n - 1);
  const float fl = floorf(f);
  AxisTap t;
  t.i0 = (int) fl;
  t.w1 = f - fl;
  t.w0 = (fl + 1.0f) - f;
  return t;
}

template <typename T>
__device__ __forceinline__ float sample8(const RenderParams& P, const T* __restrict__ vol, long cb,
                                         const AxisTap& tx, const AxisTap& ty, const AxisTap& tz) {
  // aten tap order: x fastest, then y, then z; zero padding outside the volume
  // branch-free: out-of-volume taps are clamped to a legal address and given zero weight, so
  // the eight loads are independent (a bounds branch per tap serialises the round trips)
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int iz = tz.i0 + (k >> 2), iy = ty.i0 + ((k >> 1) & 1), ix = tx.i0 + (k & 1);
    const bool in = iz >= 0 && iz < P.Z && iy >= 0 && iy < P.Y && ix >= 0 && ix < P.X;
    const float wt = in ? ((k & 1) ? tx.w1 : tx.w0) * ((k & 2) ? ty.w1 : ty.w0) * ((k & 4) ? tz.w1 : tz.w0) : 0.f;
    const long at = ((long) min(max(iz, 0), P.Z - 1) * P.Y + min(max(iy, 0), P.Y - 1)) * P.X + min(max(ix, 0), P.X - 1);
    s = __builtin_fmaf(wt, ldf(vol, cb + at), s);
  }
  return s;
}

// bilinear (x, y) sample of one volume plane, zero padding (also for a plane outside the volume)
template <typename T>
__device__ __forceinline__ float bilinear_plane(const RenderParams& P, const T* __restrict__ vol,
                                                long cb, const AxisTap& tx, const AxisTap& ty, int iz) {
  const bool zin = iz >= 0 && iz < P.Z;
  const int izc = min(max(iz, 0), P.Z - 1);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = ty.i0 + (k >> 1), ix = tx.i0 + (k & 1);
    const bool in = zin && iy >= 0 && iy < P.Y && ix >= 0 && ix < P.X;
    const float wt = in ? ((k & 1) ? tx.w1 : tx.w0) * ((k & 2) ? ty.w1 : ty.w0) : 0.f;
    s = __builtin_fmaf(wt, ldf(vol, cb + ((long) izc * P.Y + min(max(iy, 0), P.Y - 1)) * P.X + min(max(ix, 0), P.X - 1)), s);
  }
  return s;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// Heights are taken kHChunk at a time: a column is one thread and the grid is only oY * oX / 64
// waves, so the loads of one height cannot hide behind other waves -- the taps of a whole chunk
// are issued together instead of one round trip per height.
constexpr int kHChunk = 5;

template <typename T>
__global__ void __launch_bounds__(256)
bev_density_kernel(RenderParams P, const float* __restrict__ oxs, const float* __restrict__ oys,
                   const float* __restrict__ ozs, const float* __restrict__ bev_mids,
                   const float* __restrict__ beta_raw, const T* __restrict__ dens,
                   float* __restrict__ voxel_density, float* __restrict__ bev_height,
                   float* __restrict__ s0_save) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x >= P.oX || y >= P.oY) return;
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const long V = (long) P.Z * P.Y * P.X, OYX = (long) P.oY * P.oX, col = (long) y * P.oX + x;
  const AxisTap tx = axis_tap(oxs[x], P.lo[0], P.span[0], P.X);
  const AxisTap ty = axis_tap(oys[y], P.lo[1], P.span[1], P.Y);
  float cum = 0.f, height = 0.f;
  for (int j0 = 0; j0 < P.oZ; j0 += kHChunk) {
    float s0[kHChunk];
#pragma unroll
    for (int u = 0; u < kHChunk; ++u) {
      const int j = min(j0 + u, P.oZ - 1);
      const AxisTap tz = axis_tap(ozs[P.oZ - 1 - j], P.lo[2], P.span[2], P.Z);     // flip (bv2:443)
      s0[u] = sample8(P, dens, (long) b * V, tx, ty, tz);
    }
#pragma unroll
    for (int u = 0; u < kHChunk; ++u) {
      const int j = j0 + u;
      if (j >= P.oZ) break;
      const float sigma = density_fwd(dp, s0[u]);
      voxel_density[((long) b * P.oZ + j) * OYX + col] = sigma;
      if (s0_save) s0_save[((long) b * P.oZ + j) * OYX + col] = s0[u];           // for the backward's scan
      const float tau = sigma * (1.0f * P.z_step);                                  // bv2:451-453
      height = __builtin_fmaf((1.0f - expf(-tau)) * expf(-cum), bev_mids[j], height);
      cum += tau;
    }
  }
  bev_height[(long) b * OYX + col] = height;
}

// channel index space of bev_channels: [0, K) semantic, [K, K+3) rgb, [K+3, K+3+C) base.
// A thread owns one BEV column and NC consecutive channels: the column's compositing weights
// (two exps per height) and the height taps are worked out once and shared by its channels --
// with a thread per (channel, column) this kernel spent 80 % of its time in the vector ALU
// redoing them 38 times -- and the 8 * NC plane loads of a height go out together.
// cfg-B, us per launch: one thread per (channel, column) 55; NC = 1 / 2 / 4 / 8: 46 / 41.5 / 50 / 48
// (fewer waves per CU hide less latency past NC = 2).
#ifndef VAMP_BEV_NC
#define VAMP_BEV_NC 2
#endif
constexpr int kBevNC = VAMP_BEV_NC;   // channels per thread
constexpr int kBevMaxOZ = 64;         // heights whose taps fit the LDS table

template <typename T, int NC>
__global__ void __launch_bounds__(256)
bev_channels_kernel(RenderParams P, const float* __restrict__ oxs, const float* __restrict__ oys,
                    const float* __restrict__ ozs, const T* __restrict__ sem,
                    const T* __restrict__ rgb, const T* __restrict__ base,
                    const float* __restrict__ voxel_density, float* __restrict__ bev_rgb,
                    float* __restrict__ bev_seg, float* __restrict__ voxel_output,
                    float* __restrict__ ss_save) {
  __shared__ int tz_i0[kBevMaxOZ];
  __shared__ float tz_w0[kBevMaxOZ], tz_w1[kBevMaxOZ];
  if ((int) threadIdx.x < P.oZ) {
    const AxisTap tz = axis_tap(ozs[P.oZ - 1 - threadIdx.x], P.lo[2], P.span[2], P.Z);   // flip (bv2:443)
    tz_i0[threadIdx.x] = tz.i0; tz_w0[threadIdx.x] = tz.w0; tz_w1[threadIdx.x] = tz.w1;
  }
  __syncthreads();
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int nch = P.K + 3 + P.C;
  const int ngrp = (nch + NC - 1) / NC;
  const int b = blockIdx.z / ngrp, c0 = (blockIdx.z % ngrp) * NC;
  if (x >= P.oX || y >= P.oY) return;
  const long V = (long) P.Z * P.Y * P.X, OYX = (long) P.oY * P.oX, col = (long) y * P.oX + x;
  const int CO = P.C + (P.cat_seg ? P.K : 0);
  const AxisTap tx = axis_tap(oxs[x], P.lo[0], P.span[0], P.X);
  const AxisTap ty = axis_tap(oys[y], P.lo[1], P.span[1], P.Y);
  // the four (y, x) taps of the column: clamped offsets and weights (zero outside the volume)
  long off4[4];
  float w4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = ty.i0 + (k >> 1), ix = tx.i0 + (k & 1);
    const bool in = iy >= 0 && iy < P.Y && ix >= 0 && ix < P.X;
    w4[k] = in ? ((k & 1) ? tx.w1 : tx.w0) * ((k & 2) ? ty.w1 : ty.w0) : 0.f;
    off4[k] = (long) min(max(iy, 0), P.Y - 1) * P.X + min(max(ix, 0), P.X - 1);
  }
  const T* vol[NC];
  long cb[NC];
  bool on[NC];
#pragma unroll
  for (int u = 0; u < NC; ++u) {
    const int ch = min(c0 + u, nch - 1);
    on[u] = c0 + u < nch;
    if (ch < P.K) { vol[u] = sem; cb[u] = ((long) b * P.K + ch) * V; }
    else if (ch < P.K + 3) { vol[u] = rgb; cb[u] = ((long) b * 3 + (ch - P.K)) * V; }
    else { vol[u] = base; cb[u] = ((long) b * P.C + (ch - P.K - 3)) * V; }
  }
  auto plane = [&](int u, int iz) -> float {
    const bool zin = iz >= 0 && iz < P.Z;
    const long zo = cb[u] + (long) min(max(iz, 0), P.Z - 1) * P.Y * P.X;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s = __builtin_fmaf(zin ? w4[k] : 0.f, ldf(vol[u], zo + off4[k]), s);
    return s;
  };
  float cum = 0.f, acc[NC];
#pragma unroll
  for (int u = 0; u < NC; ++u) acc[u] = 0.f;
  const bool any_comp = c0 < P.K + 3;
  // Heights kHChunk at a time.  A wave's life is (heights) x (memory round trip) -- 2.3 us per
  // height under load -- so the volume planes of a whole chunk are fetched in one go: the chunk's
  // heights step down the volume by at most one plane each (det and seg grids have about the same
  // spacing), so its kHChunk + 1 planes below the first height's upper plane cover it; a height
  // outside that window (other spacings) fetches its two planes itself.
  for (int j0 = 0; j0 < P.oZ; j0 += kHChunk) {
    const int top = tz_i0[j0] + 1;
    float pl[NC][kHChunk + 1], tau[kHChunk];
#pragma unroll
    for (int t = 0; t <= kHChunk; ++t)
#pragma unroll
      for (int u = 0; u < NC; ++u) pl[u][t] = plane(u, top - t);
#pragma unroll
    for (int h = 0; h < kHChunk; ++h)
      tau[h] = any_comp ? voxel_density[((long) b * P.oZ + min(j0 + h, P.oZ - 1)) * OYX + col] * (1.0f * P.z_step) : 0.f;   // bv2:451-458
#pragma unroll
    for (int h = 0; h < kHChunk; ++h) {
      const int j = j0 + h;
      if (j >= P.oZ) break;
      const int i0 = tz_i0[j];
      const float wz0 = tz_w0[j], wz1 = tz_w1[j];
      const int d = top - 1 - i0;                  // planes below the chunk's first height (uniform)
      float n_lo[NC], n_hi[NC];
      bool found = false;
#pragma unroll
      for (int t = 0; t < kHChunk; ++t)
        if (d == t) {
#pragma unroll
          for (int u = 0; u < NC; ++u) { n_hi[u] = pl[u][t]; n_lo[u] = pl[u][t + 1]; }
          found = true;
        }
      if (!found) {
#pragma unroll
        for (int u = 0; u < NC; ++u) { n_hi[u] = plane(u, i0 + 1); n_lo[u] = plane(u, i0); }
      }
      const float wj = (1.0f - expf(-tau[h])) * expf(-cum);
      cum += tau[h];
#pragma unroll
      for (int u = 0; u < NC; ++u) {
        const float sv = __builtin_fmaf(wz1, n_hi[u], wz0 * n_lo[u]);
        const int ch = c0 + u;
        if (!on[u]) continue;
        if (ch < P.K + 3) {
          acc[u] = __builtin_fmaf(wj, sv, acc[u]);
          // training: the backward's q_j = sum_c G_c s_j[c] reads the samples back
          if (ss_save) ss_save[(((long) b * (P.K + 3) + ch) * P.oZ + j) * OYX + col] = sv;
          if (ch < P.K && P.cat_seg)
            voxel_output[(((long) b * CO + P.C + ch) * P.oZ + j) * OYX + col] = sv;     // bv2:449-450
        } else {
          voxel_output[(((long) b * CO + (ch - P.K - 3)) * P.oZ + j) * OYX + col] = sv;
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NC; ++u) {
    const int ch = c0 + u;
    if (!on[u]) continue;
    if (ch < P.K) bev_seg[((long) b * P.K + ch) * OYX + col] = acc[u];
    else if (ch < P.K + 3) bev_rgb[((long) b * 3 + (ch - P.K)) * OYX + col] = acc[u];
  }
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
// Q[b][j][col] = sum over sem / rgb channels of G_c * s_j[c]; 4 channel-lanes per column
// q_j = sum_c G_c s_j[c] over the composited channels (semantic, rgb).  A workgroup owns 64
// columns of one row and ALL heights: thread (column, channel lane cl) samples channels
// cl, cl + 4, ... and walks the heights keeping each channel's two bilinear plane values
// (consecutive heights share a volume plane: 4 new taps per height and channel instead of 8), so
// a volume plane is read once.  The per-height form this replaces (one workgroup per height, 8
// taps per sample, planes re-fetched by every height's workgroups) moved 152 MB of HBM traffic
// for 37 MB of volumes.  Partial sums of the four channel lanes meet in LDS at the end.
constexpr int kQCh = 6;               // channels per thread (4 lanes x 6 >= K + 3 = 21)

template <typename T>
__global__ void __launch_bounds__(256)
bev_q_kernel(RenderParams P, const float* __restrict__ oxs, const float* __restrict__ oys,
             const float* __restrict__ ozs, const T* __restrict__ sem, const T* __restrict__ rgb,
             const float* __restrict__ g_brgb, const float* __restrict__ g_bseg,
             float* __restrict__ Q) {
  extern __shared__ float qred[];               // [oZ][3][64]
  __shared__ int tz_i0[kBevMaxOZ];
  __shared__ float tz_w0[kBevMaxOZ], tz_w1[kBevMaxOZ];
  if ((int) threadIdx.x < P.oZ) {
    const AxisTap tz = axis_tap(ozs[P.oZ - 1 - threadIdx.x], P.lo[2], P.span[2], P.Z);   // flip (bv2:443)
    tz_i0[threadIdx.x] = tz.i0; tz_w0[threadIdx.x] = tz.w0; tz_w1[threadIdx.x] = tz.w1;
  }
  __syncthreads();
  const int lx = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + lx, y = blockIdx.y, b = blockIdx.z;
  const bool live = x < P.oX;
  const int xc = live ? x : P.oX - 1;
  const long V = (long) P.Z * P.Y * P.X, OYX = (long) P.oY * P.oX, col = (long) y * P.oX + xc;
  const AxisTap tx = axis_tap(oxs[xc], P.lo[0], P.span[0], P.X);
  const AxisTap ty = axis_tap(oys[y], P.lo[1], P.span[1], P.Y);
  const int nch = P.K + 3;
  // the four (y, x) taps of the column: clamped offsets and weights (zero outside the volume)
  long off4[4];
  float w4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = ty.i0 + (k >> 1), ix = tx.i0 + (k & 1);
    const bool in = iy >= 0 && iy < P.Y && ix >= 0 && ix < P.X;
    w4[k] = in ? ((k & 1) ? tx.w1 : tx.w0) * ((k & 2) ? ty.w1 : ty.w0) : 0.f;
    off4[k] = (long) min(max(iy, 0), P.Y - 1) * P.X + min(max(ix, 0), P.X - 1);
  }
  for (int c0 = cl; c0 < nch; c0 += 4 * kQCh) {           // one pass for K + 3 <= 24
    const T* vol[kQCh];
    long cb[kQCh];
    float gc[kQCh], v_lo[kQCh], v_hi[kQCh];
#pragma unroll
    for (int i = 0; i < kQCh; ++i) {
      const int c = c0 + 4 * i;
      const int ch = min(c, nch - 1);
      const bool is_sem = ch < P.K;
      const float* gp = is_sem ? g_bseg : g_brgb;
      const long gi = is_sem ? ((long) b * P.K + ch) * OYX + col : ((long) b * 3 + (ch - P.K)) * OYX + col;
      gc[i] = (gp && c < nch) ? gp[gi] : 0.f;
      vol[i] = is_sem ? sem : rgb;
      cb[i] = is_sem ? ((long) b * P.K + ch) * V : ((long) b * 3 + (ch - P.K)) * V;
      v_lo[i] = v_hi[i] = 0.f;
    }
    int p_lo = -0x7fffffff, p_hi = -0x7fffffff;
    for (int j = 0; j < P.oZ; ++j) {
      const int i0 = tz_i0[j];
      const float wz0 = tz_w0[j], wz1 = tz_w1[j];
      const bool lo_is_lo = i0 == p_lo, lo_is_hi = i0 == p_hi, hi_is_lo = i0 + 1 == p_lo, hi_is_hi = i0 + 1 == p_hi;
      const bool z0in = i0 >= 0 && i0 < P.Z, z1in = i0 + 1 >= 0 && i0 + 1 < P.Z;
      const long zo0 = (long) min(max(i0, 0), P.Z - 1) * P.Y * P.X;
      const long zo1 = (long) min(max(i0 + 1, 0), P.Z - 1) * P.Y * P.X;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < kQCh; ++i) {
        float n_lo, n_hi;
        if (lo_is_lo) n_lo = v_lo[i];
        else if (lo_is_hi) n_lo = v_hi[i];
        else {
          n_lo = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) n_lo = __builtin_fmaf(z0in ? w4[k] : 0.f, ldf(vol[i], cb[i] + zo0 + off4[k]), n_lo);
        }
        if (hi_is_lo) n_hi = v_lo[i];
        else if (hi_is_hi) n_hi = v_hi[i];
        else {
          n_hi = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) n_hi = __builtin_fmaf(z1in ? w4[k] : 0.f, ldf(vol[i], cb[i] + zo1 + off4[k]), n_hi);
        }
        v_lo[i] = n_lo; v_hi[i] = n_hi;
        q = __builtin_fmaf(gc[i], __builtin_fmaf(wz1, n_hi, wz0 * n_lo), q);
      }
      p_lo = i0; p_hi = i0 + 1;
      if (c0 == cl) {
        if (cl > 0) qred[(j * 3 + cl - 1) * 64 + lx] = q;
        else qred[(P.oZ * 3 + j) * 64 + lx] = q;
      } else {
        if (cl > 0) qred[(j * 3 + cl - 1) * 64 + lx] += q;
        else qred[(P.oZ * 3 + j) * 64 + lx] += q;
      }
    }
  }
  __syncthreads();
  // four waves share the final sums and stores: wave cl takes heights cl, cl + 4, ...
  for (int j = cl; j < P.oZ; j += 4)
    if (live)
      Q[((long) b * P.oZ + j) * OYX + col] = (qred[(P.oZ * 3 + j) * 64 + lx] + qred[(j * 3) * 64 + lx]) +
                                             (qred[(j * 3 + 1) * 64 + lx] + qred[(j * 3 + 2) * 64 + lx]);
}
// bev_q_saved + bev_scan as ONE kernel (round 4; the forward kept its samples): lanes = (column, height).
// A workgroup is 64 consecutive columns of the flattened (y, x) lattice times the heights, a wave per height
// (beyond kQsMaxWaves heights a wave takes several).  Per (column, height): q = G . s and tau with ALL of
// the height's loads in flight together (the column gradients, the kept samples, the density sample, the
// voxel_density gradient: one round trip); then the per-column scan WITHOUT a serial pass -- the first cut
// had wave 0 walk the heights (three expf per height, a dependent chain of ~5 us with every other wave at
// the barrier): each (column, height) forms its own prefix of the taus in the column kernel's order, its own
// weight and transmittance, and after one more barrier its own prefix of sum_k w_k q_k (the same fma chain:
// same bits as the thread-per-column scan); no Q round trip, one launch.
constexpr int kQsMaxWaves = 16;
template <int NCH>                               // K + 3 (exact), or 0: any channel count, 7 channels in flight
__global__ void __launch_bounds__(kQsMaxWaves * 64)
bev_qscan_saved_kernel(RenderParams P, const float* __restrict__ bev_mids, const float* __restrict__ beta_raw,
                       const float* __restrict__ s0_saved, const float* __restrict__ ss,
                       const float* __restrict__ g_brgb, const float* __restrict__ g_bseg,
                       const float* __restrict__ g_bh, const float* __restrict__ g_vd,
                       float* __restrict__ Wb, float* __restrict__ DS0, float* __restrict__ beta_part) {
  extern __shared__ float qs[];                 // [oZ][64] tau | q | w | T
  __shared__ float red[kQsMaxWaves];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), nwv = blockDim.x >> 6;
  const int b = blockIdx.y;
  const long OYX = (long) P.oY * P.oX;
  const long col_raw = (long) blockIdx.x * 64 + lane;
  const bool live = col_raw < OYX;
  const long col = live ? col_raw : OYX - 1;
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const int nch = NCH ? NCH : P.K + 3;
  const float dz = 1.0f * P.z_step;
  float* l_tau = qs;
  float* l_q = qs + P.oZ * 64;
  float* l_w = l_q + P.oZ * 64;
  float* l_T = l_w + P.oZ * 64;
  const float* gsem = g_bseg ? g_bseg + (long) b * P.K * OYX + col : nullptr;
  const float* grgb = g_brgb ? g_brgb + (long) b * 3 * OYX + col : nullptr;
  // the wave's first height keeps its density sample and voxel_density gradient in registers
  float s0_first = 0.f, gvd_first = 0.f;
  for (int j = wv; j < P.oZ; j += nwv) {
    const float* sp = ss + ((long) b * nch * P.oZ + j) * OYX + col;                 // channel stride: oZ * OYX
    const float Gh = g_bh ? g_bh[(long) b * OYX + col] : 0.f;
    const float s0 = s0_saved[((long) b * P.oZ + j) * OYX + col];
    const float gvd = g_vd ? g_vd[((long) b * P.oZ + j) * OYX + col] : 0.f;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    constexpr int U = 7;                          // (the accumulation pattern of the first cut: K + 3 = 21 = 3 x 7)
    if (NCH) {
      // (buffer descriptors: the column is the one per-lane offset, channels and heights are scalar offsets --
      // 64-bit addresses for 42 loads in flight do not fit the register budget of a 16-wave workgroup.  A
      // missing gradient tensor is a zero-size descriptor: its loads return 0.)
      float g[NCH ? NCH : 1], v[NCH ? NCH : 1];
      const __amdgpu_buffer_rsrc_t rs_gs = make_rsrc(g_bseg, g_bseg ? (size_t) P.B * (NCH - 3) * OYX * 4 : 0);
      const __amdgpu_buffer_rsrc_t rs_gr = make_rsrc(g_brgb, g_brgb ? (size_t) P.B * 3 * OYX * 4 : 0);
      const __amdgpu_buffer_rsrc_t rs_ss = make_rsrc(ss, (size_t) P.B * NCH * P.oZ * OYX * 4);
      const unsigned vo = (unsigned) col * 4u, oyx4 = (unsigned) OYX * 4u;
      const unsigned sj = ((unsigned) b * NCH * P.oZ + (unsigned) j) * oyx4;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool is_sem = c < NCH - 3;
        g[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
            is_sem ? rs_gs : rs_gr, vo, ((unsigned) b * (is_sem ? NCH - 3 : 3) + (unsigned) (is_sem ? c : c - (NCH - 3))) * oyx4, 0));
        v[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_ss, vo, sj + (unsigned) c * P.oZ * oyx4, 0));
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int u = c % U;
        if (u % 3 == 0) q0 = __builtin_fmaf(g[c], v[c], q0);
        else if (u % 3 == 1) q1 = __builtin_fmaf(g[c], v[c], q1);
        else q2 = __builtin_fmaf(g[c], v[c], q2);
      }
    } else {
      for (int c0 = 0; c0 < nch; c0 += U) {
        float g[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ch = min(c0 + u, nch - 1);
          const bool is_sem = ch < P.K;
          const float* gp = is_sem ? gsem : grgb;
          g[u] = (gp && c0 + u < nch) ? gp[(long) (is_sem ? ch : ch - P.K) * OYX] : 0.f;
          v[u] = sp[(long) ch * P.oZ * OYX];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (u % 3 == 0) q0 = __builtin_fmaf(g[u], v[u], q0);
          else if (u % 3 == 1) q1 = __builtin_fmaf(g[u], v[u], q1);
          else q2 = __builtin_fmaf(g[u], v[u], q2);
        }
      }
    }
    if (j == wv) { s0_first = s0; gvd_first = gvd; }
    l_tau[j * 64 + lane] = density_fwd(dp, s0) * dz;
    l_q[j * 64 + lane] = ((q0 + q1) + q2) + Gh * bev_mids[j];
  }
  __syncthreads();
  // weight and transmittance of the own heights: the prefix of the taus in height order (bev_scan_kernel's
  // cum += tau chain: same bits)
  for (int j = wv; j < P.oZ; j += nwv) {
    float cum = 0.f;
    for (int k = 0; k < j; ++k) cum += l_tau[k * 64 + lane];
    const float tau = l_tau[j * 64 + lane];
    l_w[j * 64 + lane] = (1.0f - expf(-tau)) * expf(-cum);
    cum += tau;
    l_T[j * 64 + lane] = expf(-cum);
  }
  __syncthreads();
  float dbeta = 0.f;
  for (int j = wv; j < P.oZ; j += nwv) {
    // total = sum_k w_k q_k as the column kernel's fma chain, and its prefix up to the own height
    float total = 0.f, prefix = 0.f;
    for (int k = 0; k < P.oZ; ++k) {
      total = __builtin_fmaf(l_w[k * 64 + lane], l_q[k * 64 + lane], total);
      if (k == j) prefix = total;
    }
    const float dtau = l_q[j * 64 + lane] * l_T[j * 64 + lane] - (total - prefix);
    const float s0 = j == wv ? s0_first : s0_saved[((long) b * P.oZ + j) * OYX + col];
    float dsig_ds, dsig_db;
    density_bwd(dp, s0, dsig_ds, dsig_db);
    const float gvd = j == wv ? gvd_first : (g_vd ? g_vd[((long) b * P.oZ + j) * OYX + col] : 0.f);
    const float dsigma = dtau * dz + gvd;          // sigma feeds tau and voxel_density
    dbeta = __builtin_fmaf(dsigma, dsig_db, dbeta);
    if (live) {
      Wb[((long) b * P.oZ + j) * OYX + col] = l_w[j * 64 + lane];
      DS0[((long) b * P.oZ + j) * OYX + col] = dsigma * dsig_ds;
    }
  }
  if (P.density_mode == VAMP_DENSITY_SDF_LAPLACE) {
    float v = live ? dbeta : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wv] = v;
    __syncthreads();
    // one partial per workgroup, summed in a fixed order
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int k = 0; k < nwv; ++k) t += red[k];
      beta_part[blockIdx.x + gridDim.x * blockIdx.y] = t;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
bev_scan_kernel(RenderParams P, const float* __restrict__ oxs, const float* __restrict__ oys,
                const float* __restrict__ ozs, const float* __restrict__ bev_mids,
                const float* __restrict__ beta_raw, const T* __restrict__ dens,
                const float* __restrict__ s0_saved, const float* __restrict__ g_bh,
                const float* __restrict__ g_vd, const float* __restrict__ Q, float* __restrict__ Wb,
                float* __restrict__ DS0, float* __restrict__ beta_part) {
  __shared__ float red[4];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  const bool live = x < P.oX && y < P.oY;
  const int xc = min(x, P.oX - 1), yc = min(y, P.oY - 1);
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const long V = (long) P.Z * P.Y * P.X, OYX = (long) P.oY * P.oX, col = (long) yc * P.oX + xc;
  const AxisTap tx = axis_tap(oxs[xc], P.lo[0], P.span[0], P.X);
  const AxisTap ty = axis_tap(oys[yc], P.lo[1], P.span[1], P.Y);
  const float Gh = g_bh ? g_bh[(long) b * OYX + col] : 0.f;
  const float dz = 1.0f * P.z_step;
  // the density sample of height j: kept by the forward, or sampled again
  auto sample = [&](int j) -> float {
    if (s0_saved) return s0_saved[((long) b * P.oZ + j) * OYX + col];
    const AxisTap tz = axis_tap(ozs[P.oZ - 1 - j], P.lo[2], P.span[2], P.Z);
    return sample8(P, dens, (long) b * V, tx, ty, tz);
  };
  // heights kHChunk at a time (see bev_density): the loads of a chunk are in flight together
  float total = 0.f, cum = 0.f;
  for (int j0 = 0; j0 < P.oZ; j0 += kHChunk) {
    float s0[kHChunk], qv[kHChunk];
#pragma unroll
    for (int u = 0; u < kHChunk; ++u) {
      const int j = min(j0 + u, P.oZ - 1);
      s0[u] = sample(j);
      qv[u] = Q[((long) b * P.oZ + j) * OYX + col] + Gh * bev_mids[j];
    }
#pragma unroll
    for (int u = 0; u < kHChunk; ++u) {
      if (j0 + u >= P.oZ) break;
      const float tau = density_fwd(dp, s0[u]) * dz;
      total = __builtin_fmaf((1.0f - expf(-tau)) * expf(-cum), qv[u], total);
      cum += tau;
    }
  }
  float prefix = 0.f, dbeta = 0.f;
  cum = 0.f;
  for (int j0 = 0; j0 < P.oZ; j0 += kHChunk) {
    float s0[kHChunk], qv[kHChunk], gvd[kHChunk];
#pragma unroll
    for (int u = 0; u < kHChunk; ++u) {
      const int j = min(j0 + u, P.oZ - 1);
      s0[u] = sample(j);
      qv[u] = Q[((long) b * P.oZ + j) * OYX + col] + Gh * bev_mids[j];
      gvd[u] = g_vd ? g_vd[((long) b * P.oZ + j) * OYX + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kHChunk; ++u) {
      const int j = j0 + u;
      if (j >= P.oZ) break;
      const float tau = density_fwd(dp, s0[u]) * dz;
      const float wgt = (1.0f - expf(-tau)) * expf(-cum);
      const float Tn = expf(-(cum + tau));
      cum += tau;
      prefix = __builtin_fmaf(wgt, qv[u], prefix);
      const float dtau = qv[u] * Tn - (total - prefix);
      float dsig_ds, dsig_db;
      density_bwd(dp, s0[u], dsig_ds, dsig_db);
      const float dsigma = dtau * dz + gvd[u];       // sigma feeds tau and voxel_density
      dbeta = __builtin_fmaf(dsigma, dsig_db, dbeta);
      if (live) {
        Wb[((long) b * P.oZ + j) * OYX + col] = wgt;
        DS0[((long) b * P.oZ + j) * OYX + col] = dsigma * dsig_ds;
      }
    }
  }
  if (P.density_mode == VAMP_DENSITY_SDF_LAPLACE) {
    float v = live ? dbeta : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    // one partial per workgroup; launch_beta_reduce adds them up in a fixed order
    if (threadIdx.x == 0)
      beta_part[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// candidate lattice indices along one axis for voxel index iv: the lattice positions are
// axis[k], uniformly spaced; returns [k_lo, k_hi] (clamped), possibly empty
__device__ __forceinline__ void lattice_range(const float* __restrict__ axis, int n, float lo,
                                              float span, int nvox, int iv, int& k_lo, int& k_hi) {
  // voxel tap coordinate f = (p - lo) / span * (nvox - 1); |f - iv| < 1  <=>  p in (p0, p1)
  const float e = span / (float) (nvox - 1);
  const float p0 = lo + ((float) iv - 1.0f) * e, p1 = lo + ((float) iv + 1.0f) * e;
  if (n == 1) { k_lo = 0; k_hi = 0; return; }
  const float a0 = axis[0], step = (axis[n - 1] - a0) / (float) (n - 1);
  float t0 = (p0 - a0) / step, t1 = (p1 - a0) / step;
  if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
  k_lo = max(0, (int) floorf(t0 - 0.01f));
  k_hi = min(n - 1, (int) ceilf(t1 + 0.01f));
}

__device__ __forceinline__ float tap_w(const AxisTap& t, int iv) {
  return (t.i0 == iv) ? t.w0 : ((t.i0 + 1 == iv) ? t.w1 : 0.f);
}

// thread per voxel (z, y, x); the <= kMaxT lattice taps per axis and their weights are found
// once and reused for all 1+K+3+C channels.
constexpr int kMaxT = 3;

struct AxisHits {
  int k[kMaxT];
  float w[kMaxT];
  int n;
  bool overflow;     // more than kMaxT lattice points touch this voxel: use the generic path
};

__device__ __forceinline__ AxisHits axis_hits(const float* __restrict__ axis, int n, float lo,
                                              float span, int nvox, int iv) {
  AxisHits h;
  h.n = 0;
  h.overflow = false;
#pragma unroll
  for (int i = 0; i < kMaxT; ++i) { h.k[i] = 0; h.w[i] = 0.f; }
  int k_lo, k_hi;
  lattice_range(axis, n, lo, span, nvox, iv, k_lo, k_hi);
  for (int k = k_lo; k <= k_hi; ++k) {
    const float w = tap_w(axis_tap(axis[k], lo, span, nvox), iv);
    if (w != 0.f && h.n >= kMaxT) h.overflow = true;
    if (w != 0.f && h.n < kMaxT) {
#pragma unroll
      for (int i = 0; i < kMaxT; ++i)
        if (i == h.n) { h.k[i] = k; h.w[i] = w; }
      ++h.n;
    }
  }
  return h;
}

// any number of lattice points per voxel: weights recomputed inside the loops (slow; only
// launched when the det lattice is much finer than the volume)
__global__ void __launch_bounds__(256)
bev_gather_generic_kernel(RenderParams P, const float* __restrict__ oxs,
                          const float* __restrict__ oys, const float* __restrict__ ozs,
                          const float* __restrict__ g_brgb, const float* __restrict__ g_bseg,
                          const float* __restrict__ g_vo, const float* __restrict__ Wb,
                          const float* __restrict__ DS0, float* __restrict__ gdens,
                          float* __restrict__ gsem, float* __restrict__ grgb,
                          float* __restrict__ gbase, int z_lo, int z_hi) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int nz = z_hi - z_lo + 1;
  const int z = z_lo + blockIdx.z % nz;
  const int b = blockIdx.z / nz;
  if (x >= P.X || y >= P.Y) return;
  const long V = (long) P.Z * P.Y * P.X, OYX = (long) P.oY * P.oX;
  const int CO = P.C + (P.cat_seg ? P.K : 0);
  const long vox = ((long) z * P.Y + y) * P.X + x;
  int kx0, kx1, ky0, ky1, kz0, kz1;
  lattice_range(oxs, P.oX, P.lo[0], P.span[0], P.X, x, kx0, kx1);
  lattice_range(oys, P.oY, P.lo[1], P.span[1], P.Y, y, ky0, ky1);
  lattice_range(ozs, P.oZ, P.lo[2], P.span[2], P.Z, z, kz0, kz1);
  const int nch = 1 + P.K + 3 + P.C;
  for (int ch = 0; ch < nch; ++ch) {
    float sum = 0.f;
    for (int kz = kz0; kz <= kz1; ++kz) {
      const float wz = tap_w(axis_tap(ozs[kz], P.lo[2], P.span[2], P.Z), z);
      const int j = P.oZ - 1 - kz;
      for (int ky = ky0; ky <= ky1; ++ky) {
        const float wy = tap_w(axis_tap(oys[ky], P.lo[1], P.span[1], P.Y), y);
        for (int kx = kx0; kx <= kx1; ++kx) {
          const float wx = tap_w(axis_tap(oxs[kx], P.lo[0], P.span[0], P.X), x);
          const float wt = wz * wy * wx;
          if (wt == 0.f) continue;
          const long cc = (long) ky * P.oX + kx;
          const long sc = ((long) b * P.oZ + j) * OYX + cc;
          float dsv = 0.f;
          if (ch == 0) dsv = DS0[sc];
          else if (ch <= P.K) {
            if (g_bseg) dsv = Wb[sc] * g_bseg[((long) b * P.K + ch - 1) * OYX + cc];
            if (P.cat_seg && g_vo) dsv += g_vo[(((long) b * CO + P.C + ch - 1) * P.oZ + j) * OYX + cc];
          } else if (ch <= P.K + 3) {
            if (g_brgb) dsv = Wb[sc] * g_brgb[((long) b * 3 + ch - 1 - P.K) * OYX + cc];
          } else if (g_vo) {
            dsv = g_vo[(((long) b * CO + ch - 4 - P.K) * P.oZ + j) * OYX + cc];
          }
          sum = __builtin_fmaf(wt, dsv, sum);
        }
      }
    }
    float* out = (ch == 0) ? gdens + (long) b * V
                 : (ch <= P.K) ? gsem + ((long) b * P.K + ch - 1) * V
                 : (ch <= P.K + 3) ? grgb + ((long) b * 3 + ch - 1 - P.K) * V
                                   : gbase + ((long) b * P.C + ch - 4 - P.K) * V;
    out[vox] += sum;
  }
}

// Per-axis hit tables, built once per call: for every voxel index along an axis the (<= kMaxT)
// lattice samples whose taps include it, and their weights, as one 32-byte record
// {k0, k1, k2, count | w0, w1, w2, -} so that the gather needs a single round of loads per axis.
__global__ void __launch_bounds__(256)
bev_axis_table_kernel(RenderParams P, const float* __restrict__ oxs, const float* __restrict__ oys,
                      const float* __restrict__ ozs, int4* __restrict__ tab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int NX = P.X, NY = P.Y, NZ = P.Z;
  if (i >= NX + NY + NZ) return;
  AxisHits h;
  int base, n;
  if (i < NX) { h = axis_hits(oxs, P.oX, P.lo[0], P.span[0], P.X, i); base = 0; n = i; }
  else if (i < NX + NY) { h = axis_hits(oys, P.oY, P.lo[1], P.span[1], P.Y, i - NX); base = NX; n = i - NX; }
  else { h = axis_hits(ozs, P.oZ, P.lo[2], P.span[2], P.Z, i - NX - NY); base = NX + NY; n = i - NX - NY; }
  static_assert(kMaxT == 3, "one 32-byte record per axis index: k[3], n | w[3], pad");
  tab[2 * (base + n)] = make_int4(h.k[0], h.k[1], h.k[2], h.n);
  tab[2 * (base + n) + 1] = make_int4(__float_as_int(h.w[0]), __float_as_int(h.w[1]), __float_as_int(h.w[2]), 0);
}

__device__ __forceinline__ AxisHits load_axis_hits(const int4* __restrict__ tab, int i) {
  const int4 a = tab[2 * i], w = tab[2 * i + 1];
  AxisHits h;
  h.k[0] = a.x; h.k[1] = a.y; h.k[2] = a.z; h.n = a.w;
  h.w[0] = __int_as_float(w.x); h.w[1] = __int_as_float(w.y); h.w[2] = __int_as_float(w.z);
  h.overflow = false;
  return h;
}

__device__ __forceinline__ AxisHits uniform_hits(AxisHits h) {
#pragma unroll
  for (int i = 0; i < kMaxT; ++i) {
    h.k[i] = __builtin_amdgcn_readfirstlane(h.k[i]);
    h.w[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(h.w[i])));
  }
  h.n = __builtin_amdgcn_readfirstlane(h.n);
  return h;
}

// Column gather: a thread owns one voxel column (y, x) and G channels of one tensor and walks
// the lattice heights in ascending order.  What a height gives to the column is one
// (y, x)-weighted sum per channel -- the "plane sum" -- which goes to the two voxel planes of the
// height's z taps; both running sums live in registers and a voxel plane is stored once, when
// the heights have moved past it.  No per-voxel tap search, no second visit of a plane sum, and
// the loads of the next height are issued before the current one is consumed.
//   composited channels (semantic, rgb): dL/ds_j[c] = Wb_j * gcol[c]; gcol is one value per BEV
//              column, loaded once per thread with the (y, x) weights folded in
//   pass-through channels (base; semantic with cat_seg): dL/ds_j[c] = g_voxel_output[c, j]
//   density (rides with group 0 of the rgb launch): dL/ds_j[0] = DS0_j
// Taps per height: 2 (y) x 3 (x) slots without branches (the third x slot exists for about one
// voxel in a hundred and otherwise has weight 0); a third y hit adds its three slots behind a
// wave-level test.  Needs the heights ascending (the host checks) and oZ <= kBevMaxOZ.
// The per-voxel formulation this replaces (thread per voxel and 4 channels, 8 tap loads per
// output, read-modify-write of every output) took 26 + 55 + 54 us at cfg-B.
// OW: overwrite the outputs (planes the lattice does not touch get zeros) instead of adding.
#ifndef VAMP_COLG
#define VAMP_COLG 4
#endif
constexpr int kColG = VAMP_COLG;       // channels per thread, pass-through launch
#ifndef VAMP_COLGC
#define VAMP_COLGC 4
#endif
constexpr int kColGC = VAMP_COLGC;     // channels per thread, composited launch
#ifndef VAMP_COMPG
#define VAMP_COMPG 3
#endif
#ifndef VAMP_COMPHC
#define VAMP_COMPHC 10
#endif
constexpr int kCompG = VAMP_COMPG;     // channels per thread, composited launch (round 4: 21 = 7 x 3)
constexpr int kCompHC = VAMP_COMPHC;   // heights whose taps are loaded together

template <int G, bool COL, bool VO, bool OW>
__global__ void __launch_bounds__(256)
bev_gather_col_kernel(RenderParams P, const int4* __restrict__ tab, const float* __restrict__ ozs,
                      const float* __restrict__ gcol, const float* __restrict__ gcol2,
                      const float* __restrict__ g_vo, int vo_c0, const float* __restrict__ Wb,
                      const float* __restrict__ DS0, float* __restrict__ gdens,
                      float* __restrict__ gout, float* __restrict__ gout2, int nchan, int nchan2,
                      int zseg, int with_dens, int flags, BetaTail btail) {
  beta_tail(btail);                     // the scan's d beta partials (a launch of its own before round 3)
  __shared__ int tz_i0[kBevMaxOZ];
  __shared__ float tz_w0[kBevMaxOZ], tz_w1[kBevMaxOZ];
  if ((int) threadIdx.x < P.oZ) {
    const AxisTap tz = axis_tap(ozs[threadIdx.x], P.lo[2], P.span[2], P.Z);
    tz_i0[threadIdx.x] = tz.i0; tz_w0[threadIdx.x] = tz.w0; tz_w1[threadIdx.x] = tz.w1;
  }
  __syncthreads();
  const unsigned YX = (unsigned) (P.Y * P.X);
  const unsigned col = blockIdx.x * 256 + threadIdx.x;       // rows are contiguous: full waves
  // channel space: [0, nchan) of (gcol, gout), then [0, nchan2) of (gcol2, gout2); the density
  // channel rides with the last group
  const int ngrp = (nchan + nchan2 + G - 1) / G;
  const int nseg = (P.Z + zseg - 1) / zseg;
  const int cg = blockIdx.y % ngrp;
  const int sg = (blockIdx.y / ngrp) % nseg;
  const int b = blockIdx.y / (ngrp * nseg);
  if (col >= YX) return;
  const int x = col % (unsigned) P.X, y = col / (unsigned) P.X;
  constexpr bool ow = OW, ow_d = OW;
  const bool dens = with_dens && cg == ngrp - 1;
  const AxisHits hx = load_axis_hits(tab, x), hy = load_axis_hits(tab, P.X + y);
  const bool y3 = __any(hy.n > 2);
  const unsigned V = (unsigned) P.Z * YX, OYX = (unsigned) (P.oY * P.oX);
  const int CO = P.C + (P.cat_seg ? P.K : 0);

  // the (y, x) taps of the column: lattice offsets and weights (absent slots: k = 0, w = 0)
  unsigned cc[3][3];
  float wyx[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      cc[c][e] = (unsigned) hy.k[c] * P.oX + hx.k[e];
      wyx[c][e] = hy.w[c] * hx.w[e];
    }
  bool on[G];
  unsigned vb[G];
  float* op[G];                 // output plane 0 of the channel, at this column (wave-uniform base)
  const float* gp[G];           // column gradient of the channel (null: none)
  float gw[G][2][3];
#pragma unroll
  for (int u = 0; u < G; ++u) {
    // a slot past the last channel repeats the last channel (same value to the same address from
    // the same thread) when overwriting, which keeps the loop free of per-slot branches
    const int ch = OW ? min(cg * G + u, nchan + nchan2 - 1) : cg * G + u;
    on[u] = ch < nchan + nchan2;
    const bool second = ch >= nchan;
    const int chc = on[u] ? (second ? ch - nchan : ch) : 0;
    const int nc = second ? nchan2 : nchan;
    op[u] = (second ? gout2 : gout) + ((unsigned) b * nc + chc) * V + col;
    const float* gc = second ? gcol2 : gcol;
    gp[u] = (COL && on[u] && gc) ? gc + ((unsigned) b * nc + chc) * OYX : nullptr;
    vb[u] = ((unsigned) b * CO + (VO ? vo_c0 : 0) + chc) * P.oZ * OYX;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 3; ++e) gw[u][c][e] = gp[u] ? wyx[c][e] * gp[u][cc[c][e]] : 0.f;
  }
  const unsigned od = (unsigned) b * V + col;

  // this segment's voxel planes [za, zb) and the heights whose taps reach them (ascending)
  const int za = sg * zseg, zb = min(P.Z, za + zseg);
  int k_a = P.oZ, k_b = -1;
  for (int k = 0; k < P.oZ; ++k) {
    const int i0 = tz_i0[k];
    if (i0 + 1 >= za && i0 < zb) { k_a = min(k_a, k); k_b = max(k_b, k); }
  }

  auto store_plane = [&](int z, const float (&v)[G], float vd, bool touched) __attribute__((always_inline)) {
    if (z < za || z >= zb) return;
    const unsigned zo = (unsigned) z * YX;
    if (ow) {
#pragma unroll
      for (int u = 0; u < G; ++u) op[u][zo] = v[u];
    } else if (touched) {
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (on[u]) op[u][zo] += v[u];
    }
    if (dens) {
      if (ow_d) gdens[od + zo] = vd;
      else if (touched) gdens[od + zo] += vd;
    }
  };
  float zero[G];
#pragma unroll
  for (int u = 0; u < G; ++u) zero[u] = 0.f;
  if (k_b < k_a) {                               // nothing reaches this segment
    for (int z = za; z < zb; ++z) store_plane(z, zero, 0.f, false);
    return;
  }

  // raw taps of one height: [G][2][3] pass-through values, [2][3] Wb and DS0
  constexpr int GV = VO ? G : 1;
  float nv[GV][2][3], nw[2][3], nd[2][3];
  const float* __restrict__ dsp = dens ? DS0 : Wb;
  auto fetch = [&](int k) __attribute__((always_inline)) {
    const unsigned jv = (unsigned) (P.oZ - 1 - k) * OYX;                 // flip (bv2:443)
    const unsigned jo = (unsigned) b * P.oZ * OYX + jv;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        // (a load behind a run-time test would push the tap arrays out of registers: groups
        // without the density rider read Wb twice instead)
        if (COL) nw[c][e] = Wb[jo + cc[c][e]];
        if (COL) nd[c][e] = dsp[jo + cc[c][e]];
        if (VO) {
#pragma unroll
          for (int u = 0; u < G; ++u) nv[u][c][e] = g_vo[vb[u] + jv + cc[c][e]];
        }
      }
  };
  fetch(k_a);
  int cur = tz_i0[k_a];                           // voxel plane of lo[]
  float lo[G], hi[G], lod = 0.f, hid = 0.f;
#pragma unroll
  for (int u = 0; u < G; ++u) lo[u] = hi[u] = 0.f;
  bool lo_t = false, hi_t = false;                // does any height touch the plane?
  for (int z = za; z < min(cur, zb); ++z) store_plane(z, zero, 0.f, false);

  for (int k = k_a; k <= k_b; ++k) {
    // plane sums of height k from the fetched taps
    float S[G], Sd = 0.f;
#pragma unroll
    for (int u = 0; u < G; ++u) S[u] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        if (COL) Sd = __builtin_fmaf(wyx[c][e], nd[c][e], Sd);
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (COL) S[u] = __builtin_fmaf(nw[c][e], gw[u][c][e], S[u]);
          if (VO) S[u] = __builtin_fmaf(wyx[c][e], nv[u][c][e], S[u]);
        }
      }
    if (y3) {                                     // third y hit somewhere in the wave (about 1 row in 100)
      const unsigned jv = (unsigned) (P.oZ - 1 - k) * OYX;
      const unsigned jo = (unsigned) b * P.oZ * OYX + jv;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const float w = wyx[2][e];
        if (dens) Sd = __builtin_fmaf(w, DS0[jo + cc[2][e]], Sd);
        const float wbv = COL ? w * Wb[jo + cc[2][e]] : 0.f;
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (COL && gp[u]) S[u] = __builtin_fmaf(wbv, gp[u][cc[2][e]], S[u]);
          if (VO) S[u] = __builtin_fmaf(w, g_vo[vb[u] + jv + cc[2][e]], S[u]);
        }
      }
    }
    const int i0 = tz_i0[k];
    const float w0 = tz_w0[k], w1 = tz_w1[k];
    if (k < k_b) fetch(k + 1);                    // in flight while this height is consumed
    // the heights have moved past plane `cur`: store it, shift
    while (cur < i0) {
      store_plane(cur, lo, lod, lo_t);
#pragma unroll
      for (int u = 0; u < G; ++u) { lo[u] = hi[u]; hi[u] = 0.f; }
      lod = hid; hid = 0.f;
      lo_t = hi_t; hi_t = false;
      ++cur;
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      lo[u] = __builtin_fmaf(w0, S[u], lo[u]);
      hi[u] = __builtin_fmaf(w1, S[u], hi[u]);
    }
    lod = __builtin_fmaf(w0, Sd, lod);
    hid = __builtin_fmaf(w1, Sd, hid);
    lo_t = hi_t = true;
  }
  store_plane(cur, lo, lod, lo_t);
  store_plane(cur + 1, hi, hid, hi_t);
  for (int z = max(cur + 2, za); z < zb; ++z) store_plane(z, zero, 0.f, false);
}

// Composited channels (semantic, rgb) and the density of the column gather, round 4.
//
// bev_gather_col_kernel<G, true, false, OW> was a thread per (voxel column, 4 channels, z-segment) that issued
// the 12 tap loads of a height one height ahead of their use: a wave lived for (heights) x (one L2 round trip),
// 11.8 us where its arithmetic is ~1 us, every one of the 12 (channel group, segment) threads of a column
// loaded the same Wb taps again -- 1 250 wave-level loads per 64 columns -- and the 56 MB went out at 1.2 TB/s
// (the stores alone take 12 us: ablation).  What a height gives to a composited channel is
//      S_j[c] = sum over the (y, x) taps t of  Wb_j[t] * (wyx[t] * gcol_c[t]),
// and the taps do not depend on the channel.  Here a workgroup is 64 columns x one wave per channel group
// (G channels; the last wave is the density, whose taps are DS0's and whose per-tap factor is wyx alone -- the
// same products in the same order as before, bit for bit): the waves fetch the taps of a chunk of kCompHC
// heights TOGETHER, each its share, all in flight at once, into LDS (two sources x heights x 6 (9) taps x 64
// columns), and every wave then walks the heights from LDS.  ~220 wave-level loads per 64 columns, one round
// trip per chunk instead of one per height, 70 registers, and all workgroups of cfg-B resident at once.
//   grid: x = blocks of 64 columns (XCD-major, like the forward), y = B; block = 64 x min(groups, kCompMaxW)
#ifdef VAMP_COMP_STAMPS
// diagnostic build only (tools/debug/bev_gather_stamps.py): phase stamps of wave 0 of every workgroup
__device__ long long g_comp_stamps[4096 * 8];
#define VAMP_CSTAMP(k)                                                                                     \
  do {                                                                                                     \
    const unsigned wgi_ = blockIdx.y * gridDim.x + blockIdx.x;                                             \
    if (threadIdx.x == 0 && wgi_ < 4096) g_comp_stamps[wgi_ * 8 + (k)] = (long long) wall_clock64();       \
  } while (0)
extern "C" int vamp_debug_comp_stamps(long long* host, size_t n) {
  return (int) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_comp_stamps), n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#else
#define VAMP_CSTAMP(k) do { } while (0)
#endif
constexpr int kCompMaxW = 8;           // waves per workgroup (a wave loops over its groups beyond that)
template <int G, int HC, bool OW>
__global__ void __launch_bounds__(kCompMaxW * 64, 6)      // (3 workgroups per CU: cfg-B's 625 are one round)
bev_gather_comp_kernel(RenderParams P, const int4* __restrict__ tab, const float* __restrict__ ozs,
                       const float* __restrict__ gcol, const float* __restrict__ gcol2,
                       const float* __restrict__ Wb, const float* __restrict__ DS0, float* __restrict__ gdens,
                       float* __restrict__ gout, float* __restrict__ gout2, int nchan, int nchan2,
                       BetaTail btail) {
  VAMP_CSTAMP(0);
  beta_tail(btail);                     // the scan's d beta partials
  __shared__ int tz_i0[kBevMaxOZ];
  __shared__ float tz_w0[kBevMaxOZ], tz_w1[kBevMaxOZ];
  __shared__ float taps[2][HC][9][64];  // [Wb | DS0][height of the chunk][(y, x) tap; 6..8: a third y hit][column]
  if ((int) threadIdx.x < P.oZ) {
    const AxisTap tz = axis_tap(ozs[threadIdx.x], P.lo[2], P.span[2], P.Z);
    tz_i0[threadIdx.x] = tz.i0; tz_w0[threadIdx.x] = tz.w0; tz_w1[threadIdx.x] = tz.w1;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int nwv = __builtin_amdgcn_readfirstlane((int) (blockDim.x >> 6));
  const unsigned YX = (unsigned) (P.Y * P.X);
  const int nwg = (int) ((YX + 63) / 64), per_xcd = (nwg + 7) / 8;
  const int wg = (int) (blockIdx.x & 7) * per_xcd + (int) (blockIdx.x >> 3);
  const bool wg_live = wg < nwg && (int) (blockIdx.x >> 3) < per_xcd;          // (uniform: barriers below stay matched)
  const unsigned col_raw = (unsigned) min(wg, nwg - 1) * 64u + lane;
  const bool live = wg_live && col_raw < YX;
  const unsigned col = min(col_raw, YX - 1);
  const int b = blockIdx.y;
  const int ngrp = (nchan + nchan2 + G - 1) / G + 1;           // + the density's group
  const int x = col % (unsigned) P.X, y = col / (unsigned) P.X;
  const AxisHits hx = load_axis_hits(tab, x), hy = load_axis_hits(tab, P.X + y);
  const bool y3 = __any(hy.n > 2);
  const unsigned V = (unsigned) P.Z * YX, OYX = (unsigned) (P.oY * P.oX);
  VAMP_CSTAMP(1);

  // the (y, x) taps of the column: byte offsets in a lattice plane and weights (absent slots: k = 0, w = 0)
  unsigned cc[3][3];
  float wyx[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      cc[c][e] = ((unsigned) hy.k[c] * P.oX + hx.k[e]) * 4u;
      wyx[c][e] = hy.w[c] * hx.w[e];
    }
  const __amdgpu_buffer_rsrc_t rs_g1 = make_rsrc(gcol, gcol ? (size_t) P.B * nchan * OYX * 4 : 0);
  const __amdgpu_buffer_rsrc_t rs_g2 = make_rsrc(gcol2, gcol2 ? (size_t) P.B * nchan2 * OYX * 4 : 0);
  const __amdgpu_buffer_rsrc_t rs_wb = make_rsrc(Wb, (size_t) P.B * P.oZ * OYX * 4);
  const __amdgpu_buffer_rsrc_t rs_ds = make_rsrc(DS0, (size_t) P.B * P.oZ * OYX * 4);
  const __amdgpu_buffer_rsrc_t rs_none = make_rsrc(Wb, 0);
  const unsigned ocol = live ? col * 4u : 0x7ffffff0u;          // (a lane past the volume: dropped by the hardware)
  __syncthreads();
  // the heights whose taps reach the volume's planes (ascending); values read from LDS count as per-lane
  // for the compiler: as scalar offsets of the buffer accesses they would each get a readfirstlane loop
  int k_a = P.oZ, k_b = -1;
  for (int k = 0; k < P.oZ; ++k) {
    const int i0 = tz_i0[k];
    if (i0 + 1 >= 0 && i0 < P.Z) { k_a = min(k_a, k); k_b = max(k_b, k); }
  }
  k_a = __builtin_amdgcn_readfirstlane(k_a);
  k_b = __builtin_amdgcn_readfirstlane(k_b);

  // every wave runs the same number of rounds (the barriers below), a wave without a group idles through them
  for (int g0 = 0; g0 < ngrp; g0 += nwv) {
    const int cg = g0 + wave;
    const bool active = cg < ngrp;
    const bool dgrp = cg == ngrp - 1;
    // per channel slot: output offset, raw column gradients (their factors once the first taps are issued)
    bool on[G], second[G], has_g[G];
    unsigned ob[G], gb[G];
    float gw[G][2][3], gw3[G][3];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int ch = cg * G + u;
      on[u] = active && (dgrp ? u == 0 : ch < nchan + nchan2);
      second[u] = !dgrp && ch >= nchan;
      const int chc = (on[u] && !dgrp) ? (second[u] ? ch - nchan : ch) : 0;
      const int nc = second[u] ? nchan2 : nchan;
      ob[u] = dgrp ? (unsigned) b * V * 4u : ((unsigned) b * nc + chc) * V * 4u;
      gb[u] = ((unsigned) b * nc + chc) * OYX * 4u;
      has_g[u] = on[u] && !dgrp && (second[u] ? gcol2 != nullptr : gcol != nullptr);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 3; ++e)
          gw[u][c][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(has_g[u] ? (second[u] ? rs_g2 : rs_g1) : rs_none, cc[c][e], gb[u], 0));
      // a third y hit somewhere in the workgroup's columns (about 1 row in 100): its column gradients, raw
#pragma unroll
      for (int e = 0; e < 3; ++e)
        gw3[u][e] = y3 ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(has_g[u] ? (second[u] ? rs_g2 : rs_g1) : rs_none, cc[2][e], gb[u], 0)) : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rs_o1 = make_rsrc(dgrp ? gdens : gout, (size_t) P.B * (dgrp ? 1 : nchan) * V * 4);
    const __amdgpu_buffer_rsrc_t rs_o2 = make_rsrc(gout2, gout2 ? (size_t) P.B * nchan2 * V * 4 : 0);
    auto store_plane = [&](int z, const float (&v)[G], bool touched) __attribute__((always_inline)) {
      if (z < 0 || z >= P.Z) return;
      const unsigned zo = (unsigned) z * YX * 4u;
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (!on[u]) continue;
        const __amdgpu_buffer_rsrc_t rs = second[u] ? rs_o2 : rs_o1;
        if (OW) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[u]), rs, ocol, ob[u] + zo, 0);
        } else if (touched) {
          const float old = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ocol, ob[u] + zo, 0));
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(old + v[u]), rs, ocol, ob[u] + zo, 0);
        }
      }
    };
    float zero[G], lo[G], hi[G];
#pragma unroll
    for (int u = 0; u < G; ++u) zero[u] = lo[u] = hi[u] = 0.f;
    if (k_b < k_a) {                               // the lattice misses the volume (uniform)
      for (int z = 0; z < P.Z; ++z) store_plane(z, zero, false);
      continue;
    }
    int cur = __builtin_amdgcn_readfirstlane(tz_i0[k_a]);       // voxel plane of lo[]
    bool lo_t = false, hi_t = false;                // does any height touch the plane?
    bool first = true;

    for (int k0 = k_a; k0 <= k_b; k0 += HC) {
      const int hcn = min(HC, k_b - k0 + 1);
      // ---- this wave's share of the chunk's (source, height) items: 6 tap loads each, all in flight
      constexpr int ITW = (2 * HC + kCompMaxW - 1) / kCompMaxW;      // items per wave and pass (one pass at kCompMaxW waves)
      if (g0 > 0) __syncthreads();                  // (a second round of groups: the last one's readers are done)
      // the loads write LDS themselves (buffer_load ... lds: lane i's dword lands at the row's base + 4 i):
      // no staging registers, which is what lets three of these workgroups share a CU
      for (int i0 = 0; i0 < 2 * hcn; i0 += ITW * nwv) {
#pragma unroll
        for (int u = 0; u < ITW; ++u) {
          const int it = i0 + wave + u * nwv;
          if (it < 2 * hcn) {
            const int src = it >= hcn ? 1 : 0, t = it - src * hcn;
            const unsigned jo = ((unsigned) b * P.oZ + (unsigned) (P.oZ - 1 - (k0 + t))) * OYX * 4u;   // flip (bv2:443)
            const __amdgpu_buffer_rsrc_t rs = src ? rs_ds : rs_wb;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              if (c == 2 && !y3) continue;
#pragma unroll
              for (int e = 0; e < 3; ++e)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*) &taps[src][t][c * 3 + e][0], 4,
                                                     cc[c][e], jo, 0, 0);
            }
          }
        }
      }
      if (first) {
        // (the column gradients were issued before the taps: their factors wait for them alone)
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 3; ++e)
              gw[u][c][e] = dgrp ? (u == 0 ? wyx[c][e] : 0.f) : (has_g[u] ? wyx[c][e] * gw[u][c][e] : 0.f);
        VAMP_CSTAMP(2);
        for (int z = 0; z < min(cur, P.Z); ++z) store_plane(z, zero, false);
        first = false;
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0): this wave's taps are in LDS
      __syncthreads();
      // ---- the chunk's heights from LDS
      if (active) {
        const int src = dgrp ? 1 : 0;
        for (int t = 0; t < hcn; ++t) {
          const int k = k0 + t;
          float S[G];
#pragma unroll
          for (int u = 0; u < G; ++u) S[u] = 0.f;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              const float w = taps[src][t][c * 3 + e][lane];
#pragma unroll
              for (int u = 0; u < G; ++u) S[u] = __builtin_fmaf(w, gw[u][c][e], S[u]);
            }
          if (y3) {                                   // third y hit (same products, same order as the thread-per-column kernel)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              const float w = wyx[2][e];
              const float tvv = taps[src][t][6 + e][lane];
              if (dgrp) {
                S[0] = __builtin_fmaf(w, tvv, S[0]);
              } else {
                const float wbv = w * tvv;
#pragma unroll
                for (int u = 0; u < G; ++u)
                  if (has_g[u]) S[u] = __builtin_fmaf(wbv, gw3[u][e], S[u]);
              }
            }
          }
          const int i0z = __builtin_amdgcn_readfirstlane(tz_i0[k]);
          const float w0 = tz_w0[k], w1 = tz_w1[k];
          // the heights have moved past plane `cur`: store it, shift
          while (cur < i0z) {
            store_plane(cur, lo, lo_t);
#pragma unroll
            for (int u = 0; u < G; ++u) { lo[u] = hi[u]; hi[u] = 0.f; }
            lo_t = hi_t; hi_t = false;
            ++cur;
          }
#pragma unroll
          for (int u = 0; u < G; ++u) {
            lo[u] = __builtin_fmaf(w0, S[u], lo[u]);
            hi[u] = __builtin_fmaf(w1, S[u], hi[u]);
          }
          lo_t = hi_t = true;
        }
      }
      if (k0 + HC <= k_b) __syncthreads();          // (another chunk: its taps overwrite these)
    }
    VAMP_CSTAMP(3);
    store_plane(cur, lo, lo_t);
    store_plane(cur + 1, hi, hi_t);
    for (int z = max(cur + 2, 0); z < P.Z; ++z) store_plane(z, zero, false);
  }
  VAMP_CSTAMP(4);
}

// Pass-through channels (base; semantic with cat_seg) of the column gather, round 4: dL/ds_j[c] = g_voxel_output[c, j].
// Here the taps ARE per channel, so there is nothing to share between the threads of a column; what the
// thread-per-(column, 4 channels, z-segment) kernel lost was the chain of one load round trip per height
// (wave life 12 us for ~1 us of arithmetic).  A thread owns one (column, channel), loads the 6 taps of a chunk
// of HC heights together through a buffer descriptor (the tap offsets are per-lane constants, channel and height
// a scalar offset), forms the heights' plane sums, adds the third-y-hit rows (1 row in 100) from a second
// batch of loads into the same registers, and then walks the heights.  Same products in the same order as
// bev_gather_col_kernel<G, false, true, OW>.
//   grid: x = 256 columns, y = channels * B
template <int HC, bool OW>
__global__ void __launch_bounds__(256, 4)
bev_gather_pass_kernel(RenderParams P, const int4* __restrict__ tab, const float* __restrict__ ozs,
                       const float* __restrict__ g_vo, int vo_c0, float* __restrict__ gout, int nchan,
                       BetaTail btail) {
  beta_tail(btail);
  __shared__ int tz_i0[kBevMaxOZ];
  __shared__ float tz_w0[kBevMaxOZ], tz_w1[kBevMaxOZ];
  if ((int) threadIdx.x < P.oZ) {
    const AxisTap tz = axis_tap(ozs[threadIdx.x], P.lo[2], P.span[2], P.Z);
    tz_i0[threadIdx.x] = tz.i0; tz_w0[threadIdx.x] = tz.w0; tz_w1[threadIdx.x] = tz.w1;
  }
  __syncthreads();
  const unsigned YX = (unsigned) (P.Y * P.X);
  const unsigned col_raw = blockIdx.x * 256 + threadIdx.x;
  const bool live = col_raw < YX;
  const unsigned col = live ? col_raw : YX - 1;
  // (readfirstlane: the divisions run on the vector unit, and what derives from them would count as per-lane)
  const int ch = __builtin_amdgcn_readfirstlane((int) (blockIdx.y % nchan));
  const int b = __builtin_amdgcn_readfirstlane((int) (blockIdx.y / nchan));
  const int x = col % (unsigned) P.X, y = col / (unsigned) P.X;
  const AxisHits hx = load_axis_hits(tab, x), hy = load_axis_hits(tab, P.X + y);
  const bool y3 = __any(hy.n > 2);
  const unsigned V = (unsigned) P.Z * YX, OYX = (unsigned) (P.oY * P.oX);
  const int CO = P.C + (P.cat_seg ? P.K : 0);
  unsigned cc[3][3];
  float wyx[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      cc[c][e] = ((unsigned) hy.k[c] * P.oX + hx.k[e]) * 4u;
      wyx[c][e] = hy.w[c] * hx.w[e];
    }
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(g_vo, (size_t) P.B * CO * P.oZ * OYX * 4);
  const __amdgpu_buffer_rsrc_t rs_none = make_rsrc(g_vo, 0);
  const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(gout, (size_t) P.B * nchan * V * 4);
  const unsigned vb = ((unsigned) b * CO + (unsigned) (vo_c0 + ch)) * P.oZ * OYX * 4u;
  const unsigned ob = ((unsigned) b * nchan + (unsigned) ch) * V * 4u;
  const unsigned ocol = live ? col * 4u : 0x7ffffff0u;
  int k_a = P.oZ, k_b = -1;
  for (int k = 0; k < P.oZ; ++k) {
    const int i0 = tz_i0[k];
    if (i0 + 1 >= 0 && i0 < P.Z) { k_a = min(k_a, k); k_b = max(k_b, k); }
  }
  k_a = __builtin_amdgcn_readfirstlane(k_a);
  k_b = __builtin_amdgcn_readfirstlane(k_b);
  auto store_plane = [&](int z, float v, bool touched) __attribute__((always_inline)) {
    if (z < 0 || z >= P.Z) return;
    const unsigned zo = (unsigned) z * YX * 4u;
    if (OW) {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_o, ocol, ob + zo, 0);
    } else if (touched) {
      const float old = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_o, ocol, ob + zo, 0));
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(old + v), rs_o, ocol, ob + zo, 0);
    }
  };
  if (k_b < k_a) {                               // the lattice misses the volume
    for (int z = 0; z < P.Z; ++z) store_plane(z, 0.f, false);
    return;
  }
  int cur = __builtin_amdgcn_readfirstlane(tz_i0[k_a]);
  float lo = 0.f, hi = 0.f;
  bool lo_t = false, hi_t = false;
  for (int z = 0; z < min(cur, P.Z); ++z) store_plane(z, 0.f, false);
  for (int k0 = k_a; k0 <= k_b; k0 += HC) {
    float nv[HC][6], S[HC];
#pragma unroll
    for (int t = 0; t < HC; ++t) {
      const unsigned jo = vb + (unsigned) (P.oZ - 1 - min(k0 + t, k_b)) * OYX * 4u;         // flip (bv2:443)
      const __amdgpu_buffer_rsrc_t rs = k0 + t <= k_b ? rs_v : rs_none;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 3; ++e)
          nv[t][c * 3 + e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, cc[c][e], jo, 0));
    }
#pragma unroll
    for (int t = 0; t < HC; ++t) {
      float sv = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 3; ++e) sv = __builtin_fmaf(wyx[c][e], nv[t][c * 3 + e], sv);
      S[t] = sv;
    }
    if (y3) {                                      // third y hit somewhere in the wave
#pragma unroll
      for (int t = 0; t < HC; ++t) {
        const unsigned jo = vb + (unsigned) (P.oZ - 1 - min(k0 + t, k_b)) * OYX * 4u;
        const __amdgpu_buffer_rsrc_t rs = k0 + t <= k_b ? rs_v : rs_none;
#pragma unroll
        for (int e = 0; e < 3; ++e)
          nv[t][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, cc[2][e], jo, 0));
      }
#pragma unroll
      for (int t = 0; t < HC; ++t)
#pragma unroll
        for (int e = 0; e < 3; ++e) S[t] = __builtin_fmaf(wyx[2][e], nv[t][e], S[t]);
    }
#pragma unroll
    for (int t = 0; t < HC; ++t) {
      const int k = k0 + t;
      if (k <= k_b) {
        const int i0 = __builtin_amdgcn_readfirstlane(tz_i0[k]);
        const float w0 = tz_w0[k], w1 = tz_w1[k];
        while (cur < i0) {
          store_plane(cur, lo, lo_t);
          lo = hi; hi = 0.f;
          lo_t = hi_t; hi_t = false;
          ++cur;
        }
        lo = __builtin_fmaf(w0, S[t], lo);
        hi = __builtin_fmaf(w1, S[t], hi);
        lo_t = hi_t = true;
      }
    }
  }
  store_plane(cur, lo, lo_t);
  store_plane(cur + 1, hi, hi_t);
  for (int z = max(cur + 2, 0); z < P.Z; ++z) store_plane(z, 0.f, false);
}

// d beta partial sums the scan leaves in the workspace: one per workgroup of bev_scan_kernel, or -- when the
// forward kept its samples -- of bev_qscan_saved_kernel (the workspace holds the larger count)
static size_t bev_scan_blocks(const VampRenderDesc* d, bool saved = true) {
  if (saved) return (size_t) (((long) d->oY * d->oX + 63) / 64) * d->B;
  return (size_t) ((d->oX + 63) / 64) * ((d->oY + 3) / 4) * d->B;
}

// workspace: Q, Wb, DS0 [B, oZ, oY, oX] | axis tables | beta partials | what the forward keeps for
// the backward (VAMP_BEVFWD_SAVE): density samples [B, oZ, oY, oX], composited channels' samples
// [B, K + 3, oZ, oY, oX]
static size_t bev_one(const VampRenderDesc* d) {
  return align_up((size_t) d->B * d->oZ * d->oY * d->oX * sizeof(float), 256);
}
static size_t bev_saved_offset(const VampRenderDesc* d) {
  return 3 * bev_one(d) + 2 * align_up((size_t) 2 * (d->X + d->Y + d->Z) * sizeof(int4), 256) +
         align_up(bev_scan_blocks(d) * sizeof(float), 256);
}
static size_t bev_ws_bytes(const VampRenderDesc* d) {
  return bev_saved_offset(d) + (size_t) (1 + d->K + 3) * bev_one(d);
}

}  // namespace vamp

using namespace vamp;

extern "C" {

size_t vamp_render_bev_workspace_bytes(const VampRenderDesc* d) { return d ? bev_ws_bytes(d) : 0; }

int vamp_render_bev_forward_ex(const VampRenderDesc* d, const float* oxs, const float* oys,
                               const float* ozs, const float* bev_mids, const float* beta,
                               const void* density_feature, const void* semantic, const void* rgb,
                               const void* base, float* bev_rgb, float* bev_seg, float* bev_height,
                               float* voxel_density, float* voxel_output, const float* ozs_host,
                               void* workspace, size_t workspace_bytes, int flags, void* stream) {
  if (int e = validate(d)) return e;
  float *s0_save = nullptr, *ss_save = nullptr;
  if (flags & VAMP_BEVFWD_SAVE) {
    if (!workspace || workspace_bytes < bev_ws_bytes(d))
      return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) bev_ws_bytes(d));
    s0_save = reinterpret_cast<float*>(static_cast<char*>(workspace) + bev_saved_offset(d));
    ss_save = reinterpret_cast<float*>(static_cast<char*>(workspace) + bev_saved_offset(d) + bev_one(d));
  }
  VAMP_REQUIRE(d->oZ > 0 && d->oY > 0 && d->oX > 0, "det grid must be non-empty");
  VAMP_REQUIRE(oxs && oys && ozs && bev_mids && density_feature && semantic && rgb, "null pointer");
  VAMP_REQUIRE(base || d->C == 0, "base is NULL");
  VAMP_REQUIRE(bev_rgb && bev_seg && bev_height && voxel_density && voxel_output, "null output");
  VAMP_REQUIRE(beta || d->density_mode == VAMP_DENSITY_SIGMOID, "beta is NULL");
  const RenderParams P = to_params(d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the one-kernel forward only for heights the library has checked against its plane slabs (bev_fused_heights_fit)
  if (!(flags & VAMP_BEVFWD_TWO_KERNELS) && bev_fwd_fused_supported(d) && bev_fused_heights_fit(d, ozs_host))
    return launch_bev_fwd_fused(d, P, oxs, oys, ozs, bev_mids, beta, density_feature, semantic, rgb, base, bev_rgb,
                                bev_seg, bev_height, voxel_density, voxel_output, s0_save, ss_save, s);
  dim3 g1((d->oX + 63) / 64, (d->oY + 3) / 4, d->B);
  VAMP_REQUIRE(d->oZ <= kBevMaxOZ, "at most 64 det-grid heights");
  dim3 g2((d->oX + 63) / 64, (d->oY + 3) / 4, d->B * ((d->K + 3 + d->C + kBevNC - 1) / kBevNC));
#define VAMP_BEVF(T)                                                                              \
  do {                                                                                            \
    VAMP_TIMED(kProfBevFwd, s, (bev_density_kernel<T><<<g1, 256, 0, s>>>(                         \
        P, oxs, oys, ozs, bev_mids, beta, (const T*) density_feature, voxel_density, bev_height, s0_save))); \
    if (int e = check_launch("bev_density_kernel")) return e;                                     \
    VAMP_TIMED(kProfBevFwdCh, s, (bev_channels_kernel<T, kBevNC><<<g2, 256, 0, s>>>(              \
        P, oxs, oys, ozs, (const T*) semantic, (const T*) rgb, (const T*) base, voxel_density,    \
        bev_rgb, bev_seg, voxel_output, ss_save)));                                               \
  } while (0)
  if (d->in_dtype == VAMP_F32) VAMP_BEVF(float); else VAMP_BEVF(__hip_bfloat16);
#undef VAMP_BEVF
  return check_launch("bev_channels_kernel");
}

int vamp_render_forward_merged_supported(const VampRenderDesc* d, const float* ozs_host) {
  if (!d || validate(d)) return 0;
  return render_fwd_merged_supported(d) && bev_fused_heights_fit(d, ozs_host) ? 1 : 0;
}

int vamp_render_forward_merged(const VampRenderDesc* d, const float* mats, const float* us, const float* vs,
                               const float* ds, const float* mids, const float* oxs, const float* oys,
                               const float* ozs, const float* ozs_host, const float* bev_mids, const float* beta,
                               const void* density_feature, const void* semantic, const void* rgb,
                               const void* base, float* rgb_out, float* seg_out, float* depth_out,
                               float* bev_rgb, float* bev_seg, float* bev_height, float* voxel_density,
                               float* voxel_output, void* workspace, size_t workspace_bytes,
                               void* bev_workspace, size_t bev_workspace_bytes, float* grad_beta_zero, int flags,
                               void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && us && vs && ds && mids && oxs && oys && ozs && bev_mids, "null pointer");
  VAMP_REQUIRE(density_feature && semantic && rgb && (base || d->C == 0), "null input volume");
  VAMP_REQUIRE(rgb_out && seg_out && depth_out && bev_rgb && bev_seg && bev_height && voxel_density && voxel_output,
               "null output");
  VAMP_REQUIRE(beta || d->density_mode == VAMP_DENSITY_SIGMOID, "beta is NULL");
  VAMP_REQUIRE(render_fwd_merged_supported(d), "shapes outside the merged launch's limits (vamp_render_forward_merged_supported)");
  VAMP_REQUIRE(bev_fused_heights_fit(d, ozs_host), "ozs_host is NULL or the heights do not fit the BEV plane slabs (vamp_render_forward_merged_supported)");
  const size_t base_bytes = vamp_render_workspace_bytes(d);
  int* term = (workspace && workspace_bytes >= base_bytes) ? cam_term_ptr(d, workspace) : nullptr;
  float* rows = nullptr;
  if (flags & VAMP_RENDERFWD_SAVE_SAMPLES) {
    const size_t need = base_bytes + vamp_render_samples_bytes(d);
    if (!workspace || workspace_bytes < need)
      return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
    rows = reinterpret_cast<float*>(static_cast<char*>(workspace) + base_bytes);
  }
  float *s0_save = nullptr, *ss_save = nullptr;
  if (flags & VAMP_RENDERFWD_BEV_SAVE) {
    if (!bev_workspace || bev_workspace_bytes < bev_ws_bytes(d))
      return fail(VAMP_ENOSPC, "%s: bev_workspace %ld < %ld bytes", __func__, (long) bev_workspace_bytes, (long) bev_ws_bytes(d));
    s0_save = reinterpret_cast<float*>(static_cast<char*>(bev_workspace) + bev_saved_offset(d));
    ss_save = reinterpret_cast<float*>(static_cast<char*>(bev_workspace) + bev_saved_offset(d) + bev_one(d));
  }
  // VAMP_RENDERFWD_RANK: the camera tiles draw the backward's cell ranks; the caller finishes the prepare step with
  // vamp_render_camera_prepare_ex(VAMP_CAMPREP_RANKED) -- on this stream or another
  CamRankRefs rank{nullptr, nullptr, nullptr, 0, nullptr};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (flags & VAMP_RENDERFWD_RANK) {
    VAMP_REQUIRE(term != nullptr, "VAMP_RENDERFWD_RANK needs the render workspace");
    rank = cam_rank_refs(d, workspace);
    if (flags & VAMP_RENDERFWD_COUNTERS_CLEAN) {
      if (int e = debug_expect_range(rank.cnt, 64, 0, 0, s, "VAMP_RENDERFWD_COUNTERS_CLEAN: the render workspace's cell counters are zero")) return e;
    } else if (int e = launch_cam_counters_zero(d, workspace, s)) {
      return e;
    }
  }
  rank.zero_word = grad_beta_zero;
  return launch_render_fwd_merged(d, to_params(d), mats, us, vs, ds, mids, oxs, oys, ozs, bev_mids, beta, density_feature,
                                  semantic, rgb, base, rgb_out, seg_out, depth_out, term, rows, bev_rgb, bev_seg, bev_height,
                                  voxel_density, voxel_output, s0_save, ss_save, rank, s);
}

static int bev_zero_overwritten(const VampRenderDesc* d, int flags, float* gd, float* gs, float* gr,
                                float* gb, hipStream_t s) {
  const size_t vb = (size_t) d->B * d->Z * d->Y * d->X * sizeof(float);
  if ((flags & VAMP_BEVBWD_OVERWRITE_BASE) && gb)
    if (int ze = launch_zero(gb, vb * d->C, s)) return ze;
  if (flags & VAMP_BEVBWD_OVERWRITE_CAM) {
    if (int ze = launch_zero(gd, vb, s)) return ze;
    if (int ze = launch_zero(gs, vb * d->K, s)) return ze;
    if (int ze = launch_zero(gr, vb * 3, s)) return ze;
  }
  return VAMP_OK;
}

int vamp_render_bev_backward_ex(const VampRenderDesc* d, const float* oxs, const float* oys,
                             const float* ozs, const float* bev_mids, const float* beta,
                             const void* density_feature, const void* semantic, const void* rgb,
                             const void* base, const float* g_bev_rgb, const float* g_bev_seg,
                             const float* g_bev_height, const float* g_voxel_density,
                             const float* g_voxel_output, float* grad_density_feature,
                             float* grad_semantic, float* grad_rgb, float* grad_base,
                             float* grad_beta, const float* ozs_host, void* workspace,
                             size_t workspace_bytes, int flags, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(d->oZ > 0 && d->oY > 0 && d->oX > 0, "det grid must be non-empty");
  VAMP_REQUIRE(oxs && oys && ozs && bev_mids && density_feature && semantic && rgb, "null input");
  VAMP_REQUIRE(grad_density_feature && grad_semantic && grad_rgb, "null output");
  VAMP_REQUIRE(grad_base || d->C == 0 || !g_voxel_output, "grad_base is NULL");
  VAMP_REQUIRE((beta && grad_beta) || d->density_mode == VAMP_DENSITY_SIGMOID, "beta / grad_beta is NULL");
  if (!ozs_host && (flags & VAMP_BEVBWD_ONLY_BASE)) return VAMP_OK;   // the SKIP_BASE call of the pair does it all
  if (!ozs_host) {
    // float-atomic formulation: adds, so zero what the caller asked to have overwritten
    if (int e = bev_zero_overwritten(d, flags, grad_density_feature, grad_semantic, grad_rgb, grad_base,
                                     static_cast<hipStream_t>(stream))) return e;
    return launch_bev_bwd_v1(d, oxs, oys, ozs, bev_mids, beta, density_feature, semantic, rgb, base,
                             g_bev_rgb, g_bev_seg, g_bev_height, g_voxel_density, g_voxel_output,
                             grad_density_feature, grad_semantic, grad_rgb, grad_base, grad_beta,
                             stream);
  }
  VAMP_REQUIRE(d->oZ <= kBevMaxOZ, "at most 64 det-grid heights");
  const size_t need = bev_ws_bytes(d);
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  const RenderParams P = to_params(d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t one = align_up((size_t) d->B * d->oZ * d->oY * d->oX * sizeof(float), 256);
  const int tot_ax = d->X + d->Y + d->Z;
  // VAMP_BEVBWD_ONLY_BASE / _SKIP_BASE: the pass-through (base) gather needs neither q nor the
  // scan and nobody waits for grad_base, so a caller may issue it as a call of its own -- behind
  // the event the camera gather waits for, or on another stream; that call has its own axis table
  const bool only_base = (flags & VAMP_BEVBWD_ONLY_BASE) != 0, skip_base = (flags & VAMP_BEVBWD_SKIP_BASE) != 0;
  VAMP_REQUIRE(!(only_base && skip_base), "ONLY_BASE and SKIP_BASE exclude each other");
  const size_t tab_bytes = align_up((size_t) 2 * (d->X + d->Y + d->Z) * sizeof(int4), 256);
  int4* tab = reinterpret_cast<int4*>(static_cast<char*>(workspace) + 3 * one + (only_base ? tab_bytes : 0));
  float* Q = static_cast<float*>(workspace);
  float* Wb = reinterpret_cast<float*>(static_cast<char*>(workspace) + one);
  float* DS0 = reinterpret_cast<float*>(static_cast<char*>(workspace) + 2 * one);
  float* beta_part = reinterpret_cast<float*>(static_cast<char*>(workspace) + 3 * one + 2 * tab_bytes);

  // z-range of volume planes the lattice touches (host copy of the det-grid heights)
  int z_lo = d->Z, z_hi = -1;
  for (int k = 0; k < d->oZ; ++k) {
    const float g = ((ozs_host[k] - d->lo[2]) / d->span[2]) * 2.0f - 1.0f;
    const float f = ((g + 1.0f) / 2.0f) * (float) (d->Z - 1);
    const int i0 = (int) floorf(f);
    z_lo = i0 < z_lo ? i0 : z_lo;
    z_hi = i0 + 1 > z_hi ? i0 + 1 : z_hi;
  }
  z_lo = z_lo < 0 ? 0 : z_lo;
  z_hi = z_hi > d->Z - 1 ? d->Z - 1 : z_hi;
  if (z_lo > z_hi) {
    // the det lattice misses the volume: only zeros to write.  A split pair zeroes each buffer in the
    // half that owns it (ONLY_BASE is issued behind the event the camera gather waits for: zeroing the
    // three camera tensors again there would race with, or wipe, that gather's sums)
    int zf = flags;
    if (flags & VAMP_BEVBWD_ONLY_BASE) zf &= ~VAMP_BEVBWD_OVERWRITE_CAM;
    if (flags & VAMP_BEVBWD_SKIP_BASE) zf &= ~VAMP_BEVBWD_OVERWRITE_BASE;
    return bev_zero_overwritten(d, zf, grad_density_feature, grad_semantic, grad_rgb, grad_base,
                                static_cast<hipStream_t>(stream));
  }

  dim3 gq((d->oX + 63) / 64, d->oY, d->B);
  const size_t q_lds = (size_t) d->oZ * 4 * 64 * sizeof(float);
  dim3 gs((d->oX + 63) / 64, (d->oY + 3) / 4, d->B);
  dim3 gg((d->X + 63) / 64, (d->Y + 3) / 4, d->B * (z_hi - z_lo + 1));
  const bool saved = (flags & VAMP_BEVBWD_SAVED_VALID) != 0;
  const float* s0_saved = saved ? reinterpret_cast<const float*>(static_cast<char*>(workspace) + bev_saved_offset(d)) : nullptr;
  const float* ss_saved = saved ? reinterpret_cast<const float*>(static_cast<char*>(workspace) + bev_saved_offset(d) + bev_one(d)) : nullptr;
#define VAMP_BEVB(T)                                                                              \
  do {                                                                                            \
    if (saved) {                                                                                  \
      const int nwj = d->oZ < kQsMaxWaves ? d->oZ : kQsMaxWaves;   /* a wave per height */             \
      const dim3 gqf((unsigned) (((long) d->oY * d->oX + 63) / 64), d->B);                           \
      const size_t qlds = (size_t) 4 * d->oZ * 64 * sizeof(float);                                    \
      if (d->K + 3 == 21 && (size_t) d->B * 21 * d->oZ * d->oY * d->oX * 4 < 0x7fffffffull)           \
        VAMP_TIMED(kProfBevBwd, s, (bev_qscan_saved_kernel<21><<<gqf, nwj * 64, qlds, s>>>(            \
            P, bev_mids, beta, s0_saved, ss_saved, g_bev_rgb, g_bev_seg, g_bev_height, g_voxel_density, Wb, DS0, beta_part))); \
      else                                                                                            \
        VAMP_TIMED(kProfBevBwd, s, (bev_qscan_saved_kernel<0><<<gqf, nwj * 64, qlds, s>>>(             \
            P, bev_mids, beta, s0_saved, ss_saved, g_bev_rgb, g_bev_seg, g_bev_height, g_voxel_density, Wb, DS0, beta_part))); \
      if (int e = check_launch("bev_qscan_saved_kernel")) return e;                               \
      break;                                                                                      \
    } else {                                                                                      \
    if (q_lds > 60 * 1024 &&                                                                      \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&bev_q_kernel<T>),                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int) q_lds) != hipSuccess) \
      return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);                           \
    VAMP_TIMED(kProfBevBwdQ, s, (bev_q_kernel<T><<<gq, 256, q_lds, s>>>(                              \
        P, oxs, oys, ozs, (const T*) semantic, (const T*) rgb, g_bev_rgb, g_bev_seg, Q)));        \
    if (int e = check_launch("bev_q_kernel")) return e;                                           \
    }                                                                                             \
    VAMP_TIMED(kProfBevBwd, s, (bev_scan_kernel<T><<<gs, 256, 0, s>>>(                            \
        P, oxs, oys, ozs, bev_mids, beta, (const T*) density_feature, s0_saved, g_bev_height,     \
        g_voxel_density, Q, Wb, DS0, beta_part)));                                                \
    if (int e = check_launch("bev_scan_kernel")) return e;                                        \
  } while (0)
  if (!only_base) {
    if (d->in_dtype == VAMP_F32) VAMP_BEVB(float); else VAMP_BEVB(__hip_bfloat16);
  }
#undef VAMP_BEVB
  // the beta partial sums of the scan are added up where nobody waits: in the second call of a split
  // pair (ONLY_BASE; the first, SKIP_BASE, leaves them in the workspace), else right here
  // ... by the first workgroup of the first column-gather launch of that call (beta_tail), or, where no
  // such launch follows, by a launch of its own
  BetaTail btail{nullptr, 0, nullptr, nullptr};
  if (!skip_base && d->density_mode == VAMP_DENSITY_SDF_LAPLACE)
    btail = BetaTail{beta_part, (int) bev_scan_blocks(d, saved), beta, grad_beta};
  const BetaTail no_tail{nullptr, 0, nullptr, nullptr};
  auto take_tail = [&]() { const BetaTail t = btail; btail = no_tail; return t; };
  // lattice points within one voxel's trilinear support, per axis
  bool fits = true;
  const int nvox[3] = {d->X, d->Y, d->Z};
  for (int a = 0; a < 3; ++a) {
    const float e = d->span[a] / (float) (nvox[a] - 1);
    if (!(d->det_step[a] > 0.f) || (int) floorf(2.0f * e / d->det_step[a]) + 1 > kMaxT) fits = false;
  }
  // the column gather walks the heights upwards and keeps their z taps in an LDS table
  if (d->oZ > kBevMaxOZ) fits = false;
  for (int k = 1; k < d->oZ; ++k)
    if (!(ozs_host[k] > ozs_host[k - 1])) fits = false;
  if (fits) {
    // 32-bit element offsets inside the gather
    const size_t lim = 0x7fffffffu;
    VAMP_REQUIRE((size_t) d->B * (d->K > d->C ? d->K : d->C) * d->Z * d->Y * d->X < lim &&
                 (size_t) d->B * (d->C + d->K) * d->oZ * d->oY * d->oX < lim,
                 "tensor too large for the 32-bit offsets of the BEV gather");
    // the axis table depends on the grids only: a caller that kept the workspace says so
    if (!(flags & VAMP_BEVBWD_TABLE_VALID)) {
      VAMP_TIMED(kProfAux, s, (bev_axis_table_kernel<<<(tot_ax + 255) / 256, 256, 0, s>>>(P, oxs, oys, ozs, tab)));
      if (int e = check_launch("bev_axis_table_kernel")) return e;
    }
    const bool vo_sem = g_voxel_output && d->cat_seg;
    const int ow = (flags & VAMP_BEVBWD_OVERWRITE_BASE) ? 1 : 0;
    const int owc = (flags & VAMP_BEVBWD_OVERWRITE_CAM) ? 3 : 0;
    // z-segments: a segment redoes the plane sums of the one or two heights that straddle its
    // lower edge, so as few as fill the chip (cfg-B, composited + pass-through launch, us:
    // 1 segment 40 + 29, 2 segments 40 + 25, 4 segments 43 + 25, 8 segments 73 + 29)
    const long wgs = (((long) d->Y * d->X + 255) / 256) * d->B * ((d->C + kColG - 1) / kColG);
    int nseg = (int) std::min<long>(std::max<long>(1, (1250 + wgs - 1) / std::max<long>(1, wgs)), std::max(1, d->Z / 4));
    const int zseg = (d->Z + nseg - 1) / nseg;
    nseg = (d->Z + zseg - 1) / zseg;
    auto grid = [&](int nchan, int g = kColG) {
      return dim3((unsigned) (((long) d->Y * d->X + 255) / 256), d->B * nseg * ((nchan + g - 1) / g));
    };
    // semantic + rgb + density in one launch (they share the Wb taps of a height)
    // (the descriptors of the round-4 kernel address bytes: 2 GB per tensor)
    const bool comp_ok = (size_t) d->B * d->K * d->Z * d->Y * d->X * 4 < lim && (size_t) d->B * d->oZ * d->oY * d->oX * 4 < lim;
    const int ngrp_c = (d->K + 3 + kCompG - 1) / kCompG + 1;
    const int nwv_c = ngrp_c < kCompMaxW ? ngrp_c : kCompMaxW;
    const dim3 grid_comp((unsigned) ((((long) d->Y * d->X + 63) / 64 + 7) / 8 * 8), d->B);
    if (only_base) {}
    else if (comp_ok && owc) VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_comp_kernel<kCompG, kCompHC, true><<<grid_comp, nwv_c * 64, 0, s>>>(
        P, tab, ozs, g_bev_seg, g_bev_rgb, Wb, DS0, grad_density_feature, grad_semantic, grad_rgb, d->K, 3, take_tail())));
    else if (comp_ok) VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_comp_kernel<kCompG, kCompHC, false><<<grid_comp, nwv_c * 64, 0, s>>>(
        P, tab, ozs, g_bev_seg, g_bev_rgb, Wb, DS0, grad_density_feature, grad_semantic, grad_rgb, d->K, 3, take_tail())));
    else if (owc) VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_col_kernel<kColGC, true, false, true><<<grid(d->K + 3, kColGC), 256, 0, s>>>(
        P, tab, ozs, g_bev_seg, g_bev_rgb, nullptr, -1, Wb, DS0, grad_density_feature, grad_semantic, grad_rgb,
        d->K, 3, zseg, 1, 0, take_tail())));
    else VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_col_kernel<kColGC, true, false, false><<<grid(d->K + 3, kColGC), 256, 0, s>>>(
        P, tab, ozs, g_bev_seg, g_bev_rgb, nullptr, -1, Wb, DS0, grad_density_feature, grad_semantic, grad_rgb,
        d->K, 3, zseg, 1, 0, take_tail())));
    if (int e = check_launch("bev_gather_col_kernel")) return e;
    // pass-through gradients (voxel_output): the semantic part (cat_seg) adds, base is its own tensor
    if (vo_sem && !only_base)
      VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_col_kernel<kColG, false, true, false><<<grid(d->K), 256, 0, s>>>(
          P, tab, ozs, nullptr, nullptr, g_voxel_output, d->C, Wb, DS0, grad_density_feature, grad_semantic,
          nullptr, d->K, 0, zseg, 0, 0, take_tail())));
    const bool pass_ok = (size_t) d->B * (d->C + (d->cat_seg ? d->K : 0)) * d->oZ * d->oY * d->oX * 4 < lim &&
                         (size_t) d->B * d->C * d->Z * d->Y * d->X * 4 < lim;
    const dim3 grid_pass((unsigned) (((long) d->Y * d->X + 255) / 256), (unsigned) (d->B * d->C));
    if (skip_base) {}
    else if (d->C > 0 && g_voxel_output && pass_ok && ow)
      VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_pass_kernel<kCompHC, true><<<grid_pass, 256, 0, s>>>(
          P, tab, ozs, g_voxel_output, 0, grad_base, d->C, take_tail())));
    else if (d->C > 0 && g_voxel_output && pass_ok)
      VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_pass_kernel<kCompHC, false><<<grid_pass, 256, 0, s>>>(
          P, tab, ozs, g_voxel_output, 0, grad_base, d->C, take_tail())));
    else if (d->C > 0 && g_voxel_output && ow)
      VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_col_kernel<kColG, false, true, true><<<grid(d->C), 256, 0, s>>>(
          P, tab, ozs, nullptr, nullptr, g_voxel_output, 0, Wb, DS0, grad_density_feature, grad_base, nullptr,
          d->C, 0, zseg, 0, 0, take_tail())));
    else if (d->C > 0 && g_voxel_output)
      VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_col_kernel<kColG, false, true, false><<<grid(d->C), 256, 0, s>>>(
          P, tab, ozs, nullptr, nullptr, g_voxel_output, 0, Wb, DS0, grad_density_feature, grad_base, nullptr,
          d->C, 0, zseg, 0, 0, take_tail())));
    else if (d->C > 0 && ow)
      if (int ze = launch_zero(grad_base, (size_t) d->B * d->C * d->Z * d->Y * d->X * sizeof(float), s)) return ze;
  }
  else if (!only_base) {
    // (the generic kernel does all four tensors at once: of a split pair of calls the SKIP_BASE one
    // does the whole job and the ONLY_BASE one nothing)
    // the generic kernel adds: zero what the caller asked to have overwritten
    if (int e = bev_zero_overwritten(d, flags, grad_density_feature, grad_semantic, grad_rgb, grad_base, s))
      return e;
    VAMP_TIMED(kProfBevBwdGather, s, (bev_gather_generic_kernel<<<gg, 256, 0, s>>>(
        P, oxs, oys, ozs, g_bev_rgb, g_bev_seg, g_voxel_output, Wb, DS0, grad_density_feature,
        grad_semantic, grad_rgb, grad_base, z_lo, z_hi)));
  }
  if (btail.part)          // no column-gather launch took the partial sums over
    if (int e = launch_beta_reduce(btail.part, btail.n, btail.beta_raw, btail.grad_beta, s)) return e;
  return check_launch("bev_gather_kernel");
}

int vamp_render_bev_backward(const VampRenderDesc* d, const float* oxs, const float* oys,
                             const float* ozs, const float* bev_mids, const float* beta,
                             const void* density_feature, const void* semantic, const void* rgb,
                             const void* base, const float* g_bev_rgb, const float* g_bev_seg,
                             const float* g_bev_height, const float* g_voxel_density,
                             const float* g_voxel_output, float* grad_density_feature,
                             float* grad_semantic, float* grad_rgb, float* grad_base,
                             float* grad_beta, const float* ozs_host, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return vamp_render_bev_backward_ex(d, oxs, oys, ozs, bev_mids, beta, density_feature, semantic, rgb, base,
                                     g_bev_rgb, g_bev_seg, g_bev_height, g_voxel_density, g_voxel_output,
                                     grad_density_feature, grad_semantic, grad_rgb, grad_base, grad_beta,
                                     ozs_host, workspace, workspace_bytes, 0, stream);
}

int vamp_render_bev_forward(const VampRenderDesc* d, const float* oxs, const float* oys,
                            const float* ozs, const float* bev_mids, const float* beta,
                            const void* density_feature, const void* semantic, const void* rgb,
                            const void* base, float* bev_rgb, float* bev_seg, float* bev_height,
                            float* voxel_density, float* voxel_output, void* stream) {
  return vamp_render_bev_forward_ex(d, oxs, oys, ozs, bev_mids, beta, density_feature, semantic, rgb, base,
                                    bev_rgb, bev_seg, bev_height, voxel_density, voxel_output, nullptr, nullptr, 0, 0,
                                    stream);
}

}  // extern "C"
