// LIFT kernels for gfx950: every voxel pulls a trilinear sample of
// depth[d,h,w] * feat[c,h,w] from each camera and averages over the cameras that
// hit it.  Reference call sites: base_vampire2.py:550-553 (outer product),
// :351-388 (get_pixel), :483-516 (get_voxel_feats).
//
// The outer product is never materialised: the trilinear sample of a rank-1
// (depth x feat) volume separates into
//     sum_{4 (h,w) taps} w_hw * feat[c,h,w] * ( sum_{2 d taps} w_d * depth[d,h,w] ).
// Memory-bound gather; no MFMA.  One thread per voxel, lanes along x so the
// [B,C,Z,Y,X] stores coalesce; features are read channel-last (one 64-byte run
// per tap for C = 16) from a transposed copy made by a tiny pre-pass.
#include "lift_common.hpp"
#include "depth_softmax.hpp"

namespace vamp {

// ---------------------------------------------------------------------------
// feat [BN, C, HW] (f32 or bf16) -> channel-last fp32 [BN, HW, C]
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void feat_to_channel_last_tile(const T* __restrict__ feat, float* __restrict__ out,
                                                          int C, int HW, long bn, int p0, int c0,
                                                          float (&tile)[64][65]) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    int c = c0 + j, p = p0 + tx;
    tile[j][tx] = (c < C && p < HW) ? ldf(feat, (bn * C + c) * HW + p) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    int p = p0 + j, c = c0 + tx;
    if (p < HW && c < C) out[(bn * HW + p) * C + c] = tile[tx][j];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) feat_to_channel_last(const T* __restrict__ feat,
                                                            float* __restrict__ out, int C, int HW) {
  __shared__ float tile[64][65];
  feat_to_channel_last_tile<T>(feat, out, C, HW, blockIdx.z, blockIdx.x * 64, blockIdx.y * 64, tile);
}

// ---------------------------------------------------------------------------
// Both lift operands in one launch (the producer side): workgroups [0, n_sm) turn
// the raw `mapping_along_depth` logits into the depth distribution (softmax over D,
// base_vampire2.py:550), the rest make the channel-last copy of the features.  The two jobs are
// independent, so the 6 600 softmax tiles and the 1 100 transpose tiles of cfg-B share one grid.
// ---------------------------------------------------------------------------
template <typename TL, bool REG>
__global__ void __launch_bounds__(256)
lift_operands_kernel(const TL* __restrict__ logits, float* __restrict__ depth, int D, long HW, int sm_tiles,
                     unsigned n_sm, const float* __restrict__ feat, float* __restrict__ feat_cl, int C,
                     int ptiles, int ctiles) {
  __shared__ union {
    SoftmaxLds sm;
    float tile[64][65];
  } L;
  if (blockIdx.x < n_sm) {
    depth_softmax_tile<TL, REG>(logits, depth, D, HW, blockIdx.x / sm_tiles, blockIdx.x % sm_tiles, L.sm);
  } else {
    unsigned r = blockIdx.x - n_sm;
    const int px = r % ptiles;
    r /= ptiles;
    const int ct = r % ctiles;
    feat_to_channel_last_tile<float>(feat, feat_cl, C, (int) HW, (long) (r / ctiles), px * 64, ct * 64, L.tile);
  }
}

// channel-last fp32 [BN, HW, C] -> [BN, C, HW]
__global__ void __launch_bounds__(256) feat_to_channel_first(const float* __restrict__ in,
                                                             float* __restrict__ out, int C, int HW) {
  __shared__ float tile[64][65];
  const long bn = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    int p = p0 + j, c = c0 + tx;
    tile[j][tx] = (p < HW && c < C) ? in[(bn * HW + p) * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    int c = c0 + j, p = p0 + tx;
    if (c < C && p < HW) out[(bn * C + c) * HW + p] = tile[tx][j];
  }
}

// ---------------------------------------------------------------------------
// depth interpolation at the four (h,w) taps:  dep[j] = sum_d w_d depth[d, iy, ix]
// (zero padding: out-of-range taps contribute nothing)
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void depth_taps(const LiftParams& P, const T* __restrict__ dptr,
                                           const LiftTap& t, float dep[4]) {
  dep[0] = dep[1] = dep[2] = dep[3] = 0.f;
  if (!P.use_depth) {
    // D == 1: the single plane has weight wz0 (= 1) at iz0 == 0
    const float w = (t.iz0 == 0 ? t.wz0 : 0.f) + (t.iz0 == -1 ? t.wz1 : 0.f);
    dep[0] = dep[1] = dep[2] = dep[3] = w;
    return;
  }
  // branch-free (clamped address, zero weight for taps outside the volume): the eight loads
  // are independent and issue together
  const long plane = (long) P.fH * P.fW;
  const bool x0in = t.ix0 >= 0 && t.ix0 < P.fW, x1in = t.ix0 + 1 >= 0 && t.ix0 + 1 < P.fW;
  const int x0c = min(max(t.ix0, 0), P.fW - 1), x1c = min(max(t.ix0 + 1, 0), P.fW - 1);
#pragma unroll
  for (int kz = 0; kz < 2; ++kz) {
    const int iz = t.iz0 + kz;
    const bool zin = iz >= 0 && iz < P.D;
    const float wz = zin ? (kz ? t.wz1 : t.wz0) : 0.f;
    const int izc = min(max(iz, 0), P.D - 1);
#pragma unroll
    for (int ky = 0; ky < 2; ++ky) {
      const int iy = t.iy0 + ky;
      const bool yin = iy >= 0 && iy < P.fH;
      const long row = izc * plane + (long) min(max(iy, 0), P.fH - 1) * P.fW;
      dep[ky * 2 + 0] += ((yin && x0in) ? wz : 0.f) * ldf(dptr, row + x0c);
      dep[ky * 2 + 1] += ((yin && x1in) ? wz : 0.f) * ldf(dptr, row + x1c);
    }
  }
}

// ---------------------------------------------------------------------------
// LIFT forward.  Block = TX*TY*TZ (= 256) voxels; CH channels per pass.
// ---------------------------------------------------------------------------
// EMIT (a backward will follow): the kernel also does the counting half of the backward's pixel
// sort -- every valid (voxel, camera) pair is counted in its cell and its taps are left in the
// workspace (lift_emit_pair), so nothing on the backward projects a voxel again.
template <typename T, int CH, int TX, int TY, int TZ, bool EMIT>
__global__ void __launch_bounds__(TX* TY* TZ, 4)      // (4 waves per SIMD: the EMIT variant would take 139 registers, i.e. 3: 48.5 -> 44.5 us; 5 spills: 55)
lift_fwd_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                const float* __restrict__ ys, const float* __restrict__ zs,
                const T* __restrict__ depth, const float* __restrict__ feat_cl,
                float* __restrict__ out, uint64_t* __restrict__ hits, LiftEmit E) {
  const int tid = threadIdx.x;
  const int x = blockIdx.x * TX + (tid % TX);
  const int y = blockIdx.y * TY + ((tid / TX) % TY);
  const int zblocks = (P.Z + TZ - 1) / TZ;
  const int b = blockIdx.z / zblocks;
  const int z = (blockIdx.z % zblocks) * TZ + tid / (TX * TY);
  if (x >= P.X || y >= P.Y || z >= P.Z) return;

  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long V = (long) P.Z * P.Y * P.X;
  const long vox = ((long) z * P.Y + y) * P.X + x;
  const long HW = (long) P.fH * P.fW;
  const int nchunk = P.C / CH;

  unsigned wmask = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    float sum[CH];
    uint64_t cnt = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) sum[c] = 0.f;

    for (int n = 0; n < P.N; ++n) {
      const long bn = (long) b * P.N + n;
      const LiftTap t = lift_project<true>(P, mats + bn * 48, vx, vy, vz);
      if (EMIT && chunk == 0)        // (wave-uniform here: the lanes part ways at the next line)
        if (lift_emit_pair(P, E, t, true, bn, V, vox, tid & 63)) wmask |= 1u << (n & 31);
      if (!t.valid) continue;
      float dep[4];
      depth_taps<T>(P, depth + bn * P.D * HW, t, dep);
      const float w[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
      float acc[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
        const bool in = iy >= 0 && iy < P.fH && ix >= 0 && ix < P.fW;
        const float wd = in ? w[j] * dep[j] : 0.f;         // branch-free: clamp + zero weight
        const float4* f4 = reinterpret_cast<const float4*>(
            feat_cl + (bn * HW + (long) min(max(iy, 0), P.fH - 1) * P.fW + min(max(ix, 0), P.fW - 1)) * P.C +
            chunk * CH);
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const float4 f = f4[q];
          acc[q * 4 + 0] = __builtin_fmaf(wd, f.x, acc[q * 4 + 0]);
          acc[q * 4 + 1] = __builtin_fmaf(wd, f.y, acc[q * 4 + 1]);
          acc[q * 4 + 2] = __builtin_fmaf(wd, f.z, acc[q * 4 + 2]);
          acc[q * 4 + 3] = __builtin_fmaf(wd, f.w, acc[q * 4 + 3]);
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        sum[c] += acc[c];
        cnt += (uint64_t) (fabsf(acc[c]) > 0.f) << (4 * c);   // per-channel hit count (bv2:509)
      }
    }
    float* o = out + ((long) b * P.C + chunk * CH) * V + vox;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float denom = (float) ((cnt >> (4 * c)) & 15) + 1e-6f;
      o[(long) c * V] = sum[c] / denom;
    }
    if (hits) hits[((long) b * V + vox) * nchunk + chunk] = cnt;
  }
  if (EMIT && E.amask) E.amask[(long) b * V + vox] = wmask;
}

// The same walk without the samples: counts and emits the pairs for a backward whose forward did
// not (vamp_lift_prepare; vamp_lift_backward without VAMP_LIFTBWD_CELLS_VALID).
template <int TX, int TY>
__global__ void __launch_bounds__(TX* TY)
lift_pairs_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                  const float* __restrict__ ys, const float* __restrict__ zs, LiftEmit E) {
  const int tid = threadIdx.x;
  const int x = blockIdx.x * TX + (tid % TX);
  const int y = blockIdx.y * TY + (tid / TX);
  const int b = blockIdx.z / P.Z, z = blockIdx.z % P.Z;
  if (x >= P.X || y >= P.Y) return;
  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long V = (long) P.Z * P.Y * P.X;
  const long vox = ((long) z * P.Y + y) * P.X + x;
  unsigned wmask = 0;
  for (int n = 0; n < P.N; ++n) {
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project<true>(P, mats + bn * 48, vx, vy, vz);
    if (lift_emit_pair(P, E, t, true, bn, V, vox, tid & 63)) wmask |= 1u << (n & 31);
  }
  if (E.amask) E.amask[(long) b * V + vox] = wmask;
}

// ---------------------------------------------------------------------------
// LIFT forward from the materialised frustum_feats [B,N,C,D,fH,fW]
// (signature-compatible path for get_voxel_feats, bv2:483).  8-tap gather per
// channel in aten's tap order.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
lift_fwd_dense_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                      const float* __restrict__ ys, const float* __restrict__ zs,
                      const float* __restrict__ ff, float* __restrict__ out,
                      uint64_t* __restrict__ hits) {
  const long V = (long) P.Z * P.Y * P.X;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * P.B) return;
  const int b = gid / V;
  const long vox = gid % V;
  const int x = vox % P.X, y = (vox / P.X) % P.Y, z = vox / ((long) P.X * P.Y);
  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long HW = (long) P.fH * P.fW, DHW = HW * P.D;
  const int nchunk = (P.C + 15) / 16;

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int c_lo = chunk * 16, c_n = min(16, P.C - c_lo);
    float sum[16];
    uint64_t cnt = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) sum[c] = 0.f;
    for (int n = 0; n < P.N; ++n) {
      const long bn = (long) b * P.N + n;
      const LiftTap t = lift_project(P, mats + bn * 48, vx, vy, vz);
      if (!t.valid) continue;
      float acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = t.iz0 + (k >> 2), iy = t.iy0 + ((k >> 1) & 1), ix = t.ix0 + (k & 1);
        if (iz < 0 || iz >= P.D || iy < 0 || iy >= P.fH || ix < 0 || ix >= P.fW) continue;
        const float w = ((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0) *
                        ((k & 4) ? t.wz1 : t.wz0);
        const float* src = ff + (bn * P.C + c_lo) * DHW + iz * HW + (long) iy * P.fW + ix;
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c < c_n) acc[c] += src[c * DHW] * w;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        sum[c] += acc[c];
        cnt += (uint64_t) (fabsf(acc[c]) > 0.f) << (4 * c);
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < c_n)
        out[((long) b * P.C + c_lo + c) * V + vox] =
            sum[c] / ((float) ((cnt >> (4 * c)) & 15) + 1e-6f);
    if (hits) hits[((long) b * V + vox) * nchunk + chunk] = cnt;
  }
}

// ---------------------------------------------------------------------------
// LIFT backward (v1): one thread per voxel, float atomics into channel-last
// grad_feat and into grad_depth.
// ---------------------------------------------------------------------------
template <typename T, int CH, int TX, int TY, int TZ>
__global__ void __launch_bounds__(TX* TY* TZ)
lift_bwd_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                const float* __restrict__ ys, const float* __restrict__ zs,
                const T* __restrict__ depth, const float* __restrict__ feat_cl,
                const float* __restrict__ gout, const uint64_t* __restrict__ hits,
                float* __restrict__ gdepth, float* __restrict__ gfeat_cl) {
  const int tid = threadIdx.x;
  const int x = blockIdx.x * TX + (tid % TX);
  const int y = blockIdx.y * TY + ((tid / TX) % TY);
  const int zblocks = (P.Z + TZ - 1) / TZ;
  const int b = blockIdx.z / zblocks;
  const int z = (blockIdx.z % zblocks) * TZ + tid / (TX * TY);
  if (x >= P.X || y >= P.Y || z >= P.Z) return;

  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long V = (long) P.Z * P.Y * P.X;
  const long vox = ((long) z * P.Y + y) * P.X + x;
  const long HW = (long) P.fH * P.fW;
  const int nchunk = P.C / CH;

  for (int n = 0; n < P.N; ++n) {
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project(P, mats + bn * 48, vx, vy, vz);
    if (!t.valid) continue;
    float dep[4];
    depth_taps<T>(P, depth + bn * P.D * HW, t, dep);
    const float w[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
    float gdep[4] = {0.f, 0.f, 0.f, 0.f};   // d loss / d dep[j]
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const uint64_t cnt = hits[((long) b * V + vox) * nchunk + chunk];
      float gs[CH];
      const float* g = gout + ((long) b * P.C + chunk * CH) * V + vox;
#pragma unroll
      for (int c = 0; c < CH; ++c)
        gs[c] = g[(long) c * V] / ((float) ((cnt >> (4 * c)) & 15) + 1e-6f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
        if (iy < 0 || iy >= P.fH || ix < 0 || ix >= P.fW) continue;
        const long pix = (bn * HW + (long) iy * P.fW + ix) * P.C + chunk * CH;
        const float4* f4 = reinterpret_cast<const float4*>(feat_cl + pix);
        const float wd = w[j] * dep[j];
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const float4 f = f4[q];
          dot = __builtin_fmaf(f.x, gs[q * 4 + 0], dot);
          dot = __builtin_fmaf(f.y, gs[q * 4 + 1], dot);
          dot = __builtin_fmaf(f.z, gs[q * 4 + 2], dot);
          dot = __builtin_fmaf(f.w, gs[q * 4 + 3], dot);
        }
        gdep[j] = __builtin_fmaf(w[j], dot, gdep[j]);
#pragma unroll
        for (int c = 0; c < CH; ++c) atomicAdd(gfeat_cl + pix + c, wd * gs[c]);
      }
    }
    if (P.use_depth && gdepth) {
      float* gd = gdepth + bn * P.D * HW;
#pragma unroll
      for (int kz = 0; kz < 2; ++kz) {
        const int iz = t.iz0 + kz;
        if (iz < 0 || iz >= P.D) continue;
        const float wz = kz ? t.wz1 : t.wz0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
          if (iy < 0 || iy >= P.fH || ix < 0 || ix >= P.fW) continue;
          atomicAdd(gd + iz * HW + (long) iy * P.fW + ix, wz * gdep[j]);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
lift_bwd_dense_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                      const float* __restrict__ ys, const float* __restrict__ zs,
                      const float* __restrict__ gout, const uint64_t* __restrict__ hits,
                      float* __restrict__ gff) {
  const long V = (long) P.Z * P.Y * P.X;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * P.B) return;
  const int b = gid / V;
  const long vox = gid % V;
  const int x = vox % P.X, y = (vox / P.X) % P.Y, z = vox / ((long) P.X * P.Y);
  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long HW = (long) P.fH * P.fW, DHW = HW * P.D;
  const int nchunk = (P.C + 15) / 16;
  for (int n = 0; n < P.N; ++n) {
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project(P, mats + bn * 48, vx, vy, vz);
    if (!t.valid) continue;
    for (int c = 0; c < P.C; ++c) {
      const uint64_t cnt = hits[((long) b * V + vox) * nchunk + (c >> 4)];
      const float gs = gout[((long) b * P.C + c) * V + vox] /
                       ((float) ((cnt >> (4 * (c & 15))) & 15) + 1e-6f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = t.iz0 + (k >> 2), iy = t.iy0 + ((k >> 1) & 1), ix = t.ix0 + (k & 1);
        if (iz < 0 || iz >= P.D || iy < 0 || iy >= P.fH || ix < 0 || ix >= P.fW) continue;
        const float w = ((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0) *
                        ((k & 4) ? t.wz1 : t.wz0);
        atomicAdd(gff + (bn * P.C + c) * DHW + iz * HW + (long) iy * P.fW + ix, w * gs);
      }
    }
  }
}

__global__ void __launch_bounds__(256)
lift_indices_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                    const float* __restrict__ ys, const float* __restrict__ zs,
                    uint8_t* __restrict__ valid, int16_t* __restrict__ ix0,
                    int16_t* __restrict__ iy0, int16_t* __restrict__ iz0) {
  const long V = (long) P.Z * P.Y * P.X;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * P.B * P.N) return;
  const long bn = gid / V;
  const long vox = gid % V;
  const int x = vox % P.X, y = (vox / P.X) % P.Y, z = vox / ((long) P.X * P.Y);
  const LiftTap t = lift_project(P, mats + bn * 48, xs[x], ys[y], zs[z]);
  valid[gid] = t.valid ? 1 : 0;
  ix0[gid] = (int16_t) t.ix0;
  iy0[gid] = (int16_t) t.iy0;
  iz0[gid] = (int16_t) t.iz0;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int validate(const VampLiftDesc* d) {
  VAMP_REQUIRE(d != nullptr, "desc is NULL");
  VAMP_REQUIRE(d->B > 0 && d->N > 0 && d->C > 0, "B, N, C must be positive");
  VAMP_REQUIRE(d->D > 0 && d->fH > 0 && d->fW > 0, "D, fH, fW must be positive");
  VAMP_REQUIRE(d->Z > 0 && d->Y > 0 && d->X > 0, "Z, Y, X must be positive");
  VAMP_REQUIRE(d->X < 32768 && d->Y < 32768 && d->fW < 32768 && d->fH < 32768,
               "axis too long for int16 taps");
  VAMP_REQUIRE(d->N <= 15, "at most 15 cameras (4-bit hit counters)");
  VAMP_REQUIRE(d->in_dtype == VAMP_F32 || d->in_dtype == VAMP_BF16, "in_dtype");
  VAMP_REQUIRE(d->use_depth == 1 || d->D == 1, "use_depth == 0 requires D == 1");
  return VAMP_OK;
}

static bool fused_channels_ok(int C) { return C == 4 || C == 8 || (C % 16 == 0 && C <= 64); }

template <typename T>
static void launch_to_cl(const void* feat, float* out, int BN, int C, int HW, hipStream_t s) {
  dim3 grid((HW + 63) / 64, (C + 63) / 64, BN);
  VAMP_TIMED(kProfFeatCL, s, (feat_to_channel_last<T><<<grid, 256, 0, s>>>(static_cast<const T*>(feat), out, C, HW)));
}

struct LiftWs {
  float* feat_cl;     // [B*N, HW, C]
  float* gfeat_cl;    // [B*N, HW, C] (v1 backward only)
  void* cells;        // cell lists of the backward; vamp_lift_prepare fills their offsets
  size_t bytes;
};

static LiftWs carve(const VampLiftDesc* d, void* ws) {
  LiftWs w;
  const size_t n = align_up((size_t) d->B * d->N * d->fH * d->fW * d->C * sizeof(float), 256);
  w.feat_cl = static_cast<float*>(ws);
  w.gfeat_cl = reinterpret_cast<float*>(static_cast<char*>(ws) + n);
  // the cell lists live behind the copies: the forward must not disturb prepared offsets
  w.cells = static_cast<char*>(ws) + 2 * n;
  w.bytes = 2 * n + lift_bwd_cell_ws_bytes(d);
  return w;
}

// tile shape: lanes along x for coalesced stores
// A wave is a 16 x 4 patch of voxels, not a 64 x 1 row: the exact wave-level camera cull of
// lift_project<true> skips a camera only when none of the wave's voxels has it in front, and a
// 6.4 m x 1.6 m patch is on one side of most cameras where a 25.6 m row is not (cfg-B: 47 -> 36 us;
// 8 x 32 / 16 x 16 / 32 x 8 / 64 x 4 workgroup tiles: 36.5 / 35.7 / 38.2 / 47.1).
#ifndef VAMP_LIFT_TX
#define VAMP_LIFT_TX 16
#define VAMP_LIFT_TY 16
#endif
#define VAMP_LIFT_TILE VAMP_LIFT_TX, VAMP_LIFT_TY, 1

// `cells` != nullptr: the kernel emits the backward's pairs into that part of the workspace, between
// the zero fill of the cell counters and their scan (afterwards the workspace is what
// vamp_lift_prepare leaves: VAMP_LIFTBWD_CELLS_VALID)
template <typename T>
static int lift_forward_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                          const float* xs, const float* ys, const float* zs, const void* depth,
                          const float* feat_cl, float* out, uint64_t* hits, void* cells, bool cells_clean,
                          hipStream_t s) {
  constexpr int TX = VAMP_LIFT_TX, TY = VAMP_LIFT_TY, TZ = 1;
  dim3 grid((P.X + TX - 1) / TX, (P.Y + TY - 1) / TY, ((P.Z + TZ - 1) / TZ) * P.B);
  const T* dp = static_cast<const T*>(depth);
  LiftEmit E{};
  if (cells) {
    VAMP_REQUIRE(d->N <= 15, "at most 15 cameras");
    if (int e = launch_lift_cells_begin(d, cells, s, cells_clean)) return e;
    E = lift_emit_of(d, cells);
  }
#define VAMP_FWD(CH, EM)                                                                               \
  VAMP_TIMED(kProfLiftFwd, s, (lift_fwd_kernel<T, CH, VAMP_LIFT_TILE, EM><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, out, hits, E)))
  if (cells) {
    if (P.C == 4) VAMP_FWD(4, true); else if (P.C == 8) VAMP_FWD(8, true); else VAMP_FWD(16, true);
  } else {
    if (P.C == 4) VAMP_FWD(4, false); else if (P.C == 8) VAMP_FWD(8, false); else VAMP_FWD(16, false);
  }
#undef VAMP_FWD
  if (int e = check_launch("lift_fwd_kernel")) return e;
  return cells ? launch_lift_cells_end(d, cells, s) : VAMP_OK;
}

int launch_lift_cell_prepare(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, void* scratch, hipStream_t s) {
  const LiftParams P = to_params(d);
  if (int e = launch_lift_cells_begin(d, scratch, s)) return e;
  const LiftEmit E = lift_emit_of(d, scratch);
  constexpr int TX = VAMP_LIFT_TX, TY = VAMP_LIFT_TY;
  dim3 grid((P.X + TX - 1) / TX, (P.Y + TY - 1) / TY, P.Z * P.B);
  VAMP_TIMED(kProfLiftBwdCount, s, (lift_pairs_kernel<TX, TY><<<grid, TX * TY, 0, s>>>(P, mats, xs, ys, zs, E)));
  if (int e = check_launch("lift_pairs_kernel")) return e;
  return launch_lift_cells_end(d, scratch, s);
}

template <typename T>
static int lift_backward_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                           const float* xs, const float* ys, const float* zs, const void* depth,
                           const float* feat_cl, const float* gout, const uint64_t* hits,
                           float* gdepth, float* gfeat_cl, hipStream_t s) {
  constexpr int TX = VAMP_LIFT_TX, TY = VAMP_LIFT_TY, TZ = 1;
  dim3 grid((P.X + TX - 1) / TX, (P.Y + TY - 1) / TY, ((P.Z + TZ - 1) / TZ) * P.B);
  const T* dp = static_cast<const T*>(depth);
  if (P.C == 4)
    VAMP_TIMED(kProfLiftBwdV1, s, (lift_bwd_kernel<T, 4, VAMP_LIFT_TILE><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, gout, hits, gdepth, gfeat_cl)));
  else if (P.C == 8)
    VAMP_TIMED(kProfLiftBwdV1, s, (lift_bwd_kernel<T, 8, VAMP_LIFT_TILE><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, gout, hits, gdepth, gfeat_cl)));
  else
    VAMP_TIMED(kProfLiftBwdV1, s, (lift_bwd_kernel<T, 16, VAMP_LIFT_TILE><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, gout, hits, gdepth, gfeat_cl)));
  return check_launch("lift_bwd_kernel");
}

}  // namespace vamp

using namespace vamp;

extern "C" {

size_t vamp_lift_workspace_bytes(const VampLiftDesc* d) {
  if (!d) return 0;
  return carve(d, nullptr).bytes;
}

int vamp_lift_forward(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                      const float* zs, const void* depth, const void* feat, float* out,
                      uint64_t* hits, void* workspace, size_t workspace_bytes, void* stream) {
  return vamp_lift_forward_ex(d, mats, xs, ys, zs, depth, feat, out, hits, workspace, workspace_bytes, 0, stream);
}

int vamp_lift_forward_ex(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, float* out,
                         uint64_t* hits, void* workspace, size_t workspace_bytes, int flags, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && feat && out, "null pointer");
  VAMP_REQUIRE(depth || !d->use_depth, "depth is NULL");
  VAMP_REQUIRE(fused_channels_ok(d->C), "C must be 4, 8 or a multiple of 16 (<= 64)");
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const LiftParams P = to_params(d);
  const int BN = d->B * d->N, HW = d->fH * d->fW;
  if (d->in_dtype == VAMP_F32) launch_to_cl<float>(feat, w.feat_cl, BN, d->C, HW, s);
  else launch_to_cl<__hip_bfloat16>(feat, w.feat_cl, BN, d->C, HW, s);
  if (int e = check_launch("feat_to_channel_last")) return e;
  void* cells = (flags & VAMP_LIFTFWD_EMIT_PAIRS) ? w.cells : nullptr;
  const bool clean = (flags & VAMP_LIFTFWD_CELLS_CLEAN) != 0;
  if (d->in_dtype == VAMP_F32)
    return lift_forward_t<float>(d, P, mats, xs, ys, zs, depth, w.feat_cl, out, hits, cells, clean, s);
  return lift_forward_t<__hip_bfloat16>(d, P, mats, xs, ys, zs, depth, w.feat_cl, out, hits, cells, clean, s);
}

int vamp_lift_forward_logits(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                             const float* zs, const void* logits, int32_t logits_dtype, const float* feat,
                             float* depth_out, float* out, uint64_t* hits, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return vamp_lift_forward_logits_ex(d, mats, xs, ys, zs, logits, logits_dtype, feat, depth_out, out, hits,
                                     workspace, workspace_bytes, 0, stream);
}

int vamp_lift_forward_logits_ex(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                                const float* zs, const void* logits, int32_t logits_dtype, const float* feat,
                                float* depth_out, float* out, uint64_t* hits, void* workspace,
                                size_t workspace_bytes, int flags, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && logits && feat && depth_out && out, "null pointer");
  VAMP_REQUIRE(d->use_depth == 1, "the logits entry is the depth-distribution lift");
  VAMP_REQUIRE(d->in_dtype == VAMP_F32, "feat (and the depth distribution written here) are fp32");
  VAMP_REQUIRE(logits_dtype == VAMP_F32 || logits_dtype == VAMP_BF16, "logits_dtype");
  VAMP_REQUIRE(fused_channels_ok(d->C), "C must be 4, 8 or a multiple of 16 (<= 64)");
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const LiftParams P = to_params(d);
  const long BN = (long) d->B * d->N, HW = (long) d->fH * d->fW;
  const long sm_tiles = (HW + kPix - 1) / kPix;
  const int ptiles = (int) ((HW + 63) / 64), ctiles = (d->C + 63) / 64;
  const long n_sm = BN * sm_tiles, n_cl = BN * ptiles * ctiles;
  VAMP_REQUIRE(n_sm + n_cl < 0x7fffffffL, "too many tiles");
  const unsigned grid = (unsigned) (n_sm + n_cl);
#define VAMP_OPERANDS(TL, REG)                                                                           \
  VAMP_TIMED(kProfFeatCL, s, (lift_operands_kernel<TL, REG><<<grid, 256, 0, s>>>(                        \
      static_cast<const TL*>(logits), depth_out, d->D, HW, (int) sm_tiles, (unsigned) n_sm, feat,        \
      w.feat_cl, d->C, ptiles, ctiles)))
  const bool reg = d->D <= kSplit * kRegBins;
  if (logits_dtype == VAMP_F32) {
    if (reg) VAMP_OPERANDS(float, true); else VAMP_OPERANDS(float, false);
  } else {
    if (reg) VAMP_OPERANDS(__hip_bfloat16, true); else VAMP_OPERANDS(__hip_bfloat16, false);
  }
#undef VAMP_OPERANDS
  if (int e = check_launch("lift_operands_kernel")) return e;
  return lift_forward_t<float>(d, P, mats, xs, ys, zs, depth_out, w.feat_cl, out, hits,
                               (flags & VAMP_LIFTFWD_EMIT_PAIRS) ? w.cells : nullptr,
                               (flags & VAMP_LIFTFWD_CELLS_CLEAN) != 0, s);
}

int vamp_lift_prepare(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                      const float* zs, void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs, "null pointer");
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  return launch_lift_cell_prepare(d, mats, xs, ys, zs, w.cells, static_cast<hipStream_t>(stream));
}

int vamp_lift_backward(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                       const float* zs, const void* depth, const void* feat,
                       const float* grad_out, const uint64_t* hits, float* grad_depth,
                       float* grad_feat, void* workspace, size_t workspace_bytes, void* stream) {
  return vamp_lift_backward_ex(d, mats, xs, ys, zs, depth, feat, grad_out, hits, grad_depth, grad_feat,
                               workspace, workspace_bytes, 0, stream);
}

int vamp_lift_backward_ex(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                          const float* zs, const void* depth, const void* feat,
                          const float* grad_out, const uint64_t* hits, float* grad_depth,
                          float* grad_feat, void* workspace, size_t workspace_bytes, int flags,
                          void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && feat && grad_out && hits && grad_feat, "null pointer");
  VAMP_REQUIRE((depth && grad_depth) || !d->use_depth, "depth / grad_depth is NULL");
  VAMP_REQUIRE(fused_channels_ok(d->C), "C must be 4, 8 or a multiple of 16 (<= 64)");
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const LiftParams P = to_params(d);
  const int BN = d->B * d->N, HW = d->fH * d->fW;
  // default: cell list + one wave per pixel (lift_bwd_cell.hip), no float atomics.
  // VAMP_LIFTBWD_SPLAT selects the per-voxel atomic splat below, kept as an independent cross-check.
  if (!(flags & VAMP_LIFTBWD_SPLAT)) {
    // gather variants (same kernel, smaller record chunks: the tests run them to cross the chunk
    // boundaries at every size)
    const int wpp = (flags & VAMP_LIFTBWD_WPP1) ? 1 : (flags & VAMP_LIFTBWD_WPP4) ? 4
                    : (flags & VAMP_LIFTBWD_WPP16) ? 16 : 0;
    // VAMP_LIFTBWD_HALF_LO / _HI: one half of the images (needs prepared cell lists: both halves
    // read them, and whoever prepared them inside this call would race with the other half)
    const int half = (flags & VAMP_LIFTBWD_HALF_LO) ? 1 : ((flags & VAMP_LIFTBWD_HALF_HI) ? 2 : 0);
    VAMP_REQUIRE(half == 0 || (flags & VAMP_LIFTBWD_CELLS_VALID), "HALF_LO / HALF_HI need CELLS_VALID");
    return launch_lift_bwd_cell(d, mats, xs, ys, zs, depth, feat, grad_out, hits, grad_depth,
                                grad_feat, w.cells, (flags & VAMP_LIFTBWD_CELLS_VALID) != 0, wpp, half,
                                (flags & VAMP_LIFTBWD_LOGITS) != 0, s);
  }
  VAMP_REQUIRE(!(flags & VAMP_LIFTBWD_LOGITS), "VAMP_LIFTBWD_LOGITS is a feature of the default (cell-list) backward");
  if (d->in_dtype == VAMP_F32) launch_to_cl<float>(feat, w.feat_cl, BN, d->C, HW, s);
  else launch_to_cl<__hip_bfloat16>(feat, w.feat_cl, BN, d->C, HW, s);
  if (int e = check_launch("feat_to_channel_last")) return e;
  if (int ze = launch_zero(w.gfeat_cl, (size_t) BN * HW * d->C * sizeof(float), s)) return ze;
  if (d->use_depth)
    if (int ze = launch_zero(grad_depth, (size_t) BN * d->D * HW * sizeof(float), s)) return ze;
  int e = (d->in_dtype == VAMP_F32)
              ? lift_backward_t<float>(d, P, mats, xs, ys, zs, depth, w.feat_cl, grad_out, hits, grad_depth, w.gfeat_cl, s)
              : lift_backward_t<__hip_bfloat16>(d, P, mats, xs, ys, zs, depth, w.feat_cl, grad_out, hits, grad_depth, w.gfeat_cl, s);
  if (e) return e;
  dim3 grid((HW + 63) / 64, (d->C + 63) / 64, BN);
  VAMP_TIMED(kProfFeatCF, s, (feat_to_channel_first<<<grid, 256, 0, s>>>(w.gfeat_cl, grad_feat, d->C, HW)));
  return check_launch("feat_to_channel_first");
}

int vamp_lift_forward_dense(const VampLiftDesc* d, const float* mats, const float* xs,
                            const float* ys, const float* zs, const float* frustum_feats,
                            float* out, uint64_t* hits, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && frustum_feats && out, "null pointer");
  const LiftParams P = to_params(d);
  const long total = (long) d->B * d->Z * d->Y * d->X;
  hipStream_t s = static_cast<hipStream_t>(stream);
  VAMP_TIMED(kProfLiftFwdDense, s, (lift_fwd_dense_kernel<<<(unsigned) ((total + 255) / 256), 256, 0, s>>>(
      P, mats, xs, ys, zs, frustum_feats, out, hits)));
  return check_launch("lift_fwd_dense_kernel");
}

int vamp_lift_backward_dense(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, const float* grad_out,
                             const uint64_t* hits, float* grad_frustum_feats, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && grad_out && hits && grad_frustum_feats, "null pointer");
  const LiftParams P = to_params(d);
  const long total = (long) d->B * d->Z * d->Y * d->X;
  hipStream_t s = static_cast<hipStream_t>(stream);
  VAMP_TIMED(kProfLiftBwdDense, s, (lift_bwd_dense_kernel<<<(unsigned) ((total + 255) / 256), 256, 0, s>>>(
      P, mats, xs, ys, zs, grad_out, hits, grad_frustum_feats)));
  return check_launch("lift_bwd_dense_kernel");
}

int vamp_lift_indices(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                      const float* zs, uint8_t* valid, int16_t* ix0, int16_t* iy0, int16_t* iz0,
                      void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && valid && ix0 && iy0 && iz0, "null pointer");
  const LiftParams P = to_params(d);
  const long total = (long) d->B * d->N * d->Z * d->Y * d->X;
  lift_indices_kernel<<<(unsigned) ((total + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      P, mats, xs, ys, zs, valid, ix0, iy0, iz0);
  return check_launch("lift_indices_kernel");
}

}  // extern "C"
