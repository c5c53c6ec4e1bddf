// LIFT backward, v2: owner-computes with LDS accumulators.  Autograd of
// base_vampire2.py:507-514 (grid_sampler_3d backward + the camera mean) restated as:
//
//   one workgroup per (camera, TH x TW pixel tile).  It stages the tile's feat [TH*TW][C]
//   and depth [D][TH*TW] in LDS, walks the tile's viewing pyramid in depth slabs, bounds
//   each slab by an axis-aligned voxel box, re-projects every candidate voxel with the
//   forward's bit-exact chain, and for voxels whose four (h,w) taps touch a pixel it owns
//   accumulates into LDS copies of grad_feat / grad_depth (ds_add_f32).  Each output
//   element is then stored exactly once: no global atomics, no memset, no transposes.
//
// Memory-bound gather + LDS scatter; no MFMA.
#include "lift_common.hpp"

namespace vamp {

constexpr int TH = 8, TW = 8, TP = TH * TW;

// frustum (u, v, depth) -> ego, inverse of lift_project's chain; gm = inverses of the lift
// matrices in reverse order: [inv(ida), inv(intrin inv(s2e)), bda]
__device__ __forceinline__ void unproject(const float* __restrict__ gm, float u, float v, float dd,
                                          float& x, float& y, float& z) {
  Vec4 p{u, v, dd, 1.0f};
  p = matvec(gm, p);
  p.x = p.x * p.z;
  p.y = p.y * p.z;
  p = matvec(gm + 16, p);
  p = matvec(gm + 32, p);
  x = p.x; y = p.y; z = p.z;
}

template <typename T, int CH>
__global__ void __launch_bounds__(256)
lift_bwd_tile_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ gmats,
                     const float* __restrict__ xs, const float* __restrict__ ys,
                     const float* __restrict__ zs, const T* __restrict__ depth,
                     const T* __restrict__ feat, const float* __restrict__ gout,
                     const uint64_t* __restrict__ hits, float* __restrict__ gdepth,
                     float* __restrict__ gfeat, float slab_len) {
  extern __shared__ float lds[];
  const int C = P.C, D = P.use_depth ? P.D : 0;
  const int CS = C + 1;                       // odd stride for the [pixel][channel] arrays
  float* ft = lds;                            // [TP][CS]   feat tile
  float* gf = ft + TP * CS;                   // [TP][CS]   grad_feat accumulators
  float* dt = gf + TP * CS;                   // [D][TP]    depth tile
  float* gd = dt + D * TP;                    // [D][TP]    grad_depth accumulators

  const int tid = threadIdx.x;
  const int tiles_w = (P.fW + TW - 1) / TW, tiles_h = (P.fH + TH - 1) / TH;
  const int tile = blockIdx.x % (tiles_w * tiles_h);
  const long bn = blockIdx.x / (tiles_w * tiles_h);
  const int b = bn / P.N;
  const int w0 = (tile % tiles_w) * TW, h0 = (tile / tiles_w) * TH;
  const long HW = (long) P.fH * P.fW;
  const long V = (long) P.Z * P.Y * P.X;

  // ---- stage the tile ----
  for (int e = tid; e < TP * C; e += 256) {
    const int c = e / TP, p = e % TP;
    const int h = h0 + p / TW, w = w0 + p % TW;
    ft[p * CS + c] = (h < P.fH && w < P.fW) ? ldf(feat, (bn * C + c) * HW + (long) h * P.fW + w) : 0.f;
  }
  for (int e = tid; e < TP * CS; e += 256) gf[e] = 0.f;
  for (int e = tid; e < D * TP; e += 256) {
    const int dz = e / TP, p = e % TP;
    const int h = h0 + p / TW, w = w0 + p % TW;
    dt[e] = (h < P.fH && w < P.fW) ? ldf(depth, (bn * P.D + dz) * HW + (long) h * P.fW + w) : 0.f;
    gd[e] = 0.f;
  }
  __syncthreads();

  // ---- image-space extent of voxels that can touch an owned pixel ----
  // tap coordinate fx = (u / u_div) * fW - 0.5 ; owned taps need floor(fx) in [w0-1, w0+TW-1]
  const float u_lo = fmaxf(-0.5f, ((float) w0 - 0.5f - 0.01f) * P.u_div / (float) P.fW);
  const float u_hi = fminf(P.u_max, ((float) (w0 + TW) + 0.5f + 0.01f) * P.u_div / (float) P.fW);
  const float v_lo = fmaxf(-0.5f, ((float) h0 - 0.5f - 0.01f) * P.v_div / (float) P.fH);
  const float v_hi = fminf(P.v_max, ((float) (h0 + TH) + 0.5f + 0.01f) * P.v_div / (float) P.fH);
  const float z_near = P.use_depth ? P.d_lo : 1e-3f;
  // without a depth distribution the lift accepts every z > 0: walk out to the grid diagonal
  const float z_far = P.use_depth ? P.d_hi
                                  : 2.0f * (fabsf(xs[P.X - 1] - xs[0]) + fabsf(ys[P.Y - 1] - ys[0]) +
                                            fabsf(zs[P.Z - 1] - zs[0]) + 1.0f);
  const float dx = (P.X > 1) ? (xs[P.X - 1] - xs[0]) / (float) (P.X - 1) : 1.f;
  const float dy = (P.Y > 1) ? (ys[P.Y - 1] - ys[0]) / (float) (P.Y - 1) : 1.f;
  const float dzv = (P.Z > 1) ? (zs[P.Z - 1] - zs[0]) / (float) (P.Z - 1) : 1.f;
  const float* m = mats + bn * 48;
  const float* gm = gmats + bn * 48;
  const int nchunk = C / CH;
  const int nslab = max(1, (int) ceilf((z_far - z_near) / slab_len));

  for (int sl = 0; sl < nslab; ++sl) {
    const float s_lo = z_near + (float) sl * slab_len;
    const float s_hi = (sl == nslab - 1) ? z_far : z_near + (float) (sl + 1) * slab_len;
    // voxel-index box of the pyramid slab (8 corners), one voxel of slack
    float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f, bz0 = 3e38f, bz1 = -3e38f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float x, y, z;
      unproject(gm, (k & 1) ? u_hi : u_lo, (k & 2) ? v_hi : v_lo, (k & 4) ? s_hi : s_lo, x, y, z);
      bx0 = fminf(bx0, x); bx1 = fmaxf(bx1, x);
      by0 = fminf(by0, y); by1 = fmaxf(by1, y);
      bz0 = fminf(bz0, z); bz1 = fmaxf(bz1, z);
    }
    if (!(bx0 <= bx1)) continue;                                  // NaN guard
    const int ix_lo = max(0, (int) floorf(fminf((bx0 - xs[0]) / dx, (bx1 - xs[0]) / dx)) - 1);
    const int ix_hi = min(P.X - 1, (int) ceilf(fmaxf((bx0 - xs[0]) / dx, (bx1 - xs[0]) / dx)) + 1);
    const int iy_lo = max(0, (int) floorf(fminf((by0 - ys[0]) / dy, (by1 - ys[0]) / dy)) - 1);
    const int iy_hi = min(P.Y - 1, (int) ceilf(fmaxf((by0 - ys[0]) / dy, (by1 - ys[0]) / dy)) + 1);
    const int iz_lo = max(0, (int) floorf(fminf((bz0 - zs[0]) / dzv, (bz1 - zs[0]) / dzv)) - 1);
    const int iz_hi = min(P.Z - 1, (int) ceilf(fmaxf((bz0 - zs[0]) / dzv, (bz1 - zs[0]) / dzv)) + 1);
    if (ix_lo > ix_hi || iy_lo > iy_hi || iz_lo > iz_hi) continue;
    const int nx = ix_hi - ix_lo + 1, ny = iy_hi - iy_lo + 1;
    const int count = nx * ny * (iz_hi - iz_lo + 1);

    for (int idx = tid; idx < count; idx += 256) {
      const int x = ix_lo + idx % nx;
      const int r = idx / nx;
      const int y = iy_lo + r % ny;
      const int z = iz_lo + r / ny;
      const LiftTap t = lift_project(P, m, xs[x], ys[y], zs[z]);
      if (!t.valid) continue;
      // each voxel belongs to exactly one slab (half-open in projected depth)
      if (!(t.zz >= s_lo) || (t.zz >= s_hi && sl != nslab - 1)) continue;
      if (t.ix0 + 1 < w0 || t.ix0 >= w0 + TW || t.iy0 + 1 < h0 || t.iy0 >= h0 + TH) continue;
      const long vox = ((long) z * P.Y + y) * P.X + x;
      const float wj[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
      int pj[4];                    // owned-pixel slot of each (h,w) tap, or -1
      float dep[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
        const bool own = iy >= h0 && iy < h0 + TH && ix >= w0 && ix < w0 + TW && iy < P.fH && ix < P.fW;
        pj[j] = own ? (iy - h0) * TW + (ix - w0) : -1;
        float dv = 0.f;
        if (own) {
          if (P.use_depth) {
            if (t.iz0 >= 0 && t.iz0 < P.D) dv += t.wz0 * dt[t.iz0 * TP + pj[j]];
            if (t.iz0 + 1 >= 0 && t.iz0 + 1 < P.D) dv += t.wz1 * dt[(t.iz0 + 1) * TP + pj[j]];
          } else {
            dv = (t.iz0 == 0 ? t.wz0 : 0.f) + (t.iz0 == -1 ? t.wz1 : 0.f);
          }
        }
        dep[j] = dv;
      }
      float dot[4] = {0.f, 0.f, 0.f, 0.f};
      for (int chunk = 0; chunk < nchunk; ++chunk) {
        const uint64_t cnt = hits[((long) b * V + vox) * nchunk + chunk];
        float gs[CH];
        const float* g = gout + ((long) b * C + chunk * CH) * V + vox;
#pragma unroll
        for (int c = 0; c < CH; ++c)
          gs[c] = g[(long) c * V] / ((float) ((cnt >> (4 * c)) & 15) + 1e-6f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (pj[j] < 0) continue;
          const float wd = wj[j] * dep[j];
          float* gfp = gf + pj[j] * CS + chunk * CH;
          const float* ftp = ft + pj[j] * CS + chunk * CH;
          float dj = dot[j];
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            atomicAdd(gfp + c, wd * gs[c]);
            dj = __builtin_fmaf(ftp[c], gs[c], dj);
          }
          dot[j] = dj;
        }
      }
      if (P.use_depth) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (pj[j] < 0) continue;
          const float gdj = wj[j] * dot[j];
          if (t.iz0 >= 0 && t.iz0 < P.D) atomicAdd(gd + t.iz0 * TP + pj[j], t.wz0 * gdj);
          if (t.iz0 + 1 >= 0 && t.iz0 + 1 < P.D) atomicAdd(gd + (t.iz0 + 1) * TP + pj[j], t.wz1 * gdj);
        }
      }
    }
  }
  __syncthreads();

  // ---- store the tile ----
  for (int e = tid; e < TP * C; e += 256) {
    const int c = e / TP, p = e % TP;
    const int h = h0 + p / TW, w = w0 + p % TW;
    if (h < P.fH && w < P.fW) gfeat[(bn * C + c) * HW + (long) h * P.fW + w] = gf[p * CS + c];
  }
  if (P.use_depth && gdepth)
    for (int e = tid; e < D * TP; e += 256) {
      const int dz = e / TP, p = e % TP;
      const int h = h0 + p / TW, w = w0 + p % TW;
      if (h < P.fH && w < P.fW) gdepth[(bn * P.D + dz) * HW + (long) h * P.fW + w] = gd[e];
    }
}

size_t lift_bwd_tile_ws_bytes(const VampLiftDesc* d) {
  return align_up((size_t) d->B * d->N * 48 * sizeof(float), 256);
}

template <typename T>
static int launch_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                    const float* gmats, const float* xs, const float* ys, const float* zs,
                    const void* depth, const void* feat, const float* gout, const uint64_t* hits,
                    float* gdepth, float* gfeat, float slab_len, hipStream_t s) {
  const int tiles = ((P.fW + TW - 1) / TW) * ((P.fH + TH - 1) / TH);
  const unsigned grid = (unsigned) ((long) d->B * d->N * tiles);
  const int Dd = d->use_depth ? d->D : 0;
  const size_t lds = ((size_t) 2 * TP * (d->C + 1) + (size_t) 2 * Dd * TP) * sizeof(float);
  if (lds > 150 * 1024) return fail(VAMP_EINVAL, "%s: D too large for the LDS tile", __func__);
#define VAMP_LBT(CH)                                                                              \
  do {                                                                                            \
    auto k = lift_bwd_tile_kernel<T, CH>;                                                         \
    if (lds > 64 * 1024 &&                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) \
      return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);                           \
    VAMP_TIMED(kProfLiftBwd, s, (k<<<grid, 256, lds, s>>>(                                        \
        P, mats, gmats, xs, ys, zs, static_cast<const T*>(depth), static_cast<const T*>(feat),    \
        gout, hits, gdepth, gfeat, slab_len)));                                                   \
  } while (0)
  if (P.C == 4) VAMP_LBT(4); else if (P.C == 8) VAMP_LBT(8); else VAMP_LBT(16);
#undef VAMP_LBT
  return check_launch("lift_bwd_tile_kernel");
}

// scratch: room for the inverted matrices
int launch_lift_bwd_tile(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, const float* gout,
                         const uint64_t* hits, float* gdepth, float* gfeat, void* scratch,
                         hipStream_t s) {
  const LiftParams P = to_params(d);
  float* gmats = static_cast<float*>(scratch);
  if (int e = launch_invert_mats(mats, gmats, d->B * d->N * 3, true, s)) return e;
  const char* sl = getenv("VAMP_LIFT_SLAB");
  const float slab_len = sl ? (float) atof(sl) : 3.2f;
  if (d->in_dtype == VAMP_F32)
    return launch_t<float>(d, P, mats, gmats, xs, ys, zs, depth, feat, gout, hits, gdepth, gfeat, slab_len, s);
  return launch_t<__hip_bfloat16>(d, P, mats, gmats, xs, ys, zs, depth, feat, gout, hits, gdepth, gfeat, slab_len, s);
}

}  // namespace vamp
