// LIFT backward, v2: owner-computes with LDS accumulators.  Autograd of
// base_vampire2.py:507-514 (grid_sampler_3d backward + the camera mean) restated as:
//
//   one workgroup per (camera, TH x TW pixel tile).  It stages the tile's feat [TP][C] and
//   depth [D][TP] in LDS and walks the tile's viewing pyramid in depth slabs.  Per slab:
//     A. enumerate: the slab is bounded by an axis-aligned voxel box; every candidate voxel
//        is re-projected with the forward's bit-exact chain, and voxels whose four (h,w)
//        taps touch an owned pixel are pushed to an LDS queue;
//     B. accumulate: 16-lane groups pop queue entries, lane c owning channel c: one
//        (scattered) load of grad_out[c] per lane, then conflict-free ds_add_f32 of the
//        feat gradient, a 16-lane shuffle dot product for the depth gradient.
//   Each output element is finally stored exactly once: no global atomics, no memset, no
//   layout transposes.
//
// Memory-bound gather + LDS scatter; no MFMA.
#include "lift_common.hpp"

namespace vamp {

constexpr int TH = 8, TW = 8, TP = TH * TW;
constexpr int QCAP = 1024;            // hit-queue entries per workgroup pass
constexpr int GLN = 16;               // lanes per queue entry (= channels per pass)

struct Hit {
  int vox;                            // voxel index inside the sample
  float fx, fy, fz;                   // continuous tap coordinates, as the forward computed them
};

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

// gs[b][vox][c] = grad_out[b][c][vox] / (hit_count[b][vox][c] + 1e-6): the camera-mean factor of
// bv2:512-514 folded in, channel-last so that a voxel's C values are one contiguous run.
__global__ void __launch_bounds__(256)
lift_bwd_prep_kernel(const float* __restrict__ gout, const uint64_t* __restrict__ hits,
                     float* __restrict__ gs, int C, long V, int B, int hits_per_word,
                     const int* __restrict__ total, int cap) {
  if (total && *total <= cap) return;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * B) return;
  const long b = gid / V, vox = gid % V;
  const int wpc = (C + hits_per_word - 1) / hits_per_word;
  float* dst = gs + gid * C;
  for (int c0 = 0; c0 < C; c0 += 4) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q;
      const uint64_t cnt = hits[gid * wpc + c / hits_per_word];
      v[q] = gout[(b * C + c) * V + vox] / ((float) ((cnt >> (4 * (c % hits_per_word))) & 15) + 1e-6f);
    }
    *reinterpret_cast<float4*>(dst + c0) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
lift_bwd_tile_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ gmats,
                     const float* __restrict__ xs, const float* __restrict__ ys,
                     const float* __restrict__ zs, const T* __restrict__ depth,
                     const T* __restrict__ feat, const float* __restrict__ gs_cl,
                     float* __restrict__ gdepth, float* __restrict__ gfeat, float slab_len,
                     const int* __restrict__ total, int cap) {
  if (total && *total <= cap) return;          // fallback only: the binned lists fit
  extern __shared__ float lds[];
  __shared__ Hit queue[QCAP];
  __shared__ int qn;
  __shared__ float corner[8][3];              // ego position of the 4 tile-corner rays at d = 0, 1
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
  const float* m = mats + bn * 48;
  const float* gm = gmats + bn * 48;

  // ---- image-space extent of voxels that can touch an owned pixel ----
  // tap coordinate fx = (u / u_div) * fW - 0.5 ; owned taps need floor(fx) in [w0-1, w0+TW-1]
  const float u_lo = fmaxf(-0.5f, ((float) w0 - 0.5f - 0.01f) * P.u_div / (float) P.fW);
  const float u_hi = fminf(P.u_max, ((float) (w0 + TW) + 0.5f + 0.01f) * P.u_div / (float) P.fW);
  const float v_lo = fmaxf(-0.5f, ((float) h0 - 0.5f - 0.01f) * P.v_div / (float) P.fH);
  const float v_hi = fminf(P.v_max, ((float) (h0 + TH) + 0.5f + 0.01f) * P.v_div / (float) P.fH);

  // ---- stage the tile; threads 0..7 unproject the corner rays (affine in depth) ----
  if (tid < 8) {
    float x, y, z;
    unproject(gm, (tid & 1) ? u_hi : u_lo, (tid & 2) ? v_hi : v_lo, (tid & 4) ? 1.0f : 0.0f, x, y, z);
    corner[tid][0] = x; corner[tid][1] = y; corner[tid][2] = z;
  }
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
  if (tid == 0) qn = 0;
  __syncthreads();

  const float z_near = P.use_depth ? P.d_lo : 1e-3f;
  // without a depth distribution the lift accepts every z > 0: walk out past the grid diagonal
  const float z_far = P.use_depth ? P.d_hi
                                  : 2.0f * (fabsf(xs[P.X - 1] - xs[0]) + fabsf(ys[P.Y - 1] - ys[0]) +
                                            fabsf(zs[P.Z - 1] - zs[0]) + 1.0f);
  const float x0g = xs[0], y0g = ys[0], z0g = zs[0];
  const float idx_ = (P.X > 1) ? (float) (P.X - 1) / (xs[P.X - 1] - x0g) : 1.f;
  const float idy_ = (P.Y > 1) ? (float) (P.Y - 1) / (ys[P.Y - 1] - y0g) : 1.f;
  const float idz_ = (P.Z > 1) ? (float) (P.Z - 1) / (zs[P.Z - 1] - z0g) : 1.f;
  const int nslab = max(1, (int) ceilf((z_far - z_near) / slab_len));
  const int grp = tid / GLN, gl = tid % GLN;

  // ---- B. accumulate the queued hits: one 16-lane group per entry, lane = channel ----
  auto drain = [&]() {
    const int n_hit = qn;
    for (int e = grp; e < n_hit; e += 256 / GLN) {
      const Hit hrec = queue[e];
      const float flx = floorf(hrec.fx), fly = floorf(hrec.fy), flz = floorf(hrec.fz);
      const int ix0 = (int) flx, iy0 = (int) fly, iz0 = (int) flz;
      const float wx1 = hrec.fx - flx, wx0 = (flx + 1.0f) - hrec.fx;
      const float wy1 = hrec.fy - fly, wy0 = (fly + 1.0f) - hrec.fy;
      const float wz1 = hrec.fz - flz, wz0 = (flz + 1.0f) - hrec.fz;
      const float wj[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
      int pj[4];
      float dep[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = iy0 + (j >> 1), ix = ix0 + (j & 1);
        const bool own = iy >= h0 && iy < h0 + TH && ix >= w0 && ix < w0 + TW && iy < P.fH && ix < P.fW;
        pj[j] = own ? (iy - h0) * TW + (ix - w0) : -1;
        float dv = 0.f;
        if (own) {
          if (P.use_depth) {
            if (iz0 >= 0 && iz0 < P.D) dv += wz0 * dt[iz0 * TP + pj[j]];
            if (iz0 + 1 >= 0 && iz0 + 1 < P.D) dv += wz1 * dt[(iz0 + 1) * TP + pj[j]];
          } else {
            dv = (iz0 == 0 ? wz0 : 0.f) + (iz0 == -1 ? wz1 : 0.f);
          }
        }
        dep[j] = dv;
      }
      float dot[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c0 = 0; c0 < C; c0 += GLN) {
        const int c = c0 + gl;
        const float gs = (c < C) ? gs_cl[((long) b * V + hrec.vox) * C + c] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (pj[j] < 0 || c >= C) continue;
          atomicAdd(gf + pj[j] * CS + c, wj[j] * dep[j] * gs);
          dot[j] = __builtin_fmaf(ft[pj[j] * CS + c], gs, dot[j]);
        }
      }
      if (P.use_depth) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (pj[j] < 0) continue;                         // uniform over the group
          float dj = dot[j];
#pragma unroll
          for (int o = GLN >> 1; o > 0; o >>= 1) dj += __shfl_xor(dj, o, GLN);
          const float gdj = wj[j] * dj;
          if (gl == 0 && iz0 >= 0 && iz0 < P.D) atomicAdd(gd + iz0 * TP + pj[j], wz0 * gdj);
          if (gl == 1 && iz0 + 1 >= 0 && iz0 + 1 < P.D) atomicAdd(gd + (iz0 + 1) * TP + pj[j], wz1 * gdj);
        }
      }
    }
    __syncthreads();
    if (tid == 0) qn = 0;
    __syncthreads();
  };

  for (int sl = 0; sl < nslab; ++sl) {
    const float s_lo = z_near + (float) sl * slab_len;
    const float s_hi = (sl == nslab - 1) ? z_far : z_near + (float) (sl + 1) * slab_len;
    // voxel-index box of the pyramid slab: corners are o_k + d * (r_k - o_k), d in {s_lo, s_hi}
    float fx0 = 3e38f, fx1 = -3e38f, fy0 = 3e38f, fy1 = -3e38f, fz0 = 3e38f, fz1 = -3e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float dd = e ? s_hi : s_lo;
        const float x = corner[k][0] + dd * (corner[k + 4][0] - corner[k][0]);
        const float y = corner[k][1] + dd * (corner[k + 4][1] - corner[k][1]);
        const float z = corner[k][2] + dd * (corner[k + 4][2] - corner[k][2]);
        const float gx = (x - x0g) * idx_, gy = (y - y0g) * idy_, gz = (z - z0g) * idz_;
        fx0 = fminf(fx0, gx); fx1 = fmaxf(fx1, gx);
        fy0 = fminf(fy0, gy); fy1 = fmaxf(fy1, gy);
        fz0 = fminf(fz0, gz); fz1 = fmaxf(fz1, gz);
      }
    }
    if (!(fx0 <= fx1) || !(fy0 <= fy1) || !(fz0 <= fz1)) continue;      // NaN guard (uniform)
    // one voxel of slack on each side covers the fp32 slack of the inverted matrices
    const int ix_lo = max(0, (int) floorf(fx0) - 1), ix_hi = min(P.X - 1, (int) ceilf(fx1) + 1);
    const int iy_lo = max(0, (int) floorf(fy0) - 1), iy_hi = min(P.Y - 1, (int) ceilf(fy1) + 1);
    const int iz_lo = max(0, (int) floorf(fz0) - 1), iz_hi = min(P.Z - 1, (int) ceilf(fz1) + 1);
    if (ix_lo > ix_hi || iy_lo > iy_hi || iz_lo > iz_hi) continue;      // uniform
    const int nx = ix_hi - ix_lo + 1, ny = iy_hi - iy_lo + 1;
    const int count = nx * ny * (iz_hi - iz_lo + 1);
    const float inv_nx = 1.0f / (float) nx, inv_ny = 1.0f / (float) ny;

    for (int base = 0; base < count; base += 256) {
      // ---- A. enumerate one pass of candidates into the queue ----
      const int idx = base + tid;
      if (idx < count) {
        int r = (int) (((float) idx + 0.5f) * inv_nx);
        int xi = idx - r * nx;
        if (xi < 0) { xi += nx; --r; } else if (xi >= nx) { xi -= nx; ++r; }
        int zi = (int) (((float) r + 0.5f) * inv_ny);
        int yi = r - zi * ny;
        if (yi < 0) { yi += ny; --zi; } else if (yi >= ny) { yi -= ny; ++zi; }
        const int x = ix_lo + xi, y = iy_lo + yi, z = iz_lo + zi;
        const LiftTap t = lift_project(P, m, xs[x], ys[y], zs[z]);
        // each voxel belongs to exactly one slab (half-open in projected depth)
        const bool in_slab = (t.zz >= s_lo) && (t.zz < s_hi || sl == nslab - 1);
        const bool touch = !(t.ix0 + 1 < w0 || t.ix0 >= w0 + TW || t.iy0 + 1 < h0 || t.iy0 >= h0 + TH);
        if (t.valid && in_slab && touch) {
          const int slot = atomicAdd(&qn, 1);   // a pass adds at most 256 entries
          Hit hrec;
          hrec.vox = (int) (((long) z * P.Y + y) * P.X + x);
          hrec.fx = t.fx; hrec.fy = t.fy; hrec.fz = t.fz;
          queue[slot] = hrec;
        }
      }
      __syncthreads();
      if (qn + 256 > QCAP) drain();            // uniform: qn is read after the barrier
    }
  }
  __syncthreads();
  drain();

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

template <typename T>
static int launch_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                    const float* gmats, const float* xs, const float* ys, const float* zs,
                    const void* depth, const void* feat, const float* gs_cl, float* gdepth,
                    float* gfeat, float slab_len, const int* total, int cap, hipStream_t s) {
  const int tiles = ((P.fW + TW - 1) / TW) * ((P.fH + TH - 1) / TH);
  const unsigned grid = (unsigned) ((long) d->B * d->N * tiles);
  const int Dd = d->use_depth ? d->D : 0;
  const size_t lds = ((size_t) 2 * TP * (d->C + 1) + (size_t) 2 * Dd * TP) * sizeof(float);
  if (lds + sizeof(Hit) * QCAP > 150 * 1024) return fail(VAMP_EINVAL, "%s: D too large for the LDS tile", __func__);
  auto k = lift_bwd_tile_kernel<T>;
  if (lds + sizeof(Hit) * QCAP + 256 > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int) lds) != hipSuccess)
    return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);
  VAMP_TIMED(kProfLiftBwd, s, (k<<<grid, 256, lds, s>>>(
      P, mats, gmats, xs, ys, zs, static_cast<const T*>(depth), static_cast<const T*>(feat), gs_cl,
      gdepth, gfeat, slab_len, total, cap)));
  return check_launch("lift_bwd_tile_kernel");
}

size_t lift_bwd_tile_ws_bytes(const VampLiftDesc* d) {
  return align_up((size_t) d->B * d->N * 48 * sizeof(float), 256) +
         align_up((size_t) d->B * d->Z * d->Y * d->X * d->C * sizeof(float), 256);
}

// scratch: the inverted matrices + the channel-last, mean-scaled upstream gradient
int launch_lift_bwd_tile(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, const float* gout,
                         const uint64_t* hits, float* gdepth, float* gfeat, void* scratch,
                         const int* total, int cap, hipStream_t s) {
  const LiftParams P = to_params(d);
  float* gmats = static_cast<float*>(scratch);
  float* gs_cl = reinterpret_cast<float*>(static_cast<char*>(scratch) +
                                          align_up((size_t) d->B * d->N * 48 * sizeof(float), 256));
  if (int e = launch_invert_mats(mats, gmats, d->B * d->N * 3, true, s)) return e;
  {
    const long V = (long) d->Z * d->Y * d->X;
    // the forward packs per-channel hit counters 16 to a word (4 or 8 when C is 4 or 8)
    const int hits_per_word = d->C < 16 ? d->C : 16;
    VAMP_TIMED(kProfLiftBwdPrep, s, (lift_bwd_prep_kernel<<<(unsigned) ((V * d->B + 255) / 256), 256, 0, s>>>(
        gout, hits, gs_cl, d->C, V, d->B, hits_per_word, total, cap)));
    if (int e = check_launch("lift_bwd_prep_kernel")) return e;
  }
  const char* sl = getenv("VAMP_LIFT_SLAB");
  const float slab_len = sl ? (float) atof(sl) : 3.2f;
  if (d->in_dtype == VAMP_F32)
    return launch_t<float>(d, P, mats, gmats, xs, ys, zs, depth, feat, gs_cl, gdepth, gfeat, slab_len, total, cap, s);
  return launch_t<__hip_bfloat16>(d, P, mats, gmats, xs, ys, zs, depth, feat, gs_cl, gdepth, gfeat, slab_len, total, cap, s);
}

}  // namespace vamp
