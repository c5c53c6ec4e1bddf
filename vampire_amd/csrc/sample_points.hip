// Trilinear resampling of a [B, C, Z, Y, X] volume at explicit ego-frame points: the occupancy
// and lidar-point queries of base_vampire2.py:576-609.
//
//   occ_logits  = grid_sample(semantic_logits, bda-rotated occ grid, padding_mode='border')   bv2:603
//   occ_density = grid_sample(density(density_feature), same grid)                            bv2:604
//   pts_logits  = grid_sample(semantic_logits[i], lidar points, padding_mode='border')        bv2:590
//   pts_sdf     = grid_sample(density_feature[i], lidar points) * inside-mask                 bv2:594-595
//
// all with align_corners=True and coordinates normalised by the seg bounds (bv2:581-586,
// 599-602).  Forward: thread per point, lanes along the point index (the occ grid is x-fastest,
// so neighbouring lanes read neighbouring voxels), channels looped; the density activation is
// applied to the eight taps on the fly.  Backward: the same cell-list scheme as the render
// backward (cell_list.hpp): rank -> scan -> fill -> per-voxel gather, no float atomics, with
// the activation's derivative applied once per voxel.
#include "cell_list.hpp"

#include <algorithm>

namespace vamp {

constexpr int kPtsHeavy = 256;
constexpr int PGL = 8;                 // lanes per voxel in the gather
constexpr int PVPB = 256 / PGL;

struct SampleParams {
  int B, C, Z, Y, X, Pb;
  float lo[3], span[3];
  int border, mask_outside, activation, channel_last;
  int density_mode;
  float sdf_bias, beta_min;
  int n0, n12;             // lattice hint: n0 points along axis 0, n1 * n2 per axis-0 step (0 = none)
};

static SampleParams to_params(const VampSampleDesc* d, int points_per_sample) {
  SampleParams p;
  p.B = d->B; p.C = d->C; p.Z = d->Z; p.Y = d->Y; p.X = d->X; p.Pb = points_per_sample;
  for (int i = 0; i < 3; ++i) { p.lo[i] = d->lo[i]; p.span[i] = d->span[i]; }
  p.border = d->padding == VAMP_PAD_BORDER;
  p.mask_outside = d->mask_outside;
  p.activation = d->activation;
  p.channel_last = d->channel_last_out;
  p.density_mode = d->density_mode; p.sdf_bias = d->sdf_bias; p.beta_min = d->beta_min;
  const long n12 = (long) d->lattice[1] * d->lattice[2];
  const bool lat = d->lattice[0] > 0 && n12 > 0 && (long) d->lattice[0] * n12 == points_per_sample;
  p.n0 = lat ? d->lattice[0] : 0;
  p.n12 = lat ? (int) n12 : 0;
  return p;
}

static int validate(const VampSampleDesc* d, long n) {
  VAMP_REQUIRE(d != nullptr, "descriptor is NULL");
  VAMP_REQUIRE(d->B > 0 && d->C > 0 && d->C <= 32, "B > 0, 0 < C <= 32");
  VAMP_REQUIRE(d->Z > 0 && d->Y > 0 && d->X > 0 && d->X < 2047 && d->Y < 2047 && d->Z < 511,
               "volume extents (X, Y < 2047, Z < 511: packed tap keys)");
  VAMP_REQUIRE(d->span[0] != 0.f && d->span[1] != 0.f && d->span[2] != 0.f, "zero span");
  VAMP_REQUIRE(d->padding == VAMP_PAD_ZEROS || d->padding == VAMP_PAD_BORDER, "padding mode");
  VAMP_REQUIRE(d->in_dtype == VAMP_F32 || d->in_dtype == VAMP_BF16, "in_dtype");
  VAMP_REQUIRE(!d->activation || d->C == 1, "the density activation applies to a 1-channel volume");
  VAMP_REQUIRE(n >= 0 && (long) d->B * n < 0x7fffffffL, "too many points");
  return VAMP_OK;
}

// density parameters when the activation is on (beta is only read then)
__device__ __forceinline__ DensityParams pts_density(const SampleParams& P, const float* __restrict__ beta_raw) {
  if (P.activation) return load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  DensityParams dp;
  dp.mode = VAMP_DENSITY_SIGMOID; dp.beta = 1.f; dp.ib = 1.f; dp.bias = 0.f;
  return dp;
}

// continuous tap coordinates of a point (aten: normalise by the bounds, unnormalise with
// align_corners=True, clip for 'border'), and the reference's inside mask
struct PointTap {
  float fx, fy, fz;
  bool inside;       // all(-1 <= n <= 1)        (bv2:587-589)
};
__device__ __forceinline__ PointTap point_tap(const SampleParams& P, const float* __restrict__ pt) {
  const float gx = ((pt[0] - P.lo[0]) / P.span[0]) * 2.0f - 1.0f;
  const float gy = ((pt[1] - P.lo[1]) / P.span[1]) * 2.0f - 1.0f;
  const float gz = ((pt[2] - P.lo[2]) / P.span[2]) * 2.0f - 1.0f;
  PointTap t;
  t.inside = gx >= -1.0f && gx <= 1.0f && gy >= -1.0f && gy <= 1.0f && gz >= -1.0f && gz <= 1.0f;
  t.fx = ((gx + 1.0f) / 2.0f) * (float) (P.X - 1);
  t.fy = ((gy + 1.0f) / 2.0f) * (float) (P.Y - 1);
  t.fz = ((gz + 1.0f) / 2.0f) * (float) (P.Z - 1);
  if (P.border) {                     // clip_coordinates: min(size - 1, max(coord, 0)); nan -> 0
    t.fx = fminf((float) (P.X - 1), fmaxf(t.fx, 0.f));
    t.fy = fminf((float) (P.Y - 1), fmaxf(t.fy, 0.f));
    t.fz = fminf((float) (P.Z - 1), fmaxf(t.fz, 0.f));
  }
  return t;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
sample_points_fwd_kernel(SampleParams P, const T* __restrict__ vol, const float* __restrict__ beta_raw,
                         const float* __restrict__ pts, float* __restrict__ out) {
  int p = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= P.Pb) return;
  // lattice hint: thread t takes point (t % n0, t / n0) of the [n0][n1 * n2] lattice, i.e.
  // consecutive lanes walk the lattice axis that runs along the volume's x
  if (P.n0 > 0) p = (p % P.n0) * P.n12 + p / P.n0;
  const PointTap t = point_tap(P, pts + ((long) b * P.Pb + p) * 3);
  const float flx = floorf(t.fx), fly = floorf(t.fy), flz = floorf(t.fz);
  // out-of-range coordinates (zeros mode) are clamped before the int conversion; their taps
  // get zero weight below
  const int ix0 = (int) fminf(fmaxf(flx, -2.f), (float) P.X), iy0 = (int) fminf(fmaxf(fly, -2.f), (float) P.Y),
            iz0 = (int) fminf(fmaxf(flz, -2.f), (float) P.Z);
  const float wx1 = t.fx - flx, wx0 = (flx + 1.0f) - t.fx;
  const float wy1 = t.fy - fly, wy0 = (fly + 1.0f) - t.fy;
  const float wz1 = t.fz - flz, wz0 = (flz + 1.0f) - t.fz;
  const bool finite = (t.fx == t.fx) && (t.fy == t.fy) && (t.fz == t.fz) &&
                      fabsf(t.fx) < 1e9f && fabsf(t.fy) < 1e9f && fabsf(t.fz) < 1e9f;
  float wt[8];
  long at[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int iz = iz0 + (k >> 2), iy = iy0 + ((k >> 1) & 1), ix = ix0 + (k & 1);
    const bool in = finite && iz >= 0 && iz < P.Z && iy >= 0 && iy < P.Y && ix >= 0 && ix < P.X;
    wt[k] = in ? ((k & 1) ? wx1 : wx0) * ((k & 2) ? wy1 : wy0) * ((k & 4) ? wz1 : wz0) : 0.f;
    at[k] = ((long) min(max(iz, 0), P.Z - 1) * P.Y + min(max(iy, 0), P.Y - 1)) * P.X + min(max(ix, 0), P.X - 1);
  }
  const float mask = (P.mask_outside && !t.inside) ? 0.f : 1.f;
  const long V = (long) P.Z * P.Y * P.X;
  const DensityParams dp = pts_density(P, beta_raw);
  for (int c = 0; c < P.C; ++c) {
    const long cb = ((long) b * P.C + c) * V;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = ldf(vol, cb + at[k]);
      if (P.activation) v = density_fwd(dp, v);
      s = __builtin_fmaf(wt[k], v, s);
    }
    s *= mask;
    if (P.channel_last) out[((long) b * P.Pb + p) * P.C + c] = s;
    else out[((long) b * P.C + c) * P.Pb + p] = s;
  }
}

// ---------------------------------------------------------------------------
// backward: rank, fill, gather
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sample_points_rank_kernel(SampleParams P, const float* __restrict__ pts, int* __restrict__ cnt,
                          int* __restrict__ KEY, int* __restrict__ RANK, float4* __restrict__ F,
                          long ncell_b) {
  const long gid = (long) blockIdx.x * 256 + threadIdx.x;
  const long total = (long) P.B * P.Pb;
  const int lane = threadIdx.x & 63;
  const long g = min(gid, total - 1);
  const unsigned b = (unsigned) (g / P.Pb);
  const PointTap t = point_tap(P, pts + g * 3);
  const float flx = floorf(t.fx), fly = floorf(t.fy), flz = floorf(t.fz);
  // a point contributes iff at least one tap per axis is a voxel: floor in [-1, size - 1]
  bool act = gid < total && flx >= -1.f && flx <= (float) (P.X - 1) && fly >= -1.f &&
             fly <= (float) (P.Y - 1) && flz >= -1.f && flz <= (float) (P.Z - 1);
  if (P.mask_outside && !t.inside) act = false;
  const int key = act ? pack_cell_key((int) flx, (int) fly, (int) flz) : 0;
  const long cell = key_to_cell(key, P.Y, P.X, b, ncell_b);
  const LaneRun r = lane_run(act, cell, lane);
  int base = 0;
  if (r.head) base = atomicAdd(cnt + cell, r.len);
  base = __shfl(base, act ? r.start : lane, 64);
  if (gid < total) {
    KEY[gid] = key;
    if (act) {
      RANK[gid] = base + (lane - r.start);
      F[gid] = make_float4(t.fx, t.fy, t.fz, 0.f);
    }
  }
}

__global__ void __launch_bounds__(256)
sample_points_fill_kernel(SampleParams P, const int* __restrict__ KEY, const int* __restrict__ RANK,
                          const float4* __restrict__ F, const int* __restrict__ off,
                          const int* __restrict__ boff, float4* __restrict__ R, long ncell_b) {
  const long gid = (long) blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long) P.B * P.Pb) return;
  const int key = KEY[gid];
  if (key == 0) return;
  const long cell = key_to_cell(key, P.Y, P.X, (unsigned) (gid / P.Pb), ncell_b);
  const long slot = (long) boff[cell / kScanTile] + off[cell] + RANK[gid];
  const float4 f = F[gid];
  R[slot] = make_float4(f.x, f.y, f.z, __int_as_float((int) gid));
}

// grad_out [B, C, Pb] -> channel-last rows [B * Pb][CP] (CP = C padded to a multiple of 4)
__global__ void __launch_bounds__(256)
sample_points_grad_rows_kernel(SampleParams P, const float* __restrict__ gout, float* __restrict__ G, int CP) {
  const long gid = (long) blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long) P.B * P.Pb) return;
  const long b = gid / P.Pb, p = gid % P.Pb;
  for (int c = 0; c < CP; ++c) {
    float v = 0.f;
    if (c < P.C) v = P.channel_last ? gout[gid * P.C + c] : gout[(b * P.C + c) * P.Pb + p];
    G[gid * CP + c] = v;
  }
}

template <int CP4>
__device__ __forceinline__ void point_accumulate(const CellRanges& cr, int k, const float4* __restrict__ R,
                                                 const float* __restrict__ G, float fix, float fiy,
                                                 float fiz, float (&acc)[CP4 * 4]) {
  constexpr int CP = CP4 * 4;
  const bool in = k < cr.tot;
  const float4 a = R[cell_pos(cr, min(k, cr.tot - 1))];
  const float wt = in ? cell_tap_weight(a.x, fix) * cell_tap_weight(a.y, fiy) * cell_tap_weight(a.z, fiz) : 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(G + (long) __float_as_int(a.w) * CP);
#pragma unroll
  for (int c4 = 0; c4 < CP4; ++c4) {
    const float4 f = g4[c4];
    acc[c4 * 4] = __builtin_fmaf(wt, f.x, acc[c4 * 4]);
    acc[c4 * 4 + 1] = __builtin_fmaf(wt, f.y, acc[c4 * 4 + 1]);
    acc[c4 * 4 + 2] = __builtin_fmaf(wt, f.z, acc[c4 * 4 + 2]);
    acc[c4 * 4 + 3] = __builtin_fmaf(wt, f.w, acc[c4 * 4 + 3]);
  }
}

// value stored for channel c of voxel vox: the activation's derivative is applied here, once per
// voxel; returns the d/dbeta term of this element
template <typename T>
__device__ __forceinline__ float finish_voxel(const SampleParams& P, const DensityParams& dp,
                                              const T* __restrict__ vol, long idx, float v, float& out) {
  if (!P.activation) { out = v; return 0.f; }
  float sigma, dsig_ds, dsig_db;
  density_all(dp, ldf(vol, idx), sigma, dsig_ds, dsig_db);
  out = v * dsig_ds;
  return v * dsig_db;
}

template <typename T, int CP4>
__global__ void __launch_bounds__(256)
sample_points_gather_kernel(SampleParams P, const T* __restrict__ vol, const float* __restrict__ beta_raw,
                            const int* __restrict__ off, const int* __restrict__ boff,
                            const float4* __restrict__ R, const float* __restrict__ G,
                            float* __restrict__ gvol, float* __restrict__ grad_beta,
                            float* __restrict__ beta_part, int* __restrict__ heavy,
                            int* __restrict__ nheavy, long ncell_b, int runs_x) {
  constexpr int CP = CP4 * 4;
  __shared__ float outs[CP][PVPB + 1];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int g = tid / PGL, l = tid % PGL;
  const unsigned lin = blockIdx.x;
  const int bx = lin % (unsigned) runs_x;
  const unsigned rest = lin / (unsigned) runs_x;
  const int ix = bx * PVPB + g, iy = rest % (unsigned) P.Y;
  const int zb = rest / (unsigned) P.Y;
  const int iz = zb % P.Z, b = zb / P.Z;
  const bool vox_ok = ix < P.X;
  const CellRanges cr = cell_ranges<PGL>(P.Y, P.X, off, boff, ncell_b, b, min(ix, P.X - 1), iy, iz, l);
  float acc[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) acc[c] = 0.f;
  if (vox_ok && cr.tot > kPtsHeavy) {
    if (l == 0) heavy[atomicAdd(nheavy, 1)] = ((b * P.Z + iz) * P.Y + iy) * P.X + ix;
  } else if (vox_ok) {
    const float fix = (float) ix, fiy = (float) iy, fiz = (float) iz;
    for (int k = l; k < cr.tot; k += PGL) point_accumulate<CP4>(cr, k, R, G, fix, fiy, fiz, acc);
  }
  {
    int cbase = 0;
    reduce_halving<CP, PGL / 2, PGL, CP>(acc, l, cbase);
    constexpr int NL = reduce_left<CP, PGL / 2>();
    constexpr int DUP = reduce_dups<CP, PGL / 2>();
    if ((l & DUP) == 0) {
#pragma unroll
      for (int c = 0; c < NL; ++c) outs[cbase + c][g] = acc[c];
    }
  }
  __syncthreads();
  const long V = (long) P.Z * P.Y * P.X;
  const long vox0 = ((long) iz * P.Y + iy) * P.X + (long) bx * PVPB;
  const DensityParams dp = pts_density(P, beta_raw);
  float dbeta = 0.f;
  for (int e = tid; e < P.C * PVPB; e += 256) {
    const int c = e / PVPB, gx = e % PVPB;
    if (bx * PVPB + gx >= P.X) continue;
    const long idx = ((long) b * P.C + c) * V + vox0 + gx;
    float o;
    dbeta += finish_voxel(P, dp, vol, idx, outs[c][gx], o);
    gvol[idx] = o;
  }
  if (P.activation && grad_beta && P.density_mode == VAMP_DENSITY_SDF_LAPLACE) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dbeta += __shfl_down(dbeta, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = dbeta;
    __syncthreads();
    // one partial per workgroup, summed by sample_points_beta_kernel: 40 000 atomics on one
    // address would serialise at the memory side (~7 ns each)
    if (tid == 0) beta_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

__global__ void __launch_bounds__(1024)
sample_points_beta_kernel(const float* __restrict__ part, int n, const float* __restrict__ beta_raw,
                          float* __restrict__ grad_beta) {
  __shared__ float red[16];
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) v += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float sgn = (beta_raw[0] > 0.f) ? 1.f : ((beta_raw[0] < 0.f) ? -1.f : 0.f);
    atomicAdd(grad_beta, sgn * tot);
  }
}

template <typename T, int CP4>
__global__ void __launch_bounds__(256)
sample_points_heavy_kernel(SampleParams P, const T* __restrict__ vol, const float* __restrict__ beta_raw,
                           const int* __restrict__ off, const int* __restrict__ boff,
                           const float4* __restrict__ R, const float* __restrict__ G,
                           float* __restrict__ gvol, float* __restrict__ grad_beta,
                           const int* __restrict__ heavy, const int* __restrict__ nheavy, long ncell_b) {
  constexpr int CP = CP4 * 4;
  __shared__ float part[4][CP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long V = (long) P.Z * P.Y * P.X;
  const DensityParams dp = pts_density(P, beta_raw);
  const int n = *nheavy;
  for (int item = blockIdx.x; item < n; item += gridDim.x) {
    int vid = heavy[item];
    const int ix = vid % P.X; vid /= P.X;
    const int iy = vid % P.Y; vid /= P.Y;
    const int iz = vid % P.Z, b = vid / P.Z;
    const CellRanges cr = cell_ranges<64>(P.Y, P.X, off, boff, ncell_b, b, ix, iy, iz, lane);
    float acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = 0.f;
    const float fix = (float) ix, fiy = (float) iy, fiz = (float) iz;
    for (int k = tid; k < cr.tot; k += 256) point_accumulate<CP4>(cr, k, R, G, fix, fiy, fiz, acc);
    {
      int cbase = 0;
      reduce_halving<CP, 32, 64, CP>(acc, lane, cbase);
      constexpr int NL = reduce_left<CP, 32>();
      constexpr int DUP = reduce_dups<CP, 32>();
      if ((lane & DUP) == 0) {
#pragma unroll
        for (int c = 0; c < NL; ++c) part[wv][cbase + c] = acc[c];
      }
    }
    __syncthreads();
    if (tid < P.C) {
      const float v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      const long idx = ((long) b * P.C + tid) * V + ((long) iz * P.Y + iy) * P.X + ix;
      float o;
      const float db = finish_voxel(P, dp, vol, idx, v, o);
      gvol[idx] = o;
      if (P.activation && grad_beta && P.density_mode == VAMP_DENSITY_SDF_LAPLACE && db != 0.f) {
        const float sgn = (beta_raw[0] > 0.f) ? 1.f : ((beta_raw[0] < 0.f) ? -1.f : 0.f);
        atomicAdd(grad_beta, sgn * db);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct PtsWs {
  int *cnt, *off, *bsum, *boff, *aux, *heavy, *key, *rank;
  float4 *F, *R;
  float *G, *beta_part;
  size_t bytes;
};

static PtsWs pts_ws(const VampSampleDesc* d, long n, void* scratch) {
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const long ntile = ncell / kScanTile;
  const size_t pts = (size_t) d->B * n;
  const size_t voxels = (size_t) d->B * d->Z * d->Y * d->X;
  const int CP = (d->C + 3) / 4 * 4;
  char* p = static_cast<char*>(scratch);
  PtsWs w;
  w.cnt = reinterpret_cast<int*>(p); p += align_up((size_t) (ncell + kScanPad) * sizeof(int), 256);   // + the scan's ticket word
  w.off = reinterpret_cast<int*>(p); p += align_up((size_t) ncell * sizeof(int), 256);
  w.bsum = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.boff = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.aux = reinterpret_cast<int*>(p); p += align_up((size_t) (ntile + 4) * sizeof(int), 256);
  w.heavy = reinterpret_cast<int*>(p); p += align_up(voxels * sizeof(int), 256);
  w.key = reinterpret_cast<int*>(p); p += align_up(pts * sizeof(int), 256);
  w.rank = reinterpret_cast<int*>(p); p += align_up(pts * sizeof(int), 256);
  w.F = reinterpret_cast<float4*>(p); p += align_up(pts * sizeof(float4), 256);
  w.R = reinterpret_cast<float4*>(p); p += align_up(pts * sizeof(float4), 256);
  w.G = reinterpret_cast<float*>(p); p += align_up(pts * CP * sizeof(float), 256);
  const size_t nblk = (size_t) ((d->X + PVPB - 1) / PVPB) * d->Y * d->Z * d->B;
  w.beta_part = reinterpret_cast<float*>(p); p += align_up(nblk * sizeof(float), 256);
  w.bytes = (size_t) (p - static_cast<char*>(scratch));
  return w;
}

template <typename T>
static int backward_t(const VampSampleDesc* d, const SampleParams& P, const void* volume, const float* beta,
                      const float* points, long n, const float* grad_out, float* grad_volume,
                      float* grad_beta, void* workspace, hipStream_t s) {
  const PtsWs w = pts_ws(d, n, workspace);
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const long ntile = ncell / kScanTile;
  const long ncell_b = (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1);
  const long pts = (long) d->B * n;
  const size_t voxels = (size_t) d->B * d->Z * d->Y * d->X;
  VAMP_REQUIRE(voxels < 0x7fffffffu && ncell < 0x7fffffffL, "voxel / cell count exceeds 2^31");
  const int CP = (d->C + 3) / 4 * 4, CP4 = CP / 4;
  const T* vol = static_cast<const T*>(volume);
  if (int ze = launch_zero(w.cnt, (size_t) (ncell + kScanPad) * sizeof(int), s)) return ze;
  int* nheavy = w.aux + ntile + 1;
  const unsigned pgrid = (unsigned) std::max<long>(1, (pts + 255) / 256);
  if (pts > 0) {
    sample_points_rank_kernel<<<pgrid, 256, 0, s>>>(P, points, w.cnt, w.key, w.rank, w.F, ncell_b);
    if (int e = check_launch("sample_points_rank_kernel")) return e;
  }
  if (int e = launch_cell_scan(w.cnt, w.off, w.bsum, w.boff, w.aux, ncell, s)) return e;
  if (int ze = launch_zero(nheavy, sizeof(int), s)) return ze;
  if (pts > 0) {
    sample_points_fill_kernel<<<pgrid, 256, 0, s>>>(P, w.key, w.rank, w.F, w.off, w.boff, w.R, ncell_b);
    if (int e = check_launch("sample_points_fill_kernel")) return e;
    sample_points_grad_rows_kernel<<<pgrid, 256, 0, s>>>(P, grad_out, w.G, CP);
    if (int e = check_launch("sample_points_grad_rows_kernel")) return e;
  }
  const int runs_x = (d->X + PVPB - 1) / PVPB;
  const long nblk = (long) runs_x * d->Y * d->Z * d->B;
  VAMP_REQUIRE(nblk < 0x7fffffffL, "too many x-runs");
  const unsigned hgrid = (unsigned) std::min<size_t>(voxels, 4096);
#define VAMP_PTS(CP4V)                                                                              \
  do {                                                                                              \
    sample_points_gather_kernel<T, CP4V><<<(unsigned) nblk, 256, 0, s>>>(                           \
        P, vol, beta, w.off, w.boff, w.R, w.G, grad_volume, grad_beta, w.beta_part, w.heavy, nheavy,  \
        ncell_b, runs_x);                                                                           \
    sample_points_heavy_kernel<T, CP4V><<<hgrid, 256, 0, s>>>(                                      \
        P, vol, beta, w.off, w.boff, w.R, w.G, grad_volume, grad_beta, w.heavy, nheavy, ncell_b);  \
  } while (0)
  switch (CP4) {
    case 1: VAMP_PTS(1); break;
    case 2: VAMP_PTS(2); break;
    case 3: VAMP_PTS(3); break;
    case 4: VAMP_PTS(4); break;
    case 5: VAMP_PTS(5); break;
    case 6: VAMP_PTS(6); break;
    case 7: VAMP_PTS(7); break;
    default: VAMP_PTS(8); break;
  }
#undef VAMP_PTS
  if (int e = check_launch("sample_points_gather_kernel")) return e;
  if (P.activation && grad_beta && P.density_mode == VAMP_DENSITY_SDF_LAPLACE) {
    sample_points_beta_kernel<<<1, 1024, 0, s>>>(w.beta_part, (int) nblk, beta, grad_beta);
    return check_launch("sample_points_beta_kernel");
  }
  return VAMP_OK;
}

}  // namespace vamp

using namespace vamp;

extern "C" {

size_t vamp_sample_points_workspace_bytes(const VampSampleDesc* d, int64_t points_per_sample) {
  if (!d || points_per_sample < 0) return 0;
  return pts_ws(d, (long) points_per_sample, nullptr).bytes;
}

int vamp_sample_points_forward(const VampSampleDesc* d, const void* volume, const float* beta,
                               const float* points, int64_t points_per_sample, float* out,
                               void* stream) {
  if (int e = validate(d, (long) points_per_sample)) return e;
  VAMP_REQUIRE(volume && (points || points_per_sample == 0) && (out || points_per_sample == 0), "null pointer");
  VAMP_REQUIRE(beta || !d->activation || d->density_mode == VAMP_DENSITY_SIGMOID, "beta is NULL");
  if (points_per_sample == 0) return VAMP_OK;
  const SampleParams P = to_params(d, (int) points_per_sample);
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned) ((points_per_sample + 255) / 256), d->B);
  if (d->in_dtype == VAMP_F32)
    sample_points_fwd_kernel<float><<<grid, 256, 0, s>>>(P, static_cast<const float*>(volume), beta, points, out);
  else
    sample_points_fwd_kernel<__hip_bfloat16><<<grid, 256, 0, s>>>(
        P, static_cast<const __hip_bfloat16*>(volume), beta, points, out);
  return check_launch("sample_points_fwd_kernel");
}

int vamp_sample_points_backward(const VampSampleDesc* d, const void* volume, const float* beta,
                                const float* points, int64_t points_per_sample,
                                const float* grad_out, float* grad_volume, float* grad_beta,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = validate(d, (long) points_per_sample)) return e;
  VAMP_REQUIRE(volume && grad_volume && (points_per_sample == 0 || (points && grad_out)), "null pointer");
  VAMP_REQUIRE(!d->activation || d->density_mode == VAMP_DENSITY_SIGMOID || (beta && grad_beta),
               "beta / grad_beta is NULL");
  const size_t need = vamp_sample_points_workspace_bytes(d, points_per_sample);
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  const SampleParams P = to_params(d, (int) points_per_sample);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->in_dtype == VAMP_F32)
    return backward_t<float>(d, P, volume, beta, points, (long) points_per_sample, grad_out, grad_volume,
                             grad_beta, workspace, s);
  return backward_t<__hip_bfloat16>(d, P, volume, beta, points, (long) points_per_sample, grad_out,
                                    grad_volume, grad_beta, workspace, s);
}

}  // extern "C"
