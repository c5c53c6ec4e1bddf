// LIFT backward, pixel-tile form ("own the image tile, pull the voxels that see it").
// Autograd of base_vampire2.py:507-514 (grid_sampler_3d backward + the camera mean), third
// formulation beside the cell list (lift_bwd_cell.hip) and the atomic splat (lift.hip).
//
// A workgroup owns a kTH x kTW tile of one camera's feature map: its grad_depth columns
// [D][tile], its grad_feat [C][tile] and the tile's features live in LDS.  The voxels whose
// projection touches the tile are those inside the tile's viewing pyramid.  The pyramid is cut
// into groups of kSlabGroup depth slabs; the image of a (tile +- one pixel) x (depth range) box
// under the inverse camera map is convex, so the axis-aligned bounding box of its eight corners
// in voxel-index space holds every voxel centre that can project into it.  The workgroup walks
// those boxes (~4 500 non-empty boxes and 2.1 M candidate voxels at cfg-B, against 2 x 3.84 M
// projections of the count and fill passes), projects each candidate with the forward's
// bit-exact chain -- that alone decides validity and taps, the boxes only have to be supersets --
// and queues the hits; full queues are drained lane = pair: the pair's normalised grad_out row
// and depth taps are loaded once, and its (up to) four taps inside the tile are added to the LDS
// accumulators with ds_add_f32.  A pair whose taps straddle tiles is found by each of those tiles,
// which adds only its own pixels, so every output element has one owner and is stored once:
// no sort, no records in HBM, no float atomics on global memory, no memset of the outputs.
//
// Balance: the pairs of a tile grow with the square of the depth, and a tile on the horizon sees
// seven times the average.  The depth bins are therefore cut into kRanges ranges of about equal
// pyramid volume, one workgroup per (tile, range): it owns the range's grad_depth bins, walks the
// boxes of the floor taps that reach them, and writes a partial grad_feat (each pair counted in
// the range its floor tap lies in) that lift_tile_sum_kernel adds up.
//
// Needs the image-plane matrix `ida` to be 2-D affine (rows 2, 3 = identity, no depth column:
// what the reference builds, base_exp / BEVDepth `ida_mat`); any other matrix sends the
// workgroup over the whole grid (correct, slow).  C <= 16 (one hit word per voxel).
#include "lift_common.hpp"

#include <algorithm>

// Accumulation is 64-bit fixed point: on gfx950 ds_add_f32 is served at one wave instruction per
// ~100-190 cycles per CU (tools/microbench/lds_atomic.hip: 193 cycles alone, 768 with eight waves
// issuing), ds_add_u32 / ds_add_u64 at 11-19 -- a float-atomic version of this kernel took 860 us.
// The scale is a power of two taken from bounds on the operands (see tile_scale), so the sums
// carry >= 28 bits below a typical term and, integer adds being associative, come out the same
// bits on every run.
typedef unsigned long long acc_t;
#define TILE_ADD(ptr, v) atomicAdd((ptr), (acc_t) __float2ll_rn((v) * S))

namespace vamp {

constexpr int kTW = 8, kTH = 8, kTP = kTW * kTH;    // tile of feature-map pixels
constexpr int kSlabGroup = 4;                       // depth slabs per candidate box
constexpr int kTileThreads = 512;
constexpr int kQueue = 1024;                        // queued pairs (16 bytes each)
constexpr int kMaxBoxes = 64;
constexpr int kMaxRanges = 8;

struct TileRanges {
  int n;
  int bin[kMaxRanges + 1];         // range r owns grad_depth bins [bin[r], bin[r + 1])
};

struct TileBox {
  int x0, y0, z0, nx, ny, nz;      // candidate voxels [x0, x0+nx) x ...
  int s_lo, s_hi;                  // floor depth taps iz0 this box is responsible for
};

// 4x4 inverse by cofactors in double, every index static (a pivoting elimination indexes its
// rows dynamically, which puts the matrix in scratch memory: 0.4 ms of dependent scratch round
// trips for one thread).  False when singular.
__device__ inline bool inverse4(const double (&m)[16], double (&inv)[16]) {
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (!(fabs(det) > 1e-300)) return false;
  const double id = 1.0 / det;
#pragma unroll
  for (int i = 0; i < 16; ++i) inv[i] *= id;
  return true;
}

template <typename T>
__global__ void __launch_bounds__(kTileThreads)
lift_bwd_tile_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                     const float* __restrict__ ys, const float* __restrict__ zs,
                     const T* __restrict__ depth, const T* __restrict__ feat,
                     const float* __restrict__ gout, const uint64_t* __restrict__ hits,
                     float* __restrict__ gdepth, float* __restrict__ gfeat, TileRanges R, int dmax,
                     const unsigned* __restrict__ maxima) {
  extern __shared__ acc_t smem64[];
  const int C = P.C;
  const int rng = blockIdx.x % R.n;
  // grad_depth bins [b0, b1) of this workgroup; floor taps it has to see: [b0 - 1, b1 - 1]
  const int b0 = P.use_depth ? R.bin[rng] : 0, b1 = P.use_depth ? R.bin[rng + 1] : 0;
  const int D = b1 - b0;
  acc_t* s_gd = smem64;                             // [D][kTP]  (dmax planes reserved)
  acc_t* s_gf = s_gd + dmax * kTP;                  // [C][kTP]
  float* s_ft = reinterpret_cast<float*>(s_gf + C * kTP);    // [C][kTP]
  float4* s_q = reinterpret_cast<float4*>(s_ft + C * kTP);   // [kQueue] {vox, fx, fy, fz}
  __shared__ float s_scale[2];
  __shared__ TileBox s_box[kMaxBoxes];
  __shared__ int s_nbox, s_qn;
  __shared__ float s_inv[16 + 6];                   // inverse of m1 * m0, inverse of the 2-D ida

  const int tid = threadIdx.x, lane = tid & 63;
  const int tiles_x = (P.fW + kTW - 1) / kTW, tiles_y = (P.fH + kTH - 1) / kTH;
  const int tile = (blockIdx.x / R.n) % (tiles_x * tiles_y);
  const int bn = blockIdx.x / (R.n * tiles_x * tiles_y);
  const int b = bn / P.N;
  const int X0 = (tile % tiles_x) * kTW, Y0 = (tile / tiles_x) * kTH;
  const int X1 = min(X0 + kTW, P.fW) - 1, Y1 = min(Y0 + kTH, P.fH) - 1;
  const float* m = mats + (long) bn * 48;
  const long HW = (long) P.fH * P.fW;
  const long V = (long) P.Z * P.Y * P.X;

  // ---- zero the accumulators, stage the tile's features, invert the camera --------------------
  for (int i = tid; i < (dmax + C) * kTP; i += kTileThreads) s_gd[i] = 0;
  for (int i = tid; i < C * kTP; i += kTileThreads) {
    const int c = i / kTP, p = i % kTP;
    const int iy = Y0 + p / kTW, ix = X0 + p % kTW;
    s_ft[i] = (iy < P.fH && ix < P.fW) ? ldf(feat, ((long) bn * C + c) * HW + (long) iy * P.fW + ix) : 0.f;
  }
  if (tid == 0) {
    s_qn = 0;
    // fixed-point scale: |term| <= max |grad_out / (hits + 1e-6)| * max(|depth|, C |feat|, 1);
    // up to 2^14 terms per accumulator; 2^62 of headroom
    const float gm = __uint_as_float(maxima[0]), dm = __uint_as_float(maxima[1]), fm = __uint_as_float(maxima[2]);
    const float bound = gm * fmaxf(fmaxf(dm, (float) C * fm), 1.0f) * 16384.0f;
    float sc = 1.0f;
    if (bound > 0.f && bound < 3.0e38f) sc = exp2f(floorf(62.0f - log2f(bound)));
    if (!(sc > 0.f) || !(sc < 3.0e38f)) sc = 1.0f;
    s_scale[0] = sc; s_scale[1] = 1.0f / sc;
    double a[16], ai[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v += (double) m[16 + r * 4 + k] * (double) m[k * 4 + c];
        a[r * 4 + c] = v;
      }
    bool ok = inverse4(a, ai);
    // ida = [[a b 0 tx] [c d 0 ty] [0 0 1 0] [0 0 0 1]]
    const float* q = m + 32;
    ok = ok && q[2] == 0.f && q[6] == 0.f && q[8] == 0.f && q[9] == 0.f && q[10] == 1.f && q[11] == 0.f &&
         q[12] == 0.f && q[13] == 0.f && q[14] == 0.f && q[15] == 1.f;
    const double det = (double) q[0] * q[5] - (double) q[1] * q[4];
    ok = ok && fabs(det) > 1e-300 && P.X > 1 && P.Y > 1 && P.Z > 1;
    if (ok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s_inv[i] = (float) ai[i];
      s_inv[16] = (float) (q[5] / det);  s_inv[17] = (float) (-q[1] / det);
      s_inv[18] = (float) (-q[4] / det); s_inv[19] = (float) (q[0] / det);
      s_inv[20] = q[3]; s_inv[21] = q[7];
      s_nbox = -1;                                  // boxes follow
    } else {
      s_box[0] = TileBox{0, 0, 0, P.X, P.Y, P.Z, b0 - 1, b1 - 1};
      s_nbox = 1;
    }
  }
  __syncthreads();

  // ---- candidate boxes: thread = (slab group, corner) ---------------------------------------------
  // floor depth tap iz0 = s  <=>  zz in [d_lo + (s + 0.5) dz, d_lo + (s + 1.5) dz), dz = d_span / D;
  // floor pixel tap ix0 in [X0 - 1, X1]  <=>  fx in [X0 - 1, X1 + 1)
  if (s_nbox < 0) {
    // floor taps [t0, t1] of this range, kSlabGroup per box (range 0 also takes tap -1)
    const int t0 = b0 - 1, t1 = b1 - 1;
    const int ngroups = P.use_depth ? (t1 - t0 + 1 + kSlabGroup - 1) / kSlabGroup : 1;
    const int g = tid >> 3, corner = tid & 7;
    float lo3[3] = {0.f, 0.f, 0.f}, hi3[3] = {0.f, 0.f, 0.f};
    int s_lo = -1, s_hi = P.D;
    if (g < ngroups && g < kMaxBoxes) {
      float z_near, z_far;
      if (P.use_depth) {
        const float dz = P.d_span / (float) P.D;
        s_lo = t0 + g * kSlabGroup;
        s_hi = min(s_lo + kSlabGroup - 1, t1);
        z_near = fmaxf(P.d_lo + ((float) s_lo + 0.5f) * dz, P.d_lo) * 0.9999f;
        z_far = fminf(P.d_lo + ((float) s_hi + 1.5f) * dz, P.d_hi) * 1.0001f;
      } else {
        // no depth axis: everything in front of the camera; the far end is the grid's diagonal
        const float ex = xs[P.X - 1] - xs[0], ey = ys[P.Y - 1] - ys[0], ez = zs[P.Z - 1] - zs[0];
        z_near = 1e-3f;
        z_far = 4.0f * sqrtf(ex * ex + ey * ey + ez * ez) + 100.0f;
      }
      const float fxc = (corner & 1) ? (float) (X1 + 1) : (float) (X0 - 1);
      const float fyc = (corner & 2) ? (float) (Y1 + 1) : (float) (Y0 - 1);
      const float zz = (corner & 4) ? z_far : z_near;
      // feature coordinates -> image coordinates (inverse of the normalise / unnormalise pair)
      const float u = (fxc + 0.5f) * P.u_div / (float) P.fW, v = (fyc + 0.5f) * P.v_div / (float) P.fH;
      const float du = u - s_inv[20], dv = v - s_inv[21];
      const float up = s_inv[16] * du + s_inv[17] * dv, vp = s_inv[18] * du + s_inv[19] * dv;
      const float cx = up * zz, cy = vp * zz, cz = zz;
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = s_inv[r * 4] * cx + s_inv[r * 4 + 1] * cy + s_inv[r * 4 + 2] * cz + s_inv[r * 4 + 3];
      const float iw = 1.0f / e[3];
      lo3[0] = hi3[0] = (e[0] * iw - xs[0]) / (xs[1] - xs[0]);
      lo3[1] = hi3[1] = (e[1] * iw - ys[0]) / (ys[1] - ys[0]);
      lo3[2] = hi3[2] = (e[2] * iw - zs[0]) / (zs[1] - zs[0]);
    }
    // min / max over the 8 corners (8 consecutive lanes)
#pragma unroll
    for (int o = 1; o < 8; o <<= 1)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo3[a] = fminf(lo3[a], __shfl_xor(lo3[a], o, 64));
        hi3[a] = fmaxf(hi3[a], __shfl_xor(hi3[a], o, 64));
      }
    bool keep = false;
    TileBox bx{0, 0, 0, 0, 0, 0, s_lo, s_hi};
    if (corner == 0 && g < ngroups && g < kMaxBoxes) {
      // NaN / overflow anywhere: take the whole axis
      const int n3[3] = {P.X, P.Y, P.Z};
      int a0[3], a1[3];
      keep = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float l = lo3[a] - 0.05f, h = hi3[a] + 0.05f;
        const bool fin = (l == l) && (h == h) && fabsf(l) < 1e9f && fabsf(h) < 1e9f;
        a0[a] = fin ? max(0, (int) ceilf(l)) : 0;
        a1[a] = fin ? min(n3[a] - 1, (int) floorf(h)) : n3[a] - 1;
        if (a1[a] < a0[a]) keep = false;
      }
      bx.x0 = a0[0]; bx.nx = a1[0] - a0[0] + 1;
      bx.y0 = a0[1]; bx.ny = a1[1] - a0[1] + 1;
      bx.z0 = a0[2]; bx.nz = a1[2] - a0[2] + 1;
    }
    __syncthreads();
    if (tid == 0) s_nbox = 0;
    __syncthreads();
    if (keep) s_box[atomicAdd(&s_nbox, 1)] = bx;
    __syncthreads();
    // more slab groups than box slots: one box, the whole grid
    if (tid == 0 && ngroups > kMaxBoxes) { s_box[0] = TileBox{0, 0, 0, P.X, P.Y, P.Z, t0, t1}; s_nbox = 1; }
    __syncthreads();
  }

  const float S = s_scale[0];
  // ---- drain: lane = queued pair ------------------------------------------------------------------
  auto drain = [&]() {
    const int qn = s_qn;
    // Queue order is box order: neighbours in the queue are neighbours in space and tend to land
    // on the same pixel, and ds_add_f32 lanes that share an address are served one after the
    // other at ~100 cycles each (measured: a 1 000-pair drain took 600 k cycles).  The lanes of a
    // wave therefore take entries 77 apart (an odd stride is a bijection on a power of two).
    int qm = 64;
    while (qm < qn) qm <<= 1;
    for (int k = tid; k < qm; k += kTileThreads) {
      const int i = (k * 77) & (qm - 1);
      if (i >= qn) continue;
      const float4 e = s_q[i];
      const int vox = __float_as_int(e.x);
      const float fx = e.y, fy = e.z, fz = e.w;
      const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
      const int ix0 = (int) flx, iy0 = (int) fly, iz0 = (int) flz;
      const float wx1 = fx - flx, wx0 = (flx + 1.0f) - fx;
      const float wy1 = fy - fly, wy0 = (fly + 1.0f) - fy;
      const float wz1 = fz - flz, wz0 = (flz + 1.0f) - fz;
      // grad_out / (hit count + 1e-6), the camera-mean factor of bv2:512-514
      const uint64_t hw = hits[(long) b * V + vox];
      const float* g = gout + (long) b * C * V + vox;
      float gs[16];
#pragma unroll
      for (int c = 0; c < 16; ++c)
        gs[c] = c < C ? g[(long) c * V] * __builtin_amdgcn_rcpf((float) ((hw >> (4 * c)) & 15) + 1e-6f) : 0.f;
      const float wj[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
      float dep[4] = {0.f, 0.f, 0.f, 0.f};
      if (P.use_depth) {
        const T* dptr = depth + (long) bn * P.D * HW;
#pragma unroll
        for (int kz = 0; kz < 2; ++kz) {
          const int iz = iz0 + kz;
          const bool zin = iz >= 0 && iz < P.D;
          const float wz = zin ? (kz ? wz1 : wz0) : 0.f;
          const long zo = (long) min(max(iz, 0), P.D - 1) * HW;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int iy = iy0 + (j >> 1), ix = ix0 + (j & 1);
            const bool in = iy >= 0 && iy < P.fH && ix >= 0 && ix < P.fW;
            dep[j] += (in ? wz : 0.f) *
                      ldf(dptr, zo + (long) min(max(iy, 0), P.fH - 1) * P.fW + min(max(ix, 0), P.fW - 1));
          }
        }
      } else {
        const float w = (iz0 == 0 ? wz0 : 0.f) + (iz0 == -1 ? wz1 : 0.f);
        dep[0] = dep[1] = dep[2] = dep[3] = w;
      }
      // the pair's grad_feat part belongs to the range its floor tap lies in (tap -1: range 0)
      const bool own_feat = !P.use_depth || (iz0 >= (rng == 0 ? -1 : b0) && iz0 < b1);
      // neighbouring lanes are neighbouring voxels and often share their floor pixel: starting
      // the tap loop at a different tap per lane keeps them off the same LDS word
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = (jj + lane) & 3;
        const int iy = iy0 + (j >> 1), ix = ix0 + (j & 1);
        if (iy < Y0 || iy > Y1 || ix < X0 || ix > X1) continue;         // another tile's pixel (or none)
        const int p = (iy - Y0) * kTW + (ix - X0);
        const float wjj = j == 0 ? wj[0] : (j == 1 ? wj[1] : (j == 2 ? wj[2] : wj[3]));
        const float dpj = j == 0 ? dep[0] : (j == 1 ? dep[1] : (j == 2 ? dep[2] : dep[3]));
        const float pw = own_feat ? wjj * dpj : 0.f;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (c >= C) break;
          if (own_feat) TILE_ADD(s_gf + c * kTP + p, pw * gs[c]);
          dot = __builtin_fmaf(s_ft[c * kTP + p], gs[c], dot);
        }
        if (D > 0) {
          const float wd = wjj * dot;
          if (wd != 0.f) {
            if (iz0 >= b0 && iz0 < b1) TILE_ADD(s_gd + (iz0 - b0) * kTP + p, wz0 * wd);
            if (iz0 + 1 >= b0 && iz0 + 1 < b1) TILE_ADD(s_gd + (iz0 + 1 - b0) * kTP + p, wz1 * wd);
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) s_qn = 0;
    __syncthreads();
  };

  // ---- walk the boxes -----------------------------------------------------------------------------
  const int nbox = s_nbox;
  for (int bi = 0; bi < nbox; ++bi) {
    const TileBox bx = s_box[bi];
    const int plane = bx.nx * bx.ny, cells = plane * bx.nz;
    const float rnx = 1.0f / (float) bx.nx, rpl = 1.0f / (float) plane;
    for (int c0 = 0; c0 < cells; c0 += kTileThreads) {
      const int ci = c0 + tid;
      bool hit = false;
      float4 ent = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ci < cells) {
        // decode (z, y, x) without integer division: float quotient, one correction step
        int cz = (int) ((float) ci * rpl);
        cz -= (cz * plane > ci); cz += ((cz + 1) * plane <= ci);
        const int r = ci - cz * plane;
        int cy = (int) ((float) r * rnx);
        cy -= (cy * bx.nx > r); cy += ((cy + 1) * bx.nx <= r);
        const int cx = r - cy * bx.nx;
        const int vx = bx.x0 + cx, vy = bx.y0 + cy, vz = bx.z0 + cz;
        const LiftTap t = lift_project(P, m, xs[vx], ys[vy], zs[vz]);
        hit = t.valid && t.ix0 >= X0 - 1 && t.ix0 <= X1 && t.iy0 >= Y0 - 1 && t.iy0 <= Y1 &&
              (!P.use_depth || (t.iz0 >= bx.s_lo && t.iz0 <= bx.s_hi));
        ent = make_float4(__int_as_float((vz * P.Y + vy) * P.X + vx), t.fx, t.fy, t.fz);
      }
      // one LDS atomic per wave hands out queue slots
      const unsigned long long hm = __ballot(hit);
      int base = 0;
      if (lane == 0 && hm) base = atomicAdd(&s_qn, __popcll(hm));
      base = __shfl(base, 0, 64);
      if (hit) s_q[base + __popcll(hm & ((1ull << lane) - 1ull))] = ent;
      __syncthreads();
      const int qn = s_qn;                              // everybody reads the same count ...
      __syncthreads();                                  // ... before anybody queues again
      if (qn > kQueue - kTileThreads) drain();
    }
  }
  __syncthreads();
  if (s_qn > 0) drain();

  // ---- store the tile: every element once (grad_feat: this range's partial) ---------------------
  const long gf_elems = (long) P.B * P.N * C * HW;
  for (int i = tid; i < C * kTP; i += kTileThreads) {
    const int c = i / kTP, p = i % kTP;
    const int iy = Y0 + p / kTW, ix = X0 + p % kTW;
    if (iy < P.fH && ix < P.fW) gfeat[rng * gf_elems + ((long) bn * C + c) * HW + (long) iy * P.fW + ix] = (float) ((double) (long long) s_gf[i] * (double) s_scale[1]);
  }
  if (gdepth)
    for (int i = tid; i < D * kTP; i += kTileThreads) {
      const int dz = i / kTP, p = i % kTP;
      const int iy = Y0 + p / kTW, ix = X0 + p % kTW;
      if (iy < P.fH && ix < P.fW) gdepth[((long) bn * P.D + b0 + dz) * HW + (long) iy * P.fW + ix] = (float) ((double) (long long) s_gd[i] * (double) s_scale[1]);
    }
}

// max |x| as the bits of a non-negative float (which order like unsigned integers); one atomic per
// workgroup, few workgroups: device-scope atomics on one word are served one per ~microsecond
template <typename T>
__global__ void __launch_bounds__(1024)
absmax_kernel(const T* __restrict__ x, long n, unsigned* __restrict__ out) {
  __shared__ float red[16];
  float m = 0.f;
  for (long i = (long) blockIdx.x * 1024 + threadIdx.x; i < n; i += (long) gridDim.x * 1024) m = fmaxf(m, fabsf(ldf(x, i)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    atomicMax(out, __float_as_uint(m));
  }
}

// max |grad_out / (hit count + 1e-6)| over the voxels some camera hit: the largest operand the
// accumulation will see.  Voxels whose hit word is zero are left out: no camera sees them (or every
// channel of every tap was exactly zero), they never form a pair, and their 1e6 factor would cost
// 20 bits of every sum.  A single zero-count channel of a voxel that IS hit (the forward's
// exact-zero hit test, tests/golden) keeps its 1e6 in the bound, as the reference's gradient does.
__global__ void __launch_bounds__(1024)
absmax_gs_kernel(const float* __restrict__ gout, const uint64_t* __restrict__ hits, int B, int C, long V,
                 unsigned* __restrict__ out) {
  __shared__ float red[16];
  float m = 0.f;
  for (long i = (long) blockIdx.x * 1024 + threadIdx.x; i < (long) B * V; i += (long) gridDim.x * 1024) {
    const long b = i / V, vox = i % V;
    const uint64_t hw = hits[i];
    if (hw == 0) continue;
    for (int c = 0; c < C; ++c)
      m = fmaxf(m, fabsf(gout[(b * C + c) * V + vox]) * __builtin_amdgcn_rcpf((float) ((hw >> (4 * c)) & 15) + 1e-6f));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    atomicMax(out, __float_as_uint(m));
  }
}

template <typename T>
static int launch_absmax(const T* x, long n, unsigned* out, hipStream_t s) {
  if (!x || n <= 0) return VAMP_OK;
  const unsigned grid = (unsigned) std::min<long>((n + 1023) / 1024, 512);
  VAMP_TIMED(kProfAux, s, (absmax_kernel<T><<<grid, 1024, 0, s>>>(x, n, out)));
  return check_launch("absmax_kernel");
}

// grad_feat = sum of the ranges' partials
__global__ void __launch_bounds__(256)
lift_tile_sum_kernel(const float4* __restrict__ part, float4* __restrict__ out, long n4, int nr) {
  const long i = (long) blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = part[i];
  for (int r = 1; r < nr; ++r) {
    const float4 v = part[r * n4 + i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  out[i] = a;
}

// depth-bin ranges of about equal pyramid volume: the pairs of a bin grow with zz^2 until the rays
// leave the grid (beyond its half-diagonal the population falls off again)
static TileRanges tile_ranges(const VampLiftDesc* d, const float* /*unused*/) {
  TileRanges R;
  if (!d->use_depth) { R.n = 1; R.bin[0] = 0; R.bin[1] = 0; return R; }
  const int nr = 1;   // one range: see DESIGN.md (the ranges need the grid's extent to be balanced)
  R.n = nr;
  double w[1024], tot = 0.0;
  const int Dn = std::min(d->D, 1024);
  const double dz = (double) d->d_span / d->D;
  for (int b = 0; b < Dn; ++b) { const double z = d->d_lo + (b + 0.5) * dz; w[b] = z * z; tot += w[b]; }
  R.bin[0] = 0;
  double acc = 0.0;
  int r = 1;
  for (int b = 0; b < Dn && r < nr; ++b) {
    acc += w[b];
    if (acc >= tot * r / nr) R.bin[r++] = b + 1;
  }
  for (; r < nr; ++r) R.bin[r] = Dn;
  R.bin[nr] = d->D;
  return R;
}

static int tile_dmax(const TileRanges& R) {
  int m = 0;
  for (int r = 0; r < R.n; ++r) m = std::max(m, R.bin[r + 1] - R.bin[r]);
  return m;
}

size_t lift_bwd_tile_scratch_bytes(const VampLiftDesc* d) {
  return 256 + (size_t) kMaxRanges * d->B * d->N * d->C * d->fH * d->fW * sizeof(float);
}

bool lift_bwd_tile_supported(const VampLiftDesc* d) {
  const TileRanges R = tile_ranges(d, nullptr);
  const size_t lds = ((size_t) tile_dmax(R) + d->C) * kTP * sizeof(acc_t) + (size_t) d->C * kTP * sizeof(float) + kQueue * sizeof(float4);
  return d->C <= 16 && (d->C % 4) == 0 && lds <= 150 * 1024 && (long) d->Z * d->Y * d->X < 0x7fffffffL &&
         ((long) d->B * d->N * d->C * d->fH * d->fW) % 4 == 0;
}

int launch_lift_bwd_tile(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, const float* gout,
                         const uint64_t* hits, float* gdepth, float* gfeat, void* scratch,
                         size_t scratch_bytes, hipStream_t s) {
  if (!lift_bwd_tile_supported(d)) return fail(VAMP_EINVAL, "%s: configuration not supported", __func__);
  if (!scratch || scratch_bytes < lift_bwd_tile_scratch_bytes(d))
    return fail(VAMP_ENOSPC, "%s: scratch too small", __func__);
  const LiftParams P = to_params(d);
  const TileRanges R = tile_ranges(d, nullptr);
  const int dmax = tile_dmax(R);
  const size_t lds = ((size_t) dmax + d->C) * kTP * sizeof(acc_t) + (size_t) d->C * kTP * sizeof(float) + kQueue * sizeof(float4);
  // operand bounds for the fixed-point scale: max |grad_out|, |depth|, |feat|
  unsigned* maxima = static_cast<unsigned*>(scratch);
  if (int ze = launch_zero(maxima, 16, s)) return ze;
  const long npix = (long) d->B * d->N * d->fH * d->fW;
  {
    const long V = (long) d->Z * d->Y * d->X;
    const unsigned grid = (unsigned) std::min<long>(((long) d->B * V + 1023) / 1024, 512);
    VAMP_TIMED(kProfAux, s, (absmax_gs_kernel<<<grid, 1024, 0, s>>>(gout, hits, d->B, d->C, V, maxima)));
    if (int e = check_launch("absmax_gs_kernel")) return e;
  }
  if (d->in_dtype == VAMP_F32) {
    if (d->use_depth) if (int e = launch_absmax<float>(static_cast<const float*>(depth), npix * d->D, maxima + 1, s)) return e;
    if (int e = launch_absmax<float>(static_cast<const float*>(feat), npix * d->C, maxima + 2, s)) return e;
  } else {
    if (d->use_depth) if (int e = launch_absmax<__hip_bfloat16>(static_cast<const __hip_bfloat16*>(depth), npix * d->D, maxima + 1, s)) return e;
    if (int e = launch_absmax<__hip_bfloat16>(static_cast<const __hip_bfloat16*>(feat), npix * d->C, maxima + 2, s)) return e;
  }
  const unsigned grid = (unsigned) ((long) d->B * d->N * ((d->fW + kTW - 1) / kTW) * ((d->fH + kTH - 1) / kTH) * R.n);
  float* part = R.n > 1 ? reinterpret_cast<float*>(static_cast<char*>(scratch) + 256) : gfeat;
#define VAMP_TILE(T)                                                                              \
  do {                                                                                            \
    auto k = lift_bwd_tile_kernel<T>;                                                             \
    if (lds > 48 * 1024 &&                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int) lds) != hipSuccess)                                            \
      return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);                           \
    VAMP_TIMED(kProfLiftBwd, s, (k<<<grid, kTileThreads, lds, s>>>(                               \
        P, mats, xs, ys, zs, static_cast<const T*>(depth), static_cast<const T*>(feat), gout, hits, \
        d->use_depth ? gdepth : nullptr, part, R, dmax, maxima)));                                        \
  } while (0)
  if (d->in_dtype == VAMP_F32) VAMP_TILE(float); else VAMP_TILE(__hip_bfloat16);
#undef VAMP_TILE
  if (int e = check_launch("lift_bwd_tile_kernel")) return e;
  if (R.n > 1) {
    const long n4 = (long) d->B * d->N * d->C * d->fH * d->fW / 4;
    VAMP_TIMED(kProfFeatCF, s, (lift_tile_sum_kernel<<<(unsigned) ((n4 + 255) / 256), 256, 0, s>>>(
        reinterpret_cast<const float4*>(part), reinterpret_cast<float4*>(gfeat), n4, R.n)));
  }
  return check_launch("lift_tile_sum_kernel");
}

}  // namespace vamp
