// LIFT backward as a cell list ("sort, then own").  Autograd of base_vampire2.py:507-514
// (grid_sampler_3d backward + the camera mean):
//
//   count   thread per voxel, forward's bit-exact projection: every valid (voxel, camera) pair
//           increments the counter of its cell = (camera, floor tap row + 1, floor tap column + 1),
//           (fH + 1) x (fW + 1) cells per camera
//   scan    two-level exclusive prefix sum of the counters -> cell start offsets (runtime.hip)
//   fill    same walk: the voxel's row grad_out / (hits + 1e-6) goes into a channel-last table
//           [B V, C] (consecutive lanes = consecutive 64-byte rows: streaming stores), and every
//           valid pair puts its voxel index and {wx1, wy1, wz1, iz0} -- 20 bytes -- into the next slot
//           of its cell (slots handed out by atomics on the cell cursor)
//   gather  one wave (or 4, or 16) per feature-map pixel: the pairs whose 2x2 pixel taps include
//           pixel (x, y) are exactly those of the cells (x..x+1, y..y+1), two contiguous ranges of
//           pairs.  Lane = pair: it reads the pair's 20 bytes, the voxel's 64-byte row of the table
//           and the two depth bins of the pixel's own column (staged in LDS); grad_feat partial
//           sums in registers (folded over the lanes at the end), the channel dot product feeds the
//           two depth bins of the pixel's private LDS column.  Every output element is stored once.
//
// (Until round 3 the fill wrote a 112-byte record per pair -- tap weights, depth-interpolated values
// and a copy of the voxel's row -- scattered in cell order as 16-byte pieces: 122 MB of HBM writes
// for 71 MB of records, and the pass was half of the lift backward.  A pair is now 20 bytes: the
// depth-interpolated values come from the consuming pixel's own depth column, the lower-tap weights
// are 1 - w1, and the voxel's row is shared by the pairs of a voxel.  Recomputing the projection in
// the gather instead (4-byte pairs) was measured too: fill 40 us, gather 120 -- every pair is visited
// by four pixels, and the chain with its coordinate loads sits behind the index load.)
//
// Both atomic passes aggregate runs of equal cells across the lanes of a wave (x-neighbouring
// voxels share a cell in the far field), one atomic per run: device-scope atomics are served at
// the memory side on this part and cost ~1 us each.
// No float atomics on global memory, no memset of the outputs, no layout transposes.
#include "lift_common.hpp"
#include "cell_list.hpp"

#include <algorithm>

namespace vamp {

constexpr int LGL = 16;              // lanes per record = channel lanes
constexpr int kMinWaves = 4;         // waves per gather workgroup (more when a pixel takes more)

struct LiftCells {
  int cw, ch;                        // cells per row / column of one camera
  long ncell;                        // padded to the scan tile, + 2 for the range ends
};

static LiftCells lift_cells(const VampLiftDesc* d) {
  LiftCells g;
  g.cw = d->fW + 1;
  g.ch = d->fH + 1;
  const long nc = (long) d->B * d->N * g.cw * g.ch + 2;
  g.ncell = (nc + kScanTile - 1) / kScanTile * kScanTile;
  return g;
}

// ---------------------------------------------------------------------------
// count / fill: thread per voxel, the 64 lanes of a wave are 64 x-consecutive voxels
// ---------------------------------------------------------------------------
template <typename T, int CH, bool FILL>
__global__ void __launch_bounds__(256)
lift_bwd_cell_kernel(LiftParams P, int cw, int ch, const float* __restrict__ mats,
                     const float* __restrict__ xs, const float* __restrict__ ys,
                     const float* __restrict__ zs, const T* __restrict__ depth,
                     const float* __restrict__ gout, const uint64_t* __restrict__ hits,
                     int* __restrict__ cnt, const int* __restrict__ off,
                     const int* __restrict__ boff, int* __restrict__ ids, float4* __restrict__ recs,
                     float4* __restrict__ table, unsigned* __restrict__ amask, int bn_lo, int bn_hi) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + (tid >> 6);
  const int z = blockIdx.z % P.Z, b = blockIdx.z / P.Z;
  const bool live = x < P.X && y < P.Y;
  const int xc = min(x, P.X - 1), yc = min(y, P.Y - 1);
  const float vx = xs[xc], vy = ys[yc], vz = zs[z];
  const long V = (long) P.Z * P.Y * P.X;
  const long vox = ((long) z * P.Y + yc) * P.X + xc;
  const long HW = (long) P.fH * P.fW;

  // camera mask of the voxel: written by the count pass, read by the fill pass, which then
  // projects only the cameras some lane of the wave is valid for (1-3 of 6) instead of all -- and
  // leaves at once when no voxel of the wave is seen by any camera (the half-split's images)
  unsigned vmask = 0xffffffffu, wmask = 0;
  if (FILL && amask) {
    vmask = live ? amask[(long) b * V + vox] : 0u;
    const int lo = max(0, bn_lo - b * P.N), hi = min(P.N, bn_hi - b * P.N);
    const unsigned range = hi > lo ? ((hi - lo >= 32 ? 0xffffffffu : ((1u << (hi - lo)) - 1u)) << lo) : 0u;
    if (!__any((vmask & range) != 0u)) return;
  }

  constexpr int NB = 8;                          // cameras per batch: their atomics are in flight together
  // grad_out / (hit count + 1e-6), the camera-mean factor of bv2:512-514: the same for every
  // camera of the voxel -- one channel-last row of the table per voxel (lanes = consecutive voxels
  // = consecutive rows)
  if (FILL && live) {
    const int nchunk = P.C / CH;
    float4* row = table + ((long) b * V + vox) * (P.C / 4);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const uint64_t hw = hits[((long) b * V + vox) * nchunk + chunk];
      const float* g = gout + ((long) b * P.C + chunk * CH) * V + vox;
      float v[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k)
        v[k] = g[(long) k * V] * __builtin_amdgcn_rcpf((float) ((hw >> (4 * k)) & 15) + 1e-6f);   // 1 ulp: gradients are held to 1e-4
#pragma unroll
      for (int c4 = 0; c4 < CH; c4 += 4) row[(chunk * CH + c4) / 4] = make_float4(v[c4], v[c4 + 1], v[c4 + 2], v[c4 + 3]);
    }
  }

  // images [bn_lo, bn_hi) of the flattened (sample, camera) index: all of them, or one half when
  // the caller runs two halves of the lift backward side by side (an image's records only meet
  // that image's pixels)
  const int n_lo = max(0, bn_lo - b * P.N), n_hi = min(P.N, bn_hi - b * P.N);
  for (int n0 = n_lo; n0 < n_hi; n0 += NB) {
    int base[NB], start[NB];
    long cellk[NB];
    float4 tapk[NB];                             // the pair's fractional tap coordinates and depth plane
    unsigned actm = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int n = n0 + k;
      base[k] = 0;
      start[k] = lane;
      cellk[k] = 0;
      tapk[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n >= n_hi) continue;                   // uniform
      if (FILL && !__any((vmask >> (n & 31)) & 1u)) continue;   // uniform: nobody in this wave sees camera n
      const long bn = (long) b * P.N + n;
      const LiftTap t = lift_project<true>(P, mats + bn * 48, vx, vy, vz);
      // at least one of the four pixel taps must exist
      const bool act = live && t.valid && t.ix0 >= -1 && t.ix0 < P.fW && t.iy0 >= -1 && t.iy0 < P.fH;
      if (act) wmask |= 1u << (n & 31);
      const long cell = (bn * ch + (t.iy0 + 1)) * cw + (t.ix0 + 1);
      cellk[k] = cell;
      tapk[k] = make_float4(t.wx1, t.wy1, t.wz1, __int_as_float(t.iz0));
      const LaneRun r = lane_run(act, cell, lane);
      if (r.head) {
        if (FILL) base[k] = atomicAdd(cnt + cell, r.len);
        else atomicAdd(cnt + cell, r.len);
      }
      if (act) { actm |= 1u << k; start[k] = r.start; }
    }
    if (!FILL) continue;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int n = n0 + k;
      if (n >= n_hi) continue;                   // uniform
      if (!__any((actm >> k) & 1u)) continue;    // uniform
      const int rb = __shfl(base[k], start[k], 64);
      if (!((actm >> k) & 1u)) continue;
      const long slot = (long) off[cellk[k]] + boff[cellk[k] / kScanTile] + rb + (lane - start[k]);
      ids[slot] = (int) vox;
      recs[slot] = tapk[k];
    }
  }
  if (!FILL && amask && live) amask[(long) b * V + vox] = wmask;
}

// ---------------------------------------------------------------------------
// gather: one wave per feature-map pixel (or wpp = 4 / 16 waves for dense configurations, chosen
// on the host from the expected pairs per pixel), lane = PAIR: every lane takes one pair (voxel
// index + fractional taps), loads the voxel's row of the table (64 bytes at C = 16), keeps 16
// grad_feat partial sums in registers and
// adds its two depth-plane terms to the pixel's LDS column.  The depth values a pair needs -- the
// two planes around its projected depth at THIS pixel -- come from the pixel's own depth column,
// staged in LDS once.  The 16 x 64 partial sums are folded with a recursive-halving reduction at
// the end.  C > 16 runs in chunks of 16 channels.
// max(kMinWaves, wpp) waves per workgroup: small workgroups, because a workgroup lives as long
// as its slowest pixel.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024)
lift_bwd_cell_gather_kernel(LiftParams P, int cw, int ch, int wpp, int xgroup,
                            const float* __restrict__ mats, const float* __restrict__ xs,
                            const float* __restrict__ ys, const float* __restrict__ zs,
                            const T* __restrict__ depth, const T* __restrict__ feat,
                            const int* __restrict__ off, const int* __restrict__ boff,
                            const int* __restrict__ ids, const float4* __restrict__ recs,
                            const float4* __restrict__ table,
                            float* __restrict__ gdepth, float* __restrict__ gfeat, long pix_lo, long pix_hi,
                            int softmax_bwd) {
  extern __shared__ float smem[];                // [ppb][Dp] grad columns, [ppb][Dp] depth columns, [waves][16]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int C = P.C, D = P.use_depth ? P.D : 0;
  const int Dp = D | 1;
  const int nw = blockDim.x >> 6;
  const int ppb = nw / wpp;                      // pixels per workgroup
  const int pw = wv / wpp, ws = wv % wpp;        // pixel of this wave, wave index inside the pixel
  float* gd = smem;
  float* dcolumns = smem + ppb * Dp;
  float* accbuf = smem + 2 * ppb * Dp;
  const long HW = (long) P.fH * P.fW;
  const long V = (long) P.Z * P.Y * P.X;
  const long npix = pix_hi;                      // this launch owns pixels [pix_lo, pix_hi)
  // the workgroups of one image row run on one XCD: their 16-byte pieces of a depth plane's row
  // merge into whole lines in that L2
  const long pid0 = pix_lo + (long) xcd_grouped(blockIdx.x, gridDim.x, xgroup) * ppb;
  const long pid = min(pid0 + pw, npix - 1);
  const bool pix_ok = pid0 + pw < npix;
  const long bn = __builtin_amdgcn_readfirstlane((int) (pid / HW));      // wave-uniform: the pixel
  const int pix = __builtin_amdgcn_readfirstlane((int) (pid % HW));
  const int b = (int) (bn / P.N);
  const int iy = pix / P.fW, ix = pix % P.fW;
  const float* m = mats + bn * 48;

  float* gcol = gd + pw * Dp;
  float* dcol = dcolumns + pw * Dp;
  for (int dz = ws * 64 + lane; dz < D; dz += wpp * 64) {
    gcol[dz] = 0.f;
    dcol[dz] = ldf(depth, (bn * P.D + dz) * HW + pix);
  }
  __syncthreads();

  // index ranges of cell rows iy and iy + 1, columns ix .. ix + 1 (mid = where column ix + 1
  // starts).  The cell tells which tap of the pair this pixel is: a pair of cell (row, column)
  // has iy0 = row - 1, ix0 = column - 1.
  int beg0, mid0, n0, beg1, mid1, tot;
  {
    const int l6 = min(lane, 5);
    const long c = (bn * ch + iy + l6 / 3) * cw + ix + l6 % 3;
    const int sv = off[c] + boff[c / kScanTile];
    beg0 = __shfl(sv, 0, 64);
    mid0 = __shfl(sv, 1, 64);
    n0 = __shfl(sv, 2, 64) - beg0;
    beg1 = __shfl(sv, 3, 64);
    mid1 = __shfl(sv, 4, 64);
    tot = n0 + __shfl(sv, 5, 64) - beg1;
  }
  if (!pix_ok) tot = 0;

  for (int c0 = 0; c0 < C; c0 += 16) {           // channel chunk
    const int nq = min(4, (C - c0) / 4);         // float4 pieces of this chunk
    // the pixel's features of this chunk, wave-uniform: one load, then scalar broadcasts
    const float ftv = (lane < 16 && c0 + lane < C) ? ldf(feat, (bn * C + c0 + lane) * HW + pix) : 0.f;
    float ft[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
      ft[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ftv), c));
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;

    for (int k0 = ws * 64; k0 < tot; k0 += wpp * 64) {
      const int k = k0 + lane;
      const bool in = k < tot;
      const int kc = min(k, tot - 1);
      const bool row0 = kc < n0;
      const long pos = row0 ? (long) beg0 + kc : (long) beg1 + (kc - n0);
      const int vox = ids[pos];
      // the voxel's row of grad_out / (hits + 1e-6)
      const float4* e = table + ((long) b * V + vox) * (C / 4) + c0 / 4;
      float gs[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) g4 = e[q];                    // uniform branch
        gs[4 * q] = g4.x; gs[4 * q + 1] = g4.y; gs[4 * q + 2] = g4.z; gs[4 * q + 3] = g4.w;
      }
      // the pair's taps: fractional coordinates (w1 of each axis) and the lower depth plane; the weights
      // of the lower taps are taken as 1 - w1 (the forward's (floor + 1) - f up to an ulp: gradients are
      // held to 1e-4)
      const float4 rc = recs[pos];
      const int iz0 = __float_as_int(rc.w);
      const float wz1v = rc.z, wz0v = 1.0f - rc.z;
      // which of the pair's taps this pixel is: (iy - iy0) * 2 + (ix - ix0)
      const int j = (row0 ? 2 : 0) + ((pos >= (row0 ? mid0 : mid1)) ? 0 : 1);
      const float wj = in ? (((j & 2) ? rc.y : 1.0f - rc.y) * ((j & 1) ? rc.x : 1.0f - rc.x)) : 0.f;
      // depth interpolated at this pixel: the two planes around the projected depth (zero padding)
      const bool z0in = iz0 >= 0 && iz0 < D, z1in = iz0 + 1 >= 0 && iz0 + 1 < D;
      float dep;
      if (D > 0) dep = (z0in ? wz0v * dcol[min(max(iz0, 0), D - 1)] : 0.f) + (z1in ? wz1v * dcol[min(max(iz0 + 1, 0), D - 1)] : 0.f);
      else dep = (iz0 == 0 ? wz0v : 0.f) + (iz0 == -1 ? wz1v : 0.f);      // D == 1: the single plane
      const float pwj = wj * dep;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        acc[c] = __builtin_fmaf(pwj, gs[c], acc[c]);
        dot = __builtin_fmaf(ft[c], gs[c], dot);
      }
      if (D > 0 && in) {
        const float wd = wj * dot;
        // (ds_add_f32 is slow on gfx950 -- ~190 cycles per wave instruction against ~11 for
        // ds_add_u32, tools/microbench/lds_atomic.hip -- but a fixed-point column with a per-pixel
        // scale measured slower here, 80 vs 69 us: finding the scale costs a wave reduction per batch)
        if (wd != 0.f) {
          if (z0in) atomicAdd(gcol + iz0, wz0v * wd);
          if (z1in) atomicAdd(gcol + iz0 + 1, wz1v * wd);
        }
      }
    }
    // fold the 64 lanes' partial sums: afterwards lane l holds channel c0 + (l >> 2)
    int cb = 0;
    reduce_halving<16, 32, 64, 16>(acc, lane, cb);
    static_assert(reduce_left<16, 32>() == 1 && reduce_dups<16, 32>() == 3, "16 values over 64 lanes");
    if (wpp == 1) {
      if (pix_ok && (lane & 3) == 0 && c0 + cb < C) gfeat[(bn * C + c0 + cb) * HW + pix] = acc[0];
    } else {
      if ((lane & 3) == 0) accbuf[wv * 16 + cb] = acc[0];
      __syncthreads();
      if (ws == 0 && lane < 16 && pix_ok && c0 + lane < C) {
        float v = 0.f;
        for (int s2 = 0; s2 < wpp; ++s2) v += accbuf[(wv + s2) * 16 + lane];
        gfeat[(bn * C + c0 + lane) * HW + pix] = v;
      }
      __syncthreads();
    }
  }
  if (D > 0 && gdepth) {
    __syncthreads();
    if (softmax_bwd) {
      // VAMP_LIFTBWD_LOGITS: the depth column is softmax(logits) (base_vampire2.py:550) and the caller
      // wants the gradient of the logits, p * (g - sum_d p g): both columns sit in LDS, so the
      // softmax backward costs one wave reduction per pixel and no pass over HBM
      if (ws == 0) {
        float dot = 0.f;
        for (int dz = lane; dz < D; dz += 64) dot = __builtin_fmaf(dcol[dz], gcol[dz], dot);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        if (lane == 0) accbuf[pw] = dot;
      }
      __syncthreads();
    }
    // consecutive threads = consecutive pixels of one depth plane
    for (int e = tid; e < D * ppb; e += nw * 64) {
      const int dz = e / ppb, p = e % ppb;
      const long pp = pid0 + p;
      if (pp >= npix) continue;
      float v = gd[p * Dp + dz];
      if (softmax_bwd) v = dcolumns[p * Dp + dz] * (v - accbuf[p]);
      gdepth[((pp / HW) * P.D + dz) * HW + pp % HW] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct LiftCellWs {
  int *cnt, *off, *bsum, *boff, *aux;
  unsigned* amask;                   // [B * V] cameras each voxel is valid for (N <= 32)
  int* ids;                          // [cap] voxel index (within its sample) of every pair, in cell order
  float4* recs;                      // [cap] {wx1, wy1, wz1, iz0} of every pair
  float4* table;                     // [B * V, C] grad_out / (hits + 1e-6), channel-last
  size_t bytes;
};

static LiftCellWs lift_cell_ws(const VampLiftDesc* d, void* scratch) {
  const LiftCells g = lift_cells(d);
  const long ntile = g.ncell / kScanTile;
  // every (voxel, camera) pair can be valid
  const size_t cap = (size_t) d->B * d->N * d->Z * d->Y * d->X;
  char* p = static_cast<char*>(scratch);
  LiftCellWs w;
  w.cnt = reinterpret_cast<int*>(p); p += align_up((size_t) (g.ncell + kScanPad) * sizeof(int), 256);   // + the scan's ticket word
  w.off = reinterpret_cast<int*>(p); p += align_up((size_t) g.ncell * sizeof(int), 256);
  w.bsum = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.boff = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.aux = reinterpret_cast<int*>(p); p += align_up((size_t) (ntile + 4) * sizeof(int), 256);
  w.amask = reinterpret_cast<unsigned*>(p); p += align_up((size_t) d->B * d->Z * d->Y * d->X * sizeof(unsigned), 256);
  w.ids = reinterpret_cast<int*>(p); p += align_up(cap * sizeof(int), 256);
  w.recs = reinterpret_cast<float4*>(p); p += align_up(cap * sizeof(float4), 256);
  w.table = reinterpret_cast<float4*>(p); p += align_up((size_t) d->B * d->Z * d->Y * d->X * d->C * sizeof(float), 256);
  w.bytes = (size_t) (p - static_cast<char*>(scratch));
  return w;
}

size_t lift_bwd_cell_ws_bytes(const VampLiftDesc* d) { return lift_cell_ws(d, nullptr).bytes; }

// count + scan: geometry only, so the host may run it beside the forward (vamp_lift_prepare)
int launch_lift_cell_prepare(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, void* scratch, hipStream_t s) {
  const LiftParams P = to_params(d);
  const LiftCells g = lift_cells(d);
  const LiftCellWs w = lift_cell_ws(d, scratch);
  const size_t cap = (size_t) d->B * d->N * d->Z * d->Y * d->X;
  VAMP_REQUIRE(cap < 0x7fffffffu && g.ncell < 0x7fffffffL, "pair / cell count exceeds 2^31");
  VAMP_REQUIRE(d->C % 4 == 0, "C must be a multiple of 4");
  if (int ze = launch_zero(w.cnt, (size_t) (g.ncell + kScanPad) * sizeof(int), s)) return ze;
  dim3 grid((d->X + 63) / 64, (d->Y + 3) / 4, d->Z * d->B);
  VAMP_TIMED(kProfLiftBwdCount, s, (lift_bwd_cell_kernel<float, 16, false><<<grid, 256, 0, s>>>(
      P, g.cw, g.ch, mats, xs, ys, zs, nullptr, nullptr, nullptr, w.cnt, w.off, w.boff, w.ids, w.recs, w.table, d->N <= 32 ? w.amask : nullptr, 0, d->B * d->N)));
  if (int e = check_launch("lift_bwd_cell_kernel<count>")) return e;
  return launch_cell_scan(w.cnt, w.off, w.bsum, w.boff, w.aux, g.ncell, s);
}

template <typename T>
static int launch_cell_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                         const float* xs, const float* ys, const float* zs, const void* depth,
                         const void* feat, const float* gout, const uint64_t* hits, float* gdepth,
                         float* gfeat, void* scratch, bool cells_valid, int wpp_force, int half, bool softmax_bwd,
                         hipStream_t s) {
  const LiftCells g = lift_cells(d);
  const LiftCellWs w = lift_cell_ws(d, scratch);
  const size_t cap = (size_t) d->B * d->N * d->Z * d->Y * d->X;
  if (!cells_valid)
    if (int e = launch_lift_cell_prepare(d, mats, xs, ys, zs, scratch, s)) return e;
  // (the counters are the fill cursors: the scan left them at zero)
  // half 0: all images; 1 / 2: the lower / upper half of the flattened (sample, camera) index
  const int BN = d->B * d->N;
  const int bn_lo = half == 2 ? BN / 2 : 0, bn_hi = half == 1 ? BN / 2 : BN;
  dim3 grid((d->X + 63) / 64, (d->Y + 3) / 4, d->Z * d->B);
  const T* dp = static_cast<const T*>(depth);
#define VAMP_CELL(CH)                                                                            \
  VAMP_TIMED(kProfLiftBwdFill, s, (lift_bwd_cell_kernel<T, CH, true><<<grid, 256, 0, s>>>(       \
      P, g.cw, g.ch, mats, xs, ys, zs, dp, gout, hits, w.cnt, w.off, w.boff, w.ids, w.recs, w.table,     \
      d->N <= 32 ? w.amask : nullptr, bn_lo, bn_hi)))
  if (P.C == 4) VAMP_CELL(4); else if (P.C == 8) VAMP_CELL(8); else VAMP_CELL(16);
#undef VAMP_CELL
  if (int e = check_launch("lift_bwd_cell_kernel<fill>")) return e;

  // waves per pixel from the expected records per pixel (4 taps x voxels per camera pixel)
  const long npix_all = (long) d->B * d->N * d->fH * d->fW;
  const long pix_lo = (long) bn_lo * d->fH * d->fW, pix_hi = (long) bn_hi * d->fH * d->fW;
  const long npix = pix_hi - pix_lo;
  if (npix <= 0) return VAMP_OK;
  const double per_pix = 4.0 * (double) d->B * d->Z * d->Y * d->X / (double) npix_all;
  int wpp = per_pix <= 96.0 ? 1 : (per_pix <= 768.0 ? 4 : 16);
  if (wpp_force == 1 || wpp_force == 4 || wpp_force == 16) wpp = wpp_force;
  const int nw = std::max(kMinWaves, wpp);
  const int ppb = nw / wpp;
  const int Dd = d->use_depth ? d->D : 0;
  const size_t lds = ((size_t) 2 * ppb * (Dd | 1) + (size_t) nw * 16) * sizeof(float);
  if (lds > 150 * 1024) return fail(VAMP_EINVAL, "%s: D too large for the LDS depth columns", __func__);
  const unsigned ggrid = (unsigned) ((npix + ppb - 1) / ppb);
  {
    auto k = lift_bwd_cell_gather_kernel<T>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess)
      return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);
    VAMP_TIMED(kProfLiftBwd, s, (k<<<ggrid, nw * 64, lds, s>>>(
        P, g.cw, g.ch, wpp, (d->fW % ppb == 0) ? d->fW / ppb : 0, mats, xs, ys, zs, dp, static_cast<const T*>(feat),
        w.off, w.boff, w.ids, w.recs, w.table, gdepth, gfeat, pix_lo, pix_hi, softmax_bwd ? 1 : 0)));
  }
  return check_launch("lift_bwd_cell_gather_kernel");
}

int launch_lift_bwd_cell(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, const float* gout,
                         const uint64_t* hits, float* gdepth, float* gfeat, void* scratch,
                         bool cells_valid, int wpp_force, int half, bool softmax_bwd, hipStream_t s) {
  const LiftParams P = to_params(d);
  if (d->in_dtype == VAMP_F32)
    return launch_cell_t<float>(d, P, mats, xs, ys, zs, depth, feat, gout, hits, gdepth, gfeat, scratch,
                                cells_valid, wpp_force, half, softmax_bwd, s);
  return launch_cell_t<__hip_bfloat16>(d, P, mats, xs, ys, zs, depth, feat, gout, hits, gdepth, gfeat,
                                       scratch, cells_valid, wpp_force, half, softmax_bwd, s);
}

}  // namespace vamp
