// 3x3x3 / stride 1 / pad 1 Conv3d of the UNet between lift and render, gfx950:
// nn.Conv3d(cin, cout, 3, 1, 1, bias=False) with cin, cout in {16, 32}, base_vampire2.py:20, 40-60
// (init_dres, conv2, conv4, conv5, conv6 of both Hourglass3D blocks), fp32 in / fp32 out.
//
// Implicit GEMM on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, one
// rounding per accumulate -- the same numerics as an fmaf chain):
//     out[voxel][co] = sum_{tap, ci} in[voxel + tap][ci] * w[co][ci][tap]
// M = voxels (16 consecutive x per MFMA), N = 16 output channels, K = 4 input channels per step.
//   lane l supplies A[voxel l & 15][ci k0 + (l >> 4)] straight from the NCDHW input (16 consecutive
//   floats of 4 channel planes; the 27 taps re-read the same lines from L1) and
//   B[ci k0 + (l >> 4)][co l & 15] from an LDS image of the weights laid out [tap][ci][co].
//   A wave owns 64 consecutive x of one (z, y) row x all output channels: 4 x (cout / 16)
//   accumulator tiles, every B value feeds 4 MFMAs and every A value cout / 16.
// The data gradient is the same kernel on the flipped, transposed weights (FLIP).
// The weight gradient is a GEMM over the voxels: dw[co][ci][tap] = sum_voxel dout[voxel][co] *
//   in[voxel + tap][ci]: M = co, N = ci, K = 4 voxels per step; rows are staged in LDS and the
//   per-workgroup partial sums are added up by a second kernel (see conv3d_wgrad_kernel).
// Work is priced against the fp32 matrix peak (157 TFLOP/s dense): this is the one MFMA-bound
// piece near the path.
#include "common.hpp"

#include <cstdlib>

namespace vamp {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
  int B, Z, Y, X;
};

// DPP moves inside the 16-lane rows of a wave (a row = the 16 x positions of one K channel)
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {          // lane i <- lane (i - N) mod 16
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shr1(float old, float v) {   // lane i <- lane i - 1; lane 0 keeps old
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shl1(float old, float v) {   // lane i <- lane i + 1; lane 15 keeps old
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x101, 0xf, 0xf, false));
}

// ---------------------------------------------------------------------------
// forward / data gradient
// ---------------------------------------------------------------------------
// One wave's tile: NM (2 .. 4) M tiles = 16 NM consecutive x of the row (b, z, y) x all output
// channels.  A row of T = ceil(X / 16) M tiles is cut into ceil(T / 4) wave tiles of nearly equal
// size (X = 200: 13 M tiles = 4 + 3 + 3 + 3; four 64-wide tiles would compute 16).
template <int CIN, int COUT, int NM>
__device__ __forceinline__ void conv_tile(const ConvParams& P, const float* __restrict__ inb,
                                          const float* __restrict__ wl, float* __restrict__ ob, int z,
                                          int y, int x0, int li, int lk, long plane) {
  constexpr int NT = COUT / 16;
  f32x4 acc[NM][NT];
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the three x-taps of a (dz, dy) row read the same 16 NM + 2 values: they are loaded once and
  // the shifted operands come from neighbouring lanes (the load unit, not the matrix core, limits
  // this kernel when every MFMA has its own global load)
  bool ok[NM];
  int xo[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int xx = x0 + m * 16 + li;
    ok[m] = xx < P.X;
    xo[m] = ok[m] ? xx : 0;
  }
  // lane li == 0 also fetches x0 - 1, lane li == 15 fetches x0 + 16 NM
  const int xe = li == 0 ? x0 - 1 : x0 + 16 * NM;
  const bool oke = (li == 0 || li == 15) && xe >= 0 && xe < P.X;
  for (int p = 0; p < 9; ++p) {
    const int dz = p / 3 - 1, dy = p % 3 - 1;
    const int zz = z + dz, yy = y + dy;
    if (zz < 0 || zz >= P.Z || yy < 0 || yy >= P.Y) continue;          // wave-uniform
    const float* rowp = inb + ((long) zz * P.Y + yy) * P.X;
    const float* wt = wl + p * 3 * CIN * COUT;
#pragma unroll 2
    for (int k0 = 0; k0 < CIN; k0 += 4) {
      const float* cp = rowp + (long) (k0 + lk) * plane;
      float c[NM];
#pragma unroll
      for (int m = 0; m < NM; ++m) c[m] = ok[m] ? cp[xo[m]] : 0.f;
      const float e = oke ? cp[xe] : 0.f;
      // x - 1 / x + 1 operands from the neighbouring lanes of the 16-lane row (DPP, no LDS):
      // row_shr / row_shl by one, the lane at the end of the row keeps `old`, which is the
      // rotated neighbour tile's edge value (or the halo value e)
      float lf[NM], rt[NM];
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float u0 = m > 0 ? dpp_row_ror<1>(c[m > 0 ? m - 1 : 0]) : e;
        lf[m] = dpp_row_shr1(u0, c[m]);
        const float u1 = m < NM - 1 ? dpp_row_ror<15>(c[m < NM - 1 ? m + 1 : NM - 1]) : e;
        rt[m] = dpp_row_shl1(u1, c[m]);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float bv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
          bv[n] = wt[(d * CIN + k0 + lk) * COUT + (COUT == 32 ? ((n * 16 + li) ^ ((lk & 1) << 4)) : n * 16 + li)];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const float av = d == 0 ? lf[m] : (d == 1 ? c[m] : rt[m]);
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
  }
  // C/D: col (output channel) = lane & 15, row (voxel) = (lane >> 4) * 4 + reg
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int xb = x0 + m * 16 + lk * 4;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float* op = ob + (long) (n * 16 + li) * plane + xb;
      if (xb + 3 < P.X && ((P.X & 3) == 0)) {
        *reinterpret_cast<f32x4*>(op) = acc[m][n];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (xb + r < P.X) op[r] = acc[m][n][r];
      }
    }
  }
}

template <int CIN, int COUT, bool FLIP>
__global__ void __launch_bounds__(512)
conv3d_fwd_kernel(ConvParams P, const float* __restrict__ in, const float* __restrict__ w,
                  float* __restrict__ out, int tiles_x, long ntiles) {
  extern __shared__ float wl[];                       // [27][CIN][COUT]
  // weights -> LDS.  forward: wl[t][ci][co] = w[co][ci][t];
  // data gradient (the roles of the channels swap): wl[t][k][j] = w[co = k][ci = j][26 - t]
  // (COUT == 32: the two 16-column halves of odd rows are swapped, so that the four K rows a
  // load touches fall on both halves of the 32 banks)
  for (int e = threadIdx.x; e < 27 * CIN * COUT; e += blockDim.x) {
    const int j = e % COUT, k = (e / COUT) % CIN, t = e / (COUT * CIN);
    const int js = COUT == 32 ? (j ^ ((k & 1) << 4)) : j;
    wl[(t * CIN + k) * COUT + js] = FLIP ? w[((long) k * COUT + j) * 27 + (26 - t)] : w[((long) j * CIN + k) * 27 + t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const long plane = (long) P.Z * P.Y * P.X;
  const int nw = blockDim.x >> 6;
  // M tiles of a row, dealt to tiles_x wave tiles: the first `rem` take base + 1, the rest base
  const int T = (P.X + 15) / 16, base = T / tiles_x, rem = T % tiles_x;
  for (long tile = (long) blockIdx.x * nw + wave; tile < ntiles; tile += (long) gridDim.x * nw) {
    const int tx = (int) (tile % tiles_x);
    const long row = tile / tiles_x;                  // (b, z, y)
    const int y = (int) (row % P.Y), z = (int) ((row / P.Y) % P.Z), b = (int) (row / ((long) P.Y * P.Z));
    const int x0 = 16 * (tx * base + min(tx, rem));
    const int nm = base + (tx < rem ? 1 : 0);         // wave-uniform
    const float* inb = in + (long) b * CIN * plane;
    float* ob = out + (long) b * COUT * plane + ((long) z * P.Y + y) * P.X;
    if (nm == 4) conv_tile<CIN, COUT, 4>(P, inb, wl, ob, z, y, x0, li, lk, plane);
    else if (nm == 3) conv_tile<CIN, COUT, 3>(P, inb, wl, ob, z, y, x0, li, lk, plane);
    else if (nm == 2) conv_tile<CIN, COUT, 2>(P, inb, wl, ob, z, y, x0, li, lk, plane);
    else conv_tile<CIN, COUT, 1>(P, inb, wl, ob, z, y, x0, li, lk, plane);
  }
}

// ---------------------------------------------------------------------------
// weight gradient: dw[co][ci][t] = sum_{b, voxel} dout[b][co][voxel] * in[b][ci][voxel + tap t]
// MFMA: M = co (16 per tile), N = ci (16 per tile), K = 4 consecutive x:
//   A[co l & 15][x k0 + (l >> 4)] = dout,  B[x k0 + (l >> 4)][ci l & 15] = in shifted by the tap.
// Both operands are channel-strided in NCDHW memory, so a workgroup stages rows in LDS with
// coalesced loads.  A work item is (b, z, dz, a chunk of consecutive y): it walks y with the dout
// row (z, y) and a RING of three input rows (z + dz, y - 1 .. y + 1) in LDS -- moving to the next
// y brings in one new dout row and one new input row (the first cut re-staged nine input rows per
// output row: 483 us for 32 -> 16 at 16x200x200), and the rows of the next y are loaded into
// registers before the MFMA loop of the current one, so their latency is covered.  The images are
// [channel][XS] with XS = 4 (mod 32): lane (li, lk) reads word li * XS + x + lk, bank 4 li + lk,
// every bank exactly twice per wave -- the minimum.  A wave owns one (co tile, ci tile) pair and a
// phase of the x steps and keeps the nine accumulator tiles (dy, dx) of the item's dz; partial
// sums go to a [dz][item][9 * cout * cin] buffer that a second kernel adds up (no float atomics
// from the workgroups: 27 * cout * cin addresses would each see one per workgroup).
// ---------------------------------------------------------------------------
constexpr int kWgradThreads = 512;
constexpr int kWgradPre = 24;          // staged floats per thread and y step: (cin + cout) * X <= 24 * 512

// NPRE = staged floats per thread and y step (compile-time, so that the staging loops unroll
// into plain load / store batches): the host picks the smallest of {13, 19, 24} that covers
// (cin + cout) * X / 512.  512-thread workgroups (8 waves: twice the x phases per channel-tile
// pair) took the weight gradient from ~190 to ~140 us at 16 -> 16, 16x200x200; 1024 threads the same.
template <int CIN, int COUT, int NPRE>
__global__ void __launch_bounds__(kWgradThreads)
conv3d_wgrad_kernel(ConvParams P, const float* __restrict__ in, const float* __restrict__ dout,
                    float* __restrict__ part, int nchunks, int rows_per_chunk, int XS, int nitems) {
  extern __shared__ float lds[];
  float* G = lds;                      // [COUT][XS]     dout row, G[co][x]
  float* I = lds + COUT * XS;          // [3][CIN][XS]   input rows, slot (yy + 3) % 3, I[ci][x + 1]
  constexpr int MT = COUT / 16, NTI = CIN / 16, NG = MT * NTI;      // NG in {1, 2, 4}
  constexpr int NW = kWgradThreads / 64, NPH = NW / NG;             // x phases per group
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int grp = wave % NG, phase = wave / NG;
  const int gm = grp % MT, gn = grp / MT;
  const long plane = (long) P.Z * P.Y * P.X;
  const int steps = (P.X + 3) / 4;
  // item -> (dz, b, z, chunk)
  const int item = blockIdx.x;
  const int per_dz = nitems / 3;
  const int dzi = item / per_dz, rest = item % per_dz;
  const int chunk = rest % nchunks, bz = rest / nchunks;
  const int z = bz % P.Z, b = bz / P.Z;
  const int zz = z + dzi - 1;
  float* pb = part + (long) item * 9 * COUT * CIN;
  const int y0 = chunk * rows_per_chunk, y1 = min(P.Y, y0 + rows_per_chunk);
  f32x4 acc[3][3];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int d = 0; d < 3; ++d) acc[p][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool live = zz >= 0 && zz < P.Z && y0 < y1;                  // workgroup-uniform
  if (live) {
    const float* gbase = dout + (long) b * COUT * plane + (long) z * P.Y * P.X;
    const float* ibase = in + (long) b * CIN * plane + (long) zz * P.Y * P.X;
    const int nG = COUT * P.X, nI = CIN * P.X;
    // zero the images once: pads, halo columns and rows outside the volume stay zero
    for (int e = tid; e < (COUT + 3 * CIN) * XS; e += kWgradThreads) lds[e] = 0.f;
    __syncthreads();
    // rows y0 - 1 and y0 of the input plane
    for (int yy = y0 - 1; yy <= y0; ++yy) {
      if (yy < 0 || yy >= P.Y) continue;
      float* slot = I + ((yy + 3) % 3) * CIN * XS;
      for (int e = tid; e < nI; e += kWgradThreads) {
        const int ci = e / P.X, x = e - ci * P.X;
        slot[ci * XS + x + 1] = ibase[(long) ci * plane + (long) yy * P.X + x];
      }
    }
    // registers for the rows of the coming step: dout row y, input row y + 1.  Element e = tid +
    // i * 256 of the combined [COUT + CIN][X] rows; (channel, x) advance incrementally (a division
    // per element made the staging cost more than the MFMAs).
    const int qstep = kWgradThreads / P.X, rstep = kWgradThreads % P.X;
    const int c_first = tid / P.X, x_first = tid % P.X;
    float pre[NPRE];
    // (the loads are unconditional, from clamped addresses, and nothing touches their results
    // until commit: a select on a loaded value makes the compiler wait for it right away, and
    // the prefetch then overlaps nothing -- measured 327 us = 180 staging + 110 MFMA + 31)
    auto fetch = [&](int y) {
      int c = c_first, x = x_first;
      const float* grow = gbase + (long) y * P.X;
      const float* irow = ibase + (long) min(y + 1, P.Y - 1) * P.X - (long) COUT * plane;
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int cc = min(c, COUT + CIN - 1);
        const float* src = (cc < COUT ? grow : irow) + (long) cc * plane + x;
        pre[i] = *src;
        x += rstep; c += qstep;
        if (x >= P.X) { x -= P.X; ++c; }
      }
    };
    auto commit = [&](int y) {
      float* slot = I + ((y + 1) % 3) * CIN * XS + 1 - COUT * XS;   // slot[c * XS + x] for c >= COUT
      const bool inext = y + 1 < P.Y;
      int c = c_first, x = x_first;
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        if (c < COUT) G[c * XS + x] = pre[i];
        else if (c < COUT + CIN) slot[c * XS + x] = inext ? pre[i] : 0.f;   // zeros outside the volume
        x += rstep; c += qstep;
        if (x >= P.X) { x -= P.X; ++c; }
      }
    };
    fetch(y0);
    for (int y = y0; y < y1; ++y) {
      __syncthreads();                          // the previous step's MFMAs are done with G / the ring
      commit(y);
      __syncthreads();
      if (y + 1 < y1) fetch(y + 1);             // in flight during the MFMA loop
      const float* ga = G + (gm * 16 + li) * XS + lk;
      const float* i0 = I + (((y - 1 + 3) % 3) * CIN + gn * 16 + li) * XS + lk;
      const float* i1 = I + (((y + 3) % 3) * CIN + gn * 16 + li) * XS + lk;
      const float* i2 = I + (((y + 1) % 3) * CIN + gn * 16 + li) * XS + lk;
      for (int st = phase; st < steps; st += NPH) {
        const int xk = st * 4;
        const float av = ga[xk];
        const float* rows[3] = {i0, i1, i2};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const float b0 = rows[p][xk], b1 = rows[p][xk + 1], b2 = rows[p][xk + 2];
          acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[p][0], 0, 0, 0);
          acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[p][1], 0, 0, 0);
          acc[p][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2, acc[p][2], 0, 0, 0);
        }
      }
    }
  }
  // The row y0 - 1 slot of the ring: when y0 - 1 < 0 it was never written and is zero.  (Rows
  // beyond the chunk end were committed as zeros only if outside the volume; inside the volume
  // they are real data -- the next chunk's rows -- which is what dy = +1 of the last row needs.)
  // C/D: col = ci = lane & 15, row = co = (lane >> 4) * 4 + reg.  The NPH phases of a group are
  // folded into phase 0 through LDS, one at a time; phase 0 writes the item's partial sums.
  float* red = lds;                                     // [NG][9][64][4]
  for (int ph = 1; ph < NPH; ++ph) {
    __syncthreads();
    if (phase == ph) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          *reinterpret_cast<f32x4*>(red + ((grp * 9 + p * 3 + d) * 64 + lane) * 4) = acc[p][d];
    }
    __syncthreads();
    if (phase == 0) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          acc[p][d] += *reinterpret_cast<const f32x4*>(red + ((grp * 9 + p * 3 + d) * 64 + lane) * 4);
    }
  }
  if (phase == 0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r)      // partial layout [dy dx][co][ci]: lanes li = consecutive ci
          pb[((p * 3 + d) * COUT + gm * 16 + lk * 4 + r) * CIN + gn * 16 + li] = acc[p][d][r];
  }
}

// dw[co][ci][t] = sum over the items of tap t's dz of part[item][t % 9][co][ci]; the items are split
// over gridDim.y slices whose sums meet in dw (zeroed before) through one float atomic each
__global__ void __launch_bounds__(256)
conv3d_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int per_dz, int cin,
                           int cout) {
  const int n9 = 9 * cout * cin;
  const int e = blockIdx.x * 256 + threadIdx.x;          // over [dz][9][co][ci]
  if (e >= 3 * n9) return;
  const int dzi = e / n9, r = e - dzi * n9;
  const int k0 = (int) ((long) per_dz * blockIdx.y / gridDim.y), k1 = (int) ((long) per_dz * (blockIdx.y + 1) / gridDim.y);
  const float* p = part + ((long) dzi * per_dz) * n9 + r;
  float s0 = 0.f, s1 = 0.f;
  int k = k0;
  for (; k + 1 < k1; k += 2) {
    s0 += p[(long) k * n9];
    s1 += p[(long) (k + 1) * n9];
  }
  if (k < k1) s0 += p[(long) k * n9];
  const int ci = r % cin, co = (r / cin) % cout, t9 = r / (cin * cout);
  atomicAdd(dw + ((long) co * cin + ci) * 27 + dzi * 9 + t9, s0 + s1);
}

int wgrad_xs(int X) {                  // row pitch of the LDS images: >= 4 ceil(X / 4) + 4, = 4 mod 32
  const int need = (X + 3) / 4 * 4 + 4;
  return (need - 4 + 31) / 32 * 32 + 4;
}

// y chunks per (b, z, dz): as many items as workgroups fit on the chip at once (by LDS), not more --
// 528 items on 512 slots run a second, nearly empty round
size_t wgrad_lds(const VampConvDesc* d) {
  return std::max((size_t) (d->cout + 3 * d->cin) * wgrad_xs(d->X), (size_t) 4 * 9 * 256) * sizeof(float);
}
struct WgradPlan {
  int nchunks, rows_per_chunk, nitems;
};
WgradPlan wgrad_plan(const VampConvDesc* d) {
  const long combos = (long) d->B * d->Z * 3;
  const size_t lds = wgrad_lds(d);
  const int per_cu = lds <= 53 * 1024 ? 3 : (lds <= 80 * 1024 ? 2 : 1);
  // (two or four rounds of items measured 3 - 10 % slower than one)
  const int nchunks = (int) std::max<long>(1, std::min<long>(d->Y, 256L * per_cu / combos));
  WgradPlan p;
  p.rows_per_chunk = (d->Y + nchunks - 1) / nchunks;
  p.nchunks = (d->Y + p.rows_per_chunk - 1) / p.rows_per_chunk;
  p.nitems = (int) (combos * p.nchunks);
  return p;
}

int check(const VampConvDesc* d) {
  VAMP_REQUIRE(d != nullptr, "desc is NULL");
  VAMP_REQUIRE(d->B > 0 && d->Z > 0 && d->Y > 0 && d->X > 0, "sizes must be positive");
  VAMP_REQUIRE((d->cin == 16 || d->cin == 32) && (d->cout == 16 || d->cout == 32),
               "cin, cout must be 16 or 32");
  VAMP_REQUIRE((long) d->B * std::max(d->cin, d->cout) * d->Z * d->Y * d->X < 0x7fffffffL * 4L, "tensor too large");
  return VAMP_OK;
}

template <int CIN, int COUT, bool FLIP>
int launch_fwd(const VampConvDesc* d, const float* in, const float* w, float* out, hipStream_t s) {
  ConvParams P{d->B, d->Z, d->Y, d->X};
  const int tiles_x = ((d->X + 15) / 16 + 3) / 4;        // wave tiles per row: ceil(M tiles / 4)
  const long ntiles = (long) d->B * d->Z * d->Y * tiles_x;
  const size_t lds = (size_t) 27 * CIN * COUT * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_fwd_kernel<CIN, COUT, FLIP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    attr_set = true;
  }
  // persistent workgroups of 8 waves that share one weight image: 1 / 2 / 4 per CU by LDS size
  // (8 waves: 60-65 TFLOP/s at 16x200x200 against 52-56 with 4; small volumes have too few
  // tiles for that and keep 4)
  const int wpb = ntiles >= 1024 ? 8 : 4;
  const int per_cu = lds > 80 * 1024 ? 1 : (lds > 40 * 1024 ? 2 : 4);
  const unsigned grid = (unsigned) std::min<long>((ntiles + wpb - 1) / wpb, 256L * per_cu);
  VAMP_TIMED(FLIP ? kProfConvDgrad : kProfConvFwd, s, (conv3d_fwd_kernel<CIN, COUT, FLIP><<<grid, 64 * wpb, lds, s>>>(
      P, in, w, out, tiles_x, ntiles)));
  return check_launch("conv3d_fwd_kernel");
}

}  // namespace
}  // namespace vamp

using namespace vamp;

extern "C" {

int vamp_conv3d_forward(const VampConvDesc* d, const float* in, const float* weight, float* out,
                        void* stream) {
  if (int e = check(d)) return e;
  VAMP_REQUIRE(in && weight && out, "NULL tensor");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->cin == 16 && d->cout == 16) return launch_fwd<16, 16, false>(d, in, weight, out, s);
  if (d->cin == 16 && d->cout == 32) return launch_fwd<16, 32, false>(d, in, weight, out, s);
  if (d->cin == 32 && d->cout == 16) return launch_fwd<32, 16, false>(d, in, weight, out, s);
  return launch_fwd<32, 32, false>(d, in, weight, out, s);
}

/* grad_in [B, cin, Z, Y, X] from grad_out [B, cout, Z, Y, X]: the forward kernel on the flipped,
   transposed weights (its "input channels" are cout, its "output channels" cin) */
int vamp_conv3d_backward_data(const VampConvDesc* d, const float* grad_out, const float* weight,
                              float* grad_in, void* stream) {
  if (int e = check(d)) return e;
  VAMP_REQUIRE(grad_out && weight && grad_in, "NULL tensor");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->cin == 16 && d->cout == 16) return launch_fwd<16, 16, true>(d, grad_out, weight, grad_in, s);
  if (d->cin == 16 && d->cout == 32) return launch_fwd<32, 16, true>(d, grad_out, weight, grad_in, s);
  if (d->cin == 32 && d->cout == 16) return launch_fwd<16, 32, true>(d, grad_out, weight, grad_in, s);
  return launch_fwd<32, 32, true>(d, grad_out, weight, grad_in, s);
}

/* 1 when all three entry points accept this descriptor (channel counts, row length against the
   weight-gradient's staging registers and LDS images), else 0 */
int vamp_conv3d_supported(const VampConvDesc* d) {
  if (check(d)) return 0;
  if ((long) (d->cin + d->cout) * d->X > (long) kWgradPre * kWgradThreads) return 0;
  if (wgrad_lds(d) > 160 * 1024) return 0;
  return 1;
}

size_t vamp_conv3d_workspace_bytes(const VampConvDesc* d) {
  if (check(d)) return 0;
  return align_up((size_t) wgrad_plan(d).nitems * 9 * d->cout * d->cin * sizeof(float), 256);
}

int vamp_conv3d_backward_weight(const VampConvDesc* d, const float* in, const float* grad_out,
                                float* grad_weight, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (int e = check(d)) return e;
  VAMP_REQUIRE(in && grad_out && grad_weight && workspace, "NULL tensor");
  VAMP_REQUIRE(workspace_bytes >= vamp_conv3d_workspace_bytes(d), "workspace too small");
  VAMP_REQUIRE((long) (d->cin + d->cout) * d->X <= (long) kWgradPre * kWgradThreads,
               "row too long for the staging registers ((cin + cout) * X <= 12288)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvParams P{d->B, d->Z, d->Y, d->X};
  const WgradPlan pl = wgrad_plan(d);
  const int XS = wgrad_xs(d->X);
  const size_t lds = wgrad_lds(d);
  VAMP_REQUIRE(lds <= 160 * 1024, "row too long for the LDS images");
  float* part = static_cast<float*>(workspace);
  if (int ze = launch_zero(grad_weight, (size_t) d->cout * d->cin * 27 * sizeof(float), s)) return ze;
#define VAMP_WGRAD_N(CI, CO, NP)                                                                    \
  do {                                                                                              \
    static bool attr_set = false;                                                                   \
    if (!attr_set) {                                                                                \
      (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_kernel<CI, CO, NP>),   \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
      attr_set = true;                                                                              \
    }                                                                                               \
    VAMP_TIMED(kProfConvWgrad, s, (conv3d_wgrad_kernel<CI, CO, NP><<<pl.nitems, kWgradThreads, lds, s>>>( \
        P, in, grad_out, part, pl.nchunks, pl.rows_per_chunk, XS, pl.nitems)));                     \
  } while (0)
  const int npre = (int) (((long) (d->cin + d->cout) * d->X + kWgradThreads - 1) / kWgradThreads);
#define VAMP_WGRAD(CI, CO)                                                                          \
  do {                                                                                              \
    if (npre <= 13) VAMP_WGRAD_N(CI, CO, 13);                                                       \
    else if (npre <= 19) VAMP_WGRAD_N(CI, CO, 19);                                                  \
    else VAMP_WGRAD_N(CI, CO, kWgradPre);                                                           \
  } while (0)
  if (d->cin == 16 && d->cout == 16) VAMP_WGRAD(16, 16);
  else if (d->cin == 16 && d->cout == 32) VAMP_WGRAD(16, 32);
  else if (d->cin == 32 && d->cout == 16) VAMP_WGRAD(32, 16);
  else VAMP_WGRAD(32, 32);
#undef VAMP_WGRAD_N
#undef VAMP_WGRAD
  if (int e = check_launch("conv3d_wgrad_kernel")) return e;
  const int n = 27 * d->cout * d->cin;
  const int per_dz = pl.nitems / 3;
  const dim3 rgrid((unsigned) ((n + 255) / 256), (unsigned) std::max(1, std::min(16, per_dz / 8)));
  conv3d_wgrad_reduce_kernel<<<rgrid, 256, 0, s>>>(part, grad_weight, per_dz, d->cin, d->cout);
  return check_launch("conv3d_wgrad_reduce_kernel");
}

}  // extern "C"
