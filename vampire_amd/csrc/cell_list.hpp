// Shared pieces of the "sort, then own" backward passes (render_bwd_cell.hip, lift_bwd_cell.hip,
// sample_points.hip): contributions are counted per destination cell, the counters are scanned
// (launch_cell_scan), the records are re-laid in cell order and one owner per destination streams
// exactly the records that touch it.
#pragma once
#include "common.hpp"

namespace vamp {

// Run aggregation of the counter atomics over the lanes of a wave: adjacent lanes with `act` and
// equal `cell` form a run; the head of a run adds the run length with one atomic and the lanes
// take base + (lane - start).  (Device-scope atomics are served at the memory side, ~1 us each.)
struct LaneRun {
  bool head;
  int start, len;
};
__device__ __forceinline__ LaneRun lane_run(bool act, long cell, int lane) {
  const long pcell = __shfl_up(cell, 1, 64);
  const unsigned long long vm = __ballot(act);
  const bool pact = lane > 0 && ((vm >> (lane - 1)) & 1ull);
  LaneRun r;
  r.head = act && (!pact || pcell != cell);
  const unsigned long long hm = __ballot(r.head);
  const unsigned long long upto = ~0ull >> (63 - lane);
  r.start = 63 - __clzll((long long) (hm & upto));
  const unsigned long long brk = (hm | ~vm) & ~upto;
  const int end = brk ? __ffsll((long long) brk) - 1 : 64;
  r.len = end - r.start;
  return r;
}

// ---- 3-D cells of a [Z, Y, X] voxel grid --------------------------------------------------
// cell (cz, cy, cx) = the cube whose lower corner is floor tap (cz - 1, cy - 1, cx - 1): the grid
// is padded by one on the low side, (Z+1) x (Y+1) x (X+1) cells per sample of the batch.
inline long cell_count_padded(int B, int Z, int Y, int X) {
  const long nc = (long) B * (Z + 1) * (Y + 1) * (X + 1) + 2;      // +2: the gather reads start[c + 2]
  return (nc + kScanTile - 1) / kScanTile * kScanTile;
}

// floor taps packed 11 / 11 / 10 bits, each + 1; key 0 = no contribution
__device__ __forceinline__ int pack_cell_key(int ix0, int iy0, int iz0) {
  return (ix0 + 1) | ((iy0 + 1) << 11) | ((iz0 + 1) << 22);
}
__device__ __forceinline__ long key_to_cell(int key, int Y, int X, unsigned b, long ncell_b) {
  return (long) b * ncell_b + ((long) (key >> 22) * (Y + 1) + ((key >> 11) & 2047)) * (X + 1) + (key & 2047);
}

// weight of tap index `iv` for continuous coordinate f (aten: w0 = floor+1-f, w1 = f-floor)
__device__ __forceinline__ float cell_tap_weight(float f, float iv) {
  const float fl = floorf(f);
  return (fl == iv) ? (fl + 1.0f) - f : ((fl + 1.0f == iv) ? f - fl : 0.f);
}

// The records whose trilinear support contains voxel (ix, iy, iz) are those of the 2x2x2 cells
// (ix..ix+1, iy..iy+1, iz..iz+1): four contiguous ranges (x-neighbour cells are adjacent).
// Lanes 0..7 of a W-wide group load the range ends.
struct CellRanges {
  int beg[4];
  int pre[4];     // exclusive prefix of the range lengths
  int tot;
};

template <int W>
__device__ __forceinline__ CellRanges cell_ranges(int Y, int X, const int* __restrict__ off,
                                                  const int* __restrict__ boff, long ncell_b, int b,
                                                  int ix, int iy, int iz, int l) {
  const int r = (l >> 1) & 3;
  const long c = (long) b * ncell_b +
                 ((long) (iz + (r >> 1)) * (Y + 1) + (iy + (r & 1))) * (X + 1) + ix + 2 * (l & 1);
  const int sv = off[c] + boff[c / kScanTile];
  CellRanges cr;
  int run = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    cr.beg[q] = __shfl(sv, 2 * q, W);
    const int end = __shfl(sv, 2 * q + 1, W);
    cr.pre[q] = run;
    run += end - cr.beg[q];
  }
  cr.tot = run;
  return cr;
}

// The same four ranges WITHOUT the records of "heavy" cells (more than `thresh` records): those cells are summed
// once per cell by a kernel of their own (render_bwd_cell.hip: cam_cell_splat_kernel) and reach the voxel as eight
// per-corner partial sums.  W = 8 lanes per voxel; lane l is also the voxel's cell l = (dz, dy, dx) = (l >> 2,
// (l >> 1) & 1, l & 1), and leaves with that cell's first record and count.
struct CellRangesLight {
  CellRanges cr;      // light cells only
  int all;            // records of all eight cells
  int own_start, own_n;
};
template <int W>
__device__ __forceinline__ CellRangesLight cell_ranges_light(int Y, int X, const int* __restrict__ off,
                                                             const int* __restrict__ boff, long ncell_b, int b,
                                                             int ix, int iy, int iz, int l, int thresh) {
  static_assert(W == 8, "one lane per cell of the voxel");
  const int r = (l >> 1) & 3, j = l & 1;
  const long c0 = (long) b * ncell_b + ((long) (iz + (r >> 1)) * (Y + 1) + (iy + (r & 1))) * (X + 1) + ix;
  const long c = c0 + 2 * j;
  const int sv = off[c] + boff[c / kScanTile];                   // start(c0) for j = 0, start(c0 + 2) for j = 1
  const int mid = off[c0 + 1] + boff[(c0 + 1) / kScanTile];      // start(c0 + 1)
  CellRangesLight o;
  o.own_start = j ? mid : sv;
  o.own_n = j ? sv - mid : mid - sv;
  const int hv = o.own_n > thresh;
  int run = 0, all = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int s0 = __shfl(sv, 2 * q, W), s2 = __shfl(sv, 2 * q + 1, W), s1 = __shfl(mid, 2 * q, W);
    const int ha = __shfl(hv, 2 * q, W), hb = __shfl(hv, 2 * q + 1, W);
    const int beg = ha ? s1 : s0, end = hb ? s1 : s2;
    o.cr.beg[q] = beg;
    o.cr.pre[q] = run;
    run += end - beg;
    all += s2 - s0;
  }
  o.cr.tot = run;
  o.all = all;
  return o;
}

// record position of the k-th entry of the concatenated ranges (k clamped by the caller)
__device__ __forceinline__ long cell_pos(const CellRanges& cr, int k) {
  const int q = (k >= cr.pre[1]) + (k >= cr.pre[2]) + (k >= cr.pre[3]);
  const int pre = q == 0 ? cr.pre[0] : (q == 1 ? cr.pre[1] : (q == 2 ? cr.pre[2] : cr.pre[3]));
  const int beg = q == 0 ? cr.beg[0] : (q == 1 ? cr.beg[1] : (q == 2 ? cr.beg[2] : cr.beg[3]));
  return (long) beg + (k - pre);
}

}  // namespace vamp
