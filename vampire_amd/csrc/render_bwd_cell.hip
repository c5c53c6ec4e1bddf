// Camera-branch render backward, scatter stage as a cell list ("sort, then own"):
//
//   prepare (geometry only; the host runs it beside the forward):
//     rank  every inside sample increments the counter of its cell = the voxel-grid cube whose
//           lower corner is its floor tap, grid padded by one on the low side: (Z+1) x (Y+1) x
//           (X+1) cells per sample of the batch; the atomic returns the sample's rank in the cell
//     scan  two-level exclusive prefix sum of the cell counters -> cell start offsets
//     slot  start[cell] + rank for every inside sample
//   the per-ray pass (render_bwd_ray.hip) writes each sample's record {fx, fy, fz, - | w,
//           dL/ds0, ray, -} straight to its slot: records of a cell, and of x-neighbouring cells, are
//           contiguous
//   gather  GL lanes per voxel: the samples whose trilinear support contains voxel (x, y, z)
//           are exactly those of the 2x2x2 cells (x..x+1, y..y+1, z..z+1), i.e. four contiguous
//           record ranges; the lanes stream them, accumulate weight * dL/ds[c] in registers
//           and store every output element once.  Voxels with more than kHeavy records go to
//           a queue that a second kernel drains with one workgroup per voxel (the voxels next
//           to a camera collect thousands of samples).
//
// No search, no failed candidates: the work is the 8 * (inside samples) real contributions.
// The order of a cell's records follows the order in which the count atomics were served, so
// the fp32 sums can differ in the last bit between runs (as the reference's CUDA atomics do).
// Autograd of render_utils.py:140-141 (grid_sample backward w.r.t. the volume).
#include "render_common.hpp"
#include "cell_list.hpp"
#include "ray_plan.hpp"

#include <algorithm>

namespace vamp {

#ifndef VAMP_HEAVY
#define VAMP_HEAVY 256
#endif
constexpr int kHeavy = VAMP_HEAVY;       // records per voxel beyond which the whole-workgroup kernel runs
constexpr int kGatherLanes = 8;          // lanes per voxel of the gather
constexpr int kRunVox = 256 / kGatherLanes;   // voxels (an x-run) per gather workgroup

__device__ __forceinline__ long sample_cell(const RenderParams& P, int key, unsigned b, long ncell_b) {
  return key_to_cell(key, P.Y, P.X, b, ncell_b);
}

// ---------------------------------------------------------------------------
// prepare (geometry + the termination table; the host runs it beside the forward): one workgroup per
// 8 x 8 ray tile, the mapping of the per-ray pass -- the 64 lanes of a wave are the tile's rays at ONE
// depth index and fall into a few cells, in runs.  The depth indices at which no ray of the tile can
// be inside the volume (ray_plan.hpp: 64 % of all tile-steps at cfg-B) and those behind the tile's
// last kept sample (early ray termination) are never visited: the pass costs what the kept samples
// cost, not what the frustum holds (round 3 ran a thread per (ray, depth index): 5.7 M of them for
// 0.33 M kept samples).  Each sample's frustum point and floor taps are evaluated exactly as the
// per-ray pass will (same inline chain, same bits); inside samples are counted into their cell and
// take their rank there.  Device-scope atomics are served at the memory side on this part (~1 us),
// so each run of equal cells issues ONE atomic: the run head adds the run length and the lanes of
// the run take base + position.  The per-ray pass turns (cell, rank) into the record slot itself.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cam_cells_rank_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                      const float* __restrict__ vs, const float* __restrict__ ds,
                      int* __restrict__ cnt, int* __restrict__ RANK, long ncell_b,
                      const int* __restrict__ term, int* __restrict__ tile_se) {
  __shared__ int4 plan[kPlanMax];
  const RayId id = decode_ray_wps(P);
  const int lane = threadIdx.x & 63, sub = id.sub;
  const int S = P.D - 1;
  const long bn = id.bn;
  const float* m = mats + bn * 48;
  const float u = us[id.w], v = vs[id.h];
  const int keep = term ? term[(bn * P.fH + id.h) * P.fW + id.w] : S;
  int Se = id.live ? keep : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) Se = max(Se, __shfl_xor(Se, o, 64));
  Se = __builtin_amdgcn_readfirstlane(Se);
  // the tile's depth (its longest ray): the per-ray pass takes the deepest tiles first (cam_heavy_list_kernel sorts)
  if (threadIdx.x == 0 && id.tile_ok) tile_se[id.tile] = Se;
  const bool planned = S <= kPlanMax;                              // uniform
  PlanMask mk{~0ull, ~0ull};
  if (planned) {
    plan_tile(P, m, us, vs, ds, __builtin_amdgcn_readlane(id.w, 0), __builtin_amdgcn_readlane(id.w, 63),
              __builtin_amdgcn_readlane(id.h, 0), __builtin_amdgcn_readlane(id.h, 63), sub, plan);
    __syncthreads();
    mk = plan_mask(plan);
    mask_truncate(mk, Se);
  }
  // the active depth indices below Se, dealt to the four waves in turn
  int k = 0;
  for (int i = planned ? mask_next(mk, 0) : 0; i < (planned ? kPlanMax : Se); i = planned ? mask_next(mk, i + 1) : i + 1, ++k) {
    if ((k & 3) != sub) continue;                                  // uniform
    const bool kept = id.live && i < keep;
    if (__ballot(kept) == 0ull) continue;
    float x, y, z;
    frustum_point(m, u, v, ds[i], x, y, z);
    const VolTap tp = volume_tap(P, nan_to_num_geom(x), nan_to_num_geom(y), nan_to_num_geom(z));
    const bool valid = kept && tp.inside;
    const int key = valid ? pack_cell_key(tp.ix0, tp.iy0, tp.iz0) : 0;
    const long cell = sample_cell(P, key, (unsigned) id.b, ncell_b);
    const LaneRun r = lane_run(valid, cell, lane);
    int base = 0;
    if (r.head) base = atomicAdd(cnt + cell, r.len);
    base = __shfl(base, valid ? r.start : lane, 64);
    if (valid) RANK[(id.tile * S + i) * 64 + lane] = base + (lane - r.start);
  }
}

// U entries per lane per round (k, k + stride, ...): the record loads of a round go out
// together, then the ray-gradient rows, so a round costs two memory latencies.
template <int CP4, int U>
__device__ __forceinline__ void cell_accumulate(const CellRanges& cr, int k0, int stride,
                                                const float4* __restrict__ R,
                                                const float* __restrict__ Gcl, float fix, float fiy,
                                                float fiz, float (&acc)[CP4 * 4]) {
  constexpr int CP = CP4 * 4;
  float4 a[U], g[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = k0 + u * stride;
    const long pos = cell_pos(cr, min(k, cr.tot - 1));
    a[u] = R[2 * pos];
    g[u] = R[2 * pos + 1];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool in = k0 + u * stride < cr.tot;
    float wt = cell_tap_weight(a[u].x, fix) * cell_tap_weight(a[u].y, fiy) * cell_tap_weight(a[u].z, fiz);
    wt = in ? wt : 0.f;
    // record = {fx, fy, fz, - | w, dL/ds0, ray, -}
    const float Wv = wt * g[u].x;
    acc[0] = __builtin_fmaf(wt, g[u].y, acc[0]);
    const float4* g4 = reinterpret_cast<const float4*>(Gcl + (long) __float_as_uint(g[u].z) * CP);
#pragma unroll
    for (int c4 = 0; c4 < CP4; ++c4) {
      const float4 f = g4[c4];
      if (c4 > 0) acc[c4 * 4] = __builtin_fmaf(Wv, f.x, acc[c4 * 4]);
      acc[c4 * 4 + 1] = __builtin_fmaf(Wv, f.y, acc[c4 * 4 + 1]);
      acc[c4 * 4 + 2] = __builtin_fmaf(Wv, f.z, acc[c4 * 4 + 2]);
      acc[c4 * 4 + 3] = __builtin_fmaf(Wv, f.w, acc[c4 * 4 + 3]);
    }
  }
}

template <int CP4>
__device__ __forceinline__ void cam_heavy_drain(const RenderParams& P, const int* __restrict__ off,
                                                const int* __restrict__ boff, const float4* __restrict__ R,
                                                const float* __restrict__ Gcl, float* __restrict__ gdens,
                                                float* __restrict__ gsem, float* __restrict__ grgb,
                                                const int* __restrict__ heavy, const int* __restrict__ nheavy,
                                                long ncell_b, int accumulate, int first, int stride,
                                                float (&part)[4][CP4 * 4]);

// CGL lanes per voxel, 256 / CGL voxels (an x-run) per workgroup
template <int CP4, int CGL>
__global__ void __launch_bounds__(256, 5)
cam_bwd_cell_gather_kernel(RenderParams P, const int* __restrict__ off, const int* __restrict__ boff,
                           const float4* __restrict__ R, const float* __restrict__ Gcl,
                           float* __restrict__ gdens, float* __restrict__ gsem,
                           float* __restrict__ grgb, long ncell_b, int runs_x, int heavy_thresh,
                           int accumulate, BetaTail btail, const int* __restrict__ runs,
                           const int* __restrict__ heavy, const int* __restrict__ nheavy, unsigned heavy_blocks) {
  constexpr int CP = CP4 * 4;
  constexpr int CVPB = 256 / CGL;
  beta_tail(btail);                     // the ray pass's d beta partials (a launch of its own before round 3)
  // The first `heavy_blocks` workgroups drain the heavy-voxel list (one voxel per workgroup and turn): the
  // two jobs own disjoint voxels, so they share a launch -- the long-running voxels start first, the x-run
  // workgroups fill in behind them, and no second stream or launch is needed to overlap the two.
  if (blockIdx.x < heavy_blocks) {
    __shared__ float part[4][CP];
    cam_heavy_drain<CP4>(P, off, boff, R, Gcl, gdens, gsem, grgb, heavy, nheavy, ncell_b, accumulate,
                         (int) blockIdx.x, (int) heavy_blocks, part);
    return;
  }
  const unsigned bid = blockIdx.x - heavy_blocks;
  __shared__ float outs[CP][CVPB + 1];
  __shared__ int skip[CVPB];            // the voxel is on the heavy list: its outputs are not ours
  const int tid = threadIdx.x;
  const int g = tid / CGL, l = tid % CGL;
  // (giving each XCD a contiguous slab of x-runs instead of the round-robin deal measured 12 %
  // slower: the slabs next to the cameras carry most of the records)
  // accumulate mode: only the flagged x-runs have anything to add (cam_heavy_list_kernel): one
  // scalar load decides, instead of 512 offset loads and a barrier
  if (runs && runs[bid] == 0) return;
  const unsigned lin = bid;
  const int bx = lin % (unsigned) runs_x;
  const unsigned rest = lin / (unsigned) runs_x;
  const int ix = bx * CVPB + g, iy = rest % (unsigned) P.Y;
  const int zb = rest / (unsigned) P.Y;
  const int iz = zb % P.Z, b = zb / P.Z;
  const bool vox_ok = ix < P.X;
  const int nch = 1 + P.K + 3;
  const int ixc = min(ix, P.X - 1);
  const CellRanges cr = cell_ranges<CGL>(P.Y, P.X, off, boff, ncell_b, b, ixc, iy, iz, l);
  // An x-run none of whose voxels has a record for this kernel (nothing sampled there: behind a
  // terminated ray, outside every frustum -- or all on the heavy list) has nothing to add: when the
  // buffers already hold the BEV branch's gradient the workgroup is done before it touches them.
  if (accumulate && !__syncthreads_or(vox_ok && cr.tot > 0 && cr.tot <= heavy_thresh)) return;

  // output elements this thread stores at the end; with accumulate their current values (the
  // BEV branch's gradient) are fetched now, so that the load overlaps the record streaming
  const long V = (long) P.Z * P.Y * P.X;
  const long vox0 = ((long) iz * P.Y + iy) * P.X + (long) bx * CVPB;
  constexpr int PE = (CP * CVPB + 255) / 256;
  float* optr[PE];
  float prevv[PE];
#pragma unroll
  for (int i = 0; i < PE; ++i) {
    const int e = tid + i * 256;
    const int c = e / CVPB, gx = e % CVPB;
    float* o = (c == 0) ? gdens + (long) b * V
               : (c <= P.K) ? gsem + ((long) b * P.K + (c - 1)) * V
                            : grgb + ((long) b * 3 + (c - 1 - P.K)) * V;
    optr[i] = (e < nch * CVPB && bx * CVPB + gx < P.X) ? o + vox0 + gx : nullptr;
    prevv[i] = (accumulate && optr[i]) ? *optr[i] : 0.f;
  }
  if (l == 0) skip[g] = vox_ok && cr.tot > heavy_thresh;

  float acc[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) acc[c] = 0.f;

  // (voxels with more records than the threshold are on the heavy list, built with the cell lists:
  // cam_heavy_list_kernel; the whole-workgroup kernel owns their outputs and may run beside this one)
  if (vox_ok && cr.tot <= heavy_thresh) {
    const float fix = (float) ix, fiy = (float) iy, fiz = (float) iz;
    constexpr int U = 1;              // U = 2 costs a wave of occupancy and measured slower
    for (int k = l; k < cr.tot; k += U * CGL)
      cell_accumulate<CP4, U>(cr, k, CGL, R, Gcl, fix, fiy, fiz, acc);
  }
  // reduce over the lanes of the voxel, transpose through LDS, store x-runs
  {
    int cbase = 0;
    reduce_halving<CP, CGL / 2, CGL, CP>(acc, l, cbase);
    constexpr int NL = reduce_left<CP, CGL / 2>();
    constexpr int DUP = reduce_dups<CP, CGL / 2>();
    if ((l & DUP) == 0) {
#pragma unroll
      for (int c = 0; c < NL; ++c) outs[cbase + c][g] = acc[c];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PE; ++i) {
    const int e = tid + i * 256;
    if (optr[i] && !skip[e % CVPB]) *optr[i] = prevv[i] + outs[e / CVPB][e % CVPB];
  }
}

// One workgroup per queued voxel: 256 lanes stream its records, fixed-order reduction.  `first` / `stride`:
// the workgroup's first item and the number of workgroups draining the list.
template <int CP4>
__device__ __forceinline__ void cam_heavy_drain(const RenderParams& P, const int* __restrict__ off,
                                                const int* __restrict__ boff, const float4* __restrict__ R,
                                                const float* __restrict__ Gcl, float* __restrict__ gdens,
                                                float* __restrict__ gsem, float* __restrict__ grgb,
                                                const int* __restrict__ heavy, const int* __restrict__ nheavy,
                                                long ncell_b, int accumulate, int first, int stride,
                                                float (&part)[4][CP4 * 4]) {
  constexpr int CP = CP4 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nch = 1 + P.K + 3;
  const long V = (long) P.Z * P.Y * P.X;
  const int n = *nheavy;
  for (int item = first; item < n; item += stride) {
    int vid = heavy[item];
    const int ix = vid % P.X; vid /= P.X;
    const int iy = vid % P.Y; vid /= P.Y;
    const int iz = vid % P.Z, b = vid / P.Z;
    const CellRanges cr = cell_ranges<64>(P.Y, P.X, off, boff, ncell_b, b, ix, iy, iz, lane);
    const long vox = ((long) iz * P.Y + iy) * P.X + ix;
    float* optr = nullptr;
    if (tid < nch)
      optr = (tid == 0) ? gdens + (long) b * V + vox
             : (tid <= P.K) ? gsem + ((long) b * P.K + (tid - 1)) * V + vox
                            : grgb + ((long) b * 3 + (tid - 1 - P.K)) * V + vox;
    const float prev = (accumulate && optr) ? *optr : 0.f;
    float acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = 0.f;
    const float fix = (float) ix, fiy = (float) iy, fiz = (float) iz;
    constexpr int U = 1;
    for (int k = tid; k < cr.tot; k += U * 256)
      cell_accumulate<CP4, U>(cr, k, 256, R, Gcl, fix, fiy, fiz, acc);
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
    if (optr) *optr = prev + ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
    __syncthreads();
  }
}

template <int CP4>
__global__ void __launch_bounds__(256, 5)
cam_bwd_cell_heavy_kernel(RenderParams P, const int* __restrict__ off, const int* __restrict__ boff,
                          const float4* __restrict__ R, const float* __restrict__ Gcl,
                          float* __restrict__ gdens, float* __restrict__ gsem,
                          float* __restrict__ grgb, const int* __restrict__ heavy,
                          const int* __restrict__ nheavy, long ncell_b, int accumulate) {
  __shared__ float part[4][CP4 * 4];
  cam_heavy_drain<CP4>(P, off, boff, R, Gcl, gdens, gsem, grgb, heavy, nheavy, ncell_b, accumulate,
                       (int) blockIdx.x, (int) gridDim.x, part);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct CellWs {
  int* cnt;        // [ncell] counters
  int* off;        // [ncell] tile-local exclusive offsets
  int* bsum;       // [ntile] tile totals
  int* boff;       // [ntile] exclusive scan of the tile totals
  int* aux;        // [ntile] scratch of the level-2 scan, then [ntile] = total, [ntile+1] = heavy count
  int* heavy;      // [voxels] queue
  int* runs;       // [x-runs] 1 = the gather's workgroup has something to add (accumulate mode)
  int* rank;       // [tiles][S][64] rank of the sample inside its cell (written for kept inside samples only)
  int* slot;       // [tiles][S][64] slot in R (-1 = masked): the per-ray pass's note between its two loops
  int* tile_se;    // [tiles] kept samples of the tile's longest ray
  int* tile_order; // [tiles] tiles sorted by that, longest first
  float4* R;       // [samples][2] records in cell order
  size_t bytes;
};

static CellWs cell_ws(const VampRenderDesc* d, void* scratch) {
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const long ntile = ncell / kScanTile;
  const size_t samples = (size_t) d->B * d->N * (d->D - 1) * d->fH * d->fW;
  // per-sample tables are indexed by 8 x 8 ray tile (ragged tiles padded)
  const size_t tsamples = (size_t) d->B * d->N * ((d->fH + 7) / 8) * ((d->fW + 7) / 8) * 64 * (d->D - 1);
  const size_t voxels = (size_t) d->B * d->Z * d->Y * d->X;
  char* p = static_cast<char*>(scratch);
  CellWs w;
  w.cnt = reinterpret_cast<int*>(p); p += align_up((size_t) (ncell + kScanPad) * sizeof(int), 256);   // + the scan's ticket word
  w.off = reinterpret_cast<int*>(p); p += align_up((size_t) ncell * sizeof(int), 256);
  w.bsum = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.boff = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.aux = reinterpret_cast<int*>(p); p += align_up((size_t) (ntile + 4) * sizeof(int), 256);
  w.heavy = reinterpret_cast<int*>(p); p += align_up(voxels * sizeof(int), 256);
  w.runs = reinterpret_cast<int*>(p); p += align_up((size_t) d->B * d->Z * d->Y * ((d->X + kRunVox - 1) / kRunVox) * sizeof(int), 256);
  w.rank = reinterpret_cast<int*>(p); p += align_up(tsamples * sizeof(int), 256);
  w.slot = reinterpret_cast<int*>(p); p += align_up(tsamples * sizeof(int), 256);
  const size_t tiles = tsamples / ((size_t) 64 * (d->D - 1));
  w.tile_se = reinterpret_cast<int*>(p); p += align_up(tiles * sizeof(int), 256);
  w.tile_order = reinterpret_cast<int*>(p); p += align_up(tiles * sizeof(int), 256);
  w.R = reinterpret_cast<float4*>(p); p += align_up(samples * 2 * sizeof(float4), 256);
  w.bytes = (size_t) (p - static_cast<char*>(scratch));
  return w;
}

size_t cam_bwd_cell_bytes(const VampRenderDesc* d) { return cell_ws(d, nullptr).bytes; }

// pointers the per-ray pass needs
CamCellRefs cam_cell_refs(const VampRenderDesc* d, void* scratch) {
  const CellWs w = cell_ws(d, scratch);
  CamCellRefs r;
  r.rank = w.rank; r.slot = w.slot; r.off = w.off; r.boff = w.boff; r.R = w.R;
  r.tile_order = w.tile_order;
  r.ncell_b = (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1);
  return r;
}

CamRankRefs cam_rank_refs_cells(const VampRenderDesc* d, void* scratch) {
  const CellWs w = cell_ws(d, scratch);
  return CamRankRefs{w.cnt, w.rank, w.tile_se, (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1)};
}
int launch_cam_cells_zero(const VampRenderDesc* d, void* scratch, hipStream_t s) {
  const CellWs w = cell_ws(d, scratch);
  return launch_zero(w.cnt, (size_t) (cell_count_padded(d->B, d->Z, d->Y, d->X) + kScanPad) * sizeof(int), s);
}

// rank -> scan -> slot.  Depends on (d, mats, us, vs, ds) only.
static int launch_cam_heavy_list(const VampRenderDesc* d, const RenderParams& P, const CellWs& w, hipStream_t s);

// phase 0: everything; 1: rank + scan (what needs the geometry and the termination table);
// 2: heavy list (needs the scan only) -- a caller may leave phase 2 to the backward; 3: scan + heavy list behind a
// forward that has drawn the ranks itself (mats .. term unused)
int launch_cam_cells_prepare(const VampRenderDesc* d, const RenderParams& P, const float* mats,
                             const float* us, const float* vs, const float* ds, void* scratch,
                             const int* term, int phase, hipStream_t s, bool counters_clean) {
  const CellWs w = cell_ws(d, scratch);
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const size_t samples = (size_t) d->B * d->N * (d->D - 1) * d->fH * d->fW;
  const size_t voxels = (size_t) d->B * d->Z * d->Y * d->X;
  VAMP_REQUIRE(samples > 0 && samples < 0x7fffffffu && voxels < 0x7fffffffu && ncell < 0x7fffffffL,
               "sample / voxel / cell count exceeds 2^31");
  const long ncell_b = (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1);
  if (phase == 3) {
    if (int e = launch_cell_scan(w.cnt, w.off, w.bsum, w.boff, w.aux, ncell, s)) return e;
    return launch_cam_heavy_list(d, P, w, s);
  }
  if (phase != 2) {
    if (!counters_clean) {          // (VAMP_CAMPREP_COUNTERS_CLEAN: the previous scan left them at zero)
      if (int ze = launch_zero(w.cnt, (size_t) (ncell + kScanPad) * sizeof(int), s)) return ze;
    } else if (int e = debug_expect_range(w.cnt, (size_t) (ncell + kScanPad), 0, 0, s,
                                          "VAMP_CAMPREP_COUNTERS_CLEAN: the render workspace's cell counters are zero")) {
      return e;
    }
    // (a termination table handed over with *_TERM_VALID holds a number of kept samples per ray)
    if (int e = debug_expect_range(term, (size_t) d->B * d->N * d->fH * d->fW, 0, d->D - 1, s,
                                   "VAMP_CAMPREP_TERM_VALID: the workspace holds a termination table")) return e;
    VAMP_TIMED(kProfCamBwdCount, s, (cam_cells_rank_kernel<<<ray_grid<4>(P), 256, 0, s>>>(
        P, mats, us, vs, ds, w.cnt, w.rank, ncell_b, term, w.tile_se)));
    if (int e = check_launch("cam_cells_rank_kernel")) return e;
    if (int e = launch_cell_scan(w.cnt, w.off, w.bsum, w.boff, w.aux, ncell, s)) return e;
  }
  if (phase == 1) return VAMP_OK;
  return launch_cam_heavy_list(d, P, w, s);
}

// Heavy list: the voxels whose eight cells hold more than kHeavy records, known as soon as the
// cells are scanned -- so the list belongs to the prepare pass, and the kernel that drains it can
// run beside the gather instead of behind it.  The same walk flags the gather's x-runs that hold at
// least one voxel with records of its own: when the gather adds on top of the BEV branch's gradient
// (the default) the workgroups of the other runs leave at once -- with early ray termination most of
// the volume lies behind terminated rays.  (A compacted list of the runs instead of flags: 22 400
// appends to one counter took 125 us when nothing terminates.)  Thread = voxel in the gather's own
// (run, voxel) order.
__global__ void __launch_bounds__(256)
cam_heavy_list_kernel(RenderParams P, const int* __restrict__ off, const int* __restrict__ boff,
                      int* __restrict__ heavy, int* __restrict__ nheavy, int* __restrict__ runs,
                      long ncell_b, long total_runs, int runs_x, int thresh,
                      const int* __restrict__ tile_se, int* __restrict__ tile_order, int ntiles) {
  // A duty of the first workgroup: the order in which the per-ray pass takes the ray tiles -- deepest
  // first (counting sort by the bit length of the tile's longest ray).  With early ray termination a few
  // tiles hold a ray that never saturates and march 85 samples where the others march 8: started last
  // they were the kernel's tail (23 tiles of 1 056, 42 us each, in a kernel of 56 us).
  if (blockIdx.x == 0) {
    __shared__ int cls[34];
    if (threadIdx.x < 34) cls[threadIdx.x] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += 256) atomicAdd(cls + (32 - __clz(max(tile_se[t], 0))), 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0;
      for (int k = 32; k >= 0; --k) { const int n = cls[k]; cls[k] = acc; acc += n; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += 256) tile_order[atomicAdd(cls + (32 - __clz(max(tile_se[t], 0))), 1)] = t;
  }
  const long run = (long) blockIdx.x * (256 / kRunVox) + threadIdx.x / kRunVox;
  const bool run_ok = run < total_runs;
  const long rc = run_ok ? run : total_runs - 1;
  const int bx = (int) (rc % runs_x);
  const long rest = rc / runs_x;
  const int ix = bx * kRunVox + threadIdx.x % kRunVox, iy = (int) (rest % P.Y);
  const long zb = rest / P.Y;
  const int iz = (int) (zb % P.Z);
  const long b = zb / P.Z;
  int tot = 0;
  if (run_ok && ix < P.X) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long c = b * ncell_b + ((long) (iz + (r >> 1)) * (P.Y + 1) + (iy + (r & 1))) * (P.X + 1) + ix;
      tot += (off[c + 2] + boff[(c + 2) / kScanTile]) - (off[c] + boff[c / kScanTile]);
    }
  }
  if (tot > thresh) heavy[atomicAdd(nheavy, 1)] = (int) ((((long) b * P.Z + iz) * P.Y + iy) * P.X + ix);
  static_assert(kRunVox == 32, "a run is half a wave");
  const unsigned long long any = __ballot(tot > 0 && tot <= thresh);
  const int lane = threadIdx.x & 63;
  if ((lane & 31) == 0 && run_ok) runs[run] = ((any >> lane) & 0xffffffffull) != 0ull ? 1 : 0;
}

static int launch_cam_heavy_list(const VampRenderDesc* d, const RenderParams& P, const CellWs& w, hipStream_t s) {
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const long ntile = ncell / kScanTile;
  const long ncell_b = (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1);
  const int runs_x = (d->X + kRunVox - 1) / kRunVox;
  const long total_runs = (long) runs_x * d->Y * d->Z * d->B;
  int* nheavy = w.aux + ntile + 1;              // zeroed by the scan that just ran (runtime.hip)
  VAMP_TIMED(kProfAux, s, (cam_heavy_list_kernel<<<(unsigned) ((total_runs * kRunVox + 255) / 256), 256, 0, s>>>(
      P, w.off, w.boff, w.heavy, nheavy, w.runs, ncell_b, total_runs, runs_x, kHeavy, w.tile_se, w.tile_order,
      (int) ((long) d->B * d->N * ((d->fH + 7) / 8) * ((d->fW + 7) / 8)))));
  return check_launch("cam_heavy_list_kernel");
}

#ifndef VAMP_ABL_NOHEAVY
#define VAMP_ABL_NOHEAVY 0        // (measurement build: the heavy-voxel drain is not launched -- wrong gradients near the cameras)
#endif
// per-voxel gather of the records the per-ray pass has written in cell order
int launch_cam_bwd_cell(const VampRenderDesc* d, const RenderParams& P, const float* Gcl,
                        float* gdens, float* gsem, float* grgb, void* scratch, int accumulate,
                        hipEvent_t wait_event, int parts, BetaTail btail, hipStream_t s) {
  const CellWs w = cell_ws(d, scratch);
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const long ntile = ncell / kScanTile;
  const size_t voxels = (size_t) d->B * d->Z * d->Y * d->X;
  const long ncell_b = (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1);
  const int* nheavy = w.aux + ntile + 1;
  const int* runs = accumulate ? w.runs : nullptr;   // (overwrite mode: every run is stored)

  // measured at cfg-B (gather + heavy, us): 8 lanes 145 + 56, 16 lanes 173 + 56, 32 lanes 249 + 56;
  // threshold 128 / 256 / 512 with 8 lanes: 131 + 107, 145 + 56, 159 + 40 (round 1); after early ray
  // termination 68 + 37, 68 + 34, 91 + 33
  constexpr int gl = kGatherLanes;
  const int heavy_thresh = kHeavy;
  const int vpb = 256 / gl;
  // the gradient buffers are first touched here: whoever else accumulates into them (the BEV
  // branch on another stream) must be done
  if (wait_event && hipStreamWaitEvent(s, wait_event, 0) != hipSuccess)
    return fail(VAMP_EHIP, "%s: hipStreamWaitEvent failed", __func__);
  const int runs_x = (d->X + vpb - 1) / vpb;
  const long nblk = (long) runs_x * d->Y * d->Z * d->B;
  VAMP_REQUIRE(nblk < 0x7fffffffL, "too many x-runs");
  const unsigned grid = (unsigned) nblk;
  const unsigned hgrid = (unsigned) std::min<size_t>(voxels, 8192);
  // gather and heavy drain in ONE launch when the caller asks for both in one call (the default)
  const bool merged = (parts & kCamPartGather) && (parts & kCamPartHeavy);
  const unsigned hgrid_m = (unsigned) std::min<size_t>(voxels, 2048);
  // the two kernels own disjoint voxels (the heavy list was built with the cell lists), so the
  // caller may run them on two streams: parts selects
#define VAMP_CELL(CP4)                                                                              \
  do {                                                                                              \
    if (parts & kCamPartGather)                                                                     \
      VAMP_TIMED(kProfCamBwdBrick, s, (cam_bwd_cell_gather_kernel<CP4, gl><<<grid + (merged ? hgrid_m : 0u), 256, 0, s>>>( \
          P, w.off, w.boff, w.R, Gcl, gdens, gsem, grgb, ncell_b, runs_x, heavy_thresh, accumulate, btail, runs, \
          w.heavy, nheavy, merged ? hgrid_m : 0u)));                                                \
    if ((parts & kCamPartHeavy) && !merged && !VAMP_ABL_NOHEAVY)                                    \
      VAMP_TIMED(kProfCamBwdOwn, s, (cam_bwd_cell_heavy_kernel<CP4><<<hgrid, 256, 0, s>>>(          \
          P, w.off, w.boff, w.R, Gcl, gdens, gsem, grgb, w.heavy, nheavy, ncell_b, accumulate)));   \
  } while (0)
  if (P.CP == 12) VAMP_CELL(3); else if (P.CP == 24) VAMP_CELL(6); else VAMP_CELL(8);
#undef VAMP_CELL
  return check_launch("cam_bwd_cell_gather_kernel");
}

}  // namespace vamp
