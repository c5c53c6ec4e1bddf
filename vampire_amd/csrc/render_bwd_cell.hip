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
//   splat   cells with more than kCellHeavy records (the cells next to a camera hold hundreds) are summed ONCE,
//           into eight per-corner partial rows (cam_cell_splat_kernel)
//   gather  GL lanes per voxel: the samples whose trilinear support contains voxel (x, y, z)
//           are exactly those of the 2x2x2 cells (x..x+1, y..y+1, z..z+1), i.e. four contiguous
//           record ranges; the lanes stream the records of the light cells among them, accumulate
//           weight * dL/ds[c] in registers, add the heavy cells' partial rows and store every output
//           element once.
//
// No search, no failed candidates: the work is the 8 * (inside samples) real contributions.
// The order of a cell's records follows the order in which the count atomics were served, so
// the fp32 sums can differ in the last bit between runs (as the reference's CUDA atomics do).
// Autograd of render_utils.py:140-141 (grid_sample backward w.r.t. the volume).
#include "render_common.hpp"
#include "cell_list.hpp"
#include "cam_lists.hpp"
#include "ray_plan.hpp"

#include <algorithm>

namespace vamp {

constexpr int kGatherLanes = 8;          // lanes per voxel of the gather
static_assert(kRunVox == 256 / kGatherLanes, "an x-run is what a gather workgroup owns");

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

// ---------------------------------------------------------------------------
// Heavy cells, summed ONCE per cell.  A record reaches the eight voxels at the corners of its cell with eight
// trilinear weights of the same three fractions; the per-voxel gather would stream the cell's records eight times
// (once from each corner voxel) -- and next to a camera a cell holds hundreds of records (cfg-B with early ray
// termination: 0.34 M kept samples in 3 710 cells, 94 % of them in the 2 000 cells with more than 32 records;
// tools/debug/cell_stats.py).  One workgroup per listed cell: its NW waves split the channels (CP / NW each); a
// lane is (record, z half): it takes every 32nd record and keeps the sums of the four corners of its z plane x
// CP / NW channels; the half-waves reduce them and write the cell's 8 x CP partial sums to slot start(cell) /
// kCellHeavy of the partial table (cells with more than kCellHeavy records cannot share a slot).  The gather adds
// the up to eight partial rows of a voxel in a fixed order: no float atomics, and the order of the sums inside a
// cell is the record order (the rank atomics'), as in the gather.
// The kernel is a chain of dependent fetches (list entry -> records -> gradient rows), so what counts is that every
// listed cell is resident at once: 2 000 cells x NW waves over 1 024 SIMDs -- two waves per cell (2 x 12 channels at
// cfg-B) fit at four waves per SIMD; four waves per cell with a lane per record (116 registers) ran in two batches,
// 20 us, eight waves of three channels in three, 25 us.
// ---------------------------------------------------------------------------
template <int CP4, int NW>
__global__ void __launch_bounds__(NW * 64)
cam_cell_splat_kernel(const float4* __restrict__ R, const float* __restrict__ Gcl, const int2* __restrict__ hcells,
                      const int* __restrict__ nhcells, float* __restrict__ part) {
  constexpr int CP = CP4 * 4, CW = CP / NW, NA = 4 * CW;
  static_assert(CW * NW == CP && CW % 2 == 0, "the waves split the channels evenly, in pairs at least");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  // the list entry {first record, records} goes out with the length of the list, not behind it (the list has room
  // for every cell)
  int2 ent = hcells[blockIdx.x];
  const int n = *nhcells;
  for (int item = blockIdx.x; item < n; item += gridDim.x) {
    // (sharing the cells with more than 256 records among four workgroups, each with a row set of its own, changed
    // nothing: 24.2 against 24.2 us, step 0.3945 against 0.3936 ms)
#ifdef VAMP_SPLAT_CAP       // (measurement builds: wrong sums)
    const int s0 = ent.x, s1 = ent.x + min(ent.y, VAMP_SPLAT_CAP);
#else
    const int s0 = ent.x, s1 = ent.x + ent.y;
#endif
    constexpr int U = 2;
    float4 a[U], g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long pos = min(s0 + r + u * 32, s1 - 1);
      a[u] = R[2 * pos];
      g[u] = R[2 * pos + 1];
    }
    if (item + (int) gridDim.x < n) ent = hcells[item + gridDim.x];
    float acc[NA];
#pragma unroll
    for (int e = 0; e < NA; ++e) acc[e] = 0.f;
    for (int k0 = s0 + r; k0 < s1; k0 += U * 32) {
      float gv[U][CW];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* row = Gcl + (long) __float_as_uint(g[u].z) * CP + wv * CW;
        if constexpr (CW % 4 == 0) {
#pragma unroll
          for (int q = 0; q < CW / 4; ++q) {
            const float4 f = reinterpret_cast<const float4*>(row)[q];
            gv[u][q * 4] = f.x; gv[u][q * 4 + 1] = f.y; gv[u][q * 4 + 2] = f.z; gv[u][q * 4 + 3] = f.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < CW / 2; ++q) {
            const float2 f = reinterpret_cast<const float2*>(row)[q];
            gv[u][q * 2] = f.x; gv[u][q * 2 + 1] = f.y;
          }
        }
      }
      // the records of the next round travel while this one is summed
      float4 ca[U], cg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { ca[u] = a[u]; cg[u] = g[u]; }
      if (k0 - r + U * 32 < s1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long pos = min(k0 + (U + u) * 32, s1 - 1);
          a[u] = R[2 * pos];
          g[u] = R[2 * pos + 1];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in = k0 + u * 32 < s1;
        // record = {fx, fy, fz, - | w, dL/ds0, ray, -}; the weights as cell_tap_weight forms them
        const float flx = floorf(ca[u].x), fly = floorf(ca[u].y), flz = floorf(ca[u].z);
        float wx[2] = {(flx + 1.0f) - ca[u].x, ca[u].x - flx};
        const float wy[2] = {(fly + 1.0f) - ca[u].y, ca[u].y - fly};
        const float wz = h ? ca[u].z - flz : (flz + 1.0f) - ca[u].z;
        if (!in) wx[0] = wx[1] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float wt = wx[k & 1] * wy[k >> 1] * wz;
          const float Wv = wt * cg[u].x;
#pragma unroll
          for (int q = 0; q < CW; ++q) {
            // channel 0 of the row (wave 0, q == 0) is the density: weight * dL/ds0
            if (q == 0) acc[k * CW] = wv == 0 ? __builtin_fmaf(wt, cg[u].y, acc[k * CW]) : __builtin_fmaf(Wv, gv[u][0], acc[k * CW]);
            else acc[k * CW + q] = __builtin_fmaf(Wv, gv[u][q], acc[k * CW + q]);
          }
        }
      }
    }
    // over the 32 lanes of each z half
    int cbase = 0;
#ifdef VAMP_SPLAT_NORED
    if (acc[0] != 123.f) continue;
#endif
    reduce_halving<NA, 16, 32, NA>(acc, lane, cbase);
    constexpr int NL = reduce_left<NA, 16>();
    constexpr int DUP = reduce_dups<NA, 16>();
    if ((lane & DUP) == 0) {
      float* o = part + ((long) (s0 / kCellHeavy) * 8 + 4 * h) * CP + wv * CW;
#pragma unroll
      for (int e = 0; e < NL; ++e) {
        const int idx = cbase + e;
        o[(idx / CW) * CP + idx % CW] = acc[e];
      }
    }
  }
}

// CGL lanes per voxel, 256 / CGL voxels (an x-run) per workgroup
template <int CP4, int CGL>
#ifndef VAMP_GATHER_OCC
#define VAMP_GATHER_OCC 5
#endif
__global__ void __launch_bounds__(256, VAMP_GATHER_OCC)
cam_bwd_cell_gather_kernel(RenderParams P, const int* __restrict__ off, const int* __restrict__ boff,
                           const float4* __restrict__ R, const float* __restrict__ Gcl,
                           float* __restrict__ gdens, float* __restrict__ gsem,
                           float* __restrict__ grgb, long ncell_b, int runs_x, long total_runs,
                           int accumulate, BetaTail btail, const int* __restrict__ runs,
                           const int* __restrict__ nruns, const float* __restrict__ part) {
  constexpr int CP = CP4 * 4;
  constexpr int CVPB = 256 / CGL;
  beta_tail(btail);                     // the ray pass's d beta partials (a launch of its own before round 3)
  __shared__ float outs[CP][CVPB + 1];
  const int nch = 1 + P.K + 3;
  const long V = (long) P.Z * P.Y * P.X;
  // accumulate mode: only the listed x-runs have anything to add (cam_heavy_list_kernel); the grid is fixed and a
  // workgroup takes every gridDim.x-th entry (x-neighbouring runs, which share cells, land on different CUs; a
  // contiguous slab of x-runs per XCD measured 12 % slower: the slabs next to the cameras carry most of the records)
  // (the first list entry goes out with the length of the list, not behind it: the list has room for every run)
  int ent = runs ? runs[blockIdx.x] : 0;
  const int nitems = runs ? *nruns : (int) total_runs;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    // (the thread's own indices are formed per item: hoisted out of the loop, what derives from them holds 20
    // registers across it and the kernel loses a wave of occupancy or spills)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int g = tid / CGL, l = tid % CGL;
    const unsigned lin = runs ? (unsigned) ent : (unsigned) item;
    if (runs && item + (int) gridDim.x < nitems) ent = runs[item + gridDim.x];
    const int bx = lin % (unsigned) runs_x;
    const unsigned rest = lin / (unsigned) runs_x;
    const int ix = bx * CVPB + g, iy = rest % (unsigned) P.Y;
    const int zb = rest / (unsigned) P.Y;
    const int iz = zb % P.Z, b = zb / P.Z;
    const bool vox_ok = ix < P.X;
    const int ixc = min(ix, P.X - 1);
    const CellRangesLight cl = cell_ranges_light<CGL>(P.Y, P.X, off, boff, ncell_b, b, ixc, iy, iz, l, kCellHeavy);
    const CellRanges& cr = cl.cr;

    float acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = 0.f;

    if (vox_ok) {
      // lane l's own cell (dz, dy, dx) = (l >> 2, (l >> 1) & 1, l & 1) is heavy: its records were summed per corner
      // by cam_cell_splat_kernel, and this voxel is the cell's corner (1 - dx, 1 - dy, 1 - dz)
      if (cl.own_n > kCellHeavy) {
        const int corner = l ^ 7;
        const int nchunk = splat_chunks(cl.own_n), len = cl.own_n / nchunk;
        for (int j = 0; j < nchunk; ++j) {
          const float4* p4 = reinterpret_cast<const float4*>(part + ((long) ((cl.own_start + j * len) / kCellHeavy) * 8 + corner) * CP);
#pragma unroll
          for (int c4 = 0; c4 < CP4; ++c4) {
            const float4 f = p4[c4];
            acc[c4 * 4] += f.x; acc[c4 * 4 + 1] += f.y; acc[c4 * 4 + 2] += f.z; acc[c4 * 4 + 3] += f.w;
          }
        }
      }
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
    // x-runs of every channel: with accumulate on top of what the buffers hold (the BEV branch's gradient)
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
#pragma unroll
    for (int i = 0; i < PE; ++i) {
      const int e = tid + i * 256;
      if (optr[i]) *optr[i] = prevv[i] + outs[e / CVPB][e % CVPB];
    }
    __syncthreads();                    // outs is the next item's too
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct CellWs {
  int* cnt;        // [ncell] counters
  int* off;        // [ncell] tile-local exclusive offsets
  int* bsum;       // [ntile] tile totals
  int* boff;       // [ntile] exclusive scan of the tile totals
  int* aux;        // [ntile] scratch of the level-2 scan, then [ntile] = total, [ntile+1] = heavy cells listed
  int2* hcells;    // [max(ncell, samples / 16)] list of the heavy cells' chunks: {first record, records}
  float* part;     // [samples / kCellHeavy + 2][8][CP] per-corner partial sums of the heavy cells
  int* runs;       // [x-runs] list of the x-runs with records (accumulate mode); aux[ntile + 2] = their number
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
  char* p = static_cast<char*>(scratch);
  CellWs w;
  w.cnt = reinterpret_cast<int*>(p); p += align_up((size_t) (ncell + kScanPad) * sizeof(int), 256);   // + the scan's ticket word
  w.off = reinterpret_cast<int*>(p); p += align_up((size_t) ncell * sizeof(int), 256);
  w.bsum = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.boff = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.aux = reinterpret_cast<int*>(p); p += align_up((size_t) (ntile + 4) * sizeof(int), 256);
  w.hcells = reinterpret_cast<int2*>(p); p += align_up(std::max<size_t>((size_t) ncell, samples / 16 + 4096) * sizeof(int2), 256);
  w.part = reinterpret_cast<float*>(p); p += align_up((samples / kCellHeavy + 2) * 8 * (size_t) to_params(d).CP * sizeof(float), 256);
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
  return CamRankRefs{w.cnt, w.rank, w.tile_se, (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1), nullptr};
}
int launch_cam_cells_zero(const VampRenderDesc* d, void* scratch, hipStream_t s) {
  const CellWs w = cell_ws(d, scratch);
  return launch_zero(w.cnt, (size_t) (cell_count_padded(d->B, d->Z, d->Y, d->X) + kScanPad) * sizeof(int), s);
}

// phase 0 / 1: rank + scan (what needs the geometry and the termination table); 3: the scan behind a forward that has
// drawn the ranks itself (mats .. term unused).  (2, the work lists as a launch of their own, is gone: they are built
// by the tail workgroups of the per-ray pass's launch, cam_lists.hpp.)  The scan's first workgroup sorts the ray tiles
// deepest first for the per-ray pass.
int launch_cam_cells_prepare(const VampRenderDesc* d, const RenderParams& P, const float* mats,
                             const float* us, const float* vs, const float* ds, void* scratch,
                             const int* term, int phase, hipStream_t s, bool counters_clean, const ScanJob* also) {
  const CellWs w = cell_ws(d, scratch);
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const size_t samples = (size_t) d->B * d->N * (d->D - 1) * d->fH * d->fW;
  const size_t voxels = (size_t) d->B * d->Z * d->Y * d->X;
  VAMP_REQUIRE(samples > 0 && samples < 0x7fffffffu && voxels < 0x7fffffffu && ncell < 0x7fffffffL,
               "sample / voxel / cell count exceeds 2^31");
  const long ncell_b = (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1);
  const ScanDuty duty{w.tile_se, w.tile_order, (int) ((long) d->B * d->N * ((d->fH + 7) / 8) * ((d->fW + 7) / 8))};
  if (phase == 2) return VAMP_OK;
  if (phase == 3) {
    if (also) {                     // one launch for this list's scan and another list's (the lift's pair cells)
      ScanJob mine;
      if (int e = make_scan_job(w.cnt, w.off, w.bsum, w.boff, w.aux, ncell, &mine)) return e;
      return launch_cell_scan_pair(mine, *also, s, &duty);
    }
    return launch_cell_scan(w.cnt, w.off, w.bsum, w.boff, w.aux, ncell, s, &duty);
  }
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
  if (also) {
    ScanJob mine;
    if (int e = make_scan_job(w.cnt, w.off, w.bsum, w.boff, w.aux, ncell, &mine)) return e;
    return launch_cell_scan_pair(mine, *also, s, &duty);
  }
  return launch_cell_scan(w.cnt, w.off, w.bsum, w.boff, w.aux, ncell, s, &duty);
}

// what the list-building workgroups behind the per-ray pass's tiles need (cam_lists.hpp)
CamListArgs cam_list_args(const VampRenderDesc* d, void* scratch) {
  const CellWs w = cell_ws(d, scratch);
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const long ntile = ncell / kScanTile;
  CamListArgs a;
  a.off = w.off; a.boff = w.boff; a.hcells = w.hcells;
  a.nhcells = w.aux + ntile + 1;                  // (and the run counter behind it) zeroed by the scan (runtime.hip)
  a.runs = w.runs; a.nruns = a.nhcells + 1;
  a.ncell_b = (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1);
  a.runs_x = (d->X + kRunVox - 1) / kRunVox;
  a.total_runs = (long) a.runs_x * d->Y * d->Z * d->B;
  a.ncell = ncell;
  a.cell_blocks = (unsigned) (ncell / kListCells);
  a.nblocks = a.cell_blocks + (unsigned) ((a.total_runs + 255) / 256);
  a.first_block = 0;
  return a;
}

// the heavy cells' per-corner sums, then the per-voxel gather of the records the per-ray pass has written in cell order
int launch_cam_bwd_cell(const VampRenderDesc* d, const RenderParams& P, const float* Gcl,
                        float* gdens, float* gsem, float* grgb, void* scratch, int accumulate,
                        hipEvent_t wait_event, int parts, BetaTail btail, hipStream_t s) {
  const CellWs w = cell_ws(d, scratch);
  const long ncell = cell_count_padded(d->B, d->Z, d->Y, d->X);
  const long ntile = ncell / kScanTile;
  const long ncell_b = (long) (d->Z + 1) * (d->Y + 1) * (d->X + 1);
  const int* nhcells = w.aux + ntile + 1;
  const int* runs = accumulate ? w.runs : nullptr;   // (overwrite mode: every run is stored)

  // measured at cfg-B (gather + heavy-VOXEL drain, rounds 1 - 5, us): 8 lanes 145 + 56, 16 lanes 173 + 56, 32 lanes
  // 249 + 56; after early ray termination 68 + 34
  constexpr int gl = kGatherLanes;
  const int vpb = 256 / gl;
  const int runs_x = (d->X + vpb - 1) / vpb;
  const long nblk = (long) runs_x * d->Y * d->Z * d->B;
  VAMP_REQUIRE(nblk < 0x7fffffffL, "too many x-runs");
  const unsigned grid = (unsigned) std::min<long>(nblk, kGatherGrid);
  const unsigned sgrid = (unsigned) std::min<long>(ncell, 4096);
  // PART_HEAVY: the per-cell sums (they touch the partial table only: no need to wait for whoever else writes the
  // gradient buffers); PART_GATHER reads them, so a caller that splits the parts issues HEAVY first, same stream
#define VAMP_CELL(CP4, NW)                                                                            \
  do {                                                                                              \
    if (parts & kCamPartHeavy)                                                                      \
      VAMP_TIMED(kProfCamBwdOwn, s, (cam_cell_splat_kernel<CP4, NW><<<sgrid, NW * 64, 0, s>>>(      \
          w.R, Gcl, w.hcells, nhcells, w.part)));                                                   \
    if (parts & kCamPartGather) {                                                                   \
      /* the gradient buffers are first touched here: whoever else accumulates into them (the BEV   \
         branch on another stream) must be done */                                                  \
      if (wait_event && hipStreamWaitEvent(s, wait_event, 0) != hipSuccess)                         \
        return fail(VAMP_EHIP, "%s: hipStreamWaitEvent failed", __func__);                          \
      VAMP_TIMED(kProfCamBwdBrick, s, (cam_bwd_cell_gather_kernel<CP4, gl><<<grid, 256, 0, s>>>(    \
          P, w.off, w.boff, w.R, Gcl, gdens, gsem, grgb, ncell_b, runs_x, nblk, accumulate, btail, runs, nhcells + 1, w.part))); \
    }                                                                                               \
  } while (0)
  // (waves of the splat: CP / NW channels each)
  if (P.CP == 12) VAMP_CELL(3, 2); else if (P.CP == 24) VAMP_CELL(6, VAMP_SPLAT_NW); else VAMP_CELL(8, 4);
#undef VAMP_CELL
  return check_launch("cam_bwd_cell_gather_kernel");
}

}  // namespace vamp
