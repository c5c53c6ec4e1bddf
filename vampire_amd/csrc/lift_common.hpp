// Shared by the lift kernels: parameters and the bit-exact voxel -> pixel projection.
#pragma once
#include "common.hpp"
#include "cell_list.hpp"

namespace vamp {

struct LiftParams {
  int B, N, C, D, fH, fW, Z, Y, X;
  float u_max, v_max, u_div, v_div, d_lo, d_hi, d_span;
  int use_depth;
};

inline LiftParams to_params(const VampLiftDesc* d) {
  LiftParams p;
  p.B = d->B; p.N = d->N; p.C = d->C; p.D = d->D; p.fH = d->fH; p.fW = d->fW;
  p.Z = d->Z; p.Y = d->Y; p.X = d->X;
  p.u_max = d->u_max; p.v_max = d->v_max; p.u_div = d->u_div; p.v_div = d->v_div;
  p.d_lo = d->d_lo; p.d_hi = d->d_hi; p.d_span = d->d_span; p.use_depth = d->use_depth;
  return p;
}

// Result of projecting one voxel centre into one camera.
struct LiftTap {
  bool valid;
  int ix0, iy0, iz0;
  float wx0, wx1, wy0, wy1, wz0, wz1;
  float zz;              // projected depth (camera z), used to bin voxels into depth slabs
  float fx, fy, fz;      // continuous tap coordinates
};

// get_pixel (bv2:365-388) + validity / normalisation (bv2:493-505) + aten's
// grid_sampler_unnormalize for align_corners=False.  Evaluation order is part of
// the contract (bit-exact tap indices): do not reassociate, do not fuse.
// WAVE_CULL (callers in wave-uniform control flow only): when the third row of `ida` is exactly
// (0, 0, 1, 0) -- image-plane augmentations never touch depth -- the projected depth zz equals the
// camera-space z bit for bit (or is NaN), so a camera that no lane of the wave has in front of it
// (z > d_lo, resp. z > 0) is invalid for all 64 voxels and the divisions, the third matrix and
// the normalisation are skipped: the same masks as the full chain, about half of the projections
// (the cameras facing away) at a third of the instructions.
// lift_project_from: the chain behind its first product.  `p` = inv(bda) . (x, y, z, 1), which is the same
// vector for every camera of a sample whose cameras share inv(bda) bit for bit (they do: bda is one matrix
// per sample, bv2:370-372), so callers that know it (the cull word's kLiftCullSharedBda bit) form it once.
template <bool WAVE_CULL = false>
__device__ __forceinline__ LiftTap lift_project_from(const LiftParams& P, const float* __restrict__ m, Vec4 p) {
  p = matvec(m + 16, p);   // intrin @ inv(sensor2ego)
  if (WAVE_CULL) {
    const bool e3 = m[40] == 0.0f && m[41] == 0.0f && m[42] == 1.0f && m[43] == 0.0f;   // uniform
    if (e3 && !__any(p.z > (P.use_depth ? P.d_lo : 0.0f))) {
      LiftTap t;
      t.valid = false;
      t.ix0 = t.iy0 = t.iz0 = 0;
      t.wx0 = t.wx1 = t.wy0 = t.wy1 = t.wz0 = t.wz1 = 0.f;
      t.zz = p.z;
      t.fx = t.fy = t.fz = 0.f;
      return t;
    }
  }
  float zc = (p.z < 1e-6f) ? 1e-6f : p.z;   // clamp(min=eps); NaN stays NaN
  p.x = p.x / zc;
  p.y = p.y / zc;
  p = matvec(m + 32, p);   // ida
  const float u = p.x, v = p.y, zz = p.z;
  LiftTap t;
  bool ok = (u > -0.5f) && (u < P.u_max) && (v > -0.5f) && (v < P.v_max);
  if (P.use_depth) ok = ok && (zz > P.d_lo) && (zz < P.d_hi);
  else ok = ok && (zz > 0.0f);
  t.valid = ok;
  t.zz = zz;
  float nx = 2.0f * (u / P.u_div) - 1.0f;
  float ny = 2.0f * (v / P.v_div) - 1.0f;
  float nz = P.use_depth ? (2.0f * ((zz - P.d_lo) / P.d_span) - 1.0f) : 0.0f;
  nx = fminf(fmaxf(nx, -2.0f), 2.0f);
  ny = fminf(fmaxf(ny, -2.0f), 2.0f);
  nz = fminf(fmaxf(nz, -2.0f), 2.0f);
  const float fx = ((nx + 1.0f) * (float) P.fW - 1.0f) / 2.0f;
  const float fy = ((ny + 1.0f) * (float) P.fH - 1.0f) / 2.0f;
  const float fz = ((nz + 1.0f) * (float) P.D - 1.0f) / 2.0f;
  const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
  t.ix0 = (int) flx; t.iy0 = (int) fly; t.iz0 = (int) flz;
  t.wx1 = fx - flx; t.wx0 = (flx + 1.0f) - fx;
  t.wy1 = fy - fly; t.wy0 = (fly + 1.0f) - fy;
  t.wz1 = fz - flz; t.wz0 = (flz + 1.0f) - fz;
  t.fx = fx; t.fy = fy; t.fz = fz;
  return t;
}

template <bool WAVE_CULL = false>
__device__ __forceinline__ LiftTap lift_project(const LiftParams& P, const float* __restrict__ m,
                                                float x, float y, float z) {
  return lift_project_from<WAVE_CULL>(P, m, matvec(m, Vec4{x, y, z, 1.0f}));   // inv(bda) first
}


// ---------------------------------------------------------------------------
// Cell lists of the lift backward (lift_bwd_cell.hip).  Cell = (image, floor tap row + 1, floor tap
// column + 1): (fH + 1) x (fW + 1) cells per camera image.  The FORWARD kernel (grad mode) or the
// stand-alone prepare kernel counts the valid (voxel, camera) pairs per cell and leaves every
// pair's taps and its four depth samples in `ptaps` (and `pcell`), indexed by (image, voxel); the
// backward's fill pass only re-lays them in cell order, each with its voxel's gradient row -- nothing on
// the backward projects a voxel or reads a depth plane again.
// ---------------------------------------------------------------------------
struct LiftCells {
  int cw, ch;                        // cells per row / column of one camera
  long ncell;                        // padded to the scan tile, + 2 for the range ends
};

inline LiftCells lift_cells(const VampLiftDesc* d) {
  LiftCells g;
  g.cw = d->fW + 1;
  g.ch = d->fH + 1;
  const long nc = (long) d->B * d->N * g.cw * g.ch + 2;
  g.ncell = (nc + kScanTile - 1) / kScanTile * kScanTile;
  return g;
}

struct LiftCellWs {
  int *cnt, *off, *bsum, *boff, *aux;
  unsigned* amask;                   // [B * V] cameras each voxel has a pair with (N <= 32)
  float4* ptaps;                     // [B * N * V][2] of the pair, voxel order (sparse): {wx1, wy1, wz1, (iz0 + 1) | (ix0 + 1) << 16} |
                                     // its depth samples sum_d w_d depth[d, pixel] at its four pixel taps (32 bytes side by side:
                                     // as two arrays the second store cost the training forward 8 us)
  int* pcell;                        // [B * N * V] (floor row + 1) << 16 | (floor column + 1)
  float4* recs;                      // [cap][2 + C / 4] every pair in cell order: its taps | its depth samples | its voxel's
                                     // row grad_out / (hits + 1e-6) -- everything the gather needs of a pair, one run
  int* rowq;                         // [B * N * fH] image rows, those with the most pairs first
  size_t bytes;
};

inline LiftCellWs lift_cell_ws(const VampLiftDesc* d, void* scratch) {
  const LiftCells g = lift_cells(d);
  const long ntile = g.ncell / kScanTile;
  // every (voxel, camera) pair can be valid
  const size_t V = (size_t) d->Z * d->Y * d->X;
  const size_t cap = (size_t) d->B * d->N * V;
  char* p = static_cast<char*>(scratch);
  LiftCellWs w;
  w.cnt = reinterpret_cast<int*>(p); p += align_up((size_t) (g.ncell + kScanPad) * sizeof(int), 256);   // + the scan's ticket word
  w.off = reinterpret_cast<int*>(p); p += align_up((size_t) g.ncell * sizeof(int), 256);
  w.bsum = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.boff = reinterpret_cast<int*>(p); p += align_up((size_t) ntile * sizeof(int), 256);
  w.aux = reinterpret_cast<int*>(p); p += align_up((size_t) (ntile + 4) * sizeof(int), 256);
  w.amask = reinterpret_cast<unsigned*>(p); p += align_up((size_t) d->B * V * sizeof(unsigned), 256);
  w.ptaps = reinterpret_cast<float4*>(p); p += align_up(cap * 2 * sizeof(float4), 256);
  w.pcell = reinterpret_cast<int*>(p); p += align_up(cap * sizeof(int), 256);
  w.recs = reinterpret_cast<float4*>(p); p += align_up(cap * (size_t) (2 + (d->C + 3) / 4) * sizeof(float4), 256);
  w.rowq = reinterpret_cast<int*>(p); p += align_up((size_t) d->B * d->N * d->fH * sizeof(int), 256);
  w.bytes = (size_t) (p - static_cast<char*>(scratch));
  return w;
}

// What the projecting kernels hand to lift_emit_pair.
struct LiftEmit {
  int* cnt;
  unsigned* amask;
  float4* ptaps;                     // [.][2]: taps | depth samples
  int* pcell;
  int cw, ch;
};

inline LiftEmit lift_emit_of(const VampLiftDesc* d, void* cells) {
  const LiftCells g = lift_cells(d);
  const LiftCellWs w = lift_cell_ws(d, cells);
  LiftEmit e;
  e.cnt = w.cnt; e.amask = w.amask; e.ptaps = w.ptaps; e.pcell = w.pcell;
  e.cw = g.cw; e.ch = g.ch;
  return e;
}

// One camera of one voxel, called in WAVE-UNIFORM control flow (exited lanes are fine): counts the
// pair in its cell -- one atomic per run of lanes with equal cells, x-neighbouring voxels share a
// cell in the far field -- and stores its taps and its depth samples `dep` (depth_taps: the planes around
// the projected depth interpolated at the four pixel taps, zero padding) at (image, voxel).  Returns whether
// the voxel has a pair with this camera (at least one of the four pixel taps exists).
// (The depth samples are stored by lift_emit_dep, which callers place BEHIND their other loads: the store has to
// wait for the depth planes, and issued here -- in front of the feature gather -- it held that gather's loads
// back by a round trip per camera: 37 -> 45 us for the training forward.)
__device__ __forceinline__ bool lift_emit_pair(const LiftParams& P, const LiftEmit& E, const LiftTap& t,
                                               bool live, long bn, long V, long vox, int lane) {
  const bool act = live && t.valid && t.ix0 >= -1 && t.ix0 < P.fW && t.iy0 >= -1 && t.iy0 < P.fH;
  if (!__any(act)) return false;
  const long cell = (bn * E.ch + (t.iy0 + 1)) * E.cw + (t.ix0 + 1);
  const LaneRun r = lane_run(act, cell, lane);
  if (r.head) atomicAdd(E.cnt + cell, r.len);
  if (act) {
    // (iz0 >= -1 for a valid pair; the cell column rides in the upper half of the same word)
    E.ptaps[(bn * V + vox) * 2] = make_float4(t.wx1, t.wy1, t.wz1, __int_as_float((t.iz0 + 1) | ((t.ix0 + 1) << 16)));
    E.pcell[bn * V + vox] = ((t.iy0 + 1) << 16) | (t.ix0 + 1);
  }
  return act;
}
__device__ __forceinline__ void lift_emit_dep(const LiftEmit& E, bool act, long bn, long V, long vox, const float (&dep)[4]) {
  if (act) E.ptaps[(bn * V + vox) * 2 + 1] = make_float4(dep[0], dep[1], dep[2], dep[3]);
}

// lift_bwd_cell.hip
size_t lift_bwd_cell_ws_bytes(const VampLiftDesc* d);
int launch_lift_bwd_cell(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, const float* gout,
                         const uint64_t* hits, float* gdepth, float* gfeat, void* scratch,
                         bool cells_valid, int variant, int half, bool softmax_bwd, bool feat_cl, hipStream_t s);
int launch_lift_cell_prepare(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, const void* depth, void* scratch, hipStream_t s);
// zero the cell counters (before a kernel that emits pairs) / scan them (after it)
int launch_lift_cells_begin(const VampLiftDesc* d, void* scratch, hipStream_t s, bool clean = false);
int launch_lift_cells_end(const VampLiftDesc* d, void* scratch, hipStream_t s);
int lift_cells_scan_job(const VampLiftDesc* d, void* scratch, ScanJob* job);

}  // namespace vamp
