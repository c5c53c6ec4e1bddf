// BEVDepth-style voxel pooling (north_star's "LSS frustum-to-voxel pooling op"),
// gfx950.  NOT part of /root/reference at the pinned commit -- its backbones lift by pulling
// (grid_sample, base_vampire2.py:483-516, this build's lift kernels) -- so this operator follows the
// published BEVDepth definition and is "parity unpinned" (oracle: oracle/voxel_pooling_oracle.py, a
// numpy scatter-add of the same definition):
//
//     out[b, y, x, :] = sum of feat[b, p, :] over the points p of sample b whose integer voxel index
//                       geom[b, p] = (x, y, z) lies inside [0, nx) x [0, ny) x [0, nz)
//
// The upstream op is a float atomicAdd per (point, channel).  On MI355X device-scope float atomics
// run at the memory side, so this follows the build's sort-then-own pattern:
// count the points per BEV cell (integer atomics, each point keeps the rank it drew), scan, write
// the point ids in cell order, then one wave per cell adds its points' feature rows -- C contiguous
// floats per point, coalesced -- and stores the cell's row once.  Empty cells are written as
// zeros, so the output needs no zero fill.  HBM-bound: the features are read once (4 C bytes per
// point), the output written once; no MFMA.  The backward is a pure gather.
#include "common.hpp"

namespace vamp {
namespace {

struct PoolParams {
  int B, C, nx, ny, nz;
  long P;                 // points per sample
};

__device__ __forceinline__ long pool_cell(const PoolParams& q, const int* __restrict__ geom, long gp, int b) {
  const int x = geom[gp * 3], y = geom[gp * 3 + 1], z = geom[gp * 3 + 2];
  const bool in = x >= 0 && x < q.nx && y >= 0 && y < q.ny && z >= 0 && z < q.nz;
  return in ? ((long) b * q.ny + y) * q.nx + x : -1;
}

// FILL = false: count (a point keeps the rank it drew); FILL = true: write its id into the cell's run
template <bool FILL>
__global__ void __launch_bounds__(256)
pool_cells_kernel(PoolParams q, const int* __restrict__ geom, int* __restrict__ cnt, int* __restrict__ rank,
                  const int* __restrict__ off, const int* __restrict__ boff, int* __restrict__ ids) {
  const long gp = (long) blockIdx.x * 256 + threadIdx.x;
  if (gp >= q.B * q.P) return;
  const long cell = pool_cell(q, geom, gp, (int) (gp / q.P));
  if (cell < 0) return;
  if (!FILL) rank[gp] = atomicAdd(cnt + cell, 1);
  else ids[off[cell] + boff[cell / kScanTile] + rank[gp]] = (int) (gp % q.P);
}

// four channels of one point per lane
__device__ __forceinline__ float4 ld4(const float* p, long i) { return *reinterpret_cast<const float4*>(p + i); }
__device__ __forceinline__ float4 ld4(const __hip_bfloat16* p, long i) {
  const uint2 u = *reinterpret_cast<const uint2*>(p + i);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}

// One wave per cell.  C % 4 == 0 and C <= 256 (VEC): a lane holds four channels, so a feature row takes
// Q = C / 4 lanes and R = 64 / Q points of the cell's list are added side by side (C = 80: 20 lanes x 3
// points; with a lane per channel the second pass of 64 covered 16 channels); the R partial rows are
// folded with shuffles at the end.  Otherwise lanes over channels, several passes.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
pool_gather_kernel(PoolParams q, const T* __restrict__ feat, const int* __restrict__ off,
                   const int* __restrict__ boff, const int* __restrict__ ids, float* __restrict__ out,
                   long ncells) {
  const int lane = threadIdx.x & 63;
  const long cell = (long) blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= ncells) return;
  const int b = (int) (cell / ((long) q.ny * q.nx));
  const int beg = off[cell] + boff[cell / kScanTile];
  const int end = off[cell + 1] + boff[(cell + 1) / kScanTile];
  const T* fb = feat + (long) b * q.P * q.C;
  if (VEC) {
    const int Q = q.C >> 2, R = 64 / Q;
    const int cq = lane % Q, r = lane / Q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R) {
      int k = beg + r;
      for (; k + R < end; k += 2 * R) {           // two rows in flight per lane
        const float4 v0 = ld4(fb, (long) ids[k] * q.C + 4 * cq), v1 = ld4(fb, (long) ids[k + R] * q.C + 4 * cq);
        acc.x += v0.x + v1.x; acc.y += v0.y + v1.y; acc.z += v0.z + v1.z; acc.w += v0.w + v1.w;
      }
      if (k < end) {
        const float4 v0 = ld4(fb, (long) ids[k] * q.C + 4 * cq);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      }
    }
    for (int j = 1; j < R; ++j) {                 // fold the point slots into slot 0 (wave-uniform trip count)
      const int src = min(cq + j * Q, 63);
      const float x = __shfl(acc.x, src, 64), y = __shfl(acc.y, src, 64), z = __shfl(acc.z, src, 64), w = __shfl(acc.w, src, 64);
      if (r == 0) { acc.x += x; acc.y += y; acc.z += z; acc.w += w; }
    }
    if (r == 0) *reinterpret_cast<float4*>(out + cell * q.C + 4 * cq) = acc;
    return;
  }
  for (int c0 = 0; c0 < q.C; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < q.C;
    float acc = 0.f;
    int k = beg;
    for (; k + 4 <= end; k += 4) {
      const int i0 = ids[k], i1 = ids[k + 1], i2 = ids[k + 2], i3 = ids[k + 3];
      const float v0 = live ? ldf(fb, (long) i0 * q.C + c) : 0.f;
      const float v1 = live ? ldf(fb, (long) i1 * q.C + c) : 0.f;
      const float v2 = live ? ldf(fb, (long) i2 * q.C + c) : 0.f;
      const float v3 = live ? ldf(fb, (long) i3 * q.C + c) : 0.f;
      acc += (v0 + v1) + (v2 + v3);
    }
    for (; k < end; ++k) acc += live ? ldf(fb, (long) ids[k] * q.C + c) : 0.f;
    if (live) out[cell * q.C + c] = acc;
  }
}

// grad_feat[b, p, :] = grad_out[cell(p), :] (0 for points outside the grid)
template <bool VEC>
__global__ void __launch_bounds__(256)
pool_bwd_kernel(PoolParams q, const int* __restrict__ geom, const float* __restrict__ gout,
                float* __restrict__ gfeat, int ppw) {
  const int lane = threadIdx.x & 63;
  const long wave = (long) blockIdx.x * 4 + (threadIdx.x >> 6);
  if (VEC) {
    // ppw = 64 / (C / 4) points per wave, a lane copies four channels of one point's row
    const int Q = q.C >> 2, cq = lane % Q, r = lane / Q;
    const long gp = wave * ppw + r;
    if (r >= ppw || gp >= q.B * q.P) return;
    const long cell = pool_cell(q, geom, gp, (int) (gp / q.P));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cell >= 0) v = *reinterpret_cast<const float4*>(gout + cell * q.C + 4 * cq);
    *reinterpret_cast<float4*>(gfeat + gp * q.C + 4 * cq) = v;
    return;
  }
  const long gp = wave;
  if (gp >= q.B * q.P) return;
  const long cell = pool_cell(q, geom, gp, (int) (gp / q.P));
  for (int c = lane; c < q.C; c += 64) gfeat[gp * q.C + c] = cell >= 0 ? gout[cell * q.C + c] : 0.f;
}

struct PoolWs {
  int *cnt, *off, *bsum, *boff, *aux, *rank, *ids;
  long ncell;            // rounded up to the scan tile (+ one tile: the gather reads off[cell + 1])
  size_t bytes;
};

PoolWs pool_ws(const VampPoolDesc* d, void* ws) {
  PoolWs w;
  const long cells = (long) d->B * d->ny * d->nx;
  w.ncell = (cells + 1 + kScanTile - 1) / kScanTile * kScanTile;
  const long ntile = w.ncell / kScanTile;
  char* p = static_cast<char*>(ws);
  auto take = [&](size_t n) { char* r = p; p += align_up(n, 256); return r; };
  w.cnt = reinterpret_cast<int*>(take((w.ncell + kScanPad) * sizeof(int)));
  w.off = reinterpret_cast<int*>(take(w.ncell * sizeof(int)));
  w.bsum = reinterpret_cast<int*>(take(ntile * sizeof(int)));
  w.boff = reinterpret_cast<int*>(take(ntile * sizeof(int)));
  w.aux = reinterpret_cast<int*>(take((ntile + 4) * sizeof(int)));
  w.rank = reinterpret_cast<int*>(take((size_t) d->B * d->P * sizeof(int)));
  w.ids = reinterpret_cast<int*>(take((size_t) d->B * d->P * sizeof(int)));
  w.bytes = (size_t) (p - static_cast<char*>(ws));
  return w;
}

int pool_validate(const VampPoolDesc* d) {
  VAMP_REQUIRE(d != nullptr, "desc is NULL");
  VAMP_REQUIRE(d->B > 0 && d->P > 0 && d->C > 0, "B, P, C must be positive");
  VAMP_REQUIRE(d->nx > 0 && d->ny > 0 && d->nz > 0, "voxel_num must be positive");
  VAMP_REQUIRE((long) d->B * d->P < 0x7fffffffL && d->P < 0x7fffffffL, "too many points");
  VAMP_REQUIRE((long) d->B * d->ny * d->nx < 0x7fffffffL - 2 * kScanTile, "too many cells");
  VAMP_REQUIRE(d->in_dtype == VAMP_F32 || d->in_dtype == VAMP_BF16, "in_dtype");
  return VAMP_OK;
}

}  // namespace
}  // namespace vamp

using namespace vamp;

extern "C" {

size_t vamp_voxel_pooling_workspace_bytes(const VampPoolDesc* d) {
  if (!d || pool_validate(d)) return 0;
  return pool_ws(d, nullptr).bytes;
}

int vamp_voxel_pooling_forward(const VampPoolDesc* d, const int32_t* geom_xyz, const void* feat, float* out,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = pool_validate(d)) return e;
  VAMP_REQUIRE(geom_xyz && feat && out, "NULL tensor");
  const PoolWs w = pool_ws(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const PoolParams q{d->B, d->C, d->nx, d->ny, d->nz, d->P};
  const long npts = (long) d->B * d->P, cells = (long) d->B * d->ny * d->nx;
  if (int e = launch_zero(w.cnt, (w.ncell + kScanPad) * sizeof(int), s)) return e;
  const unsigned pgrid = (unsigned) ((npts + 255) / 256);
  VAMP_TIMED(kProfAux, s, (pool_cells_kernel<false><<<pgrid, 256, 0, s>>>(q, geom_xyz, w.cnt, w.rank, w.off, w.boff, w.ids)));
  if (int e = check_launch("pool_cells_kernel<count>")) return e;
  if (int e = launch_cell_scan(w.cnt, w.off, w.bsum, w.boff, w.aux, w.ncell, s)) return e;
  VAMP_TIMED(kProfAux, s, (pool_cells_kernel<true><<<pgrid, 256, 0, s>>>(q, geom_xyz, w.cnt, w.rank, w.off, w.boff, w.ids)));
  if (int e = check_launch("pool_cells_kernel<fill>")) return e;
  const unsigned ggrid = (unsigned) ((cells + 3) / 4);
  const bool vec = d->C % 4 == 0 && d->C <= 256;
#define VAMP_POOL(T, V) VAMP_TIMED(kProfAux, s, (pool_gather_kernel<T, V><<<ggrid, 256, 0, s>>>(q, static_cast<const T*>(feat), w.off, w.boff, w.ids, out, cells)))
  if (d->in_dtype == VAMP_F32) { if (vec) VAMP_POOL(float, true); else VAMP_POOL(float, false); }
  else { if (vec) VAMP_POOL(__hip_bfloat16, true); else VAMP_POOL(__hip_bfloat16, false); }
#undef VAMP_POOL
  return check_launch("pool_gather_kernel");
}

int vamp_voxel_pooling_backward(const VampPoolDesc* d, const int32_t* geom_xyz, const float* grad_out,
                                float* grad_feat, void* stream) {
  if (int e = pool_validate(d)) return e;
  VAMP_REQUIRE(geom_xyz && grad_out && grad_feat, "NULL tensor");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const PoolParams q{d->B, d->C, d->nx, d->ny, d->nz, d->P};
  const long npts = (long) d->B * d->P;
  if (d->C % 4 == 0 && d->C <= 256) {
    const int ppw = 64 / (d->C / 4);
    const long waves = (npts + ppw - 1) / ppw;
    VAMP_TIMED(kProfAux, s, (pool_bwd_kernel<true><<<(unsigned) ((waves + 3) / 4), 256, 0, s>>>(q, geom_xyz, grad_out, grad_feat, ppw)));
  } else {
    VAMP_TIMED(kProfAux, s, (pool_bwd_kernel<false><<<(unsigned) ((npts + 3) / 4), 256, 0, s>>>(q, geom_xyz, grad_out, grad_feat, 1)));
  }
  return check_launch("pool_bwd_kernel");
}

}  // extern "C"
