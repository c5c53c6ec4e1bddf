// LDS-staged voxel bricks for the camera ray march (forward and the backward's per-ray pass).
//
// The 64 lanes of a wave are the 64 rays of an 8 x 8 pixel tile at ONE depth index.  At a fixed
// depth index the frustum points are an affine image of the pixel rectangle (get_geometry,
// bv2:328-349: inv(ida) is affine in (u, v), the depth is a constant of the plane), so the tile's
// sample points form a planar parallelogram whose bounding box in voxel coordinates is spanned by
// the tile's four corner rays.  64 samples x 8 taps hit only ~0.9 distinct voxel rows per sample
// (cfg-B: mean 50 rows per tile step), so instead of every lane pulling its own 8 x 96 B through
// the vector L1 (the L1 -> register path delivers ~32 B/clk/CU; 1.6 GB per forward at cfg-B), the
// wave copies the brick once into LDS -- LDS-DMA (global_load_lds_dwordx4): no registers, every
// load of a brick in flight together -- and the lanes gather their taps from LDS (ds_read_b128:
// 256 B/clk/CU, equal addresses broadcast).
//
//  * rows are laid out [dz][dy][dx] with a stride of RS4 float4 (odd, so that 16 consecutive rows
//    tile all 64 banks for ds_read_b128); the DMA writes wave-base + 16 * lane, so a pass moves
//    64 / RS4 rows with the padding lanes masked off;
//  * a brick larger than CAP rows (far field, diagonal views) is cut into sub-bricks that overlap
//    by one voxel; every lane belongs to the sub-brick that owns its floor tap;
//  * correctness never depends on the geometric argument: a lane whose taps are not covered by the
//    box (arbitrary caller-supplied geometry, non-finite points) falls back to a direct global
//    gather.
#pragma once
#include "render_common.hpp"

namespace vamp {

#ifdef VAMP_DUMP_BOXES
__device__ unsigned g_nbox;
__device__ int g_boxes[65536 * 8];
#endif

template <int CP4>
struct BrickLayout {
  static constexpr int RS4 = (CP4 % 2 == 0) ? CP4 + 1 : CP4;   // row stride in float4, odd
  static constexpr int RPP = 64 / RS4;                         // rows per DMA pass
};

// (n + 0.5) / d for small non-negative n (< 2^16) and 1 <= d <= 256: exact integer quotient
__device__ __forceinline__ int div_small(int n, float inv_d) {
  return (int) (((float) n + 0.5f) * inv_d);
}

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// inclusive voxel-index box [lo, hi] that contains the floor and floor + 1 taps of every INSIDE
// lane of the wave; wave-uniform
struct BrickBox {
  int lo[3], hi[3];
};

// One axis of the box from the extremes of the continuous tap coordinate over the tile's corner
// rays: 0.01 voxel of slack for the rounding of the interior lanes' chains; fmaxf(nan, 0) = 0.
__device__ __forceinline__ void brick_axis(float mn, float mx, int dim, int& lo, int& hi) {
  const float top = (float) (dim - 2);
  lo = (int) fminf(fmaxf(floorf(mn - 0.01f), 0.f), top);
  hi = (int) fminf(fmaxf(floorf(mx + 0.01f), 0.f), top) + 1;
}

// sub-brick extents: cut until a sub-brick fits, halving the owned span (extent - 1) of the
// longest axis; x last, its rows are contiguous in memory
template <int CAP>
__device__ __forceinline__ void brick_cut(const BrickBox& bb, int& sx, int& sy, int& sz) {
  sx = bb.hi[0] - bb.lo[0] + 1; sy = bb.hi[1] - bb.lo[1] + 1; sz = bb.hi[2] - bb.lo[2] + 1;
  while (sx * sy * sz > CAP) {
    if (sz >= sy && sz >= sx) sz = (sz >> 1) + 1;
    else if (sy >= sx) sy = (sy >> 1) + 1;
    else sx = (sx >> 1) + 1;
  }
}

#ifdef VAMP_ABL_NOWAIT
__device__ __forceinline__ void wait_vmem() {}
#else
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Start the copy of the ex x ey x ez brick with origin (ox, oy, oz) of the channel-last volume
// `vol4` ([Z][Y][X][CP4] float4) into `lds` (row r = (dz * ey + dy) * ex + dx at r * RS4) by
// LDS-DMA; returns at once, the data has landed after wait_vmem().  All 64 lanes take part.
template <int CP4>
__device__ __forceinline__ void brick_dma(const RenderParams& P, const float4* __restrict__ vol4,
                                          int ox, int oy, int oz, int ex, int ey, int ez,
                                          float4* lds, int lane) {
  constexpr int RS4 = BrickLayout<CP4>::RS4, RPP = BrickLayout<CP4>::RPP;
  const int g = lane / RS4, q = lane - g * RS4;
  const bool act = q < CP4 && g < RPP;
  const int rows = ex * ey * ez;
  const float inv_ex = 1.0f / (float) ex, inv_ey = 1.0f / (float) ey;
  const float4* src = vol4 + ((oz * P.Y + oy) * P.X + ox) * CP4 + q;
  const int yst = P.X * CP4, zst = P.Y * P.X * CP4;
#ifdef VAMP_STAGE_REG
  // diagnostic: the same rows through registers
  auto fetch = [&](int r) -> float4 {
    const int rc = min(r, rows - 1);
    const int l = div_small(rc, inv_ex), dx = rc - l * ex;
    const int dz = div_small(l, inv_ey), dy = l - dz * ey;
    return src[dz * zst + dy * yst + dx * CP4 - q + min(q, CP4 - 1)];
  };
  for (int r0 = g; r0 < rows + g; r0 += RPP * 4) {
    const float4 v0 = fetch(r0), v1 = fetch(r0 + RPP), v2 = fetch(r0 + 2 * RPP), v3 = fetch(r0 + 3 * RPP);
    if (act && r0 < rows) lds[r0 * RS4 + q] = v0;
    if (act && r0 + RPP < rows) lds[(r0 + RPP) * RS4 + q] = v1;
    if (act && r0 + 2 * RPP < rows) lds[(r0 + 2 * RPP) * RS4 + q] = v2;
    if (act && r0 + 3 * RPP < rows) lds[(r0 + 3 * RPP) * RS4 + q] = v3;
  }
  return;
#endif
  for (int r0 = 0; r0 < rows; r0 += RPP) {             // wave-uniform trip count
    const int r = r0 + g;
#ifdef VAMP_ABL_NODMA
    if (false) {
#else
    if (act && r < rows) {
#endif
      const int l = div_small(r, inv_ex), dx = r - l * ex;
      const int dz = div_small(l, inv_ey), dy = l - dz * ey;
#ifdef VAMP_ABL_HOTSRC
      const int off = (dz * zst + dy * yst + dx * CP4) & 0xfff;      // timing only: a 64 KB window
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (vol4 + off + q),
#else
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (src + dz * zst + dy * yst + dx * CP4),
#endif
          (__attribute__((address_space(3))) void*) (lds + r0 * RS4), 16, 0, 0);
    }
  }
}

// first sub-brick of a box (the one a prefetch loads)
template <int CP4, int CAP>
__device__ __forceinline__ void brick_prefetch(const RenderParams& P, const float* __restrict__ vol,
                                               const BrickBox& bb, float4* lds) {
  int sx, sy, sz;
  brick_cut<CAP>(bb, sx, sy, sz);
  brick_dma<CP4>(P, reinterpret_cast<const float4*>(vol), bb.lo[0], bb.lo[1], bb.lo[2],
                 min(sx, bb.hi[0] - bb.lo[0] + 1), min(sy, bb.hi[1] - bb.lo[1] + 1),
                 min(sz, bb.hi[2] - bb.lo[2] + 1), lds, threadIdx.x & 63);
}

// 8-tap trilinear sample of CP4 * 4 channels for every INSIDE lane of the wave through LDS
// bricks; s must come in zeroed.  Must be called by all 64 lanes (wave-convergent).  `lds` is
// this wave's private region of CAP * RS4 float4.  `bb` is the wave's box for this depth index;
// with `resident` the DMA of its first sub-brick has already been started into `lds`
// (brick_prefetch) and only needs to be waited for.
template <int CP4, int CAP>
__device__ __forceinline__ void brick_gather(const RenderParams& P, const float* __restrict__ vol,
                                             const VolTap& tp, bool inside, const BrickBox& bb,
                                             bool resident, float4* lds, float* __restrict__ s
#ifdef VAMP_STAMP
                                             , unsigned long long* cnt
#endif
                                             ) {
  constexpr int RS4 = BrickLayout<CP4>::RS4;
  const int lane = threadIdx.x & 63;
  const float4* vol4 = reinterpret_cast<const float4*>(vol);
  int sx, sy, sz;
  brick_cut<CAP>(bb, sx, sy, sz);
  // the floor tap that decides ownership (a sample on the upper face has its floor on the last
  // plane; it is owned through the plane below so that both of its taps are in the brick)
  const int fx = min(tp.ix0, P.X - 2), fy = min(tp.iy0, P.Y - 2), fz = min(tp.iz0, P.Z - 2);
  const int x1 = min(tp.ix0 + 1, P.X - 1) - tp.ix0;  // 0 or 1: step to the "+1" tap
  const int y1 = min(tp.iy0 + 1, P.Y - 1) - tp.iy0;
  const int z1 = min(tp.iz0 + 1, P.Z - 1) - tp.iz0;
  const float wx1 = x1 ? tp.wx1 : 0.f, wy1 = y1 ? tp.wy1 : 0.f, wz1 = z1 ? tp.wz1 : 0.f;
  bool todo = inside;
  bool first = true;
  for (int oz = bb.lo[2]; oz < bb.hi[2]; oz += sz - 1) {
    const int ez = min(sz, bb.hi[2] - oz + 1);
    for (int oy = bb.lo[1]; oy < bb.hi[1]; oy += sy - 1) {
      const int ey = min(sy, bb.hi[1] - oy + 1);
      for (int ox = bb.lo[0]; ox < bb.hi[0]; ox += sx - 1) {
        const int ex = min(sx, bb.hi[0] - ox + 1);
        const bool mine = todo && fx >= ox && fx <= ox + ex - 2 && fy >= oy && fy <= oy + ey - 2 &&
                          fz >= oz && fz <= oz + ez - 2;
        const bool have = first && resident;
        first = false;
        if (__ballot(mine) == 0ull) continue;           // (a prefetched brick nobody owns is dropped)
#ifdef VAMP_DUMP_BOXES
        if (lane == 0) {
          const unsigned k = atomicAdd(&g_nbox, 1u);
          if (k < 65536u) { int* e = g_boxes + k * 8; e[0] = ox; e[1] = oy; e[2] = oz; e[3] = ex; e[4] = ey; e[5] = ez; e[6] = blockIdx.x; e[7] = threadIdx.x >> 6; }
        }
#endif
#ifdef VAMP_STAMP
        unsigned long long ta, tb2, tc;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ta) :: "memory");
        cnt[0] += 1; if (!have) cnt[1] += 1;
        cnt[4] += ex * ey * ez;
#endif
        if (!have) {
          wait_vmem();                                   // an unused prefetch may still be landing
          wait_lds();                                    // earlier reads of this buffer are done
          brick_dma<CP4>(P, vol4, ox, oy, oz, ex, ey, ez, lds, lane);
        }
#ifdef VAMP_STAMP
        { unsigned long long ti; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ti) :: "memory"); cnt[5] += ti - ta; }
#endif
        wait_vmem();
#ifdef VAMP_STAMP
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb2) :: "memory");
        cnt[2] += tb2 - ta;
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (mine) {
          const int r0 = ((tp.iz0 - oz) * ey + (tp.iy0 - oy)) * ex + (tp.ix0 - ox);
          const int dy = y1 * ex, dz = z1 * ey * ex;
          // the x pair of each (y, z) corner: two adjacent rows
#ifdef VAMP_ABL_NOGATHER
          for (int j = 0; j < 0; ++j) {
#else
#pragma unroll 2
          for (int j = 0; j < 4; ++j) {
#endif
            const float wyz = ((j & 1) ? wy1 : tp.wy0) * ((j & 2) ? wz1 : tp.wz0);
            const float4* row = lds + (r0 + ((j & 1) ? dy : 0) + ((j & 2) ? dz : 0)) * RS4;
            const float4* rowx = row + x1 * RS4;
            const float w0 = tp.wx0 * wyz, w1 = wx1 * wyz;
#pragma unroll
            for (int q = 0; q < CP4; ++q) {
              const float4 f = row[q];
              s[q * 4 + 0] = __builtin_fmaf(w0, f.x, s[q * 4 + 0]);
              s[q * 4 + 1] = __builtin_fmaf(w0, f.y, s[q * 4 + 1]);
              s[q * 4 + 2] = __builtin_fmaf(w0, f.z, s[q * 4 + 2]);
              s[q * 4 + 3] = __builtin_fmaf(w0, f.w, s[q * 4 + 3]);
            }
#pragma unroll
            for (int q = 0; q < CP4; ++q) {
              const float4 f = rowx[q];
              s[q * 4 + 0] = __builtin_fmaf(w1, f.x, s[q * 4 + 0]);
              s[q * 4 + 1] = __builtin_fmaf(w1, f.y, s[q * 4 + 1]);
              s[q * 4 + 2] = __builtin_fmaf(w1, f.z, s[q * 4 + 2]);
              s[q * 4 + 3] = __builtin_fmaf(w1, f.w, s[q * 4 + 3]);
            }
          }
          todo = false;
        }
#ifdef VAMP_STAMP
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc) :: "memory");
        cnt[3] += tc - tb2;
#endif
      }
    }
  }
  // safety net: taps outside the box (never for frustum geometry from the matrices)
  if (__ballot(todo) != 0ull) {
    if (todo) {
#ifdef VAMP_BRICK_DEBUG
      s[1] += 1000.f;
#endif
      // one tap at a time: this path is rare and must not set the kernel's register budget
#pragma unroll 1
      for (int k = 0; k < 8; ++k) {
        const float wt = ((k & 1) ? wx1 : tp.wx0) * ((k & 2) ? wy1 : tp.wy0) * ((k & 4) ? wz1 : tp.wz0);
        const float4* row = vol4 + (((tp.iz0 + ((k & 4) ? z1 : 0)) * P.Y + tp.iy0 + ((k & 2) ? y1 : 0)) * P.X +
                                    tp.ix0 + ((k & 1) ? x1 : 0)) * CP4;
#pragma unroll
        for (int q = 0; q < CP4; ++q) {
          const float4 f = row[q];
          s[q * 4 + 0] = __builtin_fmaf(wt, f.x, s[q * 4 + 0]);
          s[q * 4 + 1] = __builtin_fmaf(wt, f.y, s[q * 4 + 1]);
          s[q * 4 + 2] = __builtin_fmaf(wt, f.z, s[q * 4 + 2]);
          s[q * 4 + 3] = __builtin_fmaf(wt, f.w, s[q * 4 + 3]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Per-tile schedule, shared by the four waves of a workgroup through LDS: which depth indices of
// the tile's rays can hold an inside sample at all, and the box of each.  One wave-wide evaluation
// of the frustum chain covers the four corner rays of 16 consecutive depth indices (lane = 4 * step
// + corner); wave k plans indices [32 k, 32 k + 32).  Depth indices whose corner box misses the
// volume are composited without evaluating the chain at all (the 64 samples are affine
// combinations of the four corners, so they are all outside), and the active indices are dealt
// evenly to the four waves (a near tile has all of its first 22 indices active and none beyond
// the 50th: fixed chunks leave three waves idle).
// ---------------------------------------------------------------------------
constexpr int kPlanMax = 128;                          // depth indices per ray the plan can hold

// plan[i] = {lo | hi << 16 of the box per axis, active flag}
__device__ __forceinline__ void plan_tile(const RenderParams& P, const float* __restrict__ m,
                                          const float* __restrict__ us, const float* __restrict__ vs,
                                          const float* __restrict__ ds, int w_lo, int w_hi, int h_lo,
                                          int h_hi, int wave, int4* __restrict__ plan) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 3;
  const float u = us[(c & 1) ? w_hi : w_lo], v = vs[(c & 2) ? h_hi : h_lo];
  const int S = P.D - 1;
#pragma unroll 1
  for (int e = 0; e < 2; ++e) {
    const int i = 32 * wave + 16 * e + (lane >> 2);
    float x, y, z;
    frustum_point(m, u, v, ds[min(i, P.D - 1)], x, y, z);
    // a non-finite corner voids the affine argument: keep the depth index
    int fin = (fabsf(x) <= 1e30f && fabsf(y) <= 1e30f && fabsf(z) <= 1e30f) ? 1 : 0;
    const VolTap tp = volume_tap(P, nan_to_num_geom(x), nan_to_num_geom(y), nan_to_num_geom(z));
    float mn[3] = {tp.fx, tp.fy, tp.fz}, mx[3] = {tp.fx, tp.fy, tp.fz};
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
        mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
      }
      fin &= __shfl_xor(fin, o, 64);
    }
    const float slack = 0.05f;                           // voxels; the interior lanes deviate < 1e-3
    const bool hit = mx[0] >= -slack && mn[0] <= (float) (P.X - 1) + slack &&
                     mx[1] >= -slack && mn[1] <= (float) (P.Y - 1) + slack &&
                     mx[2] >= -slack && mn[2] <= (float) (P.Z - 1) + slack;
    int4 ent;
    int lo, hi;
    brick_axis(mn[0], mx[0], P.X, lo, hi); ent.x = lo | (hi << 16);
    brick_axis(mn[1], mx[1], P.Y, lo, hi); ent.y = lo | (hi << 16);
    brick_axis(mn[2], mx[2], P.Z, lo, hi); ent.z = lo | (hi << 16);
    ent.w = (i < S && (hit || !fin)) ? 1 : 0;
    if (c == 0) plan[i] = ent;
  }
}

// box of depth index j; j is wave-uniform (every lane reads the same LDS word: a broadcast)
__device__ __forceinline__ BrickBox plan_box(const int4* __restrict__ plan, int j) {
  const int4 e = plan[j];
  const int px = __builtin_amdgcn_readfirstlane(e.x), py = __builtin_amdgcn_readfirstlane(e.y);
  const int pz = __builtin_amdgcn_readfirstlane(e.z);
  BrickBox b;
  b.lo[0] = px & 0xffff; b.hi[0] = px >> 16;
  b.lo[1] = py & 0xffff; b.hi[1] = py >> 16;
  b.lo[2] = pz & 0xffff; b.hi[2] = pz >> 16;
  return b;
}

// the active depth indices as two 64-bit masks (wave-uniform)
struct PlanMask {
  unsigned long long lo, hi;
};

__device__ __forceinline__ PlanMask plan_mask(const int4* __restrict__ plan) {
  const int lane = threadIdx.x & 63;
  PlanMask mk;
  mk.lo = __ballot(plan[lane].w != 0);
  mk.hi = __ballot(plan[lane + 64].w != 0);
  return mk;
}

__device__ __forceinline__ bool mask_test(const PlanMask& mk, int j) {
  return (((j < 64) ? (mk.lo >> j) : (mk.hi >> (j - 64))) & 1ull) != 0;
}

// first active depth index >= j (kPlanMax when none); j wave-uniform
__device__ __forceinline__ int mask_next(const PlanMask& mk, int j) {
  if (j < 64) {
    const unsigned long long t = mk.lo >> j;
    if (t) return j + __builtin_ctzll(t);
    j = 64;
  }
  if (j < 128) {
    const unsigned long long t = mk.hi >> (j - 64);
    if (t) return j + __builtin_ctzll(t);
  }
  return kPlanMax;
}

// depth index of the active index of rank r (0-based; r < number of active indices)
__device__ __forceinline__ int mask_select(const PlanMask& mk, int r) {
  unsigned long long t = mk.lo;
  int base = 0;
  const int nlo = __builtin_popcountll(mk.lo);
  if (r >= nlo) { t = mk.hi; base = 64; r -= nlo; }
  for (int k = 0; k < r; ++k) t &= t - 1;
  return base + __builtin_ctzll(t);
}

// wave `sub`'s contiguous range [j0, j1) of the S depth indices: equal shares of the active ones
__device__ __forceinline__ void plan_share(const PlanMask& mk, int S, int sub, int& j0, int& j1) {
  const int A = __builtin_popcountll(mk.lo) + __builtin_popcountll(mk.hi);
  if (A < 4) {                                           // nothing to balance
    const int L = (S + 3) >> 2;
    j0 = min(S, sub * L); j1 = min(S, j0 + L);
    return;
  }
  j0 = sub == 0 ? 0 : mask_select(mk, (sub * A) >> 2);
  j1 = sub == 3 ? S : mask_select(mk, ((sub + 1) * A) >> 2);
}

}  // namespace vamp
