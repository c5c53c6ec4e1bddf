// LIFT backward, v3: bin-then-own.  Autograd of base_vampire2.py:507-514
// (grid_sampler_3d backward + the camera mean) as three streaming passes:
//
//   count   thread per voxel, forward's bit-exact projection: every valid (voxel, camera)
//           pair increments the counter of each TH x TW pixel tile its four (h,w) taps touch
//   scan    exclusive prefix sum of the tile counters -> list offsets   (runtime.hip)
//   fill    same walk; the pair's record {fx, fy, fz, dep[4], gs[C]} -- continuous tap
//           coordinates, depth-interpolated weights, grad_out/(hits+1e-6) -- is appended to
//           the list of every tile it touches
//   own     one workgroup per (camera, tile) streams its list with 16-lane groups (lane =
//           channel): conflict-free ds_add_f32 into an LDS copy of grad_feat, a shuffle dot
//           product for grad_depth; each output element is stored exactly once
//
// No global float atomics, no memset, no layout transposes.  If the lists would not fit the
// workspace the caller falls back to lift_bwd_tile.hip (device-side decision, no host sync).
#include "lift_common.hpp"

namespace vamp {

constexpr int BTH = 8, BTW = 8, BTP = BTH * BTW;
constexpr int GLN = 16;

struct BinGeom {
  int tiles_w, tiles_h;
};

// tiles touched by taps {i0, i0+1} along one axis (extent n, tile size ts): up to two ids
__device__ __forceinline__ void axis_tiles(int i0, int n, int ts, int& a, int& b) {
  a = (i0 >= 0 && i0 < n) ? i0 / ts : -1;
  b = (i0 + 1 >= 0 && i0 + 1 < n) ? (i0 + 1) / ts : -1;
  if (b == a) b = -1;
  if (a < 0) { a = b; b = -1; }
}

// One atomic per distinct bin among the active lanes of a wave (wave-aggregated atomics):
// returns, for each active lane, base + rank where base is the counter value before the
// wave's batch.  Must be called by all lanes of the wave together.
__device__ __forceinline__ int wave_bin_add(int* __restrict__ counters, int bin, bool active) {
  const int lane = threadIdx.x & 63;
  int res = 0;
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long) todo) - 1;
    const int lb = __shfl(bin, leader, 64);
    const bool same = active && bin == lb;
    const unsigned long long m = __ballot(same);
    int base = 0;
    if (lane == leader) base = atomicAdd(counters + lb, (int) __popcll(m));
    base = __shfl(base, leader, 64);
    if (same) res = base + (int) __popcll(m & ((1ull << lane) - 1ull));
    todo &= ~m;
  }
  return res;
}

// Same aggregation without a result: fire-and-forget atomics, the wave never waits on them.
__device__ __forceinline__ void wave_bin_count(int* __restrict__ counters, int bin, bool active) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long) todo) - 1;
    const int lb = __shfl(bin, leader, 64);
    const unsigned long long m = __ballot(active && bin == lb);
    if (lane == leader) atomicAdd(counters + lb, (int) __popcll(m));
    todo &= ~m;
  }
}

// Per-workgroup LDS histogram over this sample's (camera, tile) bins: same-address global
// integer atomics serialise (~1 us each under contention), so a 1024-voxel workgroup touches
// each global counter at most once.
//   count: pass A builds the histogram, the non-zero bins are flushed (fire-and-forget);
//   fill : pass A, then each non-zero bin reserves a contiguous range with ONE returned
//          atomic, pass B re-walks the voxels and ranks them inside the range with LDS atomics.
constexpr int kBinThreads = 1024;

template <typename T, int CH, bool FILL>
__global__ void __launch_bounds__(kBinThreads)
lift_bwd_bin_kernel(LiftParams P, BinGeom G, const float* __restrict__ mats,
                    const float* __restrict__ xs, const float* __restrict__ ys,
                    const float* __restrict__ zs, const T* __restrict__ depth,
                    const float* __restrict__ gout, const uint64_t* __restrict__ hits,
                    int* __restrict__ cnt, const int* __restrict__ off, int* __restrict__ fill,
                    float* __restrict__ entries, const int* __restrict__ total, int cap) {
  if (FILL && *total > cap) return;                 // lists do not fit: the fallback runs instead
  extern __shared__ int ltab[];                     // [2][N * ntile]: histogram / reserved bases
  const int ntile = G.tiles_w * G.tiles_h;
  const int NT = P.N * ntile;
  int* hist = ltab;
  int* lbase = ltab + NT;
  const int tid = threadIdx.x;
  for (int e = tid; e < 2 * NT; e += kBinThreads) ltab[e] = 0;
  const int x = blockIdx.x * 64 + (tid & 63);
  const int y = blockIdx.y * (kBinThreads / 64) + (tid >> 6);
  const int z = blockIdx.z % P.Z, b = blockIdx.z / P.Z;
  const bool live = x < P.X && y < P.Y;
  const int xc = min(x, P.X - 1), yc = min(y, P.Y - 1);
  const float vx = xs[xc], vy = ys[yc], vz = zs[z];
  const long V = (long) P.Z * P.Y * P.X;
  const long vox = ((long) z * P.Y + yc) * P.X + xc;
  const long HW = (long) P.fH * P.fW;
  const int ES = 8 + P.C;
  __syncthreads();

  // ---- pass A: histogram ----
  for (int n = 0; n < P.N; ++n) {
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project(P, mats + bn * 48, vx, vy, vz);
    if (!live || !t.valid) continue;
    int txa, txb, tya, tyb;
    axis_tiles(t.ix0, P.fW, BTW, txa, txb);
    axis_tiles(t.iy0, P.fH, BTH, tya, tyb);
    if (txa < 0 || tya < 0) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tx = (q & 1) ? txb : txa, ty = (q & 2) ? tyb : tya;
      if (tx >= 0 && ty >= 0) atomicAdd(hist + n * ntile + ty * G.tiles_w + tx, 1);
    }
  }
  __syncthreads();
  if (!FILL) {
    for (int e = tid; e < NT; e += kBinThreads)
      if (hist[e]) atomicAdd(cnt + (long) b * NT + e, hist[e]);
    return;
  }
  for (int e = tid; e < NT; e += kBinThreads)
    if (hist[e]) {
      lbase[e] = atomicAdd(fill + (long) b * NT + e, hist[e]);
      hist[e] = 0;
    }
  __syncthreads();

  // ---- pass B: rank inside the reserved ranges and write the records ----
  for (int n = 0; n < P.N; ++n) {
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project(P, mats + bn * 48, vx, vy, vz);
    if (!live || !t.valid) continue;
    int txa, txb, tya, tyb;
    axis_tiles(t.ix0, P.fW, BTW, txa, txb);
    axis_tiles(t.iy0, P.fH, BTH, tya, tyb);
    if (txa < 0 || tya < 0) continue;
    float dep[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.use_depth) {
      const T* dptr = depth + bn * P.D * HW;
#pragma unroll
      for (int kz = 0; kz < 2; ++kz) {
        const int iz = t.iz0 + kz;
        const bool zin = iz >= 0 && iz < P.D;
        const float wz = zin ? (kz ? t.wz1 : t.wz0) : 0.f;
        const long zo = (long) min(max(iz, 0), P.D - 1) * HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
          const bool in = iy >= 0 && iy < P.fH && ix >= 0 && ix < P.fW;
          dep[j] += (in ? wz : 0.f) *
                    ldf(dptr, zo + (long) min(max(iy, 0), P.fH - 1) * P.fW + min(max(ix, 0), P.fW - 1));
        }
      }
    } else {
      const float w = (t.iz0 == 0 ? t.wz0 : 0.f) + (t.iz0 == -1 ? t.wz1 : 0.f);
      dep[0] = dep[1] = dep[2] = dep[3] = w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tx = (q & 1) ? txb : txa, ty = (q & 2) ? tyb : tya;
      if (tx < 0 || ty < 0) continue;
      const int le = n * ntile + ty * G.tiles_w + tx;
      const long bin = (long) b * NT + le;
      const long slot = (long) off[bin] + lbase[le] + atomicAdd(hist + le, 1);
      float* e = entries + slot * ES;
      *reinterpret_cast<float4*>(e) = make_float4(t.fx, t.fy, t.fz, 0.f);
      *reinterpret_cast<float4*>(e + 4) = make_float4(dep[0], dep[1], dep[2], dep[3]);
      // grad_out / (hit count + 1e-6), the camera-mean factor of bv2:512-514
      for (int chunk = 0; chunk < P.C / CH; ++chunk) {
        const uint64_t hw = hits[((long) b * V + vox) * (P.C / CH) + chunk];
        const float* g = gout + ((long) b * P.C + chunk * CH) * V + vox;
#pragma unroll
        for (int c4 = 0; c4 < CH; c4 += 4) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            v[k] = g[(long) (c4 + k) * V] / ((float) ((hw >> (4 * (c4 + k))) & 15) + 1e-6f);
          *reinterpret_cast<float4*>(e + 8 + chunk * CH + c4) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
}

// OT threads per tile = OT / 16 entry groups; MINW = minimum waves per SIMD the register
// allocation must allow (two 1024-thread workgroups per CU need 8).
template <typename T, int OT, int MINW>
__global__ void __launch_bounds__(OT, MINW)
lift_bwd_own_kernel(LiftParams P, BinGeom G, const T* __restrict__ feat,
                    const int* __restrict__ cnt, const int* __restrict__ off,
                    const float* __restrict__ entries, float* __restrict__ gdepth,
                    float* __restrict__ gfeat, const int* __restrict__ total, int cap) {
  if (*total > cap) return;
  extern __shared__ float lds[];
  const int C = P.C, D = P.use_depth ? P.D : 0;
  const int CS = C + 1;
  float* ft = lds;                 // [BTP][CS] feat tile
  float* gf = ft + BTP * CS;       // [BTP][CS] grad_feat accumulators
  float* gd = gf + BTP * CS;       // [D][BTP]  grad_depth accumulators
  const int tid = threadIdx.x;
  const int ntile = G.tiles_w * G.tiles_h;
  const int tile = blockIdx.x % ntile;
  const long bn = blockIdx.x / ntile;
  const int w0 = (tile % G.tiles_w) * BTW, h0 = (tile / G.tiles_w) * BTH;
  const long HW = (long) P.fH * P.fW;
  const int ES = 8 + C;

  for (int e = tid; e < BTP * C; e += OT) {
    const int c = e / BTP, p = e % BTP;
    const int h = h0 + p / BTW, w = w0 + p % BTW;
    ft[p * CS + c] = (h < P.fH && w < P.fW) ? ldf(feat, (bn * C + c) * HW + (long) h * P.fW + w) : 0.f;
  }
  for (int e = tid; e < BTP * CS; e += OT) gf[e] = 0.f;
  for (int e = tid; e < D * BTP; e += OT) gd[e] = 0.f;
  __syncthreads();

  const int grp = tid / GLN, gl = tid % GLN;
  const long first = off[blockIdx.x];
  const int n_ent = cnt[blockIdx.x];
  // Each group walks a CONTIGUOUS slice of the list (records were appended in voxel order, so
  // neighbours in the list are neighbouring voxels, which in the far field project onto the same
  // four pixels).  The grad_feat contributions of such a run are summed in registers and flushed
  // with one ds_add per pixel when the (ix0, iy0) base changes: same-address LDS float atomics
  // are slow, and the hot pixels are exactly the ones with long runs.
  constexpr int NGRP = OT / GLN;
  const int per = (n_ent + NGRP - 1) / NGRP;
  const int j_lo = min(n_ent, grp * per), j_hi = min(n_ent, j_lo + per);
  float run[4] = {0.f, 0.f, 0.f, 0.f};
  int rx = -0x40000000, ry = 0;
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = ry + (j >> 1), ix = rx + (j & 1);
      const bool own = iy >= h0 && iy < h0 + BTH && ix >= w0 && ix < w0 + BTW && iy < P.fH && ix < P.fW;
      if (own && gl < C && run[j] != 0.f) atomicAdd(gf + ((iy - h0) * BTW + (ix - w0)) * CS + gl, run[j]);
      run[j] = 0.f;
    }
  };
  constexpr int UB = 4;                          // entries per batch: their loads go out together
  for (int i0 = j_lo; i0 < j_hi; i0 += UB) {
    float4 fb[UB], db[UB];
    float gsb[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int iu = min(i0 + u, j_hi - 1);
      const float* e = entries + (first + iu) * ES;
      fb[u] = *reinterpret_cast<const float4*>(e);
      db[u] = *reinterpret_cast<const float4*>(e + 4);
      gsb[u] = (gl < C) ? e[8 + gl] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (i0 + u >= j_hi) continue;
      const float* e = entries + (first + i0 + u) * ES;
      const float4 f = fb[u];
      const float4 dp = db[u];
      const float flx = floorf(f.x), fly = floorf(f.y), flz = floorf(f.z);
      const int ix0 = (int) flx, iy0 = (int) fly, iz0 = (int) flz;
      const float wx1 = f.x - flx, wx0 = (flx + 1.0f) - f.x;
      const float wy1 = f.y - fly, wy0 = (fly + 1.0f) - f.y;
      const float wz1 = f.z - flz, wz0 = (flz + 1.0f) - f.z;
      const float wj[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
      const float dep[4] = {dp.x, dp.y, dp.z, dp.w};
      if (ix0 != rx || iy0 != ry) {              // group-uniform
        flush();
        rx = ix0; ry = iy0;
      }
      int pj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = iy0 + (j >> 1), ix = ix0 + (j & 1);
        const bool own = iy >= h0 && iy < h0 + BTH && ix >= w0 && ix < w0 + BTW && iy < P.fH && ix < P.fW;
        pj[j] = own ? (iy - h0) * BTW + (ix - w0) : -1;
      }
      float dot[4] = {0.f, 0.f, 0.f, 0.f};
      if (C <= GLN) {
        const float gs = gsb[u];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (pj[j] < 0) continue;
          run[j] = __builtin_fmaf(wj[j] * dep[j], gs, run[j]);
          if (gl < C) dot[j] = ft[pj[j] * CS + gl] * gs;
        }
      } else {
        for (int c0 = 0; c0 < C; c0 += GLN) {    // more than 16 channels: no run accumulation
          const int c = c0 + gl;
          if (c >= C) continue;
          const float gs = (c0 == 0) ? gsb[u] : e[8 + c];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (pj[j] < 0) continue;
            atomicAdd(gf + pj[j] * CS + c, wj[j] * dep[j] * gs);
            dot[j] = __builtin_fmaf(ft[pj[j] * CS + c], gs, dot[j]);
          }
        }
      }
      if (P.use_depth) {
        // channel dot products of the four taps, summed over the 16 lanes by recursive halving:
        // afterwards lane l holds tap (l >> 2); its lanes 0 / 1 (of 4) add the two depth planes
        int tj = 0;
        reduce_halving<4, GLN / 2, GLN, 4>(dot, gl, tj);
        const int pjs = tj == 0 ? pj[0] : (tj == 1 ? pj[1] : (tj == 2 ? pj[2] : pj[3]));
        const float wjs = tj == 0 ? wj[0] : (tj == 1 ? wj[1] : (tj == 2 ? wj[2] : wj[3]));
        const int sub = gl & 3;
        const int iz = iz0 + sub;
        if (sub < 2 && pjs >= 0 && iz >= 0 && iz < P.D)
          atomicAdd(gd + iz * BTP + pjs, (sub ? wz1 : wz0) * (wjs * dot[0]));
      }
    }
  }
  flush();
  __syncthreads();
  for (int e = tid; e < BTP * C; e += OT) {
    const int c = e / BTP, p = e % BTP;
    const int h = h0 + p / BTW, w = w0 + p % BTW;
    if (h < P.fH && w < P.fW) gfeat[(bn * C + c) * HW + (long) h * P.fW + w] = gf[p * CS + c];
  }
  if (P.use_depth && gdepth)
    for (int e = tid; e < D * BTP; e += OT) {
      const int dz = e / BTP, p = e % BTP;
      const int h = h0 + p / BTW, w = w0 + p % BTW;
      if (h < P.fH && w < P.fW) gdepth[(bn * P.D + dz) * HW + (long) h * P.fW + w] = gd[e];
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static constexpr int kEntriesPerVoxel = 3;     // list capacity; beyond it the tile fallback runs

static long bin_count(const VampLiftDesc* d) {
  return (long) d->B * d->N * ((d->fW + BTW - 1) / BTW) * ((d->fH + BTH - 1) / BTH);
}
static long entry_cap(const VampLiftDesc* d) {
  return (long) kEntriesPerVoxel * d->B * d->Z * d->Y * d->X;
}

size_t lift_bwd_bin_ws_bytes(const VampLiftDesc* d) {
  return align_up((size_t) (3 * bin_count(d) + 4) * sizeof(int), 256) +
         align_up((size_t) entry_cap(d) * (8 + d->C) * sizeof(float), 256);
}

template <typename T>
static int launch_bin_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                        const float* xs, const float* ys, const float* zs, const void* depth,
                        const void* feat, const float* gout, const uint64_t* hits, float* gdepth,
                        float* gfeat, void* scratch, const int** total_out, int* cap_out,
                        hipStream_t s) {
  const long nb = bin_count(d);
  int* cnt = static_cast<int*>(scratch);
  int* off = cnt + nb;
  int* fill = off + nb;
  int* total = fill + nb;
  float* entries = reinterpret_cast<float*>(static_cast<char*>(scratch) +
                                            align_up((size_t) (3 * nb + 4) * sizeof(int), 256));
  const long cap64 = entry_cap(d);
  const int cap = cap64 > 0x7fffffffL ? 0x7fffffff : (int) cap64;
  *total_out = total;
  *cap_out = cap;
  BinGeom G{(d->fW + BTW - 1) / BTW, (d->fH + BTH - 1) / BTH};
  if (hipMemsetAsync(cnt, 0, (size_t) nb * sizeof(int), s) != hipSuccess)
    return fail(VAMP_EHIP, "%s: hipMemsetAsync failed", __func__);
  dim3 grid((d->X + 63) / 64, (d->Y + kBinThreads / 64 - 1) / (kBinThreads / 64), d->Z * d->B);
  const size_t tab = (size_t) 2 * d->N * G.tiles_w * G.tiles_h * sizeof(int);
  if (tab > 60 * 1024) return fail(VAMP_EINVAL, "%s: too many (camera, tile) bins for the LDS histogram", __func__);
  const T* dp = static_cast<const T*>(depth);
#define VAMP_BIN(CH, FILLV)                                                                     \
  VAMP_TIMED(FILLV ? kProfLiftBwdFill : kProfLiftBwdCount, s,                                   \
             (lift_bwd_bin_kernel<T, CH, FILLV><<<grid, kBinThreads, tab, s>>>(                           \
                 P, G, mats, xs, ys, zs, dp, gout, hits, cnt, off, fill, entries, total, cap)))
  if (P.C == 4) VAMP_BIN(4, false); else if (P.C == 8) VAMP_BIN(8, false); else VAMP_BIN(16, false);
  if (int e = check_launch("lift_bwd_bin_kernel<count>")) return e;
  if (int e = launch_exclusive_scan(cnt, off, fill, (int) nb, total, s)) return e;
  if (P.C == 4) VAMP_BIN(4, true); else if (P.C == 8) VAMP_BIN(8, true); else VAMP_BIN(16, true);
#undef VAMP_BIN
  if (int e = check_launch("lift_bwd_bin_kernel<fill>")) return e;
  const int Dd = d->use_depth ? d->D : 0;
  const size_t lds = ((size_t) 2 * BTP * (d->C + 1) + (size_t) Dd * BTP) * sizeof(float);
  if (lds > 150 * 1024) return fail(VAMP_EINVAL, "%s: D too large for the LDS tile", __func__);
  const char* eot = getenv("VAMP_LIFT_OWN");
  const int variant = eot ? atoi(eot) : 0;
#define VAMP_OWN(OT, MINW)                                                                        \
  do {                                                                                            \
    auto k = lift_bwd_own_kernel<T, OT, MINW>;                                                    \
    if (lds > 64 * 1024 &&                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) \
      return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);                           \
    VAMP_TIMED(kProfLiftBwd, s, (k<<<(unsigned) nb, OT, lds, s>>>(                                \
        P, G, static_cast<const T*>(feat), cnt, off, entries, gdepth, gfeat, total, cap)));       \
  } while (0)
  if (variant == 1) VAMP_OWN(512, 1); else if (variant == 2) VAMP_OWN(1024, 8);
  else if (variant == 3) VAMP_OWN(512, 8); else if (variant == 4) VAMP_OWN(256, 1); else VAMP_OWN(1024, 1);
#undef VAMP_OWN
  return check_launch("lift_bwd_own_kernel");
}

int launch_lift_bwd_bin(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                        const float* zs, const void* depth, const void* feat, const float* gout,
                        const uint64_t* hits, float* gdepth, float* gfeat, void* scratch,
                        const int** total_out, int* cap_out, hipStream_t s) {
  const LiftParams P = to_params(d);
  if (d->in_dtype == VAMP_F32)
    return launch_bin_t<float>(d, P, mats, xs, ys, zs, depth, feat, gout, hits, gdepth, gfeat,
                               scratch, total_out, cap_out, s);
  return launch_bin_t<__hip_bfloat16>(d, P, mats, xs, ys, zs, depth, feat, gout, hits, gdepth, gfeat,
                                      scratch, total_out, cap_out, s);
}

}  // namespace vamp
