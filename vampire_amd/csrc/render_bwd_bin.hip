// Camera-branch render backward, scatter stage v3: bin-then-own.
//
// After cam_bwd_ray (render_bwd_brick.hip) has produced one record per sample
// {fx, fy, fz, w_i, dL/ds_i[0]}, the gradient w.r.t. the density / semantic / rgb volumes is
//     dV_c[tap] += w_tap * (c == 0 ? dL/ds_i[0] : w_i * G_c[ray])      over the sample's 8 taps.
//
//   count   thread per sample: the bricks (BX x BY x BZ voxels) its taps touch, wave-aggregated
//           fire-and-forget counter atomics (lanes are adjacent pixels -> mostly one brick)
//   scan    exclusive prefix sum -> per-brick list offsets                      (runtime.hip)
//   fill    same walk, wave-aggregated returned atomics -> slot; appends {fx,fy,fz,w_i} {g0,ray}
//   own     one 1024-thread workgroup per brick streams its list with 32-lane groups (lane =
//           channel): ds_add_f32 into an LDS copy of the brick (distinct banks per lane), then
//           stores the brick's 1+K+3 gradient channels -- every output element written once
//
// Traffic is proportional to the samples (each record read ~1.6x) instead of to the candidate
// boxes of the per-voxel gather (cam_bwd_gather: ~30x the output bytes).  If the lists would not
// fit the workspace the gather runs instead (device-side decision).
#include "render_common.hpp"

namespace vamp {

constexpr int KBX = 8, KBY = 4, KBZ = 4, KNV = KBX * KBY * KBZ;
constexpr int kOwnThreadsCam = 256;
constexpr int kChunk = 1024;             // list entries per work item

struct BrickGrid {
  int nbx, nby, nbz;
};

__device__ __forceinline__ void wave_count(int* __restrict__ counters, int bin, bool active) {
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

__device__ __forceinline__ int wave_reserve(int* __restrict__ counters, int bin, bool active) {
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

template <bool FILL>
__global__ void __launch_bounds__(256)
cam_bwd_bin_kernel(RenderParams P, BrickGrid G, const float* __restrict__ FX,
                   const float* __restrict__ FY, const float* __restrict__ FZ,
                   const float* __restrict__ Wbuf, const float* __restrict__ G0buf,
                   int* __restrict__ cnt, const int* __restrict__ off, int* __restrict__ fill,
                   float4* __restrict__ E4, float2* __restrict__ E2,
                   const int* __restrict__ total, int cap) {
  if (FILL && *total > cap) return;
  const long HW = (long) P.fH * P.fW;
  const int S = P.D - 1;
  const long nsamp = (long) P.B * P.N * S * HW;
  const long sidx = (long) blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = sidx < nsamp;
  const long sc = live ? sidx : nsamp - 1;
  const float fx = FX[sc];
  const bool inside = live && (fx == fx);                 // NaN marks a masked sample
  int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0, bz0 = 0, bz1 = 0, b = 0;
  long ray = 0;
  float fy = 0.f, fz = 0.f;
  if (inside) {
    fy = FY[sc]; fz = FZ[sc];
    const int ix0 = (int) floorf(fx), iy0 = (int) floorf(fy), iz0 = (int) floorf(fz);
    bx0 = ix0 / KBX; bx1 = min(ix0 + 1, P.X - 1) / KBX;
    by0 = iy0 / KBY; by1 = min(iy0 + 1, P.Y - 1) / KBY;
    bz0 = iz0 / KBZ; bz1 = min(iz0 + 1, P.Z - 1) / KBZ;
    const long bn = sc / (S * HW);
    b = (int) (bn / P.N);
    ray = bn * HW + sc % HW;
  }
  float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 e2 = make_float2(0.f, 0.f);
  if (FILL && inside) {
    e4 = make_float4(fx, fy, fz, Wbuf[sc]);
    e2 = make_float2(G0buf[sc], __int_as_float((int) ray));
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    // second brick along an axis only when the +1 tap crosses a brick boundary
    const bool act = inside && (!(q & 1) || bx1 != bx0) && (!(q & 2) || by1 != by0) &&
                     (!(q & 4) || bz1 != bz0);
    if (!__any(act)) continue;                              // wave-uniform
    const int bx = (q & 1) ? bx1 : bx0, by = (q & 2) ? by1 : by0, bz = (q & 4) ? bz1 : bz0;
    const int bin = act ? ((b * G.nbz + bz) * G.nby + by) * G.nbx + bx : 0;
    if (!FILL) {
      wave_count(cnt, bin, act);
    } else {
      const int rank = wave_reserve(fill, bin, act);
      if (act) {
        const long slot = (long) off[bin] + rank;
        E4[slot] = e4;
        E2[slot] = e2;
      }
    }
  }
}

// One workgroup per work item = (brick, chunk of <= kChunk list entries).  The chunk is staged in
// LDS with one coalesced read; each 32-lane group (lane = channel) then walks a CONTIGUOUS run of
// it.  Neighbouring entries are neighbouring pixels, which in the near field land on the same
// eight taps for dozens of entries: their contributions are summed in registers and flushed to
// the LDS brick only when the tap base changes (same-address LDS float atomics serialise badly).
// A brick owned by a single work item is stored; a brick split over several is accumulated into
// the (pre-zeroed) outputs with float atomics.
template <int CP4>
__global__ void __launch_bounds__(kOwnThreadsCam)
cam_bwd_own_kernel(RenderParams P, BrickGrid G, const int* __restrict__ cnt,
                   const int* __restrict__ off, const int* __restrict__ work,
                   const int* __restrict__ nwork, const float4* __restrict__ E4,
                   const float2* __restrict__ E2, const float* __restrict__ Gcl,
                   float* __restrict__ gdens, float* __restrict__ gsem, float* __restrict__ grgb,
                   const int* __restrict__ total, int cap, int dbg) {
  if (*total > cap || (int) blockIdx.x >= *nwork) return;
  constexpr int CP = CP4 * 4;
  constexpr int STRIDE = CP + 1;                 // odd: lane c of any voxel -> its own bank
  constexpr int NG = kOwnThreadsCam / 32;        // groups of 32 lanes, lane = channel
  constexpr int PER = kChunk / NG;               // contiguous entries per group
  __shared__ float acc[KNV * STRIDE];
  __shared__ float4 t4[kChunk];
  __shared__ float2 t2[kChunk];
  const int tid = threadIdx.x;
  const int bin = work[2 * blockIdx.x], chunk = work[2 * blockIdx.x + 1];
  const int bx = bin % G.nbx, by = (bin / G.nbx) % G.nby, bz = (bin / (G.nbx * G.nby)) % G.nbz;
  const int b = bin / (G.nbx * G.nby * G.nbz);
  const int x0 = bx * KBX, y0 = by * KBY, z0 = bz * KBZ;
  const int x1 = min(P.X, x0 + KBX) - 1, y1 = min(P.Y, y0 + KBY) - 1, z1 = min(P.Z, z0 + KBZ) - 1;
  for (int i = tid; i < KNV * STRIDE; i += kOwnThreadsCam) acc[i] = 0.f;
  const int n_all = cnt[bin];
  const long first = (long) off[bin] + (long) chunk * kChunk;
  const int n_ent = min(kChunk, n_all - chunk * kChunk);
  for (int i = tid; i < n_ent; i += kOwnThreadsCam) {
    t4[i] = E4[first + i];
    t2[i] = E2[first + i];
  }
  __syncthreads();

  const int nch = 1 + P.K + 3;
  const int grp = tid >> 5, c = tid & 31;
  const int cc = min(c, nch - 1);                // idle lanes load a legal address, add nothing
  float run[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) run[k] = 0.f;
  int rx = -0x40000000, ry = 0, rz = 0;          // tap base of the open run
  // flush the open run: which of the two taps per axis fall inside the brick is decided once,
  // the eight LDS addresses are the base plus constant offsets
  auto flush = [&]() {
    const bool ax0 = rx >= x0 && rx <= x1, ax1 = rx + 1 >= x0 && rx + 1 <= x1;
    const bool ay0 = ry >= y0 && ry <= y1, ay1 = ry + 1 >= y0 && ry + 1 <= y1;
    const bool az0 = rz >= z0 && rz <= z1, az1 = rz + 1 >= z0 && rz + 1 <= z1;
    float* base = acc + (((rz - z0) * KBY + (ry - y0)) * KBX + (rx - x0)) * STRIDE + c;
    const bool lane_on = c < nch;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool in = ((k & 1) ? ax1 : ax0) && ((k & 2) ? ay1 : ay0) && ((k & 4) ? az1 : az0);
      if (in && lane_on && run[k] != 0.f && !(dbg & 1))
        atomicAdd(base + ((k & 1) + ((k >> 1) & 1) * KBX + (k >> 2) * KBX * KBY) * STRIDE, run[k]);
      run[k] = 0.f;
    }
  };
  const int j_lo = grp * PER, j_hi = min(n_ent, j_lo + PER);
  constexpr int UB = 16;
  for (int j0 = j_lo; j0 < j_hi; j0 += UB) {
    float val[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {                // the dependent G loads of a batch go out together
      const int jj = min(j0 + u, j_hi - 1);
      const float2 g = t2[jj];
      const float gv = (dbg & 4) ? 1.f : Gcl[(long) __float_as_int(g.y) * CP + cc];
      val[u] = (c == 0) ? g.x : t4[jj].w * gv;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (j0 + u >= j_hi) continue;
      const float4 e = t4[j0 + u];
      const float flx = floorf(e.x), fly = floorf(e.y), flz = floorf(e.z);
      const int ix0 = (int) flx, iy0 = (int) fly, iz0 = (int) flz;
      if (ix0 != rx || iy0 != ry || iz0 != rz) {  // group-uniform
        flush();
        rx = ix0; ry = iy0; rz = iz0;
      }
      const float wx1 = e.x - flx, wx0 = (flx + 1.0f) - e.x;
      const float wy1 = e.y - fly, wy0 = (fly + 1.0f) - e.y;
      const float wz1 = e.z - flz, wz0 = (flz + 1.0f) - e.z;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        run[k] = __builtin_fmaf(((k & 1) ? wx1 : wx0) * ((k & 2) ? wy1 : wy0) * ((k & 4) ? wz1 : wz0),
                                val[u], run[k]);
    }
  }
  flush();
  __syncthreads();
  const bool sole = n_all <= kChunk;              // this work item owns the whole brick
  const long V = (long) P.Z * P.Y * P.X;
  for (int e = tid; e < nch * KNV; e += kOwnThreadsCam) {
    const int ch = e / KNV, lv = e % KNV;
    const int lx = lv % KBX, ly = (lv / KBX) % KBY, lz = lv / (KBX * KBY);
    const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
    if (x > x1 || y > y1 || z > z1) continue;
    const long vox = ((long) z * P.Y + y) * P.X + x;
    const float v = acc[lv * STRIDE + ch];
    float* dst = (ch == 0) ? gdens + (long) b * V + vox
                 : (ch <= P.K) ? gsem + ((long) b * P.K + (ch - 1)) * V + vox
                               : grgb + ((long) b * 3 + (ch - 1 - P.K)) * V + vox;
    if (dbg & 2) { if (v == 12345.f) *dst = v; }
    else if (sole) *dst = v;
    else if (v != 0.f) atomicAdd(dst, v);
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static long n_bricks(const VampRenderDesc* d) {
  return (long) d->B * ((d->X + KBX - 1) / KBX) * ((d->Y + KBY - 1) / KBY) * ((d->Z + KBZ - 1) / KBZ);
}
static long n_samples(const VampRenderDesc* d) {
  return (long) d->B * d->N * (d->D - 1) * d->fH * d->fW;
}
static long entry_cap(const VampRenderDesc* d) {
  const long c = 2 * n_samples(d);               // typical need: ~0.6-0.8 entries per sample
  return c > 0x7fffffffL ? 0x7fffffffL : c;
}

static long max_work(const VampRenderDesc* d) { return n_bricks(d) + entry_cap(d) / kChunk + 1; }

size_t cam_bwd_bin_bytes(const VampRenderDesc* d) {
  return align_up((size_t) (3 * n_bricks(d) + 4) * sizeof(int), 256) +
         align_up((size_t) entry_cap(d) * sizeof(float4), 256) +
         align_up((size_t) entry_cap(d) * sizeof(float2), 256) +
         align_up((size_t) 2 * max_work(d) * sizeof(int), 256);
}

int launch_cam_bwd_bin(const VampRenderDesc* d, const RenderParams& P, const float* FX,
                       const float* FY, const float* FZ, const float* Wbuf, const float* G0buf,
                       const float* Gcl, float* gdens, float* gsem, float* grgb, void* scratch,
                       const int** total_out, int* cap_out, hipStream_t s) {
  const long nb = n_bricks(d);
  char* p = static_cast<char*>(scratch);
  int* cnt = reinterpret_cast<int*>(p);
  int* off = cnt + nb;
  int* fill = off + nb;
  int* total = fill + nb;
  p += align_up((size_t) (3 * nb + 4) * sizeof(int), 256);
  float4* E4 = reinterpret_cast<float4*>(p);
  p += align_up((size_t) entry_cap(d) * sizeof(float4), 256);
  float2* E2 = reinterpret_cast<float2*>(p);
  p += align_up((size_t) entry_cap(d) * sizeof(float2), 256);
  int* work = reinterpret_cast<int*>(p);
  int* nwork = total + 1;
  const int cap = (int) entry_cap(d);
  *total_out = total;
  *cap_out = cap;
  BrickGrid G{(d->X + KBX - 1) / KBX, (d->Y + KBY - 1) / KBY, (d->Z + KBZ - 1) / KBZ};
  if (hipMemsetAsync(cnt, 0, (size_t) nb * sizeof(int), s) != hipSuccess)
    return fail(VAMP_EHIP, "%s: hipMemsetAsync failed", __func__);
  const long ns = n_samples(d);
  const unsigned grid = (unsigned) ((ns + 255) / 256);
  VAMP_TIMED(kProfCamBwdCount, s, (cam_bwd_bin_kernel<false><<<grid, 256, 0, s>>>(
      P, G, FX, FY, FZ, Wbuf, G0buf, cnt, off, fill, E4, E2, total, cap)));
  if (int e = check_launch("cam_bwd_bin_kernel<count>")) return e;
  if (int e = launch_exclusive_scan(cnt, off, fill, (int) nb, total, s)) return e;
  VAMP_TIMED(kProfCamBwdFill, s, (cam_bwd_bin_kernel<true><<<grid, 256, 0, s>>>(
      P, G, FX, FY, FZ, Wbuf, G0buf, cnt, off, fill, E4, E2, total, cap)));
  if (int e = check_launch("cam_bwd_bin_kernel<fill>")) return e;
  if (int e = launch_build_worklist(cnt, (int) nb, kChunk, work, nwork, s)) return e;
  {
    // bricks split over several work items accumulate with float atomics: pre-zero the outputs
    const size_t V = (size_t) d->B * d->Z * d->Y * d->X * sizeof(float);
    ProfScope sc;
    prof_begin(kProfMemset, s, &sc);
    const bool ok = hipMemsetAsync(gdens, 0, V, s) == hipSuccess &&
                    hipMemsetAsync(gsem, 0, V * d->K, s) == hipSuccess &&
                    hipMemsetAsync(grgb, 0, V * 3, s) == hipSuccess;
    prof_end(s, &sc);
    if (!ok) return fail(VAMP_EHIP, "%s: hipMemsetAsync failed", __func__);
  }
  const unsigned ogrid = (unsigned) max_work(d);
#define VAMP_OWN(CP4)                                                                            \
  VAMP_TIMED(kProfCamBwdOwn, s, (cam_bwd_own_kernel<CP4><<<ogrid, kOwnThreadsCam, 0, s>>>(       \
      P, G, cnt, off, work, nwork, E4, E2, Gcl, gdens, gsem, grgb, total, cap, getenv("VAMP_DBG") ? atoi(getenv("VAMP_DBG")) : 0)))
  if (P.CP == 12) VAMP_OWN(3); else if (P.CP == 24) VAMP_OWN(6); else VAMP_OWN(8);
#undef VAMP_OWN
  return check_launch("cam_bwd_own_kernel");
}

}  // namespace vamp
