// LIFT backward as a cell list ("sort, then own").  Autograd of base_vampire2.py:507-514
// (grid_sampler_3d backward + the camera mean):
//
//   count   done by the FORWARD kernel in grad mode (lift.hip, lift_emit_pair; lift_pairs_kernel for a
//           backward whose forward did not): every valid (voxel, camera) pair increments the counter of
//           its cell = (camera, floor tap row + 1, floor tap column + 1), (fH + 1) x (fW + 1) cells per
//           camera, and leaves its taps {wx1, wy1, wz1, iz0} at (image, voxel)
//   scan    exclusive prefix sum of the counters -> cell start offsets (runtime.hip)
//   fill    thread per voxel, NO projection: every pair of the voxel (camera mask + taps + depth samples from the
//           forward) goes to the next slot of its cell (slots handed out by atomics on the cell cursor) as ONE
//           record [taps | depth samples | the voxel's row grad_out / (hits + 1e-6)], 32 + 4 C bytes: everything
//           the gather needs of a pair, contiguous, in the order it will read it
//   gather  one workgroup per STRIP of 16 consecutive pixels of a feature-map row.  The pairs that touch
//           the strip are those of the cell rows iy, iy + 1, columns x0 .. x0 + 16: two contiguous ranges
//           of the cell-ordered records.  The workgroup stages them in LDS (lane = pair for the taps, four lanes
//           per pair for the gradient row: plain streaming reads, every load of a chunk in flight at once and
//           none of them dependent on another), then consumes them lane = (pixel, channel
//           quad): a pixel's pairs are one contiguous sub-range (its two cells are adjacent), the four
//           waves take every fourth pair of it, the channel dot product is a 4-lane DPP sum, the depth
//           terms go to the strip's LDS tile of 64-bit fixed-point sums.  No float atomics, no cross-lane folds;
//           depth-gradient planes leave as 64-byte runs.
//
// Round 6 changed what a pair carries (review item 2: "change the tile, not the order").  Through round 5 the fill
// wrote a 41 MB channel-last TABLE of normalised gradient rows, one per voxel, and 20-byte pair records beside it;
// the gather loaded a pair's voxel index, then -- a second, dependent round trip, a random 64-byte line per pair --
// its table row, and interpolated the pair's depth sample from a [D][16] tile of depth columns it had loaded at the
// head of every strip.  Now the forward keeps the four depth samples it computes anyway (16 bytes per pair), the
// fill writes the row INTO the record, and the gather's staging is one round trip of sequential reads: no table,
// no voxel indices, no depth tile (the depth columns are only loaded, late, by the fused softmax backward of the
// logits entry).
//
// (Round 3 ran one WAVE per pixel with lane = pair: 67 584 waves at cfg-B of which half held fewer
// than ten pairs, each paying the column staging, the range loads, a 64-lane fold of 16 channels and
// 16-byte pieces of 86 depth planes -- 5.85 M L2 requests and 92 us for 2.4 M pair visits; every pair
// was read by four pixels.  A strip reads a pair 2.1 times and has 16x fewer, 16x fuller units of work.)
//
// No float atomics on global memory, no memset of the outputs, no layout transposes.
#include "lift_common.hpp"

#include <algorithm>
#include <type_traits>

namespace vamp {

// ---------------------------------------------------------------------------
// fill: thread per voxel, the 64 lanes of a wave are 64 x-consecutive voxels
// ---------------------------------------------------------------------------
template <int CH>
__global__ void __launch_bounds__(256, 4)
lift_bwd_fill_kernel(LiftParams P, int cw, int ch, const float* __restrict__ gout,
                     const uint64_t* __restrict__ hits, const unsigned* __restrict__ amask,
                     const float4* __restrict__ ptaps, const int* __restrict__ pcell,
                     int* __restrict__ cnt, const int* __restrict__ off, const int* __restrict__ boff,
                     float4* __restrict__ recs, int* __restrict__ rowq, int bn_lo, int bn_hi) {
  const int tid = threadIdx.x, lane = tid & 63;
  // A duty of the first workgroup, beside its voxels: the order in which the gather takes the image
  // rows -- those with the most pairs first (counting sort by the bit length of a row's pair count), so
  // that its long-running workgroups start early and the light ones fill the tail.
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    __shared__ int cls[34];
    if (tid < 34) cls[tid] = 0;
    __syncthreads();
    const int nrow = (bn_hi - bn_lo) * P.fH;
    auto row_class = [&](int r) {
      const long c0 = ((long) bn_lo * ch + (long) (r / P.fH) * ch + r % P.fH) * cw, c1 = c0 + 2 * cw;
      const int w = (off[c1] + boff[c1 / kScanTile]) - (off[c0] + boff[c0 / kScanTile]);
      return 32 - __clz(max(w, 0));                  // 0 .. 32, heavy rows get high classes
    };
    for (int r = tid; r < nrow; r += 256) atomicAdd(cls + row_class(r), 1);
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int k = 32; k >= 0; --k) { const int n = cls[k]; cls[k] = run; run += n; }
    }
    __syncthreads();
    for (int r = tid; r < nrow; r += 256) rowq[(long) bn_lo * P.fH + atomicAdd(cls + row_class(r), 1)] = bn_lo * P.fH + r;
  }
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + (tid >> 6);
  const int z = blockIdx.z % P.Z, b = blockIdx.z / P.Z;
  const bool live = x < P.X && y < P.Y;
  const int xc = min(x, P.X - 1), yc = min(y, P.Y - 1);
  const long V = (long) P.Z * P.Y * P.X;
  const long vox = ((long) z * P.Y + yc) * P.X + xc;

  // images [bn_lo, bn_hi) of the flattened (sample, camera) index
  const int n_lo = max(0, bn_lo - b * P.N), n_hi = min(P.N, bn_hi - b * P.N);
  const unsigned range = n_hi > n_lo ? (((1u << (n_hi - n_lo)) - 1u) << n_lo) : 0u;
  const unsigned vmask = live ? (amask[(long) b * V + vox] & range) : 0u;
  if (!__any(vmask != 0u)) return;                   // (9 % of the voxels are seen by no camera: whole waves of them leave here)

  const int nchunk = P.C / CH;
  const int RS = 2 + P.C / 4;                        // float4 pieces per record
  // The records of a wave's pairs leave through LDS: a lane's 32 + 4 C bytes are one record, but stored by the lane
  // itself they are RS separate 16-byte pieces 16 RS bytes apart (the first build of this kernel: 62 us).  Staged
  // [lane][piece] and written piece by piece with consecutive lanes on consecutive pieces, a record is one run, and
  // the records of a run of lanes that share a cell -- consecutive slots -- are one longer run.
  extern __shared__ float4 fstage[];                 // [4 waves][64 lanes][RS]
  __shared__ int fslot[4][64];
  float4* my = fstage + (size_t) (tid >> 6) * 64 * RS;
  int* myslot = fslot[tid >> 6];

  // the voxel's row grad_out / (hit count + 1e-6), the camera-mean factor of bv2:512-514 -- the same for every
  // camera of the voxel -- into the lane's LDS row behind the two tap pieces; its loads are in flight under the atomics
  if (vmask != 0u) {
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const uint64_t hw = hits[((long) b * V + vox) * nchunk + chunk];
      const float* g = gout + ((long) b * P.C + chunk * CH) * V + vox;
      float v[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k)
        v[k] = g[(long) k * V] * __builtin_amdgcn_rcpf((float) ((hw >> (4 * k)) & 15) + 1e-6f);   // 1 ulp: gradients are held to 1e-4
#pragma unroll
      for (int c4 = 0; c4 < CH; c4 += 4)
        my[lane * RS + 2 + (chunk * CH + c4) / 4] = make_float4(v[c4], v[c4 + 1], v[c4 + 2], v[c4 + 3]);
    }
  }

#ifndef VAMP_FILL_NB
#define VAMP_FILL_NB 4        // (8: 122 registers, fill 43 us; 4: 74 registers, 40.5 us; 3 / 2: the same)
#endif
  constexpr int NB = VAMP_FILL_NB;                   // cameras per batch: their atomics are in flight together
  for (int n0 = n_lo; n0 < n_hi; n0 += NB) {
    int base[NB], start[NB];
    long cellk[NB];
    float4 tapk[NB], depk[NB];
    unsigned actm = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int n = n0 + k;
      base[k] = 0;
      start[k] = lane;
      cellk[k] = 0;
      tapk[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      depk[k] = tapk[k];
      if (n >= n_hi) continue;                   // uniform
      const bool act = (vmask >> n) & 1u;
      if (!__any(act)) continue;                 // uniform: nobody in this wave has a pair with camera n
      const long bn = (long) b * P.N + n;
      long cell = 0;
      if (act) {
        tapk[k] = ptaps[(bn * V + vox) * 2];
        depk[k] = ptaps[(bn * V + vox) * 2 + 1];
        const int pc = pcell[bn * V + vox];
        cell = (bn * ch + (pc >> 16)) * cw + (pc & 0xffff);
      }
      cellk[k] = cell;
      const LaneRun r = lane_run(act, cell, lane);
#ifdef VAMP_FILL_NOATOMIC          // (measurement build: wrong slots)
      if (r.head) base[k] = 0;
#else
      if (r.head) base[k] = atomicAdd(cnt + cell, r.len);
#endif
      if (act) { actm |= 1u << k; start[k] = r.start; }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int n = n0 + k;
      if (n >= n_hi) continue;                   // uniform
      if (!__any((actm >> k) & 1u)) continue;    // uniform
      const int rb = __shfl(base[k], start[k], 64);
      const bool act = (actm >> k) & 1u;
      const long slot = act ? (long) off[cellk[k]] + boff[cellk[k] / kScanTile] + rb + (lane - start[k]) : -1;
      // (a wave writes and reads its own slab: fence + wave barrier, no instruction on gfx9)
      my[lane * RS] = tapk[k];
      my[lane * RS + 1] = depk[k];
      myslot[lane] = (int) slot;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int i = lane; i < 64 * RS; i += 64) {
        const int r = i / RS, q = i - r * RS;
        const int sl = myslot[r];
#ifdef VAMP_FILL_NOSTORE           // (measurement build: no records)
        if (sl == -12345) recs[(long) sl * RS + q] = my[i];
#else
        if (sl >= 0) recs[(long) sl * RS + q] = my[i];
#endif
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------
// gather: one workgroup per strip of kS pixels of one feature-map row
// ---------------------------------------------------------------------------
constexpr int kS = 16;               // pixels per strip: depth planes move as 64-byte runs
constexpr int kW = 4;                // waves per workgroup
constexpr int kSlots = 4 * kW;       // lanes a pixel's pairs are dealt to (4 per wave)
constexpr int kRow = 24;             // floats per staged pair: 16 channels + 2 x {w dep, w wz0, w wz1, iz0}
constexpr int kFix = 44;             // fixed-point depth sums: |term| < 2^41, 2^10 .. 2^20 terms fit

__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_amdgcn_update_dpp(0.f, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// exponent e with |v| < 2^e (v = 0 -> a very small exponent)
__device__ __forceinline__ int exp_above(float v) {
  int e;
  frexpf(fmaxf(fabsf(v), 1e-30f), &e);
  return e;
}
// round(w * 2^sc) as a 64-bit integer, |result| < 2^51: the integer sits in the mantissa of
// w * 2^sc + 1.5 * 2^52 (five instructions; a float -> int64 cast is a dozen)
__device__ __forceinline__ unsigned long long fixed_of(float w, int sc) {
  const double x = ldexp((double) w, sc) + 6755399441055744.0;
  return (unsigned long long) (__double_as_longlong(x) - 0x4338000000000000LL);
}

#ifdef VAMP_LIFT_STAMPS
// diagnostic build only (tools/debug/lift_stamps.py; -DVAMP_LIFT_STAMPS): per-workgroup phase stamps of the strip gather
__device__ long long g_stamps[16384 * 8];
#define STAMP() __builtin_amdgcn_s_memtime()
#endif

static size_t strip_lds_floats(int D, int cap) {
  // 64-bit gtile + stage (staged pairs | feature tile | lane-reduction buffer | the softmax backward's depth columns)
  // + cell starts + per-pixel scale exponents + softmax partials + chunk maxima
  return (size_t) D * kS * 2 + (size_t) std::max(std::max(cap * kRow, kW * kS * 16), D * kS) + 2 * (kS + 2) + kS + 17 * kS + 8;
}

// The depth gradient of the strip is summed in 64-bit fixed point in LDS (ds_add_u64 is served at
// 11-19 cycles per wave instruction on gfx950, ds_add_f32 at 100-190, and a read-add-write per pair
// costs two LDS round trips in a loop that is nothing but latency): a term w * dot is bounded by
// max|g| * sum_c |feat_c| of its pixel, so with the power-of-two scale 2^(kFix - e_g - e_f[pixel]) every
// term is below 2^41 and keeps >= 20 bits under a typical term; integer adds are associative, so
// the sums have the same bits every run.  e_g follows the largest table row staged so far (+3 bits
// of headroom); when a later chunk exceeds it the tile is shifted down once.
template <typename T, bool VEC>
__global__ void __launch_bounds__(kW * 64, 4)
lift_bwd_strip_kernel(LiftParams P, int cw, int ch, int spr, int xgroup, int bn_lo, int cap,
                      const T* __restrict__ depth, const T* __restrict__ feat,
                      const int* __restrict__ off, const int* __restrict__ boff,
                      const float4* __restrict__ recs, const int* __restrict__ rowq, int* __restrict__ cnt,
                      float* __restrict__ gdepth, float* __restrict__ gfeat, int softmax_bwd, int fcl) {
  extern __shared__ float smem[];
  constexpr int NT = kW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int C = P.C, D = P.use_depth ? P.D : 0;
  const int DS = D * kS;
  long long* gtile = reinterpret_cast<long long*>(smem);            // [D][kS] the depth columns' gradients, fixed point
  float* stage = smem + 2 * DS;                                     // [cap][kRow] staged pairs
  float* dtile = stage;                                             // [D][kS] the depth columns (softmax backward only, at the end)
  int* offs = reinterpret_cast<int*>(stage + max(max(cap * kRow, kW * kS * 16), DS));   // [2][kS + 2] cell starts
  int* efp = offs + 2 * (kS + 2);                                   // [kS] exponent of sum_c |feat_c|
  float* red = reinterpret_cast<float*>(efp + kS);                  // [16 + 1][kS]
  float* cmax = red + 17 * kS;                                      // [kW] chunk maxima

  // the strips of one image row run on one XCD: vertical and horizontal neighbours share cells
  const unsigned lin = xcd_grouped(blockIdx.x, gridDim.x, xgroup);
  const int sx = lin % spr;
  const int row = rowq[(long) bn_lo * P.fH + lin / spr];            // heavy rows first (fill kernel)
  const int iy = row % P.fH;
  const long bn = row / P.fH;
  const int b = (int) (bn / P.N);
  const int x0 = sx * kS;
  const int np = min(kS, P.fW - x0);                 // pixels of a ragged last strip
  const long HW = (long) P.fH * P.fW;
  const long V = (long) P.Z * P.Y * P.X;
  const long pix0 = (long) iy * P.fW + x0;
  const int p = lane >> 2, sl = wv * 4 + (lane & 3); // consume phase: lane = (pixel, slot)

#ifdef VAMP_LIFT_STAMPS
  long long st0 = STAMP(), st_stage = 0, st_cons = 0, st2 = 0;
  const long long rt0 = wall_clock64();
#endif
  // start offsets of the cells (row iy + h, column x0 + j), j = 0 .. kS + 1.  A pair of cell (row,
  // column) has iy0 = row - 1, ix0 = column - 1; the cells are linear in (row, column), so the end of a
  // row is the start of the next.
  if (tid < 2 * (kS + 2)) {
    const int h = tid / (kS + 2), j = tid % (kS + 2);
    const long c = (bn * ch + iy + h) * cw + min(x0 + j, cw);
    offs[tid] = off[c] + boff[c / kScanTile];
    // the fill's cursors (= the counters the next forward counts into) go back to zero: this strip owns
    // the cells of its row and columns, the last strip of a row also column fW, the last row also row fH
    if ((h == 0 || iy == P.fH - 1) && (j < np || (j == np && x0 + np == P.fW))) cnt[c] = 0;
  }
  for (int e = tid; e < DS / 2; e += NT) reinterpret_cast<float4*>(gtile)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  // one feature of the strip: channel-first planes [C][HW], or (fcl: VAMP_LIFTBWD_FEAT_CHANNEL_LAST, uniform) the
  // caller's channel-last rows [HW][C] -- a strip's 16 channels x kS pixels are then one contiguous run
  auto feat_at = [&](int c, int pp) -> float {
    return fcl ? ldf(feat, (bn * HW + pix0 + pp) * C + c) : ldf(feat, (bn * C + c) * HW + pix0 + pp);
  };
  {
    // sum_c |feat_c| of every pixel over ALL channels: the bound of the depth terms
    const int pp = tid % kS;
    float sa = 0.f;
    for (int c = tid / kS; c < C; c += NT / kS) sa += fabsf(pp < np ? feat_at(c, pp) : 0.f);
    red[tid] = sa;
  }
  __syncthreads();
  if (tid < kS) {
    float t = 0.f;
    for (int k = 0; k < NT / kS; ++k) t += red[k * kS + tid];
    efp[tid] = exp_above(t);
  }
  __syncthreads();
#ifdef VAMP_LIFT_STAMPS
  long long st1 = STAMP();
#endif

  const int n0 = offs[np + 1] - offs[0];             // pairs of cell row iy
  const int NP = n0 + offs[kS + 2 + np + 1] - offs[kS + 2];   // pairs of both cell rows
  const int efl = efp[p];
  int eg = -1000;                                    // uniform: current exponent of the table rows

  for (int c0 = 0; c0 < C; c0 += 16) {               // channel chunk
    const int nq = min(4, (C - c0) / 4);             // float4 pieces of this chunk
    // the pixel's 16 features of this chunk, through LDS ([pixel][channel]: one load per thread)
    __syncthreads();
    {
      // (consecutive threads = consecutive pixels of a channel plane, or consecutive channels of a channel-last row)
      const int cc = fcl ? tid % 16 : tid / kS, pp = fcl ? tid / 16 : tid % kS;
      stage[pp * 16 + cc] = (c0 + cc < C && pp < np) ? feat_at(c0 + cc, pp) : 0.f;
    }
    __syncthreads();
    float ft[16], acc[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 f = reinterpret_cast<const float4*>(stage + p * 16)[k];
      ft[4 * k] = f.x; ft[4 * k + 1] = f.y; ft[4 * k + 2] = f.z; ft[4 * k + 3] = f.w;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    // this pixel's pairs in the concatenation [cell row iy | cell row iy + 1]: cells (x0 + p) [the
    // pixel is their x1 tap, role A] and (x0 + p + 1) [x0 tap, role B] of either row
    int lo_h[2], mid_h[2], hi_h[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int* oh = offs + h * (kS + 2);
      const int base = h ? n0 - oh[0] : -oh[0];
      lo_h[h] = oh[p] + base;
      mid_h[h] = oh[p + 1] + base;
      hi_h[h] = p < np ? oh[p + 2] + base : lo_h[h];
    }
    // The loads of the stage phase run one chunk ahead, and none depends on another: a pair's record sits at its
    // POSITION in the cell-ordered list (round 6: the gradient row travels in the record; through round 5 a pair's
    // voxel index had to arrive before its table row could be asked for).
    // stage phase A: lanes 0..31 of a wave = pair 32 wv + lane of the chunk (taps and depth samples; the lane works
    // out the pair's terms for its two pixels); phase B: four lanes per pair, each one 16-byte piece of the row
    const int jA = 32 * wv + lane;
    const bool laneA = lane < 32 && jA < cap;
    const int RS = 2 + C / 4;                        // float4 pieces per record
    auto rec_at = [&](int t) -> long { return (long) (t >= n0 ? offs[kS + 2] + (t - n0) : offs[0] + t) * RS; };
    auto load_taps = [&](int tc, float4& rc, float4& dp) {
      rc = make_float4(0.f, 0.f, 0.f, 0.f);
      dp = rc;
      const int t = tc + jA;
      if (laneA && t < NP) {
        const long pos = rec_at(t);
        rc = recs[pos];
        dp = recs[pos + 1];
      }
    };
    auto load_rows = [&](int tc, float4 (&g)[2]) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int jj = 32 * wv + 16 * s2 + (lane >> 2), k = lane & 3;
        g[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
        // the voxel's row of grad_out / (hits + 1e-6), this chunk of channels
        if (jj < cap && tc + jj < NP && k < nq) g[s2] = recs[rec_at(tc + jj) + 2 + c0 / 4 + k];
      }
    };
    float4 rcA, dpA, gA[2];
    load_taps(0, rcA, dpA);
    load_rows(0, gA);
    for (int t0 = 0; t0 < NP; t0 += cap) {
      const int nst = min(cap, NP - t0);
      __syncthreads();                               // the previous chunk (or the feature tile) is consumed
#ifdef VAMP_LIFT_STAMPS
      long long sa = STAMP();
#endif
      // ---- stage chunk [t0, t0 + nst) from the registers
      float gm = 0.f;
      if (laneA && jA < nst) {
        const int h = t0 + jA >= n0;
        const float4 rc = rcA;
        // the pair's taps: fractional coordinates (w1 of each axis), the lower depth plane and the
        // cell column; the weights of the lower taps are taken as 1 - w1 (the forward's
        // (floor + 1) - f up to an ulp: gradients are held to 1e-4)
        const int pk = __float_as_int(rc.w);
        const int iz0 = (pk & 0xffff) - 1, col = pk >> 16;
        const float wz1v = rc.z, wz0v = 1.0f - rc.z;
        const float wy = h ? 1.0f - rc.y : rc.y;
        const bool z0in = iz0 >= 0 && iz0 < D, z1in = iz0 + 1 >= 0 && iz0 + 1 < D;
        // the forward's depth samples at the pair's four pixel taps, [y tap][x tap]: this strip's row is the pair's
        // y1 tap when the pair comes from cell row iy (h = 0), its y0 tap from cell row iy + 1
        const float dpy[2] = {h ? dpA.x : dpA.z, h ? dpA.y : dpA.w};     // [x tap] of this row
        float* st = stage + jA * kRow;
#pragma unroll
        for (int role = 0; role < 2; ++role) {       // role A: the pixel is the x1 tap, role B: the x0 tap
          const int pr = col - role - x0;
          const bool pin = pr >= 0 && pr < np;
          const float wj = wy * (role ? 1.0f - rc.x : rc.x);
          const float dep = dpy[1 - role];
          reinterpret_cast<float4*>(st)[4 + role] =
              make_float4(pin ? wj * dep : 0.f, (pin && z0in) ? wj * wz0v : 0.f, (pin && z1in) ? wj * wz1v : 0.f,
                          __int_as_float(iz0));
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int j = 32 * wv + 16 * s2 + (lane >> 2);
        const float4 g4 = gA[s2];
        gm = fmaxf(gm, fmaxf(fmaxf(fabsf(g4.x), fabsf(g4.y)), fmaxf(fabsf(g4.z), fabsf(g4.w))));
        if (j < nst) reinterpret_cast<float4*>(stage + j * kRow)[lane & 3] = g4;
      }
      // ---- the next chunk's loads
      load_taps(t0 + cap, rcA, dpA);
      load_rows(t0 + cap, gA);
      gm = wave_max(gm);
      if (lane == 0) cmax[wv] = gm;
      __syncthreads();
      if (D > 0) {
        const int ec = exp_above(fmaxf(fmaxf(cmax[0], cmax[1]), fmaxf(cmax[2], cmax[3])));
        if (ec > eg) {                               // uniform; the first chunk, rarely a later one
          const int en = ec + 3;
          if (eg > -1000) {
            const int sh = min(en - eg, 63);
            for (int e = tid; e < DS; e += NT) gtile[e] >>= sh;
            __syncthreads();
          }
          eg = en;
        }
      }
#ifdef VAMP_LIFT_STAMPS
      long long sb = STAMP(); st_stage += sb - sa;
#endif
      // ---- consume: lane = (pixel, slot), the pair's 16 channels in the lane
      const int sce = kFix - eg - efl;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int lo = max(lo_h[h], t0), hi = min(hi_h[h], t0 + nst), mid = mid_h[h];
        for (int r = lo + sl; __any(r < hi); r += kSlots) {
          const bool in = r < hi;
          const float4* sr = reinterpret_cast<const float4*>(stage + (in ? r - t0 : 0) * kRow);
          const float4 ax = sr[r < mid ? 4 : 5];
          const float pwj = in ? ax.x : 0.f;
          float dot = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 g = sr[k];
            acc[4 * k] = __builtin_fmaf(pwj, g.x, acc[4 * k]);
            acc[4 * k + 1] = __builtin_fmaf(pwj, g.y, acc[4 * k + 1]);
            acc[4 * k + 2] = __builtin_fmaf(pwj, g.z, acc[4 * k + 2]);
            acc[4 * k + 3] = __builtin_fmaf(pwj, g.w, acc[4 * k + 3]);
            dot = __builtin_fmaf(ft[4 * k], g.x, dot);
            dot = __builtin_fmaf(ft[4 * k + 1], g.y, dot);
            dot = __builtin_fmaf(ft[4 * k + 2], g.z, dot);
            dot = __builtin_fmaf(ft[4 * k + 3], g.w, dot);
          }
          if (D > 0) {
            const float w0 = in ? ax.y * dot : 0.f, w1 = in ? ax.z * dot : 0.f;
            unsigned long long* gp = reinterpret_cast<unsigned long long*>(gtile + __float_as_int(ax.w) * kS + p);
            if (w0 != 0.f) atomicAdd(gp, fixed_of(w0, sce));
            if (w1 != 0.f) atomicAdd(gp + kS, fixed_of(w1, sce));
          }
        }
      }
#ifdef VAMP_LIFT_STAMPS
      __syncthreads();
      st_cons += STAMP() - sb;
#endif
    }
    // grad_feat: the four slots of a wave by DPP, the four waves through LDS
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = quad_sum(acc[c]);
    __syncthreads();
    if ((lane & 3) == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        reinterpret_cast<float4*>(stage + (wv * kS + p) * 16)[k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
    }
    __syncthreads();
#ifdef VAMP_LIFT_STAMPS
    st2 = STAMP();
#endif
    {
      // consecutive threads = consecutive pixels of one channel (or, channel-last, consecutive channels of a pixel)
      const int cc = fcl ? tid % 16 : tid / kS, pp = fcl ? tid / 16 : tid % kS;
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < kW; ++w2) v += stage[(w2 * kS + pp) * 16 + cc];
      if (pp < np && c0 + cc < C) gfeat[fcl ? (bn * HW + pix0 + pp) * C + c0 + cc : (bn * C + c0 + cc) * HW + pix0 + pp] = v;
    }
  }
  if (D > 0 && gdepth) {
    __syncthreads();
    // fixed point -> float, in place (the floats of a pass land below everything still unread)
    float* gf = reinterpret_cast<float*>(gtile);
    for (int e0 = 0; e0 < DS; e0 += 6 * NT) {
      float v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int e = e0 + k * NT + tid;
        v[k] = e < DS ? (float) ldexp((double) gtile[e], -(kFix - eg - efp[e % kS])) : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int e = e0 + k * NT + tid;
        if (e < DS) gf[e] = v[k];
      }
      __syncthreads();
    }
    if (softmax_bwd) {
      // VAMP_LIFTBWD_LOGITS: the depth column is softmax(logits) (base_vampire2.py:550) and the caller
      // wants the gradient of the logits, p * (g - sum_d p g): the gradient tile sits in LDS and the strip's
      // depth columns join it here (the stage region is free by now), so the softmax backward costs one
      // small reduction per strip and no pass of its own over HBM
      // the depth columns: 16-byte pieces of the planes' 64-byte runs (fW % 4 == 0), else single values
      if (VEC) {
        for (int e0 = 0; e0 < DS / 4; e0 += 2 * NT) {
          float4 v[2];
    #pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int e = min(e0 + k * NT + tid, DS / 4 - 1);
            const int dz = e / (kS / 4), p4 = min((e % (kS / 4)) * 4, np - 4);
            v[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(depth) + (bn * P.D + dz) * HW + pix0 + p4);
          }
    #pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int e = e0 + k * NT + tid;
            if (e < DS / 4) reinterpret_cast<float4*>(dtile)[e] = v[k];
          }
        }
      } else {
        for (int e0 = 0; e0 < DS; e0 += 6 * NT) {
          float v[6];
    #pragma unroll
          for (int k = 0; k < 6; ++k) {
            const int ec = min(e0 + k * NT + tid, DS - 1);
            v[k] = ldf(depth, (bn * P.D + ec / kS) * HW + pix0 + min(ec % kS, np - 1));
          }
    #pragma unroll
          for (int k = 0; k < 6; ++k) {
            const int e = e0 + k * NT + tid;
            if (e < DS) dtile[e] = v[k];
          }
        }
      }
      __syncthreads();
      const int pp = tid % kS, gq = tid / kS;        // 16 partial sums per pixel
      float s = 0.f;
      for (int dz = gq; dz < D; dz += NT / kS) s = __builtin_fmaf(dtile[dz * kS + pp], gf[dz * kS + pp], s);
      red[gq * kS + pp] = s;
      __syncthreads();
      if (tid < kS) {
        float t = 0.f;
        for (int k = 0; k < NT / kS; ++k) t += red[k * kS + tid];
        red[16 * kS + tid] = t;
      }
      __syncthreads();
    }
    if (VEC) {
      for (int e = tid; e < DS / 4; e += NT) {
        const int dz = e / (kS / 4), p4 = (e % (kS / 4)) * 4;
        if (p4 >= np) continue;
        float4 v = reinterpret_cast<const float4*>(gf)[e];
        if (softmax_bwd) {
          const float4 d = reinterpret_cast<const float4*>(dtile)[e];
          const float4 sd = *reinterpret_cast<const float4*>(red + 16 * kS + p4);
          v = make_float4(d.x * (v.x - sd.x), d.y * (v.y - sd.y), d.z * (v.z - sd.z), d.w * (v.w - sd.w));
        }
        *reinterpret_cast<float4*>(gdepth + (bn * P.D + dz) * HW + pix0 + p4) = v;
      }
    } else {
      // consecutive threads = consecutive pixels of one depth plane
      for (int e = tid; e < DS; e += NT) {
        const int dz = e / kS, pp = e % kS;
        if (pp >= np) continue;
        float v = gf[e];
        if (softmax_bwd) v = dtile[e] * (v - red[16 * kS + pp]);
        gdepth[(bn * P.D + dz) * HW + pix0 + pp] = v;
      }
    }
  }
#ifdef VAMP_LIFT_STAMPS
  __syncthreads();
  if (tid == 0 && blockIdx.x < 16384) {
    long long* o = g_stamps + (long) blockIdx.x * 8;
    o[0] = st0; o[1] = st1; o[2] = st_stage; o[3] = st_cons; o[4] = st2; o[5] = STAMP(); o[6] = rt0; o[7] = wall_clock64();
  }
#endif
}

#ifdef VAMP_LIFT_STAMPS
}  // namespace vamp
extern "C" int vamp_debug_read_stamps(long long* host, int n) {
  return (int) hipMemcpyFromSymbol(host, HIP_SYMBOL(vamp::g_stamps), (size_t) n * 8 * sizeof(long long));
}
namespace vamp {
#endif

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
size_t lift_bwd_cell_ws_bytes(const VampLiftDesc* d) { return lift_cell_ws(d, nullptr).bytes; }

// zero the counters in front of a kernel that emits pairs / scan them behind it
int launch_lift_cells_begin(const VampLiftDesc* d, void* scratch, hipStream_t s, bool clean) {
  const LiftCells g = lift_cells(d);
  const LiftCellWs w = lift_cell_ws(d, scratch);
  const size_t cap = (size_t) d->B * d->N * d->Z * d->Y * d->X;
  VAMP_REQUIRE(cap < 0x7fffffffu && g.ncell < 0x7fffffffL, "pair / cell count exceeds 2^31");
  VAMP_REQUIRE(d->C % 4 == 0, "C must be a multiple of 4");
  VAMP_REQUIRE(d->fW < 32767 && d->fH < 32767 && d->D < 65535, "feature map too large for the packed cell coordinates");
  if (clean)                        // (VAMP_LIFTFWD_CELLS_CLEAN: the caller vouches for zeroed counters)
    return debug_expect_range(w.cnt, (size_t) (g.ncell + kScanPad), 0, 0, s, "VAMP_LIFTFWD_CELLS_CLEAN: the lift workspace's cell counters are zero");
  return launch_zero(w.cnt, (size_t) (g.ncell + kScanPad) * sizeof(int), s);
}

int launch_lift_cells_end(const VampLiftDesc* d, void* scratch, hipStream_t s) {
  const LiftCells g = lift_cells(d);
  const LiftCellWs w = lift_cell_ws(d, scratch);
  return launch_cell_scan(w.cnt, w.off, w.bsum, w.boff, w.aux, g.ncell, s);
}
int lift_cells_scan_job(const VampLiftDesc* d, void* scratch, ScanJob* job) {
  const LiftCells g = lift_cells(d);
  const LiftCellWs w = lift_cell_ws(d, scratch);
  return make_scan_job(w.cnt, w.off, w.bsum, w.boff, w.aux, g.ncell, job);
}

template <typename T>
static int launch_cell_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                         const float* xs, const float* ys, const float* zs, const void* depth,
                         const void* feat, const float* gout, const uint64_t* hits, float* gdepth,
                         float* gfeat, void* scratch, bool cells_valid, int variant, int half, bool softmax_bwd,
                         bool fcl, hipStream_t s) {
  const LiftCells g = lift_cells(d);
  const LiftCellWs w = lift_cell_ws(d, scratch);
  if (!cells_valid)
    if (int e = launch_lift_cell_prepare(d, mats, xs, ys, zs, depth, scratch, s)) return e;
  // (the counters are the fill cursors: the scan left them at zero)
  // half 0: all images; 1 / 2: the lower / upper half of the flattened (sample, camera) index
  const int BN = d->B * d->N;
  const int bn_lo = half == 2 ? BN / 2 : 0, bn_hi = half == 1 ? BN / 2 : BN;
  if (bn_hi <= bn_lo) return VAMP_OK;
  dim3 grid((d->X + 63) / 64, (d->Y + 3) / 4, d->Z * d->B);
  const size_t fill_lds = (size_t) 4 * 64 * (2 + d->C / 4) * sizeof(float4);      // [wave][lane][record piece]
#define VAMP_CELL(CH)                                                                            \
  VAMP_TIMED(kProfLiftBwdFill, s, (lift_bwd_fill_kernel<CH><<<grid, 256, fill_lds, s>>>(         \
      P, g.cw, g.ch, gout, hits, w.amask, w.ptaps, w.pcell, w.cnt, w.off, w.boff, w.recs, w.rowq, bn_lo, bn_hi)))
  if (P.C == 4) VAMP_CELL(4); else if (P.C == 8) VAMP_CELL(8); else VAMP_CELL(16);
#undef VAMP_CELL
  if (int e = check_launch("lift_bwd_fill_kernel")) return e;

  // pairs staged per chunk: 128 by default; the variants exist so that the tests cross chunk
  // boundaries at every size
  const int cap = variant == 4 ? 64 : (variant == 16 ? 32 : 128);
  const int Dd = d->use_depth ? d->D : 0;
  const size_t lds = strip_lds_floats(Dd, cap) * sizeof(float);
  if (lds > 150 * 1024) return fail(VAMP_EINVAL, "%s: D too large for the LDS depth tiles", __func__);
  const int spr = (d->fW + kS - 1) / kS;
  const unsigned ggrid = (unsigned) (bn_hi - bn_lo) * d->fH * spr;
  {
    // 16-byte tile I/O needs fp32 depth planes whose strips start on 16-byte boundaries
    const bool vec = std::is_same<T, float>::value && d->fW % 4 == 0;
    auto k = vec ? lift_bwd_strip_kernel<T, true> : lift_bwd_strip_kernel<T, false>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess)
      return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);
    VAMP_TIMED(kProfLiftBwd, s, (k<<<ggrid, kW * 64, lds, s>>>(
        P, g.cw, g.ch, spr, spr, bn_lo, cap, static_cast<const T*>(depth), static_cast<const T*>(feat),
        w.off, w.boff, w.recs, w.rowq, w.cnt, gdepth, gfeat, softmax_bwd ? 1 : 0, fcl ? 1 : 0)));
  }
  return check_launch("lift_bwd_strip_kernel");
}

int launch_lift_bwd_cell(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, const float* gout,
                         const uint64_t* hits, float* gdepth, float* gfeat, void* scratch,
                         bool cells_valid, int variant, int half, bool softmax_bwd, bool fcl, hipStream_t s) {
  const LiftParams P = to_params(d);
  if (d->in_dtype == VAMP_F32)
    return launch_cell_t<float>(d, P, mats, xs, ys, zs, depth, feat, gout, hits, gdepth, gfeat, scratch,
                                cells_valid, variant, half, softmax_bwd, fcl, s);
  return launch_cell_t<__hip_bfloat16>(d, P, mats, xs, ys, zs, depth, feat, gout, hits, gdepth, gfeat,
                                       scratch, cells_valid, variant, half, softmax_bwd, fcl, s);
}

}  // namespace vamp
