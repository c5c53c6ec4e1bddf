// Device side of the one-kernel BEV forward (render_bev_fused.hip holds the description and the launcher;
// render_fwd_merged.hip runs the same column-block function behind the camera tiles in one launch).
#pragma once
#include "render_common.hpp"

namespace vamp {

typedef unsigned bev_v2u32 __attribute__((ext_vector_type(2)));

struct BevAxis {
  int i0;
  float w0, w1;
};
__device__ __forceinline__ BevAxis bev_axis(float pos, float lo, float span, int n) {
  const float g = ((pos - lo) / span) * 2.0f - 1.0f;
  const float f = ((g + 1.0f) / 2.0f) * (float) (n - 1);
  const float fl = floorf(f);
  BevAxis t;
  t.i0 = (int) fl;
  t.w1 = f - fl;
  t.w0 = (fl + 1.0f) - f;
  return t;
}

template <typename T>
__device__ __forceinline__ void bev_ld_pair(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float& a, float& b) {
  if constexpr (sizeof(T) == 4) {
    const bev_v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    a = __uint_as_float(v.x); b = __uint_as_float(v.y);
  } else {
    a = __uint_as_float(((unsigned) __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0)) << 16);
    b = __uint_as_float(((unsigned) __builtin_amdgcn_raw_buffer_load_b16(rs, voff + 2u, soff, 0)) << 16);
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bev_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int) (bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000);
}

// a column's (y, x) taps: byte offsets of the x-pair in rows y0 / y1 of a plane, and the weights
// (zero padding: a tap outside the volume keeps a legal address and gets weight zero)
struct ColTap {
  unsigned o0, o1;
  float wy0, wy1, wa, wb;
};
template <typename T>
__device__ __forceinline__ ColTap col_tap(const RenderParams& P, const BevAxis& tx, const BevAxis& ty) {
  ColTap c;
  const int xa = min(max(tx.i0, 0), P.X - 2);                 // the pair is (xa, xa + 1)
  const float w0 = (tx.i0 >= 0 && tx.i0 < P.X) ? tx.w0 : 0.f;
  const float w1 = (tx.i0 + 1 >= 0 && tx.i0 + 1 < P.X) ? tx.w1 : 0.f;
  c.wa = (tx.i0 == xa ? w0 : 0.f) + (tx.i0 + 1 == xa ? w1 : 0.f);
  c.wb = (tx.i0 == xa + 1 ? w0 : 0.f) + (tx.i0 + 1 == xa + 1 ? w1 : 0.f);
  const int y0 = min(max(ty.i0, 0), P.Y - 1), y1 = min(max(ty.i0 + 1, 0), P.Y - 1);
  c.wy0 = (ty.i0 >= 0 && ty.i0 < P.Y) ? ty.w0 : 0.f;
  c.wy1 = (ty.i0 + 1 >= 0 && ty.i0 + 1 < P.Y) ? ty.w1 : 0.f;
  c.o0 = (unsigned) (y0 * P.X + xa) * (unsigned) sizeof(T);
  c.o1 = (unsigned) (y1 * P.X + xa) * (unsigned) sizeof(T);
  return c;
}

constexpr int kFusedMaxOZ = 64;       // heights whose taps / weights fit the LDS tables
#ifndef VAMP_BEVF_PC
#define VAMP_BEVF_PC 6
#endif
constexpr int kFusedPC = VAMP_BEVF_PC; // volume planes fetched together (2 x-pair loads each)
constexpr int kFusedMaxNP = 40;       // distinct volume planes the det heights may touch

#ifndef VAMP_BEVF_PARTS
#define VAMP_BEVF_PARTS 2            // channel groups per column block (workgroups of NWV waves each)
#endif
#ifndef VAMP_BEVF_XCD
#define VAMP_BEVF_XCD 1
#endif

#ifdef VAMP_BEVF_STAMPS
// diagnostic build only (tools/debug/bev_stamps.py): phase stamps of wave 0 of both channel groups of every column block
static __device__ long long g_bevf_stamps[1024 * 16];
#define VAMP_BSTAMP(k)                                                                                  \
  do {                                                                                                  \
    if (lane == 0 && wave == 0 && bx < 1024 && part < 2)                                                \
      g_bevf_stamps[bx * 16 + (part ? 8 : 0) + (k)] = (long long) wall_clock64();         \
  } while (0)
#else
#define VAMP_BSTAMP(k) do { } while (0)
#endif
__device__ __forceinline__ void bev_store(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 0);
}

// dynamic LDS: sigma [oZ][64] | weights [oZ][64] | density planes [NPA + 1][64] (shared) | per wave: plane values [NPA][64]
//
// Round 4.  Phase stamps (tools/debug/bev_stamps.py) showed where a workgroup's 26 us went: 1 us tables, 2 us
// density planes, 6-8 us for ten heights' sigma and weights -- every one of the 8 waves evaluated all of them
// (density_fwd + two expf per height, 6 waves per SIMD doing the same arithmetic) -- and 15 us channels, each
// wave's 4-5 channels one after the other and two load round trips deep; and 625 workgroups of 8 waves are one
// uneven round on 256 CUs (2 or 3 per CU: the span was 36 us where the median workgroup took 26).  Now
//   * the heights are dealt to the waves (sigma, then the weights from a prefix of the others' sigma: two
//     short barriers instead of 8x the arithmetic),
//   * the channel loop is software-pipelined over (channel, chunk of planes): the loads of the next chunk are
//     in flight while the current one is staged, sampled and stored, and the first chunk's loads are issued
//     before the density phase,
//   * a workgroup is NWV waves and one of two channel groups of a column block: the density + the composited
//     channels, or the pass-through channels -- which need no weights, skip the density phase and stream while
//     the other group's workgroups on the same CU are in theirs (blockIdx.z does not enter the XCD of a
//     workgroup).
// (bx, b, part, nparts): blockIdx.x / .y / .z and gridDim.z of bev_fwd_fused_kernel; the merged render forward of
// render_fwd_merged.hip decodes them from its one-dimensional grid.
template <typename T, int NWV>
__device__ __forceinline__ void
bev_fwd_fused_block(const unsigned bx, const int b, const int part, const int nparts, const RenderParams& P, int NPA,
                    const float* __restrict__ oxs, const float* __restrict__ oys,
                     const float* __restrict__ ozs, const float* __restrict__ bev_mids,
                     const float* __restrict__ beta_raw, const T* __restrict__ dens,
                     const T* __restrict__ sem, const T* __restrict__ rgb, const T* __restrict__ base,
                     float* __restrict__ bev_rgb, float* __restrict__ bev_seg, float* __restrict__ bev_height,
                     float* __restrict__ voxel_density, float* __restrict__ voxel_output,
                     float* __restrict__ s0_save, float* __restrict__ ss_save) {
  __shared__ int tz_i0[kFusedMaxOZ];
  __shared__ float tz_w0[kFusedMaxOZ], tz_w1[kFusedMaxOZ];
  extern __shared__ __align__(16) float sig[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  VAMP_BSTAMP(0);
  float* wgt = sig + P.oZ * 64;
  float* dsl = wgt + P.oZ * 64;                                   // the density planes' bilinear values (all waves)
  float* bl = dsl + (NPA + 1) * 64 + wave * NPA * 64;                   // this wave's plane values (NPA planes allocated)
  const int OYX = P.oY * P.oX;
  // XCD k (workgroups with blockIdx.x % 8 == k) walks a contiguous band of the lattice: y-neighbours,
  // which read the same volume rows, share that XCD's L2 (the grid is rounded up to a multiple of 8)
  const int nwg = (OYX + 63) / 64, per_xcd = (nwg + 7) / 8;
  const int wg = VAMP_BEVF_XCD ? (int) (bx & 7) * per_xcd + (int) (bx >> 3) : (int) bx;
  if (wg >= nwg || (VAMP_BEVF_XCD && (int) (bx >> 3) >= per_xcd)) return;
  const int col_raw = wg * 64 + lane;
  const bool live = col_raw < OYX;
  const int col = live ? col_raw : OYX - 1;
  const int y = col / P.oX, x = col - y * P.oX;
  if ((int) threadIdx.x < P.oZ) {
    const BevAxis tz = bev_axis(ozs[P.oZ - 1 - threadIdx.x], P.lo[2], P.span[2], P.Z);   // flip (bv2:443)
    tz_i0[threadIdx.x] = tz.i0; tz_w0[threadIdx.x] = tz.w0; tz_w1[threadIdx.x] = tz.w1;
  }
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const unsigned V = (unsigned) (P.Z * P.Y * P.X);
  const unsigned plane_b = (unsigned) (P.Y * P.X) * (unsigned) sizeof(T);
  const unsigned vol_b = V * (unsigned) sizeof(T);
  const unsigned oplane_b = (unsigned) OYX * 4u, ovol_b = oplane_b * (unsigned) P.oZ;   // one height / one channel of the outputs
  const ColTap ct = col_tap<T>(P, bev_axis(oxs[x], P.lo[0], P.span[0], P.X), bev_axis(oys[y], P.lo[1], P.span[1], P.Y));
  // a lane outside the lattice stores beyond every descriptor's range: the hardware drops it
  const unsigned ocol = live ? (unsigned) col * 4u : 0x7ffffff0u;
  __syncthreads();
  VAMP_BSTAMP(1);
  // the volume planes the heights touch: [pmin, pmin + NP) (uniform; the launcher sized the slabs for a
  // lattice of heights with the descriptor's spacing, NPA planes)
  int pmin = tz_i0[0], pmax = tz_i0[0] + 1;
  for (int j = 1; j < P.oZ; ++j) { pmin = min(pmin, tz_i0[j]); pmax = max(pmax, tz_i0[j] + 1); }
  pmin = __builtin_amdgcn_readfirstlane(pmin);
  const int NP = min(__builtin_amdgcn_readfirstlane(pmax) - pmin + 1, NPA);
  // bilinear (y, x) value of plane p from its two x-pairs; zero padding for a plane outside the volume.
  // (The test on p is uniform, and as a uniform branch around the use of the loaded registers it would make
  // the waits in front of it conditional -- after which the compiler drains ALL loads before it reuses those
  // registers.  `vz` is a zero the compiler takes for a per-lane value: the padding becomes a select.)
  int vz = 0;
  asm volatile("" : "+v"(vz));
  auto bilin = [&](int p, float a0, float b0, float a1, float b1) -> float {
    const float r0 = __builtin_fmaf(ct.wb, b0, ct.wa * a0), r1 = __builtin_fmaf(ct.wb, b1, ct.wa * a1);
    return (unsigned) (p + vz) < (unsigned) P.Z ? __builtin_fmaf(ct.wy1, r1, ct.wy0 * r0) : 0.f;
  };
  // trilinear sample at height j from a slab (aten: z-interpolation of the two bilinear plane values)
  auto sample = [&](const float* slab, int j) -> float {
    const int k = min(tz_i0[j] - pmin, NP - 2);                    // 0 <= k, k + 1 < NP
    return __builtin_fmaf(tz_w1[j], slab[(k + 1) * 64 + lane], tz_w0[j] * slab[k * 64 + lane]);
  };

  // ---- channels: [0, K) semantic, [K, K + 3) rgb, [K + 3, K + 3 + C) base; wave g of channel group `part`
  // owns channels (g * nparts + part) + NWV * nparts * i
  const int nch = P.K + 3 + P.C;
  const int CO = P.C + (P.cat_seg ? P.K : 0);
  const __amdgpu_buffer_rsrc_t rs_s = bev_rsrc(sem + (long) b * P.K * V, (size_t) P.K * vol_b);
  const __amdgpu_buffer_rsrc_t rs_r = bev_rsrc(rgb + (long) b * 3 * V, (size_t) 3 * vol_b);
  const __amdgpu_buffer_rsrc_t rs_b = bev_rsrc(base + (long) b * P.C * V, (size_t) P.C * vol_b);
  const __amdgpu_buffer_rsrc_t rs_vo = bev_rsrc(voxel_output + (long) b * CO * P.oZ * OYX, (size_t) CO * ovol_b);
  const __amdgpu_buffer_rsrc_t rs_ss = bev_rsrc(ss_save ? ss_save + (long) b * (P.K + 3) * P.oZ * OYX : voxel_output,
                                                ss_save ? (size_t) (P.K + 3) * ovol_b : 0);
  constexpr int PC = kFusedPC;
  struct Chunk {
    float a0[PC], b0[PC], a1[PC], b1[PC];
  };
  const int nck = (NP + PC - 1) / PC;                               // chunks of planes per channel
  // Channel groups (nparts >= 2): group 0 the density and the composited channels [0, K + 3), groups 1 .. nparts - 1
  // equal slices of the pass-through channels [K + 3, nch) -- which need neither sigma nor weights: those workgroups
  // skip the density phase and stream from their first instruction, while the group-0 workgroups beside them on the
  // CU are in theirs (one round of workgroups in lockstep: for the first 9 us of the kernel almost nothing moved).
  // More than one pass-through group makes their workgroups SHORT: behind the camera tiles of the merged launch
  // (render_fwd_merged.hip) they are what fills the last microseconds.
  const bool split = nparts >= 2;
  const int per = split ? (P.C + nparts - 2) / (nparts - 1) : 0;    // pass-through channels per group
  const int c_lo = split && part >= 1 ? P.K + 3 + (part - 1) * per : 0;
  const int c_hi = split ? (part == 0 ? P.K + 3 : min(nch, c_lo + per)) : nch;
  const bool with_density = !(split && part >= 1);
  const int cstride = split ? NWV : NWV * nparts;
  // the x-pair loads of chunk k of channel c (2 * PC loads, none waited for here).  Past the last channel
  // the loads go through a zero-size descriptor (they return at once): the loop below is branch-free around
  // its loads, so that the wait in front of a chunk's use counts the younger chunk's loads as outstanding.
  auto issue = [&](Chunk& ch, int c, int k) {
    const bool is_s = c < P.K, is_r = c < P.K + 3, past = c >= c_hi;
    const __amdgpu_buffer_rsrc_t rs = is_s ? rs_s : (is_r ? rs_r : (past ? bev_rsrc(base, 0) : rs_b));
    const unsigned cso = (unsigned) (is_s ? c : (is_r ? c - P.K : (past ? 0 : c - P.K - 3))) * vol_b;
#pragma unroll
    for (int t = 0; t < PC; ++t) {
      const int p = pmin + min(k * PC + t, NP - 1);
      const unsigned so = cso + (unsigned) min(max(p, 0), P.Z - 1) * plane_b;
      bev_ld_pair<T>(rs, ct.o0, so, ch.a0[t], ch.b0[t]);
      bev_ld_pair<T>(rs, ct.o1, so, ch.a1[t], ch.b1[t]);
    }
  };
  int ci = split ? c_lo + wave : wave * nparts + part, ki = 0;       // next chunk to issue
  int cc = ci, kc = 0;                                              // next chunk to consume
  auto step = [&](int& c, int& k) {
    if (++k == nck) { k = 0; c += cstride; }
  };
  Chunk A, B;
  issue(A, ci, ki); step(ci, ki);                                   // (in flight across the density phase)

  if (with_density) {
    // ---- density planes: wave g fetches planes g, g + NWV, ... (all in flight together) into the shared slab
    {
      constexpr int DP = (kFusedMaxNP + NWV - 1) / NWV;
      const T* dbase = dens + (long) b * V;
      float a0[DP], b0[DP], a1[DP], b1[DP];
      // (branch-free: a slot past the last plane loads through a zero-size descriptor and lands in the spare
      // plane [NPA] of the slab)
  #pragma unroll
      for (int t = 0; t < DP; ++t) {
        const int q0 = wave + t * NWV;
        const __amdgpu_buffer_rsrc_t rs_d = bev_rsrc(dbase, q0 < NP ? (size_t) vol_b : 0);
        const int p = pmin + min(q0, NP - 1);
        const unsigned so = (unsigned) min(max(p, 0), P.Z - 1) * plane_b;
        bev_ld_pair<T>(rs_d, ct.o0, so, a0[t], b0[t]);
        bev_ld_pair<T>(rs_d, ct.o1, so, a1[t], b1[t]);
      }
  #pragma unroll
      for (int t = 0; t < DP; ++t) {
        const int q0 = wave + t * NWV;
        const int q = q0 < NP ? q0 : NPA;
        dsl[q * 64 + lane] = bilin(pmin + q0, a0[t], b0[t], a1[t], b1[t]);
      }
    }
    __syncthreads();
    VAMP_BSTAMP(2);

    // ---- sigma_j: wave g takes heights g, g + NWV, ...; group 0 also stores voxel_density and the raw sample
    // for the backward's scan (the other groups through zero-size descriptors: dropped)
    {
      const __amdgpu_buffer_rsrc_t rs_vd = bev_rsrc(voxel_density + (long) b * P.oZ * OYX, part == 0 ? (size_t) ovol_b : 0);
      const __amdgpu_buffer_rsrc_t rs_s0 = bev_rsrc(s0_save ? s0_save + (long) b * P.oZ * OYX : voxel_density,
                                                    s0_save && part == 0 ? (size_t) ovol_b : 0);
      for (int j = wave; j < P.oZ; j += NWV) {
        const float s0 = sample(dsl, j);
        const float sigma = density_fwd(dp, s0);
        bev_store(rs_vd, ocol, (unsigned) j * oplane_b, sigma);
        bev_store(rs_s0, ocol, (unsigned) j * oplane_b, s0);
        sig[j * 64 + lane] = sigma;
      }
    }
    __syncthreads();
    // ---- compositing weights of the same heights: w_j = (1 - exp(-tau_j)) exp(-sum_{i<j} tau_i), the sum in
    // the order of the reference's cumsum (bv2:451-453)
    for (int j = wave; j < P.oZ; j += NWV) {
      float cum = 0.f;
      for (int i = 0; i < j; ++i) cum += sig[i * 64 + lane] * (1.0f * P.z_step);
      const float tau = sig[j * 64 + lane] * (1.0f * P.z_step);
      wgt[j * 64 + lane] = (1.0f - expf(-tau)) * expf(-cum);
    }
    __syncthreads();
    if (wave == 0 && part == 0) {
      float height = 0.f;
      for (int j = 0; j < P.oZ; ++j) height = __builtin_fmaf(wgt[j * 64 + lane], bev_mids[j], height);
      if (live) bev_height[(long) b * OYX + col] = height;
    }
  }
  VAMP_BSTAMP(3);

  // chunk k into this wave's slab; after a channel's last chunk: its heights
  auto consume = [&](const Chunk& ch, int c, int k) {
#pragma unroll
    for (int t = 0; t < PC; ++t) {
      const int q = min(k * PC + t, NP - 1), p = pmin + q;          // (branch-free, as above)
      bl[q * 64 + lane] = bilin(p, ch.a0[t], ch.b0[t], ch.a1[t], ch.b1[t]);
    }
    if (k != nck - 1) return;
    if (c < P.K + 3) {
      // composited channel (semantic / rgb)
      const bool cat = c < P.K && P.cat_seg;
      float acc = 0.f;
      for (int j = 0; j < P.oZ; ++j) {
        const float sv = sample(bl, j);
        acc = __builtin_fmaf(wgt[j * 64 + lane], sv, acc);
        // training: the backward's q_j = sum_c G_c s_j[c] reads the samples back (zero-size descriptor otherwise)
        bev_store(rs_ss, ocol, (unsigned) c * ovol_b + (unsigned) j * oplane_b, sv);
        if (cat) bev_store(rs_vo, ocol, (unsigned) (P.C + c) * ovol_b + (unsigned) j * oplane_b, sv);   // bv2:449-450
      }
      if (live) {
        if (c < P.K) bev_seg[((long) b * P.K + c) * OYX + col] = acc;
        else bev_rgb[((long) b * 3 + (c - P.K)) * OYX + col] = acc;
      }
    } else {
      // pass-through channel (base -> voxel_output)
      const int cb = c - P.K - 3;
      for (int j = 0; j < P.oZ; ++j)
        bev_store(rs_vo, ocol, (unsigned) cb * ovol_b + (unsigned) j * oplane_b, sample(bl, j));
    }
  };
  while (cc < c_hi) {
    issue(B, ci, ki); step(ci, ki);
    consume(A, cc, kc); step(cc, kc);
    issue(A, ci, ki); step(ci, ki);
    if (cc < c_hi) consume(B, cc, kc);
    step(cc, kc);
  }
  VAMP_BSTAMP(4);
}

}  // namespace vamp
