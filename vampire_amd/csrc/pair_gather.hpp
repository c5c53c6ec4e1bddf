// x-neighbour pair gathers from channel-first volumes through buffer descriptors: shared by the
// one-kernel camera forward (render_cam_direct.hip) and the backward's per-ray pass
// (render_bwd_ray.hip).  A trilinear sample of one channel is four 8-byte loads -- the x pair at
// (z, y), (z, y+1), (z+1, y), (z+1, y+1) -- whose address is descriptor base + per-lane byte offset
// (VGPR) + per-channel byte offset (SGPR).
#pragma once
#include "render_common.hpp"

namespace vamp {

// x-neighbour pair through a buffer descriptor: the address is descriptor base + per-lane byte offset
// (VGPR) + per-channel byte offset (SGPR) -- no vector address arithmetic per channel, and hipcc issues
// a whole batch of such loads back to back (with 64-bit global addresses it waited after every
// second channel: eleven round trips per sample).  fp32: one 8-byte load at a 4-byte-aligned
// address; bf16: two 2-byte loads.
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
struct PairRaw {
  unsigned x, y;       // fp32: the two floats' bits; bf16: the two elements' 16 bits, zero-extended
};
template <typename T>
__device__ __forceinline__ PairRaw ld_pair(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  PairRaw r;
  if constexpr (sizeof(T) == 4) {
    const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    r.x = v.x; r.y = v.y;
  } else {
    r.x = __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0);
    r.y = __builtin_amdgcn_raw_buffer_load_b16(rs, voff + 2u, soff, 0);
  }
  return r;
}
template <typename T>
__device__ __forceinline__ float pair_lo(const PairRaw& p) { return __uint_as_float(sizeof(T) == 4 ? p.x : p.x << 16); }
template <typename T>
__device__ __forceinline__ float pair_hi(const PairRaw& p) { return __uint_as_float(sizeof(T) == 4 ? p.y : p.y << 16); }

// descriptor over `bytes` bytes at p (wave-uniform values only: the block's batch index)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int) (bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000);
}

// the four row offsets and six weights of an INSIDE sample's 2 x 2 x 2 taps, x taken as a pair
struct PairTap {
  unsigned o00, o01, o10, o11;      // (z0,y0) (z0,y1) (z1,y0) (z1,y1): BYTE offset of the pair in a channel
  float w00, w01, w10, w11;         // wz * wy
  float wa, wb;                     // weights of the pair's two elements
};

template <typename T>
__device__ __forceinline__ PairTap pair_tap(const RenderParams& P, const VolTap& tp) {
  // inside => 0 <= ix0 <= X - 1 etc.; a "+1" tap beyond the volume has weight exactly zero
  // (the coordinate is the last index itself): clamp its address, zero its weight
  PairTap t;
  const int xa = min(tp.ix0, P.X - 2);
  const bool last = tp.ix0 != xa;                   // ix0 == X - 1: the pair is (X - 2, X - 1)
  t.wa = last ? 0.f : tp.wx0;
  t.wb = last ? tp.wx0 : tp.wx1;
  const int y1 = min(tp.iy0 + 1, P.Y - 1), z1 = min(tp.iz0 + 1, P.Z - 1);
  const float wy1 = (tp.iy0 + 1 < P.Y) ? tp.wy1 : 0.f, wz1 = (tp.iz0 + 1 < P.Z) ? tp.wz1 : 0.f;
  const unsigned r0 = (unsigned) (tp.iz0 * P.Y), r1 = (unsigned) (z1 * P.Y);
  constexpr unsigned es = sizeof(T);
  t.o00 = ((r0 + tp.iy0) * P.X + xa) * es; t.o01 = ((r0 + y1) * P.X + xa) * es;
  t.o10 = ((r1 + tp.iy0) * P.X + xa) * es; t.o11 = ((r1 + y1) * P.X + xa) * es;
  t.w00 = tp.wz0 * tp.wy0; t.w01 = tp.wz0 * wy1; t.w10 = wz1 * tp.wy0; t.w11 = wz1 * wy1;
  return t;
}

template <typename T>
__device__ __forceinline__ float pair_combine(const PairTap& t, const PairRaw (&v)[4]) {
  const float r0 = __builtin_fmaf(t.wb, pair_hi<T>(v[0]), t.wa * pair_lo<T>(v[0]));
  const float r1 = __builtin_fmaf(t.wb, pair_hi<T>(v[1]), t.wa * pair_lo<T>(v[1]));
  const float r2 = __builtin_fmaf(t.wb, pair_hi<T>(v[2]), t.wa * pair_lo<T>(v[2]));
  const float r3 = __builtin_fmaf(t.wb, pair_hi<T>(v[3]), t.wa * pair_lo<T>(v[3]));
  return __builtin_fmaf(t.w11, r3, __builtin_fmaf(t.w10, r2, __builtin_fmaf(t.w01, r1, t.w00 * r0)));
}

}  // namespace vamp
