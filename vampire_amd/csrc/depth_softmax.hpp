// Depth softmax tile (`mapping_along_depth(src).softmax(dim=1)`, base_vampire2.py:550), shared by the
// standalone kernel (glue.hip) and the lift's operand producer (lift.hip).  A workgroup of 256
// threads owns 64 neighbouring pixels of one image; its four waves split the depth bins, so every
// load and store is a coalesced 256-byte row.
#pragma once
#include "common.hpp"

namespace vamp {

constexpr int kPix = 64;     // pixels per workgroup (= lanes of a wave)
constexpr int kSplit = 4;    // waves per workgroup = chunks of the depth axis
constexpr int kRegBins = 32; // depth bins a lane keeps in registers (D <= kSplit * kRegBins)

struct SoftmaxLds {
  float m[kSplit][kPix], s[kSplit][kPix];
};

// REG: the wave's chunk of the depth axis lives in registers, so the logits are read once and all
// loads of a lane are in flight together; otherwise (D > 128) the chunk is streamed twice.
template <typename T, bool REG>
__device__ __forceinline__ void depth_softmax_tile(const T* __restrict__ logits, float* __restrict__ out,
                                                   int D, long HW, long img, long tile, SoftmaxLds& L) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long pix = tile * kPix + lane;
  const bool live = pix < HW;
  const int Ld = (D + kSplit - 1) / kSplit;
  const int d0 = wv * Ld, d1 = min(D, d0 + Ld);
  const long base = img * D * HW + (live ? pix : HW - 1);
  // local (max, sum of exp) of this wave's depth chunk
  float m = -INFINITY, s = 0.f;
  float x[REG ? kRegBins : 1];
  if (REG) {
#pragma unroll
    for (int k = 0; k < kRegBins; ++k)
      x[k] = (d0 + k < d1) ? ldf(logits, base + (long) (d0 + k) * HW) : -INFINITY;
#pragma unroll
    for (int k = 0; k < kRegBins; ++k) m = fmaxf(m, x[k]);
    if (m > -INFINITY) {
#pragma unroll
      for (int k = 0; k < kRegBins; ++k) s += expf(x[k] - m);
    }
  } else {
    for (int d = d0; d < d1; ++d) {
      const float v = ldf(logits, base + (long) d * HW);
      const float mn = fmaxf(m, v);
      if (mn > -INFINITY) s = s * expf(m - mn) + expf(v - mn);
      m = mn;
    }
  }
  L.m[wv][lane] = m;
  L.s[wv][lane] = s;
  __syncthreads();
  float M = L.m[0][lane];
#pragma unroll
  for (int k = 1; k < kSplit; ++k) M = fmaxf(M, L.m[k][lane]);
  float S = 0.f;
#pragma unroll
  for (int k = 0; k < kSplit; ++k) {
    const float mk = L.m[k][lane];
    S += (mk == -INFINITY) ? 0.f : L.s[k][lane] * expf(mk - M);
  }
  if (!live) return;
  if (REG) {
#pragma unroll
    for (int k = 0; k < kRegBins; ++k)
      if (d0 + k < d1) out[base + (long) (d0 + k) * HW] = expf(x[k] - M) / S;
  } else {
    for (int d = d0; d < d1; ++d) {
      const float v = ldf(logits, base + (long) d * HW);
      out[base + (long) d * HW] = expf(v - M) / S;
    }
  }
}

}  // namespace vamp
