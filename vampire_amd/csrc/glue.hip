// Producer / consumer steps either side of the lift + render path, gfx950.
//
//  depth_softmax   `mapping_along_depth(src).softmax(dim=1)`, base_vampire2.py:550: the depth
//                  distribution the lift multiplies with the image features.  A workgroup owns 64
//                  neighbouring pixels; its four waves split the depth bins, so every load and
//                  store is a coalesced 256-byte row and four times more waves are in flight than
//                  with a thread per pixel.  One read of the logits from HBM (the second sweep hits
//                  the cache), one write of the probabilities.
//  density_gate    `voxel_output * bev_density.tanh()` (sdf) / `voxel_output * bev_density`
//                  (naive), base_vampire2.py:627-630, in front of the `voxel_output` 1x1 conv.  The
//                  backward produces both gradients in one pass over (grad, voxel_output).
// Elementwise / short reductions, HBM-bound: no MFMA.
#include "depth_softmax.hpp"

namespace vamp {
namespace {

template <typename T, bool REG>
__global__ void __launch_bounds__(256)
depth_softmax_fwd_kernel(const T* __restrict__ logits, float* __restrict__ out, int D, long HW,
                         int tiles) {
  __shared__ SoftmaxLds L;
  depth_softmax_tile<T, REG>(logits, out, D, HW, blockIdx.x / tiles, blockIdx.x % tiles, L);
}

// grad_logits = p * (g - sum_d p g)
template <bool REG>
__global__ void __launch_bounds__(256)
depth_softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ g,
                         float* __restrict__ gx, int D, long HW, int tiles) {
  __shared__ float sd[kSplit][kPix];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long img = blockIdx.x / tiles;
  const long pix = (long) (blockIdx.x % tiles) * kPix + lane;
  const bool live = pix < HW;
  const int L = (D + kSplit - 1) / kSplit;
  const int d0 = wv * L, d1 = min(D, d0 + L);
  const long base = img * D * HW + (live ? pix : HW - 1);
  float dot = 0.f;
  float pv[REG ? kRegBins : 1], gv[REG ? kRegBins : 1];
  if (REG) {
#pragma unroll
    for (int k = 0; k < kRegBins; ++k) {
      const bool in = d0 + k < d1;
      pv[k] = in ? p[base + (long) (d0 + k) * HW] : 0.f;
      gv[k] = in ? g[base + (long) (d0 + k) * HW] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < kRegBins; ++k) dot = __builtin_fmaf(pv[k], gv[k], dot);
  } else {
    for (int d = d0; d < d1; ++d)
      dot = __builtin_fmaf(p[base + (long) d * HW], g[base + (long) d * HW], dot);
  }
  sd[wv][lane] = dot;
  __syncthreads();
  const float tot = (sd[0][lane] + sd[1][lane]) + (sd[2][lane] + sd[3][lane]);
  if (!live) return;
  if (REG) {
#pragma unroll
    for (int k = 0; k < kRegBins; ++k)
      if (d0 + k < d1) gx[base + (long) (d0 + k) * HW] = pv[k] * (gv[k] - tot);
  } else {
    for (int d = d0; d < d1; ++d) {
      const long o = base + (long) d * HW;
      gx[o] = p[o] * (g[o] - tot);
    }
  }
}

__device__ __forceinline__ float gate_value(int mode, float vd) {
  return mode == VAMP_DENSITY_SDF_LAPLACE ? tanhf(vd) : vd;
}

// thread per BEV cell, channel loop (stride = cells): coalesced rows
__global__ void __launch_bounds__(256)
density_gate_fwd_kernel(const float* __restrict__ vo, const float* __restrict__ vd,
                        float* __restrict__ out, int C, long cells, int mode) {
  const long i = (long) blockIdx.x * 256 + threadIdx.x;
  const long b = blockIdx.y;
  if (i >= cells) return;
  const float gt = gate_value(mode, vd[b * cells + i]);
  const long o = b * C * cells + i;
#pragma unroll 4
  for (int c = 0; c < C; ++c) out[o + (long) c * cells] = vo[o + (long) c * cells] * gt;
}

__global__ void __launch_bounds__(256)
density_gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ vo,
                        const float* __restrict__ vd, float* __restrict__ gvo,
                        float* __restrict__ gvd, int C, long cells, int mode) {
  const long i = (long) blockIdx.x * 256 + threadIdx.x;
  const long b = blockIdx.y;
  if (i >= cells) return;
  const float gt = gate_value(mode, vd[b * cells + i]);
  const long o = b * C * cells + i;
  float dot = 0.f;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const float gc = g[o + (long) c * cells];
    gvo[o + (long) c * cells] = gc * gt;
    dot = __builtin_fmaf(gc, vo[o + (long) c * cells], dot);
  }
  // d tanh(x) = 1 - tanh(x)^2
  gvd[b * cells + i] = mode == VAMP_DENSITY_SDF_LAPLACE ? dot * (1.0f - gt * gt) : dot;
}

}  // namespace
}  // namespace vamp

using namespace vamp;

extern "C" {

int vamp_depth_softmax_forward(int64_t images, int32_t D, int64_t HW, const void* logits,
                               int32_t in_dtype, float* depth, void* stream) {
  VAMP_REQUIRE(images > 0 && D > 0 && HW > 0, "images, D, HW must be positive");
  VAMP_REQUIRE(logits && depth, "NULL tensor");
  VAMP_REQUIRE(in_dtype == VAMP_F32 || in_dtype == VAMP_BF16, "in_dtype");
  const long tiles = (HW + kPix - 1) / kPix;
  VAMP_REQUIRE(images * tiles < 0x7fffffffL, "too many pixel tiles");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned) (images * tiles);
  const bool reg = D <= kSplit * kRegBins;
#define VAMP_SOFTMAX(T, REG)                                                                      \
  VAMP_TIMED(kProfGlueSoftmax, s, (depth_softmax_fwd_kernel<T, REG><<<grid, 256, 0, s>>>(         \
      static_cast<const T*>(logits), depth, D, HW, (int) tiles)))
  if (in_dtype == VAMP_F32) {
    if (reg) VAMP_SOFTMAX(float, true); else VAMP_SOFTMAX(float, false);
  } else {
    if (reg) VAMP_SOFTMAX(__hip_bfloat16, true); else VAMP_SOFTMAX(__hip_bfloat16, false);
  }
#undef VAMP_SOFTMAX
  return check_launch("depth_softmax_fwd_kernel");
}

int vamp_depth_softmax_backward(int64_t images, int32_t D, int64_t HW, const float* depth,
                                const float* grad_depth, float* grad_logits, void* stream) {
  VAMP_REQUIRE(images > 0 && D > 0 && HW > 0, "images, D, HW must be positive");
  VAMP_REQUIRE(depth && grad_depth && grad_logits, "NULL tensor");
  const long tiles = (HW + kPix - 1) / kPix;
  VAMP_REQUIRE(images * tiles < 0x7fffffffL, "too many pixel tiles");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned) (images * tiles);
  if (D <= kSplit * kRegBins) {
    VAMP_TIMED(kProfGlueSoftmax, s, (depth_softmax_bwd_kernel<true><<<grid, 256, 0, s>>>(
        depth, grad_depth, grad_logits, D, HW, (int) tiles)));
  } else {
    VAMP_TIMED(kProfGlueSoftmax, s, (depth_softmax_bwd_kernel<false><<<grid, 256, 0, s>>>(
        depth, grad_depth, grad_logits, D, HW, (int) tiles)));
  }
  return check_launch("depth_softmax_bwd_kernel");
}

int vamp_density_gate_forward(int64_t B, int32_t C, int64_t cells, int32_t density_mode,
                              const float* voxel_output, const float* voxel_density, float* out,
                              void* stream) {
  VAMP_REQUIRE(B > 0 && B < 65536 && C > 0 && cells > 0, "B, C, cells must be positive (B < 65536)");
  VAMP_REQUIRE(voxel_output && voxel_density && out, "NULL tensor");
  VAMP_REQUIRE(density_mode == VAMP_DENSITY_SIGMOID || density_mode == VAMP_DENSITY_SDF_LAPLACE,
               "density_mode");
  VAMP_REQUIRE((cells + 255) / 256 < 0x7fffffffL, "too many cells");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned) ((cells + 255) / 256), (unsigned) B);
  VAMP_TIMED(kProfGlueGate, s, (density_gate_fwd_kernel<<<grid, 256, 0, s>>>(
      voxel_output, voxel_density, out, C, cells, density_mode)));
  return check_launch("density_gate_fwd_kernel");
}

int vamp_density_gate_backward(int64_t B, int32_t C, int64_t cells, int32_t density_mode,
                               const float* grad_out, const float* voxel_output,
                               const float* voxel_density, float* grad_voxel_output,
                               float* grad_voxel_density, void* stream) {
  VAMP_REQUIRE(B > 0 && B < 65536 && C > 0 && cells > 0, "B, C, cells must be positive (B < 65536)");
  VAMP_REQUIRE(grad_out && voxel_output && voxel_density && grad_voxel_output && grad_voxel_density,
               "NULL tensor");
  VAMP_REQUIRE(density_mode == VAMP_DENSITY_SIGMOID || density_mode == VAMP_DENSITY_SDF_LAPLACE,
               "density_mode");
  VAMP_REQUIRE((cells + 255) / 256 < 0x7fffffffL, "too many cells");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned) ((cells + 255) / 256), (unsigned) B);
  VAMP_TIMED(kProfGlueGate, s, (density_gate_bwd_kernel<<<grid, 256, 0, s>>>(
      grad_out, voxel_output, voxel_density, grad_voxel_output, grad_voxel_density, C, cells,
      density_mode)));
  return check_launch("density_gate_bwd_kernel");
}

}  // extern "C"
