// Consumer side of the path, gfx950: the density gate and the `voxel_output` 1x1
// convolution as ONE kernel each way.
//
//   reference   voxel_output = voxel_output * bev_density.tanh()          (sdf; naive: no tanh)
//               bev_feat = Conv2d(C * oZ, Cout, 1)(voxel_output.reshape(B, C * oZ, oY, oX))
//               base_vampire2.py:627-632 (the conv is `self.voxel_output[0]`, :203-209)
//
//   out[b, o, cell] = bias[o] + sum_ci W[o, ci] * vo[b, ci, cell] * gate(dens[b, ci % oZ, cell])
//
// with ci = c * oZ + z and cell = (y, x) of the BEV plane.  The gated tensor (25.6 MB at the
// reference's 256 x 256 x 10 det grid, written and read twice by the aten chain) never exists.
//
// All three products are GEMMs with the cells as the long dimension and run on the f32 matrix
// cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate -- the numerics of an fmaf
// chain); the channel-first layout makes the cells the contiguous axis of every operand, so the
// N side of each MFMA is 16 consecutive cells (64-byte rows straight from HBM, no transposes):
//   forward   out^T [Cout x cells]  = W [Cout x Cin]    . VG [Cin x cells]     A = W (LDS),  B = vo * gate (global)
//   d input   GV [Cin x cells]      = W^T [Cin x Cout]  . GO [Cout x cells]    A = W^T (LDS), B = go (LDS tile)
//   d weight  dW [Cout x Cin]       = GO [Cout x cells] . VG^T [cells x Cin]   A = go, B = vo * gate (LDS tiles)
// The backward is one persistent kernel: a workgroup stages 64 cells of go / vo / gates in LDS,
// makes grad_vo and grad_dens for them, and keeps its share of dW in accumulators across all of its
// tiles; per-workgroup partial sums are added by a second tiny kernel (no float atomics: the
// same bits on every run).
#include "common.hpp"

#include <algorithm>

namespace vamp {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxWgs = 256;         // persistent workgroups of the backward: one per CU (150 KB of LDS each)

__device__ __forceinline__ float gate_of(int mode, float vd) {
  return mode == VAMP_DENSITY_SDF_LAPLACE ? tanhf(vd) : vd;
}

// row stride (floats) of a [k][n] LDS image whose four k rows are read by the four 16-lane
// groups of a wave: pad so that the rows start in different quarters of the 64 banks
__host__ __device__ constexpr int bank_stride(int n) {
  const int r = (n + 15) / 16 * 16;
  return (r % 64 == 16 || r % 64 == 48) ? r : r + 16;
}

// ---------------------------------------------------------------------------
// forward: persistent waves, one per SIMD, each holding ITS SLICE OF THE WEIGHTS IN REGISTERS -- the A
// operand of k-step ks / row tile m is W[16 m + (lane & 15)][4 ks + (lane >> 4)], 200 values per lane
// at 160 -> 80 channels -- so a 16-cell tile is a straight run of Cin / 4 x Cout / 16 MFMAs with no
// LDS or memory operand in between; the tile's voxel_output values are fetched one tile ahead.
// (With the weights in LDS the compiler pairs every two MFMAs with a ds_read + s_waitcnt and the
// kernel runs at LDS latency: 36 us against 11 at cfg-B.)
// ---------------------------------------------------------------------------
template <int MB, int NB>            // Cout <= 16 MB, Cin <= 16 NB
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gate_conv_fwd_kernel(const float* __restrict__ vo, const float* __restrict__ dens,
                     const float* __restrict__ weight, const float* __restrict__ bias,
                     float* __restrict__ out, int Cin, int oZ, long cells, int Cout, int mode,
                     long tiles_per_b, long ntiles) {
  constexpr int NKS = NB * 4;                    // k-steps of 4 input channels
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  float* gate_s = smem + wv * oZ * 16;           // [oZ][16] of this wave's tile
  // (buffer loads: an out-of-range offset reads 0, so the padding needs no branch -- with a
  // conditional global load hipcc emits a branch and a full wait per element, 200 round trips)
  float wreg[NKS][MB];
  {
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(weight), 0, (int) ((size_t) Cout * Cin * sizeof(float)), 0x00020000);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int o = m * 16 + li, ci = ks * 4 + lk;
        const unsigned off = (o < Cout && ci < Cin) ? (unsigned) ((o * Cin + ci) * sizeof(float)) : 0xfffffff0u;
        wreg[ks][m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, off, 0, 0));
      }
  }
  // this lane's bias values (rows 4 lk + r of every M tile), likewise
  float bo[MB][4];
  {
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(bias), 0, bias ? (int) (Cout * sizeof(float)) : 0, 0x00020000);
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bo[m][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned) ((m * 16 + 4 * lk + r) * sizeof(float)), 0, 0));
  }
  // z = ci % oZ of this lane's channel at every k-step, packed 8 bits each (oZ <= 32)
  const int zstep = 4 % oZ;

  long tile = (long) blockIdx.x * 4 + wv;
  const long tstep = (long) gridDim.x * 4;
  // B operand through a buffer descriptor over the sample's voxel_output: per-lane byte offset (row
  // lk, this lane's cell) + a scalar offset per k-step; rows beyond Cin are out of range and read 0
  float cur[NKS], nxt[NKS];
  const unsigned kstride = (unsigned) (4 * cells * sizeof(float));
  auto load_tile = [&](float (&v)[NKS], long t) {
    const long b = __builtin_amdgcn_readfirstlane((int) (t / tiles_per_b));
    const long cell = min((t - b * tiles_per_b) * 16 + li, cells - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(vo + b * Cin * cells), 0, (int) ((size_t) Cin * cells * sizeof(float)), 0x00020000);
    const unsigned voff = (unsigned) (((long) lk * cells + cell) * sizeof(float));
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      v[ks] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, ks * kstride, 0));
  };
  // the tile's densities, fetched with its voxel_output values: lane group lk takes z = lk, lk + 4, ...
  // (8 slots cover oZ <= 32; slots beyond oZ are out of the descriptor's range and read 0)
  float dcur[8], dnxt[8];
  auto load_dens = [&](float (&d)[8], long t) {
    const long b = __builtin_amdgcn_readfirstlane((int) (t / tiles_per_b));
    const long cell = min((t - b * tiles_per_b) * 16 + li, cells - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dens + b * oZ * cells), 0, (int) ((size_t) oZ * cells * sizeof(float)), 0x00020000);
    const unsigned voff = (unsigned) (((long) lk * cells + cell) * sizeof(float));
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, j * kstride, 0));
  };
  if (tile < ntiles) {
    load_tile(cur, tile);
    load_dens(dcur, tile);
  }
  for (; tile < ntiles; tile += tstep) {
    const long b = tile / tiles_per_b;
    const long cell = (tile - b * tiles_per_b) * 16 + li;
    const bool live = cell < cells;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (4 * j + lk < oZ) gate_s[(4 * j + lk) * 16 + li] = gate_of(mode, dcur[j]);
    {
      int z = lk % oZ;                           // z of channel ci = 4 ks + lk
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        cur[ks] *= gate_s[z * 16 + li];
        z += zstep;
        if (z >= oZ) z -= oZ;
      }
    }
    if (tile + tstep < ntiles) {
      load_tile(nxt, tile + tstep);
      load_dens(dnxt, tile + tstep);
    }
    f32x4 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ks][m], cur[ks], acc[m], 0, 0, 0);
    // D: row (output channel) = 4 (lane >> 4) + reg, column (cell) = lane & 15
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = m * 16 + 4 * lk + r;
        if (o < Cout && live) out[(b * Cout + o) * cells + cell] = acc[m][r] + bo[m][r];
      }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) cur[ks] = nxt[ks];
#pragma unroll
    for (int j = 0; j < 8; ++j) dcur[j] = dnxt[j];
  }
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
constexpr int kTS = 84;      // row stride of the staged 64-cell tiles: 84 = 64 + 20 keeps both the
                             // [k][n] reads (4 rows x 16 columns) and the transposed [m][k] reads
                             // (16 rows x 4 columns) of a wave on (nearly) disjoint banks

template <int MB, int NB>
struct GcBwdLds {
  static constexpr int CP = MB * 16, KP = NB * 16;
  static constexpr int W2S = bank_stride(KP);
  // floats: W [CP][W2S], go tile [CP][kTS], vo tile [KP][kTS], gates [oZ][64], gd [4][oZ][64], z of ci [KP], bias sums [CP]
  __host__ __device__ static constexpr size_t floats(int oZ) {
    return (size_t) CP * W2S + (size_t) CP * kTS + (size_t) KP * kTS + (size_t) oZ * 64 + (size_t) 4 * oZ * 64 + KP + CP;
  }
};

template <int MB, int NB>
__global__ void __launch_bounds__(256)
gate_conv_bwd_kernel(const float* __restrict__ go, const float* __restrict__ vo,
                     const float* __restrict__ dens, const float* __restrict__ weight,
                     float* __restrict__ gvo, float* __restrict__ gdens, float* __restrict__ dw_part,
                     float* __restrict__ db_part, int Cin, int oZ, long cells, int Cout, int mode,
                     long tiles_per_b, long ntiles) {
  using L = GcBwdLds<MB, NB>;
  constexpr int CP = L::CP, KP = L::KP, W2S = L::W2S;
  constexpr int NBW = (NB + 3) / 4;              // N tiles of dW a wave owns (nt = wave, wave + 4, ...)
  extern __shared__ float smem[];
  float* w2_s = smem;                            // [CP][W2S]: W[o][ci], zero padded
  float* go_s = w2_s + CP * W2S;                 // [CP][kTS]
  float* vo_s = go_s + CP * kTS;                 // [KP][kTS] raw voxel_output
  float* gate_s = vo_s + KP * kTS;               // [oZ][64]
  float* gd_all = gate_s + oZ * 64;              // [4][oZ][64] per-lane partial sums of grad_dens
  int* zof_s = reinterpret_cast<int*>(gd_all + 4 * oZ * 64);      // [KP]: ci % oZ
  float* bsum_s = reinterpret_cast<float*>(zof_s + KP);           // [CP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  float* gd_s = gd_all + wv * oZ * 64;

  {
    constexpr int NLD = (CP * KP + 255) / 256;
    float wr[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * 256;
      const int o = e / KP, ci = e - o * KP;
      wr[i] = (e < CP * KP && o < Cout && ci < Cin) ? weight[(long) o * Cin + ci] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * 256;
      const int o = e / KP, ci = e - o * KP;
      if (e < CP * KP) w2_s[o * W2S + ci] = wr[i];
    }
  }
  for (int e = tid; e < KP; e += 256) zof_s[e] = e % oZ;
  for (int e = tid; e < CP; e += 256) bsum_s[e] = 0.f;

  f32x4 dacc[MB][NBW];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < NBW; ++j) dacc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // a tile's go / vo values on their way to LDS: thread t holds column t & 63 of rows (t >> 6) + 4 i
  float sg[CP / 4], sv[KP / 4];
  // (buffer loads: rows beyond Cout / Cin and cells beyond the plane are out of range and read 0 -- a
  // conditional global load costs a branch and a full wait per element)
  const unsigned rstride = (unsigned) (4 * cells * sizeof(float));     // four rows of a [rows][cells] tensor
  auto stage = [&](long tile) {
    const long b = __builtin_amdgcn_readfirstlane((int) (tile / tiles_per_b));
    const long c = (tile - b * tiles_per_b) * 64 + (tid & 63);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(go + b * Cout * cells), 0, (int) ((size_t) Cout * cells * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(vo + b * Cin * cells), 0, (int) ((size_t) Cin * cells * sizeof(float)), 0x00020000);
    const unsigned voff = c < cells ? (unsigned) (((long) (tid >> 6) * cells + c) * sizeof(float)) : 0xfffffff0u;
#pragma unroll
    for (int i = 0; i < CP / 4; ++i) sg[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grs, voff, i * rstride, 0));
#pragma unroll
    for (int i = 0; i < KP / 4; ++i) sv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, voff, i * rstride, 0));
  };
  if (blockIdx.x < ntiles) stage(blockIdx.x);
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long b = tile / tiles_per_b;
    const long c0 = (tile - b * tiles_per_b) * 64;
    __syncthreads();                             // the previous tile's readers (and the weight image)
#pragma unroll
    for (int i = 0; i < CP / 4; ++i) go_s[((tid >> 6) + 4 * i) * kTS + (tid & 63)] = sg[i];
#pragma unroll
    for (int i = 0; i < KP / 4; ++i) vo_s[((tid >> 6) + 4 * i) * kTS + (tid & 63)] = sv[i];
    {
      const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(dens + b * oZ * cells), 0, (int) ((size_t) oZ * cells * sizeof(float)), 0x00020000);
      const long c = c0 + (tid & 63);
      const unsigned voff = c < cells ? (unsigned) (((long) (tid >> 6) * cells + c) * sizeof(float)) : 0xfffffff0u;
      float dz[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) dz[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(drs, voff, j * rstride, 0));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (4 * j + (tid >> 6) < oZ) gate_s[(4 * j + (tid >> 6)) * 64 + (tid & 63)] = c < cells ? gate_of(mode, dz[j]) : 0.f;
    }
    for (int e = lane; e < oZ * 64; e += 64) gd_s[e] = 0.f;
    __syncthreads();
    // the next tile's operands start their way from HBM now; they are consumed at the top of the
    // next iteration, after this tile's MFMAs
    if (tile + gridDim.x < ntiles) stage(tile + gridDim.x);

    // grad_bias: row sums of the go tile (wave w: rows w, w + 4, ...)
    for (int o = wv; o < CP; o += 4) {
      float s = go_s[o * kTS + lane];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) bsum_s[o] += s;
    }

    // ---- d input: GV [Cin x 16 cells of this wave] = W^T . GO ----
    {
      f32x4 xacc[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) xacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int kend = min(CP, (Cout + 3) & ~3);
      for (int o0 = 0; o0 < kend; o0 += 4) {
        const float bvv = go_s[(o0 + lk) * kTS + 16 * wv + li];
#pragma unroll
        for (int m = 0; m < NB; ++m)
          xacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2_s[(o0 + lk) * W2S + m * 16 + li], bvv, xacc[m], 0, 0, 0);
      }
      // D: row (ci) = m 16 + 4 (lane >> 4) + reg, column (cell) = 16 wave + (lane & 15)
      const int cl = 16 * wv + li;
      const long c = c0 + cl;
#pragma unroll
      for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = m * 16 + 4 * lk + r;
          if (ci < Cin) {
            const int z = zof_s[ci];
            const float gv = xacc[m][r];
            if (c < cells) gvo[(b * Cin + ci) * cells + c] = gv * gate_s[z * 64 + cl];
            gd_s[z * 64 + lane] = __builtin_fmaf(gv, vo_s[ci * kTS + cl], gd_s[z * 64 + lane]);   // this lane's own slot
          }
        }
      // grad_dens of the wave's 16 cells: the four lane groups' partial sums, times d gate
      for (int z = lk; z < oZ; z += 4) {
        const float s = (gd_s[z * 64 + li] + gd_s[z * 64 + 16 + li]) + (gd_s[z * 64 + 32 + li] + gd_s[z * 64 + 48 + li]);
        const float gt = gate_s[z * 64 + cl];
        if (c < cells) gdens[(b * oZ + z) * cells + c] = mode == VAMP_DENSITY_SDF_LAPLACE ? s * (1.0f - gt * gt) : s;
      }
    }

    // ---- d weight: dW [Cout x Cin] += GO [Cout x 64 cells] . VG^T ----
    for (int k0 = 0; k0 < 64; k0 += 4) {
      float a[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) a[m] = go_s[(m * 16 + li) * kTS + k0 + lk];
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int nt = wv + 4 * j;
        if (nt < NB) {                           // wave-uniform
          const int ci = nt * 16 + li;
          const float bvv = vo_s[ci * kTS + k0 + lk] * gate_s[zof_s[ci] * 64 + k0 + lk];
#pragma unroll
          for (int m = 0; m < MB; ++m) dacc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bvv, dacc[m][j], 0, 0, 0);
        }
      }
    }
  }
  // per-workgroup partial sums: dW row (o) = m 16 + 4 (lane >> 4) + reg, column (ci) = nt 16 + (lane & 15)
  float* dwp = dw_part + (size_t) blockIdx.x * CP * KP;
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int nt = wv + 4 * j;
      if (nt < NB) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dwp[(m * 16 + 4 * lk + r) * KP + nt * 16 + li] = dacc[m][j][r];
      }
    }
  __syncthreads();
  for (int e = tid; e < CP; e += 256) db_part[(size_t) blockIdx.x * CP + e] = bsum_s[e];
}

// dW[o][ci] = sum over workgroups; grad_bias likewise.  A workgroup adds 64 neighbouring entries of
// the (padded) partial images: its four waves split the workgroups' partials, 8 loads in flight each.
__global__ void __launch_bounds__(256)
gate_conv_reduce_kernel(const float* __restrict__ dw_part, const float* __restrict__ db_part,
                        float* __restrict__ gw, float* __restrict__ gb, int Cin, int Cout, int CP, int KP,
                        int nwg) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ndw = CP * KP;
  const int e = blockIdx.x * 64 + lane;          // entry of the padded dW image, then of the bias sums
  const bool is_w = e < ndw, is_b = !is_w && e - ndw < CP;
  const float* src = is_w ? dw_part + e : db_part + (e - ndw);
  const size_t stride = is_w ? (size_t) ndw : (size_t) CP;
  float s = 0.f;
  if (is_w || is_b) {
    int g = wv;
    for (; g + 28 < nwg; g += 32) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = src[(size_t) (g + 4 * i) * stride];
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; g < nwg; g += 4) s += src[(size_t) g * stride];
  }
  part[wv][lane] = s;
  __syncthreads();
  if (wv == 0) {
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (is_w) {
      const int o = e / KP, ci = e - o * KP;
      if (o < Cout && ci < Cin) gw[o * Cin + ci] = s;
    } else if (is_b && gb && e - ndw < Cout) {
      gb[e - ndw] = s;
    }
  }
}

struct GcShape {
  int mb, nb;                        // 0: not supported
};
static GcShape gc_shape(int Cin, int Cout, int oZ) {
  GcShape s{0, 0};
  if (Cin <= 0 || Cout <= 0 || oZ <= 0 || oZ > 32) return s;
  s.mb = Cout <= 16 ? 1 : (Cout <= 80 ? 5 : 0);
  s.nb = Cin <= 64 ? 4 : (Cin <= 160 ? 10 : 0);
  if (!s.mb || !s.nb) s.mb = s.nb = 0;
  // the backward's LDS image must fit the CU (160 KB): otherwise the forward would succeed and the
  // backward fail in the middle of training -- such shapes take the caller's unfused path
  constexpr size_t kLdsLimit = 160 * 1024;
  size_t lds = 0;
  if (s.mb == 1 && s.nb == 4) lds = GcBwdLds<1, 4>::floats(oZ);
  else if (s.mb == 1 && s.nb == 10) lds = GcBwdLds<1, 10>::floats(oZ);
  else if (s.mb == 5 && s.nb == 4) lds = GcBwdLds<5, 4>::floats(oZ);
  else if (s.mb == 5 && s.nb == 10) lds = GcBwdLds<5, 10>::floats(oZ);
  if (lds * sizeof(float) > kLdsLimit) s.mb = s.nb = 0;
  return s;
}

template <int MB, int NB>
static int launch_fwd(long B, int Cin, int oZ, long cells, int Cout, int mode, const float* vo,
                      const float* dens, const float* w, const float* bias, float* out, hipStream_t s) {
  const size_t lds = (size_t) 4 * oZ * 16 * sizeof(float);
  auto k = gate_conv_fwd_kernel<MB, NB>;
  // persistent: one workgroup (4 waves, one per SIMD) per CU, a wave per 16-cell tile
  const long tiles_per_b = (cells + 15) / 16, ntiles = tiles_per_b * B;
  const unsigned grid = (unsigned) std::min<long>((ntiles + 3) / 4, kMaxWgs);
  VAMP_TIMED(kProfGlueGate, s, (k<<<grid, 256, lds, s>>>(vo, dens, w, bias, out, Cin, oZ, cells, Cout, mode,
                                                        tiles_per_b, ntiles)));
  return check_launch("gate_conv_fwd_kernel");
}

template <int MB, int NB>
static int launch_bwd(long B, int Cin, int oZ, long cells, int Cout, int mode, const float* go,
                      const float* vo, const float* dens, const float* w, float* gvo, float* gdens,
                      float* gw, float* gb, float* ws, hipStream_t s) {
  using L = GcBwdLds<MB, NB>;
  const size_t lds = L::floats(oZ) * sizeof(float);
  auto k = gate_conv_bwd_kernel<MB, NB>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess)
    return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);
  const long tiles_per_b = (cells + 63) / 64, ntiles = tiles_per_b * B;
  const int nwg = (int) std::min<long>(ntiles, kMaxWgs);
  float* dw_part = ws;
  float* db_part = ws + (size_t) kMaxWgs * L::CP * L::KP;
  VAMP_TIMED(kProfGlueGate, s, (k<<<nwg, 256, lds, s>>>(go, vo, dens, w, gvo, gdens, dw_part, db_part, Cin, oZ,
                                                       cells, Cout, mode, tiles_per_b, ntiles)));
  if (int e = check_launch("gate_conv_bwd_kernel")) return e;
  const int n = L::CP * L::KP + L::CP;
  VAMP_TIMED(kProfGlueGate, s, (gate_conv_reduce_kernel<<<(n + 63) / 64, 256, 0, s>>>(dw_part, db_part, gw, gb, Cin, Cout,
                                                                                     L::CP, L::KP, nwg)));
  return check_launch("gate_conv_reduce_kernel");
}

}  // namespace
}  // namespace vamp

using namespace vamp;

extern "C" {

int vamp_gate_conv1x1_supported(int32_t C, int32_t oZ, int32_t Cout) {
  return gc_shape(C * oZ, Cout, oZ).mb != 0;
}

size_t vamp_gate_conv1x1_workspace_bytes(int32_t C, int32_t oZ, int32_t Cout) {
  const GcShape g = gc_shape(C * oZ, Cout, oZ);
  if (!g.mb) return 0;
  return ((size_t) kMaxWgs * (g.mb * 16) * (g.nb * 16) + (size_t) kMaxWgs * (g.mb * 16)) * sizeof(float);
}

#define VAMP_GC_DISPATCH(FN, ...)                                   \
  (g.mb == 1 ? (g.nb == 4 ? FN<1, 4>(__VA_ARGS__) : FN<1, 10>(__VA_ARGS__)) \
             : (g.nb == 4 ? FN<5, 4>(__VA_ARGS__) : FN<5, 10>(__VA_ARGS__)))

int vamp_gate_conv1x1_forward(int64_t B, int32_t C, int32_t oZ, int64_t cells, int32_t Cout,
                              int32_t density_mode, const float* voxel_output,
                              const float* voxel_density, const float* weight, const float* bias,
                              float* out, void* stream) {
  VAMP_REQUIRE(B > 0 && B < 65536 && C > 0 && cells > 0, "B, C, cells must be positive (B < 65536)");
  VAMP_REQUIRE(voxel_output && voxel_density && weight && out, "NULL tensor");
  VAMP_REQUIRE(density_mode == VAMP_DENSITY_SIGMOID || density_mode == VAMP_DENSITY_SDF_LAPLACE, "density_mode");
  const GcShape g = gc_shape(C * oZ, Cout, oZ);
  VAMP_REQUIRE(g.mb != 0, "shape not supported (C * oZ <= 160, Cout <= 80, oZ <= 32): see vamp_gate_conv1x1_supported");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return VAMP_GC_DISPATCH(launch_fwd, B, C * oZ, oZ, cells, Cout, density_mode, voxel_output, voxel_density,
                          weight, bias, out, s);
}

int vamp_gate_conv1x1_backward(int64_t B, int32_t C, int32_t oZ, int64_t cells, int32_t Cout,
                               int32_t density_mode, const float* grad_out, const float* voxel_output,
                               const float* voxel_density, const float* weight,
                               float* grad_voxel_output, float* grad_voxel_density, float* grad_weight,
                               float* grad_bias, void* workspace, size_t workspace_bytes, void* stream) {
  VAMP_REQUIRE(B > 0 && B < 65536 && C > 0 && cells > 0, "B, C, cells must be positive (B < 65536)");
  VAMP_REQUIRE(grad_out && voxel_output && voxel_density && weight && grad_voxel_output && grad_voxel_density && grad_weight,
               "NULL tensor");
  VAMP_REQUIRE(density_mode == VAMP_DENSITY_SIGMOID || density_mode == VAMP_DENSITY_SDF_LAPLACE, "density_mode");
  const GcShape g = gc_shape(C * oZ, Cout, oZ);
  VAMP_REQUIRE(g.mb != 0, "shape not supported (C * oZ <= 160, Cout <= 80, oZ <= 32): see vamp_gate_conv1x1_supported");
  const size_t need = vamp_gate_conv1x1_workspace_bytes(C, oZ, Cout);
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return VAMP_GC_DISPATCH(launch_bwd, B, C * oZ, oZ, cells, Cout, density_mode, grad_out, voxel_output,
                          voxel_density, weight, grad_voxel_output, grad_voxel_density, grad_weight, grad_bias,
                          static_cast<float*>(workspace), s);
}

}  // extern "C"
