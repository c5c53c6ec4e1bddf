// 3x3x3 / stride 1 / pad 1 Conv3d of the UNet between lift and render, gfx950:
// nn.Conv3d(cin, cout, 3, 1, 1, bias=False) with cin, cout in {16, 32}, base_vampire2.py:20, 40-60
// (init_dres, conv2, conv4, conv5, conv6 of both Hourglass3D blocks), fp32 in / fp32 out.
//
// Implicit GEMM on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, one
// rounding per accumulate -- the same numerics as an fmaf chain):
//     out[voxel][co] = sum_{tap, ci} in[voxel + tap][ci] * w[co][ci][tap]
// M = voxels (16 consecutive x per MFMA), N = 16 output channels, K = 4 input channels per step.
//   lane l supplies A[voxel l & 15][ci k0 + (l >> 4)] straight from the NCDHW input (16 consecutive
//   floats of 4 channel planes; the 27 taps re-read the same lines from L1) and
//   B[ci k0 + (l >> 4)][co l & 15] from an LDS image of the weights laid out [tap][ci][co].
//   A wave owns 64 consecutive x of one (z, y) row x all output channels: 4 x (cout / 16)
//   accumulator tiles, every B value feeds 4 MFMAs and every A value cout / 16.
// The data gradient is the same kernel on the flipped, transposed weights (FLIP).
// The weight gradient is a GEMM over the voxels: dw[co][ci][tap] = sum_voxel dout[voxel][co] *
//   in[voxel + tap][ci]: M = co, N = ci, K = 4 voxels per step; a workgroup reduces a slab of rows
//   and adds its 27 x cout x cin partial sums to the result with float atomics (cout * cin * 27
//   addresses, a few hundred workgroups).
#include "common.hpp"

#include <cstdlib>

namespace vamp {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
  int B, Z, Y, X;
};

// DPP moves inside the 16-lane rows of a wave (a row = the 16 x positions of one K channel)
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {          // lane i <- lane (i - N) mod 16
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shr1(float old, float v) {   // lane i <- lane i - 1; lane 0 keeps old
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shl1(float old, float v) {   // lane i <- lane i + 1; lane 15 keeps old
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x101, 0xf, 0xf, false));
}

// ---------------------------------------------------------------------------
// forward / data gradient
// ---------------------------------------------------------------------------
template <int CIN, int COUT, bool FLIP>
__global__ void __launch_bounds__(512)
conv3d_fwd_kernel(ConvParams P, const float* __restrict__ in, const float* __restrict__ w,
                  float* __restrict__ out, int tiles_x, long ntiles) {
  extern __shared__ float wl[];                       // [27][CIN][COUT]
  constexpr int NT = COUT / 16;                       // N tiles
  // weights -> LDS.  forward: wl[t][ci][co] = w[co][ci][t];
  // data gradient (the roles of the channels swap): wl[t][k][j] = w[co = k][ci = j][26 - t]
  // (COUT == 32: the two 16-column halves of odd rows are swapped, so that the four K rows a
  // load touches fall on both halves of the 32 banks)
  for (int e = threadIdx.x; e < 27 * CIN * COUT; e += blockDim.x) {
    const int j = e % COUT, k = (e / COUT) % CIN, t = e / (COUT * CIN);
    const int js = COUT == 32 ? (j ^ ((k & 1) << 4)) : j;
    wl[(t * CIN + k) * COUT + js] = FLIP ? w[((long) k * COUT + j) * 27 + (26 - t)] : w[((long) j * CIN + k) * 27 + t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const long plane = (long) P.Z * P.Y * P.X;
  const int nw = blockDim.x >> 6;
  for (long tile = (long) blockIdx.x * nw + wave; tile < ntiles; tile += (long) gridDim.x * nw) {
    const int tx = (int) (tile % tiles_x);
    const long row = tile / tiles_x;                  // (b, z, y)
    const int y = (int) (row % P.Y), z = (int) ((row / P.Y) % P.Z), b = (int) (row / ((long) P.Y * P.Z));
    const int x0 = tx * 64;
    const float* inb = in + (long) b * CIN * plane;
    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the three x-taps of a (dz, dy) row read the same 64 + 2 values: they are loaded once and the
    // shifted operands come from neighbouring lanes (the load unit, not the matrix core, limits
    // this kernel when every MFMA has its own global load)
    bool ok[4];
    int xo[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int xx = x0 + m * 16 + li;
      ok[m] = xx < P.X;
      xo[m] = ok[m] ? xx : 0;
    }
    // lane li == 0 also fetches x0 - 1, lane li == 15 fetches x0 + 64
    const int xe = li == 0 ? x0 - 1 : x0 + 64;
    const bool oke = (li == 0 || li == 15) && xe >= 0 && xe < P.X;
    for (int p = 0; p < 9; ++p) {
      const int dz = p / 3 - 1, dy = p % 3 - 1;
      const int zz = z + dz, yy = y + dy;
      if (zz < 0 || zz >= P.Z || yy < 0 || yy >= P.Y) continue;          // wave-uniform
      const float* rowp = inb + ((long) zz * P.Y + yy) * P.X;
      const float* wt = wl + p * 3 * CIN * COUT;
#pragma unroll 2
      for (int k0 = 0; k0 < CIN; k0 += 4) {
        const float* cp = rowp + (long) (k0 + lk) * plane;
        float c[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) c[m] = ok[m] ? cp[xo[m]] : 0.f;
        const float e = oke ? cp[xe] : 0.f;
        // x - 1 / x + 1 operands from the neighbouring lanes of the 16-lane row (DPP, no LDS):
        // row_shr / row_shl by one, the lane at the end of the row keeps `old`, which is the
        // rotated neighbour tile's edge value (or the halo value e)
        float lf[4], rt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float u0 = m > 0 ? dpp_row_ror<1>(c[m > 0 ? m - 1 : 0]) : e;
          lf[m] = dpp_row_shr1(u0, c[m]);
          const float u1 = m < 3 ? dpp_row_ror<15>(c[m < 3 ? m + 1 : 3]) : e;
          rt[m] = dpp_row_shl1(u1, c[m]);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          float bv[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n)
            bv[n] = wt[(d * CIN + k0 + lk) * COUT + (COUT == 32 ? ((n * 16 + li) ^ ((lk & 1) << 4)) : n * 16 + li)];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float av = d == 0 ? lf[m] : (d == 1 ? c[m] : rt[m]);
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[n], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
    // C/D: col (output channel) = lane & 15, row (voxel) = (lane >> 4) * 4 + reg
    float* ob = out + (long) b * COUT * plane + ((long) z * P.Y + y) * P.X;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int xb = x0 + m * 16 + lk * 4;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float* op = ob + (long) (n * 16 + li) * plane + xb;
        if (xb + 3 < P.X && ((P.X & 3) == 0)) {
          *reinterpret_cast<f32x4*>(op) = acc[m][n];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (xb + r < P.X) op[r] = acc[m][n][r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient: dw[co][ci][t] = sum_{b, voxel} dout[b][co][voxel] * in[b][ci][voxel + tap t]
// MFMA: M = co (16 per tile), N = ci (16 per tile), K = 4 consecutive x:
//   A[co l & 15][x k0 + (l >> 4)] = dout,  B[x k0 + (l >> 4)][ci l & 15] = in shifted by the tap.
// Both operands are channel-strided in NCDHW memory, so a workgroup stages rows in LDS with
// coalesced loads: the dout row of its output row (b, z, y) once, then for each of the nine
// (dz, dy) pairs the input row (z + dz, y + dy) with a one-element halo.  The LDS images are
// [channel][XS] with XS = 4 (mod 32): lane (li, lk) reads word li * XS + x + lk, bank 4 li + lk,
// every bank exactly twice per wave -- the minimum.  A wave owns one (co tile, ci tile) pair with
// the three x-taps and a phase of the x steps, and keeps the 27 accumulator tiles of all nine pairs
// (108 VGPRs) over the workgroup's slab of rows; partial sums go to a [workgroup][27 cout cin]
// buffer and a second kernel adds them up in workgroup order (no atomics: the 27 * cout * cin
// addresses would each see one float atomic per workgroup).
// ---------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ void __launch_bounds__(256)
conv3d_wgrad_kernel(ConvParams P, const float* __restrict__ in, const float* __restrict__ dout,
                    float* __restrict__ part, long nrows, int rows_per_block, int XS) {
  extern __shared__ float lds[];
  float* G = lds;                      // [COUT][XS]  dout row, G[co][x]
  float* I = lds + COUT * XS;          // [CIN][XS]   input row, I[ci][x + 1] (I[ci][0] = x -1 = 0)
  constexpr int MT = COUT / 16, NTI = CIN / 16, NG = MT * NTI;      // NG in {1, 2, 4}
  constexpr int NPH = 4 / NG;                                       // x phases per group
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int grp = wave % NG, phase = wave / NG;
  const int gm = grp % MT, gn = grp / MT;
  const long plane = (long) P.Z * P.Y * P.X;
  const long r0 = (long) blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
  const int steps = (P.X + 3) / 4;
  f32x4 acc[9][3];
#pragma unroll
  for (int p = 0; p < 9; ++p)
#pragma unroll
    for (int d = 0; d < 3; ++d) acc[p][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  // zero the images once: the pads beyond X (and the halo columns) stay zero
  for (int e = tid; e < (COUT + CIN) * XS; e += 256) lds[e] = 0.f;
  for (long row = r0; row < r1; ++row) {
    const int y = (int) (row % P.Y), z = (int) ((row / P.Y) % P.Z), b = (int) (row / ((long) P.Y * P.Z));
    __syncthreads();
    {
      const float* gp = dout + (long) b * COUT * plane + ((long) z * P.Y + y) * P.X;
      for (int e = tid; e < COUT * P.X; e += 256) {
        const int co = e / P.X, x = e - co * P.X;
        G[co * XS + x] = gp[(long) co * plane + x];
      }
    }
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      const int dz = p / 3 - 1, dy = p % 3 - 1;
      const int zz = z + dz, yy = y + dy;
      if (zz < 0 || zz >= P.Z || yy < 0 || yy >= P.Y) continue;          // workgroup-uniform
      __syncthreads();                                                   // previous pair is consumed
      {
        const float* ip = in + (long) b * CIN * plane + ((long) zz * P.Y + yy) * P.X;
        for (int e = tid; e < CIN * P.X; e += 256) {
          const int ci = e / P.X, x = e - ci * P.X;
          I[ci * XS + x + 1] = ip[(long) ci * plane + x];
        }
      }
      __syncthreads();
      const float* ga = G + (gm * 16 + li) * XS + lk;
      const float* ib = I + (gn * 16 + li) * XS + lk;
      for (int st = phase; st < steps; st += NPH) {
        const int xk = st * 4;
        const float av = ga[xk];
        const float b0 = ib[xk], b1 = ib[xk + 1], b2 = ib[xk + 2];
        acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[p][0], 0, 0, 0);
        acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[p][1], 0, 0, 0);
        acc[p][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2, acc[p][2], 0, 0, 0);
      }
    }
  }
  // C/D: col = ci = lane & 15, row = co = (lane >> 4) * 4 + reg; tap t = p * 3 + d.  The NPH phases
  // of a group are folded into phase 0 through LDS, one phase at a time (27 tiles x 256 floats),
  // then phase 0 writes the workgroup's partial sums.
  float* red = lds;                                     // [NG][27][64][4]
  for (int ph = 1; ph < NPH; ++ph) {
    __syncthreads();
    if (phase == ph) {
#pragma unroll
      for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          *reinterpret_cast<f32x4*>(red + ((grp * 27 + p * 3 + d) * 64 + lane) * 4) = acc[p][d];
    }
    __syncthreads();
    if (phase == 0) {
#pragma unroll
      for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          acc[p][d] += *reinterpret_cast<const f32x4*>(red + ((grp * 27 + p * 3 + d) * 64 + lane) * 4);
    }
  }
  if (phase == 0) {
    float* pb = part + (long) blockIdx.x * 27 * COUT * CIN;
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        // partial layout [t][co][ci]: lanes li = consecutive ci
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pb[((p * 3 + d) * COUT + gm * 16 + lk * 4 + r) * CIN + gn * 16 + li] = acc[p][d][r];
      }
  }
}

// dw[co][ci][t] = sum over workgroups of part[wg][t][co][ci], in workgroup order
__global__ void __launch_bounds__(256)
conv3d_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblk, int cin,
                           int cout) {
  const int n = 27 * cout * cin;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < nblk; k += 4) {
    s0 += part[(long) k * n + e];
    s1 += part[(long) (k + 1) * n + e];
    s2 += part[(long) (k + 2) * n + e];
    s3 += part[(long) (k + 3) * n + e];
  }
  for (; k < nblk; ++k) s0 += part[(long) k * n + e];
  const int ci = e % cin, co = (e / cin) % cout, t = e / (cin * cout);
  dw[((long) co * cin + ci) * 27 + t] = (s0 + s1) + (s2 + s3);
}

int wgrad_xs(int X) {                  // row pitch of the LDS images: >= 4 ceil(X / 4) + 4, = 4 mod 32
  const int need = (X + 3) / 4 * 4 + 4;
  return (need - 4 + 31) / 32 * 32 + 4;
}

int wgrad_blocks(const VampConvDesc* d) {
  const long nrows = (long) d->B * d->Z * d->Y;
  static int cap = -1;
  if (cap < 0) {
    const char* e = getenv("VAMP_WGRAD_BLOCKS");
    cap = e ? atoi(e) : 0;
  }
  const int per_cu = cap > 0 ? cap : 2;       // measured at 16x200x200: 2 per CU 588 us fwd + bwd, 3: 616, 4: 679
  return (int) std::min<long>(nrows, 256L * per_cu);
}

int check(const VampConvDesc* d) {
  VAMP_REQUIRE(d != nullptr, "desc is NULL");
  VAMP_REQUIRE(d->B > 0 && d->Z > 0 && d->Y > 0 && d->X > 0, "sizes must be positive");
  VAMP_REQUIRE((d->cin == 16 || d->cin == 32) && (d->cout == 16 || d->cout == 32),
               "cin, cout must be 16 or 32");
  VAMP_REQUIRE((long) d->B * std::max(d->cin, d->cout) * d->Z * d->Y * d->X < 0x7fffffffL * 4L, "tensor too large");
  return VAMP_OK;
}

template <int CIN, int COUT, bool FLIP>
int launch_fwd(const VampConvDesc* d, const float* in, const float* w, float* out, hipStream_t s) {
  ConvParams P{d->B, d->Z, d->Y, d->X};
  const int tiles_x = (d->X + 63) / 64;
  const long ntiles = (long) d->B * d->Z * d->Y * tiles_x;
  const size_t lds = (size_t) 27 * CIN * COUT * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_fwd_kernel<CIN, COUT, FLIP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    attr_set = true;
  }
  // persistent workgroups of 8 waves that share one weight image: 1 / 2 / 4 per CU by LDS size
  // (8 waves: 60-65 TFLOP/s at 16x200x200 against 52-56 with 4; small volumes have too few
  // tiles for that and keep 4)
  const int wpb = ntiles >= 1024 ? 8 : 4;
  const int per_cu = lds > 80 * 1024 ? 1 : (lds > 40 * 1024 ? 2 : 4);
  const unsigned grid = (unsigned) std::min<long>((ntiles + wpb - 1) / wpb, 256L * per_cu);
  VAMP_TIMED(FLIP ? kProfConvDgrad : kProfConvFwd, s, (conv3d_fwd_kernel<CIN, COUT, FLIP><<<grid, 64 * wpb, lds, s>>>(
      P, in, w, out, tiles_x, ntiles)));
  return check_launch("conv3d_fwd_kernel");
}

}  // namespace
}  // namespace vamp

using namespace vamp;

extern "C" {

int vamp_conv3d_forward(const VampConvDesc* d, const float* in, const float* weight, float* out,
                        void* stream) {
  if (int e = check(d)) return e;
  VAMP_REQUIRE(in && weight && out, "NULL tensor");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->cin == 16 && d->cout == 16) return launch_fwd<16, 16, false>(d, in, weight, out, s);
  if (d->cin == 16 && d->cout == 32) return launch_fwd<16, 32, false>(d, in, weight, out, s);
  if (d->cin == 32 && d->cout == 16) return launch_fwd<32, 16, false>(d, in, weight, out, s);
  return launch_fwd<32, 32, false>(d, in, weight, out, s);
}

/* grad_in [B, cin, Z, Y, X] from grad_out [B, cout, Z, Y, X]: the forward kernel on the flipped,
   transposed weights (its "input channels" are cout, its "output channels" cin) */
int vamp_conv3d_backward_data(const VampConvDesc* d, const float* grad_out, const float* weight,
                              float* grad_in, void* stream) {
  if (int e = check(d)) return e;
  VAMP_REQUIRE(grad_out && weight && grad_in, "NULL tensor");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->cin == 16 && d->cout == 16) return launch_fwd<16, 16, true>(d, grad_out, weight, grad_in, s);
  if (d->cin == 16 && d->cout == 32) return launch_fwd<32, 16, true>(d, grad_out, weight, grad_in, s);
  if (d->cin == 32 && d->cout == 16) return launch_fwd<16, 32, true>(d, grad_out, weight, grad_in, s);
  return launch_fwd<32, 32, true>(d, grad_out, weight, grad_in, s);
}

size_t vamp_conv3d_workspace_bytes(const VampConvDesc* d) {
  if (check(d)) return 0;
  return align_up((size_t) wgrad_blocks(d) * 27 * d->cout * d->cin * sizeof(float), 256);
}

int vamp_conv3d_backward_weight(const VampConvDesc* d, const float* in, const float* grad_out,
                                float* grad_weight, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (int e = check(d)) return e;
  VAMP_REQUIRE(in && grad_out && grad_weight && workspace, "NULL tensor");
  VAMP_REQUIRE(workspace_bytes >= vamp_conv3d_workspace_bytes(d), "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvParams P{d->B, d->Z, d->Y, d->X};
  const long nrows = (long) d->B * d->Z * d->Y;
  const int nblk = wgrad_blocks(d);
  const int rows_per_block = (int) ((nrows + nblk - 1) / nblk);
  const unsigned grid = (unsigned) ((nrows + rows_per_block - 1) / rows_per_block);
  const int XS = wgrad_xs(d->X);
  const size_t lds = std::max((size_t) (d->cin + d->cout) * XS, (size_t) 2 * 27 * 256) * sizeof(float);
  VAMP_REQUIRE(lds <= 160 * 1024, "row too long for the LDS images (X <= about 600 at 32 + 32 channels)");
  float* part = static_cast<float*>(workspace);
#define VAMP_WGRAD(CI, CO)                                                                          \
  do {                                                                                              \
    static bool attr_set = false;                                                                   \
    if (!attr_set) {                                                                                \
      (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_kernel<CI, CO>),       \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
      attr_set = true;                                                                              \
    }                                                                                               \
    VAMP_TIMED(kProfConvWgrad, s, (conv3d_wgrad_kernel<CI, CO><<<grid, 256, lds, s>>>(              \
        P, in, grad_out, part, nrows, rows_per_block, XS)));                                        \
  } while (0)
  if (d->cin == 16 && d->cout == 16) VAMP_WGRAD(16, 16);
  else if (d->cin == 16 && d->cout == 32) VAMP_WGRAD(16, 32);
  else if (d->cin == 32 && d->cout == 16) VAMP_WGRAD(32, 16);
  else VAMP_WGRAD(32, 32);
#undef VAMP_WGRAD
  if (int e = check_launch("conv3d_wgrad_kernel")) return e;
  const int n = 27 * d->cout * d->cin;
  conv3d_wgrad_reduce_kernel<<<(n + 255) / 256, 256, 0, s>>>(part, grad_weight, (int) grid, d->cin, d->cout);
  return check_launch("conv3d_wgrad_reduce_kernel");
}

}  // extern "C"
