// 3x3x3 / stride 1 / pad 1 Conv3d of the UNet in bf16 (SURVEY 8f N3 under the reference's own
// `precision=16`, base_cli.py:77: under autocast the layers of base_vampire2.py:20, 40-60 see bf16
// activations and weights), gfx950: bf16 in, fp32 accumulate on the matrix cores
// (v_mfma_f32_16x16x32_bf16), bf16 out; NCDHW tensors as autocast hands them over, weight
// [cout, cin, 3, 3, 3] bf16.
//
// The bf16 MFMA wants 8 consecutive K per lane, and K = (tap, input channel) -- but NCDHW keeps a
// voxel's channels a whole volume apart.  So a workgroup stages its input tile (1 z x 4 y x 64 x
// outputs -> 3 x 6 x 68 voxels) in LDS CHANNEL-INNER: every thread loads x-pairs of 8 channel
// planes (4-byte loads through a buffer descriptor: voxels outside the volume read 0), packs them
// with v_perm into the two voxels' 16-byte channel vectors and writes those with ds_write_b128.
// Then   out^T [cout x voxels] = W [cout x K] . IN [K x voxels]:
//   A = weights, LDS image [k-step][cout][32 k] (one ds_read_b128 per fragment),
//   B = the tile: lane (voxel n, k-group) reads the 16 bytes of (voxel + tap, 8 channels),
//   cin = 16: a k-step is two taps x 16 channels (14 steps, the 28th tap has zero weights);
//   cin = 32: one tap x 32 channels (27 steps).
// D: row (cout) = 4 (lane >> 4) + reg, column (voxel) = lane & 15 -> 32-byte rows of the NCDHW output.
// The data gradient is the same kernel on the flipped, transposed weights (FLIP).
// The weight gradient has K = voxels, contiguous along x in NCDHW: both operands are plain
// 16-byte global loads (the shifted-by-one-voxel windows at 2-byte alignment), no LDS at all.
#include "common.hpp"

#include <algorithm>
#include <type_traits>

namespace vamp {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 __attribute__((aligned(2))) u32x4_u;     // a 16-byte global load at 2-byte alignment
typedef bf16x8 __attribute__((aligned(2))) bf16x8_u;

struct CvP {
  int B, Z, Y, X;
  int cin, cout;        // real channel counts of the kernel's input / output tensors (<= the template tile sizes)
};

constexpr int kTY = 4;        // output rows per tile = waves per workgroup
constexpr int kTX = 64;       // output x per tile (four 16-voxel N tiles per wave)
constexpr int kHX = 68;       // staged x: x0 - 2 .. x0 + 65 (pairs at even x)
constexpr int kRows = 3 * (kTY + 2);

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// the 16-bit matrix-core product: bf16 or (F16) IEEE half operands, fp32 accumulate; same fragment maps
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16) {
    f16x8 ha, hb;
    __builtin_memcpy(&ha, &a, 16);
    __builtin_memcpy(&hb, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
}

__device__ __forceinline__ unsigned short f2bf(float f) {          // round to nearest even (NaN kept quiet)
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short) ((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short) (u >> 16);
}

template <bool F16>
__device__ __forceinline__ unsigned short f2e(float f) {              // fp32 -> the 16-bit element type, round to nearest even
  if constexpr (F16) {
    const _Float16 h = (_Float16) f;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
  } else {
    return f2bf(f);
  }
}

template <int CIN, int COUT>
struct CvLds {
  static constexpr int NTP = CIN == 16 ? 14 : 27;                  // k-steps
  static constexpr int VS = CIN * 2 + 16;                          // bytes per staged voxel (padded: bank spread)
  static constexpr size_t w_bytes = (size_t) NTP * COUT * 64;
  static constexpr size_t in_bytes = (size_t) kRows * kHX * VS;
  static constexpr size_t bytes = w_bytes + in_bytes;
};

template <int CIN, int COUT, bool FLIP, bool F16>
__global__ void __launch_bounds__(256)
conv3d_bf16_fwd_kernel(CvP P, const unsigned short* __restrict__ in, const unsigned short* __restrict__ w,
                       unsigned short* __restrict__ out, int tiles_x, long ntiles) {
  using L = CvLds<CIN, COUT>;
  constexpr int NTP = L::NTP, VS = L::VS, MT = COUT / 16, CG = CIN / 8;
  constexpr int NITEM = kRows * (kHX / 2) * CG;                    // (row, x pair, 8-channel group)
  constexpr int NI = (NITEM + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* w_s = smem;                                       // [NTP][COUT][32] bf16
  unsigned char* in_s = smem + L::w_bytes;                         // [kRows][kHX] voxels of VS bytes
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const long plane = (long) P.Z * P.Y * P.X;

  // weight image: w_s[tp][co][k], k -> (tap, ci)
  for (int e = tid; e < NTP * COUT * 32; e += 256) {
    const int k = e & 31, co = (e >> 5) % COUT, tp = e / (32 * COUT);
    const int tap = CIN == 16 ? 2 * tp + (k >> 4) : tp, ci = CIN == 16 ? (k & 15) : k;
    unsigned short v = 0;
    if (tap < 27 && ci < P.cin && co < P.cout)       // channels beyond the real counts: zero weights
      v = FLIP ? w[((long) ci * P.cout + co) * 27 + (26 - tap)] : w[((long) co * P.cin + ci) * 27 + tap];
    reinterpret_cast<unsigned short*>(w_s)[e] = v;
  }

  // staging: this thread's items of a tile, registers first (the next tile's loads fly during the MFMAs)
  unsigned st[NI][8];
  auto fetch = [&](long tile) {
    const int tx = (int) (tile % tiles_x);
    long r = tile / tiles_x;
    const int ty = (int) (r % ((P.Y + kTY - 1) / kTY));
    r /= ((P.Y + kTY - 1) / kTY);
    const int z = (int) (r % P.Z), b = (int) (r / P.Z);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(in + (long) b * P.cin * plane), 0, (int) ((size_t) P.cin * plane * 2), 0x00020000);   // planes >= cin read 0
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int it = tid + i * 256;
      const int px = it % (kHX / 2), rr = (it / (kHX / 2)) % kRows, cg = it / ((kHX / 2) * kRows);
      const int zz = z + rr / (kTY + 2) - 1, yy = ty * kTY + rr % (kTY + 2) - 1, xx = tx * kTX - 2 + 2 * px;
      const bool ok = it < NITEM && zz >= 0 && zz < P.Z && yy >= 0 && yy < P.Y && xx >= 0 && xx < P.X;
      const unsigned voff = ok ? (unsigned) ((((long) cg * 8 * P.Z + zz) * P.Y + yy) * P.X + xx) * 2u : 0xfffffff0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) st[i][j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (unsigned) (j * plane * 2), 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int it = tid + i * 256;
      if (it < NITEM) {
        const int px = it % (kHX / 2), rr = (it / (kHX / 2)) % kRows, cg = it / ((kHX / 2) * kRows);
        u32x4 lo, hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          lo[q] = __builtin_amdgcn_perm(st[i][2 * q + 1], st[i][2 * q], 0x05040100u);   // the even-x voxel: low halves
          hi[q] = __builtin_amdgcn_perm(st[i][2 * q + 1], st[i][2 * q], 0x07060302u);   // the odd-x voxel: high halves
        }
        unsigned char* p = in_s + ((size_t) rr * kHX + 2 * px) * VS + cg * 16;
        *reinterpret_cast<u32x4*>(p) = lo;
        *reinterpret_cast<u32x4*>(p + VS) = hi;
      }
    }
  };

  long tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                           // the previous tile's readers (and the weight image)
    commit();
    __syncthreads();
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);

    const int tx = (int) (tile % tiles_x);
    long r = tile / tiles_x;
    const int ty = (int) (r % ((P.Y + kTY - 1) / kTY));
    r /= ((P.Y + kTY - 1) / kTY);
    const int z = (int) (r % P.Z), b = (int) (r / P.Z);
    const int y = ty * kTY + wv, x0 = tx * kTX;
    const int nnt = min(4, (P.X - x0 + 15) / 16);       // N tiles of this wave inside the volume (uniform)
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // NN = N tiles computed, a compile-time count: no branch between the LDS reads and the MFMAs of a
    // k-step, so hipcc issues the step's reads as a batch (with a per-tile `if` every MFMA waited for
    // its own ds_read: 73 us instead of 30 at 16 -> 16, 16 x 200 x 200)
    auto run = [&](auto nn_tag) {
      constexpr int NN = decltype(nn_tag)::value;
#pragma unroll
      for (int tp = 0; tp < NTP; ++tp) {
        // this lane's tap of the k-step (cin = 16: lanes 32..63 take the second tap) and channel chunk
        int toff;
        if (CIN == 16) {
          const int t0 = 2 * tp, t1 = min(2 * tp + 1, 26);
          const int o0 = ((t0 / 9) * (kTY + 2) + (t0 / 3) % 3) * kHX + t0 % 3;
          const int o1 = ((t1 / 9) * (kTY + 2) + (t1 / 3) % 3) * kHX + t1 % 3;
          toff = ((kg >> 1) ? o1 : o0) * VS + (kg & 1) * 16;
        } else {
          toff = (((tp / 9) * (kTY + 2) + (tp / 3) % 3) * kHX + tp % 3) * VS + kg * 16;
        }
        bf16x8 a[MT], bv[NN];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          a[m] = *reinterpret_cast<const bf16x8*>(w_s + ((size_t) (tp * COUT + m * 16 + li) * 64 + kg * 16));
        // voxel (row wv, x0 + 16 n + li) + tap: staged x index = (x - x0 + 2) + (dx - 1)
#pragma unroll
        for (int n = 0; n < NN; ++n)
          bv[n] = *reinterpret_cast<const bf16x8*>(in_s + (size_t) (wv * kHX + n * 16 + li + 1) * VS + toff);
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = mfma16<F16>(a[m], bv[n], acc[m][n]);
      }
    };
    if (nnt == 4) run(std::integral_constant<int, 4>());
    else if (nnt == 3) run(std::integral_constant<int, 3>());
    else if (nnt == 2) run(std::integral_constant<int, 2>());
    else run(std::integral_constant<int, 1>());
    if (y < P.Y) {
      unsigned short* ob = out + (long) b * P.cout * plane + ((long) z * P.Y + y) * P.X;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const int x = x0 + n * 16 + li;
          if (x < P.X) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
              if (m * 16 + 4 * kg + rg < P.cout) ob[(long) (m * 16 + 4 * kg + rg) * plane + x] = f2e<F16>(acc[m][n][rg]);
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient: dw[co][ci][t] = sum_voxels dout[co][v] * in[ci][v + tap t]
// M = co, N = ci, K = 32 consecutive x of one (b, z, y) row per MFMA.  A = 16 bytes of dout at
// (co, x0 + 8 kg), B = 16 bytes of in at (ci, x0 + 8 kg + dx) of row (z + dz, y + dy): plain global
// loads (L1 serves the 27 shifted windows).  A wave owns a subset of the 27 taps (wave w: taps w,
// w + 4, ...: 7 accumulator sets) and walks the rows its workgroup is given; per-workgroup partial
// sums [27][cout][cin] are added by a second kernel.
// ---------------------------------------------------------------------------
template <int CIN, int COUT, bool X8, bool F16>       // X8: X % 8 == 0, every window is one aligned 16-byte load; else X % 4 == 0, two 8-byte halves
__global__ void __launch_bounds__(192)
conv3d_bf16_wgrad_kernel(CvP P, const unsigned short* __restrict__ in, const unsigned short* __restrict__ dout,
                         float* __restrict__ part, long nrows) {
  constexpr int MT = COUT / 16, NT = CIN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;     // wave = dz + 1: nine taps (dy, dx) each
  const int li = lane & 15, kg = lane >> 4;
  const long plane = (long) P.Z * P.Y * P.X;
  f32x4 acc[9][MT][NT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ksteps = (P.X + 31) / 32;
  // One step = 32 voxels of one output row.  Per step a wave loads the dout fragment and ONE aligned
  // 16-byte window per (dy, input-channel tile); the windows shifted by one voxel (dx = -1 / +1) are made
  // in registers from the window and the neighbouring lane group's edge dword (ds_bpermute; the previous /
  // next step supplies the edge of lane groups 0 / 3).  (27 separate windows, 18 of them at 2-byte
  // alignment, cost 214 us at 16 -> 16, 16 x 200 x 200: a misaligned 16-byte global load is served at
  // less than half the rate.)  All loads are unconditional, from rows clamped into the volume; the
  // next step's loads are issued before this step's MFMAs; padding is applied to registers afterwards.
  struct Step {
    bf16x8 a[MT];
    u32x4 v[3][NT];
    int z, y, xb;
  };
  auto load = [&](Step& S, long step) {
    const long row = blockIdx.x + (step / ksteps) * (long) gridDim.x;
    const int ks = (int) (step % ksteps);
    S.y = (int) (row % P.Y);
    S.z = (int) ((row / P.Y) % P.Z);
    const int b = (int) (row / ((long) P.Y * P.Z));
    S.xb = ks * 32 + 8 * kg;
    const unsigned short* gb = dout + (long) b * P.cout * plane + ((long) S.z * P.Y + S.y) * P.X;
    const unsigned short* ib = in + (long) b * P.cin * plane;
    const int zz = min(max(S.z + wv - 1, 0), P.Z - 1);
    // the window [xb, xb + 8): one 16-byte load, or (rows that are only 8-byte aligned, last window half
    // outside) its two halves from starts clamped into the row -- the compute step zeroes what lies outside
    const int xc = min(S.xb, P.X - (X8 ? 8 : 4)), xh = min(S.xb + 4, P.X - 4);
    auto window = [&](const unsigned short* rowp) {
      if (X8) return *reinterpret_cast<const u32x4*>(rowp + xc);
      const u32x2 lo = *reinterpret_cast<const u32x2*>(rowp + xc), hi = *reinterpret_cast<const u32x2*>(rowp + xh);
      return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const u32x4 t = window(gb + (long) min(m * 16 + li, P.cout - 1) * plane);      // rows beyond cout: clamped here, zeroed in compute
      __builtin_memcpy(&S.a[m], &t, 16);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = min(max(S.y + r - 1, 0), P.Y - 1);
#pragma unroll
      for (int n = 0; n < NT; ++n) S.v[r][n] = window(ib + (((long) min(n * 16 + li, P.cin - 1) * P.Z + zz) * P.Y + yy) * P.X);
    }
  };
  // zero the part of a window that lies beyond the row end (whole window, or its upper half when X % 8 == 4)
  auto clip = [&](const u32x4& v, int xb) {
    u32x4 r = xb < P.X ? v : u32x4{0u, 0u, 0u, 0u};
    if (!X8 && xb + 4 >= P.X) { r[2] = 0u; r[3] = 0u; }
    return r;
  };
  auto as_frag = [](const u32x4& v) {
    bf16x8 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
  };
  // prev3[r][n]: dword 3 of the previous step's windows (the voxels just left of this step's lane group 0)
  u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
  unsigned prev3[3][NT];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) prev3[r][n] = 0u;
  auto compute = [&](const Step& S, const Step& N) {
    const bool in_row = S.xb < P.X;
    bf16x8 a[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      u32x4 t;
      __builtin_memcpy(&t, &S.a[m], 16);
      t = clip(t, S.xb);
      if (m * 16 + li >= P.cout) t = zero4;
      __builtin_memcpy(&a[m], &t, 16);
    }
    const int zz = S.z + wv - 1;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = S.y + r - 1;
      const bool row_ok = zz >= 0 && zz < P.Z && yy >= 0 && yy < P.Y;       // wave-uniform
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const u32x4 v = (row_ok && n * 16 + li < P.cin) ? clip(S.v[r][n], S.xb) : zero4;
        // the dword left of this lane's window: lane group kg - 1 (same step), or the previous step's group 3
        const unsigned lsend = kg == 3 ? prev3[r][n] : v[3];
        unsigned left = (unsigned) __shfl((int) lsend, (lane + 48) & 63, 64);
        // the dword right of it: lane group kg + 1, or the next step's group 0
        const unsigned rsend = kg == 0 ? N.v[r][n][0] : v[0];
        unsigned right = (unsigned) __shfl((int) rsend, (lane + 16) & 63, 64);
        if (S.xb == 0 || !row_ok) left = 0u;                        // the row's first voxel has no left neighbour
        if (S.xb + 8 >= P.X || !row_ok) right = 0u;                 // ... its last no right neighbour
        const u32x4 vl = u32x4{__builtin_amdgcn_alignbit(v[0], left, 16), __builtin_amdgcn_alignbit(v[1], v[0], 16),
                               __builtin_amdgcn_alignbit(v[2], v[1], 16), __builtin_amdgcn_alignbit(v[3], v[2], 16)};
        const u32x4 vr = u32x4{__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                               __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(right, v[3], 16)};
        const bf16x8 b0 = as_frag(in_row ? vl : zero4), b1 = as_frag(v), b2 = as_frag(in_row ? vr : zero4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[r * 3 + 0][m][n] = mfma16<F16>(a[m], b0, acc[r * 3 + 0][m][n]);
          acc[r * 3 + 1][m][n] = mfma16<F16>(a[m], b1, acc[r * 3 + 1][m][n]);
          acc[r * 3 + 2][m][n] = mfma16<F16>(a[m], b2, acc[r * 3 + 2][m][n]);
        }
        prev3[r][n] = v[3];
      }
    }
  };
  const long my_rows = blockIdx.x < nrows ? (nrows - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const long nsteps = my_rows * ksteps;
  Step cur, nxt;
  if (nsteps > 0) load(cur, 0);
  for (long st = 0; st < nsteps; ++st) {
    load(nxt, min(st + 1, nsteps - 1));
    compute(cur, nxt);
    cur = nxt;
  }
  // partial sums: [workgroup][tap][co][ci]; D row (co) = 4 kg + reg, column (ci) = li
  float* pb = part + (size_t) blockIdx.x * 27 * COUT * CIN;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int tap = wv * 9 + t;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) pb[((size_t) tap * COUT + m * 16 + 4 * kg + rg) * CIN + n * 16 + li] = acc[t][m][n][rg];
  }
}

// dw[co][ci][tap] (bf16 or fp32 out) = sum over workgroups of part[wg][tap][co][ci]
__global__ void __launch_bounds__(256)
conv3d_bf16_wreduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int cin, int cout, int nwg,
                           int cin_r, int cout_r) {
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
  __shared__ float red[4][64];
  const int n = 27 * cout * cin;
  float s = 0.f;
  if (e < n) {
    int g = wv;
    for (; g + 28 < nwg; g += 32) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = part[(size_t) (g + 4 * i) * n + e];
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; g < nwg; g += 4) s += part[(size_t) g * n + e];
  }
  red[wv][threadIdx.x & 63] = s;
  __syncthreads();
  if (wv == 0 && e < n) {
    s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const int ci = e % cin, co = (e / cin) % cout, tap = e / (cin * cout);
    if (co < cout_r && ci < cin_r) dw[((size_t) co * cin_r + ci) * 27 + tap] = s;
  }
}

constexpr int kWgradWgs = 512;       // two workgroups per CU

int tile_ch(int c) { return c <= 16 ? 16 : 32; }       // channel tile a real channel count runs under

bool bf16_shape_ok(const VampConvDesc* d) {
  return d && d->cin >= 1 && d->cin <= 32 && d->cout >= 1 && d->cout <= 32 && d->B > 0 && d->Z > 0 &&
         d->Y > 0 && d->X >= 8 && d->X % 4 == 0 &&
         (long) d->Z * d->Y * d->X * 32 * 2 < 0x7fffffffL;
}

template <int CIN, int COUT, bool FLIP, bool F16>
int launch_fwd(const VampConvDesc* d, const void* in, const void* w, void* out, hipStream_t s) {
  using L = CvLds<CIN, COUT>;
  auto k = conv3d_bf16_fwd_kernel<CIN, COUT, FLIP, F16>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) L::bytes) != hipSuccess)
    return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);
  const CvP P{d->B, d->Z, d->Y, d->X, FLIP ? d->cout : d->cin, FLIP ? d->cin : d->cout};
  const int tiles_x = (d->X + kTX - 1) / kTX;
  const long ntiles = (long) d->B * d->Z * ((d->Y + kTY - 1) / kTY) * tiles_x;
  const int per_cu = L::bytes <= 78 * 1024 ? 2 : 1;
  const unsigned grid = (unsigned) std::min<long>(ntiles, 256 * per_cu);
  VAMP_TIMED(FLIP ? kProfConvDgrad : kProfConvFwd, s, (k<<<grid, 256, L::bytes, s>>>(
      P, static_cast<const unsigned short*>(in), static_cast<const unsigned short*>(w), static_cast<unsigned short*>(out), tiles_x, ntiles)));
  return check_launch("conv3d_bf16_fwd_kernel");
}

template <int CIN, int COUT, bool F16>
int launch_wgrad(const VampConvDesc* d, const void* in, const void* dout, float* dw, float* ws, hipStream_t s) {
  const CvP P{d->B, d->Z, d->Y, d->X, d->cin, d->cout};
  const long nrows = (long) d->B * d->Z * d->Y;
  const int nwg = (int) std::min<long>(nrows, kWgradWgs);
  if (d->X % 8 == 0)
    VAMP_TIMED(kProfConvWgrad, s, (conv3d_bf16_wgrad_kernel<CIN, COUT, true, F16><<<nwg, 192, 0, s>>>(
        P, static_cast<const unsigned short*>(in), static_cast<const unsigned short*>(dout), ws, nrows)));
  else
    VAMP_TIMED(kProfConvWgrad, s, (conv3d_bf16_wgrad_kernel<CIN, COUT, false, F16><<<nwg, 192, 0, s>>>(
        P, static_cast<const unsigned short*>(in), static_cast<const unsigned short*>(dout), ws, nrows)));
  if (int e = check_launch("conv3d_bf16_wgrad_kernel")) return e;
  const int n = 27 * CIN * COUT;
  VAMP_TIMED(kProfConvWgrad, s, (conv3d_bf16_wreduce_kernel<<<(n + 63) / 64, 256, 0, s>>>(ws, dw, CIN, COUT, nwg, d->cin, d->cout)));
  return check_launch("conv3d_bf16_wreduce_kernel");
}

}  // namespace
}  // namespace vamp

using namespace vamp;

extern "C" {

int vamp_conv3d_bf16_supported(const VampConvDesc* d) { return bf16_shape_ok(d) ? 1 : 0; }

size_t vamp_conv3d_bf16_workspace_bytes(const VampConvDesc* d) {
  if (!bf16_shape_ok(d)) return 0;
  return (size_t) kWgradWgs * 27 * tile_ch(d->cin) * tile_ch(d->cout) * sizeof(float);
}

// dtype: VAMP_BF16 or VAMP_F16 (the reference's `precision=16` is IEEE half)
int vamp_conv3d_half_forward(const VampConvDesc* d, int32_t dtype, const void* in, const void* weight, void* out,
                             void* stream) {
  VAMP_REQUIRE(bf16_shape_ok(d), "unsupported shape: see vamp_conv3d_bf16_supported");
  VAMP_REQUIRE(in && weight && out, "NULL tensor");
  VAMP_REQUIRE(dtype == VAMP_BF16 || dtype == VAMP_F16, "dtype must be VAMP_BF16 or VAMP_F16");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define VAMP_CV(F16)                                                                                       \
  if (tile_ch(d->cin) == 16) return tile_ch(d->cout) == 16 ? launch_fwd<16, 16, false, F16>(d, in, weight, out, s)           \
                                         : launch_fwd<16, 32, false, F16>(d, in, weight, out, s);          \
  return tile_ch(d->cout) == 16 ? launch_fwd<32, 16, false, F16>(d, in, weight, out, s)                             \
                       : launch_fwd<32, 32, false, F16>(d, in, weight, out, s);
  if (dtype == VAMP_F16) { VAMP_CV(true) }
  VAMP_CV(false)
#undef VAMP_CV
}

int vamp_conv3d_half_backward_data(const VampConvDesc* d, int32_t dtype, const void* grad_out, const void* weight,
                                   void* grad_in, void* stream) {
  VAMP_REQUIRE(bf16_shape_ok(d), "unsupported shape: see vamp_conv3d_bf16_supported");
  VAMP_REQUIRE(grad_out && weight && grad_in, "NULL tensor");
  VAMP_REQUIRE(dtype == VAMP_BF16 || dtype == VAMP_F16, "dtype must be VAMP_BF16 or VAMP_F16");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the kernel's input tensor is grad_out (cout channels), its output grad_in (cin channels)
#define VAMP_CV(F16)                                                                                       \
  if (tile_ch(d->cout) == 16) return tile_ch(d->cin) == 16 ? launch_fwd<16, 16, true, F16>(d, grad_out, weight, grad_in, s)  \
                                         : launch_fwd<16, 32, true, F16>(d, grad_out, weight, grad_in, s); \
  return tile_ch(d->cin) == 16 ? launch_fwd<32, 16, true, F16>(d, grad_out, weight, grad_in, s)                     \
                      : launch_fwd<32, 32, true, F16>(d, grad_out, weight, grad_in, s);
  if (dtype == VAMP_F16) { VAMP_CV(true) }
  VAMP_CV(false)
#undef VAMP_CV
}

int vamp_conv3d_half_backward_weight(const VampConvDesc* d, int32_t dtype, const void* in, const void* grad_out,
                                     float* grad_weight, void* workspace, size_t workspace_bytes, void* stream) {
  VAMP_REQUIRE(bf16_shape_ok(d), "unsupported shape: see vamp_conv3d_bf16_supported");
  VAMP_REQUIRE(in && grad_out && grad_weight, "NULL tensor");
  VAMP_REQUIRE(dtype == VAMP_BF16 || dtype == VAMP_F16, "dtype must be VAMP_BF16 or VAMP_F16");
  const size_t need = vamp_conv3d_bf16_workspace_bytes(d);
  if (!workspace || workspace_bytes < need)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
#define VAMP_CV(F16)                                                                                       \
  if (tile_ch(d->cin) == 16) return tile_ch(d->cout) == 16 ? launch_wgrad<16, 16, F16>(d, in, grad_out, grad_weight, ws, s)  \
                                         : launch_wgrad<16, 32, F16>(d, in, grad_out, grad_weight, ws, s); \
  return tile_ch(d->cout) == 16 ? launch_wgrad<32, 16, F16>(d, in, grad_out, grad_weight, ws, s)                    \
                       : launch_wgrad<32, 32, F16>(d, in, grad_out, grad_weight, ws, s);
  if (dtype == VAMP_F16) { VAMP_CV(true) }
  VAMP_CV(false)
#undef VAMP_CV
}

int vamp_conv3d_bf16_forward(const VampConvDesc* d, const void* in, const void* weight, void* out, void* stream) {
  return vamp_conv3d_half_forward(d, VAMP_BF16, in, weight, out, stream);
}
int vamp_conv3d_bf16_backward_data(const VampConvDesc* d, const void* grad_out, const void* weight, void* grad_in,
                                   void* stream) {
  return vamp_conv3d_half_backward_data(d, VAMP_BF16, grad_out, weight, grad_in, stream);
}
int vamp_conv3d_bf16_backward_weight(const VampConvDesc* d, const void* in, const void* grad_out, float* grad_weight,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  return vamp_conv3d_half_backward_weight(d, VAMP_BF16, in, grad_out, grad_weight, workspace, workspace_bytes, stream);
}

}  // extern "C"
