// Trilinear resize of the 3-D UNet between lift and render (first piece), gfx950:
// `F.interpolate(x, size, mode='trilinear', align_corners=True)`, base_vampire2.py:66, 72 (twice
// per Hourglass3D), forward and backward.
//
// aten's backward scatters every output gradient into its 8 source voxels with float atomics
// (2.2 ms per call for the [32, 16, 200, 200] level on MI355X: four calls are 8.8 ms of the
// module's 23 ms backward).  Here the backward is a gather: the outputs that touch source index
// i along one axis form one short contiguous range (source coordinate = o * (in-1)/(out-1), taps
// floor and floor + 1), so a tiny per-axis table holds {first output, count, weights} per source
// index and each source voxel sums its nz * ny * nx contributions (about 4^3 for a x2 resize),
// coalesced along x, every gradient element stored once, no atomics.
//
// Index / weight arithmetic follows aten (area_pixel_compute_source_index with
// align_corners=True, UpSampleTrilinear3d.cu): scale = (in - 1) / (out - 1) in fp32,
// src = scale * o, i0 = (int) src, i1 = i0 + (i0 < in - 1), lambda1 = src - i0, lambda0 = 1 - lambda1.
// HBM-bound streaming: no MFMA.
#include "common.hpp"

namespace vamp {
namespace {

constexpr int kMaxHits = 14;           // outputs per source index and axis the table can hold

struct AxisHit {                       // 64 bytes
  int begin, n;
  float w[kMaxHits];
};

__device__ __forceinline__ float axis_scale(int in, int out) {
  return out > 1 ? (float) (in - 1) / (float) (out - 1) : 0.f;
}

// element types of the tensors: fp32, or the 16-bit types a mixed-precision UNet hands over (compute stays fp32)
struct Bf16 { unsigned short u; };
struct F16 { unsigned short u; };
__device__ __forceinline__ float lde(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float lde(const Bf16* p, long i) { return __uint_as_float((unsigned) p[i].u << 16); }
__device__ __forceinline__ float lde(const F16* p, long i) {
  _Float16 h;
  __builtin_memcpy(&h, &p[i].u, 2);
  return (float) h;
}
__device__ __forceinline__ void ste(float* p, long i, float v) { p[i] = v; }
__device__ __forceinline__ void ste(Bf16* p, long i, float v) {
  unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x400000u; else u += 0x7fffu + ((u >> 16) & 1u);
  p[i].u = (unsigned short) (u >> 16);
}
__device__ __forceinline__ void ste(F16* p, long i, float v) {
  const _Float16 h = (_Float16) v;
  __builtin_memcpy(&p[i].u, &h, 2);
}

struct Tap {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Tap axis_tap(float scale, int o, int in) {
  const float src = scale * (float) o;
  Tap t;
  t.i0 = (int) src;
  t.i1 = t.i0 + ((t.i0 < in - 1) ? 1 : 0);
  t.l1 = src - (float) t.i0;
  t.l0 = 1.0f - t.l1;
  return t;
}

// one thread per (axis, source index): the contiguous run of outputs whose taps include it
__global__ void __launch_bounds__(256)
upsample_axis_table_kernel(AxisHit* __restrict__ tab, int iz, int iy, int ix, int oz, int oy, int ox) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= iz + iy + ix) return;
  const int axis = t < iz ? 0 : (t < iz + iy ? 1 : 2);
  const int i = axis == 0 ? t : (axis == 1 ? t - iz : t - iz - iy);
  const int in = axis == 0 ? iz : (axis == 1 ? iy : ix);
  const int out = axis == 0 ? oz : (axis == 1 ? oy : ox);
  const float scale = axis_scale(in, out);
  // first candidate: a little before (i - 1) / scale; the run ends when floor(src) > i
  int o = 0;
  if (scale > 0.f) o = max(0, (int) ((float) (i - 1) / scale) - 2);
  AxisHit h;
  h.begin = 0; h.n = 0;
#pragma unroll
  for (int k = 0; k < kMaxHits; ++k) h.w[k] = 0.f;
  for (; o < out; ++o) {
    const Tap tp = axis_tap(scale, o, in);
    if (tp.i0 > i) break;
    if (tp.i0 != i && tp.i1 != i) continue;
    const float w = (tp.i0 == i ? tp.l0 : 0.f) + (tp.i1 == i ? tp.l1 : 0.f);
    if (h.n == 0) h.begin = o;
    // (no dynamic register indexing: select the slot; the host has checked that the run fits)
#pragma unroll
    for (int k = 0; k < kMaxHits; ++k)
      if (k == h.n) h.w[k] = w;
    if (h.n < kMaxHits) ++h.n;
  }
  tab[t] = h;
}

constexpr int kPlanesPerBlock = 8;     // (batch, channel) planes a thread walks with one set of taps
constexpr int kBwdPlanes = 1;          // the backward has 64 loads per plane already: parallelism wins

// thread per output voxel, kPlanesPerBlock planes per thread (the taps are computed once)
template <typename T>
__global__ void __launch_bounds__(256)
upsample_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, int iz, int iy, int ix,
                    int oz, int oy, int ox, int planes) {
  const unsigned ovox = (unsigned) oz * oy * ox, ivox = (unsigned) iz * iy * ix;
  const unsigned o = blockIdx.x * 256u + threadIdx.x;
  if (o >= ovox) return;
  const unsigned x = o % (unsigned) ox, yz = o / (unsigned) ox;
  const unsigned y = yz % (unsigned) oy, z = yz / (unsigned) oy;
  const Tap tz = axis_tap(axis_scale(iz, oz), (int) z, iz);
  const Tap ty = axis_tap(axis_scale(iy, oy), (int) y, iy);
  const Tap tx = axis_tap(axis_scale(ix, ox), (int) x, ix);
  const unsigned r00 = ((unsigned) tz.i0 * iy + ty.i0) * ix, r01 = ((unsigned) tz.i0 * iy + ty.i1) * ix;
  const unsigned r10 = ((unsigned) tz.i1 * iy + ty.i0) * ix, r11 = ((unsigned) tz.i1 * iy + ty.i1) * ix;
  const int p0 = blockIdx.y * kPlanesPerBlock, p1 = min(planes, p0 + kPlanesPerBlock);
  for (int pl = p0; pl < p1; ++pl) {
    const T* p = in + (long) pl * ivox;
    const float v000 = lde(p, r00 + tx.i0), v001 = lde(p, r00 + tx.i1), v010 = lde(p, r01 + tx.i0), v011 = lde(p, r01 + tx.i1);
    const float v100 = lde(p, r10 + tx.i0), v101 = lde(p, r10 + tx.i1), v110 = lde(p, r11 + tx.i0), v111 = lde(p, r11 + tx.i1);
    // aten's nesting (UpSampleTrilinear3d.cu)
    ste(out, (long) pl * ovox + o,
        tz.l0 * (ty.l0 * (tx.l0 * v000 + tx.l1 * v001) + ty.l1 * (tx.l0 * v010 + tx.l1 * v011)) +
        tz.l1 * (ty.l0 * (tx.l0 * v100 + tx.l1 * v101) + ty.l1 * (tx.l0 * v110 + tx.l1 * v111)));
  }
}

template <int MH>
__device__ __forceinline__ float hit_weight(const AxisHit& h, int k) {
  float w = 0.f;
#pragma unroll
  for (int q = 0; q < MH; ++q)
    if (q == k) w = h.w[q];
  return w;
}

// thread per source voxel: sum of the (at most nz * ny * nx) output gradients, kBwdPlanes
// planes per thread; MH = hits per axis the x-loop is unrolled for
template <int MH, typename T>
__global__ void __launch_bounds__(256)
upsample_bwd_kernel(const T* __restrict__ g, T* __restrict__ gin, const AxisHit* __restrict__ tab,
                    int iz, int iy, int ix, int oz, int oy, int ox, int planes) {
  const unsigned ivox = (unsigned) iz * iy * ix;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= ivox) return;
  const unsigned x = i % (unsigned) ix, yz = i / (unsigned) ix;
  const unsigned y = yz % (unsigned) iy, z = yz / (unsigned) iy;
  const AxisHit hz = tab[z], hy = tab[iz + y], hx = tab[iz + iy + x];
  const long ovox = (long) oz * oy * ox;
  const int p0 = blockIdx.y * kBwdPlanes, p1 = min(planes, p0 + kBwdPlanes);
  for (int pl = p0; pl < p1; ++pl) {
    const T* gp = g + (long) pl * ovox;
    float acc = 0.f;
    for (int a = 0; a < hz.n; ++a) {
      const float wz = hit_weight<MH>(hz, a);
      for (int b = 0; b < hy.n; ++b) {
        const float wzy = wz * hit_weight<MH>(hy, b);
        const T* row = gp + ((unsigned) (hz.begin + a) * oy + (hy.begin + b)) * (unsigned) ox + hx.begin;
#pragma unroll
        for (int c = 0; c < MH; ++c)
          if (c < hx.n) acc += wzy * hx.w[c] * lde(row, c);
      }
    }
    ste(gin, (long) pl * ivox + i, acc);
  }
}

size_t table_bytes(int iz, int iy, int ix) {
  return align_up((size_t) (iz + iy + ix) * sizeof(AxisHit), 256);
}

bool run_fits(int in, int out, int hits = kMaxHits) {
  if (in == 1) return out <= hits;
  return 2L * (out - 1) / (in - 1) + 2 <= hits;
}

int check_dims(int64_t planes, int iz, int iy, int ix, int oz, int oy, int ox) {
  VAMP_REQUIRE(planes > 0 && planes < 65536, "0 < batch * channels < 65536");
  VAMP_REQUIRE(iz > 0 && iy > 0 && ix > 0 && oz > 0 && oy > 0 && ox > 0, "sizes must be positive");
  VAMP_REQUIRE((long) oz * oy * ox < 0x7fffffffL && (long) iz * iy * ix < 0x7fffffffL, "volume too large");
  return VAMP_OK;
}

}  // namespace
}  // namespace vamp

using namespace vamp;

extern "C" {

}  // extern "C"

template <typename T>
static int up_forward(int64_t planes, int iz, int iy, int ix, int oz, int oy, int ox, const void* in, void* out, hipStream_t s) {
  const long ovox = (long) oz * oy * ox;
  const dim3 grid((unsigned) ((ovox + 255) / 256), (unsigned) ((planes + kPlanesPerBlock - 1) / kPlanesPerBlock));
  VAMP_TIMED(kProfUpsample, s, (upsample_fwd_kernel<T><<<grid, 256, 0, s>>>(static_cast<const T*>(in), static_cast<T*>(out),
                                                                           iz, iy, ix, oz, oy, ox, (int) planes)));
  return check_launch("upsample_fwd_kernel");
}

extern "C" {

/* dtype: VAMP_F32, VAMP_BF16 or VAMP_F16 for `in` and `out` alike (fp32 arithmetic) */
int vamp_upsample_trilinear_forward_ex(int64_t planes, int32_t iz, int32_t iy, int32_t ix, int32_t oz,
                                       int32_t oy, int32_t ox, int32_t dtype, const void* in, void* out,
                                       void* stream) {
  if (int e = check_dims(planes, iz, iy, ix, oz, oy, ox)) return e;
  VAMP_REQUIRE(in && out, "NULL tensor");
  VAMP_REQUIRE(dtype == VAMP_F32 || dtype == VAMP_BF16 || dtype == VAMP_F16, "dtype");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VAMP_F32) return up_forward<float>(planes, iz, iy, ix, oz, oy, ox, in, out, s);
  if (dtype == VAMP_BF16) return up_forward<Bf16>(planes, iz, iy, ix, oz, oy, ox, in, out, s);
  return up_forward<F16>(planes, iz, iy, ix, oz, oy, ox, in, out, s);
}

int vamp_upsample_trilinear_forward(int64_t planes, int32_t iz, int32_t iy, int32_t ix, int32_t oz,
                                    int32_t oy, int32_t ox, const float* in, float* out,
                                    void* stream) {
  return vamp_upsample_trilinear_forward_ex(planes, iz, iy, ix, oz, oy, ox, VAMP_F32, in, out, stream);
}

int vamp_upsample_trilinear_supported(int32_t iz, int32_t iy, int32_t ix, int32_t oz, int32_t oy, int32_t ox) {
  if (iz <= 0 || iy <= 0 || ix <= 0 || oz <= 0 || oy <= 0 || ox <= 0) return 0;
  return (run_fits(iz, oz) && run_fits(iy, oy) && run_fits(ix, ox)) ? 1 : 0;
}

size_t vamp_upsample_trilinear_workspace_bytes(int32_t iz, int32_t iy, int32_t ix) {
  if (iz <= 0 || iy <= 0 || ix <= 0) return 0;
  return table_bytes(iz, iy, ix);
}

}  // extern "C"

template <typename T>
static int up_backward(int64_t planes, int iz, int iy, int ix, int oz, int oy, int ox, const void* grad_out, void* grad_in,
                       AxisHit* tab, hipStream_t s) {
  const long ivox = (long) iz * iy * ix;
  const dim3 grid((unsigned) ((ivox + 255) / 256), (unsigned) ((planes + kBwdPlanes - 1) / kBwdPlanes));
  const T* g = static_cast<const T*>(grad_out);
  T* gi = static_cast<T*>(grad_in);
  if (run_fits(ix, ox, 6) && run_fits(iy, oy, 6) && run_fits(iz, oz, 6)) {
    VAMP_TIMED(kProfUpsample, s, (upsample_bwd_kernel<6, T><<<grid, 256, 0, s>>>(g, gi, tab, iz, iy, ix, oz, oy, ox, (int) planes)));
  } else {
    VAMP_TIMED(kProfUpsample, s, (upsample_bwd_kernel<kMaxHits, T><<<grid, 256, 0, s>>>(g, gi, tab, iz, iy, ix, oz, oy, ox, (int) planes)));
  }
  return check_launch("upsample_bwd_kernel");
}

extern "C" {

int vamp_upsample_trilinear_backward_ex(int64_t planes, int32_t iz, int32_t iy, int32_t ix, int32_t oz,
                                        int32_t oy, int32_t ox, int32_t dtype, const void* grad_out, void* grad_in,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = check_dims(planes, iz, iy, ix, oz, oy, ox)) return e;
  VAMP_REQUIRE(grad_out && grad_in && workspace, "NULL tensor");
  VAMP_REQUIRE(dtype == VAMP_F32 || dtype == VAMP_BF16 || dtype == VAMP_F16, "dtype");
  VAMP_REQUIRE(workspace_bytes >= table_bytes(iz, iy, ix), "workspace too small");
  // outputs whose source coordinate lies in [i - 1, i + 1): at most floor(2 (out-1)/(in-1)) + 1
  VAMP_REQUIRE(run_fits(iz, oz) && run_fits(iy, oy) && run_fits(ix, ox),
               "scale factor too large for the gather table (about out / in <= 6 per axis)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AxisHit* tab = static_cast<AxisHit*>(workspace);
  const int n = iz + iy + ix;
  upsample_axis_table_kernel<<<(n + 255) / 256, 256, 0, s>>>(tab, iz, iy, ix, oz, oy, ox);
  if (int e = check_launch("upsample_axis_table_kernel")) return e;
  if (dtype == VAMP_F32) return up_backward<float>(planes, iz, iy, ix, oz, oy, ox, grad_out, grad_in, tab, s);
  if (dtype == VAMP_BF16) return up_backward<Bf16>(planes, iz, iy, ix, oz, oy, ox, grad_out, grad_in, tab, s);
  return up_backward<F16>(planes, iz, iy, ix, oz, oy, ox, grad_out, grad_in, tab, s);
}

int vamp_upsample_trilinear_backward(int64_t planes, int32_t iz, int32_t iy, int32_t ix, int32_t oz,
                                     int32_t oy, int32_t ox, const float* grad_out, float* grad_in,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  return vamp_upsample_trilinear_backward_ex(planes, iz, iy, ix, oz, oy, ox, VAMP_F32, grad_out, grad_in, workspace,
                                             workspace_bytes, stream);
}

}  // extern "C"
