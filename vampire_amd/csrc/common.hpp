// Shared device/host helpers for the gfx950 lift + render kernels.
// Compiled with -ffp-contract=off: every a*b+c below is two IEEE roundings unless
// written as __builtin_fmaf.  The projection chains rely on that to reproduce the
// reference's fp32 results bit for bit (DESIGN.md "Bit-exact projection").
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vampire_hip.h"

namespace vamp {

constexpr int kWave = 64;

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return code;
}

#define VAMP_REQUIRE(cond, msg)                                              \
  do {                                                                       \
    if (!(cond)) return ::vamp::fail(VAMP_EINVAL, "%s: requirement failed: " msg, __func__); \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return VAMP_EHIP;
  }
  return VAMP_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------
// optional HIP-event kernel timer (runtime.hip); slots name the kernels
// ---------------------------------------------------------------------------
enum ProfSlot {
  kProfFeatCL = 0, kProfLiftFwd, kProfLiftBwd, kProfFeatCF, kProfLiftFwdDense, kProfLiftBwdDense,
  kProfPack, kProfCamFwd, kProfBevFwd, kProfCamBwd, kProfUnpack, kProfBevBwd, kProfMemset,
  kProfAux, kProfCamBwdBrick, kProfBevFwdCh, kProfBevBwdQ, kProfBevBwdGather, kProfLiftBwdV1,
  kProfLiftBwdCount, kProfLiftBwdFill, kProfCamBwdCount, kProfCamBwdFill, kProfCamBwdOwn, kProfCamBwdV1,
  kProfGlueSoftmax, kProfGlueGate, kProfUpsample, kProfConvFwd, kProfConvDgrad, kProfConvWgrad, kProfCamTerm, kProfRenderFwdMerged, kProfSlots
};
struct ProfScope { int idx; };
bool prof_enabled();
void prof_begin(int slot, hipStream_t s, ProfScope* sc);
void prof_end(hipStream_t s, ProfScope* sc);
// device-side exclusive scan of bin counts (runtime.hip): off[i] = sum_{j<i} cnt[j], fill[i] = 0
int launch_exclusive_scan(const int* cnt, int* off, int* fill, int n, int* total, hipStream_t s);
// Two-level exclusive scan of `ncell` counters (ncell a multiple of kScanTile) for the cell lists:
// afterwards the start offset of cell c is off[c] + boff[c / kScanTile]; bsum / boff / aux hold
// ncell / kScanTile ints (aux four more) and aux[ncell / kScanTile] receives the grand total; the word
// and the two words behind it are zeroed (list counters of the cell-list users).
// grad_beta[0] += sign(beta_raw[0]) * sum(part[0..n)), summed in a fixed order by one workgroup
// zero `bytes` (multiple of 4) at ptr with a kernel (graph-safe, see runtime.hip)
int launch_zero(void* ptr, size_t bytes, hipStream_t s);
// vamp_debug_checks(1): verify (synchronously) that p[0 .. n) lies in [lo, hi]; VAMP_OK when the mode is off
extern bool g_debug_checks;
int debug_expect_range(const int* p, size_t n, int lo, int hi, hipStream_t s, const char* what);
int launch_beta_reduce(const float* part, int n, const float* beta_raw, float* grad_beta, hipStream_t s);
constexpr int kScanTile = 2048;
// `cnt` holds ncell + kScanPad ints, all zeroed by the caller before the count pass: cnt[ncell] is the
// scan's arrival ticket (the workgroup that takes the last ticket scans the tile totals, so the whole
// scan is one launch)
constexpr int kScanPad = 64;
// A duty the scan's first workgroup can take over from a launch of its own: the order in which a later kernel takes
// its `n` work items -- those with the largest key first (counting sort by the bit length of key[i] >= 0).  The camera
// backward's per-ray pass takes its ray tiles deepest first that way (render_bwd_cell.hip).
struct ScanDuty {
  const int* key;      // [n], nullptr: no duty
  int* order;          // [n] out
  int n;
};
int launch_cell_scan(int* cnt, int* off, int* bsum, int* boff, int* aux, long ncell,
                     hipStream_t s, const ScanDuty* duty = nullptr);
// the same scan as a job description, and two jobs in one launch (runtime.hip: cell_scan_pair_kernel)
struct ScanJob {
  int *cnt, *off, *bsum, *boff, *fill, *total, *ticket;
  int ntile;
};
int make_scan_job(int* cnt, int* off, int* bsum, int* boff, int* aux, long ncell, ScanJob* job);
int launch_cell_scan_pair(const ScanJob& a, const ScanJob& b, hipStream_t s, const ScanDuty* duty = nullptr);
// A duty a consumer kernel takes over from a launch of its own: the first workgroup adds up, in a
// fixed order, the per-workgroup partial sums of d loss / d beta_eff that an EARLIER kernel of the same
// stream left in `part` (kernel boundary: visible), and adds sign(beta_raw) * sum to grad_beta with one
// atomic (the camera and BEV branches add to the same word from two streams; two addends commute, so
// the sum has the same bits every run).  part == nullptr: nothing to do.
struct BetaTail {
  const float* part;
  int n;
  const float* beta_raw;
  float* grad_beta;
};
__device__ __forceinline__ void beta_tail(const BetaTail& t) {
  if (!t.part || blockIdx.x != 0 || blockIdx.y != 0 || blockIdx.z != 0) return;     // uniform per workgroup
  __shared__ float beta_red[16];
  const int nthr = blockDim.x * blockDim.y * blockDim.z;
  float v = 0.f;
  for (int i = threadIdx.x; i < t.n; i += nthr) v += t.part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) beta_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < (nthr + 63) / 64; ++i) tot += beta_red[i];
    const float b = t.beta_raw[0];
    atomicAdd(t.grad_beta, ((b > 0.f) ? 1.f : ((b < 0.f) ? -1.f : 0.f)) * tot);
  }
  __syncthreads();
}

// usage: VAMP_TIMED(slot, stream, kernel<<<...>>>(...));
#define VAMP_TIMED(slot, stream, launch)              \
  do {                                                \
    ::vamp::ProfScope _sc;                            \
    ::vamp::prof_begin(slot, stream, &_sc);           \
    launch;                                           \
    ::vamp::prof_end(stream, &_sc);                   \
  } while (0)

// ---------------------------------------------------------------------------
// 4x4 * 4x1 in the evaluation order of torch's CPU bmm for tiny matrices:
// ((m0*x + m1*y) + m2*z) + m3*w, no fused multiply-add.
// ---------------------------------------------------------------------------
struct Vec4 {
  float x, y, z, w;
};

__device__ __forceinline__ Vec4 matvec(const float* __restrict__ m, Vec4 p) {
  Vec4 r;
  r.x = ((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3] * p.w;
  r.y = ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7] * p.w;
  r.z = ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11] * p.w;
  r.w = ((m[12] * p.x + m[13] * p.y) + m[14] * p.z) + m[15] * p.w;
  return r;
}

// element loads with fp32 promotion
__device__ __forceinline__ float ldf(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float ldf(const __hip_bfloat16* p, long i) {
  return __uint_as_float(((uint32_t) reinterpret_cast<const uint16_t*>(p)[i]) << 16);
}

__device__ __forceinline__ float nan_to_num(float v) {
  // torch.nan_to_num defaults: nan -> 0, +-inf -> +-FLT_MAX
  if (v != v) return 0.f;
  return fminf(fmaxf(v, -3.402823466e+38f), 3.402823466e+38f);
}

// ---------------------------------------------------------------------------
// XCD-aware workgroup order.  The hardware deals consecutive workgroups round-robin to the 8
// XCDs, each with its own L2, so spatial neighbours (which share input lines) land in eight
// different caches and every line is fetched up to eight times.  xcd_block() returns the logical
// (x, y, z) block such that XCD k walks the contiguous eighth [k n/8, (k+1) n/8) of the grid in
// launch order (identity when the workgroup count is not a multiple of 8).
// ---------------------------------------------------------------------------
__device__ __forceinline__ dim3 xcd_block() {
  const unsigned n = gridDim.x * gridDim.y * gridDim.z;
  unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((n & 7u) == 0) lin = (lin & 7u) * (n >> 3) + (lin >> 3);
  dim3 r;
  r.x = lin % gridDim.x;
  lin /= gridDim.x;
  r.y = lin % gridDim.y;
  r.z = lin / gridDim.y;
  return r;
}

// Same idea for 1-D grids with uneven work: runs of `group` consecutive logical blocks stay on
// one XCD (so that neighbours share an L2 and their partial-line stores merge there), and the
// runs are dealt round-robin, which keeps the load of the XCDs even.  Identity unless the grid
// is a multiple of 8 * group.
__device__ __forceinline__ unsigned xcd_grouped(unsigned lin, unsigned n, unsigned group) {
  const unsigned span = 8u * group;
  if (group == 0 || n % span != 0) return lin;
  return (lin / span) * span + (lin & 7u) * group + (lin % span) / 8u;
}

// ---------------------------------------------------------------------------
// lane-group reductions
// ---------------------------------------------------------------------------
// Sum N per-lane values over the lanes of a W-wide group with recursive halving: at every xor
// step with an even count each lane keeps one half of the values and hands the other half to its
// partner, so the step moves N/2 values instead of N (an odd count falls back to a plain
// butterfly).  On return a[0 .. reduce_left<N, W/2>()) are the complete sums of channels
// cbase .. ; lanes that differ only in the bits of reduce_dups<N, W/2>() hold copies.
template <int N, int O>
constexpr int reduce_left() {
  if constexpr (O == 0) return N;
  else if constexpr (N % 2 == 0) return reduce_left<N / 2, O / 2>();
  else return reduce_left<N, O / 2>();
}
template <int N, int O>
constexpr int reduce_dups() {
  if constexpr (O == 0) return 0;
  else if constexpr (N % 2 == 0) return reduce_dups<N / 2, O / 2>();
  else return O | reduce_dups<N, O / 2>();
}
// The exchanges never touch the LDS crossbar (ds_bpermute, what __shfl_xor compiles to, made this
// reduction 16 of the lift gather's 82 us): across 32 and 16 lanes gfx950's v_permlane32_swap /
// v_permlane16_swap trade the two halves in one instruction; inside a row of 16 the partner comes
// through DPP -- row_ror:8 (l ^ 8), row_half_mirror (l ^ 7: differs from l in bit 2, which is all
// a pairing step needs, and the two steps that follow pair l ^ 2 and l ^ 1, so the eight lanes
// are still covered), quad_perm (l ^ 2, l ^ 1).
template <int O>
__device__ __forceinline__ float lane_partner(float v) {
  static_assert(O == 8 || O == 4 || O == 2 || O == 1, "DPP partner inside a row of 16");
  constexpr int ctrl = O == 8 ? 0x128 : (O == 4 ? 0x141 : (O == 2 ? 0x4E : 0xB1));
  return __builtin_amdgcn_update_dpp(0.f, v, ctrl, 0xf, 0xf, false);
}
// lanes whose bit O is clear get lo(l) + lo(l ^ O), the others hi(l) + hi(l ^ O)   (O = 32, 16)
template <int O>
__device__ __forceinline__ float swap_add(float lo, float hi) {
  static_assert(O == 32 || O == 16, "row swaps");
  if constexpr (O == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}
template <int N, int O, int W, int CP>
__device__ __forceinline__ void reduce_halving(float (&a)[CP], int l, int& cbase) {
  if constexpr (O == 0) {
    return;
  } else if constexpr (N % 2 == 0) {
    constexpr int H = N / 2;
    const bool up = (l & O) != 0;
    if constexpr (O >= 16) {
#pragma unroll
      for (int c = 0; c < H; ++c) a[c] = swap_add<O>(a[c], a[c + H]);
    } else {
#pragma unroll
      for (int c = 0; c < H; ++c) {
        const float send = up ? a[c] : a[c + H];
        const float keep = up ? a[c + H] : a[c];
        a[c] = keep + lane_partner<O>(send);
      }
    }
    cbase += up ? H : 0;
    reduce_halving<H, O / 2, W, CP>(a, l, cbase);
  } else {
#pragma unroll
    for (int c = 0; c < N; ++c) {
      if constexpr (O >= 16) a[c] = swap_add<O>(a[c], a[c]);
      else a[c] += lane_partner<O>(a[c]);
    }
    reduce_halving<N, O / 2, W, CP>(a, l, cbase);
  }
}

// ---------------------------------------------------------------------------
// density activations (render_utils.py:30-46 / nn.Sigmoid)
// ---------------------------------------------------------------------------
struct DensityParams {
  int mode;
  float beta;   // |beta_raw| + beta_min
  float ib;     // 1 / beta, divided once per thread: |t| * ib stands for the reference's |t| / beta
                // (one ulp in the exponent's argument; the values are held to 1e-4, not bit-exact)
  float bias;
};

__device__ __forceinline__ float density_fwd(const DensityParams& dp, float s) {
  if (dp.mode == VAMP_DENSITY_SIGMOID) return 1.f / (1.f + expf(-s));
  float t = s - dp.bias;
  float sg = (t > 0.f) ? 1.f : ((t < 0.f) ? -1.f : 0.f);
  return dp.ib * (0.5f + 0.5f * sg * expm1f(-fabsf(t) * dp.ib));
}

// sigma, d sigma / d s and d sigma / d beta_eff (beta_eff = |beta_raw| + beta_min) with one
// transcendental: exp(x) is taken as expm1(x) + 1 (x <= 0)
__device__ __forceinline__ void density_all(const DensityParams& dp, float s, float& sigma,
                                            float& dsig_ds, float& dsig_dbeta) {
  if (dp.mode == VAMP_DENSITY_SIGMOID) {
    sigma = 1.f / (1.f + expf(-s));
    dsig_ds = sigma * (1.f - sigma);
    dsig_dbeta = 0.f;
    return;
  }
  const float t = s - dp.bias;
  const float a = fabsf(t);
  const float sg = (t > 0.f) ? 1.f : ((t < 0.f) ? -1.f : 0.f);
  const float ib = dp.ib;
  const float em1 = expm1f(-a * ib);         // exp(-|t|/beta) - 1
  const float e = em1 + 1.0f;
  sigma = ib * (0.5f + 0.5f * sg * em1);
  // d/ds: ib * 0.5*sg * e * (-sg*ib) = -0.5*ib^2*e*sg^2   (sign(t)' = 0 a.e.)
  dsig_ds = -0.5f * ib * ib * e * (sg * sg);
  // d/dbeta: -ib^2*(0.5+0.5*sg*em1) + ib*0.5*sg*e*(a*ib^2)
  dsig_dbeta = -ib * ib * (0.5f + 0.5f * sg * em1) + 0.5f * sg * e * a * ib * ib * ib;
}

__device__ __forceinline__ void density_bwd(const DensityParams& dp, float s, float& dsig_ds,
                                            float& dsig_dbeta) {
  float sigma;
  density_all(dp, s, sigma, dsig_ds, dsig_dbeta);
}

__device__ __forceinline__ DensityParams load_density(int mode, const float* beta_raw,
                                                      float beta_min, float bias) {
  DensityParams dp;
  dp.mode = mode;
  dp.bias = bias;
  dp.beta = (mode == VAMP_DENSITY_SDF_LAPLACE) ? (fabsf(beta_raw[0]) + beta_min) : 1.f;
  dp.ib = 1.f / dp.beta;
  return dp;
}

}  // namespace vamp
