// LIFT kernels for gfx950: every voxel pulls a trilinear sample of
// depth[d,h,w] * feat[c,h,w] from each camera and averages over the cameras that
// hit it.  Reference call sites: base_vampire2.py:550-553 (outer product),
// :351-388 (get_pixel), :483-516 (get_voxel_feats).
//
// The outer product is never materialised: the trilinear sample of a rank-1
// (depth x feat) volume separates into
//     sum_{4 (h,w) taps} w_hw * feat[c,h,w] * ( sum_{2 d taps} w_d * depth[d,h,w] ).
// Memory-bound gather; no MFMA.  One thread per voxel, lanes along x so the
// [B,C,Z,Y,X] stores coalesce; features are read channel-last (one 64-byte run
// per tap for C = 16) from a transposed copy made by a tiny pre-pass.
#include "lift_common.hpp"
#include "depth_softmax.hpp"

namespace vamp {

// tile shape: lanes along x for coalesced stores
// A wave is a 16 x 4 patch of voxels, not a 64 x 1 row: the exact wave-level camera cull of
// lift_project<true> skips a camera only when none of the wave's voxels has it in front, and a
// 6.4 m x 1.6 m patch is on one side of most cameras where a 25.6 m row is not (cfg-B: 47 -> 36 us;
// 8 x 32 / 16 x 16 / 32 x 8 / 64 x 4 workgroup tiles: 36.5 / 35.7 / 38.2 / 47.1).
#ifndef VAMP_LIFT_TX
#define VAMP_LIFT_TX 16
#define VAMP_LIFT_TY 16
#endif
#define VAMP_LIFT_TILE VAMP_LIFT_TX, VAMP_LIFT_TY, 1
#ifndef VAMP_LIFT_COOP
#define VAMP_LIFT_COOP 1
#endif

// ---------------------------------------------------------------------------
// feat [BN, C, HW] (f32 or bf16) -> channel-last fp32 [BN, HW, C]
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void feat_to_channel_last_tile(const T* __restrict__ feat, float* __restrict__ out,
                                                          int C, int HW, long bn, int p0, int c0,
                                                          float (&tile)[64][65]) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    int c = c0 + j, p = p0 + tx;
    tile[j][tx] = (c < C && p < HW) ? ldf(feat, (bn * C + c) * HW + p) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    int p = p0 + j, c = c0 + tx;
    if (p < HW && c < C) out[(bn * HW + p) * C + c] = tile[tx][j];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) feat_to_channel_last(const T* __restrict__ feat,
                                                            float* __restrict__ out, int C, int HW) {
  __shared__ float tile[64][65];
  feat_to_channel_last_tile<T>(feat, out, C, HW, blockIdx.z, blockIdx.x * 64, blockIdx.y * 64, tile);
}

// ---------------------------------------------------------------------------
// Camera cull words of the forward (round 5).  The waves of a lift_fwd_kernel workgroup -- a TX x TY patch
// of voxel centres in one z plane -- read ONE word before they project anything: bit n set = camera n's
// frustum may contain a voxel of the patch.  The word comes from the matrices alone.  With ida's rows
// (a00, a01, 0, a03), (a10, a11, 0, a13), (0, 0, 1, 0) -- image-plane augmentations, the reference's only
// kind (nusc_det_seg_dataset.py:490-498) -- the point q = (K inv(s2e)) (inv(bda) c) is affine in the voxel
// centre c and, wherever q.z > 0,
//     u > -0.5   <=>   a00 q.x + a01 q.y + (a03 q.w + 0.5) q.z > 0            (likewise u_max, v, depth)
// so each of the six bounds of `valid` (bv2:493-497) is a half-space of c.  If the four corners of the
// patch's bounding rectangle GROWN BY ONE VOXEL on every side lie outside ONE of them by a further margin of
// one feature pixel / one depth bin -- both four orders of magnitude above the rounding of the fp32 chain and
// of the fp32 product (K inv(s2e)) inv(bda) the corners go through -- the whole patch does, and no lane can
// find the camera valid: skipping it changes no bit of the result.
// Anything else (another shape of ida, NaN / Inf anywhere) leaves the bit set; the exact chain of
// lift_project still decides every survivor, and its wave-level test (lift_project_from<true>) still drops a
// camera the wave's own 16 x 4 voxels are behind.  cfg-B: 1.6 of 6 cameras per patch survive; the forward
// kernel went from 1 369 to 587 vector instructions per wave.
// Bit 15: all cameras of the sample share inv(bda) bit for bit (its product is then formed once per voxel).
// ---------------------------------------------------------------------------
constexpr unsigned kLiftCullSharedBda = 1u << 15;

struct LiftCull {
  unsigned* words;      // [B][Z][nyp][nxp]
  int nxp, nyp;         // patches per row / column of one z plane
  int px, py;           // patch shape in voxels
  int group;            // lanes per patch in a cull workgroup: the power of two >= N
  int bps;              // stand-alone cull: workgroups (256 / group patches each) per sample
  int ppt;              // forward's first launch: patches per feature tile (N * ptiles tiles per sample)
};

// What a cull workgroup keeps per camera of its sample (LDS): A = (K inv(s2e)) inv(bda) in fp32 -- its
// rounding is what the margins are four orders of magnitude above -- and the image-plane rows of ida.
struct LiftCullCam {
  float A[16];
  float a00, a01, a03, a10, a11, a13;
  int plane;          // ida has the image-plane shape the half-space form needs
  int same;           // inv(bda) is camera 0's, bit for bit
};
constexpr int kLiftCullMaxCams = 15;
#ifndef VAMP_CULL_WPW
#define VAMP_CULL_WPW 4      // (1 / 2 / 4: first launch 9.9 / 9.0 / 9.0 us, forward kernel 28.3 / 28.4 / 28.4 at cfg-B)
#endif
constexpr int kCullWpw = VAMP_CULL_WPW;     // waves of lift_fwd_kernel (stacked along y) that share a word: 1, 2 or 4
constexpr int kCullPX = VAMP_LIFT_TX, kCullPY = 64 / VAMP_LIFT_TX * kCullWpw;

// The cull in three pieces, so that a launch can put its loads beside other loads and its arithmetic behind
// a barrier it has anyway.  (1) per camera of the sample, once per workgroup: the constants above (wave 0);
// (2) per patch of the workgroup's share, once: its bounding rectangle (wave 1); barrier; (3) a lane per
// (patch, camera), the cameras of a patch in G consecutive lanes (G = the power of two >= N): the test from
// LDS, the group's word a ballot.
struct LiftCullBox { float xlo, xhi, ylo, yhi, zc; int inside; };
constexpr int kCullBoxes = 64;        // patches per round of a workgroup (256 / G <= 64 for G >= 4; G < 4: rounds of 64)

struct LiftCullLds {
  LiftCullCam cams[kLiftCullMaxCams];
  LiftCullBox box[kCullBoxes];
};

__device__ __forceinline__ int lift_cull_round(const LiftCull& K) { return min(256 / K.group, kCullBoxes); }

// (1): per camera `cn` of sample b.  16-byte loads: the matrices are 64-byte rows of a 16-byte aligned array
// (checked at the entry points).
struct LiftCullMats { float4 r[12], z[4]; };    // the camera's three matrices | inv(bda) of the sample's first camera
__device__ __forceinline__ LiftCullMats lift_cull_load(const LiftParams& P, const float* __restrict__ mats, int b, int cn) {
  const float4* m = reinterpret_cast<const float4*>(mats + ((long) b * P.N + cn) * 48);
  const float4* m0 = reinterpret_cast<const float4*>(mats + (long) b * P.N * 48);
  LiftCullMats L;
#pragma unroll
  for (int i = 0; i < 12; ++i) L.r[i] = m[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) L.z[i] = m0[i];
  return L;
}
__device__ __forceinline__ LiftCullCam lift_cull_cam_of(const LiftCullMats& L) {
  const float4 (&r)[12] = L.r;
  const float4 (&z)[4] = L.z;
  LiftCullCam c;
  unsigned diff = 0;                     // (bitwise, no short-circuit: the comparison stays inside the loads' basic block)
#pragma unroll
  for (int i = 0; i < 4; ++i)
    diff |= (__float_as_uint(r[i].x) ^ __float_as_uint(z[i].x)) | (__float_as_uint(r[i].y) ^ __float_as_uint(z[i].y)) |
            (__float_as_uint(r[i].z) ^ __float_as_uint(z[i].z)) | (__float_as_uint(r[i].w) ^ __float_as_uint(z[i].w));
  const bool same = diff == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {          // A = M2 . M1, row i
    const float4 k = r[4 + i];
    c.A[i * 4 + 0] = __builtin_fmaf(k.x, r[0].x, __builtin_fmaf(k.y, r[1].x, __builtin_fmaf(k.z, r[2].x, k.w * r[3].x)));
    c.A[i * 4 + 1] = __builtin_fmaf(k.x, r[0].y, __builtin_fmaf(k.y, r[1].y, __builtin_fmaf(k.z, r[2].y, k.w * r[3].y)));
    c.A[i * 4 + 2] = __builtin_fmaf(k.x, r[0].z, __builtin_fmaf(k.y, r[1].z, __builtin_fmaf(k.z, r[2].z, k.w * r[3].z)));
    c.A[i * 4 + 3] = __builtin_fmaf(k.x, r[0].w, __builtin_fmaf(k.y, r[1].w, __builtin_fmaf(k.z, r[2].w, k.w * r[3].w)));
  }
  c.a00 = r[8].x; c.a01 = r[8].y; c.a03 = r[8].w; c.a10 = r[9].x; c.a11 = r[9].y; c.a13 = r[9].w;
  c.plane = r[8].z == 0.f && r[9].z == 0.f && r[10].x == 0.f && r[10].y == 0.f && r[10].z == 1.f && r[10].w == 0.f;
  c.same = same;
  return c;
}
__device__ __forceinline__ LiftCullCam lift_cull_cam(const LiftParams& P, const float* __restrict__ mats, int b, int cn) {
  return lift_cull_cam_of(lift_cull_load(P, mats, b, cn));
}
// threads [0, N) of a workgroup: into LDS
__device__ __forceinline__ void lift_cull_constants(const LiftParams& P, const float* __restrict__ mats, int b,
                                                    LiftCullLds& S) {
  if ((int) threadIdx.x >= P.N) return;
  S.cams[threadIdx.x] = lift_cull_cam(P, mats, b, threadIdx.x);
}

// The bounding rectangle of the patch of kCullPX x kCullPY voxel centres whose first voxel is (x0, y0) in plane z,
// grown by one voxel on every side (the axes are arrays of the caller: no monotonicity assumed; loads issued together)
__device__ __forceinline__ LiftCullBox lift_cull_box_of(const LiftParams& P, const float* __restrict__ xs,
                                                        const float* __restrict__ ys, const float* __restrict__ zs,
                                                        int x0, int y0, int z) {
  float xv[kCullPX], yv[kCullPY];
#pragma unroll
  for (int i = 0; i < kCullPX; ++i) xv[i] = xs[min(x0 + i, P.X - 1)];
#pragma unroll
  for (int i = 0; i < kCullPY; ++i) yv[i] = ys[min(y0 + i, P.Y - 1)];
  LiftCullBox B;
  B.zc = zs[z];
  float xlo = xv[0], xhi = xlo, ylo = yv[0], yhi = ylo;
#pragma unroll
  for (int i = 1; i < kCullPX; ++i) { xlo = fminf(xlo, xv[i]); xhi = fmaxf(xhi, xv[i]); }
#pragma unroll
  for (int i = 1; i < kCullPY; ++i) { ylo = fminf(ylo, yv[i]); yhi = fmaxf(yhi, yv[i]); }
  const int nx = min(kCullPX, P.X - x0), ny = min(kCullPY, P.Y - y0);
  const float gx = nx > 1 ? (xhi - xlo) / (float) (nx - 1) : 0.f, gy = ny > 1 ? (yhi - ylo) / (float) (ny - 1) : 0.f;
  B.xlo = xlo - gx; B.xhi = xhi + gx; B.ylo = ylo - gy; B.yhi = yhi + gy;
  B.inside = 1;
  return B;
}

// (2): threads [64, 64 + round): the bounding rectangle of patch `first + slot` of the sample, grown by one
// voxel on every side (the axes are arrays of the caller: no monotonicity assumed; loads issued together)
__device__ __forceinline__ void lift_cull_boxes(const LiftParams& P, const float* __restrict__ xs,
                                                const float* __restrict__ ys, const float* __restrict__ zs,
                                                const LiftCull& K, long first, long count, LiftCullLds& S) {
  const int slot = (int) threadIdx.x - 64;
  if (slot < 0 || slot >= lift_cull_round(K) || slot >= count) return;
  const long per_sample = (long) P.Z * K.nyp * K.nxp;
  const long pl = min(first + slot, per_sample - 1);
  const int xp = (int) (pl % K.nxp);
  const long r = pl / K.nxp;
  const int yp = (int) (r % K.nyp);
  const int x0 = min(xp * K.px, P.X - 1), y0 = min(yp * K.py, P.Y - 1);
  LiftCullBox B = lift_cull_box_of(P, xs, ys, zs, x0, y0, (int) (r / K.nyp));
  B.inside = first + slot < per_sample && xp * K.px < P.X && yp * K.py < P.Y;
  S.box[slot] = B;
}

// the test itself: may a voxel of the patch `B` be valid in camera `c`?  (false = provably none)
__device__ __forceinline__ bool lift_cull_test(const LiftParams& P, const LiftCullCam& c, const LiftCullBox& B) {
  const float mu = P.u_div / (float) P.fW, mv = P.v_div / (float) P.fH;      // one feature pixel
  // one depth bin; without depth bounds (D == 1: valid needs z > 0) a centimetre -- three orders above the rounding
  // of the fused product A = M2 M1 the corners go through
  const float mz = P.use_depth ? P.d_span / (float) P.D : 0.01f;
  const float zlo = P.use_depth ? P.d_lo : 0.f;
  bool keep = true;
  if (c.plane) {
    // corners q = A0 x + A1 y + (A2 z + A3)
    float base[4], ax[2][4], ay[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      base[i] = __builtin_fmaf(c.A[i * 4 + 2], B.zc, c.A[i * 4 + 3]);
      ax[0][i] = c.A[i * 4] * B.xlo; ax[1][i] = c.A[i * 4] * B.xhi;
      ay[0][i] = c.A[i * 4 + 1] * B.ylo; ay[1][i] = c.A[i * 4 + 1] * B.yhi;
    }
    // the range of a03 q.w / a13 q.w over the patch (q.w is 1 up to the rounding of inv(s2e))
    const float wlo = base[3] + fminf(ax[0][3], ax[1][3]) + fminf(ay[0][3], ay[1][3]);
    const float whi = base[3] + fmaxf(ax[0][3], ax[1][3]) + fmaxf(ay[0][3], ay[1][3]);
    const float u3lo = fminf(c.a03 * wlo, c.a03 * whi), u3hi = fmaxf(c.a03 * wlo, c.a03 * whi);
    const float v3lo = fminf(c.a13 * wlo, c.a13 * whi), v3hi = fmaxf(c.a13 * wlo, c.a13 * whi);
    bool o_near = true, o_far = P.use_depth != 0, o_l = true, o_r = true, o_t = true, o_b = true, front = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float qx = base[0] + ax[k & 1][0] + ay[k >> 1][0], qy = base[1] + ax[k & 1][1] + ay[k >> 1][1];
      const float qz = base[2] + ax[k & 1][2] + ay[k >> 1][2];
      o_near = o_near && qz <= zlo - mz;
      o_far = o_far && qz >= P.d_hi + mz;
      const float lu = c.a00 * qx + c.a01 * qy, lv = c.a10 * qx + c.a11 * qy;
      o_l = o_l && lu + (u3hi + 0.5f + mu) * qz <= 0.f;
      o_r = o_r && lu + (u3lo - P.u_max - mu) * qz >= 0.f;
      o_t = o_t && lv + (v3hi + 0.5f + mv) * qz <= 0.f;
      o_b = o_b && lv + (v3lo - P.v_max - mv) * qz >= 0.f;
      front = front && qz >= 1.0f;
    }
    // with the depth bounds a valid voxel has q.z > d_lo; without them (D == 1) or with a near bound
    // below a metre the side planes are used only for patches wholly a metre in front of the camera
    // (the chain clamps q.z at 1e-6, and the margins are worth a feature pixel only away from q.z = 0)
    const bool sides = (P.use_depth && P.d_lo >= 1.0f) || front;
    if (o_near || o_far || (sides && (o_l || o_r || o_t || o_b))) keep = false;
  }
  return keep;
}

// (3): every wave that holds a live lane (wave-uniform: it ballots)
__device__ __forceinline__ void lift_cull_eval(const LiftParams& P, const LiftCull& K, int b, long first, long count,
                                               const LiftCullLds& S) {
  const int G = K.group, round = lift_cull_round(K);
  const int slot = threadIdx.x / G, n = threadIdx.x & (G - 1);
  const int wave_slot0 = (threadIdx.x & ~63) / G;                 // first slot of this wave
  if (wave_slot0 >= round || wave_slot0 >= count) return;
  const long per_sample = (long) P.Z * K.nyp * K.nxp;
  const bool live = slot < round && slot < count && first + slot < per_sample && n < P.N;
  const LiftCullBox B = S.box[min(slot, round - 1)];
  const LiftCullCam c = S.cams[min(n, P.N - 1)];
  const bool keep = lift_cull_test(P, c, B);
  const int lane = threadIdx.x & 63, g0 = lane & ~(G - 1);
  const uint64_t gm = ((G == 64) ? ~0ull : ((1ull << G) - 1)) << g0;
  const uint64_t kept = __ballot(live && keep) & gm, differ = __ballot(live && !c.same) & gm;
  if (live && n == 0)
    K.words[(long) b * per_sample + first + slot] =
        B.inside ? ((unsigned) (kept >> g0) | (differ ? 0u : kLiftCullSharedBda)) : 0u;
}

// `count` patches of sample b from `first` on, by one workgroup whose caller has run (1) and (2) for the
// first round and a barrier; further rounds (a share larger than 256 / G patches: huge grids on tiny images)
__device__ __forceinline__ void lift_cull_share(const LiftParams& P, const float* __restrict__ xs,
                                                const float* __restrict__ ys, const float* __restrict__ zs,
                                                const LiftCull& K, int b, long first, long count, LiftCullLds& S) {
  const int round = lift_cull_round(K);
  lift_cull_eval(P, K, b, first, count, S);
  for (long done = round; done < count; done += round) {       // (uniform over the workgroup)
    __syncthreads();
    lift_cull_boxes(P, xs, ys, zs, K, first + done, count - done, S);
    __syncthreads();
    lift_cull_eval(P, K, b, first + done, count - done, S);
  }
}

// The word of ONE workgroup's own patch, formed at its head: what the forward kernels do when no first launch has
// left the words in the workspace -- a caller that hands the features over channel-last needs no transpose, and with
// it the whole first launch goes (VAMP_LIFTFWD_FEAT_CHANNEL_LAST).  All threads call it.
//
// The same test as lift_cull_test -- the patch's bounding rectangle grown by a voxel, its four corners against the six
// half-spaces of `valid` with the same margins -- laid out over the lanes of wave 0 instead of down one lane: lane =
// (camera, corner), 4 N <= 60 lanes.  A lane sends ONE corner through inv(bda) and K inv(s2e) (two matrix-vector
// products, 28 fma, where the lane-per-camera form built the 4 x 4 product and then walked the corners: ~400 dependent
// vector instructions per workgroup, 2.5 us at the head of each of them and 7 us on the kernel), the range of q.w and
// the "all four corners outside" conjunctions are two-step exchanges inside the camera's quad of lanes, and the patch's
// axes arrive as one load (lanes 0 .. 15 x, 16 .. 31 y) + a 16-lane min / max.  The corners go through the two matrices
// one after the other here and through their fp32 product there: roundings four orders of magnitude inside the margins
// either way (a feature pixel, a depth bin); both are conservative, so the forward's bits do not depend on which ran.
__device__ __forceinline__ unsigned lift_cull_inline(const LiftParams& P, const float* __restrict__ mats,
                                                     const float* __restrict__ xs, const float* __restrict__ ys,
                                                     const float* __restrict__ zs, int b, int x0, int y0, int z) {
  static_assert(kCullWpw == 4 && kCullPX == 16 && kCullPY == 16, "one word per workgroup of 16 x 16 voxels");
  __shared__ unsigned word_s;
  if (threadIdx.x < 64) {                                         // wave 0
    const int lane = threadIdx.x, n = lane >> 2, k = lane & 3;
    const bool act = n < P.N;
    const float4* m = reinterpret_cast<const float4*>(mats + ((long) b * P.N + (act ? n : 0)) * 48);
    const float4* m0 = reinterpret_cast<const float4*>(mats + (long) b * P.N * 48);
    float4 r[11], z4[4];
#pragma unroll
    for (int i = 0; i < 11; ++i) r[i] = m[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) z4[i] = m0[i];
    // the patch's axes (arrays of the caller: no monotonicity assumed) and their extremes
    const int x0c = min(x0, P.X - 1), y0c = min(y0, P.Y - 1);
    const float av = (lane & 16) ? ys[min(y0c + (lane & 15), P.Y - 1)] : xs[min(x0c + (lane & 15), P.X - 1)];
    const float zc = zs[z];
    // (exchanges inside a quad and inside a row of 16 lanes are DPP operands of the min / max itself; through the LDS
    // crossbar, which is what __shfl compiles to, each of the nine dependent steps of this function was a round trip)
    float lo = av, hi = av;
    lo = fminf(lo, __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false)); hi = fmaxf(hi, __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    lo = fminf(lo, __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false)); hi = fmaxf(hi, __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    lo = fminf(lo, __builtin_amdgcn_update_dpp(lo, lo, 0x124, 0xf, 0xf, false)); hi = fmaxf(hi, __builtin_amdgcn_update_dpp(hi, hi, 0x124, 0xf, 0xf, false));   // row_ror:4
    lo = fminf(lo, __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xf, 0xf, false)); hi = fmaxf(hi, __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xf, 0xf, false));   // row_ror:8
    const float xlo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo), 0)), xhi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi), 0));
    const float ylo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo), 16)), yhi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi), 16));
    const int nx = min(kCullPX, P.X - x0c), ny = min(kCullPY, P.Y - y0c);
    const float gx = nx > 1 ? (xhi - xlo) / (float) (nx - 1) : 0.f, gy = ny > 1 ? (yhi - ylo) / (float) (ny - 1) : 0.f;
    // this lane's corner of the grown rectangle
    const float cx = (k & 1) ? xhi + gx : xlo - gx, cy = (k & 2) ? yhi + gy : ylo - gy;
    float pv[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pv[i] = __builtin_fmaf(r[i].x, cx, __builtin_fmaf(r[i].y, cy, __builtin_fmaf(r[i].z, zc, r[i].w)));
#pragma unroll
    for (int i = 0; i < 4; ++i)
      q[i] = __builtin_fmaf(r[4 + i].x, pv[0], __builtin_fmaf(r[4 + i].y, pv[1], __builtin_fmaf(r[4 + i].z, pv[2], r[4 + i].w * pv[3])));
    // the range of q.w over the rectangle (affine in the centre: its corners' extremes)
    float wlo = q[3], whi = q[3];
    wlo = fminf(wlo, __builtin_amdgcn_update_dpp(wlo, wlo, 0xB1, 0xf, 0xf, false)); whi = fmaxf(whi, __builtin_amdgcn_update_dpp(whi, whi, 0xB1, 0xf, 0xf, false));
    wlo = fminf(wlo, __builtin_amdgcn_update_dpp(wlo, wlo, 0x4E, 0xf, 0xf, false)); whi = fmaxf(whi, __builtin_amdgcn_update_dpp(whi, whi, 0x4E, 0xf, 0xf, false));
    const float a00 = r[8].x, a01 = r[8].y, a03 = r[8].w, a10 = r[9].x, a11 = r[9].y, a13 = r[9].w;
    const bool plane = r[8].z == 0.f && r[9].z == 0.f && r[10].x == 0.f && r[10].y == 0.f && r[10].z == 1.f && r[10].w == 0.f;
    const float mu = P.u_div / (float) P.fW, mv = P.v_div / (float) P.fH;      // one feature pixel
    const float mz = P.use_depth ? P.d_span / (float) P.D : 0.01f;            // one depth bin (D == 1: a centimetre)
    const float zlo = P.use_depth ? P.d_lo : 0.f;
    const float u3lo = fminf(a03 * wlo, a03 * whi), u3hi = fmaxf(a03 * wlo, a03 * whi);
    const float v3lo = fminf(a13 * wlo, a13 * whi), v3hi = fmaxf(a13 * wlo, a13 * whi);
    const float lu = a00 * q[0] + a01 * q[1], lv = a10 * q[0] + a11 * q[1], qz = q[2];
    // this corner's side of each bound (lift_cull_test's o_near .. front), then the conjunction over the four corners
    unsigned f = (qz <= zlo - mz ? 1u : 0u) | ((P.use_depth != 0 && qz >= P.d_hi + mz) ? 2u : 0u) |
                 (lu + (u3hi + 0.5f + mu) * qz <= 0.f ? 4u : 0u) | (lu + (u3lo - P.u_max - mu) * qz >= 0.f ? 8u : 0u) |
                 (lv + (v3hi + 0.5f + mv) * qz <= 0.f ? 16u : 0u) | (lv + (v3lo - P.v_max - mv) * qz >= 0.f ? 32u : 0u) |
                 (qz >= 1.0f ? 64u : 0u);
    f &= (unsigned) __builtin_amdgcn_update_dpp((int) f, (int) f, 0xB1, 0xf, 0xf, false);
    f &= (unsigned) __builtin_amdgcn_update_dpp((int) f, (int) f, 0x4E, 0xf, 0xf, false);
    const bool sides = (P.use_depth && P.d_lo >= 1.0f) || (f & 64u);
    const bool out = plane && ((f & 3u) || (sides && (f & 60u)));
    unsigned diff = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      diff |= (__float_as_uint(r[i].x) ^ __float_as_uint(z4[i].x)) | (__float_as_uint(r[i].y) ^ __float_as_uint(z4[i].y)) |
              (__float_as_uint(r[i].z) ^ __float_as_uint(z4[i].z)) | (__float_as_uint(r[i].w) ^ __float_as_uint(z4[i].w));
    // camera n's verdict sits in lanes 4 n .. 4 n + 3: bit 4 n of a ballot -> bit n of the word (scalar)
    const uint64_t kb = __ballot(act && !out && k == 0), df = __ballot(act && diff != 0);
    unsigned kept = 0;
#pragma unroll
    for (int c = 0; c < kLiftCullMaxCams; ++c) kept |= (unsigned) ((kb >> (4 * c)) & 1ull) << c;
    if (lane == 0) word_s = kept | (df ? 0u : kLiftCullSharedBda);
  }
  __syncthreads();
  return (unsigned) __builtin_amdgcn_readfirstlane((int) word_s);
}

// stand-alone form (vamp_lift_cull_words): workgroup `blk` = sample * K.bps + chunk of 256 / G patches
__global__ void __launch_bounds__(256)
lift_cull_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                 const float* __restrict__ ys, const float* __restrict__ zs, LiftCull K) {
  __shared__ LiftCullLds S;
  const int b = blockIdx.x / K.bps, ppr = lift_cull_round(K);
  const long first = (long) (blockIdx.x % K.bps) * ppr;
  lift_cull_constants(P, mats, b, S);
  lift_cull_boxes(P, xs, ys, zs, K, first, ppr, S);
  __syncthreads();
  lift_cull_share(P, xs, ys, zs, K, b, first, ppr, S);
}

// One tile of the forward's first launch: 64 pixels x all channels of feat go channel-last, and -- in the
// shadow of those loads and behind the same barrier -- the workgroup's share of its sample's cull words.
// `r` = tile index in [0, B * N * ptiles); the N * ptiles tiles of a sample share its words evenly.
template <typename T>
__device__ __forceinline__ void lift_prologue_tile(const LiftParams& P, const T* __restrict__ feat,
                                                   float* __restrict__ feat_cl, int ptiles, unsigned r,
                                                   const float* __restrict__ mats, const float* __restrict__ xs,
                                                   const float* __restrict__ ys, const float* __restrict__ zs,
                                                   const LiftCull& K, float (&tile)[64][65], LiftCullLds& S) {
  const int C = P.C, HW = P.fH * P.fW;
  const long bn = r / ptiles;
  const int p0 = (r % ptiles) * 64, b = (int) (bn / P.N);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long first = (long) (r % ((unsigned) P.N * ptiles)) * K.ppt;
  for (int c = ty; c < C; c += 4) tile[c][tx] = (p0 + tx < HW) ? ldf(feat, (bn * C + c) * HW + p0 + tx) : 0.f;
  lift_cull_constants(P, mats, b, S);
  lift_cull_boxes(P, xs, ys, zs, K, first, K.ppt, S);
  __syncthreads();
  const int np = min(64, HW - p0);
  float* o = feat_cl + (bn * HW + p0) * C;
  if ((C & (C - 1)) == 0) {              // (no integer division on the way out: it costs more than the cull)
    const int sh = __builtin_ctz(C);
    for (int i = threadIdx.x; i < np * C; i += 256) o[i] = tile[i & (C - 1)][i >> sh];
  } else {
    for (int i = threadIdx.x; i < np * C; i += 256) o[i] = tile[i % C][i / C];
  }
  lift_cull_share(P, xs, ys, zs, K, b, first, K.ppt, S);
}

// The forward's first launch on the plain (depth, feat) entry: channel-last feature copy + cull words.
template <typename T>
__global__ void __launch_bounds__(256, 6)
lift_prologue_kernel(LiftParams P, const T* __restrict__ feat, float* __restrict__ feat_cl, int ptiles,
                     const float* __restrict__ mats, const float* __restrict__ xs,
                     const float* __restrict__ ys, const float* __restrict__ zs, LiftCull K) {
  __shared__ float tile[64][65];
  __shared__ LiftCullLds S;
  lift_prologue_tile<T>(P, feat, feat_cl, ptiles, blockIdx.x, mats, xs, ys, zs, K, tile, S);
}

// ---------------------------------------------------------------------------
// Both lift operands in one launch (the producer side): workgroups [0, n_sm) turn
// the raw `mapping_along_depth` logits into the depth distribution (softmax over D,
// base_vampire2.py:550), the rest make the channel-last copy of the features.  The two jobs are
// independent, so the 6 600 softmax tiles and the 1 100 transpose tiles of cfg-B share one grid.
// ---------------------------------------------------------------------------
template <typename TL, bool REG>
__global__ void __launch_bounds__(256, 6)      // (6 waves per SIMD, what the softmax tiles ran at before they shared the launch with the cull)
lift_operands_kernel(LiftParams P, const TL* __restrict__ logits, float* __restrict__ depth, int sm_tiles,
                     unsigned n_sm, const float* __restrict__ feat, float* __restrict__ feat_cl,
                     int ptiles, const float* __restrict__ mats, const float* __restrict__ xs,
                     const float* __restrict__ ys, const float* __restrict__ zs, LiftCull K) {
  __shared__ union {
    SoftmaxLds sm;
    float tile[64][65];
  } L;
  __shared__ LiftCullLds S;
  const long HW = (long) P.fH * P.fW;
  if (blockIdx.x < n_sm) {
    depth_softmax_tile<TL, REG>(logits, depth, P.D, HW, blockIdx.x / sm_tiles, blockIdx.x % sm_tiles, L.sm);
  } else {
    lift_prologue_tile<float>(P, feat, feat_cl, ptiles, blockIdx.x - n_sm, mats, xs, ys, zs, K, L.tile, S);
  }
}

// channel-last fp32 [BN, HW, C] -> [BN, C, HW]
__global__ void __launch_bounds__(256) feat_to_channel_first(const float* __restrict__ in,
                                                             float* __restrict__ out, int C, int HW) {
  __shared__ float tile[64][65];
  const long bn = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    int p = p0 + j, c = c0 + tx;
    tile[j][tx] = (p < HW && c < C) ? in[(bn * HW + p) * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    int c = c0 + j, p = p0 + tx;
    if (c < C && p < HW) out[(bn * C + c) * HW + p] = tile[tx][j];
  }
}

// ---------------------------------------------------------------------------
// depth interpolation at the four (h,w) taps:  dep[j] = sum_d w_d depth[d, iy, ix]
// (zero padding: out-of-range taps contribute nothing)
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void depth_taps(const LiftParams& P, const T* __restrict__ dptr,
                                           const LiftTap& t, float dep[4]) {
  dep[0] = dep[1] = dep[2] = dep[3] = 0.f;
  if (!P.use_depth) {
    // D == 1: the single plane has weight wz0 (= 1) at iz0 == 0
    const float w = (t.iz0 == 0 ? t.wz0 : 0.f) + (t.iz0 == -1 ? t.wz1 : 0.f);
    dep[0] = dep[1] = dep[2] = dep[3] = w;
    return;
  }
  // branch-free (clamped address, zero weight for taps outside the volume): the loads are
  // independent and issue together
  const long plane = (long) P.fH * P.fW;
  if constexpr (sizeof(T) == 4) {
    // fp32 planes: the two x taps of a row are neighbours in memory -- ONE 8-byte load (dword aligned, which
    // is all a multi-dword global load needs) instead of two: four line look-ups per pair instead of eight
    if (P.fW >= 2) {
      struct __attribute__((packed, aligned(4))) Pair { float a, b; };
      const int xb = min(max(t.ix0, 0), P.fW - 2);                 // the pair (xb, xb + 1) holds every in-range tap
      const bool x0in = t.ix0 >= 0 && t.ix0 < P.fW, x1in = t.ix0 + 1 >= 0 && t.ix0 + 1 < P.fW;
      const bool x0hi = t.ix0 != xb, x1lo = t.ix0 + 1 == xb;      // which half each tap is
#pragma unroll
      for (int kz = 0; kz < 2; ++kz) {
        const int iz = t.iz0 + kz;
        const bool zin = iz >= 0 && iz < P.D;
        const float wz = zin ? (kz ? t.wz1 : t.wz0) : 0.f;
        const int izc = min(max(iz, 0), P.D - 1);
#pragma unroll
        for (int ky = 0; ky < 2; ++ky) {
          const int iy = t.iy0 + ky;
          const bool yin = iy >= 0 && iy < P.fH;
          const Pair pr = *reinterpret_cast<const Pair*>(reinterpret_cast<const float*>(dptr) + izc * plane +
                                                        (long) min(max(iy, 0), P.fH - 1) * P.fW + xb);
          dep[ky * 2 + 0] += ((yin && x0in) ? wz : 0.f) * (x0hi ? pr.b : pr.a);
          dep[ky * 2 + 1] += ((yin && x1in) ? wz : 0.f) * (x1lo ? pr.a : pr.b);
        }
      }
      return;
    }
  }
  const bool x0in = t.ix0 >= 0 && t.ix0 < P.fW, x1in = t.ix0 + 1 >= 0 && t.ix0 + 1 < P.fW;
  const int x0c = min(max(t.ix0, 0), P.fW - 1), x1c = min(max(t.ix0 + 1, 0), P.fW - 1);
#pragma unroll
  for (int kz = 0; kz < 2; ++kz) {
    const int iz = t.iz0 + kz;
    const bool zin = iz >= 0 && iz < P.D;
    const float wz = zin ? (kz ? t.wz1 : t.wz0) : 0.f;
    const int izc = min(max(iz, 0), P.D - 1);
#pragma unroll
    for (int ky = 0; ky < 2; ++ky) {
      const int iy = t.iy0 + ky;
      const bool yin = iy >= 0 && iy < P.fH;
      const long row = izc * plane + (long) min(max(iy, 0), P.fH - 1) * P.fW;
      dep[ky * 2 + 0] += ((yin && x0in) ? wz : 0.f) * ldf(dptr, row + x0c);
      dep[ky * 2 + 1] += ((yin && x1in) ? wz : 0.f) * ldf(dptr, row + x1c);
    }
  }
}

// ---------------------------------------------------------------------------
// LIFT forward.  Block = TX*TY*TZ (= 256) voxels; CH channels per pass.
// ---------------------------------------------------------------------------
// EMIT (a backward will follow): the kernel also does the counting half of the backward's pixel
// sort -- every valid (voxel, camera) pair is counted in its cell and its taps are left in the
// workspace (lift_emit_pair), so nothing on the backward projects a voxel again.
template <typename T, int CH, int TX, int TY, int TZ, bool EMIT>
__global__ void __launch_bounds__(TX* TY* TZ, 4)      // (4 waves per SIMD: the EMIT variant would take 139 registers, i.e. 3: 48.5 -> 44.5 us; 5 spills: 55)
lift_fwd_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                const float* __restrict__ ys, const float* __restrict__ zs,
                const T* __restrict__ depth, const float* __restrict__ feat_cl,
                float* __restrict__ out, uint64_t* __restrict__ hits, LiftEmit E,
                const unsigned* __restrict__ cull) {
  static_assert(TZ == 1 && TX * TY == 256 && 64 % TX == 0, "a wave is a TX x (64 / TX) patch of one z plane");
  const int tid = threadIdx.x;
  const int x = blockIdx.x * TX + (tid % TX);
  const int y = blockIdx.y * TY + ((tid / TX) % TY);
  const int zblocks = (P.Z + TZ - 1) / TZ;
  const int b = blockIdx.z / zblocks;
  const int z = (blockIdx.z % zblocks) * TZ + tid / (TX * TY);
  // the wave's cull word (lift_cull_eval): one scalar load
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (the voxel's centre and its first product in front of the word: with the inline cull their loads are in flight
  // while the first lanes form it)
  const float vx = xs[min(x, P.X - 1)], vy = ys[min(y, P.Y - 1)], vz = zs[min(z, P.Z - 1)];
  const Vec4 p1 = matvec(mats + (long) b * P.N * 48, Vec4{vx, vy, vz, 1.0f});       // inv(bda) . c
  const unsigned cword = cull ? cull[((long) blockIdx.z * (gridDim.y * (4 / kCullWpw)) + blockIdx.y * (4 / kCullWpw) + wave / kCullWpw) * gridDim.x + blockIdx.x]
                              : lift_cull_inline(P, mats, xs, ys, zs, b, blockIdx.x * TX, blockIdx.y * TY, min(z, P.Z - 1));
  if (x >= P.X || y >= P.Y || z >= P.Z) return;

  const long V = (long) P.Z * P.Y * P.X;
  const long vox = ((long) z * P.Y + y) * P.X + x;
  const long HW = (long) P.fH * P.fW;
  const int nchunk = P.C / CH;

  unsigned wmask = 0;
  const bool shared_bda = (cword & kLiftCullSharedBda) != 0;                        // uniform
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    float sum[CH];
    uint64_t cnt = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) sum[c] = 0.f;

    for (unsigned cams = cword & (kLiftCullSharedBda - 1); cams; cams &= cams - 1) {
      const int n = __builtin_ctz(cams);
      const long bn = (long) b * P.N + n;
      const LiftTap t = lift_project_from<true>(P, mats + bn * 48, shared_bda ? p1 : matvec(mats + bn * 48, Vec4{vx, vy, vz, 1.0f}));
      bool pair = false;
      if (EMIT && chunk == 0) {      // (wave-uniform here: the lanes part ways at the next line)
        pair = lift_emit_pair(P, E, t, true, bn, V, vox, tid & 63);
        if (pair) wmask |= 1u << (n & 31);
      }
      if (!t.valid) continue;
      float dep[4];
      depth_taps<T>(P, depth + bn * P.D * HW, t, dep);
      const float w[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
      float acc[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
        const bool in = iy >= 0 && iy < P.fH && ix >= 0 && ix < P.fW;
        const float wd = in ? w[j] * dep[j] : 0.f;         // branch-free: clamp + zero weight
        const float4* f4 = reinterpret_cast<const float4*>(
            feat_cl + (bn * HW + (long) min(max(iy, 0), P.fH - 1) * P.fW + min(max(ix, 0), P.fW - 1)) * P.C +
            chunk * CH);
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const float4 f = f4[q];
          acc[q * 4 + 0] = __builtin_fmaf(wd, f.x, acc[q * 4 + 0]);
          acc[q * 4 + 1] = __builtin_fmaf(wd, f.y, acc[q * 4 + 1]);
          acc[q * 4 + 2] = __builtin_fmaf(wd, f.z, acc[q * 4 + 2]);
          acc[q * 4 + 3] = __builtin_fmaf(wd, f.w, acc[q * 4 + 3]);
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        sum[c] += acc[c];
        cnt += (uint64_t) (fabsf(acc[c]) > 0.f) << (4 * c);   // per-channel hit count (bv2:509)
      }
      if (EMIT && chunk == 0) lift_emit_dep(E, pair, bn, V, vox, dep);     // (behind the feature gather's loads)
    }
    float* o = out + ((long) b * P.C + chunk * CH) * V + vox;
    // the mean over the cameras that hit (bv2:509-514).  Unless a feature channel holds exact zeros the CH
    // counts of a voxel are one number: one division per lane gives its reciprocal, a product and one
    // residual step per channel the quotient (within an ulp of sum / denom; the outputs are held to 1e-4)
    const unsigned k0 = (unsigned) cnt & 15u;
    if (__all(cnt == (uint64_t) k0 * (0x1111111111111111ull >> (64 - 4 * CH)))) {
      const float denom = (float) k0 + 1e-6f, r = 1.0f / denom;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float q = sum[c] * r;
        o[(long) c * V] = __builtin_fmaf(__builtin_fmaf(-q, denom, sum[c]), r, q);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float denom = (float) ((cnt >> (4 * c)) & 15) + 1e-6f;
        o[(long) c * V] = sum[c] / denom;
      }
    }
    if (hits) hits[((long) b * V + vox) * nchunk + chunk] = cnt;
  }
  if (EMIT && E.amask) E.amask[(long) b * V + vox] = wmask;
}

// ---------------------------------------------------------------------------
// LIFT forward, C = 16, with a COOPERATIVE feature gather (round 5).
// In lift_fwd_kernel a lane fetches the 64-byte channel row of each of its four pixel taps as four 16-byte
// loads: 16 load instructions per camera whose 64 lanes touch 64 different lines each -- 1 024 line look-ups
// per wave and camera for 256 rows, and the ablation that fetched one piece per row (4 x fewer look-ups)
// was 8.7 us faster.  Here the wave works in two roles per camera: lane = voxel (projection, depth taps,
// the weights of its four pixel taps -> LDS), then lane = (voxel of a 16-voxel row of the patch, channel
// quad): the four lanes of a voxel fetch the four pieces of a row in ONE instruction, so a load touches 16
// lines, not 64 (same bytes, same number of instructions).  A lane then holds 4 channels of 4 voxels; sums,
// hit counts and the stores follow that layout (a store still writes 64-byte runs of 16 x-neighbours).
// The arithmetic per (voxel, camera, channel) is the old kernel's, in the same order: same bits.
// ---------------------------------------------------------------------------
// (waves per SIMD / rows in flight, cfg-B: 4 / 2: 29.2 us, 4 / 1: 29.3, 3 / 2: 38.3, 3 / 4: 31.7, 5 / 1: 48.0)
template <typename T>
__global__ void __launch_bounds__(256, 4)
lift_fwd_coop_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                     const float* __restrict__ ys, const float* __restrict__ zs,
                     const T* __restrict__ depth, const float* __restrict__ feat_cl,
                     float* __restrict__ out, uint64_t* __restrict__ hits, const unsigned* __restrict__ cull) {
  constexpr int TX = 16, TY = 16;                // a wave is a 16 x 4 patch: row g of it = 16 x-neighbours
  __shared__ float4 xch[4][64][2];               // per wave and voxel: the four taps' pixel indices | weights
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = blockIdx.x * TX + (tid % TX);
  const int y = blockIdx.y * TY + (tid / TX);
  const int b = blockIdx.z / P.Z, z = blockIdx.z % P.Z;
  // (the voxel's centre and its first product in front of the word: with the inline cull their loads are in flight
  // while the first lanes form it)
  const float vx = xs[min(x, P.X - 1)], vy = ys[min(y, P.Y - 1)], vz = zs[z];
  const Vec4 p1 = matvec(mats + (long) b * P.N * 48, Vec4{vx, vy, vz, 1.0f});       // inv(bda) . c
  const unsigned cword = cull ? cull[((long) blockIdx.z * (gridDim.y * (4 / kCullWpw)) + blockIdx.y * (4 / kCullWpw) + wave / kCullWpw) * gridDim.x + blockIdx.x]
                              : lift_cull_inline(P, mats, xs, ys, zs, b, blockIdx.x * TX, blockIdx.y * TY, z);
  // (no early exit: a lane whose own voxel is outside the grid still fetches for the others')
  const bool live = x < P.X && y < P.Y;
  const long V = (long) P.Z * P.Y * P.X;
  const long HW = (long) P.fH * P.fW;
  const int q = lane & 3, vq = lane >> 2;        // second role: channel quad, voxel within a row of the patch

  float sum[16];                                 // [row g of the patch][channel 4 q + k]
  uint64_t cnt = 0;                              // their hit counts, 4 bits each
#pragma unroll
  for (int i = 0; i < 16; ++i) sum[i] = 0.f;
  const bool shared_bda = (cword & kLiftCullSharedBda) != 0;                        // uniform

  for (unsigned cams = cword & (kLiftCullSharedBda - 1); cams; cams &= cams - 1) {
    const int n = __builtin_ctz(cams);
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project_from<true>(P, mats + bn * 48, shared_bda ? p1 : matvec(mats + bn * 48, Vec4{vx, vy, vz, 1.0f}));
    const bool valid = live && t.valid;
    if (!__any(valid)) continue;                 // uniform
    // ---- lane = voxel: the weights w_hw * (sum_d w_d depth) of the four pixel taps
    {
      float dep[4] = {0.f, 0.f, 0.f, 0.f};
      if (valid) depth_taps<T>(P, depth + bn * P.D * HW, t, dep);
      const float w[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
      int pix[4];
      float wd[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
        const bool in = valid && iy >= 0 && iy < P.fH && ix >= 0 && ix < P.fW;
        wd[j] = in ? w[j] * dep[j] : 0.f;
        pix[j] = in ? iy * P.fW + ix : -1;       // -1: nothing to fetch
      }
      xch[wave][lane][0] = make_float4(__int_as_float(pix[0]), __int_as_float(pix[1]), __int_as_float(pix[2]), __int_as_float(pix[3]));
      xch[wave][lane][1] = make_float4(wd[0], wd[1], wd[2], wd[3]);
    }
    // (one wave writes and reads its own slab.  The writes and the reads below touch the same words from different
    // lanes: a wave-level fence + barrier between them -- no instruction on gfx9, but it forbids the compiler to
    // reorder them should it ever prove the two index expressions distinct)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- lane = (voxel vq of row g, channel quad q): two rows of the patch at a time
    const float4* frow = reinterpret_cast<const float4*>(feat_cl) + bn * HW * 4 + q;
constexpr int R = 2;                          // rows of the patch whose loads are in flight together
#pragma unroll
    for (int g0 = 0; g0 < 4; g0 += R) {
      float4 px[R], wv[R], f[R][4];
#pragma unroll
      for (int h = 0; h < R; ++h) {
        px[h] = xch[wave][(g0 + h) * 16 + vq][0];
        wv[h] = xch[wave][(g0 + h) * 16 + vq][1];
      }
#pragma unroll
      for (int h = 0; h < R; ++h) {
        const int pj[4] = {__float_as_int(px[h].x), __float_as_int(px[h].y), __float_as_int(px[h].z), __float_as_int(px[h].w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[h][j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (pj[j] >= 0) f[h][j] = frow[(long) pj[j] * 4];
        }
      }
#pragma unroll
      for (int h = 0; h < R; ++h) {
        const float wj[4] = {wv[h].x, wv[h].y, wv[h].z, wv[h].w};
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0] = __builtin_fmaf(wj[j], f[h][j].x, acc[0]);
          acc[1] = __builtin_fmaf(wj[j], f[h][j].y, acc[1]);
          acc[2] = __builtin_fmaf(wj[j], f[h][j].z, acc[2]);
          acc[3] = __builtin_fmaf(wj[j], f[h][j].w, acc[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = (g0 + h) * 4 + k;
          sum[i] += acc[k];
          cnt += (uint64_t) (fabsf(acc[k]) > 0.f) << (4 * i);   // per-channel hit count (bv2:509)
        }
      }
    }
    // (the next camera overwrites the slab: its writes stay behind these reads)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- the mean over the cameras that hit (bv2:509-514) and the stores, row by row of the patch
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int xo = blockIdx.x * TX + vq, yo = blockIdx.y * TY + wave * 4 + g;
    const bool live_o = xo < P.X && yo < P.Y;
    const long vo = ((long) z * P.Y + min(yo, P.Y - 1)) * P.X + min(xo, P.X - 1);
    const unsigned c16 = (unsigned) (cnt >> (16 * g)) & 0xffffu;     // this lane's four counts of the row's voxel
    const unsigned k0 = c16 & 15u;
    float* o = out + ((long) b * P.C + q * 4) * V + vo;
    if (__all(c16 == k0 * 0x1111u)) {
      // (one count per voxel unless a feature channel holds exact zeros: one division, then a product and a
      // residual step per channel -- within an ulp of sum / denom; the outputs are held to 1e-4)
      const float denom = (float) k0 + 1e-6f, r = 1.0f / denom;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float qv = sum[g * 4 + k] * r;
        if (live_o) o[(long) k * V] = __builtin_fmaf(__builtin_fmaf(-qv, denom, sum[g * 4 + k]), r, qv);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (live_o) o[(long) k * V] = sum[g * 4 + k] / ((float) ((c16 >> (4 * k)) & 15u) + 1e-6f);
    }
    if (hits) {
      // the voxel's word: channel c's count in nibble c = the four lanes' 16-bit pieces side by side
      unsigned lo = q < 2 ? c16 << (16 * q) : 0u, hi = q >= 2 ? c16 << (16 * (q - 2)) : 0u;
      lo |= __shfl_xor(lo, 1, 64); hi |= __shfl_xor(hi, 1, 64);
      lo |= __shfl_xor(lo, 2, 64); hi |= __shfl_xor(hi, 2, 64);
      if (q == 0 && live_o) hits[(long) b * V + vo] = ((uint64_t) hi << 32) | lo;
    }
  }
}

// The same walk without the samples: counts and emits the pairs for a backward whose forward did
// not (vamp_lift_prepare; vamp_lift_backward without VAMP_LIFTBWD_CELLS_VALID).
template <typename T, int TX, int TY>
__global__ void __launch_bounds__(TX* TY)
lift_pairs_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                  const float* __restrict__ ys, const float* __restrict__ zs, const T* __restrict__ depth, LiftEmit E) {
  const int tid = threadIdx.x;
  const int x = blockIdx.x * TX + (tid % TX);
  const int y = blockIdx.y * TY + (tid / TX);
  const int b = blockIdx.z / P.Z, z = blockIdx.z % P.Z;
  if (x >= P.X || y >= P.Y) return;
  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long V = (long) P.Z * P.Y * P.X;
  const long HW = (long) P.fH * P.fW;
  const long vox = ((long) z * P.Y + y) * P.X + x;
  unsigned wmask = 0;
  for (int n = 0; n < P.N; ++n) {
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project<true>(P, mats + bn * 48, vx, vy, vz);
    float dep[4] = {0.f, 0.f, 0.f, 0.f};
    if (t.valid) depth_taps<T>(P, depth + bn * P.D * HW, t, dep);
    const bool pair = lift_emit_pair(P, E, t, true, bn, V, vox, tid & 63);
    if (pair) wmask |= 1u << (n & 31);
    lift_emit_dep(E, pair, bn, V, vox, dep);
  }
  if (E.amask) E.amask[(long) b * V + vox] = wmask;
}

// ---------------------------------------------------------------------------
// LIFT forward from the materialised frustum_feats [B,N,C,D,fH,fW]
// (signature-compatible path for get_voxel_feats, bv2:483).  8-tap gather per
// channel in aten's tap order.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
lift_fwd_dense_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                      const float* __restrict__ ys, const float* __restrict__ zs,
                      const float* __restrict__ ff, float* __restrict__ out,
                      uint64_t* __restrict__ hits) {
  const long V = (long) P.Z * P.Y * P.X;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * P.B) return;
  const int b = gid / V;
  const long vox = gid % V;
  const int x = vox % P.X, y = (vox / P.X) % P.Y, z = vox / ((long) P.X * P.Y);
  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long HW = (long) P.fH * P.fW, DHW = HW * P.D;
  const int nchunk = (P.C + 15) / 16;

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int c_lo = chunk * 16, c_n = min(16, P.C - c_lo);
    float sum[16];
    uint64_t cnt = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) sum[c] = 0.f;
    for (int n = 0; n < P.N; ++n) {
      const long bn = (long) b * P.N + n;
      const LiftTap t = lift_project(P, mats + bn * 48, vx, vy, vz);
      if (!t.valid) continue;
      float acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = t.iz0 + (k >> 2), iy = t.iy0 + ((k >> 1) & 1), ix = t.ix0 + (k & 1);
        if (iz < 0 || iz >= P.D || iy < 0 || iy >= P.fH || ix < 0 || ix >= P.fW) continue;
        const float w = ((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0) *
                        ((k & 4) ? t.wz1 : t.wz0);
        const float* src = ff + (bn * P.C + c_lo) * DHW + iz * HW + (long) iy * P.fW + ix;
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c < c_n) acc[c] += src[c * DHW] * w;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        sum[c] += acc[c];
        cnt += (uint64_t) (fabsf(acc[c]) > 0.f) << (4 * c);
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < c_n)
        out[((long) b * P.C + c_lo + c) * V + vox] =
            sum[c] / ((float) ((cnt >> (4 * c)) & 15) + 1e-6f);
    if (hits) hits[((long) b * V + vox) * nchunk + chunk] = cnt;
  }
}

// ---------------------------------------------------------------------------
// LIFT backward (v1): one thread per voxel, float atomics into channel-last
// grad_feat and into grad_depth.
// ---------------------------------------------------------------------------
template <typename T, int CH, int TX, int TY, int TZ>
__global__ void __launch_bounds__(TX* TY* TZ)
lift_bwd_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                const float* __restrict__ ys, const float* __restrict__ zs,
                const T* __restrict__ depth, const float* __restrict__ feat_cl,
                const float* __restrict__ gout, const uint64_t* __restrict__ hits,
                float* __restrict__ gdepth, float* __restrict__ gfeat_cl) {
  const int tid = threadIdx.x;
  const int x = blockIdx.x * TX + (tid % TX);
  const int y = blockIdx.y * TY + ((tid / TX) % TY);
  const int zblocks = (P.Z + TZ - 1) / TZ;
  const int b = blockIdx.z / zblocks;
  const int z = (blockIdx.z % zblocks) * TZ + tid / (TX * TY);
  if (x >= P.X || y >= P.Y || z >= P.Z) return;

  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long V = (long) P.Z * P.Y * P.X;
  const long vox = ((long) z * P.Y + y) * P.X + x;
  const long HW = (long) P.fH * P.fW;
  const int nchunk = P.C / CH;

  for (int n = 0; n < P.N; ++n) {
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project(P, mats + bn * 48, vx, vy, vz);
    if (!t.valid) continue;
    float dep[4];
    depth_taps<T>(P, depth + bn * P.D * HW, t, dep);
    const float w[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
    float gdep[4] = {0.f, 0.f, 0.f, 0.f};   // d loss / d dep[j]
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const uint64_t cnt = hits[((long) b * V + vox) * nchunk + chunk];
      float gs[CH];
      const float* g = gout + ((long) b * P.C + chunk * CH) * V + vox;
#pragma unroll
      for (int c = 0; c < CH; ++c)
        gs[c] = g[(long) c * V] / ((float) ((cnt >> (4 * c)) & 15) + 1e-6f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
        if (iy < 0 || iy >= P.fH || ix < 0 || ix >= P.fW) continue;
        const long pix = (bn * HW + (long) iy * P.fW + ix) * P.C + chunk * CH;
        const float4* f4 = reinterpret_cast<const float4*>(feat_cl + pix);
        const float wd = w[j] * dep[j];
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const float4 f = f4[q];
          dot = __builtin_fmaf(f.x, gs[q * 4 + 0], dot);
          dot = __builtin_fmaf(f.y, gs[q * 4 + 1], dot);
          dot = __builtin_fmaf(f.z, gs[q * 4 + 2], dot);
          dot = __builtin_fmaf(f.w, gs[q * 4 + 3], dot);
        }
        gdep[j] = __builtin_fmaf(w[j], dot, gdep[j]);
#pragma unroll
        for (int c = 0; c < CH; ++c) atomicAdd(gfeat_cl + pix + c, wd * gs[c]);
      }
    }
    if (P.use_depth && gdepth) {
      float* gd = gdepth + bn * P.D * HW;
#pragma unroll
      for (int kz = 0; kz < 2; ++kz) {
        const int iz = t.iz0 + kz;
        if (iz < 0 || iz >= P.D) continue;
        const float wz = kz ? t.wz1 : t.wz0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int iy = t.iy0 + (j >> 1), ix = t.ix0 + (j & 1);
          if (iy < 0 || iy >= P.fH || ix < 0 || ix >= P.fW) continue;
          atomicAdd(gd + iz * HW + (long) iy * P.fW + ix, wz * gdep[j]);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
lift_bwd_dense_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                      const float* __restrict__ ys, const float* __restrict__ zs,
                      const float* __restrict__ gout, const uint64_t* __restrict__ hits,
                      float* __restrict__ gff) {
  const long V = (long) P.Z * P.Y * P.X;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * P.B) return;
  const int b = gid / V;
  const long vox = gid % V;
  const int x = vox % P.X, y = (vox / P.X) % P.Y, z = vox / ((long) P.X * P.Y);
  const float vx = xs[x], vy = ys[y], vz = zs[z];
  const long HW = (long) P.fH * P.fW, DHW = HW * P.D;
  const int nchunk = (P.C + 15) / 16;
  for (int n = 0; n < P.N; ++n) {
    const long bn = (long) b * P.N + n;
    const LiftTap t = lift_project(P, mats + bn * 48, vx, vy, vz);
    if (!t.valid) continue;
    for (int c = 0; c < P.C; ++c) {
      const uint64_t cnt = hits[((long) b * V + vox) * nchunk + (c >> 4)];
      const float gs = gout[((long) b * P.C + c) * V + vox] /
                       ((float) ((cnt >> (4 * (c & 15))) & 15) + 1e-6f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = t.iz0 + (k >> 2), iy = t.iy0 + ((k >> 1) & 1), ix = t.ix0 + (k & 1);
        if (iz < 0 || iz >= P.D || iy < 0 || iy >= P.fH || ix < 0 || ix >= P.fW) continue;
        const float w = ((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0) *
                        ((k & 4) ? t.wz1 : t.wz0);
        atomicAdd(gff + (bn * P.C + c) * DHW + iz * HW + (long) iy * P.fW + ix, w * gs);
      }
    }
  }
}

__global__ void __launch_bounds__(256)
lift_indices_kernel(LiftParams P, const float* __restrict__ mats, const float* __restrict__ xs,
                    const float* __restrict__ ys, const float* __restrict__ zs,
                    uint8_t* __restrict__ valid, int16_t* __restrict__ ix0,
                    int16_t* __restrict__ iy0, int16_t* __restrict__ iz0) {
  const long V = (long) P.Z * P.Y * P.X;
  const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= V * P.B * P.N) return;
  const long bn = gid / V;
  const long vox = gid % V;
  const int x = vox % P.X, y = (vox / P.X) % P.Y, z = vox / ((long) P.X * P.Y);
  const LiftTap t = lift_project(P, mats + bn * 48, xs[x], ys[y], zs[z]);
  valid[gid] = t.valid ? 1 : 0;
  ix0[gid] = (int16_t) t.ix0;
  iy0[gid] = (int16_t) t.iy0;
  iz0[gid] = (int16_t) t.iz0;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int validate(const VampLiftDesc* d) {
  VAMP_REQUIRE(d != nullptr, "desc is NULL");
  VAMP_REQUIRE(d->B > 0 && d->N > 0 && d->C > 0, "B, N, C must be positive");
  VAMP_REQUIRE(d->D > 0 && d->fH > 0 && d->fW > 0, "D, fH, fW must be positive");
  VAMP_REQUIRE(d->Z > 0 && d->Y > 0 && d->X > 0, "Z, Y, X must be positive");
  VAMP_REQUIRE(d->X < 32768 && d->Y < 32768 && d->fW < 32768 && d->fH < 32768,
               "axis too long for int16 taps");
  VAMP_REQUIRE(d->N <= 15, "at most 15 cameras (4-bit hit counters)");
  VAMP_REQUIRE(d->in_dtype == VAMP_F32 || d->in_dtype == VAMP_BF16, "in_dtype");
  VAMP_REQUIRE(d->use_depth == 1 || d->D == 1, "use_depth == 0 requires D == 1");
  return VAMP_OK;
}

static bool fused_channels_ok(int C) { return C == 4 || C == 8 || (C % 16 == 0 && C <= 64); }

template <typename T>
static void launch_to_cl(const void* feat, float* out, int BN, int C, int HW, hipStream_t s) {
  dim3 grid((HW + 63) / 64, (C + 63) / 64, BN);
  VAMP_TIMED(kProfFeatCL, s, (feat_to_channel_last<T><<<grid, 256, 0, s>>>(static_cast<const T*>(feat), out, C, HW)));
}


struct LiftWs {
  float* feat_cl;     // [B*N, HW, C]
  float* gfeat_cl;    // [B*N, HW, C] (v1 backward only)
  void* cells;        // cell lists of the backward; vamp_lift_prepare fills their offsets
  LiftCull cull;      // camera cull words of the forward's waves
  long ncull;
  size_t bytes;
};

static LiftWs carve(const VampLiftDesc* d, void* ws) {
  LiftWs w;
  const size_t n = align_up((size_t) d->B * d->N * d->fH * d->fW * d->C * sizeof(float), 256);
  w.feat_cl = static_cast<float*>(ws);
  w.gfeat_cl = reinterpret_cast<float*>(static_cast<char*>(ws) + n);
  // the cell lists live behind the copies: the forward must not disturb prepared offsets
  w.cells = static_cast<char*>(ws) + 2 * n;
  const size_t ncell_bytes = align_up(lift_bwd_cell_ws_bytes(d), 256);
  w.cull.words = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + 2 * n + ncell_bytes);
  w.cull.px = VAMP_LIFT_TX;
  w.cull.py = kCullPY;
  w.cull.nxp = (d->X + VAMP_LIFT_TX - 1) / VAMP_LIFT_TX;
  w.cull.nyp = (d->Y + VAMP_LIFT_TY - 1) / VAMP_LIFT_TY * (4 / kCullWpw);
  w.ncull = (long) d->B * d->Z * w.cull.nyp * w.cull.nxp;
  w.cull.group = 1;
  while (w.cull.group < d->N) w.cull.group *= 2;
  const int ppb = std::min(256 / w.cull.group, 64);
  const long per_sample = (long) d->Z * w.cull.nyp * w.cull.nxp;
  w.cull.bps = (int) ((per_sample + ppb - 1) / ppb);
  const long tiles = (long) d->N * (((long) d->fH * d->fW + 63) / 64);
  w.cull.ppt = (int) ((per_sample + tiles - 1) / tiles);
  w.bytes = 2 * n + ncell_bytes + align_up((size_t) w.ncull * sizeof(unsigned), 256);
  return w;
}


// `cells` != nullptr: the kernel emits the backward's pairs into that part of the workspace, between
// the zero fill of the cell counters and their scan (afterwards the workspace is what
// vamp_lift_prepare leaves: VAMP_LIFTBWD_CELLS_VALID)
template <typename T>
static int lift_forward_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                          const float* xs, const float* ys, const float* zs, const void* depth,
                          const float* feat_cl, float* out, uint64_t* hits, void* cells, bool cells_clean,
                          const unsigned* cull, hipStream_t s, bool defer_scan = false) {
  constexpr int TX = VAMP_LIFT_TX, TY = VAMP_LIFT_TY, TZ = 1;
  dim3 grid((P.X + TX - 1) / TX, (P.Y + TY - 1) / TY, ((P.Z + TZ - 1) / TZ) * P.B);
  const T* dp = static_cast<const T*>(depth);
  LiftEmit E{};
  if (cells) {
    VAMP_REQUIRE(d->N <= 15, "at most 15 cameras");
    if (int e = launch_lift_cells_begin(d, cells, s, cells_clean)) return e;
    E = lift_emit_of(d, cells);
  }
#define VAMP_FWD(CH, EM)                                                                               \
  VAMP_TIMED(kProfLiftFwd, s, (lift_fwd_kernel<T, CH, VAMP_LIFT_TILE, EM><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, out, hits, E, cull)))
  // C = 16 (the reference's mid_channels), no backward to follow: the kernel with the cooperative feature gather
  // (cfg-B: 28.6 against 31.8 us.  With the pair emission of a training forward its registers spill and it
  // loses, 44.3 against 36.3 -- those calls keep lift_fwd_kernel)
  const bool coop = P.C == 16 && TX == 16 && TY == 16 && VAMP_LIFT_COOP && !cells;
  if (coop)
    VAMP_TIMED(kProfLiftFwd, s, (lift_fwd_coop_kernel<T><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, out, hits, cull)));
  else if (cells) {
    if (P.C == 4) VAMP_FWD(4, true); else if (P.C == 8) VAMP_FWD(8, true); else VAMP_FWD(16, true);
  } else {
    if (P.C == 4) VAMP_FWD(4, false); else if (P.C == 8) VAMP_FWD(8, false); else VAMP_FWD(16, false);
  }
#undef VAMP_FWD
  if (int e = check_launch("lift_fwd_kernel")) return e;
  // (VAMP_LIFTFWD_DEFER_SCAN: the counters stay as counted; vamp_lift_finish_cells or
  // vamp_render_camera_prepare_with_lift scans them)
  return (cells && !defer_scan) ? launch_lift_cells_end(d, cells, s) : VAMP_OK;
}

int launch_lift_cell_prepare(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, const void* depth, void* scratch, hipStream_t s) {
  const LiftParams P = to_params(d);
  if (int e = launch_lift_cells_begin(d, scratch, s)) return e;
  const LiftEmit E = lift_emit_of(d, scratch);
  constexpr int TX = VAMP_LIFT_TX, TY = VAMP_LIFT_TY;
  dim3 grid((P.X + TX - 1) / TX, (P.Y + TY - 1) / TY, P.Z * P.B);
  if (d->in_dtype == VAMP_F32)
    VAMP_TIMED(kProfLiftBwdCount, s, (lift_pairs_kernel<float, TX, TY><<<grid, TX * TY, 0, s>>>(P, mats, xs, ys, zs, static_cast<const float*>(depth), E)));
  else
    VAMP_TIMED(kProfLiftBwdCount, s, (lift_pairs_kernel<__hip_bfloat16, TX, TY><<<grid, TX * TY, 0, s>>>(P, mats, xs, ys, zs, static_cast<const __hip_bfloat16*>(depth), E)));
  if (int e = check_launch("lift_pairs_kernel")) return e;
  return launch_lift_cells_end(d, scratch, s);
}

template <typename T>
static int lift_backward_t(const VampLiftDesc* d, const LiftParams& P, const float* mats,
                           const float* xs, const float* ys, const float* zs, const void* depth,
                           const float* feat_cl, const float* gout, const uint64_t* hits,
                           float* gdepth, float* gfeat_cl, hipStream_t s) {
  constexpr int TX = VAMP_LIFT_TX, TY = VAMP_LIFT_TY, TZ = 1;
  dim3 grid((P.X + TX - 1) / TX, (P.Y + TY - 1) / TY, ((P.Z + TZ - 1) / TZ) * P.B);
  const T* dp = static_cast<const T*>(depth);
  if (P.C == 4)
    VAMP_TIMED(kProfLiftBwdV1, s, (lift_bwd_kernel<T, 4, VAMP_LIFT_TILE><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, gout, hits, gdepth, gfeat_cl)));
  else if (P.C == 8)
    VAMP_TIMED(kProfLiftBwdV1, s, (lift_bwd_kernel<T, 8, VAMP_LIFT_TILE><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, gout, hits, gdepth, gfeat_cl)));
  else
    VAMP_TIMED(kProfLiftBwdV1, s, (lift_bwd_kernel<T, 16, VAMP_LIFT_TILE><<<grid, 256, 0, s>>>(P, mats, xs, ys, zs, dp, feat_cl, gout, hits, gdepth, gfeat_cl)));
  return check_launch("lift_bwd_kernel");
}

}  // namespace vamp

using namespace vamp;

extern "C" {

size_t vamp_lift_workspace_bytes(const VampLiftDesc* d) {
  if (!d) return 0;
  return carve(d, nullptr).bytes;
}

int vamp_lift_forward(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                      const float* zs, const void* depth, const void* feat, float* out,
                      uint64_t* hits, void* workspace, size_t workspace_bytes, void* stream) {
  return vamp_lift_forward_ex(d, mats, xs, ys, zs, depth, feat, out, hits, workspace, workspace_bytes, 0, stream);
}

int vamp_lift_forward_ex(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, const void* depth, const void* feat, float* out,
                         uint64_t* hits, void* workspace, size_t workspace_bytes, int flags, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && feat && out, "null pointer");
  VAMP_REQUIRE(((uintptr_t) mats & 15) == 0, "mats must be 16-byte aligned (the cull reads whole matrix rows)");
  VAMP_REQUIRE(depth || !d->use_depth, "depth is NULL");
  VAMP_REQUIRE(fused_channels_ok(d->C), "C must be 4, 8 or a multiple of 16 (<= 64)");
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const LiftParams P = to_params(d);
  const long BN = (long) d->B * d->N, HW = (long) d->fH * d->fW;
  void* cells = (flags & VAMP_LIFTFWD_EMIT_PAIRS) ? w.cells : nullptr;
  const bool clean = (flags & VAMP_LIFTFWD_CELLS_CLEAN) != 0;
  const bool defer = (flags & VAMP_LIFTFWD_DEFER_SCAN) != 0;
  if (flags & VAMP_LIFTFWD_FEAT_CHANNEL_LAST) {
    // the features are already [B, N, fH, fW, C] fp32: no copy, and every workgroup of the forward kernel forms
    // its own patch's cull word at its head -- the forward is ONE launch
    VAMP_REQUIRE(d->in_dtype == VAMP_F32, "VAMP_LIFTFWD_FEAT_CHANNEL_LAST takes fp32 features (and depth)");
    VAMP_REQUIRE(((uintptr_t) feat & 15) == 0, "channel-last feat must be 16-byte aligned");
    return lift_forward_t<float>(d, P, mats, xs, ys, zs, depth, static_cast<const float*>(feat), out, hits, cells, clean,
                                 nullptr, s, defer);
  }
  // first launch: channel-last copy of the features + the camera cull words of the forward's waves
  const int ptiles = (int) ((HW + 63) / 64);
  const long n_cl = BN * ptiles;
  VAMP_REQUIRE(n_cl < 0x7fffffffL, "too many tiles");
  const unsigned grid = (unsigned) n_cl;
  if (d->in_dtype == VAMP_F32)
    VAMP_TIMED(kProfFeatCL, s, (lift_prologue_kernel<float><<<grid, 256, 0, s>>>(P, static_cast<const float*>(feat), w.feat_cl, ptiles, mats, xs, ys, zs, w.cull)));
  else
    VAMP_TIMED(kProfFeatCL, s, (lift_prologue_kernel<__hip_bfloat16><<<grid, 256, 0, s>>>(P, static_cast<const __hip_bfloat16*>(feat), w.feat_cl, ptiles, mats, xs, ys, zs, w.cull)));
  if (int e = check_launch("lift_prologue_kernel")) return e;
  if (d->in_dtype == VAMP_F32)
    return lift_forward_t<float>(d, P, mats, xs, ys, zs, depth, w.feat_cl, out, hits, cells, clean, w.cull.words, s, defer);
  return lift_forward_t<__hip_bfloat16>(d, P, mats, xs, ys, zs, depth, w.feat_cl, out, hits, cells, clean, w.cull.words, s, defer);
}

int vamp_lift_forward_logits(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                             const float* zs, const void* logits, int32_t logits_dtype, const float* feat,
                             float* depth_out, float* out, uint64_t* hits, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return vamp_lift_forward_logits_ex(d, mats, xs, ys, zs, logits, logits_dtype, feat, depth_out, out, hits,
                                     workspace, workspace_bytes, 0, stream);
}

int vamp_lift_forward_logits_ex(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                                const float* zs, const void* logits, int32_t logits_dtype, const float* feat,
                                float* depth_out, float* out, uint64_t* hits, void* workspace,
                                size_t workspace_bytes, int flags, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && logits && feat && depth_out && out, "null pointer");
  VAMP_REQUIRE(((uintptr_t) mats & 15) == 0, "mats must be 16-byte aligned (the cull reads whole matrix rows)");
  VAMP_REQUIRE(d->use_depth == 1, "the logits entry is the depth-distribution lift");
  VAMP_REQUIRE(d->in_dtype == VAMP_F32, "feat (and the depth distribution written here) are fp32");
  VAMP_REQUIRE(logits_dtype == VAMP_F32 || logits_dtype == VAMP_BF16, "logits_dtype");
  VAMP_REQUIRE(fused_channels_ok(d->C), "C must be 4, 8 or a multiple of 16 (<= 64)");
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const LiftParams P = to_params(d);
  const long BN = (long) d->B * d->N, HW = (long) d->fH * d->fW;
  const long sm_tiles = (HW + kPix - 1) / kPix;
  const int ptiles = (int) ((HW + 63) / 64);
  // VAMP_LIFTFWD_FEAT_CHANNEL_LAST: only the softmax tiles; the forward kernel reads the caller's features and forms
  // its cull words itself
  const bool fcl = (flags & VAMP_LIFTFWD_FEAT_CHANNEL_LAST) != 0;
  VAMP_REQUIRE(!fcl || ((uintptr_t) feat & 15) == 0, "channel-last feat must be 16-byte aligned");
  const long n_sm = BN * sm_tiles, n_cl = fcl ? 0 : BN * ptiles;
  VAMP_REQUIRE(n_sm + n_cl < 0x7fffffffL, "too many tiles");
  const unsigned grid = (unsigned) (n_sm + n_cl);
#define VAMP_OPERANDS(TL, REG)                                                                           \
  VAMP_TIMED(kProfFeatCL, s, (lift_operands_kernel<TL, REG><<<grid, 256, 0, s>>>(                        \
      P, static_cast<const TL*>(logits), depth_out, (int) sm_tiles, (unsigned) n_sm, feat, w.feat_cl,    \
      ptiles, mats, xs, ys, zs, w.cull)))
  const bool reg = d->D <= kSplit * kRegBins;
  if (logits_dtype == VAMP_F32) {
    if (reg) VAMP_OPERANDS(float, true); else VAMP_OPERANDS(float, false);
  } else {
    if (reg) VAMP_OPERANDS(__hip_bfloat16, true); else VAMP_OPERANDS(__hip_bfloat16, false);
  }
#undef VAMP_OPERANDS
  if (int e = check_launch("lift_operands_kernel")) return e;
  return lift_forward_t<float>(d, P, mats, xs, ys, zs, depth_out, fcl ? feat : w.feat_cl, out, hits,
                               (flags & VAMP_LIFTFWD_EMIT_PAIRS) ? w.cells : nullptr,
                               (flags & VAMP_LIFTFWD_CELLS_CLEAN) != 0, fcl ? nullptr : w.cull.words, s,
                               (flags & VAMP_LIFTFWD_DEFER_SCAN) != 0);
}

// the scan a forward with VAMP_LIFTFWD_DEFER_SCAN left out
int vamp_lift_finish_cells(const VampLiftDesc* d, void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = validate(d)) return e;
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  return launch_lift_cells_end(d, w.cells, static_cast<hipStream_t>(stream));
}

// the same scan as a job for a launch shared with another cell list (render_bwd.hip: vamp_render_camera_prepare_with_lift)
int lift_scan_job(const VampLiftDesc* d, void* workspace, size_t workspace_bytes, ScanJob* job) {
  if (int e = validate(d)) return e;
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  return lift_cells_scan_job(d, w.cells, job);
}

int vamp_lift_prepare(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                      const float* zs, const void* depth, void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs, "null pointer");
  VAMP_REQUIRE(depth || !d->use_depth, "depth is NULL");
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  return launch_lift_cell_prepare(d, mats, xs, ys, zs, depth, w.cells, static_cast<hipStream_t>(stream));
}

int vamp_lift_backward(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                       const float* zs, const void* depth, const void* feat,
                       const float* grad_out, const uint64_t* hits, float* grad_depth,
                       float* grad_feat, void* workspace, size_t workspace_bytes, void* stream) {
  return vamp_lift_backward_ex(d, mats, xs, ys, zs, depth, feat, grad_out, hits, grad_depth, grad_feat,
                               workspace, workspace_bytes, 0, stream);
}

int vamp_lift_backward_ex(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                          const float* zs, const void* depth, const void* feat,
                          const float* grad_out, const uint64_t* hits, float* grad_depth,
                          float* grad_feat, void* workspace, size_t workspace_bytes, int flags,
                          void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && feat && grad_out && hits && grad_feat, "null pointer");
  VAMP_REQUIRE((depth && grad_depth) || !d->use_depth, "depth / grad_depth is NULL");
  VAMP_REQUIRE(fused_channels_ok(d->C), "C must be 4, 8 or a multiple of 16 (<= 64)");
  const LiftWs w = carve(d, workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(VAMP_ENOSPC, "%s: workspace %ld < %ld bytes", __func__, (long) workspace_bytes, (long) w.bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const LiftParams P = to_params(d);
  const int BN = d->B * d->N, HW = d->fH * d->fW;
  // VAMP_LIFTBWD_FEAT_CHANNEL_LAST: feat is read, and grad_feat written, as [B, N, fH, fW, C] fp32
  const bool fcl = (flags & VAMP_LIFTBWD_FEAT_CHANNEL_LAST) != 0;
  VAMP_REQUIRE(!fcl || d->in_dtype == VAMP_F32, "VAMP_LIFTBWD_FEAT_CHANNEL_LAST takes fp32 features (and depth)");
  // default: cell list + one wave per pixel (lift_bwd_cell.hip), no float atomics.
  // VAMP_LIFTBWD_SPLAT selects the per-voxel atomic splat below, kept as an independent cross-check.
  if (!(flags & VAMP_LIFTBWD_SPLAT)) {
    // gather variants (same kernel, smaller record chunks: the tests run them to cross the chunk
    // boundaries at every size)
    const int wpp = (flags & VAMP_LIFTBWD_WPP1) ? 1 : (flags & VAMP_LIFTBWD_WPP4) ? 4
                    : (flags & VAMP_LIFTBWD_WPP16) ? 16 : 0;
    return launch_lift_bwd_cell(d, mats, xs, ys, zs, depth, feat, grad_out, hits, grad_depth,
                                grad_feat, w.cells, (flags & VAMP_LIFTBWD_CELLS_VALID) != 0, wpp, /*half=*/0,
                                (flags & VAMP_LIFTBWD_LOGITS) != 0, fcl, s);
  }
  VAMP_REQUIRE(!(flags & VAMP_LIFTBWD_LOGITS), "VAMP_LIFTBWD_LOGITS is a feature of the default (cell-list) backward");
  // (channel-last features: they are what the splat reads, and the gradient it accumulates is what the caller wants)
  const float* fsrc = fcl ? static_cast<const float*>(feat) : w.feat_cl;
  float* gdst = fcl ? grad_feat : w.gfeat_cl;
  if (!fcl) {
    if (d->in_dtype == VAMP_F32) launch_to_cl<float>(feat, w.feat_cl, BN, d->C, HW, s);
    else launch_to_cl<__hip_bfloat16>(feat, w.feat_cl, BN, d->C, HW, s);
    if (int e = check_launch("feat_to_channel_last")) return e;
  }
  if (int ze = launch_zero(gdst, (size_t) BN * HW * d->C * sizeof(float), s)) return ze;
  if (d->use_depth)
    if (int ze = launch_zero(grad_depth, (size_t) BN * d->D * HW * sizeof(float), s)) return ze;
  int e = (d->in_dtype == VAMP_F32)
              ? lift_backward_t<float>(d, P, mats, xs, ys, zs, depth, fsrc, grad_out, hits, grad_depth, gdst, s)
              : lift_backward_t<__hip_bfloat16>(d, P, mats, xs, ys, zs, depth, fsrc, grad_out, hits, grad_depth, gdst, s);
  if (e || fcl) return e;
  dim3 grid((HW + 63) / 64, (d->C + 63) / 64, BN);
  VAMP_TIMED(kProfFeatCF, s, (feat_to_channel_first<<<grid, 256, 0, s>>>(w.gfeat_cl, grad_feat, d->C, HW)));
  return check_launch("feat_to_channel_first");
}

int vamp_lift_forward_dense(const VampLiftDesc* d, const float* mats, const float* xs,
                            const float* ys, const float* zs, const float* frustum_feats,
                            float* out, uint64_t* hits, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && frustum_feats && out, "null pointer");
  const LiftParams P = to_params(d);
  const long total = (long) d->B * d->Z * d->Y * d->X;
  hipStream_t s = static_cast<hipStream_t>(stream);
  VAMP_TIMED(kProfLiftFwdDense, s, (lift_fwd_dense_kernel<<<(unsigned) ((total + 255) / 256), 256, 0, s>>>(
      P, mats, xs, ys, zs, frustum_feats, out, hits)));
  return check_launch("lift_fwd_dense_kernel");
}

int vamp_lift_backward_dense(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, const float* grad_out,
                             const uint64_t* hits, float* grad_frustum_feats, void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && grad_out && hits && grad_frustum_feats, "null pointer");
  const LiftParams P = to_params(d);
  const long total = (long) d->B * d->Z * d->Y * d->X;
  hipStream_t s = static_cast<hipStream_t>(stream);
  VAMP_TIMED(kProfLiftBwdDense, s, (lift_bwd_dense_kernel<<<(unsigned) ((total + 255) / 256), 256, 0, s>>>(
      P, mats, xs, ys, zs, grad_out, hits, grad_frustum_feats)));
  return check_launch("lift_bwd_dense_kernel");
}

int vamp_lift_cull_words(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, uint32_t* words, int32_t patch[2], int32_t grid[2], void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(patch && grid, "null pointer");
  LiftWs w = carve(d, nullptr);
  patch[0] = w.cull.px; patch[1] = w.cull.py;
  grid[0] = w.cull.nxp; grid[1] = w.cull.nyp;
  if (!words) return VAMP_OK;
  VAMP_REQUIRE(mats && xs && ys && zs, "null pointer");
  VAMP_REQUIRE(((uintptr_t) mats & 15) == 0, "mats must be 16-byte aligned (the cull reads whole matrix rows)");
  w.cull.words = words;
  lift_cull_kernel<<<(unsigned) ((long) d->B * w.cull.bps), 256, 0, static_cast<hipStream_t>(stream)>>>(
      to_params(d), mats, xs, ys, zs, w.cull);
  return check_launch("lift_cull_kernel");
}

int vamp_lift_indices(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                      const float* zs, uint8_t* valid, int16_t* ix0, int16_t* iy0, int16_t* iz0,
                      void* stream) {
  if (int e = validate(d)) return e;
  VAMP_REQUIRE(mats && xs && ys && zs && valid && ix0 && iy0 && iz0, "null pointer");
  const LiftParams P = to_params(d);
  const long total = (long) d->B * d->N * d->Z * d->Y * d->X;
  lift_indices_kernel<<<(unsigned) ((total + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      P, mats, xs, ys, zs, valid, ix0, iy0, iz0);
  return check_launch("lift_indices_kernel");
}

}  // extern "C"
