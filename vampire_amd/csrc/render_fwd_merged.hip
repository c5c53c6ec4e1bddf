// The render forward as ONE launch: volume_rendering_from_multiple_views, bv2:396-467 -- the camera
// branch (bv2:396-440) and the BEV branch (bv2:408-418, 442-461) consume the same four volumes in the
// reference's one function, and here in one grid.
//
// Why one grid.  The camera kernel's second half is a tail: of its 1 056 ray tiles at cfg-B (1 024
// resident) half are done at 23 us, 90 % at 31, the kernel ends at 43 - 48 (tools/debug/cam_stamps.py);
// the BEV kernel behind it then starts on an idle chip and streams for 31 us.  Running the two side by
// side on two streams lost (136 against 124 us replayed: the fork starts the camera kernel 6 - 13 us late
// and both kernels compete from their first microsecond).  In one grid the order is the hardware's
// dispatch order: workgroups [0, ncam) are the camera tiles (cam_fwd_direct_tile), workgroups behind them
// the BEV forward's column blocks (bev_fwd_fused_block) -- they get a slot only as a camera tile retires,
// i.e. exactly in the window the stamps show half empty, with no fork, no event and no second queue.
//
// Both bodies are 256-thread workgroups; the launch takes the larger register count (the camera tile's
// 126, four waves per SIMD) and the larger dynamic LDS of the two.  ncam and the BEV grid's x extent are
// multiples of 8, so blockIdx.x % 8 -- the XCD the hardware deals a workgroup to -- is what each body's
// XCD-banded order expects.  Gathers and streaming reads: no MFMA.
#include "render_cam_direct_dev.hpp"
#include "render_bev_fused_dev.hpp"

#include <algorithm>

namespace vamp {

static_assert(VAMP_DIRECT_NW == 4 && VAMP_BEVF_NWV == 4, "both bodies are four-wave workgroups");
// channel groups of a BEV column block in this launch: the composited group + this many minus one pass-through groups
#ifndef VAMP_MERGED_BEV_PARTS
#define VAMP_MERGED_BEV_PARTS 3              // (2 / 3 / 5 at cfg-B: forward pair 107.0 / 105.7 / 107.8 us)
#endif

struct MergedCam {
  const float *mats, *us, *vs, *ds, *mids;
  float *rgb_out, *seg_out, *depth_out;
  int* term_out;
  float* rows;
  CamRankRefs rank;       // training: the tiles draw the camera backward's cell ranks (cnt == nullptr: no)
};
struct MergedBev {
  const float *oxs, *oys, *ozs, *bev_mids;
  float *bev_rgb, *bev_seg, *bev_height, *voxel_density, *voxel_output, *s0_save, *ss_save;
  int NPA;                // planes of a wave's slab (bev_planes_alloc)
  unsigned gx;            // column blocks per (sample, channel group), a multiple of 8
  int parts;              // channel groups per column block
};
template <typename T>
struct MergedArgs {
  unsigned ncam;          // workgroups [0, ncam): camera tiles (a multiple of 8)
  RenderParams P;
  const float* beta_raw;
  const T *dens, *sem, *rgb, *base;
  MergedCam c;
  MergedBev v;
};

#ifdef VAMP_MERGED_STAMPS
// diagnostic build only (tools/debug/merged_stamps.py): wall-clock start / end of every workgroup
static __device__ long long g_merged_stamps[8192 * 2];
extern "C" int vamp_debug_merged_stamps(long long* host, size_t n) {
  return (int) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_merged_stamps), n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

// a field group of the kernel-argument segment into registers (scalar loads from the constant address space)
template <typename S, typename KS>
__device__ __forceinline__ S kernarg_copy(KS src) {
  static_assert(sizeof(S) % 4 == 0, "dword-sized groups");
  S out;
  typedef const int __attribute__((address_space(4))) * KI;
  KI w = (KI) src;
#pragma unroll
  for (unsigned i = 0; i < sizeof(S) / 4; ++i) reinterpret_cast<int*>(&out)[i] = w[i];
  return out;
}

// The arguments are read INSIDE the branch that uses them, through the kernel-argument segment pointer: taken as
// plain by-value parameters the compiler loads all of them in the entry block and keeps the union of both bodies'
// 70-odd scalars alive across the branch -- 48 scalar spills and, through the lanes that hold them, 52 bytes of
// scratch per lane in a camera tile that has none on its own.
template <typename T, int NCH, bool RANK>
__global__ void __launch_bounds__(256, 4)
render_fwd_merged_kernel(MergedArgs<T> args_in_kernarg_segment) {
#ifdef VAMP_MERGED_STAMPS
  const long long t0 = wall_clock64();
#endif
  typedef const MergedArgs<T> __attribute__((address_space(4))) * KP;
  KP ap = (KP) __builtin_amdgcn_kernarg_segment_ptr();
  // (Camera tiles FIRST is the order that wins: with BEV groups dealt in between -- one per 2 / 3 / 5 / 8 camera groups,
  // so that the streaming blocks would overlap the tiles' latency-bound phases from the first microsecond -- the pair
  // takes 113.7 / 114.1 / 110.2 / 108.6 us against 106.1: a camera tile that starts late ends late.)
  const unsigned ncam = ap->ncam;
  const bool is_cam = blockIdx.x < ncam;
  const unsigned bid = is_cam ? blockIdx.x : blockIdx.x - ncam;
  if (is_cam) {
    asm volatile("" : "+s"(ap));
    const RenderParams P = kernarg_copy<RenderParams>(&ap->P);
    const MergedCam c = kernarg_copy<MergedCam>(&ap->c);
    // a duty of the first workgroup: the word the backward will accumulate d loss / d beta into starts at zero (a
    // one-element fill launch of its own costs a training step ~10 us: 5 of kernel, 5 of hand-over behind it)
    if (bid == 0 && threadIdx.x == 0 && c.rank.zero_word) *c.rank.zero_word = 0.f;
    cam_fwd_direct_tile<T, NCH, true, 4, RANK>(bid, P, c.mats, c.us, c.vs, c.ds, c.mids, ap->beta_raw, ap->dens, ap->sem,
                                               ap->rgb, c.rgb_out, c.seg_out, c.depth_out, c.term_out, c.rows, c.rank);
  } else {
    asm volatile("" : "+s"(ap));
    const RenderParams P = kernarg_copy<RenderParams>(&ap->P);
    const MergedBev v = kernarg_copy<MergedBev>(&ap->v);
    const unsigned q = bid / v.gx, bx = bid - q * v.gx;
    const int part = (int) (q / (unsigned) P.B), b = (int) (q - (unsigned) part * (unsigned) P.B);
    bev_fwd_fused_block<T, 4>(bx, b, part, v.parts, P, v.NPA, v.oxs, v.oys, v.ozs, v.bev_mids, ap->beta_raw, ap->dens,
                              ap->sem, ap->rgb, ap->base, v.bev_rgb, v.bev_seg, v.bev_height, v.voxel_density,
                              v.voxel_output, v.s0_save, v.ss_save);
  }
#ifdef VAMP_MERGED_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    g_merged_stamps[blockIdx.x * 2] = t0;
    g_merged_stamps[blockIdx.x * 2 + 1] = wall_clock64();
  }
#endif
}

bool render_fwd_merged_supported(const VampRenderDesc* d) {
  return d->D - 1 <= kPlanMax && bev_fwd_fused_supported(d) && d->oZ > 0 && d->oY > 0 && d->oX > 0;
}

int launch_render_fwd_merged(const VampRenderDesc* d, const RenderParams& P, const float* mats, const float* us,
                             const float* vs, const float* ds, const float* mids, const float* oxs, const float* oys,
                             const float* ozs, const float* bev_mids, const float* beta, const void* dens,
                             const void* sem, const void* rgb, const void* base, float* rgb_out, float* seg_out,
                             float* depth_out, int* term_out, float* rows, float* bev_rgb, float* bev_seg,
                             float* bev_height, float* voxel_density, float* voxel_output, float* s0_save,
                             float* ss_save, const CamRankRefs& rank, hipStream_t s) {
  const int S = P.D - 1, nch = cam_direct_nch(P.K + 3);
  const long tiles = (long) P.B * P.N * ((P.fH + 7) / 8) * ((P.fW + 7) / 8);
  const long ncam = (tiles + 7) / 8 * 8;
  MergedBev v{oxs, oys, ozs, bev_mids, bev_rgb, bev_seg, bev_height, voxel_density, voxel_output, s0_save, ss_save, 0, 0, 0};
  v.NPA = bev_planes_alloc(d);
  const long cols = (long) P.oY * P.oX;
  v.gx = (unsigned) (((cols + 63) / 64 + 7) / 8 * 8);
  v.parts = P.C > 0 ? std::min(VAMP_MERGED_BEV_PARTS, P.C + 1) : 1;
  const long nbev = (long) v.gx * P.B * v.parts;
  VAMP_REQUIRE(ncam + nbev < 0x7fffffffL, "too many workgroups");
  const size_t dyn_cam = cam_direct_dyn_bytes(S, nch), dyn_bev = bev_fused_dyn_bytes(P.oZ, v.NPA);
  const size_t dyn = dyn_cam > dyn_bev ? dyn_cam : dyn_bev;
  const unsigned grid = (unsigned) (ncam + nbev);
#define VAMP_MRG(T, NCH)                                                                                   \
  do {                                                                                                     \
    const MergedArgs<T> a{(unsigned) ncam, P, beta, static_cast<const T*>(dens), static_cast<const T*>(sem), \
                          static_cast<const T*>(rgb), static_cast<const T*>(base),                         \
                          MergedCam{mats, us, vs, ds, mids, rgb_out, seg_out, depth_out, term_out, rows, rank}, v}; \
    if (rank.cnt)                                                                                          \
      VAMP_TIMED(kProfRenderFwdMerged, s, (render_fwd_merged_kernel<T, NCH, true><<<grid, 256, dyn, s>>>(a)));  \
    else                                                                                                   \
      VAMP_TIMED(kProfRenderFwdMerged, s, (render_fwd_merged_kernel<T, NCH, false><<<grid, 256, dyn, s>>>(a))); \
  } while (0)
#define VAMP_MRG_T(T)                                                                                      \
  do {                                                                                                     \
    if (nch == 8) VAMP_MRG(T, 8);                                                                          \
    else if (nch == 12) VAMP_MRG(T, 12);                                                                   \
    else if (nch == 21) VAMP_MRG(T, 21);                                                                   \
    else if (nch == 24) VAMP_MRG(T, 24);                                                                   \
    else VAMP_MRG(T, 32);                                                                                  \
  } while (0)
  if (d->in_dtype == VAMP_F32) VAMP_MRG_T(float);
  else VAMP_MRG_T(__hip_bfloat16);
#undef VAMP_MRG_T
#undef VAMP_MRG
  return check_launch("render_fwd_merged_kernel");
}

}  // namespace vamp
