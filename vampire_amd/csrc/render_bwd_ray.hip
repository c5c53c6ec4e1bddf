// Camera-branch render backward, per-ray pass (the default path; render_bwd.hip holds the v1
// float-atomic splat kept as an independent cross-check, and the compositing algebra).
//
//  cam_bwd_ray   four waves per 8x8 ray tile, one per depth chunk: each re-marches its samples
//                once (x-pair gathers from the channel-first volumes as they are, pair_gather.hpp:
//                no channel-last copy anywhere on the default path), keeps (s0, delta, q) per sample in
//                LDS, merges the chunks through LDS and writes one record per inside sample --
//                the continuous tap coordinates (fx, fy, fz) exactly as the forward computed
//                them, the compositing weight w_i, dL/ds_i[0], the ray id -- straight to the
//                sample's slot in the cell-ordered record array (the slots come from the
//                geometry-only prepare pass of render_bwd_cell.hip); plus the ray's upstream
//                gradient row (Gcl).
//  The records are then gathered per voxel: render_bwd_cell.hip.
#include "render_common.hpp"
#include "cell_list.hpp"
#include "cam_lists.hpp"
#include "pair_gather.hpp"

namespace vamp {

// ---------------------------------------------------------------------------
// per-ray pass
// ---------------------------------------------------------------------------
// KT: the number of semantic classes when known at compile time (the channel -> tensor mapping of the
// gather then costs no scalar selects; 0 = read P.K)
template <typename T, int LPR, int CP4, int KT>
__global__ void __launch_bounds__(256)
cam_bwd_ray_kernel(RenderParams P, const float* __restrict__ mats, const float* __restrict__ us,
                   const float* __restrict__ vs, const float* __restrict__ ds,
                   const float* __restrict__ mids, const float* __restrict__ beta_raw,
                   const T* __restrict__ dens, const T* __restrict__ sem, const T* __restrict__ rgbv,
                   const float* __restrict__ g_rgb,
                   const float* __restrict__ g_seg, const float* __restrict__ g_depth,
                   CamCellRefs cells,
                   float* __restrict__ Gcl, float* __restrict__ beta_part,
                   const float* __restrict__ samples, const int* __restrict__ term, int L, CamListArgs lists) {
  constexpr int CP = CP4 * 4;
  // the workgroups behind the ray tiles build the work lists of the two kernels that follow (cam_lists.hpp): they
  // need the scanned cells only, and start as the tiles' tail frees slots
  if (blockIdx.x >= lists.first_block) {
    cam_lists_block(P, lists, blockIdx.x - lists.first_block);
    return;
  }
  extern __shared__ float lds[];              // [3][L][256]: s0, delta (sign = no-grad flag), q
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float* l_s0 = lds;
  float* l_dl = lds + (long) L * 256;
  float* l_q = lds + (long) 2 * L * 256;

  // wave-per-depth-chunk mapping (render_common.hpp): lanes = the 64 rays of an 8x8 tile
  static_assert(LPR == 4, "the four waves of the workgroup are the four depth chunks");
  __shared__ float xm[2 * 4 * 64];
  // tiles deepest first (cells.tile_order, from the prepare pass); the grid is padded to a multiple of 8
  const long ntiles = (long) P.B * P.N * ((P.fH + 7) / 8) * ((P.fW + 7) / 8);
  const RayId id = cells.tile_order ? decode_ray_tile(P, blockIdx.x < ntiles ? cells.tile_order[blockIdx.x] : -1)
                                    : decode_ray_wps(P);
  const bool live = id.live;
  float4* __restrict__ REC = cells.R;
  const int w = id.w, h = id.h, sub = id.sub, b = id.b;
  const long bn = id.bn;
  const long ray = (bn * P.fH + h) * P.fW + w;
  const DensityParams dp = load_density(P.density_mode, beta_raw, P.beta_min, P.sdf_bias);
  const int S = P.D - 1;
  // early ray termination: this ray's samples from index `keep` on are dropped (they have no
  // record either: render_bwd_cell.hip), and the tile's four waves share its first Se indices
  const int keep = term ? term[ray] : S;
  int Se = keep;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) Se = max(Se, __shfl_xor(Se, o, 64));
  const int Le = term ? (Se + LPR - 1) / LPR : L;
  const int i0 = min(Se, sub * Le), i1 = min(Se, i0 + Le);
  const float* m = mats + bn * 48;
  const float u = us[w], v = vs[h];
  const long V = (long) P.Z * P.Y * P.X;
  const int bu = __builtin_amdgcn_readfirstlane(b);                 // (a tile is one camera of one sample)
  const __amdgpu_buffer_rsrc_t rs_d = make_rsrc(dens + (long) bu * V, (size_t) V * sizeof(T));
  const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(sem + (long) bu * P.K * V, (size_t) P.K * V * sizeof(T));
  const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(rgbv + (long) bu * 3 * V, (size_t) 3 * V * sizeof(T));
  const unsigned vbytes = (unsigned) V * (unsigned) sizeof(T);
  const int Kc = KT > 0 ? KT : P.K;
  const int nch = 1 + Kc + 3;
  const long HW = (long) P.fH * P.fW;
  const long pix = (long) h * P.fW + w;

  float G[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    float gv = 0.f;
    if (live) {
      if (c >= 1 && c <= P.K) gv = g_seg ? g_seg[(bn * P.K + (c - 1)) * HW + pix] : 0.f;
      else if (c > P.K && c <= P.K + 3) gv = g_rgb ? g_rgb[(bn * 3 + (c - 1 - P.K)) * HW + pix] : 0.f;
    }
    G[c] = gv;
  }
  const float Gd = (live && g_depth) ? g_depth[bn * HW + pix] : 0.f;
  if (live && sub == 0) {
    float4* dst = reinterpret_cast<float4*>(Gcl + ray * CP);
#pragma unroll
    for (int q = 0; q < CP4; ++q) dst[q] = make_float4(G[q * 4], G[q * 4 + 1], G[q * 4 + 2], G[q * 4 + 3]);
  }

  // ---- march the chunk once ----
  float px, py, pz, qx, qy, qz;
  auto point = [&](int i, float& x, float& y, float& z) {
    frustum_point(m, u, v, ds[i], x, y, z);
    x = nan_to_num_geom(x); y = nan_to_num_geom(y); z = nan_to_num_geom(z);
  };
  if (i0 < i1) point(i0, px, py, pz);
  float cum = 0.f, A = 0.f, E1 = 1.f;
  for (int i = i0; i < i1; ++i) {
    point(i + 1, qx, qy, qz);
    VolTap tp = volume_tap(P, px, py, pz);
    tp.inside = tp.inside && i < keep;
    // the sample's slot in the cell-ordered record array: start of its cell + the rank it drew in the
    // prepare pass (render_bwd_cell.hip); requested ahead of the gather so that the latency hides behind it
    const long sidx = (id.tile * S + i) * 64 + (tid & 63);
    int slot = -1;
    if (live && tp.inside) {
      const long cell = key_to_cell(pack_cell_key(tp.ix0, tp.iy0, tp.iz0), P.Y, P.X, (unsigned) b, cells.ncell_b);
      slot = cells.off[cell] + cells.boff[cell / kScanTile] + cells.rank[sidx];
    }
    float s[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) s[c] = 0.f;
    if (tp.inside) {
      if (samples) {
        // the values the forward sampled (render_cam_direct.hip / render_cam_fwd_plan_kernel<SAVE>): one
        // coalesced 256-byte run per channel for the tile's 64 rays
        const float* rr = samples + ((id.tile * S + i) * CP) * 64 + (tid & 63);
#pragma unroll
        for (int c = 0; c < CP; ++c) s[c] = (c < nch) ? rr[c * 64] : 0.f;
      } else {
        // channel 0 = density feature, 1..K semantic, K+1..K+3 rgb; four x-pair loads per channel, in
        // batches of four channels (their 16 loads are issued back to back)
        const PairTap pt = pair_tap<T>(P, tp);
        unsigned vb = vbytes;
        asm volatile("" : "+s"(vb));                      // (per-channel offsets recomputed: see render_cam_direct.hip)
#pragma unroll
        for (int c0 = 0; c0 < CP; c0 += 4) {
          PairRaw raw[4][4];
#pragma unroll
          for (int uu = 0; uu < 4; ++uu) {
            const int cc = min(c0 + uu, nch - 1);
            const __amdgpu_buffer_rsrc_t rs = cc == 0 ? rs_d : (cc <= Kc ? rs_s : rs_r);
            const unsigned so = (unsigned) (cc == 0 ? 0 : (cc <= Kc ? cc - 1 : cc - 1 - Kc)) * vb;
            raw[uu][0] = ld_pair<T>(rs, pt.o00, so);
            raw[uu][1] = ld_pair<T>(rs, pt.o01, so);
            raw[uu][2] = ld_pair<T>(rs, pt.o10, so);
            raw[uu][3] = ld_pair<T>(rs, pt.o11, so);
          }
#pragma unroll
          for (int uu = 0; uu < 4; ++uu) s[c0 + uu] = (c0 + uu < nch) ? pair_combine<T>(pt, raw[uu]) : 0.f;
        }
      }
    }
    const bool fin = (s[0] == s[0]) && (fabsf(s[0]) <= 3.402823466e+38f);
    const float s0 = nan_to_num(s[0]);
    // q = G_depth (mid - d_far) + sum_c G_c nan_to_num(s_c).  A non-finite s_c always leaves the
    // plain chain non-finite (inf * 0 and nan * 0 are nan), so the 4-instruction sanitiser per
    // channel only runs on the rare lanes whose plain result is not finite; otherwise the plain
    // chain IS the sanitised one, bit for bit.
    const float qbase = Gd * (mids[i] - P.d_far);
    float qv = qbase;
#pragma unroll
    for (int c = 1; c < CP; ++c) qv = __builtin_fmaf(G[c], s[c], qv);
    if (!(fabsf(qv) <= 3.402823466e+38f)) {
      qv = qbase;
#pragma unroll
      for (int c = 1; c < CP; ++c) qv = __builtin_fmaf(G[c], nan_to_num(s[c]), qv);
    }
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float delta = sqrtf(dx * dx + dy * dy + dz * dz);
    // transmittance inside the chunk as a running product E = prod exp(-tau_j) (one exp per
    // sample instead of two; the second loop forms the same products, so A and its prefix agree
    // to the bit)
    const float tau = density_fwd(dp, s0) * delta;
    const float etau = expf(-tau);
    A = __builtin_fmaf((1.0f - etau) * E1, qv, A);
    E1 *= etau;
    cum += tau;
    const int j = i - i0;
    l_s0[j * 256 + tid] = s0;
    // a sample passes gradient to s[0] only if it is inside and finite: flag in the sign
    l_dl[j * 256 + tid] = (tp.inside && fin) ? delta : -delta;
    l_q[j * 256 + tid] = qv;
    if (live) {
      // the continuous tap coordinates go to the record now, the weights follow in the second loop
      if (tp.inside) REC[2 * (long) slot] = make_float4(tp.fx, tp.fy, tp.fz, 0.f);
      cells.slot[sidx] = slot;
    }
    px = qx; py = qy; pz = qz;
  }

  // ---- merge the LPR chunks of the ray ----
  float scale = 1.f, suffix = 0.f;
  {
    const int lane = tid & 63;
    xm[sub * 64 + lane] = cum;
    __syncthreads();
    float excl = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < sub) excl += xm[k * 64 + lane];
    scale = expf(-excl);
    xm[(4 + sub) * 64 + lane] = scale * A;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k > sub) suffix += xm[(4 + k) * 64 + lane];     // sum_{m > sub} scale_m A_m
  }

  // ---- second loop over the LDS copies: emit w_i and dL/ds_i[0] ----
  float E2 = 1.f, prefix = 0.f, dbeta = 0.f;
  for (int i = i0; i < i1; ++i) {
    const int j = i - i0;
    const int slot = live ? cells.slot[(id.tile * S + i) * 64 + (tid & 63)] : -1;   // this thread's own note
    const float s0 = l_s0[j * 256 + tid];
    const float dl = l_dl[j * 256 + tid];
    const float qv = l_q[j * 256 + tid];
    const float delta = fabsf(dl);
    float sigma, dsig_ds, dsig_db;
    density_all(dp, s0, sigma, dsig_ds, dsig_db);
    const float tau = sigma * delta;
    const float etau = expf(-tau);
    const float wloc = (1.0f - etau) * E2;
    E2 *= etau;                                        // exp(-(cl + tau)) of this chunk
    const float Tn = scale * E2;
    prefix = __builtin_fmaf(wloc, qv, prefix);
    const float R = scale * (A - prefix) + suffix;
    const float dtau = qv * Tn - R;
    dbeta = __builtin_fmaf(dtau * delta, dsig_db, dbeta);
    if (live) {
      if (slot >= 0) {
        REC[2 * (long) slot + 1] = make_float4(scale * wloc, (dl > 0.f) ? dtau * delta * dsig_ds : 0.f,
                                               __uint_as_float((unsigned) ray), 0.f);
      }
    }
  }
  if (P.density_mode == VAMP_DENSITY_SDF_LAPLACE) {
    float vsum = live ? dbeta : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vsum += __shfl_down(vsum, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = vsum;
    __syncthreads();
    // one partial per workgroup; launch_beta_reduce adds them up in a fixed order
    if (tid == 0) beta_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// render_bwd_cell.hip
size_t cam_bwd_cell_bytes(const VampRenderDesc* d);
CamCellRefs cam_cell_refs(const VampRenderDesc* d, void* scratch);
int launch_cam_cells_prepare(const VampRenderDesc* d, const RenderParams& P, const float* mats,
                             const float* us, const float* vs, const float* ds, void* scratch,
                             const int* term, int phase, hipStream_t s, bool counters_clean = false,
                             const ScanJob* also = nullptr);
int launch_cam_bwd_cell(const VampRenderDesc* d, const RenderParams& P, const float* Gcl,
                        float* gdens, float* gsem, float* grgb, void* scratch, int accumulate,
                        hipEvent_t wait_event, int parts, BetaTail btail, hipStream_t s);

static size_t gcl_bytes(const VampRenderDesc* d) {
  const RenderParams P = to_params(d);
  return align_up((size_t) d->B * d->N * d->fH * d->fW * P.CP * sizeof(float), 256);
}

static size_t beta_part_bytes(const VampRenderDesc* d) {
  return align_up((size_t) ray_grid<4>(to_params(d)) * sizeof(float), 256);
}

size_t cam_bwd_v2_bytes(const VampRenderDesc* d) { return gcl_bytes(d) + cam_bwd_cell_bytes(d) + beta_part_bytes(d); }

// the cell lists inside the render workspace: [packed volume | Gcl | cell lists | beta partials]
CamRankRefs cam_rank_refs_cells(const VampRenderDesc* d, void* scratch);
int launch_cam_cells_zero(const VampRenderDesc* d, void* scratch, hipStream_t s);
static void* cell_scratch_of(const VampRenderDesc* d, void* workspace) {
  return static_cast<char*>(workspace) + packed_bytes(d) + gcl_bytes(d);
}
CamRankRefs cam_rank_refs(const VampRenderDesc* d, void* workspace) { return cam_rank_refs_cells(d, cell_scratch_of(d, workspace)); }
int launch_cam_counters_zero(const VampRenderDesc* d, void* workspace, hipStream_t s) {
  return launch_cam_cells_zero(d, cell_scratch_of(d, workspace), s);
}
int launch_cam_prepare_ranked(const VampRenderDesc* d, void* workspace, hipStream_t s, const ScanJob* also) {
  return launch_cam_cells_prepare(d, to_params(d), nullptr, nullptr, nullptr, nullptr, cell_scratch_of(d, workspace), nullptr,
                                  /*phase=*/3, s, false, also);
}

// scratch = workspace region after the packed volume: [Gcl | cell lists | beta partials]
int launch_cam_prepare(const VampRenderDesc* d, const RenderParams& P, const float* mats,
                       const float* us, const float* vs, const float* ds, void* scratch,
                       const int* term, int phase, hipStream_t s, bool counters_clean, const ScanJob* also) {
  return launch_cam_cells_prepare(d, P, mats, us, vs, ds, static_cast<char*>(scratch) + gcl_bytes(d), term, phase, s,
                                  counters_clean, also);
}

int launch_cam_bwd_v2(const VampRenderDesc* d, const RenderParams& P, const float* mats,
                      const float* us, const float* vs, const float* ds, const float* mids,
                      const float* beta, const void* dens, const void* sem, const void* rgbv, const float* g_rgb,
                      const float* g_seg, const float* g_depth, float* gdens, float* gsem,
                      float* grgb, float* grad_beta, void* scratch, int accumulate,
                      hipEvent_t wait_event, int cells_valid, const float* samples, const int* term,
                      int parts, hipStream_t s) {
  float* Gcl = static_cast<float*>(scratch);
  void* cell_scratch = static_cast<char*>(scratch) + gcl_bytes(d);
  // the ray pass leaves one d beta partial per workgroup; the gather's first workgroup adds them up
  BetaTail btail{nullptr, 0, nullptr, nullptr};
  if (P.density_mode == VAMP_DENSITY_SDF_LAPLACE && (parts & kCamPartGather))
    btail = BetaTail{reinterpret_cast<const float*>(static_cast<char*>(cell_scratch) + cam_bwd_cell_bytes(d)),
                     (int) ray_grid<4>(P), beta, grad_beta};
  if (!(parts & kCamPartRay))
    return launch_cam_bwd_cell(d, P, Gcl, gdens, gsem, grgb, cell_scratch, accumulate, wait_event, parts, btail, s);
  // the sample -> slot table depends on the geometry only; the caller may have prepared it
  // (cells_valid 1), or its rank + scan half (2, a caller of rounds 2 - 5: nothing left to do here -- the work lists are built by this launch's tail)
  if (cells_valid != 1)
    if (int e = launch_cam_cells_prepare(d, P, mats, us, vs, ds, cell_scratch, term, cells_valid == 2 ? 2 : 0, s)) return e;
  const CamCellRefs cells = cam_cell_refs(d, cell_scratch);
  float* beta_part = reinterpret_cast<float*>(static_cast<char*>(cell_scratch) + cam_bwd_cell_bytes(d));

  constexpr int LPR = 4;
  const int S = d->D - 1;
  const int L = (S + LPR - 1) / LPR;
  const size_t lds = (size_t) 3 * L * 256 * sizeof(float);
  if (lds > 150 * 1024) return fail(VAMP_EINVAL, "%s: too many depth samples for the LDS staging", __func__);
  const unsigned grid = ray_grid<LPR>(P);
  CamListArgs lists = cam_list_args(d, cell_scratch);
  lists.first_block = grid;
  // (the list counters are zeroed by the cell scan and spent by this launch: VAMP_CAMBWD_CELLS_VALID holds for ONE
  // backward per prepare pass, as the header says)
  if (int e = debug_expect_range(lists.nhcells, 2, 0, 0, s,
                                 "VAMP_CAMBWD_CELLS_VALID: no camera backward has run on this workspace since the prepare pass")) return e;
#define VAMP_RAY_T(T, CP4, KT)                                                                    \
  do {                                                                                            \
    auto kr = cam_bwd_ray_kernel<T, LPR, CP4, KT>;                                                \
    if (lds > 64 * 1024 &&                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kr),                                    \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) \
      return fail(VAMP_EHIP, "%s: cannot raise dynamic LDS", __func__);                           \
    VAMP_TIMED(kProfCamBwd, s, (kr<<<grid + lists.nblocks, 256, lds, s>>>(P, mats, us, vs, ds, mids, beta, \
        static_cast<const T*>(dens), static_cast<const T*>(sem), static_cast<const T*>(rgbv),     \
        g_rgb, g_seg, g_depth, cells, Gcl, beta_part, samples, term, L, lists)));               \
  } while (0)
#define VAMP_RAY(CP4, KT)                                                                         \
  do {                                                                                            \
    if (d->in_dtype == VAMP_F32) VAMP_RAY_T(float, CP4, KT); else VAMP_RAY_T(__hip_bfloat16, CP4, KT); \
  } while (0)
  if (P.CP == 12) VAMP_RAY(3, 0); else if (P.CP == 24 && P.K == 18) VAMP_RAY(6, 18); else if (P.CP == 24) VAMP_RAY(6, 0); else VAMP_RAY(8, 0);
#undef VAMP_RAY
#undef VAMP_RAY_T
  if (int e = check_launch("cam_bwd_ray_kernel")) return e;
  if (!(parts & (kCamPartGather | kCamPartHeavy))) return VAMP_OK;
  return launch_cam_bwd_cell(d, P, Gcl, gdens, gsem, grgb, cell_scratch, accumulate, wait_event, parts, btail, s);
}

}  // namespace vamp
