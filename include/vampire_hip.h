/*
 * vampire_hip.h -- C ABI of the MI355X (gfx950) lift + volume-render hot path.
 *
 * The reference (cskkxjk/Vampire) has no native code and no FFI: its hot path is
 * Python calling PyTorch aten ops.  Each entry point below therefore names the
 * reference *Python call site* it replaces (paths relative to the reference
 * repository root; "bv2" = src/layers/backbones/base_vampire2.py).  A Python host
 * binds these with ctypes; nothing here depends on torch.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host;
 *  - tensors are dense, row-major, fp32 unless a desc field says otherwise;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *    enqueued asynchronously on it, no call synchronises or allocates;
 *  - inputs are borrowed for the duration of the enqueued work, outputs are fully
 *    overwritten unless documented as accumulated;
 *  - return value: 0 on success, a negative VAMP_E* code otherwise;
 *    vamp_last_error() gives a thread-local message.
 */
#ifndef VAMPIRE_HIP_H_
#define VAMPIRE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAMP_ABI_VERSION 6   /* bumped whenever entry points or flags are added (round 2: 2, round 3: 3, round 4: 4, round 5: 5, round 6: 6) */

enum {
  VAMP_OK = 0,
  VAMP_EINVAL = -1,  /* bad descriptor / null pointer / unsupported shape */
  VAMP_ENOSPC = -2,  /* workspace too small */
  VAMP_EHIP = -3     /* a HIP runtime call failed (launch error) */
};

enum { VAMP_F32 = 0, VAMP_BF16 = 1, VAMP_F16 = 2 };   /* VAMP_F16: only the vamp_conv3d_half_* entry points take it */
enum { VAMP_DENSITY_SIGMOID = 0, VAMP_DENSITY_SDF_LAPLACE = 1 };

int vamp_abi_version(void);
const char* vamp_last_error(void);

/*
 * In-library kernel timing with HIP events on the launch stream (used by bench.py
 * for the roofline figure; off by default, zero cost when off).  While enabled,
 * every launcher brackets each of its kernels with an event pair.
 * vamp_profile_read synchronises the recorded events and returns, for the kernel
 * slot `slot` (0 <= slot < vamp_profile_slots()), its name, number of launches
 * and total milliseconds since the last vamp_profile_enable(1).
 */
/*
 * Debugging mode (off by default): while on, the promises a caller makes with flags are verified before the
 * library relies on them -- VAMP_LIFTFWD_CELLS_CLEAN / VAMP_CAMPREP_COUNTERS_CLEAN (the cell counters in the
 * workspace are zero), VAMP_CAM{FWD,PREP,BWD}_TERM_VALID (the workspace holds a termination table: every entry a
 * number of kept samples) -- and a broken one returns VAMP_EINVAL ("promise broken: ...") instead of corrupting a
 * result silently.  Each check drains the stream it is issued on.
 */
int vamp_debug_checks(int on);

int vamp_profile_enable(int on);
int vamp_profile_slots(void);
/* time only this kernel slot (-1 = all): one event pair per step instead of ~40 */
int vamp_profile_select(int slot);
int vamp_profile_read(int slot, const char** name, int* launches, double* total_ms);

/* ------------------------------------------------------------------------- *
 * LIFT: voxel <- mean over cameras of trilinear samples of depth (x) feat.
 * Replaces bv2:550-553 (outer product), bv2:351-388 (get_pixel) and
 * bv2:483-516 (get_voxel_feats); with use_depth = 0 it is the D == 1 variant of
 * src/layers/backbones/base_bilinear.py:471-519.
 * ------------------------------------------------------------------------- */
typedef struct VampLiftDesc {
  int32_t B, N;        /* samples, cameras per sample                          */
  int32_t C;           /* feature channels (mid_channels), multiple of 4, <= 64 */
  int32_t D, fH, fW;   /* depth planes and feature-map size                    */
  int32_t Z, Y, X;     /* voxel grid, output is [B, C, Z, Y, X]                */
  float u_max, v_max;  /* final_dim[1] - 0.5, final_dim[0] - 0.5  (bv2:494-495) */
  float u_div, v_div;  /* final_dim[1] - 1,   final_dim[0] - 1    (bv2:499-500) */
  float d_lo, d_hi;    /* d_bound[0], d_bound[1]                  (bv2:496)     */
  float d_span;        /* (float)(d_bound[1] - d_bound[0])        (bv2:501)     */
  int32_t use_depth;   /* 1: depth-distribution lift; 0: bilinear (z > 0) lift  */
  int32_t in_dtype;    /* VAMP_F32 or VAMP_BF16 for `depth` and `feat`          */
} VampLiftDesc;

/* Bytes of scratch vamp_lift_forward / vamp_lift_backward need in `workspace`. */
size_t vamp_lift_workspace_bytes(const VampLiftDesc* d);

/*
 * mats   [B, N, 3, 4, 4]  inv(bda), intrin @ inv(sensor2ego), ida   (bv2:370-387)
 * xs,ys,zs                voxel-centre coordinates along each axis  (bv2:273-293)
 * depth  [B, N, D, fH, fW]  depth distribution (ignored when use_depth == 0)
 * feat   [B, N, C, fH, fW]
 * out    [B, C, Z, Y, X]
 * hits   [B, Z, Y, X, ceil(C/16)] u64, 4 bits per channel = number of cameras
 *        with a non-zero sample (bv2:509-512); may be NULL when no backward follows.
 */
int vamp_lift_forward(const VampLiftDesc* d, const float* mats, const float* xs,
                      const float* ys, const float* zs, const void* depth,
                      const void* feat, float* out, uint64_t* hits,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * flags = VAMP_LIFTFWD_EMIT_PAIRS (a backward will follow): the forward kernel -- which projects every
 * voxel into every camera anyway -- also does the counting half of the backward's pixel sort and leaves
 * every valid (voxel, camera) pair's taps in `workspace`.  Afterwards the workspace is in the state
 * vamp_lift_prepare leaves (VAMP_LIFTBWD_CELLS_VALID), and no kernel of the backward projects a voxel
 * again (the backward's prepare pass and half of its fill pass were that projection).  flags == 0 is
 * vamp_lift_forward.
 */
#define VAMP_LIFTFWD_EMIT_PAIRS 1
/* with EMIT_PAIRS: the caller asserts that the cell counters in `workspace` are zero -- as a zero-filled
   buffer has them, and as every completed vamp_lift_forward_ex(EMIT_PAIRS) / vamp_lift_prepare /
   vamp_lift_backward on this workspace leaves them (the scan zeroes what it has read, the backward's gather
   the cursors of the fill) -- so the zero fill in front of the kernel is skipped */
#define VAMP_LIFTFWD_CELLS_CLEAN 2
/* (ABI 6) `feat` is handed over CHANNEL-LAST, [B, N, fH, fW, C] fp32 (in_dtype VAMP_F32; 16-byte aligned) -- the
   memory of a torch.channels_last [B*N, C, fH, fW] tensor, what the producing convolution (channel_lower,
   bv2:551-553) emits natively in that memory format.  The lift samples a pixel's C features as one run, so this is
   the layout it wants: the forward's first launch (the transposing copy into the workspace + the camera cull
   words) disappears -- every workgroup forms its own patch's cull word at its head -- and the forward is one
   kernel.  Without the flag feat is [B, N, C, fH, fW] as in the reference and the first launch runs. */
#define VAMP_LIFTFWD_FEAT_CHANNEL_LAST 4
/* (ABI 6) with VAMP_LIFTFWD_EMIT_PAIRS: the forward counts the pairs per cell but leaves the scan of the counters
 * to a later call on the same stream -- vamp_lift_finish_cells, or vamp_render_camera_prepare_with_lift, which scans
 * them in the launch that scans the camera backward's cells (a training step has both lists due between the render
 * forward and the backward).  Until then the workspace is NOT what VAMP_LIFTBWD_CELLS_VALID promises, and its
 * counters are not clean. */
#define VAMP_LIFTFWD_DEFER_SCAN 8
int vamp_lift_forward_ex(const VampLiftDesc* d, const float* mats, const float* xs,
                         const float* ys, const float* zs, const void* depth,
                         const void* feat, float* out, uint64_t* hits,
                         void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * Producer fusion (bv2:550 `mapping_along_depth(src).softmax(dim=1)` feeding bv2:553):
 * `logits` [B, N, D, fH, fW] (logits_dtype VAMP_F32 | VAMP_BF16) are the raw depth logits.  ONE launch
 * makes both lift operands -- the softmax over D into `depth_out` (fp32 [B, N, D, fH, fW]; keep it for
 * the backward, where it is the `depth` argument) and the channel-last feature copy -- and the lift
 * follows.  d->in_dtype must be VAMP_F32 (feat fp32), d->use_depth 1.  Everything else as
 * vamp_lift_forward.
 */
int vamp_lift_forward_logits(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, const void* logits,
                             int32_t logits_dtype, const float* feat, float* depth_out, float* out,
                             uint64_t* hits, void* workspace, size_t workspace_bytes, void* stream);
/* the same with the flags of vamp_lift_forward_ex */
int vamp_lift_forward_logits_ex(const VampLiftDesc* d, const float* mats, const float* xs,
                                const float* ys, const float* zs, const void* logits,
                                int32_t logits_dtype, const float* feat, float* depth_out, float* out,
                                uint64_t* hits, void* workspace, size_t workspace_bytes, int flags,
                                void* stream);

/*
 * Backward of vamp_lift_forward w.r.t. depth and feat (autograd of bv2:507-514,
 * i.e. grid_sampler_3d_backward + the mean).  grad_depth / grad_feat are fp32 and
 * fully overwritten.  grad_depth may be NULL when use_depth == 0.
 */
int vamp_lift_backward(const VampLiftDesc* d, const float* mats, const float* xs,
                       const float* ys, const float* zs, const void* depth,
                       const void* feat, const float* grad_out, const uint64_t* hits,
                       float* grad_depth, float* grad_feat, void* workspace,
                       size_t workspace_bytes, void* stream);

/*
 * The backward first sorts the valid (voxel, camera) pairs by feature-map pixel; the counting
 * half of that depends on the geometry only.  A forward with VAMP_LIFTFWD_EMIT_PAIRS has done it;
 * vamp_lift_prepare does it alone (a projection-only kernel) into `workspace`; and
 * vamp_lift_backward_ex with VAMP_LIFTBWD_CELLS_VALID then skips it: the caller asserts that
 * `workspace` is the same buffer, d / mats / xs / ys / zs are unchanged and no other lift
 * backward has run on it since.  flags == 0 is vamp_lift_backward.
 */
#define VAMP_LIFTBWD_CELLS_VALID 1
/* implementation selectors (tests cross-check them; 0 = the default cell-list gather):
   SPLAT = the per-voxel float-atomic splat; WPP4 / WPP16 (names from round 3's wave-per-pixel gather) make
   the strip gather stage its pairs in chunks of 64 / 32 instead of 256, so that small inputs cross chunk
   boundaries too; WPP1 = default */
#define VAMP_LIFTBWD_SPLAT 2
#define VAMP_LIFTBWD_WPP1 4
#define VAMP_LIFTBWD_WPP4 8
#define VAMP_LIFTBWD_WPP16 16
/* (64, 128: VAMP_LIFTBWD_HALF_LO / _HI of rounds 2-4, the backward by halves of the images on two streams: measured
   slower twice, removed in round 5) */
/* LOGITS: `depth` is softmax(logits) over D as written by vamp_lift_forward_logits, and grad_depth receives
   the gradient w.r.t. the LOGITS, p * (g - sum_d p g) (autograd of bv2:550): applied to the pixel's column
   while it sits in LDS, no extra pass.  Default (cell-list) backward only. */
#define VAMP_LIFTBWD_LOGITS 256
/* (ABI 6) feat is read, and grad_feat written, channel-last [B, N, fH, fW, C] fp32 (see VAMP_LIFTFWD_FEAT_CHANNEL_LAST) */
#define VAMP_LIFTBWD_FEAT_CHANNEL_LAST 512
/* (ABI 6: takes `depth` -- a pair now carries its four depth samples, so that the backward reads no depth plane) */
int vamp_lift_prepare(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                      const float* zs, const void* depth, void* workspace, size_t workspace_bytes, void* stream);
/* (ABI 6) the scan a forward with VAMP_LIFTFWD_DEFER_SCAN left out (one small kernel); afterwards the workspace
 * is what VAMP_LIFTBWD_CELLS_VALID promises */
int vamp_lift_finish_cells(const VampLiftDesc* d, void* workspace, size_t workspace_bytes, void* stream);
int vamp_lift_backward_ex(const VampLiftDesc* d, const float* mats, const float* xs,
                          const float* ys, const float* zs, const void* depth,
                          const void* feat, const float* grad_out, const uint64_t* hits,
                          float* grad_depth, float* grad_feat, void* workspace,
                          size_t workspace_bytes, int flags, void* stream);

/*
 * Signature-compatible path for get_voxel_feats(frustum_feats, ...) (bv2:483):
 * gathers from the materialised [B, N, C, D, fH, fW] fp32 tensor.
 */
int vamp_lift_forward_dense(const VampLiftDesc* d, const float* mats, const float* xs,
                            const float* ys, const float* zs, const float* frustum_feats,
                            float* out, uint64_t* hits, void* stream);
/* grad_frustum_feats [B, N, C, D, fH, fW] must be zero-filled by the caller; it is
 * accumulated into with atomics. */
int vamp_lift_backward_dense(const VampLiftDesc* d, const float* mats, const float* xs,
                             const float* ys, const float* zs, const float* grad_out,
                             const uint64_t* hits, float* grad_frustum_feats, void* stream);

/*
 * Diagnostics for the "voxel indices bit-exact" contract: the validity mask
 * (bv2:494-497) and floor-corner taps of the lift's trilinear sample, per
 * (b, n, z, y, x).  Shares the projection code with the lift kernels.
 */
int vamp_lift_indices(const VampLiftDesc* d, const float* mats, const float* xs,
                      const float* ys, const float* zs, uint8_t* valid, int16_t* ix0,
                      int16_t* iy0, int16_t* iz0, void* stream);

/*
 * Diagnostics for the forward's camera cull (lift.hip: lift_cull_eval): before a wave of the forward
 * kernel projects its patch[0] x patch[1] voxels of one z plane it reads one word, computed from the
 * matrices alone by the forward's first launch -- bit n set: camera n may hold a voxel of the patch that
 * passes `valid` (bv2:493-497); bit 15: every camera of the sample shares inv(bda) bit for bit.  A clear
 * bit is a proof (conservative half-space test of the patch against the six bounds of `valid`), so
 * skipping the camera changes no bit of the result.  This entry runs the same code into `words`
 * [B, Z, grid[1], grid[0]] (uint32; pass words == NULL to query patch / grid only).
 */
int vamp_lift_cull_words(const VampLiftDesc* d, const float* mats, const float* xs, const float* ys,
                         const float* zs, uint32_t* words, int32_t patch[2], int32_t grid[2], void* stream);

/* ------------------------------------------------------------------------- *
 * RENDER: volume_rendering_from_multiple_views (bv2:391-467) with the density
 * activation of src/utils/render_utils.py:30-46 (or nn.Sigmoid, bv2:191-194)
 * fused in.
 * ------------------------------------------------------------------------- */
typedef struct VampRenderDesc {
  int32_t B, N;          /* samples, cameras                                     */
  int32_t D, fH, fW;     /* frustum planes (D - 1 samples per ray), map size     */
  int32_t K;             /* semantic classes                                     */
  int32_t C;             /* base (voxel_features) channels, BEV branch only      */
  int32_t Z, Y, X;       /* seg volume [B, c, Z, Y, X]                           */
  int32_t oZ, oY, oX;    /* det grid sampled by the BEV branch                   */
  float lo[3];           /* (x, y, z)_bound_seg[0]                   (bv2:397)   */
  float span[3];         /* (float)(bound[1] - bound[0])             (bv2:399)   */
  float d_far;           /* d_bound[1], background depth             (bv2:436)   */
  float z_step_det;      /* z_bound_det[2], BEV delta                (bv2:451)   */
  float det_step[3];     /* (x, y, z)_bound_det[2]: spacing of the det-grid lattice      */
  int32_t density_mode;  /* VAMP_DENSITY_*                                       */
  float sdf_bias;        /* ModifyLaplaceDensity.bias                            */
  float beta_min;        /* ModifyLaplaceDensity.beta_min (1e-4)                 */
  int32_t cat_seg;       /* voxel_output = cat(base, seg)            (bv2:449)   */
  int32_t in_dtype;      /* VAMP_F32 or VAMP_BF16 for the four volumes           */
} VampRenderDesc;

size_t vamp_render_workspace_bytes(const VampRenderDesc* d);

/*
 * Camera branch, forward (bv2:396-407, 419-440 and the nan_to_num of bv2:612).
 *   geom        [B, N, D, fH, fW, 3] ego-frame frustum points, or NULL to have the
 *               kernel evaluate get_geometry (bv2:314-349) itself from:
 *   mats        [B, N, 3, 4, 4]  inv(ida), sensor2ego @ inv(intrin), bda
 *   us[fW], vs[fH], ds[D]        frustum axes (bv2:253-271)
 *   mids        [D - 1]          camera_mids (bv2:243-246)
 *   beta        device pointer to the raw learnable beta (ignored for sigmoid)
 *   density_feature [B,1,Z,Y,X], semantic [B,K,Z,Y,X], rgb [B,3,Z,Y,X]
 * outputs: rgb_out [B,N,3,fH,fW], seg_out [B,N,K,fH,fW], depth_out [B,N,1,fH,fW]
 */
int vamp_render_camera_forward(const VampRenderDesc* d, const float* geom, const float* mats,
                               const float* us, const float* vs, const float* ds,
                               const float* mids, const float* beta,
                               const void* density_feature, const void* semantic,
                               const void* rgb, float* rgb_out, float* seg_out,
                               float* depth_out, void* workspace, size_t workspace_bytes,
                               void* stream);

/*
 * Same, for training: with VAMP_CAMFWD_SAVE_SAMPLES the march also stores the trilinear sample
 * row (the 1 + K + 3 channels, as gathered) of every inside sample, so that the backward's
 * per-ray pass reads 96 contiguous bytes per sample instead of repeating the 8-tap gather.
 * The rows live behind the base region of `workspace`, which must then hold
 * vamp_render_workspace_bytes(d) + vamp_render_samples_bytes(d) bytes (dense addressing by
 * (camera, depth index, pixel): only the rows of inside samples are ever touched).  Needs the
 * in-kernel geometry (geom == NULL); otherwise the flag is ignored and the backward gathers.
 */
#define VAMP_CAMFWD_SAVE_SAMPLES 1
/*
 * Early ray termination (geom == NULL only).  A ray's samples behind the point where its
 * transmittance has fallen below exp(-18) = 1.5e-8 are dropped: together they weigh less than
 * that in every output and gradient (the compositing weights telescope), three orders below the
 * 1e-4 the outputs are held to, and with the reference's sdf density they are most of the inside
 * samples.  vamp_render_camera_terminate marches the density channel alone and leaves the number
 * of kept samples per ray in a table inside `workspace` (needs vamp_render_workspace_bytes(d));
 * forward, prepare and backward take the table from there:
 *   VAMP_CAMFWD_NO_ERT       march every sample (bit-identical to the v1 results)
 *   VAMP_CAMFWD_TERM_VALID   the table is already in `workspace` (vamp_render_camera_terminate has
 *                            run for these volumes / matrices / beta); otherwise the forward
 *                            builds it first
 */
#define VAMP_CAMFWD_NO_ERT 2
#define VAMP_CAMFWD_TERM_VALID 4
#define VAMP_CAMFWD_PACK_ONLY 8      /* only the channel-last copy of (density, semantic, rgb) into the workspace */
#define VAMP_CAMFWD_PACKED_VALID 16   /* the workspace already holds that copy: march only */
/* geom == NULL only: the whole camera branch as ONE kernel that reads the channel-first volumes
 * directly (render_cam_direct.hip) -- no channel-last copy, no separate termination pass; the
 * per-ray table is written into `workspace` as a by-product when workspace_bytes >=
 * vamp_render_workspace_bytes(d) (workspace may be NULL otherwise).  Combines with NO_ERT. */
#define VAMP_CAMFWD_DIRECT 32
/* With DIRECT the DENSITY samples' tap coordinates come from the reference's own fp32 chain (bv2:328-349, 397-404):
 * the ray's exact line in fp64 deviates from that chain by the chain's rounding, a few ulp of a tap coordinate, which
 * the Laplace density's slope (1 / (2 beta^2) = 50) turns into up to 2.2e-4 m of rendered depth at cfg-A -- outside
 * north_star's 1e-4.  Through ABI 5 the chain was this flag and the line the default of the C entry point; since
 * ABI 6 the chain is the only behaviour (rendered depth within 4.3e-5 m, semantic logits 7.7e-5 at cfg-A) and the flag
 * is accepted and ignored. */
#define VAMP_CAMFWD_EXACT_TAPS 64
int vamp_render_camera_terminate(const VampRenderDesc* d, const float* mats, const float* us,
                                 const float* vs, const float* ds, const float* beta,
                                 const void* density_feature, void* workspace, size_t workspace_bytes,
                                 void* stream);
size_t vamp_render_samples_bytes(const VampRenderDesc* d);
/* Byte offset of that table inside `workspace`: int32 [B, N, fH, fW], the number of leading
 * samples each ray keeps (D - 1 = nothing dropped).  For callers that report how much of the
 * march early ray termination removed (bench.py). */
size_t vamp_render_term_offset(const VampRenderDesc* d);
int vamp_render_camera_forward_ex(const VampRenderDesc* d, const float* geom, const float* mats,
                                  const float* us, const float* vs, const float* ds,
                                  const float* mids, const float* beta,
                                  const void* density_feature, const void* semantic,
                                  const void* rgb, float* rgb_out, float* seg_out,
                                  float* depth_out, void* workspace, size_t workspace_bytes,
                                  int flags, void* stream);

/*
 * Camera branch, backward.  g_* are the upstream gradients of the three outputs
 * (any may be NULL = zero).  grad_density_feature / grad_semantic / grad_rgb are
 * fp32 [B,c,Z,Y,X], fully overwritten.  grad_beta (1 float) is ACCUMULATED into.
 */
int vamp_render_camera_backward(const VampRenderDesc* d, const float* geom, const float* mats,
                                const float* us, const float* vs, const float* ds,
                                const float* mids, const float* beta,
                                const void* density_feature, const void* semantic,
                                const void* rgb, const float* g_rgb, const float* g_seg,
                                const float* g_depth, float* grad_density_feature,
                                float* grad_semantic, float* grad_rgb, float* grad_beta,
                                void* workspace, size_t workspace_bytes, void* stream);

/*
 * Same, for callers that run the two render branches on two streams (they are independent up
 * to these three buffers):
 *   flags       VAMP_CAMBWD_ACCUMULATE: the gradients are ADDED to what grad_density_feature /
 *               grad_semantic / grad_rgb hold (e.g. the BEV branch's contribution, written first)
 *               VAMP_CAMBWD_PACKED_VALID: `workspace` is the buffer vamp_render_camera_forward
 *               ran with for these same volumes and nothing has written to it since, so its
 *               channel-last copy of the volumes is reused instead of rebuilt
 *               VAMP_CAMBWD_CELLS_VALID: vamp_render_camera_prepare has run on this workspace
 *               for the same d / mats / us / vs / ds and no camera backward since
 *   wait_event  a hipEvent_t (or NULL) the stream waits for right before those buffers are first
 *               touched, i.e. after the per-ray pass and the sample sort have been queued
 * ACCUMULATE and wait_event need the default (cell-list) implementation with mats (geom == NULL).
 */
/*
 * The backward sorts the ray samples by the voxel cell of their floor tap; which sample goes to
 * which slot depends on the geometry only.  vamp_render_camera_prepare computes that table into
 * `workspace` ahead of time (e.g. on a second stream beside vamp_render_camera_forward, which
 * only touches the head of the workspace); without it the backward computes the table itself.
 */
int vamp_render_camera_prepare(const VampRenderDesc* d, const float* mats, const float* us,
                               const float* vs, const float* ds, void* workspace,
                               size_t workspace_bytes, void* stream);
/* flags: VAMP_CAMPREP_TERM_VALID -- sort only the samples the early-termination table in
   `workspace` keeps (the backward must then be given VAMP_CAMBWD_TERM_VALID too) */
#define VAMP_CAMPREP_TERM_VALID 1
#define VAMP_CAMPREP_RANKED 8   /* (ABI 6) the ranks have been drawn by vamp_render_forward_merged(VAMP_RENDERFWD_RANK): only the cell scan runs (mats .. ds may be NULL; the backward's work lists are built inside the per-ray pass's launch) */
#define VAMP_CAMPREP_COUNTERS_CLEAN 4   /* the caller asserts that the cell counters in `workspace` are zero (a zero-filled buffer, or one a completed prepare pass has run on: its scan zeroes what it reads): no zero fill */
/* (2: VAMP_CAMPREP_RANK_ONLY of rounds 2-4, the prepare pass in two phases: measured slower twice, removed in round 5) */
int vamp_render_camera_prepare_ex(const VampRenderDesc* d, const float* mats, const float* us,
                                  const float* vs, const float* ds, void* workspace,
                                  size_t workspace_bytes, int flags, void* stream);
/* (ABI 6) vamp_render_camera_prepare_ex whose scan launch ALSO scans the pair cells of a lift forward that ran with
 * VAMP_LIFTFWD_DEFER_SCAN on `lift_workspace` (on this stream, or on one this call is ordered behind): one launch of
 * two independent scans instead of two launches -- it stands for vamp_lift_finish_cells.  Whoever consumes the lift's
 * cells (vamp_lift_backward_ex) must be ordered behind this call. */
int vamp_render_camera_prepare_with_lift(const VampRenderDesc* d, const float* mats, const float* us,
                                         const float* vs, const float* ds, void* workspace, size_t workspace_bytes,
                                         int flags, const VampLiftDesc* lift_desc, void* lift_workspace,
                                         size_t lift_workspace_bytes, void* stream);
#define VAMP_CAMBWD_ACCUMULATE 1
#define VAMP_CAMBWD_PACKED_VALID 2
#define VAMP_CAMBWD_CELLS_VALID 4
/* implementation selector: the float-atomic splat instead of the default cell-list gather
   (the independent cross-check of the tests; also what a caller-supplied geom tensor takes) */
#define VAMP_CAMBWD_SPLAT 8
/* the workspace holds the sample rows of vamp_render_camera_forward_ex(VAMP_CAMFWD_SAVE_SAMPLES)
   for these same volumes / matrices and nothing has written to it since (cell-list path only) */
#define VAMP_CAMBWD_SAMPLES_VALID 16
/* early ray termination in the cell-list path: TERM_VALID = the table of the forward is in
   `workspace`; without it the backward builds the table itself; NO_ERT = every sample */
#define VAMP_CAMBWD_TERM_VALID 32
#define VAMP_CAMBWD_NO_ERT 64
/* Parts of the call, for a caller with two streams (none set = all three: ray, heavy, gather):
 *   PART_RAY     the per-ray pass (and the channel-last copy / termination table / cell lists and
 *                work lists), d loss / d beta
 *   PART_HEAVY   the kernel that sums the heavy cells (more than 32 records) once per cell into per-corner
 *                partial rows in the workspace; it does not touch the gradient buffers.  (Rounds 1 - 5: a
 *                drain of heavy VOXELS that could run beside the gather; since round 6 the gather reads
 *                this part's output, so a caller that splits the parts issues it before PART_GATHER on
 *                the same stream, or behind an event of its own.)
 *   PART_GATHER  the per-voxel gather (waits for wait_event first)
 * Give every part the same VALID / ACCUMULATE flags. */
/* (1024: VAMP_CAMBWD_SLOTS_PENDING, the second phase of VAMP_CAMPREP_RANK_ONLY: removed with it) */
#define VAMP_CAMBWD_PART_RAY 128
#define VAMP_CAMBWD_PART_GATHER 256
#define VAMP_CAMBWD_PART_HEAVY 512
int vamp_render_camera_backward_acc(const VampRenderDesc* d, const float* geom, const float* mats,
                                    const float* us, const float* vs, const float* ds,
                                    const float* mids, const float* beta,
                                    const void* density_feature, const void* semantic,
                                    const void* rgb, const float* g_rgb, const float* g_seg,
                                    const float* g_depth, float* grad_density_feature,
                                    float* grad_semantic, float* grad_rgb, float* grad_beta,
                                    void* workspace, size_t workspace_bytes, int flags,
                                    void* wait_event, void* stream);

/*
 * BEV (top-down) branch, forward (bv2:408-418, 442-461).
 *   oxs[oX], oys[oY], ozs[oZ]  det-grid centres (bv2:160), bev_mids [oZ] (bv2:248-251)
 *   base [B,C,Z,Y,X]
 * outputs: bev_rgb [B,3,oY,oX], bev_seg [B,K,oY,oX], bev_height [B,1,oY,oX],
 *          voxel_density [B,1,oZ,oY,oX], voxel_output [B,C(+K),oZ,oY,oX]
 */
int vamp_render_bev_forward(const VampRenderDesc* d, const float* oxs, const float* oys,
                            const float* ozs, const float* bev_mids, const float* beta,
                            const void* density_feature, const void* semantic,
                            const void* rgb, const void* base, float* bev_rgb,
                            float* bev_seg, float* bev_height, float* voxel_density,
                            float* voxel_output, void* stream);

/*
 * The same with flags.  VAMP_BEVFWD_SAVE (training): the density samples and the composited
 * channels' samples of every det-grid point are kept in `workspace`
 * (vamp_render_bev_workspace_bytes(d) bytes; +35 MB per sample at cfg-B); a backward call on the
 * same workspace with VAMP_BEVBWD_SAVED_VALID reads them back instead of sampling again.
 */
#define VAMP_BEVFWD_SAVE 1
#define VAMP_BEVFWD_TWO_KERNELS 2   /* the first implementation (density pass + channel-pair pass) instead of the one-kernel forward of render_bev_fused.hip: kept as the cross-check of the tests */
/* ozs_host (ABI 6; was the caller-asserted flag VAMP_BEVFWD_HEIGHTS_LATTICE = 4): a HOST copy of ozs, or NULL.  The
 * one-kernel forward sizes its per-wave plane slabs for a lattice of heights with spacing det_step[2] (what the
 * reference's create_voxel_coords makes, bv2:273-293) and would clamp -- silently -- a plane outside them, so it runs
 * only when the library has CHECKED, on ozs_host, that the heights' z taps fit the slabs; NULL, or an array that does
 * not fit (any jittered / non-uniform ozs), takes the two-kernel path, which handles every height on its own. */
int vamp_render_bev_forward_ex(const VampRenderDesc* d, const float* oxs, const float* oys,
                            const float* ozs, const float* bev_mids, const float* beta,
                            const void* density_feature, const void* semantic,
                            const void* rgb, const void* base, float* bev_rgb,
                            float* bev_seg, float* bev_height, float* voxel_density,
                            float* voxel_output, const float* ozs_host, void* workspace,
                            size_t workspace_bytes, int flags, void* stream);

/*
 * The render forward as ONE launch (ABI 6): volume_rendering_from_multiple_views, bv2:396-467 -- camera branch
 * (bv2:396-440) and BEV branch (bv2:408-418, 442-461) consume the same four volumes in the reference's one function,
 * and here in one grid: the camera branch's 8 x 8 ray tiles are the first workgroups, the BEV branch's column blocks
 * the ones behind them, so the BEV blocks fill the slots the camera tiles' long tail leaves idle (no fork, no event,
 * no second queue).  Same results as vamp_render_camera_forward_ex(VAMP_CAMFWD_DIRECT) + vamp_render_bev_forward_ex
 * bit for bit (the same device functions).  Early ray termination is on (the data-independent forward keeps its two
 * launches).  Arguments as in those two calls; `workspace` (>= vamp_render_workspace_bytes(d), plus
 * vamp_render_samples_bytes(d) with VAMP_RENDERFWD_SAVE_SAMPLES) receives the termination table and the kept sample
 * rows, `bev_workspace` (vamp_render_bev_workspace_bytes(d), only with VAMP_RENDERFWD_BEV_SAVE) the BEV samples.
 * vamp_render_forward_merged_supported: 1 when the shapes qualify (at most 128 samples per ray, the one-kernel BEV
 * forward's limits) AND ozs_host fits the BEV slabs (see vamp_render_bev_forward_ex); the call itself returns
 * VAMP_EINVAL otherwise -- the caller then issues the two calls.
 * grad_beta_zero (may be NULL): one float the launch sets to zero -- the accumulator a training caller will hand to
 * vamp_render_bev_backward* / vamp_render_camera_backward* as grad_beta, which ADD to it: a one-element fill launch
 * at the head of the backward costs a replayed step ~10 us.
 */
#define VAMP_RENDERFWD_SAVE_SAMPLES 1   /* = VAMP_CAMFWD_SAVE_SAMPLES */
#define VAMP_RENDERFWD_BEV_SAVE 2       /* = VAMP_BEVFWD_SAVE */
/* training: the camera tiles also do the RANK PASS of the camera backward's cell sort -- every kept inside sample is
   counted into its cell and takes its rank there, behind its channel loads.  The caller finishes the prepare step with
   vamp_render_camera_prepare_ex(.., VAMP_CAMPREP_RANKED, stream) (the cell scan: one small kernel, on any
   stream ordered behind this call); afterwards `workspace` holds what vamp_render_camera_prepare_ex
   (VAMP_CAMPREP_TERM_VALID) leaves, and the backward takes VAMP_CAMBWD_CELLS_VALID.  The three launches on two streams
   of a training forward (camera kernel, BEV forward, prepare pass beside it) become one launch + one small one.
   COUNTERS_CLEAN: as VAMP_CAMPREP_COUNTERS_CLEAN (otherwise the counters are zero-filled first). */
#define VAMP_RENDERFWD_RANK 4
#define VAMP_RENDERFWD_COUNTERS_CLEAN 8
int vamp_render_forward_merged_supported(const VampRenderDesc* d, const float* ozs_host);
int vamp_render_forward_merged(const VampRenderDesc* d, const float* mats, const float* us, const float* vs,
                               const float* ds, const float* mids, const float* oxs, const float* oys,
                               const float* ozs, const float* ozs_host, const float* bev_mids, const float* beta,
                               const void* density_feature, const void* semantic, const void* rgb,
                               const void* base, float* rgb_out, float* seg_out, float* depth_out,
                               float* bev_rgb, float* bev_seg, float* bev_height, float* voxel_density,
                               float* voxel_output, void* workspace, size_t workspace_bytes,
                               void* bev_workspace, size_t bev_workspace_bytes, float* grad_beta_zero, int flags,
                               void* stream);

/*
 * BEV branch, backward.  The four volume gradients are ACCUMULATED into (so that
 * the camera-branch backward can run first into the same buffers); grad_base is
 * written only by this call and must be zero-filled (or hold a running sum).
 * ozs_host is a HOST copy of ozs (used to find the volume planes the det grid
 * touches); workspace needs vamp_render_bev_workspace_bytes(d) bytes.  With
 * ozs_host == NULL the call takes the slower float-atomic formulation (which the
 * tests use as the independent cross-check of the gather).
 */
size_t vamp_render_bev_workspace_bytes(const VampRenderDesc* d);
int vamp_render_bev_backward(const VampRenderDesc* d, const float* oxs, const float* oys,
                             const float* ozs, const float* bev_mids, const float* beta,
                             const void* density_feature, const void* semantic,
                             const void* rgb, const void* base, const float* g_bev_rgb,
                             const float* g_bev_seg, const float* g_bev_height,
                             const float* g_voxel_density, const float* g_voxel_output,
                             float* grad_density_feature, float* grad_semantic,
                             float* grad_rgb, float* grad_base, float* grad_beta,
                             const float* ozs_host, void* workspace, size_t workspace_bytes,
                             void* stream);

/*
 * The same with flags: an OVERWRITE bit makes the call write the named gradients instead of
 * adding to them (voxel planes the det grid does not touch get zeros), which saves the zero fill
 * and the read-modify-write; run it BEFORE vamp_render_camera_backward_acc(..ACCUMULATE) then.
 */
#define VAMP_BEVBWD_OVERWRITE_BASE 1   /* grad_base */
#define VAMP_BEVBWD_OVERWRITE_CAM 2    /* grad_density_feature, grad_semantic, grad_rgb */
#define VAMP_BEVBWD_SKIP_BASE 16        /* first half of a split pair: q, scan, the composited channels' gather (what the camera gather waits for) ... */
#define VAMP_BEVBWD_ONLY_BASE 8         /* ... second half: the beta reduction and the pass-through (grad_base) gather, which nobody waits for; give both calls the same OVERWRITE flags, and issue both */
#define VAMP_BEVBWD_TABLE_VALID 32      /* the workspace still holds the axis tables of an earlier call with the same grids (both halves write their own) */
#define VAMP_BEVBWD_SAVED_VALID 4      /* the workspace holds what vamp_render_bev_forward_ex(.., VAMP_BEVFWD_SAVE) kept */
int vamp_render_bev_backward_ex(const VampRenderDesc* d, const float* oxs, const float* oys,
                             const float* ozs, const float* bev_mids, const float* beta,
                             const void* density_feature, const void* semantic,
                             const void* rgb, const void* base, const float* g_bev_rgb,
                             const float* g_bev_seg, const float* g_bev_height,
                             const float* g_voxel_density, const float* g_voxel_output,
                             float* grad_density_feature, float* grad_semantic,
                             float* grad_rgb, float* grad_base, float* grad_beta,
                             const float* ozs_host, void* workspace, size_t workspace_bytes,
                             int flags, void* stream);

/*
 * Diagnostics: inside-mask (bv2:405-407) and floor taps of the camera branch's
 * trilinear sample per (b, n, d < D-1, h, w); geometry as in the forward.
 */
int vamp_render_indices(const VampRenderDesc* d, const float* geom, const float* mats,
                        const float* us, const float* vs, const float* ds, uint8_t* inside,
                        int16_t* ix0, int16_t* iy0, int16_t* iz0, void* stream);

/*
 * The same diagnostic for the one-kernel camera forward (render_cam_direct.hip): inside mask and floor
 * taps of every sample [B, N, D-1, fH, fW] as THAT kernel evaluates them (its sample coordinates come
 * from an fp64 line per ray with a wave-level fallback to the fp32 chain of bv2:328-349 next to the
 * faces of the volume); fxyz (may be NULL) [B, N, D-1, fH, fW, 3] receives the continuous tap
 * coordinates.  The mask is the reference's bit for bit; a floor tap can differ from the reference's
 * where a coordinate lies within ~1e-5 of an integer (the trilinear sample is continuous there).
 */
int vamp_render_camera_direct_taps(const VampRenderDesc* d, const float* mats, const float* us,
                                   const float* vs, const float* ds, uint8_t* inside, int16_t* ix0,
                                   int16_t* iy0, int16_t* iz0, float* fxyz, void* stream);

/* get_geometry (bv2:314-349) + nan_to_num(-1e3) (bv2:612): geom [B,N,D,fH,fW,3]. */
int vamp_frustum_geometry(const VampRenderDesc* d, const float* mats, const float* us,
                          const float* vs, const float* ds, float* geom, void* stream);

/* --------------------------------------------------------------------------
 * Point resampling: the occupancy and lidar-point queries of
 * base_vampire2.py:576-609 -- F.grid_sample(volume, points, align_corners=True) with the
 * points normalised by the seg bounds.
 *   occ_logits  bv2:603   padding BORDER                     (points = bda-rotated occ grid, bv2:599-602)
 *   occ_density bv2:604   padding ZEROS, activation = 1      (density(density_feature) is sampled)
 *   pts_logits  bv2:590   padding BORDER, channel_last_out   (lidar points of one sample, B = 1)
 *   pts_sdf     bv2:594   padding ZEROS, mask_outside = 1
 * -------------------------------------------------------------------------- */
#define VAMP_PAD_ZEROS 0
#define VAMP_PAD_BORDER 1
typedef struct {
  int32_t B, C;            /* samples, channels of the volume (C <= 32; C == 1 with activation) */
  int32_t Z, Y, X;
  float lo[3], span[3];    /* (p - lo) / span * 2 - 1 is the normalised coordinate (bv2:581-586) */
  int32_t padding;         /* VAMP_PAD_ZEROS | VAMP_PAD_BORDER */
  int32_t mask_outside;    /* multiply by all(-1 <= n <= 1) (bv2:587-589, 595) */
  int32_t activation;      /* sample density(volume) instead of volume (bv2:604) */
  int32_t density_mode;    /* VAMP_DENSITY_*; with sdf_bias / beta_min as in VampRenderDesc */
  float sdf_bias, beta_min;
  int32_t channel_last_out;/* out / grad_out are [B, P, C] instead of [B, C, P] */
  int32_t in_dtype;        /* dtype of `volume` */
  int32_t lattice[3];      /* optional hint: the P points are a row-major n0 x n1 x n2 lattice (n2
                              fastest) whose axis 0 runs along the volume's x, as the occ grid does
                              ([200, 200, 16, 3], bv2:295-312): threads then walk axis 0 so that
                              neighbouring lanes read neighbouring voxels.  0, 0, 0 = no structure */
} VampSampleDesc;

/*
 * volume [B, C, Z, Y, X]; points [B, P, 3] fp32 ego-frame (x, y, z), P = points_per_sample;
 * out [B, C, P] (or [B, P, C]) fp32; beta as in the render entry points (read only with
 * activation and the sdf density).
 */
int vamp_sample_points_forward(const VampSampleDesc* d, const void* volume, const float* beta,
                               const float* points, int64_t points_per_sample, float* out,
                               void* stream);
size_t vamp_sample_points_workspace_bytes(const VampSampleDesc* d, int64_t points_per_sample);
/* grad_volume [B, C, Z, Y, X] fp32 is fully overwritten; grad_beta (1 float) is ACCUMULATED into. */
int vamp_sample_points_backward(const VampSampleDesc* d, const void* volume, const float* beta,
                                const float* points, int64_t points_per_sample,
                                const float* grad_out, float* grad_volume, float* grad_beta,
                                void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------
 * Producer / consumer steps either side of the path.
 *   depth softmax  base_vampire2.py:550  `mapping_along_depth(src).softmax(dim=1)`:
 *                  logits [images, D, HW] (fp32 | bf16, images = B * N cameras, HW = fH * fW)
 *                  -> depth [images, D, HW] fp32, the `depth` argument of vamp_lift_forward.
 *   density gate   base_vampire2.py:627-630  `voxel_output * bev_density.tanh()` (sdf density) or
 *                  `voxel_output * bev_density` (naive): voxel_output [B, C, cells], voxel_density
 *                  [B, 1, cells] (cells = oZ * oY * oX, both outputs of vamp_render_forward) ->
 *                  out [B, C, cells], the input of the `voxel_output` 1x1 conv.
 * All tensors contiguous; nothing is retained.
 * -------------------------------------------------------------------------- */
int vamp_depth_softmax_forward(int64_t images, int32_t D, int64_t HW, const void* logits,
                               int32_t in_dtype, float* depth, void* stream);
/* grad_logits = depth * (grad_depth - sum_d depth * grad_depth), fully overwritten */
int vamp_depth_softmax_backward(int64_t images, int32_t D, int64_t HW, const float* depth,
                                const float* grad_depth, float* grad_logits, void* stream);
int vamp_density_gate_forward(int64_t B, int32_t C, int64_t cells, int32_t density_mode,
                              const float* voxel_output, const float* voxel_density, float* out,
                              void* stream);
/* grad_voxel_output [B, C, cells] and grad_voxel_density [B, 1, cells] are fully overwritten */
int vamp_density_gate_backward(int64_t B, int32_t C, int64_t cells, int32_t density_mode,
                               const float* grad_out, const float* voxel_output,
                               const float* voxel_density, float* grad_voxel_output,
                               float* grad_voxel_density, void* stream);

/*
 * Consumer fusion (base_vampire2.py:627-632 with the conv of :203-209): the density gate and
 * the `voxel_output` 1x1 convolution in one kernel each way, on the fp32 matrix cores --
 *   out[b, o, cell] = bias[o] + sum_ci weight[o, ci] * voxel_output[b, ci, cell] * gate(voxel_density[b, ci % oZ, cell])
 * with ci = c * oZ + z (the reference's reshape(B, C * oZ, oY, oX)), cell = oY * oX positions of the BEV
 * plane, gate = tanh (sdf density) or identity (naive).  voxel_output [B, C, oZ, cells], voxel_density
 * [B, 1, oZ, cells] (outputs of vamp_render_forward), weight [Cout, C * oZ] (= Conv2d.weight), bias [Cout] or
 * NULL, out [B, Cout, cells]; fp32, contiguous.  The gated tensor is never materialised.
 * Shapes: C * oZ <= 160, Cout <= 80, oZ <= 32 (vamp_gate_conv1x1_supported; else callers keep aten).
 */
int vamp_gate_conv1x1_supported(int32_t C, int32_t oZ, int32_t Cout);
size_t vamp_gate_conv1x1_workspace_bytes(int32_t C, int32_t oZ, int32_t Cout);
int vamp_gate_conv1x1_forward(int64_t B, int32_t C, int32_t oZ, int64_t cells, int32_t Cout,
                              int32_t density_mode, const float* voxel_output,
                              const float* voxel_density, const float* weight, const float* bias,
                              float* out, void* stream);
/* grad_voxel_output [B, C, oZ, cells], grad_voxel_density [B, 1, oZ, cells], grad_weight [Cout, C * oZ] and
 * grad_bias [Cout] (may be NULL) are fully overwritten; deterministic (no float atomics). */
int vamp_gate_conv1x1_backward(int64_t B, int32_t C, int32_t oZ, int64_t cells, int32_t Cout,
                               int32_t density_mode, const float* grad_out, const float* voxel_output,
                               const float* voxel_density, const float* weight,
                               float* grad_voxel_output, float* grad_voxel_density, float* grad_weight,
                               float* grad_bias, void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------
 * BEVDepth-style voxel pooling (north_star's "LSS frustum-to-voxel pooling op").
 * NOT in /root/reference at the pinned commit (its backbones lift with grid_sample = vamp_lift_*); this
 * follows the published BEVDepth operator `voxel_pooling(geom_xyz, input_features, voxel_num)`:
 *   out[b, y, x, :] = sum of feat[b, p, :] over the points p with 0 <= geom[b, p] = (x, y, z) < (nx, ny, nz).
 * geom_xyz [B, P, 3] int32 voxel indices (P = N * D * H * W frustum points), feat [B, P, C] (in_dtype),
 * out [B, ny, nx, C] fp32, fully overwritten (the caller permutes to [B, C, ny, nx] as upstream does).
 * Sort-then-own instead of upstream's float atomics; sums run in list order (last-bit run-to-run
 * variation, like upstream).  Parity unpinned: checked against a numpy scatter-add of the definition.
 * -------------------------------------------------------------------------- */
typedef struct VampPoolDesc {
  int32_t B, C;          /* samples, channels                                   */
  int64_t P;             /* frustum points per sample                           */
  int32_t nx, ny, nz;    /* voxel_num                                           */
  int32_t in_dtype;      /* VAMP_F32 | VAMP_BF16 for feat                       */
} VampPoolDesc;
size_t vamp_voxel_pooling_workspace_bytes(const VampPoolDesc* d);
int vamp_voxel_pooling_forward(const VampPoolDesc* d, const int32_t* geom_xyz, const void* feat, float* out,
                               void* workspace, size_t workspace_bytes, void* stream);
/* grad_feat [B, P, C] fp32 is fully overwritten: the row of the point's cell, zeros outside the grid */
int vamp_voxel_pooling_backward(const VampPoolDesc* d, const int32_t* geom_xyz, const float* grad_out,
                                float* grad_feat, void* stream);

/* --------------------------------------------------------------------------
 * Trilinear resize inside the 3-D UNet between lift and render (first piece):
 * F.interpolate(x, size, mode='trilinear', align_corners=True), base_vampire2.py:66, 72.
 * in [planes, iz, iy, ix] -> out [planes, oz, oy, ox], planes = batch * channels, fp32 contiguous.
 * The backward is a gather over a small per-axis table built in `workspace` (no float atomics);
 * it supports resize factors up to about 6 per axis and returns VAMP_EINVAL beyond.
 * -------------------------------------------------------------------------- */
int vamp_upsample_trilinear_forward(int64_t planes, int32_t iz, int32_t iy, int32_t ix, int32_t oz,
                                    int32_t oy, int32_t ox, const float* in, float* out,
                                    void* stream);
size_t vamp_upsample_trilinear_workspace_bytes(int32_t iz, int32_t iy, int32_t ix);
/* the same with 16-bit tensors (a mixed-precision UNet): dtype = VAMP_F32 | VAMP_BF16 | VAMP_F16 for in / out
   (and grad_out / grad_in below); fp32 arithmetic */
int vamp_upsample_trilinear_forward_ex(int64_t planes, int32_t iz, int32_t iy, int32_t ix, int32_t oz,
                                       int32_t oy, int32_t ox, int32_t dtype, const void* in, void* out,
                                       void* stream);
int vamp_upsample_trilinear_backward_ex(int64_t planes, int32_t iz, int32_t iy, int32_t ix, int32_t oz,
                                        int32_t oy, int32_t ox, int32_t dtype, const void* grad_out, void* grad_in,
                                        void* workspace, size_t workspace_bytes, void* stream);
/* 1 when the backward's gather table covers this scale (about out / in <= 6 per axis), else 0:
 * callers fall back to F.interpolate then */
int vamp_upsample_trilinear_supported(int32_t iz, int32_t iy, int32_t ix, int32_t oz, int32_t oy, int32_t ox);
/* grad_in [planes, iz, iy, ix] is fully overwritten */
int vamp_upsample_trilinear_backward(int64_t planes, int32_t iz, int32_t iy, int32_t ix, int32_t oz,
                                     int32_t oy, int32_t ox, const float* grad_out, float* grad_in,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------
 * 3x3x3 / stride 1 / padding 1 / no-bias Conv3d of the UNet between lift and render (SURVEY 8f
 * N3): nn.Conv3d(cin, cout, 3, 1, 1, bias=False) with cin, cout in {16, 32} -- init_dres, conv2,
 * conv4, conv5, conv6 of the two Hourglass3D blocks, base_vampire2.py:20, 40-60.  fp32 NCDHW
 * tensors, weight [cout, cin, 3, 3, 3]; fp32 matrix cores (exact fp32 products).
 * -------------------------------------------------------------------------- */
typedef struct {
  int32_t B, cin, cout, Z, Y, X;
} VampConvDesc;
/* 1 when the three entry points below accept the descriptor (cin, cout in {16, 32};
   (cin + cout) * X <= 12288 and the LDS row images of the weight gradient fit), else 0 */
int vamp_conv3d_supported(const VampConvDesc* d);
int vamp_conv3d_forward(const VampConvDesc* d, const float* in, const float* weight, float* out,
                        void* stream);
int vamp_conv3d_backward_data(const VampConvDesc* d, const float* grad_out, const float* weight,
                              float* grad_in, void* stream);
size_t vamp_conv3d_workspace_bytes(const VampConvDesc* d);
/* grad_weight [cout, cin, 3, 3, 3] is fully overwritten; workspace holds the per-workgroup partial
   sums (vamp_conv3d_workspace_bytes) */
int vamp_conv3d_backward_weight(const VampConvDesc* d, const float* in, const float* grad_out,
                                float* grad_weight, void* workspace, size_t workspace_bytes,
                                void* stream);

/*
 * The same layers under the reference's `precision=16` (base_cli.py:77; autocast hands the layers of
 * base_vampire2.py:20, 40-60 bf16 activations and weights): bf16 NCDHW in, fp32 accumulate on the bf16
 * matrix cores (v_mfma_f32_16x16x32_bf16), bf16 NCDHW out; weight [cout, cin, 3, 3, 3] bf16; 1 <= cin, cout
 * <= 32, X % 4 == 0 (vamp_conv3d_bf16_supported).  grad_weight is fp32 [cout, cin, 3, 3, 3].
 */
int vamp_conv3d_bf16_supported(const VampConvDesc* d);
int vamp_conv3d_bf16_forward(const VampConvDesc* d, const void* in, const void* weight, void* out, void* stream);
int vamp_conv3d_bf16_backward_data(const VampConvDesc* d, const void* grad_out, const void* weight, void* grad_in,
                                   void* stream);
size_t vamp_conv3d_bf16_workspace_bytes(const VampConvDesc* d);
int vamp_conv3d_bf16_backward_weight(const VampConvDesc* d, const void* in, const void* grad_out, float* grad_weight,
                                     void* workspace, size_t workspace_bytes, void* stream);
/* the same three with the 16-bit type as an argument: dtype = VAMP_BF16 or VAMP_F16 (IEEE half, what Lightning's
   `precision=16` autocasts to: v_mfma_f32_16x16x32_f16); shapes / workspace as the bf16 entry points */
int vamp_conv3d_half_forward(const VampConvDesc* d, int32_t dtype, const void* in, const void* weight, void* out,
                             void* stream);
int vamp_conv3d_half_backward_data(const VampConvDesc* d, int32_t dtype, const void* grad_out, const void* weight,
                                   void* grad_in, void* stream);
int vamp_conv3d_half_backward_weight(const VampConvDesc* d, int32_t dtype, const void* in, const void* grad_out,
                                     float* grad_weight, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAMPIRE_HIP_H_ */
