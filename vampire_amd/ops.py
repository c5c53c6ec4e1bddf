"""Host-side operators of the lift + render hot path (Python over the C ABI).

`HotPath` owns the per-configuration device constants (axis arrays, descriptors)
and exposes the operators with autograd support:

  lift(depth, feat, lift_mats)                       ~ bv2:550-553 + get_voxel_feats
  lift_dense(frustum_feats, lift_mats)               ~ get_voxel_feats signature (bv2:483)
  render(geom|render_mats, density_feature, semantic_logits, base, rgb, beta)
                                                     ~ volume_rendering_from_multiple_views (bv2:391)

All heavy work happens in hand-written HIP kernels reached through ctypes
(`_capi`); torch supplies device memory and the current stream only.
"""
import ctypes as C
import math
import os

import numpy as np

import torch

from . import _capi
from .config import PathConfig
from . import geometry as G


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(stream=None):
    return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _capi.VAMP_F32
    if t.dtype == torch.bfloat16:
        return _capi.VAMP_BF16
    raise TypeError(f"unsupported dtype {t.dtype}: the hot path takes fp32 or bf16 inputs")


def _accept(t):
    """The kernels read fp32 or bf16; anything else (fp16 under the reference's precision=16
    autocast, base_cli.py:77, fp64) is promoted to fp32, as aten's autocast does for these ops."""
    if t is None or t.dtype in (torch.float32, torch.bfloat16):
        return t
    return t.float()


def _is_channel_last(feat: torch.Tensor) -> bool:
    """feat [B, N, C, fH, fW] whose memory is [B, N, fH, fW, C] (a torch.channels_last producer's output, reshaped):
    what the lift wants -- it samples a pixel's C features as one run -- and takes zero-copy."""
    return (feat.dim() == 5 and feat.dtype == torch.float32 and feat.shape[2] > 1
            and feat.permute(0, 1, 3, 4, 2).is_contiguous() and feat.data_ptr() % 16 == 0)


def _chk(t: torch.Tensor, shape, name):
    if not t.is_cuda:
        raise _capi.VampireHipError(f"{name} must be a device tensor (no CPU fallback)")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.contiguous()


class HotPath:
    """Device constants + operators for one PathConfig on one device."""

    def __init__(self, cfg: PathConfig, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.lib = _capi.load()
        geo = G.PathGeometry(cfg)
        dev = self.device
        f32 = torch.float32
        # axis arrays (the kernels rebuild points from these; bit-identical to the buffers)
        self.xs = G.axis_centres(cfg.x_bound_seg).to(dev)
        self.ys = G.axis_centres(cfg.y_bound_seg).to(dev)
        self.zs = G.axis_centres(cfg.z_bound_seg).to(dev)
        self.oxs = G.axis_centres(cfg.x_bound_det).to(dev)
        self.oys = G.axis_centres(cfg.y_bound_det).to(dev)
        self.ozs = G.axis_centres(cfg.z_bound_det).to(dev)
        self.us = geo.frustum[0, 0, :, 0].contiguous().to(dev)
        self.vs = geo.frustum[0, :, 0, 1].contiguous().to(dev)
        self.ds = geo.frustum[:, 0, 0, 2].contiguous().to(dev)
        self.camera_mids = geo.camera_mids.to(dev)
        self.bev_mids = geo.bev_mids.to(dev)
        ozs_cpu = G.axis_centres(cfg.z_bound_det)
        self.ozs_host = (C.c_float * ozs_cpu.numel())(*ozs_cpu.tolist())
        self._ws = {}
        # state of the camera-forward choice ("auto"): mode, call counter, the statistic on the device / in pinned memory
        self._cam_sel = {"mode": "direct", "calls": 0,
                         "dev": torch.full((1,), -1.0, dtype=f32, device=dev),
                         "host": torch.full((1,), -1.0, dtype=f32).pin_memory() if dev.type == "cuda" else torch.full((1,), -1.0)}
        self._dirty = set()         # workspaces whose cell counters a call in flight (or one that raised) may have left non-zero
        # Implementation selectors (host-side state of this object; the library itself keeps none):
        # "cell" = the default cell-list gathers, "v1" = the float-atomic splats kept as independent
        # cross-checks for the tests; lift_wpp forces the lift gather's waves per pixel (0 = auto).
        # The environment gives the initial values only.
        self.impl = {"cam_bwd": os.environ.get("VAMP_CAM_BWD", "cell"),
                     # lift backward: "cell" = cell list + gather (default), "v1" = float-atomic splat
                     "lift_bwd": os.environ.get("VAMP_LIFT_BWD", "cell"),
                     "bev_bwd": os.environ.get("VAMP_BEV_BWD", "cell"),
                     "lift_wpp": int(os.environ.get("VAMP_LIFT_WPP", "0")),
                     # BEV branch on a second stream in training steps (forward-only calls stay on one
                     # stream): eager step 0.679 vs 0.699 ms, replayed from a HIP graph 0.629 vs 0.695 ms
                     "overlap": os.environ.get("VAMP_OVERLAP", "1") == "1",
                     # early ray termination in the camera branch (include/vampire_hip.h)
                     "ert": os.environ.get("VAMP_ERT", "1") != "0",
                     # the camera branch as ONE kernel on the channel-first volumes (no packed copy, no
                     # separate termination pass; render_cam_direct.hip), or the channel-last copy + planned march.
                     # The one kernel wins where rays saturate early (48 against 19 + 26 + ... us) and loses where they
                     # do not (cfg-B: 185 - 222 us against 26 + 126: 64 far-field rays of a wave touch 64 lines per pair
                     # load of a channel-first volume).  "auto" (default): chosen per call from the termination tables
                     # of the last calls (_camera_forward_choice); True / False force one
                     "cam_direct": {"1": True, "0": False}.get(os.environ.get("VAMP_CAM_DIRECT", "auto"), "auto"),
                     # the render forward's two branches as ONE launch (render_fwd_merged.hip: camera tiles first, BEV
                     # column blocks filling the slots the camera tiles' tail leaves idle) wherever the one-kernel camera
                     # forward with early termination runs; False = the two launches (the tests' cross-check: same bits)
                     "fwd_merged": True,
                     # the BEV forward as one kernel (render_bev_fused.hip); False = the two-kernel first
                     # implementation, the cross-check of the tests
                     "bev_fused": True,
                     # forward-only calls: BEV branch on the side stream beside the camera branch.  Off for
                     # eager launches (the fork / join costs more than it hides); a caller that captures the
                     # forward into a HIP graph switches it on (bench.py)
                     "fwd_overlap": False,
                     # training: the camera forward keeps every inside sample's values for the backward's per-ray pass
                     # (+0.5 GB of workspace per sample at cfg-B, touched only where samples are kept); False = that pass
                     # gathers again (the tests' cross-check of the re-sampling path; no environment switch)
                     "save_rows": True,
                     # a training lift forward leaves the scan of its pair cells to the render forward's prepare step,
                     # which scans both operators' cell lists in ONE launch (vamp_render_camera_prepare_with_lift); a lift
                     # backward, or another lift forward, that finds the scan still pending runs it itself
                     # (vamp_lift_finish_cells).  False = the scan inside the lift forward's call (no environment switch)
                     "defer_lift_scan": True}
        self._lift_scan_pending = None      # (desc, workspace, stream) of a lift forward whose cell scan is still due

    # ---------------------------------------------------------------- descs
    def lift_desc(self, B, N, C_, dtype_code, use_depth=True, fhw=None) -> _capi.VampLiftDesc:
        """`fhw`: size of the image feature map when it is not final_dim / downsample_factor -- the lift
        samples it in normalised coordinates (bv2:499-507), so any resolution works: the reference's own
        ResNet-50 + SECONDFPN(upsample_strides=[0.5, 1, 2, 4]) delivers stride-8 maps (32 x 88) beside
        the stride-4 frustum of the renderer."""
        c = self.cfg
        d = _capi.VampLiftDesc()
        d.B, d.N, d.C = B, N, C_
        d.D = c.D if use_depth else 1
        d.fH, d.fW = (c.fH, c.fW) if fhw is None else (int(fhw[0]), int(fhw[1]))
        d.Z, d.Y, d.X = c.vZ, c.vY, c.vX
        d.u_max, d.v_max = float(c.final_dim[1] - 0.5), float(c.final_dim[0] - 0.5)
        d.u_div, d.v_div = float(c.final_dim[1] - 1), float(c.final_dim[0] - 1)
        d.d_lo, d.d_hi = c.d_bound[0], c.d_bound[1]
        d.d_span = c.d_bound[1] - c.d_bound[0]          # python double -> fp32, as torch does
        d.use_depth = 1 if use_depth else 0
        d.in_dtype = dtype_code
        return d

    def render_desc(self, B, N, dtype_code, C_=None) -> _capi.VampRenderDesc:
        c = self.cfg
        d = _capi.VampRenderDesc()
        d.B, d.N = B, N
        d.D, d.fH, d.fW = c.D, c.fH, c.fW
        d.K, d.C = c.num_classes, (c.mid_channels if C_ is None else C_)
        d.Z, d.Y, d.X = c.vZ, c.vY, c.vX
        d.oZ, d.oY, d.oX = c.oZ, c.oY, c.oX
        bounds = (c.x_bound_seg, c.y_bound_seg, c.z_bound_seg)
        for i, b in enumerate(bounds):
            d.lo[i] = b[0]
            d.span[i] = b[1] - b[0]
        d.d_far = c.d_bound[1]
        d.z_step_det = c.z_bound_det[2]
        for i, bnd in enumerate((c.x_bound_det, c.y_bound_det, c.z_bound_det)):
            d.det_step[i] = bnd[2]
        d.density_mode = (_capi.VAMP_DENSITY_SDF_LAPLACE if c.density_mode == "sdf"
                          else _capi.VAMP_DENSITY_SIGMOID)
        d.sdf_bias = c.sdf_bias
        d.beta_min = 1e-4
        d.cat_seg = 1 if c.cat_seg else 0
        d.in_dtype = dtype_code
        return d

    def _side_stream(self):
        """Second HIP stream for the BEV branch of the renderer: it shares no kernel with the camera
        branch, and at batch 1 neither fills the 256 CUs alone (VAMP_OVERLAP=0 disables)."""
        if not self.impl["overlap"]:
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _finish_lift_scan(self):
        """Run the cell scan a training lift forward deferred (impl["defer_lift_scan"]), if it is still due."""
        p, self._lift_scan_pending = self._lift_scan_pending, None
        if p is None:
            return
        d, ws, st = p
        cur = torch.cuda.current_stream()
        if cur != st:
            cur.wait_stream(st)
        _capi.check(self.lib.vamp_lift_finish_cells(C.byref(d), _ptr(ws), ws.numel(), _stream(cur)),
                    "vamp_lift_finish_cells")
        self._dirty.discard("lift")         # scanned: the counters are back at zero

    def _cam_clean_flag(self):
        """VAMP_CAMPREP_COUNTERS_CLEAN when the render workspace's cell counters are known to be zero (fresh
        zero-filled buffer, or the last prepare pass on it was issued in full); marks them in flight."""
        clean = "render" not in self._dirty
        self._dirty.add("render")
        return _capi.VAMP_CAMPREP_COUNTERS_CLEAN if clean else 0

    # ------------------------------------------------- which camera forward (bv2:191-194: both density modes are first class)
    # Two candidates when early termination is allowed: the ONE KERNEL with termination, whose cost follows the
    # samples the rays keep, and the channel-last COPY + PLANNED MARCH of every inside sample without it (the
    # termination pre-pass of the planned march never pays: where rays saturate the one kernel is faster, where
    # they do not the pre-pass marches every sample itself -- cfg-B, sigmoid density: forward 291 us with it, 230
    # without, 257 for the one kernel; training step 899 / 878 / 979).
    # The statistic: mean over the 8 x 8 ray tiles of (largest number of leading samples a ray of the tile keeps)
    # / (samples per ray) -- what the one kernel's cost follows -- from the termination table in the workspace
    # (0.09 on the synthetic sdf workload, 0.69 under the sigmoid density; the two forwards cost the same near
    # 0.5 - 0.6).
    # DETERMINISTIC (round 6): the statistic is formed on the device at fixed call indices (every _PROBE_EVERY-th
    # eager call), copied to pinned host memory behind an event, and APPLIED at a fixed later call index
    # (_PROBE_LAG calls on), where the host waits for that event -- long done by then, so the wait costs nothing.
    # Which call switches is therefore a function of the call index and the data alone: two runs of one seed (and
    # the ranks of a data-parallel job on equal data) take the same kernels in every call.  (Round 5 read the
    # pinned word whenever it happened to have landed: the switch depended on host timing.)
    _PROBE_EVERY = 16            # one-kernel mode: the table is the forward's by-product; the probe is two small reductions
    _PROBE_EVERY_PLANNED = 64    # planned mode leaves no table: the density-only pre-pass (20 - 60 us) is run for the probe
    _PROBE_LAG = 8               # calls between a probe and the call that applies it
    _TO_PLANNED, _TO_DIRECT = 0.5, 0.4                           # hysteresis

    def camera_forward_mode(self):
        """The camera forward the next "auto" call takes ("direct" = one kernel + early termination, "planned" = copy +
        planned march), for logs: bench.py reports it beside the captured step."""
        return self._cam_sel["mode"]

    def _camera_forward_choice(self, can_direct, no_geom):
        """(ert, direct) of this call."""
        ert_allowed = no_geom and self.impl["ert"]
        sel = self.impl["cam_direct"]
        if sel != "auto" or not ert_allowed or not can_direct:
            want = ert_allowed if sel == "auto" else bool(sel)
            return ert_allowed, bool(can_direct and want)
        st = self._cam_sel
        pend = st.get("pending")
        # a probe is applied at ITS call index + lag, never earlier or later (calls under graph capture do not count:
        # a captured graph keeps the variant of its capture)
        if pend is not None and st["calls"] + 1 >= pend[0] and not torch.cuda.is_current_stream_capturing():
            pend[1].synchronize()
            r = float(st["host"][0])
            st["last"] = r
            st["pending"] = None
            if st["mode"] == "direct" and r > self._TO_PLANNED:
                st["mode"] = "planned"
            elif st["mode"] == "planned" and r < self._TO_DIRECT:
                st["mode"] = "direct"
        return (True, True) if st["mode"] == "direct" else (False, False)

    def _camera_forward_probe(self, d, ws, tensors, ert, no_geom, stream):
        """The eager calls of the auto mode at the probe indices: the statistic of this call's termination table."""
        if self.impl["cam_direct"] != "auto" or not (no_geom and self.impl["ert"]):
            return
        if torch.cuda.is_current_stream_capturing():
            return
        st = self._cam_sel
        st["calls"] += 1
        every = self._PROBE_EVERY if ert else max(self._PROBE_EVERY, self._PROBE_EVERY_PLANNED if self._PROBE_EVERY > 1 else 1)
        if (st["calls"] - 1) % every or st.get("pending") is not None:
            return
        c = self.cfg
        if not ert:
            mats, beta, dens = tensors
            _capi.check(self.lib.vamp_render_camera_terminate(
                C.byref(d), _ptr(mats), _ptr(self.us), _ptr(self.vs), _ptr(self.ds), _ptr(beta), _ptr(dens), _ptr(ws),
                ws.numel(), _stream(stream)), "vamp_render_camera_terminate")
        off = self.lib.vamp_render_term_offset(C.byref(d))
        n = d.B * d.N * c.fH * c.fW
        term = ws[off:off + 4 * n].view(torch.int32).view(d.B * d.N, 1, c.fH, c.fW)
        tile_max = torch.nn.functional.max_pool2d(term.float(), 8, ceil_mode=True)
        st["dev"].copy_((tile_max.mean() / float(c.D - 1)).reshape(1))
        st["host"].copy_(st["dev"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        st["pending"] = [st["calls"] + min(self._PROBE_LAG, every - 1 if every > 1 else 1), ev]

    def _workspace(self, key, nbytes):
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            # zero-filled once: the cell counters inside start clean, and every completed pass leaves them
            # clean again (VAMP_LIFTFWD_CELLS_CLEAN / VAMP_CAMPREP_COUNTERS_CLEAN), so no step zeroes them
            t = torch.zeros(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = t
            self._dirty.discard(key)
        return t

    # ----------------------------------------------------------------- lift
    def lift(self, depth, feat, lift_mats, use_depth=True):
        """depth [B,N,D,fH,fW], feat [B,N,C,fH,fW], lift_mats [B,N,3,4,4] -> [B,C,Z,Y,X]."""
        with torch.cuda.device(self.device):      # launches go to this object's device, whatever is current
            return _LiftFn.apply(self, depth, feat, lift_mats, use_depth, torch.is_grad_enabled())

    def lift_logits(self, logits, feat, lift_mats):
        """The lift fed with the raw depth LOGITS [B,N,D,fH,fW] (fp32 | bf16): the softmax of bv2:550 runs
        in the launch that prepares the lift's operands and its backward inside the lift backward's
        gather (producer side) -- same values as ``lift(logits.softmax(2), feat, mats)``."""
        with torch.cuda.device(self.device):
            return _LiftFn.apply(self, logits, feat, lift_mats, True, torch.is_grad_enabled(), True)

    def lift_dense(self, frustum_feats, lift_mats):
        """frustum_feats [B,N,C,D,fH,fW] (materialised, fp32) -> [B,C,Z,Y,X]."""
        return _LiftDenseFn.apply(self, frustum_feats, lift_mats)

    def lift_indices(self, lift_mats, use_depth=True):
        c = self.cfg
        B, N = lift_mats.shape[:2]
        d = self.lift_desc(B, N, 4, _capi.VAMP_F32, use_depth)
        shp = (B, N, c.vZ, c.vY, c.vX)
        valid = torch.empty(shp, dtype=torch.uint8, device=self.device)
        ix0, iy0, iz0 = (torch.empty(shp, dtype=torch.int16, device=self.device) for _ in range(3))
        mats = _chk(lift_mats.float(), (B, N, 3, 4, 4), "lift_mats")
        _capi.check(self.lib.vamp_lift_indices(C.byref(d), _ptr(mats), _ptr(self.xs), _ptr(self.ys),
                                               _ptr(self.zs), _ptr(valid), _ptr(ix0), _ptr(iy0),
                                               _ptr(iz0), _stream()), "vamp_lift_indices")
        return valid, ix0, iy0, iz0

    def lift_cull_words(self, lift_mats, use_depth=True):
        """The forward's camera cull words [B, Z, grid_y, grid_x] (uint32 as int64) and the patch shape
        (px, py) in voxels: bit n clear = no voxel of the patch can be valid in camera n (diagnostic)."""
        c = self.cfg
        B, N = lift_mats.shape[:2]
        d = self.lift_desc(B, N, 4, _capi.VAMP_F32, use_depth)
        patch, grid = (C.c_int32 * 2)(), (C.c_int32 * 2)()
        mats = _chk(lift_mats.float(), (B, N, 3, 4, 4), "lift_mats")
        _capi.check(self.lib.vamp_lift_cull_words(C.byref(d), None, None, None, None, None, patch, grid, None),
                    "vamp_lift_cull_words")
        words = torch.empty(B, c.vZ, grid[1], grid[0], dtype=torch.int32, device=self.device)
        _capi.check(self.lib.vamp_lift_cull_words(C.byref(d), _ptr(mats), _ptr(self.xs), _ptr(self.ys),
                                                  _ptr(self.zs), _ptr(words), patch, grid, _stream()),
                    "vamp_lift_cull_words")
        return words.long() & 0xffffffff, (patch[0], patch[1])

    # --------------------------------------------------------------- render
    def frustum_geometry(self, render_mats):
        c = self.cfg
        B, N = render_mats.shape[:2]
        d = self.render_desc(B, N, _capi.VAMP_F32)
        mats = _chk(render_mats.float(), (B, N, 3, 4, 4), "render_mats")
        geom = torch.empty(B, N, c.D, c.fH, c.fW, 3, dtype=torch.float32, device=self.device)
        _capi.check(self.lib.vamp_frustum_geometry(C.byref(d), _ptr(mats), _ptr(self.us), _ptr(self.vs),
                                                   _ptr(self.ds), _ptr(geom), _stream()),
                    "vamp_frustum_geometry")
        return geom

    def render_indices(self, geom=None, render_mats=None):
        c = self.cfg
        src = geom if geom is not None else render_mats
        B, N = src.shape[:2]
        d = self.render_desc(B, N, _capi.VAMP_F32)
        shp = (B, N, c.D - 1, c.fH, c.fW)
        inside = torch.empty(shp, dtype=torch.uint8, device=self.device)
        ix0, iy0, iz0 = (torch.empty(shp, dtype=torch.int16, device=self.device) for _ in range(3))
        g = None if geom is None else _chk(geom.float(), (B, N, c.D, c.fH, c.fW, 3), "geom")
        m = None if render_mats is None else _chk(render_mats.float(), (B, N, 3, 4, 4), "render_mats")
        _capi.check(self.lib.vamp_render_indices(C.byref(d), _ptr(g), _ptr(m), _ptr(self.us),
                                                 _ptr(self.vs), _ptr(self.ds), _ptr(inside), _ptr(ix0),
                                                 _ptr(iy0), _ptr(iz0), _stream()), "vamp_render_indices")
        return inside, ix0, iy0, iz0

    def render_direct_taps(self, render_mats, coords=False):
        """Diagnostics: (inside, ix0, iy0, iz0[, fxyz]) of every ray sample as the one-kernel camera forward
        evaluates them (vamp_render_camera_direct_taps)."""
        c = self.cfg
        B, N = render_mats.shape[:2]
        d = self.render_desc(B, N, _capi.VAMP_F32)
        shp = (B, N, c.D - 1, c.fH, c.fW)
        inside = torch.empty(shp, dtype=torch.uint8, device=self.device)
        ix0, iy0, iz0 = (torch.empty(shp, dtype=torch.int16, device=self.device) for _ in range(3))
        fxyz = torch.empty(shp + (3,), dtype=torch.float32, device=self.device) if coords else None
        m = _chk(render_mats.float(), (B, N, 3, 4, 4), "render_mats")
        _capi.check(self.lib.vamp_render_camera_direct_taps(C.byref(d), _ptr(m), _ptr(self.us), _ptr(self.vs),
                                                            _ptr(self.ds), _ptr(inside), _ptr(ix0), _ptr(iy0),
                                                            _ptr(iz0), _ptr(fxyz), _stream()),
                    "vamp_render_camera_direct_taps")
        return (inside, ix0, iy0, iz0) + ((fxyz,) if coords else ())

    def render(self, density_feature, semantic_logits, base, rgb, beta=None, *, geom=None,
               render_mats=None):
        """The reference's 8-tuple (bv2:462-467).  Give either ``geom`` [B,N,D,fH,fW,3]
        (API-compatible) or ``render_mats`` [B,N,3,4,4] (geometry evaluated in-kernel)."""
        if (geom is None) == (render_mats is None):
            raise ValueError("give exactly one of geom / render_mats")
        if beta is None:
            beta = torch.zeros((), device=self.device)
            if self.cfg.density_mode == "sdf":
                raise ValueError("density_mode='sdf' needs the beta parameter")
        with torch.cuda.device(self.device):
            return _RenderFn.apply(self, density_feature, semantic_logits, base, rgb, beta, geom,
                                   render_mats, torch.is_grad_enabled())

    def ert_statistics(self, density_feature, beta, render_mats):
        """What early ray termination removes on these inputs: (inside samples, inside samples the
        per-ray table keeps).  Diagnostic for bench.py (the gain is data-dependent); one termination
        pre-pass + the index kernel, a host sync."""
        c = self.cfg
        B, N = render_mats.shape[:2]
        dens = _chk(_accept(density_feature), (B, 1, c.vZ, c.vY, c.vX), "density_feature")
        d = self.render_desc(B, N, _dtype_code(dens))
        mats = _chk(render_mats.float(), (B, N, 3, 4, 4), "render_mats")
        beta = (torch.zeros(1, device=self.device) if beta is None else beta.detach().reshape(1).float().contiguous())
        ws = self._workspace("render", self.lib.vamp_render_workspace_bytes(C.byref(d)))
        with torch.cuda.device(self.device):
            _capi.check(self.lib.vamp_render_camera_terminate(
                C.byref(d), _ptr(mats), _ptr(self.us), _ptr(self.vs), _ptr(self.ds), _ptr(beta), _ptr(dens),
                _ptr(ws), ws.numel(), _stream()), "vamp_render_camera_terminate")
            off = self.lib.vamp_render_term_offset(C.byref(d))
            term = ws[off:off + 4 * B * N * c.fH * c.fW].view(torch.int32).reshape(B, N, 1, c.fH, c.fW).clone()
            inside = self.render_indices(render_mats=mats)[0].bool()
        idx = torch.arange(c.D - 1, device=self.device).reshape(1, 1, -1, 1, 1)
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1      # the workspace no longer matches a saved forward
        return int(inside.sum()), int((inside & (idx < term)).sum())

    # ------------------------------------------------------ point resampling
    def sample_points(self, volume, points, *, padding="zeros", mask_outside=False, activation=False,
                      beta=None, channel_last=False, lattice=None):
        """F.grid_sample(volume, normalised points, padding_mode=padding, align_corners=True) as the
        reference uses it for the occupancy and lidar-point queries (bv2:576-609).

        volume [B,C,Z,Y,X]; points [B,P,3] ego-frame xyz (normalised by the seg bounds inside);
        returns [B,C,P] (or [B,P,C] with channel_last).  ``activation`` samples density(volume)
        (occ_density, bv2:604); ``mask_outside`` multiplies by all(-1 <= n <= 1) (pts_sdf, bv2:595)."""
        if beta is None:
            beta = torch.zeros((), device=self.device)
            if activation and self.cfg.density_mode == "sdf":
                raise ValueError("density_mode='sdf' needs the beta parameter")
        return _SamplePointsFn.apply(self, volume, points, beta, padding, bool(mask_outside),
                                     bool(activation), bool(channel_last), lattice)

    def occupancy_queries(self, semantic_logits, density_feature, occ_coords, bda_mat, beta=None):
        """bv2:596-604: (occ_logits [B,K,oz,oy,ox], occ_density [B,1,oz,oy,ox]) on the occ grid
        rotated by bda[:3,:3] (a tiny torch matmul; the two resamplings are HIP kernels)."""
        B = semantic_logits.shape[0]
        # R @ c for every grid point as one [P,3] x [3,3] product per sample (the reference's
        # broadcast of 640k 3x3 matmuls, bv2:599, costs 8 ms on the GPU)
        if bda_mat is None:      # the static grid of BaseLSSImpaintor / BaseLSS / BaseBiLinear
            pts = occ_coords.reshape(1, -1, 3).float().expand(B, -1, 3).contiguous()
        else:
            pts = torch.matmul(occ_coords.reshape(1, -1, 3).float(), bda_mat[:, :3, :3].float().transpose(1, 2))
        shp = tuple(occ_coords.shape[:3])
        logits = self.sample_points(semantic_logits, pts, padding="border", lattice=shp)
        dens = self.sample_points(density_feature, pts, activation=True, beta=beta, lattice=shp)
        return logits.reshape(B, -1, *shp), dens.reshape(B, 1, *shp)


    # ------------------------------------------------- producer / consumer glue
    def depth_softmax(self, logits):
        """`mapping_along_depth(src).softmax(dim=1)` (bv2:550): logits [B*N, D, fH, fW] fp32 | bf16
        -> fp32 depth distribution of the same shape (what `lift` takes as `depth`)."""
        return _DepthSoftmaxFn.apply(self, logits)

    def density_gate(self, voxel_output, voxel_density):
        """`voxel_output * bev_density.tanh()` (sdf) / `voxel_output * bev_density` (naive),
        bv2:627-630: [B,C,oZ,oY,oX] x [B,1,oZ,oY,oX] -> [B,C,oZ,oY,oX]."""
        return _DensityGateFn.apply(self, voxel_output, voxel_density)


    def gate_conv1x1_supported(self, C_, oZ, cout):
        return bool(self.lib.vamp_gate_conv1x1_supported(int(C_), int(oZ), int(cout)))

    def gate_conv1x1(self, voxel_output, voxel_density, weight, bias=None):
        """The density gate and the `voxel_output` 1x1 conv in one kernel (bv2:627-632; SURVEY 8f N2, consumer
        side): voxel_output [B,C,oZ,oY,oX], voxel_density [B,1,oZ,oY,oX], weight [Cout, C*oZ(,1,1)], bias
        [Cout] | None -> [B,Cout,oY,oX] = conv1x1((voxel_output * gate(voxel_density)).reshape(B, C*oZ, oY, oX))."""
        with torch.cuda.device(self.device):
            return _GateConvFn.apply(self, voxel_output, voxel_density, weight, bias)


# ===========================================================================
# autograd glue
# ===========================================================================
class _LiftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hp: HotPath, depth, feat, mats, use_depth, grad_mode=True, logits=False):
        c = hp.cfg
        B, N, C_ = feat.shape[:3]
        ctx.logits = logits
        ctx.in_dtypes = (depth.dtype if use_depth else None, feat.dtype)
        feat = _accept(feat)
        if logits:
            # `depth` holds the logits (their own dtype); the kernel writes the fp32 distribution
            lg = _chk(_accept(depth), (B, N, c.D) + tuple(feat.shape[-2:]), "depth logits")
            feat = feat.float()
            depth = torch.empty(lg.shape, dtype=torch.float32, device=lg.device)
        elif use_depth:
            depth = _accept(depth)
            if depth.dtype != feat.dtype:               # mixed inputs (e.g. fp32 softmax output + half features)
                depth, feat = depth.float(), feat.float()
        code = _dtype_code(feat)
        fhw = tuple(feat.shape[-2:])
        d = hp.lift_desc(B, N, C_, code, use_depth, fhw=fhw)
        # channel-last fp32 features (the memory of a torch.channels_last producer) go in as they are
        # (VAMP_LIFTFWD_FEAT_CHANNEL_LAST: no transposing first launch); anything else is made [B,N,C,fH,fW]-contiguous
        fcl = _is_channel_last(feat) and (not use_depth or logits or depth.dtype == torch.float32)
        ctx.feat_cl = fcl
        if fcl:
            if not feat.is_cuda:
                raise _capi.VampireHipError("feat must be a device tensor (no CPU fallback)")
            if tuple(feat.shape) != (B, N, C_) + fhw:
                raise ValueError(f"feat: expected shape {(B, N, C_) + fhw}, got {tuple(feat.shape)}")
        else:
            feat = _chk(feat, (B, N, C_) + fhw, "feat")
        mats = _chk(mats.float(), (B, N, 3, 4, 4), "lift_mats")
        out = torch.empty(B, C_, c.vZ, c.vY, c.vX, dtype=torch.float32, device=feat.device)
        # (the caller's grad mode comes in as an argument: inside forward it is always off and
        # needs_input_grad ignores torch.no_grad(), under which the backward's hit words / prepare
        # pass would be wasted work)
        need_grad = grad_mode and (ctx.needs_input_grad[2] or (use_depth and ctx.needs_input_grad[1]))
        nchunk = (C_ + 15) // 16
        hits = (torch.empty(B, c.vZ, c.vY, c.vX, nchunk, dtype=torch.int64, device=feat.device)
                if need_grad else None)
        nbytes = hp.lib.vamp_lift_workspace_bytes(C.byref(d))
        ws = hp._workspace("lift", nbytes)
        cur = torch.cuda.current_stream()
        hp._lift_gen = getattr(hp, "_lift_gen", 0) + 1
        ctx.cells_key = None
        flags = _capi.VAMP_LIFTFWD_FEAT_CHANNEL_LAST if fcl else 0
        if need_grad and hp.impl["lift_bwd"] == "cell":
            # the forward kernel projects every voxel into every camera anyway: in grad mode it also
            # counts the backward's (voxel, camera) pairs per pixel cell and leaves their taps and depth samples in
            # the workspace, so the backward never projects (a backward whose lists another call has overwritten
            # since -- cells_key below -- builds them itself)
            flags |= _capi.VAMP_LIFTFWD_EMIT_PAIRS
            hp._finish_lift_scan()          # (an earlier forward's deferred scan: its counters must be spent first)
            if hp.impl["defer_lift_scan"]:
                flags |= _capi.VAMP_LIFTFWD_DEFER_SCAN
            if "lift" not in hp._dirty:
                flags |= _capi.VAMP_LIFTFWD_CELLS_CLEAN
            hp._dirty.add("lift")           # (until this call has been issued in full)
            ctx.cells_key = (hp._lift_gen, ws.data_ptr())
        if logits:
            _capi.check(hp.lib.vamp_lift_forward_logits_ex(C.byref(d), _ptr(mats), _ptr(hp.xs), _ptr(hp.ys), _ptr(hp.zs),
                                                           _ptr(lg), _dtype_code(lg), _ptr(feat), _ptr(depth), _ptr(out),
                                                           _ptr(hits), _ptr(ws), ws.numel(), flags, _stream(cur)),
                        "vamp_lift_forward_logits_ex")
        else:
            _capi.check(hp.lib.vamp_lift_forward_ex(C.byref(d), _ptr(mats), _ptr(hp.xs), _ptr(hp.ys), _ptr(hp.zs),
                                                    _ptr(depth if use_depth else None), _ptr(feat), _ptr(out),
                                                    _ptr(hits), _ptr(ws), ws.numel(), flags, _stream(cur)),
                        "vamp_lift_forward_ex")
        if flags & _capi.VAMP_LIFTFWD_DEFER_SCAN:
            hp._lift_scan_pending = (d, ws, cur)      # (the counters stay in flight until the scan)
        else:
            hp._dirty.discard("lift")       # emit + scan issued: the counters are back at zero
        if need_grad:
            ctx.hp, ctx.desc, ctx.use_depth = hp, d, use_depth
            ctx.save_for_backward(depth if use_depth else feat, feat, mats, hits)
        return out

    @staticmethod
    def backward(ctx, g):
        hp, d, use_depth = ctx.hp, ctx.desc, ctx.use_depth
        depth, feat, mats, hits = ctx.saved_tensors
        g = g.contiguous().float()
        if ctx.feat_cl:      # the gradient in the features' own (channel-last) memory layout
            B_, N_, C_, fH_, fW_ = feat.shape
            gfeat = torch.empty(B_, N_, fH_, fW_, C_, dtype=torch.float32, device=feat.device).permute(0, 1, 4, 2, 3)
        else:
            gfeat = torch.empty(feat.shape, dtype=torch.float32, device=feat.device)
        gdepth = (torch.empty(depth.shape, dtype=torch.float32, device=feat.device)
                  if use_depth else None)
        nbytes = hp.lib.vamp_lift_workspace_bytes(C.byref(d))
        ws = hp._workspace("lift", nbytes)
        hp._finish_lift_scan()              # (nobody has run the forward's deferred scan: a lift without a render forward)
        valid = 1 if ctx.cells_key == (getattr(hp, "_lift_gen", 0), ws.data_ptr()) else 0
        hp._lift_gen = getattr(hp, "_lift_gen", 0) + 1       # the backward consumes the prepared counters
        if hp.impl["lift_bwd"] == "v1":
            valid = _capi.VAMP_LIFTBWD_SPLAT
        elif ctx.logits:
            valid |= _capi.VAMP_LIFTBWD_LOGITS
        if ctx.feat_cl:
            valid |= _capi.VAMP_LIFTBWD_FEAT_CHANNEL_LAST
        valid |= {1: _capi.VAMP_LIFTBWD_WPP1, 4: _capi.VAMP_LIFTBWD_WPP4,
                  16: _capi.VAMP_LIFTBWD_WPP16}.get(hp.impl["lift_wpp"], 0)
        hp._dirty.add("lift")

        def call(flags, stream):
            _capi.check(hp.lib.vamp_lift_backward_ex(C.byref(d), _ptr(mats), _ptr(hp.xs), _ptr(hp.ys), _ptr(hp.zs),
                                                     _ptr(depth if use_depth else None), _ptr(feat), _ptr(g),
                                                     _ptr(hits), _ptr(gdepth), _ptr(gfeat), _ptr(ws),
                                                     ws.numel(), flags, _stream(stream)), "vamp_lift_backward_ex")

        call(valid, None)
        if hp.impl["lift_bwd"] == "cell":
            hp._dirty.discard("lift")       # the gather has re-zeroed the fill's cursors
        if ctx.logits and hp.impl["lift_bwd"] == "v1":
            # (the cross-check implementation returns the gradient of the distribution)
            gdepth = depth * (gdepth - (depth * gdepth).sum(2, keepdim=True))
        gd = gdepth.to(ctx.in_dtypes[0]) if use_depth else None
        return None, gd, gfeat.to(ctx.in_dtypes[1]), None, None, None, None


class _LiftDenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hp: HotPath, ff, mats):
        c = hp.cfg
        B, N, C_, D = ff.shape[:4]
        use_depth = D > 1
        d = hp.lift_desc(B, N, C_, _capi.VAMP_F32, use_depth, fhw=tuple(ff.shape[-2:]))
        d.D = D
        ff = _chk(ff.float(), (B, N, C_, D) + tuple(ff.shape[-2:]), "frustum_feats")
        mats = _chk(mats.float(), (B, N, 3, 4, 4), "lift_mats")
        out = torch.empty(B, C_, c.vZ, c.vY, c.vX, dtype=torch.float32, device=ff.device)
        nchunk = (C_ + 15) // 16
        hits = (torch.empty(B, c.vZ, c.vY, c.vX, nchunk, dtype=torch.int64, device=ff.device)
                if ctx.needs_input_grad[1] else None)
        _capi.check(hp.lib.vamp_lift_forward_dense(C.byref(d), _ptr(mats), _ptr(hp.xs), _ptr(hp.ys),
                                                   _ptr(hp.zs), _ptr(ff), _ptr(out), _ptr(hits),
                                                   _stream()), "vamp_lift_forward_dense")
        if ctx.needs_input_grad[1]:
            ctx.hp, ctx.desc, ctx.shape = hp, d, ff.shape
            ctx.save_for_backward(mats, hits)
        return out

    @staticmethod
    def backward(ctx, g):
        hp, d = ctx.hp, ctx.desc
        mats, hits = ctx.saved_tensors
        g = g.contiguous().float()
        gff = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        _capi.check(hp.lib.vamp_lift_backward_dense(C.byref(d), _ptr(mats), _ptr(hp.xs), _ptr(hp.ys),
                                                    _ptr(hp.zs), _ptr(g), _ptr(hits), _ptr(gff),
                                                    _stream()), "vamp_lift_backward_dense")
        return None, gff, None


def render_forward_plan(train, two, prep_ok, direct, ert, merged=False):
    """The C calls of one render forward in issue order: [(op, stream, flags, waits, records)].

    op: "term" (termination table), "pack" (channel-last copy of the volumes), "cam" (camera branch: the one
    kernel when `direct`, else the planned march), "prep" (the backward's geometry-only prepare pass), "bev",
    "render" (camera tiles + BEV column blocks in ONE launch, vamp_render_forward_merged: stands for "cam" + "bev";
    only with `merged`, which the caller gives when direct and ert hold and the library supports the shapes).
    stream: "cur" | "side"; flags: beyond the op's base flags; waits / records: names of events.
    `two`: a side stream is available; `prep_ok`: the cell-list backward will follow on matrices (no explicit
    geometry, not the v1 splat); `direct` / `ert`: one-kernel camera forward / early ray termination.
    Every schedule holds exactly one "bev" and one "cam"; the rest is what the measured timelines of
    DESIGN.md 7g-7h settled on (one table instead of four hand-unrolled branches)."""
    F = _capi
    if not (direct and ert):
        merged = False
    if merged and not train:
        return [("render", "cur", 0, (), ())]
    if merged and train and prep_ok:
        # training: the camera tiles also draw the backward's cell ranks, and the call finishes the prepare step (the cell
        # scan, together with the lift's deferred one) behind the launch -- one launch + one small one on ONE stream where
        # rounds 3 - 5 ran the camera kernel, the BEV forward and the prepare pass as three launches on two
        # (the scan on the side stream, with the backward's camera chain behind it there and its BEV chain here, so that
        # the BEV chain would start beside it: 0.411 against 0.395 ms -- it stays on this stream)
        return [("render", "cur", F.VAMP_RENDERFWD_RANK, (), ()),
                ("prep", "cur", F.VAMP_CAMPREP_RANKED, (), ())]
    if train and two and prep_ok and direct:
        # the camera kernel (which leaves the termination table), then the BEV forward; beside them, once the
        # table is there, the prepare pass
        return [("cam", "cur", F.VAMP_CAMFWD_DIRECT, (), ("table",)),
                ("prep", "side", F.VAMP_CAMPREP_TERM_VALID, ("table",), ()),
                ("bev", "cur", 0, (), ())]
    if train and two and prep_ok and ert:
        # planned march: the side stream takes what only the camera branch needs later -- table, copy, prepare
        # pass -- this stream the BEV forward and, once table and copy are there, the march
        return [("term", "side", 0, (), ()),
                ("pack", "side", F.VAMP_CAMFWD_TERM_VALID | F.VAMP_CAMFWD_PACK_ONLY, (), ("packed",)),
                ("prep", "side", F.VAMP_CAMPREP_TERM_VALID, (), ()),
                ("bev", "cur", 0, (), ()),
                ("cam", "cur", F.VAMP_CAMFWD_TERM_VALID | F.VAMP_CAMFWD_PACKED_VALID, ("packed",), ())]
    if direct and not train:
        # forward only (BEV issued first: second, it starts behind the camera kernel and finds the CUs taken)
        return [("bev", "side" if two else "cur", 0, (), ()), ("cam", "cur", F.VAMP_CAMFWD_DIRECT, (), ())]
    steps, camf = [], (F.VAMP_CAMFWD_DIRECT if direct else 0)
    if ert and not direct:
        steps.append(("term", "cur", 0, (), ("table",)))
        camf |= F.VAMP_CAMFWD_TERM_VALID
    prep = train and two and prep_ok and not direct
    # without early termination the march is the long kernel of the step, and a BEV forward beside it costs it more
    # than it hides (207 us beside, 150 alone).  The BEV forward therefore OPENS the side stream: it runs beside the
    # channel-last copy in front of the march (a kernel of 1.5 TB/s: latency, not bytes), with the prepare pass behind
    # it -- replayed step 0.788 against 0.808 ms with the BEV forward behind the march (rounds 4 - 6)
    bev_first = prep and not ert
    if bev_first or not prep:
        steps.append(("bev", "side" if two else "cur", 0, (), ()))
    if prep:
        steps.append(("prep", "side", F.VAMP_CAMPREP_TERM_VALID if ert else 0, ("table",) if ert else (), ()))
    if prep and not bev_first:
        steps.append(("bev", "side", 0, (), ()))
    steps.append(("cam", "cur", camf, (), ()))
    return steps


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hp: HotPath, dens, sem, base, rgb, beta, geom, mats, grad_mode=True):
        c = hp.cfg
        # will a backward follow?  (grad_mode is the caller's: inside forward it is always off, and
        # needs_input_grad ignores torch.no_grad())
        train = grad_mode and any(ctx.needs_input_grad)
        B = dens.shape[0]
        N = (geom if geom is not None else mats).shape[1]
        C_ = base.shape[1]
        ctx.in_dtypes = tuple(t.dtype for t in (dens, sem, base, rgb))
        dens, sem, base, rgb = (_accept(t) for t in (dens, sem, base, rgb))
        if len({t.dtype for t in (dens, sem, base, rgb)}) > 1:
            dens, sem, base, rgb = (t.float() for t in (dens, sem, base, rgb))
        d = hp.render_desc(B, N, _dtype_code(dens), C_)
        vshape = (c.vZ, c.vY, c.vX)
        dens = _chk(dens, (B, 1) + vshape, "density_feature")
        sem = _chk(sem, (B, c.num_classes) + vshape, "semantic_logits")
        base = _chk(base, (B, C_) + vshape, "voxel_features")
        rgb = _chk(rgb, (B, 3) + vshape, "rgb")
        if geom is not None:
            geom = _chk(geom.float(), (B, N, c.D, c.fH, c.fW, 3), "geom_xyz")
        else:
            mats = _chk(mats.float(), (B, N, 3, 4, 4), "render_mats")
        ctx.beta_shape = beta.shape
        beta = beta.reshape(1).float().contiguous()
        dev, f32 = dens.device, torch.float32
        K = c.num_classes
        CO = C_ + (K if c.cat_seg else 0)
        rgb_p = torch.empty(B, N, 3, c.fH, c.fW, dtype=f32, device=dev)
        seg_p = torch.empty(B, N, K, c.fH, c.fW, dtype=f32, device=dev)
        dep_p = torch.empty(B, N, 1, c.fH, c.fW, dtype=f32, device=dev)
        bev_rgb = torch.empty(B, 3, c.oY, c.oX, dtype=f32, device=dev)
        bev_seg = torch.empty(B, K, c.oY, c.oX, dtype=f32, device=dev)
        bev_h = torch.empty(B, 1, c.oY, c.oX, dtype=f32, device=dev)
        vdens = torch.empty(B, 1, c.oZ, c.oY, c.oX, dtype=f32, device=dev)
        vout = torch.empty(B, CO, c.oZ, c.oY, c.oX, dtype=f32, device=dev)
        nbytes = hp.lib.vamp_render_workspace_bytes(C.byref(d))
        # training: the camera forward also keeps every inside sample's gathered values (tile-major rows, 256
        # contiguous bytes per tile, depth index and channel), and the backward's per-ray pass reads them back
        # instead of repeating the 8-tap gathers
        save = bool(train and geom is None and hp.impl["save_rows"] and hp.impl["cam_bwd"] != "v1" and (c.D - 1) <= 128)
        if save:
            nbytes += hp.lib.vamp_render_samples_bytes(C.byref(d))
        ws = hp._workspace("render", nbytes)
        # which camera forward: the one kernel with early termination, or copy + planned march (with the
        # termination pre-pass or without) -- from the switches, or ("auto") from what the rays did lately
        ert, direct = hp._camera_forward_choice(geom is None and (c.D - 1) <= 128, geom is None)
        # the two branches share only their inputs: in a training step they (and the prepare pass) use two
        # streams; forward-only calls stay on one unless the caller captures them into a graph (fwd_overlap)
        cur = torch.cuda.current_stream()
        side = hp._side_stream() if (train or hp.impl["fwd_overlap"]) else None
        prep_ok = geom is None and hp.impl["cam_bwd"] != "v1"
        # camera tiles + BEV column blocks in one launch where the library takes the shapes (it checks the heights
        # hp.ozs_host against the BEV kernel's plane slabs itself)
        merged = bool(direct and ert and hp.impl["fwd_merged"] and hp.impl["bev_fused"] and
                      hp.lib.vamp_render_forward_merged_supported(C.byref(d), hp.ozs_host))
        # training calls take the one launch (camera tiles + rank pass + BEV blocks) while the camera tiles are at most
        # four rounds of the chip's workgroup slots: cfg-B, replayed step, merged against the three launches on two
        # streams -- 1 / 2 / 3 samples per GPU 0.395 / 0.710 / 1.03 against 0.410 / 0.723 / 1.05 ms, 4 samples equal,
        # 8 samples 2.69 against 2.63 (a full chip hides the prepare pass beside the BEV forward better than it
        # fills the camera tiles' tail)
        if merged and train and B * N * ((c.fH + 7) // 8) * ((c.fW + 7) // 8) > int(hp.impl.get("merged_tiles", 4096)):
            merged = False
        plan = render_forward_plan(train, side is not None, prep_ok, direct, ert, merged)
        bev_flags = 0 if hp.impl["bev_fused"] else _capi.VAMP_BEVFWD_TWO_KERNELS
        bev_save = train and hp.impl["bev_bwd"] != "v1"
        ws_bev = hp._workspace("bev", hp.lib.vamp_render_bev_workspace_bytes(C.byref(d))) if bev_save else None
        cam_base = 0 if ert else _capi.VAMP_CAMFWD_NO_ERT
        streams, events = {"cur": cur, "side": side}, {}
        ctx.cells, ctx.ert, ctx.bev_key = False, ert, None
        # the backward's d loss / d beta accumulator is zeroed by the FORWARD: as the first node of the backward the
        # one-element fill sat alone on the step's critical path (5 us + the same again of hand-over behind a kernel
        # that short).  The merged launch zeroes the word itself; the other schedules with a fill on their side stream
        gbeta_in_launch = train and any(op == "render" for op, *_ in plan)
        ctx.gbeta0 = torch.empty(1, dtype=f32, device=dev) if gbeta_in_launch else None
        # (a fork / join with nothing on the other side still costs a replayed graph a barrier: only plans that use the
        # side stream touch it)
        if side is not None and not any(where == "side" for _, where, *_ in plan):
            side = None
            streams["side"] = None
        if side is not None:
            side.wait_stream(cur)
        if train and not gbeta_in_launch:
            if side is not None:
                with torch.cuda.stream(side):
                    ctx.gbeta0 = torch.zeros(1, dtype=f32, device=dev)
                ctx.gbeta0.record_stream(cur)
            else:
                ctx.gbeta0 = torch.zeros(1, dtype=f32, device=dev)
        for op, where, flags, waits, records in plan:
            st = streams[where]
            for w in waits:
                st.wait_event(events[w])
            if op == "term":
                _capi.check(hp.lib.vamp_render_camera_terminate(
                    C.byref(d), _ptr(mats), _ptr(hp.us), _ptr(hp.vs), _ptr(hp.ds), _ptr(beta), _ptr(dens), _ptr(ws),
                    ws.numel(), _stream(st)), "vamp_render_camera_terminate")
            elif op in ("cam", "pack"):
                keep = _capi.VAMP_CAMFWD_SAVE_SAMPLES if (save and op == "cam") else 0
                _capi.check(hp.lib.vamp_render_camera_forward_ex(
                    C.byref(d), _ptr(geom), _ptr(mats), _ptr(hp.us), _ptr(hp.vs), _ptr(hp.ds),
                    _ptr(hp.camera_mids), _ptr(beta), _ptr(dens), _ptr(sem), _ptr(rgb), _ptr(rgb_p),
                    _ptr(seg_p), _ptr(dep_p), _ptr(ws), ws.numel(), cam_base | flags | keep, _stream(st)),
                    "vamp_render_camera_forward_ex")
            elif op == "prep":
                ranked = bool(flags & _capi.VAMP_CAMPREP_RANKED)       # (the forward drew the ranks: scan + work lists only)
                pend = hp._lift_scan_pending
                pflags = flags | (0 if ranked else hp._cam_clean_flag())
                if pend is not None and pend[2] == cur:
                    # a lift forward's pair cells are due too: both scans in one launch.  (The lift ran on `cur`; this
                    # op's stream is `cur` or the side stream, which has waited for `cur` above and which `cur` waits
                    # for below: the lift backward is ordered behind the scan either way.)
                    hp._lift_scan_pending = None
                    _capi.check(hp.lib.vamp_render_camera_prepare_with_lift(
                        C.byref(d), _ptr(mats), _ptr(hp.us), _ptr(hp.vs), _ptr(hp.ds), _ptr(ws), ws.numel(), pflags,
                        C.byref(pend[0]), _ptr(pend[1]), pend[1].numel(), _stream(st)),
                        "vamp_render_camera_prepare_with_lift")
                    hp._dirty.discard("lift")
                else:
                    _capi.check(hp.lib.vamp_render_camera_prepare_ex(
                        C.byref(d), _ptr(mats), _ptr(hp.us), _ptr(hp.vs), _ptr(hp.ds), _ptr(ws), ws.numel(),
                        pflags, _stream(st)), "vamp_render_camera_prepare_ex")
                hp._dirty.discard("render")
                ctx.cells = True
            elif op == "render":
                _capi.check(hp.lib.vamp_render_forward_merged(
                    C.byref(d), _ptr(mats), _ptr(hp.us), _ptr(hp.vs), _ptr(hp.ds), _ptr(hp.camera_mids),
                    _ptr(hp.oxs), _ptr(hp.oys), _ptr(hp.ozs), hp.ozs_host, _ptr(hp.bev_mids), _ptr(beta),
                    _ptr(dens), _ptr(sem), _ptr(rgb), _ptr(base), _ptr(rgb_p), _ptr(seg_p), _ptr(dep_p),
                    _ptr(bev_rgb), _ptr(bev_seg), _ptr(bev_h), _ptr(vdens), _ptr(vout), _ptr(ws), ws.numel(),
                    _ptr(ws_bev), ws_bev.numel() if bev_save else 0, _ptr(ctx.gbeta0) if train else None,
                    (_capi.VAMP_RENDERFWD_SAVE_SAMPLES if save else 0) | (_capi.VAMP_RENDERFWD_BEV_SAVE if bev_save else 0)
                    | flags | (_capi.VAMP_RENDERFWD_COUNTERS_CLEAN if (flags & _capi.VAMP_RENDERFWD_RANK) and hp._cam_clean_flag() else 0),
                    _stream(st)), "vamp_render_forward_merged")
                # (with VAMP_RENDERFWD_RANK the counters stay "in flight" until the "prep" op behind has scanned them)
            else:
                _capi.check(hp.lib.vamp_render_bev_forward_ex(
                    C.byref(d), _ptr(hp.oxs), _ptr(hp.oys), _ptr(hp.ozs), _ptr(hp.bev_mids), _ptr(beta),
                    _ptr(dens), _ptr(sem), _ptr(rgb), _ptr(base), _ptr(bev_rgb), _ptr(bev_seg), _ptr(bev_h),
                    _ptr(vdens), _ptr(vout), hp.ozs_host, _ptr(ws_bev), ws_bev.numel() if bev_save else 0,
                    (_capi.VAMP_BEVFWD_SAVE if bev_save else 0) | bev_flags, _stream(st)), "vamp_render_bev_forward_ex")
            if op in ("render", "bev"):
                hp._bev_gen = getattr(hp, "_bev_gen", 0) + 1
                ctx.bev_key = (hp._bev_gen, ws_bev.data_ptr()) if bev_save else None
            for r in records:
                events[r] = torch.cuda.Event()
                events[r].record(st)
        if side is not None:
            cur.wait_stream(side)
        hp._camera_forward_probe(d, ws, (mats, beta, dens), ert, geom is None, cur)
        ctx.samples = save
        ctx.hp, ctx.desc = hp, d
        # the workspace now holds the termination table, the channel-last copy of (dens, sem, rgb) unless the
        # one kernel ran, and the cell lists if a prepare pass did; the backward reuses them if no other render
        # call has touched the workspace in between
        hp._pack_gen = getattr(hp, "_pack_gen", 0) + 1
        ctx.pack_key = None if (direct and not train) else (hp._pack_gen, ws.data_ptr())
        ctx.packed = not direct                 # (the one-kernel forward makes no channel-last copy)
        ctx.has_geom = geom is not None
        ctx.save_for_backward(dens, sem, base, rgb, beta, geom if geom is not None else mats)
        return rgb_p, seg_p, dep_p, bev_rgb, bev_seg, bev_h, vdens, vout

    @staticmethod
    def backward(ctx, g_rgb, g_seg, g_dep, g_brgb, g_bseg, g_bh, g_vd, g_vo):
        hp, d = ctx.hp, ctx.desc
        dens, sem, base, rgb, beta, gm = ctx.saved_tensors
        geom, mats = (gm, None) if ctx.has_geom else (None, gm)
        f32 = torch.float32
        cont = lambda t: None if t is None else t.contiguous().float()
        g_rgb, g_seg, g_dep, g_brgb, g_bseg, g_bh, g_vd, g_vo = map(
            cont, (g_rgb, g_seg, g_dep, g_brgb, g_bseg, g_bh, g_vd, g_vo))
        nbytes = hp.lib.vamp_render_workspace_bytes(C.byref(d))
        if ctx.samples:
            nbytes += hp.lib.vamp_render_samples_bytes(C.byref(d))
        ws = hp._workspace("render", nbytes)
        ws_bev = hp._workspace("bev", hp.lib.vamp_render_bev_workspace_bytes(C.byref(d)))
        # every gradient buffer is written in full by the calls below (the BEV branch overwrites,
        # the camera branch adds, or the other way round): no zero fills
        gb = torch.empty(base.shape, dtype=f32, device=dens.device)
        gd = torch.empty(dens.shape, dtype=f32, device=dens.device)
        gs = torch.empty(sem.shape, dtype=f32, device=dens.device)
        gr = torch.empty(rgb.shape, dtype=f32, device=dens.device)
        gbeta, ctx.gbeta0 = getattr(ctx, "gbeta0", None), None        # (a second backward of the same graph zeroes its own)
        if gbeta is None:
            gbeta = torch.zeros(1, dtype=f32, device=dens.device)

        bev_saved = (ctx.bev_key is not None and hp.impl["bev_bwd"] != "v1"
                     and ctx.bev_key == (getattr(hp, "_bev_gen", 0), ws_bev.data_ptr()))

        # the axis tables in the BEV workspace depend on the grids only (constants of this object):
        # valid once both halves of a split pair have written theirs into this very buffer
        tab_key = (ws_bev.data_ptr(), d.B)
        tab_valid = hp.impl["bev_bwd"] != "v1" and getattr(hp, "_bev_tab_key", None) == tab_key

        def bev_backward(stream, overwrite_cam, part=0):
            flags = _capi.VAMP_BEVBWD_OVERWRITE_BASE | (_capi.VAMP_BEVBWD_OVERWRITE_CAM if overwrite_cam else 0) | part
            if tab_valid and part:
                flags |= _capi.VAMP_BEVBWD_TABLE_VALID
            if bev_saved:
                flags |= _capi.VAMP_BEVBWD_SAVED_VALID
            _capi.check(hp.lib.vamp_render_bev_backward_ex(
                C.byref(d), _ptr(hp.oxs), _ptr(hp.oys), _ptr(hp.ozs), _ptr(hp.bev_mids), _ptr(beta),
                _ptr(dens), _ptr(sem), _ptr(rgb), _ptr(base), _ptr(g_brgb), _ptr(g_bseg), _ptr(g_bh),
                _ptr(g_vd), _ptr(g_vo), _ptr(gd), _ptr(gs), _ptr(gr), _ptr(gb), _ptr(gbeta),
                None if hp.impl["bev_bwd"] == "v1" else hp.ozs_host, _ptr(ws_bev), ws_bev.numel(),
                flags, _stream(stream)), "vamp_render_bev_backward_ex")

        cam_args = (C.byref(d), _ptr(geom), _ptr(mats), _ptr(hp.us), _ptr(hp.vs), _ptr(hp.ds),
                    _ptr(hp.camera_mids), _ptr(beta), _ptr(dens), _ptr(sem), _ptr(rgb), _ptr(g_rgb),
                    _ptr(g_seg), _ptr(g_dep))
        cur, side = torch.cuda.current_stream(), hp._side_stream()
        default_impl = hp.impl["cam_bwd"] != "v1"
        if not default_impl or geom is not None or not ctx.cells:
            # the v1 splat keeps a packed gradient copy where the cell lists live, and a backward that builds
            # its own cell lists may stop half way: either way the counters there are no longer known zero
            hp._dirty.add("render")
        packed_valid = 2 if ctx.pack_key == (getattr(hp, "_pack_gen", 0), ws.data_ptr()) else 0
        if packed_valid and ctx.cells:
            packed_valid |= 4                                    # VAMP_CAMBWD_CELLS_VALID
        if packed_valid and ctx.samples and default_impl:
            packed_valid |= _capi.VAMP_CAMBWD_SAMPLES_VALID
        if not ctx.ert:
            packed_valid |= _capi.VAMP_CAMBWD_NO_ERT
        elif packed_valid:
            packed_valid |= _capi.VAMP_CAMBWD_TERM_VALID           # same validity as the packed copy
        else:
            packed_valid &= ~4                                   # no table: the cells are rebuilt with a fresh one
        if not ctx.packed:
            packed_valid &= ~2                                   # table / cells are there, a packed copy (v1 splat only) is not
        hp._pack_gen = getattr(hp, "_pack_gen", 0) + 1          # the backward scribbles after the copy only,
        ctx.pack_key = None                                     # but a second backward must not assume so
        if side is not None and geom is None and default_impl:
            # Two streams: the BEV branch writes the buffers on the side stream while the camera
            # branch marches its rays and sorts its samples on this one; the camera gather then
            # waits for the BEV event and adds on top.
            # The pass-through (grad_base) gather of the BEV branch is issued behind the event: nobody
            # waits for grad_base, so it runs beside the camera gather instead of in front of it.
            s_bev, s_cam = side, cur
            side.wait_stream(cur)
            # The camera backward in two calls on its own stream: the ray pass and the heavy cells' per-corner sums
            # (neither touches the gradient buffers), then -- behind the BEV event -- the gather, which adds both on top.
            # Issue order: the camera chain FIRST.  It is the longer chain and the lift backward follows it, and a
            # replayed graph keeps the branch whose first node was created first on the queue of the nodes around the
            # fork -- the other branch pays the cross-queue hand-over (5 - 13 us at its start, and again where it joins).
            cam_flags = 1 | packed_valid

            def cam_part(part, stream, event=None):
                _capi.check(hp.lib.vamp_render_camera_backward_acc(
                    *cam_args, _ptr(gd), _ptr(gs), _ptr(gr), _ptr(gbeta), _ptr(ws), ws.numel(), cam_flags | part,
                    event, _stream(stream)), "vamp_render_camera_backward_acc")

            # (cfg-B replayed: 0.3625 - 0.370 ms/step against 0.3735 with the BEV chain issued first.)
            cam_part(_capi.VAMP_CAMBWD_PART_RAY | _capi.VAMP_CAMBWD_PART_HEAVY, s_cam)
            bev_backward(s_bev, True, _capi.VAMP_BEVBWD_SKIP_BASE)
            done = torch.cuda.Event()
            done.record(s_bev)
            bev_backward(s_bev, True, _capi.VAMP_BEVBWD_ONLY_BASE)
            hp._bev_tab_key = tab_key
            cam_part(_capi.VAMP_CAMBWD_PART_GATHER, s_cam, C.c_void_p(done.cuda_event))
            cur.wait_stream(side)
        elif geom is None and default_impl:
            bev_backward(cur, True)
            _capi.check(hp.lib.vamp_render_camera_backward_acc(
                *cam_args, _ptr(gd), _ptr(gs), _ptr(gr), _ptr(gbeta), _ptr(ws), ws.numel(), 1 | packed_valid,
                None, _stream(cur)), "vamp_render_camera_backward_acc")
        else:
            _capi.check(hp.lib.vamp_render_camera_backward_acc(
                *cam_args, _ptr(gd), _ptr(gs), _ptr(gr), _ptr(gbeta), _ptr(ws), ws.numel(),
                packed_valid | (0 if default_impl else _capi.VAMP_CAMBWD_SPLAT), None, _stream(cur)),
                "vamp_render_camera_backward_acc")
            bev_backward(cur, False)
        grad_beta = gbeta.reshape(ctx.beta_shape) if hp.cfg.density_mode == "sdf" else None
        dt = ctx.in_dtypes
        return (None, gd.to(dt[0]), gs.to(dt[1]), gb.to(dt[2]), gr.to(dt[3]), grad_beta, None, None, None)


class _SamplePointsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hp: HotPath, volume, points, beta, padding, mask_outside, activation, channel_last,
                lattice=None):
        c = hp.cfg
        B, C_ = volume.shape[:2]
        ctx.in_dtype = volume.dtype
        volume = _chk(_accept(volume), (B, C_, c.vZ, c.vY, c.vX), "volume")
        P = points.shape[1]
        points = _chk(points.float(), (B, P, 3), "points")
        d = _capi.VampSampleDesc()
        d.B, d.C, d.Z, d.Y, d.X = B, C_, c.vZ, c.vY, c.vX
        for i, bnd in enumerate((c.x_bound_seg, c.y_bound_seg, c.z_bound_seg)):
            d.lo[i] = bnd[0]
            d.span[i] = bnd[1] - bnd[0]
        if padding not in ("zeros", "border"):
            raise ValueError("padding must be 'zeros' or 'border'")
        d.padding = _capi.VAMP_PAD_BORDER if padding == "border" else _capi.VAMP_PAD_ZEROS
        d.mask_outside = 1 if mask_outside else 0
        d.activation = 1 if activation else 0
        d.density_mode = (_capi.VAMP_DENSITY_SDF_LAPLACE if c.density_mode == "sdf"
                          else _capi.VAMP_DENSITY_SIGMOID)
        d.sdf_bias, d.beta_min = c.sdf_bias, 1e-4
        d.channel_last_out = 1 if channel_last else 0
        d.in_dtype = _dtype_code(volume)
        for i in range(3):
            d.lattice[i] = int(lattice[i]) if lattice is not None else 0
        ctx.beta_shape = beta.shape
        beta = beta.reshape(1).float().contiguous()
        out = torch.empty((B, P, C_) if channel_last else (B, C_, P), dtype=torch.float32,
                          device=volume.device)
        _capi.check(hp.lib.vamp_sample_points_forward(C.byref(d), _ptr(volume), _ptr(beta), _ptr(points),
                                                      P, _ptr(out), _stream()),
                    "vamp_sample_points_forward")
        ctx.hp, ctx.desc, ctx.P, ctx.activation = hp, d, P, activation
        ctx.save_for_backward(volume, points, beta)
        return out

    @staticmethod
    def backward(ctx, g):
        hp, d, P = ctx.hp, ctx.desc, ctx.P
        volume, points, beta = ctx.saved_tensors
        g = g.contiguous().float()
        gvol = torch.empty(volume.shape, dtype=torch.float32, device=volume.device)
        gbeta = torch.zeros(1, dtype=torch.float32, device=volume.device)
        ws = hp._workspace("sample", hp.lib.vamp_sample_points_workspace_bytes(C.byref(d), P))
        _capi.check(hp.lib.vamp_sample_points_backward(
            C.byref(d), _ptr(volume), _ptr(beta), _ptr(points), P, _ptr(g), _ptr(gvol), _ptr(gbeta),
            _ptr(ws), ws.numel(), _stream()), "vamp_sample_points_backward")
        grad_beta = (gbeta.reshape(ctx.beta_shape)
                     if (ctx.activation and hp.cfg.density_mode == "sdf") else None)
        return None, gvol.to(ctx.in_dtype), None, grad_beta, None, None, None, None, None


class _DepthSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hp: HotPath, logits):
        if logits.dim() < 3:
            raise ValueError("logits: expected [images, D, ...]")
        if not logits.is_cuda:
            raise _capi.VampireHipError("logits must be a device tensor (no CPU fallback)")
        ctx.in_dtype = logits.dtype
        logits = _accept(logits).contiguous()
        images, D = logits.shape[:2]
        HW = logits[0, 0].numel()
        out = torch.empty(logits.shape, dtype=torch.float32, device=logits.device)
        if logits.numel():
            _capi.check(hp.lib.vamp_depth_softmax_forward(images, D, HW, _ptr(logits), _dtype_code(logits),
                                                          _ptr(out), _stream()), "vamp_depth_softmax_forward")
        ctx.hp, ctx.dims = hp, (images, D, HW)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        g = g.contiguous().float()
        gx = torch.empty_like(p)
        if p.numel():
            _capi.check(ctx.hp.lib.vamp_depth_softmax_backward(*ctx.dims, _ptr(p), _ptr(g), _ptr(gx), _stream()),
                        "vamp_depth_softmax_backward")
        return None, gx.to(ctx.in_dtype)


class _DensityGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hp: HotPath, vo, vd):
        if not (vo.is_cuda and vd.is_cuda):
            raise _capi.VampireHipError("voxel_output / voxel_density must be device tensors (no CPU fallback)")
        B, C_ = vo.shape[:2]
        vd = _chk(vd.float(), (B, 1) + tuple(vo.shape[2:]), "voxel_density")
        vo = vo.float().contiguous()
        cells = vd[0].numel()
        mode = (_capi.VAMP_DENSITY_SDF_LAPLACE if hp.cfg.density_mode == "sdf" else _capi.VAMP_DENSITY_SIGMOID)
        out = torch.empty_like(vo)
        if vo.numel():
            _capi.check(hp.lib.vamp_density_gate_forward(B, C_, cells, mode, _ptr(vo), _ptr(vd), _ptr(out),
                                                         _stream()), "vamp_density_gate_forward")
        ctx.hp, ctx.dims = hp, (B, C_, cells, mode)
        ctx.save_for_backward(vo, vd)
        return out

    @staticmethod
    def backward(ctx, g):
        vo, vd = ctx.saved_tensors
        g = g.contiguous().float()
        gvo, gvd = torch.empty_like(vo), torch.empty_like(vd)
        if vo.numel():
            _capi.check(ctx.hp.lib.vamp_density_gate_backward(*ctx.dims, _ptr(g), _ptr(vo), _ptr(vd), _ptr(gvo),
                                                              _ptr(gvd), _stream()), "vamp_density_gate_backward")
        return None, gvo, gvd


class _GateConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hp: HotPath, vo, vd, weight, bias):
        if not (vo.is_cuda and vd.is_cuda and weight.is_cuda):
            raise _capi.VampireHipError("gate_conv1x1 needs device tensors (no CPU fallback)")
        B, C_, oZ = vo.shape[:3]
        plane = tuple(vo.shape[3:])
        cells = int(math.prod(plane))
        cout = weight.shape[0]
        vd = _chk(vd.float(), (B, 1, oZ) + plane, "voxel_density")
        vo = vo.float().contiguous()
        w = _chk(weight.float().reshape(cout, -1), (cout, C_ * oZ), "weight")
        bs = None if bias is None else _chk(bias.float(), (cout,), "bias")
        mode = (_capi.VAMP_DENSITY_SDF_LAPLACE if hp.cfg.density_mode == "sdf" else _capi.VAMP_DENSITY_SIGMOID)
        out = torch.empty((B, cout) + plane, dtype=torch.float32, device=vo.device)
        dims = (B, C_, oZ, cells, cout, mode)
        _capi.check(hp.lib.vamp_gate_conv1x1_forward(*dims, _ptr(vo), _ptr(vd), _ptr(w), _ptr(bs), _ptr(out),
                                                     _stream()), "vamp_gate_conv1x1_forward")
        ctx.hp, ctx.dims, ctx.wshape, ctx.has_bias = hp, dims, tuple(weight.shape), bias is not None
        ctx.dtypes = (weight.dtype, None if bias is None else bias.dtype)
        ctx.save_for_backward(vo, vd, w)
        return out

    @staticmethod
    def backward(ctx, g):
        hp = ctx.hp
        vo, vd, w = ctx.saved_tensors
        g = g.contiguous().float()
        gvo, gvd, gw = torch.empty_like(vo), torch.empty_like(vd), torch.empty_like(w)
        gb = torch.empty(w.shape[0], dtype=torch.float32, device=w.device) if ctx.has_bias else None
        B, C_, oZ, cells, cout, mode = ctx.dims
        ws = hp._workspace("gate_conv", hp.lib.vamp_gate_conv1x1_workspace_bytes(C_, oZ, cout))
        _capi.check(hp.lib.vamp_gate_conv1x1_backward(*ctx.dims, _ptr(g), _ptr(vo), _ptr(vd), _ptr(w), _ptr(gvo),
                                                      _ptr(gvd), _ptr(gw), _ptr(gb), _ptr(ws), ws.numel(), _stream()),
                    "vamp_gate_conv1x1_backward")
        return (None, gvo, gvd, gw.reshape(ctx.wshape).to(ctx.dtypes[0]),
                None if gb is None else gb.to(ctx.dtypes[1]))


# ===========================================================================
# BEVDepth-style voxel pooling (north_star; SURVEY 8 row a11 -- not in the reference tree, parity unpinned)
# ===========================================================================
_pool_ws = {}


def voxel_pooling(geom_xyz, input_features, voxel_num):
    """The published BEVDepth operator `voxel_pooling(geom_xyz, input_features, voxel_num)`:
    geom_xyz [B, N, D, H, W, 3] integer voxel indices (x, y, z), input_features [B, N, D, H, W, C]
    (fp32 | bf16), voxel_num (nx, ny, nz) -> [B, C, ny, nx] fp32, the sum of the features of the points
    falling into each BEV cell (points outside the grid are dropped).  HIP kernels, no CPU fallback."""
    return _VoxelPoolingFn.apply(geom_xyz, input_features, tuple(int(v) for v in voxel_num))


class _VoxelPoolingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geom, feat, voxel_num):
        if not (geom.is_cuda and feat.is_cuda):
            raise _capi.VampireHipError("voxel_pooling needs device tensors (no CPU fallback)")
        lib = _capi.load()
        B, C_ = feat.shape[0], feat.shape[-1]
        P = feat[0].numel() // C_
        ctx.in_dtype, ctx.fshape = feat.dtype, tuple(feat.shape)
        feat = _accept(feat).reshape(B, P, C_).contiguous()
        geom = _chk(geom.reshape(B, P, 3).to(torch.int32), (B, P, 3), "geom_xyz")
        d = _capi.VampPoolDesc(B, C_, P, voxel_num[0], voxel_num[1], voxel_num[2], _dtype_code(feat))
        out = torch.empty(B, voxel_num[1], voxel_num[0], C_, dtype=torch.float32, device=feat.device)
        nbytes = lib.vamp_voxel_pooling_workspace_bytes(C.byref(d))
        key = (feat.device, torch.cuda.current_stream().cuda_stream)
        ws = _pool_ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _pool_ws[key] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=feat.device)
        with torch.cuda.device(feat.device):
            _capi.check(lib.vamp_voxel_pooling_forward(C.byref(d), _ptr(geom), _ptr(feat), _ptr(out), _ptr(ws),
                                                       ws.numel(), _stream()), "vamp_voxel_pooling_forward")
        ctx.desc = d
        ctx.save_for_backward(geom)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        (geom,) = ctx.saved_tensors
        d = ctx.desc
        g = g.permute(0, 2, 3, 1).contiguous().float()
        gfeat = torch.empty(d.B, d.P, d.C, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _capi.check(_capi.load().vamp_voxel_pooling_backward(C.byref(d), _ptr(geom), _ptr(g), _ptr(gfeat), _stream()),
                        "vamp_voxel_pooling_backward")
        return None, gfeat.reshape(ctx.fshape).to(ctx.in_dtype), None


# ===========================================================================
# trilinear resize of the 3-D UNet (first piece)
# ===========================================================================
_resize_ws = {}


def upsample_trilinear(x, size):
    """F.interpolate(x, size, mode='trilinear', align_corners=True) (bv2:66, 72) on the HIP
    kernels: x [B,C,z,y,x] fp32 device tensor -> [B,C,*size]; the backward is a gather (aten's
    float-atomic scatter takes 2.2 ms per call at the UNet's full-resolution level)."""
    return _UpsampleTrilinearFn.apply(x, tuple(int(v) for v in size))


class _UpsampleTrilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size):
        if not x.is_cuda:
            raise _capi.VampireHipError("x must be a device tensor (no CPU fallback)")
        if x.dim() != 5 or len(size) != 3:
            raise ValueError("expected a [B,C,z,y,x] tensor and a 3-tuple size")
        lib = _capi.load()
        ctx.in_dtype = x.dtype
        codes = {torch.float32: _capi.VAMP_F32, torch.bfloat16: _capi.VAMP_BF16, torch.float16: _capi.VAMP_F16}
        if x.dtype not in codes:
            x = x.float()
        x = x.contiguous()
        ctx.code = codes[x.dtype]
        B, C_ = x.shape[:2]
        out = torch.empty((B, C_) + size, dtype=x.dtype, device=x.device)
        if out.numel() and x.numel():
            _capi.check(lib.vamp_upsample_trilinear_forward_ex(B * C_, *x.shape[2:], *size, ctx.code, _ptr(x), _ptr(out),
                                                               _stream()), "vamp_upsample_trilinear_forward_ex")
        ctx.lib, ctx.dims = lib, (B * C_,) + tuple(x.shape[2:]) + size
        ctx.in_shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        dt = {_capi.VAMP_F32: torch.float32, _capi.VAMP_BF16: torch.bfloat16, _capi.VAMP_F16: torch.float16}[ctx.code]
        g = g.contiguous().to(dt)
        gin = torch.empty(ctx.in_shape, dtype=dt, device=g.device)
        if gin.numel() and g.numel():
            lib = ctx.lib
            nbytes = lib.vamp_upsample_trilinear_workspace_bytes(*ctx.dims[1:4])
            key = (g.device, torch.cuda.current_stream().cuda_stream, nbytes)   # one table per stream
            ws = _resize_ws.get(key)
            if ws is None:
                ws = _resize_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
            _capi.check(lib.vamp_upsample_trilinear_backward_ex(*ctx.dims, ctx.code, _ptr(g), _ptr(gin), _ptr(ws), ws.numel(),
                                                                _stream()), "vamp_upsample_trilinear_backward_ex")
        else:
            gin.zero_()
        return gin.to(ctx.in_dtype), None


# ===========================================================================
# 3x3x3 convolutions of the 3-D UNet
# ===========================================================================
def conv3d_3x3x3(x, weight):
    """nn.Conv3d(cin, cout, 3, 1, 1, bias=False) (bv2:20, 40-60) on the fp32 matrix cores:
    x [B,cin,Z,Y,X], weight [cout,cin,3,3,3], cin / cout in {16, 32}; fp32 device tensors."""
    return _Conv3dFn.apply(x, weight)


def conv3d_bf16(x, weight):
    """The same layer in bf16 (what the reference's `precision=16` training hands it): x bf16 [B,cin,Z,Y,X],
    weight bf16 [cout,cin,3,3,3] -> bf16 [B,cout,Z,Y,X]; fp32 accumulation on the bf16 matrix cores."""
    return _Conv3dBf16Fn.apply(x, weight)


def conv3d_bf16_supported(x, weight, stride, padding, bias):
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and bias is None and tuple(stride) == (1, 1, 1)
            and tuple(padding) == (1, 1, 1) and tuple(weight.shape[2:]) == (3, 3, 3) and x.dim() == 5
            and weight.shape[1] == x.shape[1]):
        return False
    d = _capi.VampConvDesc()
    d.B, d.cin, d.Z, d.Y, d.X = x.shape
    d.cout = weight.shape[0]
    return bool(_capi.load().vamp_conv3d_bf16_supported(C.byref(d)))


class _Conv3dBf16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        if not (x.is_cuda and w.is_cuda):
            raise _capi.VampireHipError("x / weight must be device tensors (no CPU fallback)")
        if x.dtype not in (torch.bfloat16, torch.float16) or w.dtype != x.dtype or x.dim() != 5 or w.dim() != 5:
            raise TypeError("conv3d_bf16 takes bf16 (or fp16) [B,cin,Z,Y,X] and [cout,cin,3,3,3] tensors of one dtype")
        code = _capi.VAMP_BF16 if x.dtype == torch.bfloat16 else _capi.VAMP_F16
        lib = _capi.load()
        x, w = x.contiguous(), w.contiguous()
        d = _capi.VampConvDesc()
        d.B, d.cin, d.Z, d.Y, d.X = x.shape
        d.cout = w.shape[0]
        if w.shape[1] != d.cin:
            raise ValueError("weight / input channel mismatch")
        out = torch.empty((d.B, d.cout, d.Z, d.Y, d.X), dtype=x.dtype, device=x.device)
        _capi.check(lib.vamp_conv3d_half_forward(C.byref(d), code, _ptr(x), _ptr(w), _ptr(out), _stream()),
                    "vamp_conv3d_half_forward")
        ctx.lib, ctx.desc, ctx.code = lib, d, code
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        lib, d = ctx.lib, ctx.desc
        g = g.contiguous().to(x.dtype)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _capi.check(lib.vamp_conv3d_half_backward_data(C.byref(d), ctx.code, _ptr(g), _ptr(w), _ptr(gx), _stream()),
                        "vamp_conv3d_half_backward_data")
        if ctx.needs_input_grad[1]:
            gw32 = torch.empty(w.shape, dtype=torch.float32, device=w.device)
            nbytes = lib.vamp_conv3d_bf16_workspace_bytes(C.byref(d))
            key = (g.device, torch.cuda.current_stream().cuda_stream, "conv16", nbytes)
            ws = _resize_ws.get(key)
            if ws is None:
                ws = _resize_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
            _capi.check(lib.vamp_conv3d_half_backward_weight(C.byref(d), ctx.code, _ptr(x), _ptr(g), _ptr(gw32), _ptr(ws),
                                                             ws.numel(), _stream()), "vamp_conv3d_half_backward_weight")
            gw = gw32.to(w.dtype)                 # the gradient of the 16-bit copy autocast made of the fp32 parameter
        return gx, gw


def conv3d_supported(x, weight, stride, padding, bias):
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and bias is None
            and tuple(stride) == (1, 1, 1) and tuple(padding) == (1, 1, 1) and tuple(weight.shape[2:]) == (3, 3, 3)
            and x.dim() == 5 and weight.shape[1] == x.shape[1] and not torch.is_autocast_enabled()):
        return False
    d = _capi.VampConvDesc()
    d.B, d.cin, d.Z, d.Y, d.X = x.shape
    d.cout = weight.shape[0]
    return bool(_capi.load().vamp_conv3d_supported(C.byref(d)))


class _Conv3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        if not (x.is_cuda and w.is_cuda):
            raise _capi.VampireHipError("x / weight must be device tensors (no CPU fallback)")
        if x.dtype != torch.float32 or w.dtype != torch.float32 or x.dim() != 5 or w.dim() != 5:
            raise TypeError("conv3d_3x3x3 takes fp32 [B,cin,Z,Y,X] and [cout,cin,3,3,3] tensors")
        lib = _capi.load()
        x, w = x.contiguous(), w.contiguous()
        d = _capi.VampConvDesc()
        d.B, d.cin, d.Z, d.Y, d.X = x.shape
        d.cout = w.shape[0]
        if w.shape[1] != d.cin:
            raise ValueError("weight / input channel mismatch")
        out = torch.empty((d.B, d.cout, d.Z, d.Y, d.X), dtype=torch.float32, device=x.device)
        _capi.check(lib.vamp_conv3d_forward(C.byref(d), _ptr(x), _ptr(w), _ptr(out), _stream()),
                    "vamp_conv3d_forward")
        ctx.lib, ctx.desc = lib, d
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        lib, d = ctx.lib, ctx.desc
        g = g.contiguous().float()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _capi.check(lib.vamp_conv3d_backward_data(C.byref(d), _ptr(g), _ptr(w), _ptr(gx), _stream()),
                        "vamp_conv3d_backward_data")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            nbytes = lib.vamp_conv3d_workspace_bytes(C.byref(d))
            key = (g.device, torch.cuda.current_stream().cuda_stream, "conv", nbytes)
            ws = _resize_ws.get(key)
            if ws is None:
                ws = _resize_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
            _capi.check(lib.vamp_conv3d_backward_weight(C.byref(d), _ptr(x), _ptr(g), _ptr(gw), _ptr(ws),
                                                        ws.numel(), _stream()), "vamp_conv3d_backward_weight")
        return gx, gw
