"""ctypes binding of include/vampire_hip.h (the C-ABI HIP library).

There is deliberately no fallback: if the shared library is missing or a symbol
does not resolve, importing the ops raises.  The product path never routes
through oracle/ or any CPU implementation.
"""
import ctypes as C
import os

from .build import lib_path

ABI_VERSION = 6

VAMP_F32, VAMP_BF16, VAMP_F16 = 0, 1, 2
VAMP_DENSITY_SIGMOID, VAMP_DENSITY_SDF_LAPLACE = 0, 1


class VampLiftDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("C", C.c_int32),
                ("D", C.c_int32), ("fH", C.c_int32), ("fW", C.c_int32),
                ("Z", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32),
                ("u_max", C.c_float), ("v_max", C.c_float),
                ("u_div", C.c_float), ("v_div", C.c_float),
                ("d_lo", C.c_float), ("d_hi", C.c_float), ("d_span", C.c_float),
                ("use_depth", C.c_int32), ("in_dtype", C.c_int32)]


class VampPoolDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("P", C.c_int64),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("in_dtype", C.c_int32)]


class VampRenderDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32),
                ("D", C.c_int32), ("fH", C.c_int32), ("fW", C.c_int32),
                ("K", C.c_int32), ("C", C.c_int32),
                ("Z", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32),
                ("oZ", C.c_int32), ("oY", C.c_int32), ("oX", C.c_int32),
                ("lo", C.c_float * 3), ("span", C.c_float * 3),
                ("d_far", C.c_float), ("z_step_det", C.c_float), ("det_step", C.c_float * 3),
                ("density_mode", C.c_int32), ("sdf_bias", C.c_float), ("beta_min", C.c_float),
                ("cat_seg", C.c_int32), ("in_dtype", C.c_int32)]


class VampConvDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("Z", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32)]


class VampSampleDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32),
                ("Z", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32),
                ("lo", C.c_float * 3), ("span", C.c_float * 3),
                ("padding", C.c_int32), ("mask_outside", C.c_int32), ("activation", C.c_int32),
                ("density_mode", C.c_int32), ("sdf_bias", C.c_float), ("beta_min", C.c_float),
                ("channel_last_out", C.c_int32), ("in_dtype", C.c_int32), ("lattice", C.c_int32 * 3)]


VAMP_PAD_ZEROS, VAMP_PAD_BORDER = 0, 1
# flag bits of vamp_lift_backward_ex / vamp_render_camera_backward_acc (include/vampire_hip.h)
VAMP_LIFTFWD_EMIT_PAIRS, VAMP_LIFTFWD_CELLS_CLEAN, VAMP_LIFTFWD_FEAT_CHANNEL_LAST, VAMP_LIFTFWD_DEFER_SCAN = 1, 2, 4, 8
VAMP_LIFTBWD_CELLS_VALID, VAMP_LIFTBWD_SPLAT = 1, 2
VAMP_LIFTBWD_WPP1, VAMP_LIFTBWD_WPP4, VAMP_LIFTBWD_WPP16 = 4, 8, 16
VAMP_LIFTBWD_LOGITS, VAMP_LIFTBWD_FEAT_CHANNEL_LAST = 256, 512
VAMP_CAMBWD_ACCUMULATE, VAMP_CAMBWD_PACKED_VALID, VAMP_CAMBWD_CELLS_VALID, VAMP_CAMBWD_SPLAT = 1, 2, 4, 8
VAMP_CAMBWD_SAMPLES_VALID, VAMP_CAMBWD_TERM_VALID, VAMP_CAMBWD_NO_ERT = 16, 32, 64
VAMP_CAMBWD_PART_RAY, VAMP_CAMBWD_PART_GATHER, VAMP_CAMBWD_PART_HEAVY = 128, 256, 512
VAMP_CAMFWD_SAVE_SAMPLES, VAMP_CAMFWD_NO_ERT, VAMP_CAMFWD_TERM_VALID = 1, 2, 4
VAMP_CAMPREP_TERM_VALID, VAMP_CAMPREP_COUNTERS_CLEAN, VAMP_CAMPREP_RANKED = 1, 4, 8
VAMP_BEVBWD_OVERWRITE_BASE, VAMP_BEVBWD_OVERWRITE_CAM, VAMP_BEVBWD_SAVED_VALID = 1, 2, 4
VAMP_BEVFWD_SAVE, VAMP_BEVFWD_TWO_KERNELS = 1, 2
VAMP_RENDERFWD_SAVE_SAMPLES, VAMP_RENDERFWD_BEV_SAVE, VAMP_RENDERFWD_RANK, VAMP_RENDERFWD_COUNTERS_CLEAN = 1, 2, 4, 8
VAMP_CAMFWD_PACK_ONLY, VAMP_CAMFWD_PACKED_VALID, VAMP_CAMFWD_DIRECT, VAMP_CAMFWD_EXACT_TAPS = 8, 16, 32, 64
VAMP_BEVBWD_ONLY_BASE, VAMP_BEVBWD_SKIP_BASE, VAMP_BEVBWD_TABLE_VALID = 8, 16, 32

_P = C.c_void_p
_LD = C.POINTER(VampLiftDesc)
_RD = C.POINTER(VampRenderDesc)
_SD = C.POINTER(VampSampleDesc)
_CD = C.POINTER(VampConvDesc)

# name -> (restype, argtypes); must list every symbol declared in include/vampire_hip.h
SIGNATURES = {
    "vamp_abi_version": (C.c_int, []),
    "vamp_debug_checks": (C.c_int, [C.c_int]),
    "vamp_last_error": (C.c_char_p, []),
    "vamp_profile_enable": (C.c_int, [C.c_int]),
    "vamp_profile_slots": (C.c_int, []),
    "vamp_profile_select": (C.c_int, [C.c_int]),
    "vamp_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                    C.POINTER(C.c_double)]),
    "vamp_lift_workspace_bytes": (C.c_size_t, [_LD]),
    "vamp_lift_forward": (C.c_int, [_LD] + [_P] * 8 + [_P, C.c_size_t, _P]),
    "vamp_lift_forward_logits": (C.c_int, [_LD] + [_P] * 5 + [C.c_int32] + [_P] * 4 + [_P, C.c_size_t, _P]),
    "vamp_lift_forward_ex": (C.c_int, [_LD] + [_P] * 8 + [_P, C.c_size_t, C.c_int, _P]),
    "vamp_lift_forward_logits_ex": (C.c_int, [_LD] + [_P] * 5 + [C.c_int32] + [_P] * 4 + [_P, C.c_size_t, C.c_int, _P]),
    "vamp_lift_backward": (C.c_int, [_LD] + [_P] * 10 + [_P, C.c_size_t, _P]),
    "vamp_lift_backward_ex": (C.c_int, [_LD] + [_P] * 10 + [_P, C.c_size_t, C.c_int, _P]),
    "vamp_lift_prepare": (C.c_int, [_LD] + [_P] * 5 + [_P, C.c_size_t, _P]),
    "vamp_lift_finish_cells": (C.c_int, [_LD, _P, C.c_size_t, _P]),
    "vamp_render_camera_prepare_with_lift": (C.c_int, [_RD] + [_P] * 4 + [_P, C.c_size_t, C.c_int, _LD, _P, C.c_size_t, _P]),
    "vamp_lift_forward_dense": (C.c_int, [_LD] + [_P] * 7 + [_P]),
    "vamp_lift_backward_dense": (C.c_int, [_LD] + [_P] * 7 + [_P]),
    "vamp_lift_indices": (C.c_int, [_LD] + [_P] * 8 + [_P]),
    "vamp_lift_cull_words": (C.c_int, [_LD] + [_P] * 7 + [_P]),
    "vamp_render_workspace_bytes": (C.c_size_t, [_RD]),
    "vamp_render_camera_forward": (C.c_int, [_RD] + [_P] * 13 + [_P, C.c_size_t, _P]),
    "vamp_render_samples_bytes": (C.c_size_t, [_RD]),
    "vamp_render_term_offset": (C.c_size_t, [_RD]),
    "vamp_render_camera_terminate": (C.c_int, [_RD] + [_P] * 6 + [_P, C.c_size_t, _P]),
    "vamp_render_camera_prepare_ex": (C.c_int, [_RD] + [_P] * 4 + [_P, C.c_size_t, C.c_int, _P]),
    "vamp_render_camera_forward_ex": (C.c_int, [_RD] + [_P] * 13 + [_P, C.c_size_t, C.c_int, _P]),
    "vamp_render_camera_backward": (C.c_int, [_RD] + [_P] * 17 + [_P, C.c_size_t, _P]),
    "vamp_render_camera_prepare": (C.c_int, [_RD] + [_P] * 4 + [_P, C.c_size_t, _P]),
    "vamp_render_camera_backward_acc": (C.c_int, [_RD] + [_P] * 17 + [_P, C.c_size_t, C.c_int, _P, _P]),
    "vamp_render_bev_forward": (C.c_int, [_RD] + [_P] * 14 + [_P]),
    "vamp_render_bev_forward_ex": (C.c_int, [_RD] + [_P] * 14 + [C.POINTER(C.c_float), _P, C.c_size_t, C.c_int, _P]),
    "vamp_render_forward_merged_supported": (C.c_int, [_RD, C.POINTER(C.c_float)]),
    "vamp_render_forward_merged": (C.c_int, [_RD] + [_P] * 8 + [C.POINTER(C.c_float)] + [_P] * 14
                                   + [_P, C.c_size_t, _P, C.c_size_t, _P, C.c_int, _P]),
    "vamp_render_bev_workspace_bytes": (C.c_size_t, [_RD]),
    "vamp_render_bev_backward": (C.c_int, [_RD] + [_P] * 19 + [C.POINTER(C.c_float), _P, C.c_size_t, _P]),
    "vamp_render_bev_backward_ex": (C.c_int, [_RD] + [_P] * 19 + [C.POINTER(C.c_float), _P, C.c_size_t, C.c_int, _P]),
    "vamp_render_indices": (C.c_int, [_RD] + [_P] * 9 + [_P]),
    "vamp_render_camera_direct_taps": (C.c_int, [_RD] + [_P] * 9 + [_P]),
    "vamp_frustum_geometry": (C.c_int, [_RD] + [_P] * 5 + [_P]),
    "vamp_sample_points_forward": (C.c_int, [_SD, _P, _P, _P, C.c_int64, _P, _P]),
    "vamp_sample_points_workspace_bytes": (C.c_size_t, [_SD, C.c_int64]),
    "vamp_sample_points_backward": (C.c_int, [_SD, _P, _P, _P, C.c_int64, _P, _P, _P, _P, C.c_size_t, _P]),
    "vamp_depth_softmax_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_int64, _P, C.c_int32, _P, _P]),
    "vamp_depth_softmax_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_int64, _P, _P, _P, _P]),
    "vamp_density_gate_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    "vamp_upsample_trilinear_forward": (C.c_int, [C.c_int64] + [C.c_int32] * 6 + [_P, _P, _P]),
    "vamp_upsample_trilinear_forward_ex": (C.c_int, [C.c_int64] + [C.c_int32] * 7 + [_P, _P, _P]),
    "vamp_upsample_trilinear_backward_ex": (C.c_int, [C.c_int64] + [C.c_int32] * 7 + [_P, _P, _P, C.c_size_t, _P]),
    "vamp_upsample_trilinear_workspace_bytes": (C.c_size_t, [C.c_int32] * 3),
    "vamp_upsample_trilinear_supported": (C.c_int, [C.c_int32] * 6),
    "vamp_upsample_trilinear_backward": (C.c_int, [C.c_int64] + [C.c_int32] * 6 + [_P, _P, _P, C.c_size_t, _P]),
    "vamp_conv3d_supported": (C.c_int, [_CD]),
    "vamp_conv3d_forward": (C.c_int, [_CD, _P, _P, _P, _P]),
    "vamp_conv3d_backward_data": (C.c_int, [_CD, _P, _P, _P, _P]),
    "vamp_conv3d_workspace_bytes": (C.c_size_t, [_CD]),
    "vamp_conv3d_backward_weight": (C.c_int, [_CD, _P, _P, _P, _P, C.c_size_t, _P]),
    "vamp_conv3d_bf16_supported": (C.c_int, [_CD]),
    "vamp_conv3d_bf16_forward": (C.c_int, [_CD, _P, _P, _P, _P]),
    "vamp_conv3d_bf16_backward_data": (C.c_int, [_CD, _P, _P, _P, _P]),
    "vamp_conv3d_bf16_workspace_bytes": (C.c_size_t, [_CD]),
    "vamp_conv3d_bf16_backward_weight": (C.c_int, [_CD, _P, _P, _P, _P, C.c_size_t, _P]),
    "vamp_conv3d_half_forward": (C.c_int, [_CD, C.c_int32, _P, _P, _P, _P]),
    "vamp_conv3d_half_backward_data": (C.c_int, [_CD, C.c_int32, _P, _P, _P, _P]),
    "vamp_conv3d_half_backward_weight": (C.c_int, [_CD, C.c_int32, _P, _P, _P, _P, C.c_size_t, _P]),
    "vamp_density_gate_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, _P, _P,
                                             _P]),
    "vamp_voxel_pooling_workspace_bytes": (C.c_size_t, [_P]),
    "vamp_voxel_pooling_forward": (C.c_int, [_P] * 4 + [_P, C.c_size_t, _P]),
    "vamp_voxel_pooling_backward": (C.c_int, [_P] * 5),
    "vamp_gate_conv1x1_supported": (C.c_int, [C.c_int32] * 3),
    "vamp_gate_conv1x1_workspace_bytes": (C.c_size_t, [C.c_int32] * 3),
    "vamp_gate_conv1x1_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32]
                                  + [_P] * 6),
    "vamp_gate_conv1x1_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32]
                                   + [_P] * 9 + [C.c_size_t, _P]),
}

_lib = None


class VampireHipError(RuntimeError):
    pass


def load():
    """Load the library and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must bring in ITS libamdhip64 first: a second HIP runtime (the system copy this
    # library would otherwise pull in) sees no device inside a torch process.
    import torch  # noqa: F401
    path = os.environ.get("VAMPIRE_HIP_LIB", lib_path())
    if not os.path.exists(path):
        raise VampireHipError(
            f"{path} not found: build it with `python -m vampire_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.vamp_abi_version()
    if got != ABI_VERSION:
        raise VampireHipError(f"ABI version mismatch: library {got}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().vamp_last_error().decode("utf-8", "replace")
        raise VampireHipError(f"{what} failed with code {code}: {msg}")


def profile_enable(on: bool):
    check(load().vamp_profile_enable(1 if on else 0), "vamp_profile_enable")


def profile_select(name=None):
    """Restrict the event timer to one kernel slot (by name); None = all slots."""
    lib = load()
    slot = -1
    if name is not None:
        for i in range(lib.vamp_profile_slots()):
            nm, n, ms = C.c_char_p(), C.c_int(), C.c_double()
            check(lib.vamp_profile_read(i, C.byref(nm), C.byref(n), C.byref(ms)), "vamp_profile_read")
            if nm.value.decode() == name:
                slot = i
        if slot < 0:
            raise KeyError(name)
    check(lib.vamp_profile_select(slot), "vamp_profile_select")


def profile_read():
    """{kernel name: (launches, total_ms)} since the last profile_enable(True)."""
    lib = load()
    out = {}
    for slot in range(lib.vamp_profile_slots()):
        name, n, ms = C.c_char_p(), C.c_int(), C.c_double()
        check(lib.vamp_profile_read(slot, C.byref(name), C.byref(n), C.byref(ms)), "vamp_profile_read")
        if n.value:
            out[name.value.decode()] = (n.value, ms.value)
    return out
