"""ctypes binding of libpixie_hip.so (declared in include/pixie_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, the caller gets an
exception.  Torch is used only for device memory (tensor.data_ptr()) and the current stream.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpixie_hip.so")
DIAG_LIB_PATH = os.path.join(_HERE, "libpixie_hip_diag.so")   # the -DPIXIE_DIAG build of the same sources (tests, profilers)


class PixieHipError(RuntimeError):
    pass


class BCDesc(C.Structure):
    """struct pixie_bc_desc"""
    _fields_ = [("type", C.c_int32), ("surface_type", C.c_int32), ("reset", C.c_int32), ("pad_", C.c_int32),
                ("point", C.c_double * 3), ("size", C.c_double * 3), ("velocity", C.c_double * 3),
                ("normal", C.c_double * 3), ("start_time", C.c_double), ("end_time", C.c_double),
                ("friction", C.c_double)]


class PModDesc(C.Structure):
    """struct pixie_pmod_desc"""
    _fields_ = [("type", C.c_int32), ("pad_", C.c_int32),
                ("point", C.c_double * 3), ("size", C.c_double * 3), ("force", C.c_double * 3),
                ("velocity", C.c_double * 3), ("normal", C.c_double * 3), ("h1", C.c_double * 3),
                ("h2", C.c_double * 3), ("half_height", C.c_double), ("radius", C.c_double),
                ("rotation_scale", C.c_double), ("translation_scale", C.c_double),
                ("start_time", C.c_double), ("end_time", C.c_double)]


class ConvDesc(C.Structure):
    """struct pixie_conv_desc"""
    _fields_ = [("d_in0", C.c_void_p), ("c0", C.c_int32),
                ("d_in1", C.c_void_p), ("c1", C.c_int32),
                ("in_d", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("upsample", C.c_int32), ("stride", C.c_int32), ("ksize", C.c_int32),
                ("d_pro_a", C.c_void_p), ("d_pro_b", C.c_void_p),
                ("d_gamma", C.c_void_p), ("d_beta", C.c_void_p),
                ("act", C.c_int32),
                ("d_w", C.c_void_p), ("d_bias", C.c_void_p),
                ("c_out", C.c_int32),
                ("d_residual", C.c_void_p), ("d_out", C.c_void_p),
                ("d_w16", C.c_void_p), ("d_in_amax0", C.c_void_p), ("d_in_amax1", C.c_void_p),
                ("in_bound", C.c_float),
                ("d_out_stats", C.c_void_p), ("d_out_amax", C.c_void_p),
                ("d_workspace", C.c_void_p),
                ("out_d", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("d_skip_in0", C.c_void_p), ("skip_c0", C.c_int32),
                ("d_skip_in1", C.c_void_p), ("skip_c1", C.c_int32),
                ("d_skip_w16", C.c_void_p), ("d_skip_bias", C.c_void_p),
                ("d_skip_amax0", C.c_void_p), ("d_skip_amax1", C.c_void_p)]


class UNetConfigC(C.Structure):
    """struct pixie_unet_config"""
    _fields_ = [("feature_channels", C.c_int32), ("cond_dim", C.c_int32), ("model_channels", C.c_int32), ("num_res_blocks", C.c_int32),
                ("n_channel_mult", C.c_int32), ("channel_mult", C.c_int32 * 8),
                ("n_attention_resolutions", C.c_int32), ("attention_resolutions", C.c_int32 * 8),
                ("grid_size", C.c_int32), ("out_channels", C.c_int32), ("precision", C.c_int32)]


class FieldDesc(C.Structure):
    """struct pixie_field_desc"""
    _fields_ = [("d_pred", C.c_void_p), ("d_mask", C.c_void_p),
                ("d_axis_x", C.c_void_p), ("d_axis_y", C.c_void_p), ("d_axis_z", C.c_void_p),
                ("n_classes", C.c_int32), ("d", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("min_spacing", C.c_double),
                ("density_min", C.c_double), ("density_max", C.c_double), ("E_min", C.c_double), ("E_max", C.c_double),
                ("nu_min", C.c_double), ("nu_max", C.c_double)]


# every symbol include/pixie_hip.h declares: name -> (restype, argtypes)
_VP, _I, _I64, _D, _S = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_char_p
_D3 = C.POINTER(C.c_double)
SIGNATURES = {
    "pixie_last_error": (C.c_char_p, []),
    "pixie_build_arch": (C.c_char_p, []),
    "pixie_mpm_create": (_I, [C.POINTER(_VP), _I, _I, _D]),
    "pixie_mpm_destroy": (_I, [_VP]),
    "pixie_mpm_regrid": (_I, [_VP, _I, _D, _VP]),
    "pixie_mpm_set_field": (_I, [_VP, _S, _VP, _I64, _VP]),
    "pixie_mpm_get_field": (_I, [_VP, _S, _VP, _I64, _VP]),
    "pixie_mpm_fill_field": (_I, [_VP, _S, _D, _VP]),
    "pixie_mpm_set_scalar": (_I, [_VP, _S, _D]),
    "pixie_mpm_get_scalar": (_I, [_VP, _S, C.POINTER(_D)]),
    "pixie_mpm_update_mass": (_I, [_VP, _VP]),
    "pixie_mpm_finalize_mu_lam": (_I, [_VP, _I, _VP]),
    "pixie_mpm_apply_additional_params": (_I, [_VP, _D3, _D3, _D, _D, _D, _I, _VP]),
    "pixie_mpm_apply_additional_params_batch": (_I, [_VP, _I64, _VP, _VP, _VP, _VP]),
    "pixie_mpm_add_bc": (_I, [_VP, C.POINTER(BCDesc)]),
    "pixie_mpm_add_particle_modifier": (_I, [_VP, C.POINTER(PModDesc), _VP]),
    "pixie_mpm_step": (_I, [_VP, _D, _I, _VP]),
    "pixie_mpm_export_cov": (_I, [_VP, _VP, _VP]),
    "pixie_mpm_export_R": (_I, [_VP, _VP, _VP]),
    "pixie_mpm_export_frame": (_I, [_VP, _I, _D3, _D, _D3, C.POINTER(C.c_double), _VP, _VP, _VP]),
    "pixie_mpm_out_of_bounds": (_I, [_VP, C.POINTER(_I64), _VP]),
    "pixie_pack_fields": (_I, [_VP, _VP, _I64, _I64, _VP, _I64, _VP]),
    "pixie_conv_cout_padded": (_I, [_I]),
    "pixie_conv_pack_weights": (_I, [_VP, _VP, _I, _I, _I, _VP]),
    "pixie_conv_packed16_bytes": (_I64, [_I, _I, _I]),
    "pixie_conv_pack_weights_f16x2": (_I, [_VP, _VP, _I, _I, _I, _VP]),
    "pixie_conv3d_forward": (_I, [C.POINTER(ConvDesc), _VP]),
    "pixie_conv_stats_floats": (_I64, [C.POINTER(ConvDesc)]),
    "pixie_conv_workspace_bytes": (_I64, [C.POINTER(ConvDesc)]),
    "pixie_stats_finalize": (_I, [_VP, C.POINTER(ConvDesc), _VP, _VP]),
    "pixie_channel_stats": (_I, [_VP, _I, _I64, _VP, _VP, _VP]),
    "pixie_tensor_amax": (_I, [_VP, _I64, _VP, _VP]),
    "pixie_channel_sums": (_I, [_VP, _I, _I64, _VP, _VP]),
    "pixie_norm_finalize": (_I, [_VP, _I, _I64, _I, _I, _D, _VP, _VP, _VP, _VP, _VP]),
    "pixie_attention_forward": (_I, [_VP, _VP, _I, _I, _VP]),
    "pixie_channel_affine": (_I, [_VP, _VP, _VP, _VP, _I, _I64, _VP]),
    "pixie_projector_conv0": (_I, [_VP, _I64, _I, _I, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), _I, _VP]),
    "pixie_conv_skip_foldable": (_I, [C.POINTER(ConvDesc)]),
    "pixie_unet_create": (_I, [C.POINTER(_VP), C.POINTER(UNetConfigC)]),
    "pixie_unet_destroy": (_I, [_VP]),
    "pixie_unet_param_count": (_I, [_VP]),
    "pixie_unet_param_info": (_I, [_VP, _I, C.POINTER(_S), C.POINTER(_I64), C.POINTER(C.c_int32), C.POINTER(_I64)]),
    "pixie_unet_set_param": (_I, [_VP, _S, _VP, _I64]),
    "pixie_unet_workspace_bytes": (_I64, [_VP, _I, _I, _I]),
    "pixie_unet_set_option": (_I, [_VP, _S, _I]),
    "pixie_unet_forward": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _I64, _VP]),
    "pixie_combine_class_ids": (_I, [_VP, _I, _VP, _I64, _VP, _VP]),
    "pixie_combine_predictions": (_I, [_VP, _I, _VP, _I64, _VP, _VP, _VP]),
    "pixie_voxel_grid_to_ncdhw": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "pixie_unscale_prediction": (_I, [_VP, _I, _I64, _D, _D, _D, _D, _D, _D, _VP, _VP]),
    "pixie_field_points_scratch_bytes": (_I64, [C.POINTER(FieldDesc)]),
    "pixie_field_points": (_I, [C.POINTER(FieldDesc), _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pixie_fill_densify": (_I, [_VP, _VP, _VP, _I, _I, _D, _VP, _VP, _VP]),
    "pixie_fill_dense_cells": (_I, [_VP, _VP, _I, _D, _D, _I, _VP, _I64, _VP, C.c_uint32, _VP]),
    "pixie_fill_internal_cells": (_I, [_VP, _VP, _I, _D, _I, _I, _I, _D, _VP, _I64, _VP, C.c_uint32, _VP]),
    "pixie_particle_volume": (_I, [_VP, _I, _I, _D, _VP, _VP, _VP]),
    "pixie_nearest_particle": (_I, [_VP, _I, _VP, _I, _VP, _VP]),
    "pixie_dbscan_roots": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, C.POINTER(_D), _D, _D, _I, _VP, _VP, _VP, _VP]),
    "pixie_field_to_particles": (_I, [C.POINTER(FieldDesc), _VP, _I, _I, _D, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
}

# entry points that exist only in the PIXIE_DIAG build (include/pixie_hip.h, last section)
DIAG_SIGNATURES = {
    "pixie_mpm_phase": (_I, [_VP, _I, _D, _VP]),
    "pixie_mpm_kernel_times": (_I, [_VP, C.POINTER(_D), C.POINTER(_D), C.POINTER(_I64)]),
    "pixie_conv_kernel_variant": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int)]),
}

_libs = {}


def load(diag: bool = False):
    """dlopen libpixie_hip.so -- or, diag=True, libpixie_hip_diag.so, the same sources built with -DPIXIE_DIAG, which adds the
    diagnostic entry points -- and type every entry point; raises if absent.  The two are independent libraries: a handle
    belongs to the one that created it."""
    if diag in _libs:
        return _libs[diag]
    path = DIAG_LIB_PATH if diag else LIB_PATH
    if not os.path.exists(path):
        raise PixieHipError(
            f"{path} not found: build it with `python -m pixie_amd.build` (hipcc, gfx950). "
            "pixie_amd has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  The process must have ONE HIP runtime -- streams and
    # device pointers cross the C ABI in both directions -- so torch's copy has to be in the process before ours is resolved
    # (same SONAME: the loader then binds libpixie_hip.so to it).  Loaded the other way round, the system ROCm runtime and
    # torch's coexist and the first hipMalloc / launch on a torch stream fails.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in list(SIGNATURES.items()) + (list(DIAG_SIGNATURES.items()) if diag else []):
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _libs[diag] = lib
    return lib


def check(rc: int, what: str = "", lib=None):
    if rc != 0:
        msg = (lib or load()).pixie_last_error()
        raise PixieHipError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def current_stream_ptr():
    """hipStream_t of torch's current stream as a void pointer (value 0 / None = the default stream).  The raw getter is ~5x
    cheaper than building a torch.cuda.Stream object, which matters for callers that ask once per substep (the deferred
    p2g2p of the MPM shim: 1000 calls per frame)."""
    import torch
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    except AttributeError:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])
