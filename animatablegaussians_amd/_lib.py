"""ctypes binding of ``libag_hip.so`` (the C ABI declared in ``include/ag_raster.h``).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved this module raises, and every
operator that needs it raises with it.  ``torch`` is imported first on purpose: the library links against
``libamdhip64.so.7`` and must bind to the HIP runtime instance torch already loaded, so that torch's streams and
device pointers are valid inside our launches.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime the library must share)

_HERE = os.path.dirname(os.path.abspath(__file__))
# AG_LIB_PATH selects another build of the same library (same-box A/B measurements of kernel variants); default: in-tree
LIB_PATH = os.environ.get("AG_LIB_PATH") or os.path.join(_HERE, "lib", "libag_hip.so")

c_i32 = ctypes.c_int32
c_f = ctypes.c_float
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t


class AgRasterForwardArgs(ctypes.Structure):
    _fields_ = [
        ("P", c_i32), ("W", c_i32), ("H", c_i32),
        ("sh_degree", c_i32), ("sh_coeffs", c_i32), ("prefiltered", c_i32),
        ("tan_fovx", c_f), ("tan_fovy", c_f), ("scale_modifier", c_f),
        ("bg", c_vp), ("means3D", c_vp), ("colors_precomp", c_vp), ("shs", c_vp), ("opacities", c_vp),
        ("scales", c_vp), ("rotations", c_vp), ("cov3D_precomp", c_vp),
        ("viewmatrix", c_vp), ("projmatrix", c_vp), ("campos", c_vp),
        ("out_color", c_vp), ("out_depth", c_vp), ("out_alpha", c_vp), ("radii", c_vp),
        ("geom_buffer", c_vp), ("geom_bytes", c_sz),
        ("image_buffer", c_vp), ("image_bytes", c_sz),
        ("binning_buffer", c_vp), ("binning_bytes", c_sz),
    ]


class AgRasterBackwardArgs(ctypes.Structure):
    _fields_ = [
        ("P", c_i32), ("W", c_i32), ("H", c_i32),
        ("sh_degree", c_i32), ("sh_coeffs", c_i32), ("num_rendered", c_i32),
        ("tan_fovx", c_f), ("tan_fovy", c_f), ("scale_modifier", c_f),
        ("bg", c_vp), ("means3D", c_vp), ("radii", c_vp), ("colors_precomp", c_vp), ("shs", c_vp),
        ("scales", c_vp), ("rotations", c_vp), ("cov3D_precomp", c_vp),
        ("viewmatrix", c_vp), ("projmatrix", c_vp), ("campos", c_vp), ("alphas", c_vp),
        ("dL_dout_color", c_vp), ("dL_dout_depth", c_vp), ("dL_dout_alpha", c_vp),
        ("geom_buffer", c_vp), ("image_buffer", c_vp), ("binning_buffer", c_vp),
        ("dL_dmeans2D", c_vp), ("dL_dcolors", c_vp), ("dL_dopacity", c_vp), ("dL_dmeans3D", c_vp),
        ("dL_dcov3D", c_vp), ("dL_dsh", c_vp), ("dL_dscales", c_vp), ("dL_drotations", c_vp),
        ("accum_buffer", c_vp), ("accum_bytes", c_sz), ("accumulate", c_i32), ("reserved", c_i32),
    ]


class AgRasterScratchLayout(ctypes.Structure):
    _fields_ = [(n, c_sz) for n in (
        "geom_rec_off", "geom_rec_stride", "geom_cov3d_off", "geom_tiles_touched_off",
        "img_ranges_off", "img_n_contrib_off", "img_tile_count_off", "img_num_rendered_off",
        "bin_point_list_off", "bin_keys_off")]


class AgGatherArgs(ctypes.Structure):
    _fields_ = [("N", c_i32), ("S", c_i32)] + [(n, c_vp) for n in (
        "pix", "position_map", "other_map", "color_map", "xyz", "opacity_raw", "scaling_raw", "rotation_raw",
        "positions", "opacity", "scales", "rotations", "colors")]


class AgLbsArgs(ctypes.Structure):
    _fields_ = [("N", c_i32), ("J", c_i32)] + [(n, c_vp) for n in (
        "lbs", "jnt_mats", "positions", "rotations", "out_positions", "out_rotations", "sp_idx", "sp_w")] + [("K", c_i32), ("reserved", c_i32)]


class AgHandFuseArgs(ctypes.Structure):
    _fields_ = [("N", c_i32), ("reserved", c_i32)] + [(n, c_vp) for n in (
        "xyz", "left_box", "right_box", "centre", "hand_positions", "hand_opacity", "hand_scales", "hand_rotations",
        "positions", "opacity", "scales", "rotations")]


class AgConvDesc(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("kind", "Cin", "Cout", "H", "W", "k", "stride", "padding")] + [("weight_scale", c_f)]


class AgLayerArgs(ctypes.Structure):          # include/ag_layers.h
    _fields_ = ([(n, c_i32) for n in ("Cin", "Cout", "H", "W", "k", "resample", "modulated", "reserved")]
                + [(n, c_f) for n in ("scale", "slope", "act_scale", "reserved_f")]
                + [(n, c_vp) for n in ("x", "weight", "style", "noise", "noise_weight", "act_bias", "k_blur", "w_mod", "demod", "x_blur", "out",
                                       "scratch", "workspace")]
                + [("workspace_bytes", c_sz)]
                + [(n, c_vp) for n in ("g_out", "g_x", "g_weight", "g_style", "g_bias_noise")]
                + [("want_bias", c_i32), ("want_noise_weight", c_i32)])


AG_MAX_GROUPS = 16
_PTRS = c_vp * AG_MAX_GROUPS


class AgGroupedLayerArgs(ctypes.Structure):   # include/ag_layers.h
    _fields_ = ([(n, c_i32) for n in ("G", "Cin", "Cout", "H", "W", "k", "resample", "modulated")]
                + [(n, c_f) for n in ("scale", "slope", "act_scale", "reserved_f")]
                + [("x", c_vp), ("x_group_stride", ctypes.c_int64)]
                + [(n, _PTRS) for n in ("weight", "style", "noise", "noise_weight", "act_bias")]
                + [(n, c_vp) for n in ("k_blur", "w_mod", "demod", "x_blur", "out", "scratch", "workspace")]
                + [("workspace_bytes", c_sz)]
                + [(n, c_vp) for n in ("g_out", "g_x", "g_weight", "g_style", "g_bias_noise")]
                + [("want_bias", c_i32), ("want_noise_weight", c_i32), ("operand_maxima", c_vp), ("x_maxima", c_vp), ("out_maxima", c_vp)]
                + [("packed_weights", c_vp), ("weights_cached", c_i32), ("reserved_i", c_i32)])


class AgGroupedToRgbArgs(ctypes.Structure):   # include/ag_layers.h
    _fields_ = ([(n, c_i32) for n in ("G", "Cin", "Cout", "H", "W")] + [("scale", c_f), ("x", c_vp)]
                + [(n, _PTRS) for n in ("weight", "style", "bias")]
                + [(n, c_vp) for n in ("skip", "skip_taps", "w_mod", "out", "scratch", "workspace")]
                + [("workspace_bytes", c_sz)]
                + [(n, c_vp) for n in ("g_out", "g_x", "g_weight", "g_style", "g_skip")]
                + [("weights_cached", c_i32), ("reserved_i", c_i32)])


class AgGroupedCombArgs(ctypes.Structure):    # include/ag_layers.h
    _fields_ = ([(n, c_i32) for n in ("M", "N", "C1", "C2", "Cout", "H", "W")] + [("member_begin", c_i32 * (AG_MAX_GROUPS + 1))]
                + [(n, c_f) for n in ("scale", "slope", "act_scale", "reserved_f")]
                + [("x", c_vp), ("lev", c_vp), ("weight", _PTRS), ("act_bias", _PTRS)]
                + [(n, c_vp) for n in ("out", "scratch", "workspace")] + [("workspace_bytes", c_sz)]
                + [(n, c_vp) for n in ("g_out", "g_x", "g_lev", "g_weight_x", "g_weight_lev", "g_bias", "g_weight", "operand_maxima", "x_maxima", "out_maxima",
                                       "packed_x", "packed_lev")]
                + [("weights_cached", c_i32), ("reserved_i", c_i32)])


AG_ADAM_MAX_TENSORS = 48


class AgAdamArgs(ctypes.Structure):           # include/ag_optim.h
    _fields_ = ([("n", c_i32), ("maximize", c_i32)] + [(n, c_vp * AG_ADAM_MAX_TENSORS) for n in ("param", "grad", "exp_avg", "exp_avg_sq")]
                + [("numel", ctypes.c_int64 * AG_ADAM_MAX_TENSORS)]
                + [(n, c_f) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "one_minus_beta1", "one_minus_beta2")]
                + [(n, c_f * AG_ADAM_MAX_TENSORS) for n in ("bias_correction1", "bias_correction2_sqrt")])


AG_LINEAR_MAX_JOBS = 32
_LPTRS = c_vp * AG_LINEAR_MAX_JOBS


class AgEqualLinearArgs(ctypes.Structure):    # include/ag_linear.h
    _fields_ = ([(n, c_i32) for n in ("n_jobs", "B", "in_features", "act", "normalize_input", "reserved")]
                + [("x", _LPTRS), ("weight", _LPTRS), ("bias", _LPTRS), ("out_features", c_i32 * AG_LINEAR_MAX_JOBS),
                   ("alpha", c_f * AG_LINEAR_MAX_JOBS), ("bias_mul", c_f * AG_LINEAR_MAX_JOBS), ("y", _LPTRS), ("g_y", _LPTRS),
                   ("g_x", _LPTRS), ("g_weight", _LPTRS), ("g_bias", _LPTRS), ("scratch", c_vp)])


class AgSmplxModel(ctypes.Structure):
    _fields_ = [("V", c_i32), ("J", c_i32), ("NB", c_i32), ("reserved", c_i32)] + [(n, c_vp) for n in (
        "v_template", "shapedirs", "posedirs", "J_regressor", "parents", "lbs_weights", "joint_template", "joint_dirs")]


# every symbol include/*.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("ag_abi_version", ctypes.c_int, []),
    ("ag_last_error", ctypes.c_char_p, []),
    ("ag_raster_geom_bytes", c_sz, [c_i32]),
    ("ag_raster_image_bytes", c_sz, [c_i32, c_i32]),
    ("ag_raster_binning_bytes", c_sz, [c_i32]),
    ("ag_raster_accum_bytes", c_sz, [c_i32]),
    ("ag_raster_describe_scratch", ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(AgRasterScratchLayout)]),
    ("ag_raster_forward_plan", ctypes.c_int, [ctypes.POINTER(AgRasterForwardArgs), c_vp, ctypes.POINTER(c_i32)]),
    ("ag_raster_forward_render", ctypes.c_int, [ctypes.POINTER(AgRasterForwardArgs), c_i32, c_vp]),
    ("ag_raster_forward_optimistic", ctypes.c_int, [ctypes.POINTER(AgRasterForwardArgs), c_i32, c_vp, ctypes.POINTER(c_i32)]),
    ("ag_raster_backward", ctypes.c_int, [ctypes.POINTER(AgRasterBackwardArgs), c_vp]),
    ("ag_raster_forward_backward", ctypes.c_int, [ctypes.POINTER(AgRasterForwardArgs), ctypes.POINTER(AgRasterBackwardArgs), c_i32, c_vp,
                                                  ctypes.POINTER(c_i32)]),
    ("ag_raster_forward_backward_enqueue", ctypes.c_int, [ctypes.POINTER(AgRasterForwardArgs), ctypes.POINTER(AgRasterBackwardArgs), c_i32, c_vp,
                                                          ctypes.POINTER(c_i32)]),
    ("ag_raster_collect", ctypes.c_int, [c_i32, ctypes.POINTER(c_i32)]),
    ("ag_raster_large_tile_sort", ctypes.c_int, [c_i32]),
    ("ag_raster_mark_visible", ctypes.c_int, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    ("ag_prof_kernel_name", ctypes.c_char_p, [c_i32]),
    ("ag_prof_enable", ctypes.c_int, [ctypes.c_uint32]),
    ("ag_prof_collect", ctypes.c_int, [ctypes.POINTER(c_i32), ctypes.POINTER(c_f), ctypes.POINTER(ctypes.c_double)]),
    ("ag_prof_collect_to", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(c_i32), ctypes.POINTER(c_f), ctypes.POINTER(ctypes.c_double)]),
    ("ag_debug_atomic_rate", ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    ("ag_noise_bias_act_forward", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                               ctypes.c_float, c_vp]),
    ("ag_noise_bias_act_partial_floats", c_sz, [c_i32, c_i32]),
    ("ag_noise_bias_act_backward", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_float, ctypes.c_float, c_vp]),
    ("ag_modulate_weight_forward", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_float, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp]),
    ("ag_modulate_weight_partial_floats", c_sz, [c_i32, c_i32]),
    ("ag_modulate_weight_backward", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_float, ctypes.c_int32,
                                                 ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp]),
    ("ag_block2x2_transform", ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp]),
    ("ag_skip_chain_forward", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    ("ag_skip_chain_backward", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    # include/ag_lpips.h
    ("ag_maxpool2x2_forward", ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp]),
    ("ag_maxpool2x2_backward", ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp]),
    ("ag_lpips_level_forward", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, c_vp]),
    ("ag_lpips_level_backward", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, c_vp]),
    ("ag_conv_set_math", ctypes.c_int, [ctypes.c_int]),
    ("ag_conv_get_math", ctypes.c_int, []),
    ("ag_conv_status", ctypes.c_int, [ctypes.c_int]),
    ("ag_adam_args_bytes", c_sz, []),
    ("ag_adam_step", ctypes.c_int, [ctypes.POINTER(AgAdamArgs), c_vp]),
    # include/ag_linear.h
    ("ag_equal_linear_args_bytes", c_sz, []),
    ("ag_equal_linear_scratch_floats", c_sz, [ctypes.POINTER(AgEqualLinearArgs)]),
    ("ag_equal_linear_forward", ctypes.c_int, [ctypes.POINTER(AgEqualLinearArgs), c_vp]),
    ("ag_equal_linear_backward", ctypes.c_int, [ctypes.POINTER(AgEqualLinearArgs), c_vp]),
    ("ag_bilinear_resize_forward", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    ("ag_bilinear_resize_backward", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    ("ag_select_add_rows", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    ("ag_plane_sums_scratch_floats", c_sz, [c_i32, ctypes.c_int64]),
    ("ag_plane_sums", ctypes.c_int, [c_vp, c_vp, c_i32, ctypes.c_int64, c_vp, c_vp]),
    ("ag_debug_mfma_rate", ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_vp, c_vp]),
    ("ag_debug_mfma_rate_bf16", ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_vp, c_vp]),
    # include/ag_smplx.h
    ("ag_smplx_workspace_floats", c_sz, [ctypes.POINTER(AgSmplxModel), c_i32]),
    ("ag_smplx_forward", ctypes.c_int, [ctypes.POINTER(AgSmplxModel), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    ("ag_smplx_prepare", ctypes.c_int, [ctypes.POINTER(AgSmplxModel), c_vp, c_vp, c_vp]),
    ("ag_smplx_shape", ctypes.c_int, [ctypes.POINTER(AgSmplxModel), c_i32, c_vp, c_vp, c_vp]),
    ("ag_mat4_mul_inverse", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    ("ag_smplx_keypoints", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    # include/ag_avatar.h
    ("ag_gather_activate_forward", ctypes.c_int, [ctypes.POINTER(AgGatherArgs), c_vp]),
    ("ag_gather_activate_backward", ctypes.c_int, [ctypes.POINTER(AgGatherArgs), c_vp, c_vp, c_vp, c_vp]),
    ("ag_lbs_forward", ctypes.c_int, [ctypes.POINTER(AgLbsArgs), c_vp]),
    ("ag_lbs_backward", ctypes.c_int, [ctypes.POINTER(AgLbsArgs), c_vp, c_vp, c_vp]),
    ("ag_hand_fuse", ctypes.c_int, [ctypes.POINTER(AgHandFuseArgs), c_vp]),
    # include/ag_styleunet.h
    ("ag_fused_bias_act", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f, c_f, ctypes.c_int64, ctypes.c_int64, c_i32, c_vp]),
    ("ag_upfirdn2d", ctypes.c_int, [c_vp, c_vp, c_vp] + [c_i32] * 13 + [c_vp]),
    # include/ag_layers.h
    ("ag_layer_args_bytes", c_sz, []),
    ("ag_layer_output_size", ctypes.c_int, [ctypes.POINTER(AgLayerArgs), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    ("ag_layer_scratch_floats", c_sz, [ctypes.POINTER(AgLayerArgs), c_i32]),
    ("ag_layer_forward", ctypes.c_int, [ctypes.POINTER(AgLayerArgs), c_vp]),
    ("ag_layer_backward", ctypes.c_int, [ctypes.POINTER(AgLayerArgs), c_vp]),
    ("ag_grouped_layer_args_bytes", c_sz, []),
    ("ag_grouped_layer_output_size", ctypes.c_int, [ctypes.POINTER(AgGroupedLayerArgs), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    ("ag_grouped_layer_scratch_floats", c_sz, [ctypes.POINTER(AgGroupedLayerArgs), c_i32]),
    ("ag_grouped_layer_workspace_bytes", c_sz, [ctypes.POINTER(AgGroupedLayerArgs)]),
    ("ag_grouped_layer_maxima_floats", c_sz, []),
    ("ag_grouped_layer_packed_bytes", c_sz, [ctypes.POINTER(AgGroupedLayerArgs)]),
    ("ag_grouped_comb_maxima_floats", c_sz, []),
    ("ag_grouped_comb_packed_bytes", c_sz, [ctypes.POINTER(AgGroupedCombArgs), c_i32]),
    ("ag_grouped_layer_forward", ctypes.c_int, [ctypes.POINTER(AgGroupedLayerArgs), c_vp]),
    ("ag_grouped_layer_backward", ctypes.c_int, [ctypes.POINTER(AgGroupedLayerArgs), c_vp]),
    ("ag_grouped_to_rgb_args_bytes", c_sz, []),
    ("ag_grouped_to_rgb_scratch_floats", c_sz, [ctypes.POINTER(AgGroupedToRgbArgs), c_i32]),
    ("ag_grouped_to_rgb_workspace_bytes", c_sz, [ctypes.POINTER(AgGroupedToRgbArgs)]),
    ("ag_grouped_to_rgb_forward", ctypes.c_int, [ctypes.POINTER(AgGroupedToRgbArgs), c_vp]),
    ("ag_grouped_to_rgb_backward", ctypes.c_int, [ctypes.POINTER(AgGroupedToRgbArgs), c_vp]),
    ("ag_grouped_block2x2", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    ("ag_grouped_comb_args_bytes", c_sz, []),
    ("ag_grouped_comb_scratch_floats", c_sz, [ctypes.POINTER(AgGroupedCombArgs), c_i32]),
    ("ag_grouped_comb_workspace_bytes", c_sz, [ctypes.POINTER(AgGroupedCombArgs)]),
    ("ag_grouped_comb_forward", ctypes.c_int, [ctypes.POINTER(AgGroupedCombArgs), c_vp]),
    ("ag_grouped_comb_backward", ctypes.c_int, [ctypes.POINTER(AgGroupedCombArgs), c_vp]),
    # include/ag_conv.h
    ("ag_conv_output_size", ctypes.c_int, [ctypes.POINTER(AgConvDesc), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    ("ag_conv_workspace_bytes", c_sz, [ctypes.POINTER(AgConvDesc)]),
    ("ag_conv_forward", ctypes.c_int, [ctypes.POINTER(AgConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    ("ag_conv_backward_input", ctypes.c_int, [ctypes.POINTER(AgConvDesc), c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    ("ag_conv_backward_weight", ctypes.c_int, [ctypes.POINTER(AgConvDesc), c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
]

AG_ERR_SCRATCH_TOO_SMALL = -2      # include/ag_raster.h

_lib = None


class AgNativeError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the library; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AgNativeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(animatablegaussians_amd/csrc/build.sh).  There is no CPU or PyTorch fallback for this path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)   # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        if L.ag_abi_version() != 1:
            raise AgNativeError(f"libag_hip.so ABI version {L.ag_abi_version()} != 1")
        _lib = L
    return _lib


AG_K_COUNT = 9
AG_K_BLEND_BACKWARD, AG_K_GATHER_CONV, AG_K_WGRAD = 5, 7, 8      # include/ag_raster.h AgKernelId


def prof_enable(kernel_ids) -> None:
    mask = 0
    for k in kernel_ids:
        mask |= 1 << int(k)
    check(lib().ag_prof_enable(mask), "ag_prof_enable")


def prof_collect():
    """{kernel name: (launches, total ms)} for the launches bracketed since the last collect."""
    n, ms, work = prof_collect_work()
    return {k: (n[k], ms[k]) for k in n}


def prof_collect_work(records_path=None):
    """({name: launches}, {name: total ms}, {name: declared work (FLOPs for the convolution kernels)}); ``records_path``: also one CSV line
    per bracketed launch (kernel,tag,work,ms) into that file."""
    n = (c_i32 * AG_K_COUNT)()
    ms = (c_f * AG_K_COUNT)()
    wk = (ctypes.c_double * AG_K_COUNT)()
    if records_path is not None:
        check(lib().ag_prof_collect_to(str(records_path).encode(), n, ms, wk), "ag_prof_collect_to")
    else:
        check(lib().ag_prof_collect(n, ms, wk), "ag_prof_collect")
    names = [lib().ag_prof_kernel_name(i).decode() for i in range(AG_K_COUNT)]
    return ({k: int(n[i]) for i, k in enumerate(names)}, {k: float(ms[i]) for i, k in enumerate(names)},
            {k: float(wk[i]) for i, k in enumerate(names)})


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(dev):
    """Context for a native call on ``dev``: ``torch.cuda.device(dev)`` only when ``dev`` is not already the current device (one
    process per GPU: it always is; the unconditional device switch cost ~3 us in each of the ~1300 native calls of a training step)."""
    import torch
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(dev)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ag_last_error().decode("utf-8", "replace")
        raise AgNativeError(f"{what} failed (code {rc}): {msg}")
