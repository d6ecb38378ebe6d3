"""ctypes binding of libhaet_hip.so (the C ABI declared in include/haet_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails,
an exception is raised.  (The CPU oracle under oracle/ is test infrastructure and is
never imported from here.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_longlong, POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhaet_hip.so")

MAX_LEVELS = 4
MAX_DEPTH = 4


class HaetError(RuntimeError):
    pass


class ErwinBlockW(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "norm1", "norm2", "pe_w", "pe_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
        "w1_w", "w1_b", "w2_w", "w2_b", "w3_w", "w3_b", "small_off", "small_len")]


class ErwinLayer(ctypes.Structure):
    _fields_ = [("c", c_int32), ("heads", c_int32), ("ball", c_int32), ("depth", c_int32),
                ("hidden", c_int32), ("blocks", ErwinBlockW * MAX_DEPTH)]


class ErwinResample(ctypes.Structure):
    _fields_ = [("stride", c_int32), ("w", c_int32), ("b", c_int32),
                ("bn_scale", c_int32), ("bn_shift", c_int32), ("small_off", c_int32), ("small_len", c_int32),
                ("bn_eps", c_float), ("bn_momentum", c_float)]


class ErwinDesc(ctypes.Structure):
    _fields_ = [("n_levels", c_int32), ("d", c_int32), ("rotate", c_int32), ("decode", c_int32),
                ("bn_train", c_int32), ("enc", ErwinLayer * MAX_LEVELS), ("dec", ErwinLayer * MAX_LEVELS),
                ("pool", ErwinResample * MAX_LEVELS), ("unpool", ErwinResample * MAX_LEVELS)]


class ErwinBlockF32(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "norm1", "norm2", "pe_w", "pe_b", "qkv_w", "qkv_wT", "qkv_b", "proj_w", "proj_wT", "proj_b",
        "w1_w", "w1_wT", "w1_b", "w2_w", "w2_wT", "w2_b", "w3_w", "w3_wT", "w3_b")]


class ErwinResampleF32(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("w", "wT", "b", "gamma", "beta", "rmean", "rvar")]


class ErwinF32Layout(ctypes.Structure):
    """haet_erwin_f32_layout (include/haet_hip.h): float offsets of the fp32 weight blob of haet_erwin_bwd."""
    _fields_ = [("total", c_int32), ("pad", c_int32 * 3),
                ("enc", (ErwinBlockF32 * MAX_DEPTH) * MAX_LEVELS), ("dec", (ErwinBlockF32 * MAX_DEPTH) * MAX_LEVELS),
                ("pool", ErwinResampleF32 * MAX_LEVELS), ("unpool", ErwinResampleF32 * MAX_LEVELS)]


class ErwinTrain(ctypes.Structure):
    """haet_erwin_train (include/haet_hip.h): device pointers of the train-mode BatchNorm state."""
    _fields_ = [("running_mean", c_void_p * (2 * MAX_LEVELS)), ("running_var", c_void_p * (2 * MAX_LEVELS)),
                ("num_batches_tracked", c_void_p * (2 * MAX_LEVELS)), ("batch_stats", c_void_p),
                ("workspace", c_void_p), ("tape", c_void_p), ("tape_op_floats", c_int32)]


# name -> (restype, argtypes); every symbol include/haet_hip.h declares
_P = c_void_p
SIGNATURES = {
    "haet_abi_version": (c_int, []),
    "haet_last_error_string": (c_char_p, []),
    "haet_split_bf16": (c_int, [_P, c_size_t, _P, _P, _P]),
    "haet_split_bf16_tiled": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "haet_linear_supported": (c_int, [c_int, c_int]),
    "haet_linear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "haet_slice_num_parts": (c_int, [c_int] * 5),
    "haet_slice_part_floats": (c_size_t, [c_int] * 4),
    "haet_slice_fwd": (c_int, [_P, _P, _P, _P, c_float, c_float, c_float] + [c_int] * 5 + [_P, _P]),
    "haet_slice_merge": (c_int, [_P] + [c_int] * 6 + [_P, _P, _P, _P, _P]),
    "haet_token_pos": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "haet_balltree_build": (c_int, [_P, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int),
                                    c_float, _P, _P, _P, _P]),
    "haet_tokens_tree_build": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int),
                                       c_float, _P, _P, _P, _P]),
    "haet_project_split_supported": (c_int, [c_int, c_int, c_int]),
    "haet_project_split_fwd": (c_int, [_P, _P, _P, _P, _P, c_float, c_float, c_int, c_int, c_int, _P, _P, _P]),
    "haet_slice_split_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "haet_slice_mfma_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "haet_slice_fwd_split": (c_int, [_P, _P, _P, _P, c_float, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "haet_deslice_fwd_split": (c_int, [_P, _P, _P, _P, c_float, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                       _P, _P]),
    "haet_conv_project_supported": (c_int, [c_int, c_int]),
    "haet_conv_cells_supported": (c_int, [c_int, c_int]),
    "haet_conv_cells_bytes": (c_size_t, [c_longlong, c_int]),
    "haet_conv_presplit": (c_int, [_P, c_longlong, c_int, _P, _P, _P, c_float, c_int, _P]),
    "haet_conv_project_cells_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P, c_int, _P]),
    "haet_conv_project_cells_split_fwd": (c_int, [_P, _P, _P, _P, _P, c_float, c_float] + [c_int] * 9 + [_P, _P, c_int, _P]),
    "haet_conv_project_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      _P, _P]),
    "haet_conv_project_det_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      _P, _P]),
    "haet_conv_project_split_fwd": (c_int, [_P, _P, _P, _P, _P, c_float, c_float] + [c_int] * 9 + [_P, _P, _P]),
    "haet_conv_project_f16_fwd": (c_int, [_P, _P, _P] + [c_int] * 9 + [_P, _P]),
    "haet_conv_project_split_f16_fwd": (c_int, [_P, _P, _P, _P, c_float, c_float] + [c_int] * 9 + [_P, _P, _P]),
    "haet_slice_bwd_num_parts": (c_int, [c_int] * 5),
    "haet_slice_bwd_param_floats": (c_size_t, [c_int, c_int]),
    "haet_deslice_bwd_tokens": (c_int, [_P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P] + [c_int] * 5 + [_P, _P]),
    "haet_slice_bwd_points": (c_int, [_P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P]
                              + [c_int] * 5 + [_P, _P, _P]),
    "haet_linear_wgrad_supported": (c_int, [c_int, c_int]),
    "haet_linear_wgrad_num_parts": (c_int, [c_int]),
    "haet_linear_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "haet_conv_wgrad_tap": (c_int, [_P, _P] + [c_int] * 9 + [_P, _P]),
    "haet_linear_act_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "haet_linear_ln_fwd": (c_int, [_P, _P, _P, c_float, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "haet_project_split_ln_fwd": (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_float, c_float, c_int, c_int, c_int,
                                          _P, _P, _P]),
    "haet_linear_swiglu_supported": (c_int, [c_int, c_int, c_int]),
    "haet_linear_swiglu_fwd": (c_int, [_P, _P, c_float, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "haet_layernorm_supported": (c_int, [c_int]),
    "haet_layernorm_fwd": (c_int, [_P, _P, _P, c_float, c_int, c_int, _P, _P]),
    "haet_add_layernorm_fwd": (c_int, [_P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P]),
    "haet_ln_linear_small_supported": (c_int, [c_int, c_int]),
    "haet_ln_linear_small_fwd": (c_int, [_P, _P, _P, _P, c_float, _P, _P, c_int, c_int, c_int, _P, _P]),
    "haet_layernorm_bwd_num_parts": (c_int, [c_int]),
    "haet_layernorm_bwd": (c_int, [_P, _P, _P, c_float, c_int, c_int, _P, _P, _P]),
    "haet_linear_narrow_supported": (c_int, [c_int, c_int]),
    "haet_linear_narrow_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "haet_erwin_lds_bytes": (c_size_t, [POINTER(ErwinDesc), c_int]),
    "haet_erwin_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, POINTER(ErwinDesc), POINTER(ErwinTrain), c_int, c_int, _P, _P]),
    "haet_deslice_out_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "haet_deslice_out_fwd_split": (c_int, [_P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P] + [c_int] * 6 + [_P, _P]),
    "haet_erwin_tape_floats": (c_size_t, [POINTER(ErwinDesc), c_int, POINTER(c_int), POINTER(c_int)]),
    "haet_erwin_bwd_supported": (c_int, [POINTER(ErwinDesc), c_int, c_int]),
    "haet_erwin_bwd": (c_int, [_P] * 8 + [POINTER(ErwinDesc), _P, _P, c_int, c_int, _P, _P, _P]),
    "haet_erwin_train_workspace_bytes": (c_size_t, [POINTER(ErwinDesc), c_int]),
    "haet_set_deterministic": (None, [c_int]),
    "haet_get_deterministic": (c_int, []),
    "haet_resident_workgroup_limit": (c_int, []),
    "haet_erwin_barrier_status": (c_int, [c_void_p]),
    "haet_token_stage_supported": (c_int, [POINTER(ErwinDesc), c_int, c_int]),
    "haet_token_stage_fwd": (c_int, [_P, c_int, c_int, c_float, _P, c_size_t, POINTER(ErwinDesc), POINTER(ErwinTrain),
                                     _P, _P, _P, _P, _P, _P]),
    "haet_erwin_frag_bytes": (c_size_t, [POINTER(ErwinDesc), c_int]),
    "haet_erwin_frag_pack": (c_int, [_P, c_size_t, POINTER(ErwinDesc), c_int, _P, c_size_t, _P]),
    "haet_token_stage_fwd_frag": (c_int, [_P, c_int, c_int, c_float, _P, c_size_t, _P, c_size_t, POINTER(ErwinDesc),
                                          _P, _P, _P, _P, _P, _P]),
    "haet_deslice_fwd": (c_int, [_P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P]
                         + [c_int] * 5 + [_P, _P]),
    # stand-alone ErwinFlash on raw point clouds (include/haet_hip.h, last section)
    "haet_balltree_big_slots": (c_size_t, [_P, c_int]),
    "haet_balltree_big_workspace_bytes": (c_size_t, [_P, c_int, c_int]),
    "haet_balltree_big_build": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "haet_pc_radius_neighbours": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_int, _P, _P, _P]),
    "haet_pc_radius_cell_keys": (c_int, [_P, _P, c_int, c_int, _P, c_float, _P, _P, _P]),
    "haet_pc_radius_neighbours_grid": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, c_float, _P, c_float, c_int, _P, _P, _P]),
    "haet_pc_radius_fill": (c_int, [_P, _P, _P, c_int, c_int, c_longlong, _P, _P]),
    "haet_pc_mpnn_message": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, c_longlong, c_int, c_float, _P, c_int, _P]),
    "haet_pc_ln_residual": (c_int, [_P, c_int, c_longlong, c_int, c_float, _P, c_int, _P, c_int, _P]),
    "haet_pc_gather_rows": (c_int, [_P, c_int, _P, c_longlong, c_int, _P, c_int, _P]),
    "haet_pc_scatter_masked": (c_int, [_P, c_int, _P, _P, c_longlong, c_int, _P, c_int, _P]),
    "haet_pc_rows_add": (c_int, [_P, c_int, _P, _P, c_int, c_longlong, c_int, _P]),
    "haet_pc_ball_rel": (c_int, [_P, _P, c_longlong, c_int, c_int, _P, _P]),
    "haet_pc_rms_pe": (c_int, [_P, c_int, _P, _P, c_longlong, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "haet_pc_ball_attention_supported": (c_int, [c_int, c_int, c_int]),
    "haet_pc_ball_attention": (c_int, [_P, c_int, c_longlong, c_int, c_int, c_int, _P, c_int, _P]),
    "haet_pc_swiglu": (c_int, [_P, c_int, c_longlong, c_int, _P, c_int, _P]),
    "haet_pc_pool_gather": (c_int, [_P, c_int, _P, c_longlong, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "haet_pc_unpool_input": (c_int, [_P, c_int, _P, _P, c_longlong, c_int, c_int, c_int, _P, c_int, _P]),
    "haet_pc_unpool_merge": (c_int, [_P, c_int, _P, c_int, c_longlong, c_int, c_int, _P, c_int, _P]),
    "haet_pc_rms_bwd": (c_int, [_P, c_int, _P, _P, c_int, _P, c_longlong, c_int, _P, c_int, c_int, _P]),
    "haet_pc_swiglu_bwd": (c_int, [_P, c_int, _P, c_int, c_longlong, c_int, _P, c_int, _P]),
    "haet_pc_ball_attention_bwd_supported": (c_int, [c_int, c_int, c_int]),
    "haet_pc_ball_attention_bwd": (c_int, [_P, c_int, _P, c_int, c_longlong, c_int, c_int, c_int, _P, c_int, _P]),
    "haet_pc_ln_bwd": (c_int, [_P, c_int, _P, c_int, c_longlong, c_int, c_float, _P, c_int, _P]),
    "haet_pc_mpnn_bwd_waves": (c_int, [c_longlong]),
    "haet_pc_mpnn_message_bwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, c_int, c_longlong, c_int, c_float, _P, c_int, _P, _P]),
    "haet_pc_bn_blocks": (c_int, [c_longlong]),
    "haet_pc_bn_stats": (c_int, [_P, c_int, c_longlong, c_int, c_float, _P, _P, _P, _P, _P, _P]),
    "haet_pc_bn_apply": (c_int, [_P, c_int, c_longlong, c_int, _P, _P, c_float, _P, _P, _P]),
}

_lib = None

# --------------------------------------------------------------------------- cache generations
# The modules cache weight-derived operands (bf16 planes, the packed Erwin blob) keyed on
# (data_ptr, _version) of their parameters, and the LIST of those parameter objects to avoid walking the
# module tree per forward.  A parameter or buffer OBJECT re-assigned by hand on any submodule goes through
# nn.Module.register_parameter / __setattr__, which fire torch's global registration hooks: every such
# event bumps this generation and the cached lists are re-walked at the next forward -- immediately, not
# after N calls.  (In-place writes through `.data` bump neither `_version` nor this: call
# `module.invalidate_caches()` after EMA swaps / manual clipping done that way.)
_generation = [0]


def _bump_generation(*_args):
    _generation[0] += 1
    return None


def cache_generation() -> int:
    return _generation[0]


def invalidate_all_caches() -> None:
    """Force every haet_amd module to re-derive its cached operands at its next forward."""
    _generation[0] += 1
    _invalidate_epoch[0] += 1


_invalidate_epoch = [0]


def invalidate_epoch() -> int:
    return _invalidate_epoch[0]


def _install_registration_hooks():
    import torch.nn.modules.module as _m

    _m.register_module_parameter_registration_hook(_bump_generation)
    _m.register_module_buffer_registration_hook(_bump_generation)


_install_registration_hooks()


_PRECISIONS = ("bf16x3", "fp16x2")
_precision = ["bf16x3"]


def precision() -> str:
    """Arithmetic of the matrix-core products: "bf16x3" (default: split bf16, three products per term, fp32-grade
    ~5e-6) or "fp16x2" (opt-in: two fp16 products per term where a kernel offers it -- today the convolution
    projection of the structured variants -- ~2e-4; north_star's bar is 1e-3).  HAET_PRECISION sets the initial value."""
    return _precision[0]


def set_precision(mode: str) -> None:
    """Range contract of "fp16x2": weights and activations inside the fp16 range (|v| <= 65504; a layer whose weight
    exceeds it stays on bf16x3 -- ops.SplitConv.f16_ok; activations are the caller's: post-LayerNorm inputs are O(1))."""
    if mode not in _PRECISIONS:
        raise ValueError(f"precision {mode!r}: expected 'bf16x3' or 'fp16x2'")
    _precision[0] = mode


# the environment's initial value goes through the same validation: a typo ("fp16") raises at import instead of
# silently running bf16x3
if "HAET_PRECISION" in os.environ:
    try:
        set_precision(os.environ["HAET_PRECISION"])
    except ValueError as e:
        raise ValueError(f"HAET_PRECISION: {e}") from None


def load():
    """Load libhaet_hip.so (raises HaetError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HaetError(
                f"{LIB_PATH} not found: build it with `make -C haet_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback for the HIP path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
            fn.restype, fn.argtypes = res, args
        if lib.haet_abi_version() != 9:
            raise HaetError(f"libhaet_hip.so ABI {lib.haet_abi_version()} != 9")
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().haet_last_error_string().decode(errors="replace")
        raise HaetError(f"{what} failed (code {rc}): {msg}")
