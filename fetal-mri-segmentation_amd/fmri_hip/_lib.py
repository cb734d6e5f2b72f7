"""Loads libfmri_hip.so and declares the argtypes of every entry point of include/fmri_hip.h."""
import ctypes as C
import os

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
IMPL_AUTO, IMPL_GENERIC, IMPL_MFMA = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMRI_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libfmri_hip.so")   # FMRI_LIB: A/B builds


class LibraryMissing(RuntimeError):
    pass


class FmriError(RuntimeError):
    pass


p, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
u64, u32 = C.c_uint64, C.c_uint32

# name -> argtypes; must list exactly the functions declared in include/fmri_hip.h (tests/test_abi.py checks both ways)
SIGNATURES = {
    "fmri_version": [],
    "fmri_error_string": [i32],
    "fmri_clock_stamp": [p, p],
    "fmri_conv3d_uses_mfma": [i32] * 7,
    "fmri_conv3d_fwd": [p, i32, i32, p, i32, p, p, p, p, i32, i32, i32, i32, i32, i32, f32, i32, i32, i32, p],
    "fmri_conv3d_fwd_tail_ok": [i32] * 7,
    "fmri_conv3d_fwd_tail": [p, i32, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, f32, i32, p],
    "fmri_conv3d_fwd_tail_planar_ok": [i32] * 7,
    "fmri_conv3d_fwd_tail_planar": [p, i32, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, f32, i32, p],
    "fmri_conv3d_dgrad": [p, i32, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_conv3d_wgrad": [p, i32, i32, p, i32, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p, i64, p],
    "fmri_conv3d_wgrad_workspace_bytes": [i32] * 9,
    "fmri_conv3d_wgrad_cout32_ok": [i32] * 10,
    "fmri_conv3d_pack_weights": [p, p, p, i32, i32, i32, p],
    "fmri_pack_weights_batched": [p, i32, i32, i32, p],
    "fmri_conv1x1_fwd": [p, p, p, p, i64, i32, i32, i32, p],
    "fmri_conv1x1_bwd": [p, p, p, p, p, p, i64, i32, i32, i32, i32, p],
    "fmri_sigmoid_dice_fwd": [p, p, p, p, i64, p],
    "fmri_sigmoid_dice_bwd": [p, p, p, p, i64, f32, f32, p],
    "fmri_sigmoid_loss_bwd": [p, p, p, p, i64, i32, f32, f32, f32, p],
    "fmri_weighted_dice_fwd": [p, p, p, p, i32, i64, i32, f32, p],
    "fmri_weighted_dice_bwd": [p, p, p, p, p, i32, i64, i32, f32, f32, p],
    "fmri_maxpool3d_2x_fwd": [p, p, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_maxpool3d_2x_bwd": [p, p, p, i32, i32, p, i32, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_upsample_nearest2x_fwd": [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_upsample_nearest2x_bwd": [p, i32, i32, p, p, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_norm_act_fwd": [p, p, p, p, p, p, i32, i64, i32, i32, f32, i32, i32, f32, i32, p],
    "fmri_norm_act_bwd": [p, p, p, p, p, p, p, p, p, i32, i64, i32, i32, i32, f32, i32, p],
    "fmri_norm_act_bwd_x": [p, p, p, p, p, p, p, p, p, i32, i64, i32, i32, i32, f32, i32, p],
    "fmri_conv3d_fwd_ntail_ok": [i32, i32, i32, i32, i32, i32, i32, i32],
    "fmri_norm_tail_ws_doubles": [i32, i32],
    "fmri_conv3d_fwd_stats": [p, i32, i32, p, i32, p, p, p, i32, i32, i32, i32, i32, i32, f32, p, i32, i32, p],
    "fmri_conv3d_upcat_fwd_stats": [p, i32, p, i32, p, p, p, p, i32, i32, i32, i32, i32, i32, f32, p, i32, i32, p],
    "fmri_norm_act_fwd_pre": [p, p, p, p, p, p, i32, i64, i32, i32, f32, i32, i32, f32, i32, p],
    "fmri_norm_scale_shift": [p, p, p, p, i32, i32, p],
    "fmri_norm_moving_update": [p, p, p, i32, C.c_double, f32, f32, p],
    "fmri_conv3d_dgrad_norm": [p, i32, p, p, p, p, i32, i32, i32, i32, i32, i32, f32, p, i32, i32, p],
    "fmri_norm_act_bwd_pre": [p, p, p, p, p, p, p, p, i32, i64, i32, i32, i32, p],
    "fmri_deconv3d_k2s2_fwd": [p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_deconv3d_k2s2_bwd": [p, p, p, i32, i32, p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_conv3d_direct_fwd": [p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, p],
    "fmri_conv3d_direct_bwd": [p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_conv2d_direct_fwd": [p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, p],
    "fmri_conv2d_direct_bwd": [p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_correlate1d_f64": [p, p, i32, i32, i32, i32, p, i32, p],
    "fmri_threshold_f64": [p, p, i64, C.c_double, p],
    "fmri_fill_holes_step": [p, p, p, i32, i32, i32, i32, i32, p, p],
    "fmri_largest_component_step": [p, p, p, p, p, i32, i32, i32, i32, i32, p, p],
    "fmri_add": [p, p, p, i64, i32, p],
    "fmri_act_bwd": [p, p, p, i32, f32, i64, i32, p],
    "fmri_slice_channels": [p, i32, i32, p, i32, i64, i32, i32, p],
    "fmri_channel_scale": [p, p, p, i32, i64, i32, i32, p],
    "fmri_adam_step": [p, p, p, p, i64, f32, f32, f32, f32, f32, p],
    "fmri_set_deterministic": [p, p, i64],
    "fmri_deterministic_finish": [p, p, i64, p],
    "fmri_tile_gather": [p, i32, i32, i32, p, i32, i32, i32, i32, p, i32, p],
    "fmri_tile_scatter_accumulate": [p, p, i32, i32, i32, i32, i32, p, p, i32, i32, i32, p],
    "fmri_tile_finalize": [p, p, p, p, i64, i32, p],
    "fmri_cast": [p, i32, p, i32, i64, p],
    "fmri_conv3d_upcat_ok": [i32, i32, i32, i32, i32, i32, i32],
    "fmri_conv3d_upcat_fwd_bias27": [p, i32, p, i32, p, p, p, p, i32, i32, i32, i32, i32, i32, f32, i32, p],
    "fmri_conv3d_upcat_wgrad_parts": [p, i32, p, i32, p, p, p, p, i32, i32, i32, i32, i32, i32, p, i64, p],
    "fmri_border_class_sums": [p, p, i32, i32, i32, i32, i32, i32, p],
    "fmri_conv3d_pack_up_weights": [p, i32, i32, i32, p, p, p, p, i32, p],
    "fmri_conv3d_upcat_fwd": [p, i32, p, i32, p, p, p, p, i32, i32, i32, i32, i32, i32, f32, i32, p],
    "fmri_conv3d_upcat_dgrad": [p, i32, p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_conv3d_stride2_fwd": [p, i32, p, p, p, i32, i32, i32, i32, i32, i32, p],
    "fmri_conv3d_upcat_wgrad": [p, i32, p, i32, p, p, p, p, i32, i32, i32, i32, i32, i32, p, i64, p],
    "fmri_conv2d_upcat_ok": [i32, i32, i32, i32, i32, i32, i32],
    "fmri_conv2d_pack_up_weights": [p, i32, i32, i32, p, p, p, p, i32, p],
    "fmri_conv2d_upcat_fwd": [p, i32, p, i32, p, p, p, p, i32, i32, i32, i32, i32, f32, i32, p],
    "fmri_conv2d_upcat_dgrad": [p, i32, p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, p],
    "fmri_conv2d_upcat_wgrad": [p, i32, p, i32, p, p, p, p, i32, i32, i32, i32, i32, p, i64, p],
    "fmri_sigmoid_dice_fwd_weighted": [p, p, p, p, p, i64, p],
    "fmri_sigmoid_loss_bwd_weighted": [p, p, p, p, p, i64, i32, f32, f32, f32, p],
    "fmri_affine_sample": [p, i32, i32, i32, i32, p, i32, i32, i32, i32, i32, i32, i32, f32, p, i32, i32, p],
    "fmri_minmax": [p, i64, i32, p, p],
    "fmri_rescale_intensity": [p, i64, i32, p, i32, f32, f32, f32, p],
    "fmri_noise_augment": [p, i64, i32, p, p, i32, f32, p],
    "fmri_shot_noise_step": [p, i64, i32, p, p, p, p, i32, p],
    "fmri_correlate1d_f32": [p, p, i32, i32, i32, i32, p, i32, i32, p],
    "fmri_elastic_warp": [p, i32, i32, i32, i32, i32, p, p, i32, p, i32, p],
    "fmri_coarse_dropout": [p, i32, i32, i32, i32, i32, p, i32, i32, i32, p, p],
    "fmri_affine_sample_batch": [i32, p, p, p, p, p, i32, i32, i32, i32, i32, p, i32, i32, i64, p],
    "fmri_minmax_ws_batch": [p, i64, i64, i32, i32, p, p, p],
    "fmri_rescale_intensity_ws_batch": [p, i64, i64, i32, i32, p, p, p, p],
    "fmri_noise_rng_batch": [p, i64, i64, i32, i32, p, p, i32, f32, u64, p, p],
    "fmri_shot_noise_rng_batch": [p, i64, i64, i32, i32, p, p, u64, p, p],
    "fmri_elastic_fields_rng_batch": [p, i32, i32, i32, p, p, u64, p, i32, p],
    "fmri_elastic_warp_batch": [p, i32, i32, i32, i32, i32, i64, p, i32, p, i32, i64, i32, p],
    "fmri_coarse_dropout_rng_batch": [p, i32, i32, i32, i32, i32, i64, i32, p, i32, f32, p, u64, p, p],
    "fmri_piecewise_affine2": [p, i32, i32, i32, i32, i32, p, i32, p, i32, p],
    "fmri_avgpool3d_2x_fwd": [p, p, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_avgpool3d_2x_bwd": [p, p, i32, i32, i32, i32, i32, i32, i32, p],
    "fmri_global_avgpool_fwd": [p, p, i32, i64, i32, i32, p],
    "fmri_global_avgpool_bwd": [p, p, i32, i64, i32, i32, p],
    "fmri_dense_fwd": [p, p, p, p, i32, i32, i32, i32, f32, p],
    "fmri_dense_bwd": [p, p, p, p, p, p, p, i32, i32, i32, i32, f32, p],
    "fmri_sigmoid_bce_fwd": [p, p, p, p, i64, p],
    "fmri_sigmoid_bce_bwd": [p, p, p, i64, f32, p],
    "fmri_sigmoid_chain": [p, p, i32, i32, p, i64, f32, i32, i32, p],
    "fmri_discriminator_input": [p, i32, p, i32, i32, p, i32, i32, i64, i32, p],
}

_lib = None


def lib():
    """The loaded library; raises LibraryMissing (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing("%s not found - build it with `python __graft_entry__.py` or `make -C "
                                 "fetal-mri-segmentation_amd/csrc` (no CPU fallback exists)" % LIB_PATH)
        # torch first: its wheel carries its own HIP runtime, and the library's kernels must register with the runtime whose streams
        # and allocations they are handed (loading this library before torch puts a second runtime in the process: every launch fails)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_char_p if name == "fmri_error_string" else (i64 if name.endswith(("_workspace_bytes", "_ws_doubles")) else i32)
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise FmriError("%s failed: %s (%d)" % (what, lib().fmri_error_string(rc).decode(), rc))
