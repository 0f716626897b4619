"""ctypes binding of libadsorbdiff_hip.so (the C ABI in include/adsorbdiff_hip.h).

This is the stub a reference maintainer would add.  There is
no fallback: if the library is missing or a call fails, an exception is
raised — ``RuntimeError`` for out-of-memory / HIP errors (so that
``ml_diffuse``'s split-and-retry contract works, reference
relaxation/ml_relaxation.py:146-165) and ``ValueError`` for an image without
neighbours (reference painn_denoising.py:370-375).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

ADF_OK, ADF_EINVAL, ADF_EOOM, ADF_ENONEIGHBOR, ADF_EHIP, ADF_EOVERFLOW, ADF_ENUMERIC = 0, 1, 2, 3, 4, 5, 6


class NumericRangeError(ArithmeticError):
    """Non-finite model output (ADF_ENUMERIC).  Not a RuntimeError on purpose: ``ml_diffuse`` must not answer it by
    splitting the batch; the engine answers it by re-running in exact f32."""

EXPORTS = (
    "adf_painn_create", "adf_painn_destroy", "adf_painn_set_weights", "adf_graph_build", "adf_graph_set_moving",
    "adf_check_flags", "adf_painn_set_arithmetic", "adf_painn_set_incremental", "adf_painn_set_fused_mlp",
    "adf_graph_export", "adf_painn_forward", "adf_painn_forward_subset", "adf_linear_forward", "adf_painn_message_layer", "adf_painn_update_layer",
    "adf_sde_init_placement", "adf_sde_step", "adf_sde_step_scheduled", "adf_sample", "adf_sample_traj",
    "adf_frames_create", "adf_frames_destroy", "adf_frames_push", "adf_frames_wait", "adf_frames_release", "adf_frames_pushed", "adf_frames_abort",
    "adf_get_counters", "adf_profile_enable", "adf_profile_read", "adf_measure_peaks",
    "adf_lift_adsorbates", "adf_comm_unique_id", "adf_comm_create", "adf_comm_destroy", "adf_allgather_sites",
    "adf_op_linear_fwd", "adf_op_linear_bwd_scratch", "adf_op_linear_bwd", "adf_op_ssilu_fwd", "adf_op_ssilu_bwd", "adf_op_layernorm_fwd", "adf_op_layernorm_bwd", "adf_op_embed_fwd", "adf_op_embed_bwd", "adf_op_rbf", "adf_op_message_fwd", "adf_op_message_fwd_fused", "adf_op_message_bwd", "adf_op_message_bwd_fused", "adf_op_message_bwd_fused_supported", "adf_op_message_bwd_perm", "adf_op_edge_owner", "adf_op_rbf_image_bytes", "adf_op_rbf_image", "adf_op_rbf_wgrad_fused_scratch", "adf_op_rbf_wgrad_fused", "adf_op_vdot_fwd", "adf_op_vdot_bwd", "adf_op_update_out_fwd", "adf_op_update_out_bwd", "adf_op_vnorm_fwd", "adf_op_vnorm_bwd", "adf_op_gate_fwd", "adf_op_gate_bwd", "adf_op_copy_rows", "adf_op_score_loss", "adf_op_sqnorm_accumulate", "adf_op_adamw_step",
    "adf_eqv2_create", "adf_eqv2_destroy", "adf_eqv2_set_constants", "adf_eqv2_set_weights", "adf_eqv2_set_arithmetic",
    "adf_eqv2_set_edges", "adf_eqv2_set_moving", "adf_eqv2_set_incremental", "adf_eqv2_forward", "adf_eqv2_forward_subset", "adf_eqv2_check_flags", "adf_eqv2_init_placement",
    "adf_eqv2_sde_step", "adf_eqv2_sample", "adf_eqv2_sample_traj", "adf_eqv2_linear_forward", "adf_eqv2_get_counters", "adf_eqv2_profile_enable", "adf_eqv2_profile_read",
    "adf_last_error", "adf_version",
)


class Hparams(C.Structure):
    _fields_ = [
        ("hidden_channels", C.c_int32), ("num_layers", C.c_int32), ("num_rbf", C.c_int32),
        ("num_elements", C.c_int32), ("max_neighbors", C.c_int32), ("envelope_exponent", C.c_int32),
        ("num_heads", C.c_int32), ("cutoff", C.c_float),
    ]


class EqV2Hparams(C.Structure):
    _fields_ = [
        ("lmax", C.c_int32), ("mmax", C.c_int32), ("num_layers", C.c_int32), ("sphere_channels", C.c_int32),
        ("attn_hidden_channels", C.c_int32), ("num_heads", C.c_int32), ("attn_alpha_channels", C.c_int32),
        ("attn_value_channels", C.c_int32), ("ffn_hidden_channels", C.c_int32), ("grid_resolution", C.c_int32),
        ("edge_channels", C.c_int32), ("num_distance_basis", C.c_int32), ("max_num_elements", C.c_int32),
        ("max_neighbors", C.c_int32), ("max_radius", C.c_float), ("avg_degree", C.c_float),
    ]


class EqV2Counters(C.Structure):
    _fields_ = [("num_edges", C.c_int64), ("num_atoms", C.c_int64), ("dense_flops", C.c_int64), ("conv_flops", C.c_int64),
                ("inc_rows", C.c_int64), ("inc_rows_full", C.c_int64), ("forwards_total", C.c_int64),
                ("conv_flops_total", C.c_int64)]


class BatchDesc(C.Structure):
    _fields_ = [
        ("num_systems", C.c_int32), ("num_atoms", C.c_int32), ("pos", C.c_void_p), ("cell", C.c_void_p),
        ("atomic_numbers", C.c_void_p), ("batch", C.c_void_p), ("atom_offset", C.c_void_p), ("reps", C.c_int32 * 3),
    ]


class StepCoef(C.Structure):
    _fields_ = [
        ("coef_tr", C.c_float), ("rot_pre", C.c_float), ("rot_dt", C.c_float), ("rot_g2", C.c_float),
        ("noise_tr", C.c_float), ("noise_rot", C.c_float),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("num_edges", C.c_int64), ("num_atoms", C.c_int64), ("message_bytes_per_layer", C.c_int64),
        ("dense_flops", C.c_int64),
        ("inc_rows", C.c_int64), ("inc_rows_full", C.c_int64), ("inc_msg_launches", C.c_int64),
        ("inc_msg_edges", C.c_int64),
    ]


_LIB = None


def lib_path() -> Path:
    import os

    override = os.environ.get("ADF_LIB_PATH")  # development only: A/B a differently built library
    if override:
        return Path(override)
    return Path(__file__).resolve().parent / "libadsorbdiff_hip.so"


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: build it with `python -m adsorbdiff_amd.build` "
            "(the HIP path has no CPU fallback)"
        )
    lib = C.CDLL(str(path))
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.adf_last_error.restype = C.c_char_p
    lib.adf_version.restype = C.c_char_p
    sigs = {
        "adf_painn_create": [C.POINTER(Hparams), C.POINTER(vp)],
        "adf_painn_destroy": [vp],
        "adf_painn_set_weights": [vp, i32, C.POINTER(vp), C.POINTER(C.c_float), vp],
        "adf_graph_build": [vp, C.POINTER(BatchDesc), vp, C.POINTER(i64)],
        "adf_graph_set_moving": [vp, vp, vp, vp],
        "adf_check_flags": [vp, vp],
        "adf_painn_set_arithmetic": [vp, i32],
        "adf_painn_set_incremental": [vp, i32],
        "adf_painn_set_fused_mlp": [vp, i32],
        "adf_graph_export": [vp, vp, vp, vp, i64, vp, vp, vp, vp, C.POINTER(i64), vp],
        "adf_painn_forward": [vp, C.POINTER(BatchDesc), vp, vp, vp],
        "adf_painn_forward_subset": [vp, C.POINTER(BatchDesc), vp, i32, vp, vp, vp],
        "adf_painn_message_layer": [vp, i32, i32, vp, vp, vp, vp, vp],
        "adf_linear_forward": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
        "adf_painn_update_layer": [vp, i32, i32, vp, vp, vp],
        "adf_sde_init_placement": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp],
        "adf_sde_step": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp, vp, C.POINTER(StepCoef), vp, vp, i32, vp, vp, vp, vp],
        "adf_sde_step_scheduled": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp],
        "adf_get_counters": [vp, C.POINTER(Counters), vp],
        "adf_profile_enable": [vp, i32],
        "adf_profile_read": [vp, C.POINTER(C.c_float), C.POINTER(i64), C.POINTER(i64), vp],
        "adf_measure_peaks": [C.POINTER(C.c_float), vp],
        "adf_lift_adsorbates": [vp, vp, vp, i32, C.c_float, vp, vp],
        "adf_comm_unique_id": [vp],
        "adf_comm_create": [vp, i32, i32, C.POINTER(vp)],
        "adf_comm_destroy": [vp],
        "adf_allgather_sites": [vp, vp, i64, vp, vp],
        "adf_op_linear_fwd": [vp, i32, vp, vp, vp, i32, i64, i32, i32, vp],
        "adf_op_linear_bwd": [vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, i32, i64, i32, i32, vp, vp],
        "adf_op_ssilu_fwd": [vp, vp, i64, vp],
        "adf_op_ssilu_bwd": [vp, vp, vp, i64, vp],
        "adf_op_layernorm_fwd": [vp, vp, vp, vp, vp, i32, i32, vp],
        "adf_op_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp],
        "adf_op_embed_fwd": [vp, vp, i32, vp, vp],
        "adf_op_embed_bwd": [vp, vp, vp, i32, i32, vp],
        "adf_op_rbf": [vp, vp, vp],
        "adf_op_message_fwd": [vp, vp, vp, vp, vp, vp, vp, i32, vp],
        "adf_op_message_fwd_fused": [vp, i32, vp, vp, vp, vp, vp, i32, vp],
        "adf_op_message_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp],
        "adf_op_message_bwd_fused": [vp, i32, vp, vp, vp, vp, vp, vp, C.c_int64, vp, vp, i32, vp, vp],
        "adf_op_edge_owner": [vp, vp, C.c_int64, vp],
        "adf_op_rbf_image": [vp, vp, C.c_int64, vp, vp],
        "adf_op_rbf_wgrad_fused": [vp, vp, vp, vp, vp, C.c_int64, vp, vp, i32, vp],
        "adf_op_message_bwd_fused_supported": [vp],
        "adf_op_message_bwd_perm": [vp, vp, i32],
        "adf_op_vdot_fwd": [vp, vp, vp, i32, i64, i32, C.c_float, vp],
        "adf_op_vdot_bwd": [vp, vp, i32, vp, vp, i32, vp, vp, i64, i32, vp],
        "adf_op_update_out_fwd": [vp, vp, vp, vp, vp, C.c_float, vp, vp, i64, i32, vp],
        "adf_op_update_out_bwd": [vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp],
        "adf_op_vnorm_fwd": [vp, vp, i32, i64, i32, vp],
        "adf_op_vnorm_bwd": [vp, vp, i32, vp, i32, vp, i64, i32, vp],
        "adf_op_gate_fwd": [vp, vp, vp, i32, vp, i64, i32, vp],
        "adf_op_gate_bwd": [vp, vp, vp, i32, vp, vp, vp, i64, i32, vp],
        "adf_op_copy_rows": [vp, i32, vp, i32, i64, i32, i32, vp],
        "adf_op_score_loss": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp],
        "adf_op_sqnorm_accumulate": [vp, i64, vp, vp],
        "adf_op_adamw_step": [vp, vp, vp, vp, vp, i64, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, C.c_float, vp],
        "adf_eqv2_create": [C.POINTER(EqV2Hparams), C.POINTER(vp)],
        "adf_eqv2_destroy": [vp],
        "adf_eqv2_set_constants": [vp, vp, vp, vp, vp, vp],
        "adf_eqv2_set_weights": [vp, i32, C.POINTER(vp), vp],
        "adf_eqv2_set_arithmetic": [vp, i32],
        "adf_eqv2_set_edges": [vp, i64, vp, vp, vp, i32, vp],
        "adf_eqv2_set_moving": [vp, vp, vp, vp],
        "adf_eqv2_set_incremental": [vp, i32],
        "adf_eqv2_forward": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp],
        "adf_eqv2_forward_subset": [vp, C.POINTER(BatchDesc), vp, i32, vp, vp, vp],
        "adf_eqv2_check_flags": [vp, vp],
        "adf_eqv2_init_placement": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp],
        "adf_eqv2_sde_step": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp, vp, C.POINTER(StepCoef), vp, i32, vp, vp, i32, vp, vp, vp, vp],
        "adf_eqv2_sample": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp],
        "adf_eqv2_get_counters": [vp, C.POINTER(EqV2Counters), vp],
        "adf_eqv2_linear_forward": [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp],
        "adf_eqv2_profile_enable": [vp, i32],
        "adf_eqv2_profile_read": [vp, C.POINTER(C.c_float), C.POINTER(i64), vp],
        "adf_sample": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp],
        "adf_sample_traj": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, i32, vp],
        "adf_eqv2_sample_traj": [vp, C.POINTER(BatchDesc), vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, i32, vp],
        "adf_frames_create": [i32, i64, i32, C.POINTER(vp)],
        "adf_frames_destroy": [vp],
        "adf_frames_push": [vp, vp, vp],
        "adf_frames_wait": [vp, i64, i32, C.POINTER(C.POINTER(C.c_float))],
        "adf_frames_release": [vp, i64],
        "adf_frames_abort": [vp],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = i32
    lib.adf_op_linear_bwd_scratch.argtypes = [i64, i32, i32]
    lib.adf_op_linear_bwd_scratch.restype = i64
    lib.adf_op_rbf_image_bytes.argtypes = [i64]
    lib.adf_op_rbf_image_bytes.restype = i64
    lib.adf_op_rbf_wgrad_fused_scratch.argtypes = [vp]
    lib.adf_op_rbf_wgrad_fused_scratch.restype = i64
    lib.adf_frames_pushed.argtypes = [vp]
    lib.adf_frames_pushed.restype = i64
    _LIB = lib
    return lib


def check(status: int) -> None:
    """Translate a status code into the exception the reference's callers expect."""
    if status == ADF_OK:
        return
    msg = load().adf_last_error().decode("utf-8", "replace")
    if status == ADF_ENONEIGHBOR:
        raise ValueError(msg)
    if status == ADF_EOOM:
        raise RuntimeError(f"HIP out of memory: {msg}")
    if status == ADF_ENUMERIC:
        raise NumericRangeError(msg)
    if status == ADF_EINVAL:
        raise ValueError(f"adsorbdiff_hip: invalid argument: {msg}")
    raise RuntimeError(f"adsorbdiff_hip error {status}: {msg}")
