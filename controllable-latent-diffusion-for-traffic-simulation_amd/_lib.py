"""ctypes binding of libcld_hip.so (C-ABI: include/cld.h).

The product path has NO fallback: if the HIP library is missing or a call
fails, a `CldError` is raised.  Nothing under `oracle/` is ever imported here.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# CLD_LIB_PATH selects an experimental build of the same library (never a different backend)
LIB_PATH = os.environ.get("CLD_LIB_PATH") or os.path.join(PKG_DIR, "libcld_hip.so")


class CldError(RuntimeError):
    pass


class CldConfig(C.Structure):
    _fields_ = [
        ("horizon", C.c_int32), ("latent_dim", C.c_int32), ("cond_dim", C.c_int32), ("base_dim", C.c_int32),
        ("dim_mults", C.c_int32 * 3), ("hidden", C.c_int32), ("n_timesteps", C.c_int32),
        ("step_time", C.c_float), ("acce_bound", C.c_float * 2), ("v_bound", C.c_float * 2),
        ("max_steer", C.c_float), ("max_yawvel", C.c_float),
        ("norm_mean", C.c_float * 6), ("norm_std", C.c_float * 6),
        ("precision", C.c_int32),
    ]


PRECISIONS = {"f32": 0, "f16x2": 1}
OPTIMIZERS = {"adam": 0, "sgd": 1}
KERNELS = {"guide": 0, "decode": 1, "encode": 2, "unet": 3, "context": 4, "conv5": 5}        # cld_debug_force_kernel
FORMS = {"auto": 0, "valu": 1, "mfma": 2, "quad": 3,         # "quad": guide kernel only
         "layers": 1, "chain": 2, "chain1": 3, "chain4": 4, "chainw": 5, "chainw2": 6, "chainw1": 7,                              # "unet" only: one launch per layer / LDS-resident layer chains
         "direct": 1, "winograd": 2, "winograd_whole": 3, "winograd_ksplit": 4, "winograd_f2": 3}                                                     # "context": the 3x3 / stride-1 convolutions of the ResNet-18; "conv5": the 256 -> 256 k5 layers of the U-Net


class CldGuidance(C.Structure):
    """include/cld.h `cld_guidance` (device pointers + optimiser settings of the sampling-time guidance step)."""
    _fields_ = [("curr_states", C.c_void_p), ("target_speed", C.c_void_p), ("loss_scale", C.c_void_p),
                ("lr", C.c_float), ("perturb_th", C.c_float), ("optimizer", C.c_int32),
                ("speed_limit", C.c_float), ("acc_limit", C.c_float),
                ("speed_limit_scale", C.c_void_p), ("acc_limit_scale", C.c_void_p),
                ("target_pos", C.c_void_p), ("target_time", C.c_void_p), ("target_pos_scale", C.c_void_p),
                ("ext_grad", C.c_void_p),
                ("apply_output", C.c_int32), ("no_intermediate", C.c_int32),
                ("final_lr", C.c_float), ("final_perturb_th", C.c_float), ("final_optimizer", C.c_int32),
                ("grad_steps", C.c_int32), ("final_grad_steps", C.c_int32), ("guide_clean", C.c_int32),
                ("collision", C.c_void_p),        # const cld_collision*
                ("map_collision", C.c_void_p)]    # const cld_map_collision*


class CldCollision(C.Structure):
    """include/cld.h `cld_collision` (upstream's AgentCollisionLoss configured per scene, device pointers)."""
    _fields_ = [("extent", C.c_void_p), ("world_from_agent", C.c_void_p), ("curr_speed", C.c_void_p),
                ("scene_start", C.c_void_p), ("scene_weight", C.c_void_p), ("guided", C.c_void_p),
                ("num_scenes", C.c_int32), ("num_samp", C.c_int32), ("num_disks", C.c_int32), ("max_scene_agents", C.c_int32),
                ("buffer_dist", C.c_float), ("decay_rate", C.c_float), ("moving_speed_th", C.c_float), ("excluded", C.c_void_p)]


class CldMapCollision(C.Structure):
    """include/cld.h `cld_map_collision` (upstream's MapCollisionLoss configured per scene, device pointers)."""
    _fields_ = [("extent", C.c_void_p), ("raster_from_agent", C.c_void_p), ("drivable_map", C.c_void_p), ("curr_speed", C.c_void_p),
                ("scene_start", C.c_void_p), ("scene_weight", C.c_void_p),
                ("num_scenes", C.c_int32), ("num_samp", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("num_points_l", C.c_int32), ("num_points_w", C.c_int32), ("decay_rate", C.c_float), ("moving_speed_th", C.c_float)]


_P = C.c_void_p
# name -> (restype, argtypes); every symbol include/cld.h declares
SIGNATURES = {
    "cld_default_config": (None, [C.POINTER(CldConfig)]),
    "cld_create": (C.c_int, [C.POINTER(CldConfig), C.POINTER(_P)]),
    "cld_destroy": (C.c_int, [_P]),
    "cld_last_error": (C.c_char_p, [_P]),
    "cld_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    "cld_finalize": (C.c_int, [_P, _P]),
    "cld_workspace_bytes": (C.c_size_t, [_P, C.c_int32]),
    "cld_get_schedule": (C.c_int, [_P, _P, _P, _P]),
    "cld_unet_forward": (C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int32, _P, C.c_size_t, _P]),
    "cld_unet_forward_t": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P, C.c_size_t, _P]),
    "cld_denoise_loss": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, C.c_size_t, _P]),
    "cld_ddpm_step": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P, C.POINTER(C.c_float), C.c_int32, _P, C.c_size_t, _P]),
    "cld_sample": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, _P, _P, C.c_int32, C.c_uint64, _P, C.c_size_t, _P]),
    "cld_sample_cfg": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.c_int32, _P, _P, _P, C.c_int32, C.c_uint64, _P, C.c_size_t, _P]),
    "cld_sample_guided": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.POINTER(CldGuidance), C.c_int32, _P, _P, _P, C.c_int32,
                                    C.c_uint64, _P, C.c_size_t, _P]),
    "cld_sample_step": (C.c_int, [_P, _P, _P, _P, C.c_float, C.POINTER(CldGuidance), C.c_int32, _P, _P, _P, _P, _P, C.POINTER(C.c_float), C.c_int32, _P, C.c_size_t, _P]),
    "cld_guidance_step": (C.c_int, [_P, _P, _P, C.POINTER(CldGuidance), C.c_float, _P, _P, _P, _P, C.c_int32, _P, C.c_size_t, _P]),
    "cld_guidance_losses": (C.c_int, [_P, _P, C.POINTER(CldGuidance), _P, C.c_int32, _P]),
    "cld_log_prob": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, C.c_int32, _P, C.c_size_t, _P]),
    "cld_lstm_decode": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P]),
    "cld_action_to_state": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cld_decode": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "cld_traj2z": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "cld_vae_loss": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, _P, C.c_int32, _P, C.c_size_t, _P]),
    "cld_state_to_state_and_action": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "cld_context_workspace_bytes": (C.c_size_t, [_P, C.c_int32]),
    "cld_context_encode": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P, C.c_size_t, _P]),
    "cld_context_combine": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, _P]),
    "cld_compute_reward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_float, _P, _P, _P,
                                     C.c_int32, _P]),
    "cld_agent_collision": (C.c_int, [_P, _P, C.POINTER(CldCollision), _P, _P, _P, C.c_int32, _P]),
    "cld_map_collision_loss": (C.c_int, [_P, _P, C.POINTER(CldMapCollision), _P, _P, _P, C.c_int32, _P]),
    "cld_world_step": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, _P, C.c_int32, _P]),
    "cld_profile_enable": (C.c_int, [_P, C.c_int32]),
    "cld_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "cld_profile_read_executed": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "cld_set_stride": (C.c_int, [_P, C.c_int32]),
    "cld_debug_lds_floor": (C.c_int, [_P, C.c_size_t]),
    "cld_debug_force_kernel": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "cld_debug_conv5_form": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    "cld_debug_stamps": (C.c_int, [_P, _P, C.c_int32]),
    "cld_debug_guide_stamps": (C.c_int, [_P]),
    "cld_get_precision": (C.c_int, [_P]),
    "cld_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libcld_hip.so (built by `__graft_entry__.build()` / `build.sh`); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CldError(f"{LIB_PATH} not found: build the HIP library first "
                       f"(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # ONE HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 (same SONAME as
    # /opt/rocm's).  Importing torch first makes libcld_hip.so bind to that already-loaded copy, so
    # torch's streams / device pointers and our launches live in the same runtime.  Loading this
    # library first would pull /opt/rocm's runtime in beside torch's and break device init.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(handle, rc, what):
    if rc != 0:
        msg = load().cld_last_error(handle)
        raise CldError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
