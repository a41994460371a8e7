"""ctypes binding of libgsr_hip.so — field-for-field mirror of include/gsr.h.

The product path has no CPU fallback: if the HIP library is missing or fails to
load, importing anything that needs it raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_HIP_LIB", os.path.join(_HERE, "libgsr_hip.so"))  # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gsr.h")

GSR_OK, GSR_E_INVALID_ARG, GSR_E_OOM, GSR_E_HIP, GSR_E_STATE = 0, -1, -2, -3, -4
DEFAULT = 0  # GSR_DEFAULT (ABI 6: 0, so that a zero-initialised gsr_config is the default config)
SSIM_FAST, SSIM_EXACT = 1, 2                      # gsr_config.ssim_precision
PREPROCESS_DIRECT, PREPROCESS_AGGREGATING = 1, 2  # gsr_config.preprocess_form
TUNER_OFF, TUNER_ON = 1, 2                        # gsr_config.form_tuner
GRAD_FP32_REFERENCE, GRAD_ACCURATE = 1, 2         # gsr_config.grad_precision
MODES = {"rgb": 3, "rgbd": 5, "rgbdn": 8}
FORWARD_ONLY = 1  # gsr_aux.flags: no backward state is kept (inference render)
FLAG_REFERENCE_TILE_LISTS = 2  # flags = 0: exact footprint culling (the default); bit 1 is retired (rejected)
ABI_VERSION = 6  # GSR_ABI_VERSION of the include/gsr.h this mirror was written against

(BUF_RADII, BUF_GRAD_MEANS2D, BUF_N_CONTRIB, BUF_FINAL_T, BUF_TILE_RANGES, BUF_VALUES_SORTED, BUF_GEOM, BUF_NORMALS,
 BUF_GRAD_ROWS, BUF_INSTANCE_AUX) = range(10)


class GsrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gsr error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("mode", C.c_int32), ("near_plane", C.c_float),
                ("far_plane", C.c_float), ("radius_clip", C.c_int32), ("blur_eps", C.c_float), ("flags", C.c_uint32),
                ("bins_budget_bytes", C.c_uint64), ("ssim_precision", C.c_int32), ("preprocess_form", C.c_int32),
                ("form_tuner", C.c_int32), ("grad_precision", C.c_int32)]


class Inputs(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_coeffs", C.c_int32), ("sh_degree", C.c_int32), ("means", C.c_void_p),
                ("shs", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
                ("background", C.c_float * 3)]


class CameraS(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("t", C.c_float * 3), ("focal", C.c_float * 2), ("principal", C.c_float * 2),
                ("camera_center", C.c_float * 3), ("R_dev", C.c_void_p), ("t_dev", C.c_void_p)]


class Aux(C.Structure):
    _fields_ = [("covisibilities", C.c_void_p), ("uncertainties", C.c_void_p), ("radii", C.c_void_p),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("n_rendered", C.c_int64), ("n_visible", C.c_int32), ("max_tile_instances", C.c_int32),
                ("generation", C.c_uint64), ("bins_bytes", C.c_int64), ("compact_binning", C.c_int32), ("preprocess_form", C.c_int32),
                # ABI 6: the handle's view history (gsr_policy.h), cumulative since gsr_create
                ("bins_regrowths", C.c_uint32), ("compact_fallbacks", C.c_uint32), ("tuner_rearms", C.c_uint32),
                ("scratch_regrowths", C.c_uint32), ("fused_relaunches", C.c_uint32), ("held_views", C.c_uint32),
                ("bin_capacity", C.c_uint32), ("tuner_form", C.c_int32), ("tuner_ms", C.c_float * 2),
                ("tier_tiles", C.c_uint32 * 3), ("reserved", C.c_uint32)]

    HISTORY = ("bins_regrowths", "compact_fallbacks", "tuner_rearms", "scratch_regrowths", "fused_relaunches", "held_views")

    def history(self) -> dict:
        return {k: int(getattr(self, k)) for k in self.HISTORY}


class Grads(C.Structure):
    _fields_ = [("vmeans", C.c_void_p), ("vshs", C.c_void_p), ("vopacities", C.c_void_p), ("vscales", C.c_void_p),
                ("vrotations", C.c_void_p), ("vR", C.c_void_p), ("vt", C.c_void_p), ("vcolors", C.c_void_p),
                ("vmeans2d", C.c_void_p), ("forward_generation", C.c_uint64), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


GRADS_COLOR_COTANGENT = 0x1  # gsr_grads.flags / gsr_tail_state.flags: channels >= 3 of vpixels are zeros (the loss head's cotangent)
ADAM_MAX_GROUPS = 8


class AdamGroup(C.Structure):
    _fields_ = [("theta", C.c_void_p), ("grad", C.c_void_p), ("mu", C.c_void_p), ("nu", C.c_void_p),
                ("count", C.c_int64), ("lr", C.c_float), ("current_step", C.c_uint32)]


class TailGrads(C.Structure):
    _fields_ = [("vmeans", C.c_void_p), ("vshs", C.c_void_p), ("vopacities", C.c_void_p), ("vscales", C.c_void_p),
                ("vrotations", C.c_void_p)]


class TailState(C.Structure):  # gsr_tail_state
    _fields_ = [("theta", C.c_void_p * 6), ("mu", C.c_void_p * 6), ("nu", C.c_void_p * 6), ("lr", C.c_float * 6),
                ("current_step", C.c_uint32 * 6), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("scale_dims", C.c_int32), ("shs", C.c_void_p), ("opacities_act", C.c_void_p),
                ("scales_act", C.c_void_p), ("vmeans2d", C.c_void_p), ("forward_generation", C.c_uint64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class ComposeGroup(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_words", C.c_int32), ("new_zero", C.c_int32)]


COMPOSE_MAX_GROUPS = 24
DENSIFY_CLONE, DENSIFY_SPLIT, DENSIFY_PRUNE = 0, 1, 2


class GatherGroup(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_words", C.c_int32)]


# ---- include/gsr_policy.h: the library's policies as GPU-free functions ----
class FormTuner(C.Structure):
    _fields_ = [("phase", C.c_int32), ("form", C.c_int32), ("n_ref", C.c_int32), ("age", C.c_uint32), ("ms", C.c_float * 2)]


class PolicyConfig(C.Structure):
    _fields_ = [("grid_x", C.c_int32), ("grid_y", C.c_int32), ("bins_budget_bytes", C.c_uint64), ("preprocess_form", C.c_int32),
                ("form_tuner", C.c_int32), ("beside_max_tiles", C.c_uint32), ("bwd_split_max_tiles", C.c_uint32),
                ("agg_max_bands", C.c_int32), ("reserved", C.c_int32)]


class PolicyState(C.Structure):
    _fields_ = [("bin_cap", C.c_uint32), ("compact_sticky", C.c_uint32), ("last_n", C.c_int32), ("last_max_tile", C.c_uint32),
                ("last_n_rendered", C.c_int64), ("tier_n", C.c_uint32 * 3), ("bin_cap_view", C.c_uint32), ("views", C.c_uint64),
                ("tuner", FormTuner), ("bins_regrowths", C.c_uint32), ("compact_fallbacks", C.c_uint32),
                ("compact_views", C.c_uint32), ("overflow_views", C.c_uint32), ("tuner_rearms", C.c_uint32),
                ("fused_relaunches", C.c_uint32), ("held_views", C.c_uint32), ("reserved", C.c_uint32)]


class ViewPlan(C.Structure):
    _fields_ = [("bin_cap_view", C.c_uint32), ("form_request", C.c_int32), ("form", C.c_int32), ("timed_slot", C.c_int32),
                ("skewed", C.c_int32), ("hold_fused", C.c_int32), ("tuner_decided", C.c_int32), ("spec_mid4", C.c_uint32),
                ("spec_mid8", C.c_uint32), ("reserved", C.c_int32)]


class ViewOutcome(C.Structure):
    _fields_ = [("binning", C.c_int32), ("fused_done", C.c_int32), ("long_tiles", C.c_int32), ("beside", C.c_int32),
                ("launch_fused_now", C.c_int32), ("reserved", C.c_int32), ("sorted_mid4", C.c_uint32), ("sorted_mid8", C.c_uint32),
                ("bin_cap_next", C.c_uint32), ("bins_regrown", C.c_uint32)]


class BwdSplit(C.Structure):
    _fields_ = [("n_big", C.c_uint32), ("n_mid8", C.c_uint32), ("n_mid4", C.c_uint32), ("split_len", C.c_uint32)]


POLICY_EXPORTS = ["gsr_policy_config_init", "gsr_policy_state_init", "gsr_policy_agg_plan", "gsr_policy_preprocess_form",
                  "gsr_policy_form_is_open", "gsr_policy_begin_view", "gsr_policy_end_view", "gsr_policy_bwd_split"]


def bind_policy(lib):
    """argtypes of the gsr_policy_* exports on `lib` (libgsr_hip.so, or the g++-only build tests/test_policy.py makes)."""
    P = C.POINTER
    lib.gsr_policy_config_init.argtypes = [P(PolicyConfig), C.c_int32, C.c_int32, C.c_uint64, C.c_int32]
    lib.gsr_policy_config_init.restype = None
    lib.gsr_policy_state_init.argtypes = [P(PolicyState)]
    lib.gsr_policy_state_init.restype = None
    lib.gsr_policy_agg_plan.argtypes = [C.c_int32, C.c_int32, C.c_uint32, P(C.c_int32), P(C.c_int32), P(C.c_size_t), P(C.c_int32)]
    lib.gsr_policy_agg_plan.restype = None
    lib.gsr_policy_preprocess_form.argtypes = [P(PolicyConfig), C.c_int32, C.c_int32, C.c_uint32, C.c_int32]
    lib.gsr_policy_preprocess_form.restype = C.c_int32
    lib.gsr_policy_form_is_open.argtypes = [P(PolicyConfig), C.c_int32, C.c_uint32]
    lib.gsr_policy_form_is_open.restype = C.c_int32
    lib.gsr_policy_begin_view.argtypes = [P(PolicyConfig), P(PolicyState), C.c_int32, P(C.c_float), P(ViewPlan)]
    lib.gsr_policy_begin_view.restype = None
    lib.gsr_policy_end_view.argtypes = [P(PolicyConfig), P(PolicyState), P(ViewPlan), C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_uint64, C.c_int32, P(ViewOutcome)]
    lib.gsr_policy_end_view.restype = None
    lib.gsr_policy_bwd_split.argtypes = [P(PolicyConfig), C.c_uint32, C.c_uint32, C.c_uint32, P(BwdSplit)]
    lib.gsr_policy_bwd_split.restype = None
    lib.gsr_bins_capacity_after.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32]
    lib.gsr_bins_capacity_after.restype = C.c_uint32
    return lib


EXPORTS = ["gsr_create", "gsr_destroy", "gsr_release_scene_buffers", "gsr_memory_usage", "gsr_reserve", "gsr_bins_capacity_after", "gsr_forward",
           "gsr_backward", "gsr_host_wait_policy", "gsr_buffer", "gsr_copy_buffer", "gsr_ssim_forward", "gsr_ssim_backward", "gsr_loss_l1_ssim",
           "gsr_ssim_precision", "gsr_get_ssim_precision", "gsr_preprocess_form", "gsr_get_preprocess_form",
           "gsr_allreduce_grads", "gsr_last_error_string", "gsr_version", "gsr_abi_version", "gsr_check_abi", "gsr_profile_enable",
           "gsr_profile_stage_count", "gsr_profile_stages", "gsr_profile_stage_name", "gsr_profile_read", "gsr_profile_read_intervals", "gsr_update_stats",
           "gsr_prologue_forward", "gsr_prologue_backward", "gsr_adam_step", "gsr_stream_triad",
           "gsr_mask_findall_scratch_bytes", "gsr_mask_findall", "gsr_gather_rows", "gsr_sh_grad_from_views", "gsr_sh_grad_from_views_tail", "gsr_trainer_tail_step",
           "gsr_backward_trainer_tail",
           "gsr_densify_grad_mean", "gsr_densify_mask", "gsr_compose_rows", "gsr_split_transform", "gsr_reset_opacity", "gsr_morton_codes",
           "gsr_ply_pack_rows", "gsr_ply_unpack_rows", "gsr_count_nonfinite"] + POLICY_EXPORTS

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).  The product only: the native
    test program of tests/test_gpu_host_wait.py is `build_tools()` (a link failure of a test tool must not read
    as 'building libgsr_hip.so failed')."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libgsr_hip.so failed")
    return LIB_PATH


def build_tools(verbose: bool = False) -> bool:
    """`make tools`: host_wait_threads (host code through the C ABI).  Returns False — with the compiler's words on stdout —
    when it does not build; never raises."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "tools"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-2000:])
    return r.returncode == 0


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
    # torch bundles its own libamdhip64.so.7; it must be the first HIP runtime in the process,
    # otherwise two runtimes get loaded and the second one finds no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    lib.gsr_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.gsr_destroy.argtypes = [vp]
    lib.gsr_release_scene_buffers.argtypes = [vp]
    lib.gsr_memory_usage.argtypes = [vp]
    lib.gsr_memory_usage.restype = C.c_int64
    lib.gsr_reserve.argtypes = [vp, C.c_int64, C.c_int64]
    bind_policy(lib)
    lib.gsr_forward.argtypes = [vp, C.POINTER(Inputs), C.POINTER(CameraS), vp, C.POINTER(Aux), vp, C.POINTER(Stats)]
    lib.gsr_backward.argtypes = [vp, C.POINTER(Inputs), C.POINTER(CameraS), vp, C.POINTER(Grads), vp]
    lib.gsr_host_wait_policy.argtypes = [i32, i32, i32]
    lib.gsr_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.gsr_copy_buffer.argtypes = [vp, i32, vp, C.c_size_t, vp]
    lib.gsr_ssim_forward.argtypes = [i32, i32, i32, i32, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp]
    lib.gsr_ssim_backward.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_loss_l1_ssim.argtypes = [vp, vp, vp, f32, vp, vp, vp]
    lib.gsr_allreduce_grads.argtypes = [vp, vp, C.c_size_t, vp]
    lib.gsr_update_stats.argtypes = [vp, vp, vp, vp, vp]
    lib.gsr_prologue_forward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_prologue_backward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_adam_step.argtypes = [C.POINTER(AdamGroup), i32, f32, f32, f32, vp]
    lib.gsr_mask_findall_scratch_bytes.argtypes = [C.c_int64]
    lib.gsr_mask_findall_scratch_bytes.restype = C.c_size_t
    lib.gsr_mask_findall.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
    lib.gsr_gather_rows.argtypes = [C.POINTER(GatherGroup), i32, vp, C.c_int64, vp]
    i64 = C.c_int64
    lib.gsr_densify_grad_mean.argtypes = [i64, vp, vp, vp, vp]
    lib.gsr_densify_mask.argtypes = [C.c_int32, i64, i64, vp, vp, C.c_int32, vp, vp, f32, f32, f32, C.c_int32, vp, vp]
    lib.gsr_compose_rows.argtypes = [C.POINTER(ComposeGroup), C.c_int32, vp, i64, vp, i64, C.c_int32, vp]
    lib.gsr_split_transform.argtypes = [i64, C.c_int32, vp, vp, vp, C.c_uint32, vp]
    lib.gsr_reset_opacity.argtypes = [i64, vp, vp]
    lib.gsr_morton_codes.argtypes = [i64, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp]
    lib.gsr_count_nonfinite.argtypes = [C.POINTER(vp), C.POINTER(C.c_int32), C.c_int32, i64, vp, vp, vp]
    lib.gsr_ply_pack_rows.argtypes = [i64, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_ply_unpack_rows.argtypes = [i64, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_sh_grad_from_views.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.gsr_sh_grad_from_views_tail.argtypes = [i32, i32, i32, i32, vp, vp, C.POINTER(TailGrads), C.POINTER(TailState), vp]
    lib.gsr_trainer_tail_step.argtypes = [i32, i32, i32, C.POINTER(TailGrads), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                          C.POINTER(f32), C.POINTER(C.c_uint32), f32, f32, f32, vp, vp, vp, vp]
    lib.gsr_backward_trainer_tail.argtypes = [vp, C.POINTER(Inputs), C.POINTER(CameraS), vp, C.POINTER(TailState), vp]
    lib.gsr_stream_triad.argtypes = [vp, vp, vp, C.c_size_t, f32, vp]
    lib.gsr_ssim_precision.argtypes = [i32]
    lib.gsr_preprocess_form.argtypes = [i32]
    lib.gsr_profile_enable.argtypes = [vp, i32]
    lib.gsr_profile_stages.argtypes = [vp, C.c_uint32]
    lib.gsr_profile_stage_name.argtypes = [i32]
    lib.gsr_profile_stage_name.restype = C.c_char_p
    lib.gsr_profile_read_intervals.argtypes = [vp, i32, C.POINTER(C.c_double), i32, C.POINTER(C.c_int)]
    lib.gsr_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int), i32]
    lib.gsr_last_error_string.restype = C.c_char_p
    lib.gsr_version.restype = C.c_char_p
    lib.gsr_check_abi.argtypes = [i32] + [C.c_size_t] * 7
    # a stale library next to a newer mirror (or the reverse) must fail here, loudly, not mis-read structs later
    rc = lib.gsr_check_abi(ABI_VERSION, *[C.sizeof(t) for t in (Config, Inputs, CameraS, Aux, Stats, Grads, TailState)])
    if rc != 0:
        raise GsrError(rc, lib.gsr_last_error_string().decode())
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise GsrError(rc, load().gsr_last_error_string().decode())
