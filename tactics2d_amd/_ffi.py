"""ctypes binding of libt2d_hip.so -- the ONLY compute path of this package.

There is no CPU fallback: if the shared library (built in-tree by `python -m
tactics2d_amd.build`) is missing, or no HIP device is usable, the calls raise.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("T2D_LIB_NAME", "libt2d_hip.so"))

OK, ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_GEOMETRY, ERR_ACTION = 0, 1, 2, 3, 4, 5, 6


class T2DError(RuntimeError):
    """A libt2d_hip.so call returned a non-zero status."""

    def __init__(self, code, msg):
        super().__init__(f"libt2d_hip error {code}: {msg}")
        self.code = code


class GeometryError(T2DError, ValueError):
    pass


class StatusConfig(C.Structure):
    """t2d_status_config (include/t2d.h)."""
    _fields_ = [("max_step", C.c_int32), ("ego_index", C.c_int32), ("check_dynamic", C.c_int32),
                ("check_off_lane", C.c_int32), ("reward_collision", C.c_float),
                ("reward_time_exceed", C.c_float), ("reward_out_bound", C.c_float),
                ("reward_completed", C.c_float), ("time_penalty_scale", C.c_float),
                ("check_arrival", C.c_int32), ("check_no_action", C.c_int32),
                ("no_action_max_step", C.c_int32), ("shaped_reward", C.c_int32),
                ("arrival_threshold", C.c_float), ("no_action_iou", C.c_float),
                ("dist_reward_scale", C.c_float)]


class FrameLayout(C.Structure):
    """t2d_frame_layout (include/t2d.h): byte offsets of the sections of one host frame, -1 = absent."""
    _fields_ = [(k, C.c_int64) for k in ("bytes", "off_obs", "off_rel", "off_reward", "off_status", "off_iou", "off_frame_ms",
                                         "off_cnt_step", "off_episode", "off_target", "off_target_heading", "off_lidar")] + \
               [("n_env", C.c_int32), ("n_beams", C.c_int32)]


# every symbol include/t2d.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    "t2d_last_error": (C.c_char_p, [_vp]),
    "t2d_abi_version": (C.c_int, []),
    "t2d_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "t2d_destroy": (C.c_int, [_vp]),
    "t2d_set_param_table": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32]),
    "t2d_set_static_geometry": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "t2d_set_lane_geometry": (C.c_int, [_vp, _vp, _vp, _vp]),
    "t2d_set_status_config": (C.c_int, [_vp, C.POINTER(StatusConfig)]),
    "t2d_set_target_areas": (C.c_int, [_vp, _vp, _vp]),
    "t2d_reset": (C.c_int, [_vp] * 10),
    "t2d_bind_actions": (C.c_int, [_vp, _vp, _vp]),
    "t2d_bind_actions_strided": (C.c_int, [_vp, _vp, _vp, C.c_int32]),
    "t2d_set_action_extent": (C.c_int, [_vp, C.c_int64]),
    "t2d_integrate": (C.c_int, [_vp, C.c_int32, _vp]),
    "t2d_collide": (C.c_int, [_vp, _vp]),
    "t2d_check_status": (C.c_int, [_vp, C.c_int32, _vp]),
    "t2d_step": (C.c_int, [_vp, C.c_int32, _vp]),
    "t2d_step_n": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int64, _vp]),
    "t2d_set_step_chaining": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "t2d_set_split_step": (C.c_int, [_vp, C.c_int32]),
    "t2d_step_form": (C.c_int, [_vp, C.c_int32]),
    "t2d_set_fused_step": (C.c_int, [_vp, C.c_int32]),
    "t2d_set_ego_kernel": (C.c_int, [_vp, C.c_int32]),
    "t2d_step_groups": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32]),
    "t2d_get_field": (C.c_int, [_vp, C.c_int32, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "t2d_download": (C.c_int, [_vp, C.c_int32, _vp, C.c_size_t]),
    "t2d_upload": (C.c_int, [_vp, C.c_int32, _vp, C.c_size_t]),
    "t2d_sync": (C.c_int, [_vp]),
    "t2d_frame_config": (C.c_int, [_vp, C.c_uint32, C.c_int32, C.POINTER(FrameLayout)]),
    "t2d_set_target_headings": (C.c_int, [_vp, _vp]),
    "t2d_step_host": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, C.c_int32, C.POINTER(_vp)]),
    "t2d_frame_fetch": (C.c_int, [_vp, _vp, C.c_int32, C.POINTER(_vp)]),
    "t2d_host_action_buffer": (C.c_int, [_vp, C.POINTER(_vp)]),
    "t2d_snapshot": (C.c_int, [_vp]),
    "t2d_restore": (C.c_int, [_vp, C.c_int32, _vp]),
    "t2d_set_auto_reset": (C.c_int, [_vp, C.c_int32]),
    "t2d_lidar_config": (C.c_int, [_vp, C.c_int32, C.c_float, C.c_int32, _vp, _vp]),
    "t2d_lidar_scan": (C.c_int, [_vp, _vp, _vp]),
    "t2d_set_idm": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    "t2d_idm_actions": (C.c_int, [_vp, _vp, _vp]),
    "t2d_verify_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "t2d_generate_parking": (C.c_int, [C.c_int32, C.c_uint64, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_double]
                             + [_vp] * 8),
    "t2d_parking_scenes": (C.c_int, [_vp, C.c_uint64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int32]),
    "t2d_get_parking_scenes": (C.c_int, [_vp] * 10),
    "t2d_set_integrator_variant": (C.c_int, [_vp, C.c_int32]),
    "t2d_set_outputs": (C.c_int, [_vp, C.c_uint32]),
    "t2d_lane_safe_rects": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp]),
    "t2d_geometry_budget": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "t2d_profile_enable": (C.c_int, [_vp, C.c_int32]),
    "t2d_profile_read": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "t2d_comm_unique_id": (C.c_int, [_vp]),
    "t2d_comm_init": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32]),
    "t2d_gather": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp]),
    "t2d_gather_wait": (C.c_int, [_vp, _vp, C.c_int32]),
    "t2d_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "t2d_step_count": (C.c_int64, [_vp]),
    "t2d_step_occupancy": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "t2d_stream_create": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "t2d_stream_destroy": (C.c_int, [_vp]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7; if this
    library pulled in /opt/rocm's copy first, a later `import torch` would bring a second runtime
    that finds no GPU ("No HIP GPUs are available").  So when torch is installed but not imported
    yet, load ITS runtime first: libt2d_hip.so's DT_NEEDED libamdhip64.so.7 then resolves to it."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def _load(path, symbols, what):
    _share_torch_hip_runtime()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -m tactics2d_amd.build{what}` "
            "(hipcc, gfx950). tactics2d_amd has no CPU fallback.")
    handle = C.CDLL(path)
    for name, (res, args) in symbols.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            # (A/B measurements against a library built from older sources of the same ABI version: scripts/ab_step.py)
            if os.environ.get("T2D_ALLOW_MISSING_SYMBOLS"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    from . import layout
    have = handle.t2d_abi_version()
    if have != layout.ABI_VERSION:   # e.g. a stale .so whose record ring is sized differently
        raise ImportError(f"{path} has ABI version {have}, this package expects {layout.ABI_VERSION}: "
                          "rebuild it with `python -m tactics2d_amd.build --force`")
    return handle


def lib():
    """Load libt2d_hip.so (raises if it has not been built -- no fallback)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, SYMBOLS, "")
    return _lib


def check(rc, pool=None, library=None):
    if rc == OK:
        return
    msg = (library or lib()).t2d_last_error(pool)
    msg = msg.decode() if msg else ""
    raise (GeometryError if rc == ERR_GEOMETRY else T2DError)(rc, msg)
