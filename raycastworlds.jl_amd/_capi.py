"""ctypes binding of librcw_hip.so (include/rcw.h).

There is no CPU fallback: if the shared library is missing this module raises at load
time, and `rcw_create` itself fails on a machine without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "librcw_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "rcw.h")

RCW_OK = 0
RCW_ERR_INVALID_ARGUMENT = -1
RCW_ERR_INVALID_ACTION = -2
RCW_ERR_NO_DEVICE = -3
RCW_ERR_OUT_OF_MEMORY = -4
RCW_ERR_OUT_OF_BOUNDS = -5
RCW_ERR_HIP = -6
RCW_ERR_UNSUPPORTED = -7
RCW_WARN_SAMPLER_GAVE_UP = 1      # a per-agent status WARNING (utils.jl:34), not an error
RCW_ABI_VERSION = 4

# R of SingleRoom(; R = ...) SR:266 (include/rcw.h RCW_REWARD_*)
RCW_REWARD_FLOAT32, RCW_REWARD_FLOAT64, RCW_REWARD_INT32, RCW_REWARD_INT64 = 0, 1, 2, 3
RCW_GATHER_COLUMNS, RCW_GATHER_FRAMES = 0, 1
RCW_UNIQUE_ID_BYTES = 128
RCW_TOP_VIEW_NONE, RCW_TOP_VIEW_IN_PLACE, RCW_TOP_VIEW_ONE_KERNEL, RCW_TOP_VIEW_TWO_KERNELS = 0, 1, 2, 3
RCW_STEP_TWO_LAUNCHES, RCW_STEP_ONE_LAUNCH = 1, 2   # rcw_step_form / rcw_set_step_form


class RcwConfig(C.Structure):
    """`rcw_config` (include/rcw.h): kwargs of SingleRoom(; ...) SR:258-272 + colours SR:288-296."""

    _fields_ = [
        ("abi_version", C.c_int32),
        ("height_tile_map_tu", C.c_int32),
        ("width_tile_map_tu", C.c_int32),
        ("num_directions", C.c_int32),
        ("num_rays", C.c_int32),
        ("height_camera_view_pu", C.c_int32),
        ("pu_per_tu", C.c_int32),
        ("player_radius_wu", C.c_float),
        ("position_increment_wu", C.c_float),
        ("semi_field_of_view_wu", C.c_float),
        ("camera_height_tile_wu", C.c_float),
        ("goal_reward", C.c_float),
        ("floor_color", C.c_uint32),
        ("ceiling_color", C.c_uint32),
        ("wall_dim_1_color", C.c_uint32),
        ("wall_dim_2_color", C.c_uint32),
        ("goal_dim_1_color", C.c_uint32),
        ("goal_dim_2_color", C.c_uint32),
        ("dda_tie_break", C.c_int32),
        ("dda_distance", C.c_int32),
        ("normalize_mode", C.c_int32),
        ("auto_reset", C.c_int32),
        ("agent_id_offset", C.c_int64),
        ("reward_type", C.c_int32),
        ("out_of_bounds", C.c_int32),
        ("render_top_view", C.c_int32),
        ("world_unit_bits", C.c_int32),
        ("player_radius_wu_f64", C.c_double),
        ("position_increment_wu_f64", C.c_double),
        ("semi_field_of_view_wu_f64", C.c_double),
        ("camera_height_tile_wu_f64", C.c_double),
        ("goal_reward_f64", C.c_double),
        ("reserved", C.c_int32 * 2),
    ]


class RcwError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"librcw_hip error {code}: {message}")
        self.code = code
        self.message = message


_vp = C.c_void_p
_i32 = C.c_int32
_u64 = C.c_uint64

# name -> argtypes; every function returns int unless listed in _RESTYPE
SIGNATURES = {
    "rcw_abi_version": [],
    "rcw_last_error": [],
    "rcw_config_default": [C.POINTER(RcwConfig)],
    "rcw_create": [C.POINTER(RcwConfig), _i32, _i32, _u64, C.POINTER(_vp)],
    "rcw_destroy": [_vp],
    "rcw_set_direction_table": [_vp, _vp],
    "rcw_set_stream": [_vp, _vp],
    "rcw_get_stream": [_vp, C.POINTER(_vp)],
    "rcw_bind_obs": [_vp, _vp],
    "rcw_reset": [_vp, _vp, _u64],
    "rcw_set_state": [_vp, _vp, _vp, _vp, _vp],
    "rcw_set_state64": [_vp, _vp, _vp, _vp, _vp],
    "rcw_position64": [_vp, _vp],
    "rcw_rays64": [_vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "rcw_set_direction_table64": [_vp, _vp],
    "rcw_ray_table64": [_vp, _vp],
    "rcw_direction_table64": [_vp, _vp],
    "rcw_step": [_vp, _vp],
    "rcw_step_device": [_vp, _vp],
    "rcw_cast_rays": [_vp],
    "rcw_update_camera_view": [_vp],
    "rcw_update_top_view": [_vp],
    "rcw_sync": [_vp],
    "rcw_clear_error": [_vp],
    "rcw_obs_device_ptr": [_vp, C.POINTER(_vp)],
    "rcw_obs_copy": [_vp, _vp, _i32, _i32],
    "rcw_top_view_device_ptr": [_vp, C.POINTER(_vp)],
    "rcw_top_view_copy": [_vp, _vp, _i32, _i32],
    "rcw_reward": [_vp, _vp],
    "rcw_reward_typed": [_vp, _vp],
    "rcw_done": [_vp, _vp],
    "rcw_reward_device_ptr": [_vp, C.POINTER(_vp)],
    "rcw_done_device_ptr": [_vp, C.POINTER(_vp)],
    "rcw_position": [_vp, _vp],
    "rcw_direction": [_vp, _vp],
    "rcw_goal": [_vp, _vp],
    "rcw_episode": [_vp, _vp],
    "rcw_status": [_vp, _vp],
    "rcw_tile_map_num_chunks": [_vp, C.POINTER(_i32)],
    "rcw_tile_map_chunks": [_vp, _vp],
    "rcw_rays": [_vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "rcw_columns": [_vp, _i32, _i32, _vp, _vp],
    "rcw_columns_device_ptr": [_vp, C.POINTER(_vp), C.POINTER(_vp)],
    "rcw_expand_columns": [_vp, _vp, _vp, _i32, _vp],
    "rcw_ray_table": [_vp, _vp],
    "rcw_direction_table": [_vp, _vp],
    "rcw_timer_start": [_vp],
    "rcw_timer_stop": [_vp, C.POINTER(C.c_float)],
    "rcw_profile": [_vp, _i32],
    "rcw_profile_read": [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_i32)],
    "rcw_top_view_form": [_vp, C.POINTER(_i32)],
    "rcw_update_top_view_form": [_vp, C.POINTER(_i32)],
    "rcw_set_top_view_form": [_vp, _i32, _i32],
    "rcw_step_form": [_vp, C.POINTER(_i32)],
    "rcw_set_step_form": [_vp, _i32],
    "rcw_fill_kernel_name": [_vp, C.c_char_p, _i32],
    "rcw_comm_unique_id": [_vp],
    "rcw_comm_init": [_vp, _vp, _i32, _i32],
    "rcw_comm_destroy": [_vp],
    "rcw_comm_info": [_vp, C.POINTER(_i32), C.POINTER(_i32)],
    "rcw_gather_columns": [_vp, _vp, _vp],
    "rcw_gather_observations": [_vp, _i32, _vp],
    "rcw_device_malloc": [_vp, _u64, C.POINTER(_vp)],
    "rcw_device_free": [_vp, _vp],
    "rcw_memcpy_to_host": [_vp, _vp, _vp, _u64],
    "rcw_batch": [_vp, C.POINTER(_i32)],
    "rcw_get_config": [_vp, C.POINTER(RcwConfig)],
    "rcw_device_name": [_vp, C.c_char_p, _i32],
}
_RESTYPE = {"rcw_last_error": C.c_char_p}

_lib = None


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch's ROCm wheel bundles its own libamdhip64.so
    (soname libamdhip64.so.7, same as /opt/rocm's).  If librcw_hip pulled in the system copy
    first, a later `import torch` would load a second runtime that finds no GPU.  Loading
    torch's copy first (without importing torch) makes both resolve to the same one."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def preload_rccl() -> None:
    """One RCCL per process, for the same reason: PyTorch's wheel bundles its own librccl.so (soname
    librccl.so.1).  librcw_hip loads RCCL lazily with dlopen("librccl.so.1"), which returns whatever copy
    is resident; loading torch's first makes rcw_comm_* and torch.distributed share one instance."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


DEV_LIB_PATH = os.path.join(_PKG, "lib", "librcw_hip_dev.so")
_libs = {}


def load(library=None) -> C.CDLL:
    """Load librcw_hip.so — the shipped library, which reads no RCW_* environment variable but RCW_RCCL_LIBRARY; loud
    failure when it has not been built.  `library="dev"` (or a path) loads the development build instead
    (`make dev`: -DRCW_DEV_SWITCHES, the tuning knobs and the measured-and-rejected kernel variants); development
    tools may also point every default load at another build with RCW_LIBRARY=<path> — an explicit path, never a
    silent switch inside the library."""
    global _lib
    if library is None:
        library = os.environ.get("RCW_LIBRARY") or None
    path = LIB_PATH if library is None else (DEV_LIB_PATH if library == "dev" else os.path.abspath(library))
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python __graft_entry__.py build` "
            f"(or `make -C raycastworlds.jl_amd/csrc`). There is no CPU fallback."
        )
    _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    if lib.rcw_abi_version() != RCW_ABI_VERSION:
        raise ImportError(f"librcw_hip ABI {lib.rcw_abi_version()} != {RCW_ABI_VERSION}: rebuild with `make -C raycastworlds.jl_amd/csrc`")
    _libs[path] = lib
    if path == LIB_PATH or _lib is None:
        _lib = lib
    return lib


def last_error(lib=None) -> str:
    msg = (lib or load()).rcw_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, lib=None) -> None:
    if rc == RCW_OK:
        return
    msg = last_error(lib)
    if rc == RCW_ERR_INVALID_ACTION:
        # the reference raises AssertionError("Invalid action: ...") at SR:140
        raise AssertionError(msg or "Invalid action")
    if rc == RCW_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    if rc == RCW_ERR_OUT_OF_BOUNDS:
        raise IndexError(msg)   # Julia: BoundsError
    if rc == RCW_ERR_OUT_OF_MEMORY:
        raise MemoryError(msg)
    raise RcwError(rc, msg)


def default_config() -> RcwConfig:
    cfg = RcwConfig()
    check(load().rcw_config_default(C.byref(cfg)))
    return cfg
