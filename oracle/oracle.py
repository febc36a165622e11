"""ctypes front end of the CPU oracle (oracle/rcw_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of rcw_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package.
PARITY UNPINNED for cast_ray / normalize / LinRange (no Julia toolchain, no golden vectors
in the reference); pinned by tests/golden/ hand-derived vectors and oracle/pyref.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librcw_oracle.so")
_LIB64_PATH = os.path.join(_HERE, "_build", "librcw_oracle64.so")   # the same source with T = Float64
if os.environ.get("RCW_ORACLE_SANITIZED"):                           # tests/test_oracle_sanitizers.py: the ASan + UBSan build (make san)
    _LIB_PATH = os.path.join(_HERE, "_build", "san", "librcw_oracle.so")
    _LIB64_PATH = os.path.join(_HERE, "_build", "san", "librcw_oracle64.so")
if os.environ.get("RCW_ORACLE_COVERAGE"):                            # make cov: gcov-instrumented (which lines of the checker do its tests run?)
    _LIB_PATH = os.path.join(_HERE, "_build", "cov", "librcw_oracle.so")
    _LIB64_PATH = os.path.join(_HERE, "_build", "cov", "librcw_oracle64.so")


class RcwConfig(C.Structure):
    """Mirror of `rcw_config` (include/rcw.h); kwargs of SingleRoom(; ...) SR:258-272."""

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


def default_config(**overrides) -> RcwConfig:
    """Reference defaults SR:258-272, SR:288-296 (restated, not read from the product)."""
    cfg = RcwConfig()
    cfg.abi_version = 4
    cfg.height_tile_map_tu = 8
    cfg.width_tile_map_tu = 16
    cfg.num_directions = 128
    cfg.num_rays = 512
    cfg.height_camera_view_pu = 256
    cfg.pu_per_tu = 32
    cfg.player_radius_wu = np.float32(1 / 8)
    cfg.position_increment_wu = np.float32(1 / 8)
    cfg.semi_field_of_view_wu = np.float32(2 / 3)
    cfg.camera_height_tile_wu = 1.0
    cfg.goal_reward = 1.0
    cfg.floor_color = 0x00404040
    cfg.ceiling_color = 0x00FFFFFF
    cfg.wall_dim_1_color = 0x00808080
    cfg.wall_dim_2_color = 0x00C0C0C0
    cfg.goal_dim_1_color = 0x00800000
    cfg.goal_dim_2_color = 0x00C00000
    cfg.reward_type = 0          # R = Float32 SR:266
    cfg.goal_reward_f64 = 1.0    # one(R) SR:82
    cfg.world_unit_bits = 32
    cfg.player_radius_wu_f64 = 1 / 8
    cfg.position_increment_wu_f64 = 1 / 8
    cfg.semi_field_of_view_wu_f64 = 2 / 3
    cfg.camera_height_tile_wu_f64 = 1.0
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown config field {k!r}")
        setattr(cfg, k, v)
    return cfg


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, strict IEEE).  Building the checker is not using it."""
    src = os.path.join(_HERE, "rcw_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "rcw.h")
    newest = max(os.path.getmtime(src), os.path.getmtime(hdr))
    stale = force or any(not os.path.exists(p) or os.path.getmtime(p) < newest for p in (_LIB_PATH, _LIB64_PATH))
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"] + (["san"] if os.environ.get("RCW_ORACLE_SANITIZED") else ["cov"] if os.environ.get("RCW_ORACLE_COVERAGE") else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_libs = {}


def lib(bits: int = 32) -> C.CDLL:
    """The oracle for world-unit type Float32 (bits=32) or Float64 (bits=64)."""
    if bits not in _libs:
        build()
        L = C.CDLL(_LIB_PATH if bits == 32 else _LIB64_PATH)
        real = C.c_float if bits == 32 else C.c_double
        vp = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(RcwConfig), C.c_int32, C.c_uint64, C.c_int, C.POINTER(vp)]
        L.orc_create.restype = C.c_int
        L.orc_destroy.argtypes = [vp]
        L.orc_destroy.restype = None
        L.orc_reset.argtypes = [vp, vp, C.c_uint64]
        L.orc_set_state.argtypes = [vp, vp, vp, vp, vp]
        L.orc_set_direction_table.argtypes = [vp, vp]
        L.orc_step.argtypes = [vp, vp]
        L.orc_step_lenient.argtypes = [vp, vp]
        L.orc_clear_status.argtypes = [vp]
        L.orc_clear_status.restype = None
        L.orc_tile_map_chunks.argtypes = [vp, vp]
        L.orc_tile_map_chunks.restype = None
        L.orc_num_chunks.argtypes = [vp]
        L.orc_num_chunks.restype = C.c_int32
        L.orc_direction_table.argtypes = [C.c_int32, vp]
        L.orc_direction_table.restype = None
        L.orc_ray_fan.argtypes = [C.POINTER(RcwConfig), vp, vp]
        L.orc_ray_fan.restype = None
        L.orc_cast_ray.argtypes = [vp, C.c_int32, C.c_int32, real, real, real, real, C.c_int32, C.c_int32,
                                   vp, vp, vp, vp]
        L.orc_is_player_colliding.argtypes = [vp, C.c_int32, C.c_int32, real, real, real, C.c_int32]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads.restype = None
        for name in ("camera_view", "top_view", "reward", "done", "position", "direction", "goal", "episode",
                     "status", "ray_stop", "ray_dim", "ray_dist", "ray_dirs", "col_height",
                     "col_colour", "directions", "ray_table"):
            f = getattr(L, "orc_" + name)
            f.argtypes = [vp]
            f.restype = vp
        _libs[bits] = L
    return _libs[bits]


def _view(ptr, dtype, shape):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleBatch:
    """B independent reference worlds stepped on the CPU.  Arrays are returned with the
    BATCH AXIS FIRST (numpy C order of the Julia column-major layouts): camera_view is
    (B, N, H_cam) == Julia (H_cam, N, B)."""

    def __init__(self, batch: int, seed: int = 0, render: bool = True, config: RcwConfig | None = None,
                 **overrides):
        self.cfg = config if config is not None else default_config(**overrides)
        self.B = int(batch)
        self.bits = 64 if self.cfg.world_unit_bits == 64 else 32
        self.real = np.float64 if self.bits == 64 else np.float32   # the reference's T
        self._h = C.c_void_p()
        rc = lib(self.bits).orc_create(C.byref(self.cfg), self.B, seed, 1 if render else 0, C.byref(self._h))
        if rc != 0:
            raise ValueError(f"orc_create failed: {rc}")
        self.render = render
        self.N = self.cfg.num_rays
        self.Hc = self.cfg.height_camera_view_pu
        self.H = self.cfg.height_tile_map_tu
        self.W = self.cfg.width_tile_map_tu
        self.nd = self.cfg.num_directions

    def close(self):
        if self._h:
            lib(self.bits).orc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- RCW.reset! / act! ---------------------------------------------------------
    def reset(self, mask=None, seed: int = 0):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        rc = lib(self.bits).orc_reset(self._h, _p(m), seed)
        if rc:
            raise ValueError(rc)

    def set_state(self, goal_ij, position_wu, direction_au, mask=None):
        g = np.ascontiguousarray(goal_ij, dtype=np.int32).reshape(self.B, 2)
        p = np.ascontiguousarray(position_wu, dtype=self.real).reshape(self.B, 2)
        d = np.ascontiguousarray(direction_au, dtype=np.int32).reshape(self.B)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        rc = lib(self.bits).orc_set_state(self._h, _p(g), _p(p), _p(d), _p(m))
        if rc:
            raise ValueError(f"orc_set_state: {rc}")

    def set_direction_table(self, dirs):
        d = np.ascontiguousarray(dirs, dtype=self.real).reshape(self.nd, 2)
        lib(self.bits).orc_set_direction_table(self._h, _p(d))

    def step(self, actions) -> int:
        a = np.ascontiguousarray(actions, dtype=np.uint8).reshape(self.B)
        return lib(self.bits).orc_step(self._h, _p(a))

    def step_lenient(self, actions) -> int:
        """rcw_step_device semantics: agents with an invalid action are skipped, not rejected."""
        a = np.ascontiguousarray(actions, dtype=np.uint8).reshape(self.B)
        return lib(self.bits).orc_step_lenient(self._h, _p(a))

    def clear_status(self):
        lib(self.bits).orc_clear_status(self._h)

    # --- state ---------------------------------------------------------------------
    def _g(self, name, dtype, shape):
        return _view(getattr(lib(self.bits), "orc_" + name)(self._h), dtype, shape)

    @property
    def camera_view(self):
        assert self.render
        return self._g("camera_view", np.uint32, (self.B, self.N, self.Hc))

    @property
    def top_view(self):
        """(B, W*pu, H*pu) == Julia (H*pu, W*pu, B); needs render_top_view=1."""
        assert self.cfg.render_top_view
        pu = self.cfg.pu_per_tu
        return self._g("top_view", np.uint32, (self.B, self.W * pu, self.H * pu))

    @property
    def reward(self):
        """world.reward in the batch's R (cfg.reward_type: Float32 / Float64 / Int32 / Int64)."""
        dt = (np.float32, np.float64, np.int32, np.int64)[self.cfg.reward_type]
        return self._g("reward", dt, (self.B,))

    @property
    def done(self):
        return self._g("done", np.uint8, (self.B,))

    @property
    def position(self):
        return self._g("position", self.real, (self.B, 2))

    @property
    def direction(self):
        return self._g("direction", np.int32, (self.B,))

    @property
    def goal(self):
        return self._g("goal", np.int32, (self.B, 2))

    @property
    def episode(self):
        return self._g("episode", np.uint32, (self.B,))

    @property
    def status(self):
        return self._g("status", np.int32, (self.B,))

    @property
    def ray_stop(self):
        return self._g("ray_stop", np.int64, (self.B, self.N, 2))

    @property
    def ray_dim(self):
        return self._g("ray_dim", np.int64, (self.B, self.N))

    @property
    def ray_dist(self):
        return self._g("ray_dist", self.real, (self.B, self.N))

    @property
    def ray_dirs(self):
        return self._g("ray_dirs", self.real, (self.B, self.N, 2))

    @property
    def col_height(self):
        return self._g("col_height", np.int32, (self.B, self.N))

    @property
    def col_colour(self):
        return self._g("col_colour", np.uint8, (self.B, self.N))

    @property
    def directions(self):
        return self._g("directions", self.real, (self.nd, 2))

    @property
    def ray_table(self):
        """(nd, N, 2) normalized ray directions per heading."""
        return self._g("ray_table", self.real, (self.nd, self.N, 2))

    def tile_map_chunks(self):
        n = lib(self.bits).orc_num_chunks(self._h)
        out = np.zeros((self.B, n), dtype=np.uint64)
        lib(self.bits).orc_tile_map_chunks(self._h, _p(out))
        return out


def direction_table(nd: int, bits: int = 32) -> np.ndarray:
    out = np.zeros((nd, 2), dtype=np.float64 if bits == 64 else np.float32)
    lib(bits).orc_direction_table(nd, _p(out))
    return out


def ray_fan(cfg: RcwConfig, direction, bits: int = 32) -> np.ndarray:
    real = np.float64 if bits == 64 else np.float32
    d = np.ascontiguousarray(direction, dtype=real)
    out = np.zeros((cfg.num_rays, 2), dtype=real)
    lib(bits).orc_ray_fan(C.byref(cfg), _p(d), _p(out))
    return out


def cast_ray(obstacle_map, x, y, dx, dy, tie_break=0, dist_mode=0, bits: int = 32):
    """obstacle_map: bool array indexed [i-1, j-1] (shape (H, W))."""
    om = np.asfortranarray(np.asarray(obstacle_map, dtype=np.uint8))
    H, W = om.shape
    i = C.c_int64(); j = C.c_int64(); dim = C.c_int64(); dist = C.c_double() if bits == 64 else C.c_float()
    rc = lib(bits).orc_cast_ray(om.ctypes.data_as(C.c_void_p), H, W, x, y, dx, dy, tie_break, dist_mode,
                            C.addressof(i), C.addressof(j), C.addressof(dim), C.addressof(dist))
    if rc:
        raise IndexError("BoundsError")
    return i.value, j.value, dim.value, (np.float64 if bits == 64 else np.float32)(dist.value)


def is_player_colliding(layer, px, py, radius, oob_empty=False, bits: int = 32) -> bool:
    lm = np.asfortranarray(np.asarray(layer, dtype=np.uint8))
    H, W = lm.shape
    rc = lib(bits).orc_is_player_colliding(lm.ctypes.data_as(C.c_void_p), H, W, px, py, radius,
                                       1 if oob_empty else 0)
    if rc < 0:
        raise IndexError("BoundsError")
    return bool(rc)


def set_num_threads(n: int):
    lib().orc_set_num_threads(n)


def num_threads() -> int:
    return lib().orc_num_threads()
