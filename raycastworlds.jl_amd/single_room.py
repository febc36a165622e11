"""Host-side mirror of RayCastWorlds.jl's SingleRoomModule for a BATCH of agents.

The reference is Julia (src/single_room.jl, "SR" below); no Julia toolchain exists in this
pipeline, so this Python layer plays the role of the Julia host code above the C ABI
(include/rcw.h) with the reference's names and argument meanings:

    reference (Julia)                       here (Python)
    SingleRoom(; kwargs...)      SR:258     SingleRoom(batch, **same kwargs)
    RCW.reset!(env)              SR:326     reset_(env)
    RCW.act!(env, action)        SR:333     act_(env, actions)        # one action per agent
    RCW.get_action_names(env)    SR:486     get_action_names(env)
    RCW.RLBaseEnv(env)           rlbase.jl  RLBaseEnv(env)            # see rlbase.py
    env.world.reward / .done ... SR:21-40   env.world.reward / .done ...

All compute happens in librcw_hip.so on the GPU; nothing here falls back to the CPU.
Arrays come back batch-first in numpy C order, which is byte-identical to the Julia
column-major layout with a trailing batch axis: camera_view (B, N, H_cam) == Julia
UInt32 (H_cam, N, B).
"""
from __future__ import annotations

import ctypes as C
import sys
from typing import Optional, Sequence

import numpy as np

from . import _capi

NUM_OBJECTS = 2   # SR:16
WALL = 1          # SR:17
GOAL = 2          # SR:18
NUM_ACTIONS = 4   # SR:19

NUM_VIEWS = 2     # SR:237
CAMERA_VIEW = 1   # SR:238
TOP_VIEW = 2      # SR:239


class _Handle:
    """Owner of the native `rcw_handle`: whoever holds one keeps the engine's device memory alive.

    It refers to nothing but the library, so it can be held from places Python's collector cannot see (the deleter of a
    torch tensor made over engine memory, `_CudaExport` below) without tying the environment into an uncollectable
    cycle: environment -> DeviceArray -> tensor -> export -> _Handle, and nothing leads back."""

    __slots__ = ("lib", "h")

    def __init__(self, lib):
        self.lib = lib
        self.h = C.c_void_p()

    def close(self):
        if self.h:
            self.lib.rcw_destroy(self.h)         # waits for the handle's stream
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:                        # noqa: BLE001 — interpreter shutdown
            pass


class _CudaExport:
    """What `DeviceArray.torch()` hands to `torch.as_tensor`: pointer, shape and type of the device memory plus the
    `_Handle` that keeps it allocated.  torch takes a reference to the exporting object inside the tensor's storage
    deleter — a reference invisible to Python's collector — so the exporter must not lead back to whoever caches the
    tensor (exporting the DeviceArray itself leaked every environment whose reward / done / state
    tensor had been made)."""

    __slots__ = ("__cuda_array_interface__", "_handle")

    def __init__(self, interface: dict, handle):
        self.__cuda_array_interface__ = interface
        self._handle = handle


class DeviceArray(np.lib.mixins.NDArrayOperatorsMixin):
    """A typed view of library-owned device memory (no copy).

    Exposes `__cuda_array_interface__` (v3), so `torch.as_tensor(x, device="cuda")` and
    friends alias it exactly as `RLBase.state(env)` aliases `camera_view` (SR:576).

    A GPU-resident consumer takes `x.torch(sync=False)` — one cached tensor over the stable device pointer, ordered on
    the engine's stream, no host synchronisation ever.  A host consumer just uses it as an array: `np.asarray(x)`,
    `x == 0`, `x.any()`, `total += x` copy the current contents to the host at that moment (one device-to-host copy
    behind the engine's stream — the explicit request; a sticky device error such as the reference's BoundsError
    surfaces there).
    """

    def __init__(self, ptr: int, shape: Sequence[int], dtype, owner, sync, host_getter=None):
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._owner = owner   # keeps the handle alive
        self._sync = sync
        self._host_getter = host_getter
        self._torch = None

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape)) * self.dtype.itemsize

    @property
    def __cuda_array_interface__(self):
        # One host synchronisation when the alias is made (what a casual `torch.as_tensor(env.camera_view)`
        # needs to read valid data).  After that the alias is NOT re-synchronised: torch work on a stream other
        # than the engine's must `wait_stream(env.torch_stream())` (or share one stream via `env.set_stream`)
        # before reading frames a later step writes — INTEGRATION.md "Streams".
        self._sync()
        return self._interface()

    def _interface(self) -> dict:
        # (device memory behind a Bool view is one byte per element: exported as uint8, viewed as bool by .torch())
        return {
            "shape": self.shape,
            "typestr": "|u1" if self.dtype == np.bool_ else self.dtype.str,
            "data": (self.ptr, False),
            "version": 3,
            "strides": None,
        }

    @staticmethod
    def _tensor_over(export, device: int):
        import torch

        return torch.as_tensor(export, device=f"cuda:{device}")

    def torch(self, sync: bool = True):
        """Zero-copy torch tensor on the handle's device (made once per DeviceArray, the pointer is stable).
        `sync=False` skips the host synchronisation: for callers that order their work on the engine's stream
        themselves (a policy on the shared stream, the observation gather)."""
        if sync:
            self._sync()
        if self._torch is None:
            # exported through a proxy that holds the native handle, not this object: see _CudaExport
            t = self._tensor_over(_CudaExport(self._interface(), getattr(self._owner, "_handle", None)), self._owner.device)
            if self.dtype == np.bool_:
                import torch

                t = t.view(torch.bool)
            self._torch = t
        return self._torch

    # DLPack (the array-API exchange protocol: `jax.dlpack.from_dlpack(x)`, `cupy.from_dlpack(x)`, `torch.from_dlpack(x)`):
    # the same zero-copy alias, handed over by the cached torch tensor.  The consumer's stream is ordered behind the
    # engine's by one host synchronisation at export, as with __cuda_array_interface__.
    def __dlpack__(self, stream=None, **kwargs):
        t = self.torch(sync=True)
        return t.__dlpack__(**kwargs) if stream is None else t.__dlpack__(stream=stream, **kwargs)

    def __dlpack_device__(self):
        return self.torch(sync=False).__dlpack_device__()

    def numpy(self) -> np.ndarray:
        """Copy to host (waits for the engine's stream)."""
        if self._host_getter is not None:
            return self._host_getter()
        return self._owner._copy_device_array(self)

    # ---- host-side use as an array: every operation below starts with one device-to-host copy --------------
    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        if out is not None:
            if any(isinstance(o, DeviceArray) for o in out):
                # `x += 1`, `np.add(a, b, out=x)`: engine memory is the engine's to write (the next step overwrites it)
                raise TypeError("a DeviceArray is read-only from the host: operate on np.asarray(x) or x.torch()")
            kwargs["out"] = out
        inputs = tuple(np.asarray(x) if isinstance(x, DeviceArray) else x for x in inputs)
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __bool__(self):
        """As ndarray: the value of a one-element array, ValueError for more (`if is_terminated(env):` with a batch)."""
        return bool(self.numpy())

    def __iter__(self):
        """One device-to-host copy, then the host array's iterator (not one copy per element)."""
        return iter(self.numpy())

    def any(self, *a, **k):
        return self.numpy().any(*a, **k)

    def all(self, *a, **k):
        return self.numpy().all(*a, **k)

    def sum(self, *a, **k):
        return self.numpy().sum(*a, **k)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __repr__(self):
        return f"DeviceArray(ptr=0x{self.ptr:x}, shape={self.shape}, dtype={self.dtype})"


def _torch_owns(stream) -> bool:
    """True for a stream out of torch's own pool (`torch.cuda.Stream()`, the current / default stream): torch never
    destroys those.  `torch.cuda.ExternalStream` is a Stream subclass with the same `.cuda_stream` attribute, but wraps
    a hipStream_t its creator may destroy at any time — anything that is not provably torch's is the caller's."""
    torch = sys.modules.get("torch")
    if torch is None:
        return False
    return isinstance(stream, torch.cuda.Stream) and not isinstance(stream, torch.cuda.ExternalStream)


def _as_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SingleRoomWorld:
    """Accessors for the batched `SingleRoomWorld` state (SR:21-40). Each property copies
    the current device state to host."""

    def __init__(self, env: "SingleRoom"):
        self._env = env

    def _get(self, fn, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        self._env.host_syncs += 1                      # every host getter waits for the engine's stream
        self._env._check(fn(self._env._h, _as_ptr(out)))
        return out

    @property
    def num_directions(self) -> int:
        return self._env.cfg.num_directions

    @property
    def num_rays(self) -> int:
        return self._env.cfg.num_rays

    @property
    def player_radius_wu(self) -> float:
        return self._env.cfg.player_radius_wu

    @property
    def position_increment_wu(self) -> float:
        return self._env.cfg.position_increment_wu

    @property
    def semi_field_of_view_wu(self) -> float:
        return self._env.cfg.semi_field_of_view_wu

    @property
    def goal_reward(self):                             # SR:34-35, one(R) SR:82
        return self._env.R(1)

    @property
    def reward(self) -> np.ndarray:                    # SR:33, in R
        return self._get(self._env._lib.rcw_reward_typed, self._env.R, (self._env.batch,))

    @property
    def done(self) -> np.ndarray:                      # SR:35
        return self._get(self._env._lib.rcw_done, np.uint8, (self._env.batch,)).astype(bool)

    @property
    def player_position_wu(self) -> np.ndarray:        # SR:24, (B, 2) in T
        if self._env.T is np.float64:
            return self._get(self._env._lib.rcw_position64, np.float64, (self._env.batch, 2))
        return self._get(self._env._lib.rcw_position, np.float32, (self._env.batch, 2))

    @property
    def player_direction_au(self) -> np.ndarray:       # SR:25
        return self._get(self._env._lib.rcw_direction, np.int32, (self._env.batch,))

    @property
    def goal_position(self) -> np.ndarray:             # SR:32, (B, 2) 1-based (i, j)
        return self._get(self._env._lib.rcw_goal, np.int32, (self._env.batch, 2))

    @property
    def status(self) -> np.ndarray:
        """Per-agent sticky status: 0, -5 where the reference would have raised BoundsError, -2 for an invalid device action, or the
        warning 1 (`RCW_WARN_SAMPLER_GAVE_UP`) where `sample_empty_position` found no empty tile and gave up (utils.jl:34)."""
        return self._get(self._env._lib.rcw_status, np.int32, (self._env.batch,))

    @property
    def episode(self) -> np.ndarray:
        return self._get(self._env._lib.rcw_episode, np.uint32, (self._env.batch,))

    @property
    def directions_wu(self) -> np.ndarray:             # SR:28, (nd, 2)
        out = np.empty((self.num_directions, 2), dtype=self._env.T)
        fn = self._env._lib.rcw_direction_table64 if self._env.T is np.float64 else self._env._lib.rcw_direction_table
        self._env._check(fn(self._env._h, _as_ptr(out)))
        return out

    @property
    def tile_map_chunks(self) -> np.ndarray:
        """BitArray{3}(2, H, W).chunks per agent: uint64 (B, nchunks)  SR:22."""
        n = C.c_int32()
        self._env._check(self._env._lib.rcw_tile_map_num_chunks(self._env._h, C.byref(n)))
        return self._get(self._env._lib.rcw_tile_map_chunks, np.uint64, (self._env.batch, n.value))

    @property
    def tile_map(self) -> np.ndarray:
        """bool (B, 2, H, W): tile_map[b, o-1, i-1, j-1] == Julia tile_map[o, i, j] of agent b."""
        env = self._env
        return unpack_tile_map(self.tile_map_chunks, env.cfg.height_tile_map_tu, env.cfg.width_tile_map_tu)

    def rays(self, first: int = 0, count: Optional[int] = None):
        """(ray_stop_position_tu (n, N, 2) int64, ray_hit_dimension (n, N) int64,
        ray_distance_wu (n, N) float32, ray_directions_wu (n, N, 2) float32)  SR:29-31,39."""
        env = self._env
        n = env.batch - first if count is None else count
        N = env.cfg.num_rays
        stop = np.empty((n, N, 2), dtype=np.int64)
        dim = np.empty((n, N), dtype=np.int64)
        dist = np.empty((n, N), dtype=env.T)
        dirs = np.empty((n, N, 2), dtype=env.T)
        fn = env._lib.rcw_rays64 if env.T is np.float64 else env._lib.rcw_rays
        env._check(fn(env._h, first, n, _as_ptr(stop), _as_ptr(dim), _as_ptr(dist), _as_ptr(dirs)))
        return stop, dim, dist, dirs


class SingleRoom:
    """`SingleRoom(; kwargs...)` (SR:258-324) for `batch` independent agents on one MI355X.

    Keyword arguments and defaults are the reference's (SR:258-272). `T` is "Float32" (default)
    or "Float64"; `R` (the reward type, SR:266) is "Float32" (default), "Float64", "Int32" or "Int64";
    `seed` keys the device generator (the default: resets are sampled on the GPU, reproducibly, in the reference's
    distribution).  `rng` (SR:49,265) gives the reference's keyword back: a `numpy.random.Generator` — or one per agent —
    from which construction and every `reset_(env, rng=...)` draw goal, player tile and heading on the HOST in exactly
    the reference's order (`reference_reset_draws`), injected with `rcw_set_state`.  `device` is the HIP device index.
    """

    def __init__(
        self,
        batch: int = 1,
        *,
        T="Float32",
        height_tile_map_tu: int = 8,
        width_tile_map_tu: int = 16,
        num_directions: int = 128,
        player_radius_wu: float = 1 / 8,
        position_increment_wu: float = 1 / 8,
        seed: int = 0,
        rng=None,
        R="Float32",
        semi_field_of_view_wu: float = 2 / 3,
        num_rays: int = 512,
        pu_per_tu: int = 32,
        camera_height_tile_wu: float = 1.0,
        height_camera_view_pu: int = 256,
        device: int = 0,
        auto_reset: bool = False,
        agent_id_offset: int = 0,
        dda_tie_break: int = 0,
        dda_distance: int = 0,
        normalize_mode: int = 0,
        out_of_bounds: int = 0,
        render_top_view: bool = False,
        library: Optional[str] = None,
    ):
        f32_names = ("Float32", "float32", "<class 'numpy.float32'>")
        f64_names = ("Float64", "float64", "<class 'numpy.float64'>", "<class 'float'>")
        if str(T) not in f32_names + f64_names:
            raise NotImplementedError(f"world-unit type T = {T!r}: only Float32 and Float64 are built")
        r_names = {**{n: (np.float32, _capi.RCW_REWARD_FLOAT32) for n in f32_names},
                   **{n: (np.float64, _capi.RCW_REWARD_FLOAT64) for n in f64_names},
                   **{n: (np.int32, _capi.RCW_REWARD_INT32) for n in ("Int32", "int32", "<class 'numpy.int32'>")},
                   **{n: (np.int64, _capi.RCW_REWARD_INT64) for n in ("Int64", "Int", "int64", "<class 'numpy.int64'>",
                                                                     "<class 'int'>")}}
        if str(R) not in r_names:
            raise NotImplementedError(f"reward type R = {R!r}: Float32, Float64, Int32 and Int64 are built")
        self.T = np.float64 if str(T) in f64_names else np.float32
        self.R, reward_type = r_names[str(R)]
        # `library`: None = the shipped librcw_hip.so; "dev" (or a path) = the development build, which also reads the
        # RCW_* tuning knobs and carries the measured-and-rejected kernel variants (csrc/Makefile `dev`)
        self._lib = _capi.load(library)
        cfg = _capi.RcwConfig()
        _capi.check(self._lib.rcw_config_default(C.byref(cfg)), self._lib)
        cfg.height_tile_map_tu = height_tile_map_tu
        cfg.width_tile_map_tu = width_tile_map_tu
        cfg.num_directions = num_directions
        cfg.player_radius_wu = np.float32(player_radius_wu)          # convert(T, .) SR:263
        cfg.position_increment_wu = np.float32(position_increment_wu)
        cfg.semi_field_of_view_wu = np.float32(semi_field_of_view_wu)
        cfg.num_rays = num_rays
        cfg.pu_per_tu = pu_per_tu
        cfg.camera_height_tile_wu = np.float32(camera_height_tile_wu)
        cfg.height_camera_view_pu = height_camera_view_pu
        cfg.auto_reset = 1 if auto_reset else 0
        cfg.agent_id_offset = agent_id_offset
        cfg.dda_tie_break = dda_tie_break
        cfg.dda_distance = dda_distance
        cfg.normalize_mode = normalize_mode
        cfg.out_of_bounds = out_of_bounds
        cfg.render_top_view = 1 if render_top_view else 0
        cfg.reward_type = reward_type                 # R, with goal_reward = one(R) SR:82
        cfg.goal_reward = 1.0
        cfg.goal_reward_f64 = 1.0
        # convert(T, .) of the caller's values (SR:263-270): for T = Float64 the Float64 value itself
        cfg.world_unit_bits = 64 if self.T is np.float64 else 32
        cfg.player_radius_wu_f64 = float(player_radius_wu)
        cfg.position_increment_wu_f64 = float(position_increment_wu)
        cfg.semi_field_of_view_wu_f64 = float(semi_field_of_view_wu)
        cfg.camera_height_tile_wu_f64 = float(camera_height_tile_wu)
        self.cfg = cfg
        self.batch = int(batch)
        self.device = int(device)
        self.seed = int(seed)
        self._handle = _Handle(self._lib)    # owns the native handle; `self._h` reads it
        self._check(self._lib.rcw_create(C.byref(cfg), self.batch, self.device, self.seed, C.byref(self._handle.h)))
        self.world = SingleRoomWorld(self)
        self._held = []          # (event, tensors): torch tensors a NON-torch stream may still be reading (_release_after_use)
        self._free_events = []
        self._torch_owned_stream = None
        # In a process that already uses torch the engine runs on a stream TORCH owns (never destroyed, so
        # `Tensor.record_stream` on it is safe for tensors that outlive the engine).  A caller that has not imported torch
        # does not get it imported here and keeps the library's own stream (`stream_ptr()`).
        torch = sys.modules.get("torch")
        if torch is not None:
            try:
                if torch.cuda.is_available():
                    self.set_stream(torch.cuda.Stream(device=self.device))
            except BaseException:
                self._handle.close()             # nothing of a half-made environment stays allocated
                raise
        self.host_syncs = 0      # host synchronisations / device-to-host getters issued through this object (bench.py --api rlbase)
        self.rng = rng
        if rng is not None:
            # the reference's constructor consumes its rng TWICE: the draws of SR:62-74, then reset!(world) SR:105 -> SR:120-128
            try:
                _reset_from_rng(self, rng, None, construction=True)
            except BaseException:
                self._handle.close()
                raise
        # colour fields of the reference struct SR:241-256
        self.floor_color = cfg.floor_color
        self.ceiling_color = cfg.ceiling_color
        self.wall_dim_1_color = cfg.wall_dim_1_color
        self.wall_dim_2_color = cfg.wall_dim_2_color
        self.goal_dim_1_color = cfg.goal_dim_1_color
        self.goal_dim_2_color = cfg.goal_dim_2_color
        self.camera_height_tile_wu = cfg.camera_height_tile_wu
        self.height_camera_view_pu = cfg.height_camera_view_pu

    def _check(self, rc: int) -> None:
        _capi.check(rc, self._lib)

    @property
    def _h(self):
        """The native `rcw_handle*` (NULL once closed)."""
        return self._handle.h

    # ---- lifetime -------------------------------------------------------------------
    def close(self):
        """`rcw_destroy`.  Device aliases made earlier (`camera_view`, `reward(env).torch()`, ...) dangle from here on,
        as any view of freed memory does; without an explicit close the memory lives as long as the last of them."""
        handle = getattr(self, "_handle", None)
        if handle is not None:
            handle.close()
        self._held = []
        for name in ("_reward_dev", "_done_dev", "_done_dev_bool", "_state_alias", "_constant_action_buffers"):
            self.__dict__.pop(name, None)

    # (no __del__: an environment that is dropped without close() lets go of its _Handle, and the engine is destroyed
    # when the last holder does — this object, or the last torch tensor made over engine memory, whichever lives longer)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- device views ---------------------------------------------------------------
    def _sync(self):
        self.host_syncs += 1
        self._check(self._lib.rcw_sync(self._h))

    def sync(self):
        self._sync()

    def _copy_device_array(self, arr: DeviceArray) -> np.ndarray:
        if arr.ptr == self._obs_ptr():
            out = np.empty(arr.shape, dtype=arr.dtype)
            self._check(self._lib.rcw_obs_copy(self._h, _as_ptr(out), 0, self.batch))
            return out
        import torch

        return torch.as_tensor(arr, device=f"cuda:{self.device}").cpu().numpy()

    def _obs_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self._lib.rcw_obs_device_ptr(self._h, C.byref(p)))
        return int(p.value)

    @property
    def camera_view(self) -> DeviceArray:
        """The observation batch, aliased (SR:300, SR:576): uint32 (B, N, H_cam) in C order."""
        return DeviceArray(self._obs_ptr(), (self.batch, self.cfg.num_rays, self.cfg.height_camera_view_pu),
                           np.uint32, self, self._sync)

    @property
    def top_view(self) -> DeviceArray:
        """`env.top_view` (SR:302, needs render_top_view=True): uint32 (B, W*pu, H*pu) in C order ==
        Julia (H*pu, W*pu, B)."""
        p = C.c_void_p()
        self._check(self._lib.rcw_top_view_device_ptr(self._h, C.byref(p)))
        pu = self.cfg.pu_per_tu
        return DeviceArray(p.value, (self.batch, self.cfg.width_tile_map_tu * pu, self.cfg.height_tile_map_tu * pu),
                           np.uint32, self, self._sync)

    def top_view_host(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        n = self.batch - first if count is None else count
        pu = self.cfg.pu_per_tu
        out = np.empty((n, self.cfg.width_tile_map_tu * pu, self.cfg.height_tile_map_tu * pu), dtype=np.uint32)
        self._check(self._lib.rcw_top_view_copy(self._h, _as_ptr(out), first, n))
        return out

    def camera_view_host(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        n = self.batch - first if count is None else count
        out = np.empty((n, self.cfg.num_rays, self.cfg.height_camera_view_pu), dtype=np.uint32)
        self._check(self._lib.rcw_obs_copy(self._h, _as_ptr(out), first, n))
        return out

    def columns(self, first: int = 0, count: Optional[int] = None):
        """Compact per-column descriptors (height_line_pu int32 (n, N), colour id uint8 (n, N))."""
        n = self.batch - first if count is None else count
        h = np.empty((n, self.cfg.num_rays), dtype=np.int32)
        c = np.empty((n, self.cfg.num_rays), dtype=np.uint8)
        self._check(self._lib.rcw_columns(self._h, first, n, _as_ptr(h), _as_ptr(c)))
        return h, c

    def columns_device(self):
        hp, cp = C.c_void_p(), C.c_void_p()
        self._check(self._lib.rcw_columns_device_ptr(self._h, C.byref(hp), C.byref(cp)))
        shape = (self.batch, self.cfg.num_rays)
        return (DeviceArray(hp.value, shape, np.int32, self, self._sync),
                DeviceArray(cp.value, shape, np.uint8, self, self._sync))

    def expand_columns(self, height_line_pu, colour_id, out=None):
        """Descriptors (CUDA tensors (n, N) int32 / uint8, e.g. gathered from other GPUs) ->
        frames (n, N, H_cam) as a torch CUDA tensor, with this engine's colours (rcw_expand_columns).
        Stream-ordered on the engine's stream behind the producer of the descriptors (torch's current
        stream); no host synchronisation.  Consumers on another stream wait for `env.torch_stream()`."""
        import torch

        n = int(height_line_pu.shape[0])
        if tuple(height_line_pu.shape) != (n, self.cfg.num_rays) or tuple(colour_id.shape) != (n, self.cfg.num_rays):
            raise ValueError("descriptor shape must be (n, num_rays)")
        if height_line_pu.dtype != torch.int32 or colour_id.dtype != torch.uint8:
            raise ValueError("descriptors must be int32 / uint8")
        h = height_line_pu.contiguous()
        c = colour_id.contiguous()
        if out is None:
            out = torch.empty((n, self.cfg.num_rays, self.cfg.height_camera_view_pu), dtype=torch.uint32,
                              device=f"cuda:{self.device}")
        cross = self._order_behind_torch()
        self._check(self._lib.rcw_expand_columns(self._h, C.c_void_p(h.data_ptr()), C.c_void_p(c.data_ptr()), n,
                                                 C.c_void_p(out.data_ptr())))
        if cross:
            self._release_after_use(h, c, out)
        return out

    def _constant_actions(self, action: int):
        """A device array holding `action` for every agent (None without torch: the host path is taken)."""
        cache = self.__dict__.setdefault("_constant_action_buffers", {})
        buf = cache.get(action)
        if buf is None and action not in cache:
            try:
                import torch

                buf = torch.full((self.batch,), action, dtype=torch.uint8, device=f"cuda:{self.device}")
                torch.cuda.synchronize(self.device)      # (made on torch's stream, used on the engine's from now on)
            except Exception:                            # noqa: BLE001 — no torch / no CUDA torch: stay on the host path
                buf = None
            cache[action] = buf
        return buf

    def _order_behind_torch(self, *tensors):
        """Make the engine's stream wait for torch's current stream (the producer of `tensors`), and keep the
        tensors alive until the engine has consumed them: `_release_after_use` must follow the launch.  Without
        the hold, a tensor the caller drops right after the call goes back to torch's caching allocator, which
        may hand the block out again — to be overwritten on torch's stream — before the engine has read it."""
        import torch

        producer = torch.cuda.current_stream(self.device)
        if producer.cuda_stream == self.stream_ptr():
            return False
        self.torch_stream().wait_stream(producer)
        return True

    def _release_after_use(self, *tensors):
        """Keep `tensors` (handed to the engine on its stream) from being recycled by torch's caching allocator before the
        engine has used them, although the caller may drop them right after the call.

        On a stream torch owns (the default: see __init__) this is `Tensor.record_stream` — no bookkeeping, never blocks.
        On any other stream (the library's own, a caller's raw hipStream_t) record_stream must NOT be used: the
        allocator records an event on every recorded stream when the tensor is finally freed, which may be long after
        `close()` has destroyed that stream — an event record on a destroyed HIP stream is a use after free inside the
        runtime (the segmentation fault of rounds 2 and 3).  There, references are held instead until an
        event recorded now on the engine's stream has completed (a ring of 16 reusable events; the host only ever blocks
        when it runs 16 launches ahead of the GPU)."""
        import torch

        if self._stream_is_torch_owned():
            ts = self.torch_stream()
            for t in tensors:
                t.record_stream(ts)
            return
        ring, free = self._held, self._free_events
        while ring and ring[0][0].query():
            free.append(ring.pop(0)[0])
        if len(ring) >= 16:
            ev, _ = ring.pop(0)
            ev.synchronize()
            free.append(ev)
        ev = free.pop() if free else torch.cuda.Event()
        ev.record(self.torch_stream())
        ring.append((ev, tensors))

    def reward_device(self) -> DeviceArray:
        """`world.reward` (SR:33) where it lives: R (B,) in device memory, stable for the handle's lifetime, rewritten
        by every step in stream order.  The same object on every call (its torch alias is made once)."""
        if getattr(self, "_reward_dev", None) is None:
            p = C.c_void_p()
            self._check(self._lib.rcw_reward_device_ptr(self._h, C.byref(p)))
            self._reward_dev = DeviceArray(p.value, (self.batch,), self.R, self, self._sync,
                                           host_getter=lambda: self.world.reward)
        return self._reward_dev

    def done_device(self, as_bool: bool = False) -> DeviceArray:
        """`world.done` (SR:35) in device memory: one byte per agent (`as_bool`: the same bytes viewed as Bool)."""
        name = "_done_dev_bool" if as_bool else "_done_dev"
        if getattr(self, name, None) is None:
            p = C.c_void_p()
            self._check(self._lib.rcw_done_device_ptr(self._h, C.byref(p)))
            getter = (lambda: self.world.done) if as_bool else (lambda: self.world.done.astype(np.uint8))
            setattr(self, name, DeviceArray(p.value, (self.batch,), np.bool_ if as_bool else np.uint8, self, self._sync,
                                            host_getter=getter))
        return getattr(self, name)

    def ray_table(self) -> np.ndarray:
        """(nd, 5, N) float32: per heading [dx | dy | |1/dx| | |1/dy| | dir·ray]."""
        out = np.empty((self.cfg.num_directions, 5, self.cfg.num_rays), dtype=self.T)
        fn = self._lib.rcw_ray_table64 if self.T is np.float64 else self._lib.rcw_ray_table
        self._check(fn(self._h, _as_ptr(out)))
        return out

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._check(self._lib.rcw_device_name(self._h, buf, 256))
        return buf.value.decode()

    # ---- state injection (how "identical seeds" is realised) ---------
    def set_state(self, goal_position, player_position_wu, player_direction_au, mask=None):
        g = np.ascontiguousarray(goal_position, dtype=np.int32).reshape(self.batch, 2)
        p = np.ascontiguousarray(player_position_wu, dtype=self.T).reshape(self.batch, 2)
        d = np.ascontiguousarray(player_direction_au, dtype=np.int32).reshape(self.batch)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.batch)
        fn = self._lib.rcw_set_state64 if self.T is np.float64 else self._lib.rcw_set_state
        self._check(fn(self._h, _as_ptr(g), _as_ptr(p), _as_ptr(d), _as_ptr(m)))

    def set_direction_table(self, directions_wu):
        d = np.ascontiguousarray(directions_wu, dtype=self.T).reshape(self.cfg.num_directions, 2)
        fn = self._lib.rcw_set_direction_table64 if self.T is np.float64 else self._lib.rcw_set_direction_table
        self._check(fn(self._h, _as_ptr(d)))

    def stream_ptr(self) -> int:
        """The hipStream_t (as an int) the engine's work is ordered on."""
        p = C.c_void_p()
        self._check(self._lib.rcw_get_stream(self._h, C.byref(p)))
        return int(p.value or 0)

    def _stream_is_torch_owned(self) -> bool:
        ts = getattr(self, "_torch_owned_stream", None)
        return ts is not None and ts.cuda_stream == self.stream_ptr()

    def torch_stream(self):
        """The engine's stream as a torch stream object (for wait_stream / wait_event / record_stream): the torch Stream
        itself where torch owns it, else ONE `ExternalStream` wrapper per raw stream, kept for the engine's lifetime."""
        import torch

        if self._stream_is_torch_owned():
            return self._torch_owned_stream
        ptr = self.stream_ptr()
        cached = getattr(self, "_torch_stream", None)
        if cached is None or cached[0] != ptr:
            cached = self._torch_stream = (ptr, torch.cuda.ExternalStream(ptr, device=f"cuda:{self.device}"))
        return cached[1]

    def set_stream(self, stream):
        """Order the engine's work on another stream: a `torch.cuda.Stream` (kept alive here; torch never destroys its
        streams, so tensors may be handed over with record_stream); a `torch.cuda.ExternalStream` or a raw hipStream_t as
        an int — a stream the CALLER owns and keeps alive for as long as the engine, and any tensor it was given, lives
        (tensors are then held by reference until the engine has used them, never `record_stream`ed: torch would record
        an event on that stream whenever the tensor is finally freed, possibly after the caller destroyed it); or None:
        the library's own stream."""
        self._torch_owned_stream = None
        if stream is not None and hasattr(stream, "cuda_stream"):
            if _torch_owns(stream):
                self._torch_owned_stream = stream
            stream = stream.cuda_stream
        self._check(self._lib.rcw_set_stream(self._h, C.c_void_p(stream) if stream else None))

    def bind_obs(self, device_ptr: Optional[int]):
        self._check(self._lib.rcw_bind_obs(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def clear_error(self):
        self._check(self._lib.rcw_clear_error(self._h))

    def profile(self, enable: bool):
        """Bracket the cast and fill kernels of every step with HIP events (<= 256 steps)."""
        self._check(self._lib.rcw_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        """(mean cast kernel ms, mean top view kernel ms (0 without it), mean fill kernel ms, steps recorded).
        With the two-kernel top view the second is its store kernel; its draw kernel runs beside the fill."""
        c, t, f, n = C.c_float(), C.c_float(), C.c_float(), C.c_int32()
        self._check(self._lib.rcw_profile_read(self._h, C.byref(c), C.byref(t), C.byref(f), C.byref(n)))
        return float(c.value), float(t.value), float(f.value), int(n.value)

    def set_top_view_form(self, form: Optional[str] = None, runs: int = 0) -> None:
        """Choose the form update_top_view! takes instead of the library's rule (rcw_set_top_view_form): None = automatic,
        "in-place", "one-kernel" or "two-kernels"; `runs` = 0 (automatic) or 1..8 runs of agents for the two-kernel form.
        All forms write the same pixels.  Raises RcwError (unsupported) when the geometry cannot take the form."""
        forms = {None: 0, "auto": 0, "in-place": _capi.RCW_TOP_VIEW_IN_PLACE, "one-kernel": _capi.RCW_TOP_VIEW_ONE_KERNEL,
                 "two-kernels": _capi.RCW_TOP_VIEW_TWO_KERNELS}
        if form not in forms:
            raise ValueError(f"unknown top view form {form!r}")
        self._check(self._lib.rcw_set_top_view_form(self._h, forms[form], int(runs)))

    def step_form(self) -> str:
        """How many launches a step of this handle takes (rcw_step_form): "two-launches" (cast kernel, then fill kernel) or
        "one-launch" (the fill picks each frame among the successors the previous launch cast: rcw_fill256_cast_kernel)."""
        f = C.c_int32()
        self._check(self._lib.rcw_step_form(self._h, C.byref(f)))
        return {_capi.RCW_STEP_TWO_LAUNCHES: "two-launches", _capi.RCW_STEP_ONE_LAUNCH: "one-launch"}[f.value]

    def set_step_form(self, form: Optional[str] = None) -> None:
        """Choose the step's form instead of the library's rule (rcw_set_step_form): None = automatic, "two-launches" or
        "one-launch".  Both leave the same state and the same pixels.  Raises RcwError (unsupported) when the geometry cannot
        take the one-launch form (a 256-row camera view without a top view)."""
        forms = {None: 0, "auto": 0, "two-launches": _capi.RCW_STEP_TWO_LAUNCHES, "one-launch": _capi.RCW_STEP_ONE_LAUNCH}
        if form not in forms:
            raise ValueError(f"unknown step form {form!r}")
        self._check(self._lib.rcw_set_step_form(self._h, forms[form]))

    def fill_kernel_name(self) -> str:
        """The kernel update_camera_view! takes for this camera height (rcw_fill_kernel_name)."""
        buf = C.create_string_buffer(64)
        self._check(self._lib.rcw_fill_kernel_name(self._h, buf, 64))
        return buf.value.decode()

    def top_view_form(self) -> str:
        """Which kernel form update_top_view! takes for this geometry: "none", "in-place", "one-kernel", "two-kernels"."""
        f = C.c_int32()
        self._check(self._lib.rcw_top_view_form(self._h, C.byref(f)))
        return ("none", "in-place", "one-kernel", "two-kernels")[f.value]

    def update_top_view_form(self) -> str:
        """The form `update_top_view_(env)` takes when called alone, outside a step (rcw_update_top_view_form)."""
        f = C.c_int32()
        self._check(self._lib.rcw_update_top_view_form(self._h, C.byref(f)))
        return ("none", "in-place", "one-kernel", "two-kernels")[f.value]

    def timer_start(self):
        self._check(self._lib.rcw_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self._lib.rcw_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)


# ---- the generic functions of RayCastWorlds.jl:7-14 that are on the path ------------------
def reference_reset_draws(rng, H: int, W: int, nd: int, old_goal=None):
    """The random draws of ONE `reset!(world)` (SR:110-137) from `rng` (a `numpy.random.Generator`: `integers(lo, hi)`,
    `hi` exclusive), in the reference's order and number:

        rand(rng, 2:H-1), rand(rng, 2:W-1)                     the goal tile                      SR:120
        rand(rng, CartesianIndices((1:H, 1:W)))                the player's tile; drawn AGAIN while the tile is occupied
                                                               (wall ring or the new goal), at most 1024 H W times  UT:23-37
        rand(rng, 0:nd-1)                                      the heading                        SR:128

    A position is ONE draw of a linear index into the column-major region (i = lin mod H + 1, j = lin div H + 1), as
    `rand(rng, ::AbstractArray)` indexes its argument with one `rand(rng, 1:length)`.  Returns (goal_i, goal_j, tile_i,
    tile_j, heading), tiles 1-based.  `old_goal` is not needed: reset! clears the old goal bit before it draws (SR:118)."""
    gi = int(rng.integers(2, H))                     # 2 : H - 1
    gj = int(rng.integers(2, W))

    def occupied(lin):
        i, j = lin % H + 1, lin // H + 1
        return i == 1 or i == H or j == 1 or j == W or (i == gi and j == gj)      # any(@view tile_map[:, position]) UT:27

    lin = int(rng.integers(0, H * W))                # UT:24
    for _ in range(1024 * H * W):                    # UT:26
        if occupied(lin):
            lin = int(rng.integers(0, H * W))        # UT:28
        else:
            break
    else:                                            # UT:34 (@warn, then the occupied tile is returned)
        import warnings

        warnings.warn(f"Could not sample an empty position in max_tries = {1024 * H * W}. "
                      f"Returning non-empty position: ({lin % H + 1}, {lin // H + 1})", RuntimeWarning, stacklevel=2)
    d = int(rng.integers(0, nd))                     # 0 : nd - 1
    return gi, gj, lin % H + 1, lin // H + 1, d


def _reset_from_rng(env: "SingleRoom", rng, mask, construction: bool = False, first: int = 0, global_batch: Optional[int] = None) -> None:
    """reset!(world) for every (unmasked) agent with the caller's generator(s): agent after agent from ONE generator —
    B reference worlds sharing an rng, reset in order — or agent a from `rng[a]` — each agent the reference world built
    with that rng.  The state goes to the engine with one rcw_set_state (tile_map goal bit, pose, heading, reward 0,
    done false, rays cast, views rendered: SR:118-134, SR:328-329).

    `first` / `global_batch` (ShardedSingleRoom): this engine holds the global agents [first, first + batch) of `global_batch`;
    `rng` and `mask` are then the GLOBAL ones — one generator is advanced through every unmasked global agent's draws and this
    shard keeps its own, so that the states do not depend on the sharding (every rank passes generators in the same state)."""
    H, W, nd, B = env.cfg.height_tile_map_tu, env.cfg.width_tile_map_tu, env.cfg.num_directions, env.batch
    G = B if global_batch is None else int(global_batch)
    per_agent = not hasattr(rng, "integers")
    if per_agent and len(rng) != G:
        raise ValueError(f"expected one generator or {G} of them, got {len(rng)}")
    gm = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(G)
    goal = np.ones((B, 2), dtype=np.int32)
    pos = np.ones((B, 2), dtype=env.T)
    head = np.zeros(B, dtype=np.int32)
    for ga in (range(first, first + B) if per_agent else range(G)):          # (one generator: the other shards' draws are made and dropped)
        if gm is not None and not gm[ga]:
            continue
        g = rng[ga] if per_agent else rng
        if construction:
            reference_reset_draws(g, H, W, nd)       # SR:62-74: drawn, then overwritten by reset!(world) SR:105
        gi, gj, ti, tj, d = reference_reset_draws(g, H, W, nd)
        a = ga - first
        if 0 <= a < B:
            goal[a] = (gi, gj)
            pos[a] = (env.T(ti - 0.5), env.T(tj - 0.5))  # convert(T, tile - 0.5) SR:125
            head[a] = d
    # masked-out agents keep their state: rcw_set_state skips them, the placeholders above are never read
    env.set_state(goal, pos, head, mask=None if gm is None else gm[first:first + B])


def reset_(env: SingleRoom, mask=None, seed: Optional[int] = None, rng=None) -> None:
    """`RCW.reset!(env)` SR:326-331 — all agents, or those with a non-zero `mask` byte.

    Default: sampled on the device by the engine's counter-based generator keyed (seed, global agent id, episode).
    `rng` (or an environment constructed with one, SR:49): the reference's own keyword — the draws come from the caller's
    generator on the host, in the reference's order (`reference_reset_draws`)."""
    if rng is None and seed is None:
        rng = getattr(env, "rng", None)
    if rng is not None:
        _reset_from_rng(env, rng, mask)
        return None
    if seed is not None:
        env.seed = int(seed)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(env.batch)
    env._check(env._lib.rcw_reset(env._h, _as_ptr(m), env.seed))
    return None


def host_actions(batch: int, action) -> np.ndarray:
    """Normalise host-side actions to a contiguous uint8 (batch,) array; anything outside 1..4
    raises AssertionError like `@assert action in Base.OneTo(NUM_ACTIONS)` (SR:140)."""
    if np.isscalar(action):
        a = np.full(batch, action)
    else:
        a = np.asarray(action).reshape(-1)
        if a.size != batch:
            raise ValueError(f"expected {batch} actions, got {a.size}")
        if a.dtype == np.uint8 and a.flags.c_contiguous and a.size:          # the common case: two reductions, no temporaries
            if a.min() >= 1 and a.max() <= NUM_ACTIONS:
                return a
    ok = (a >= 1) & (a <= NUM_ACTIONS)
    if a.dtype.kind == "f":
        ok &= a == np.floor(a)
    if not np.all(ok):
        raise AssertionError(f"Invalid action: {a[~ok][0]}")
    return np.ascontiguousarray(a.astype(np.uint8, copy=False))


def unpack_tile_map(chunks: np.ndarray, H: int, W: int) -> np.ndarray:
    """BitArray{3}(2, H, W).chunks per agent (uint64 (B, nchunks)) -> bool (B, 2, H, W) with
    tile_map[b, o-1, i-1, j-1] == Julia tile_map[o, i, j]; bit index (o-1) + 2(i-1) + 2H(j-1)."""
    chunks = np.ascontiguousarray(chunks, dtype=np.uint64)
    bits = np.unpackbits(chunks.view(np.uint8), axis=1, bitorder="little")[:, : 2 * H * W]
    return bits.reshape(chunks.shape[0], W, H, 2).transpose(0, 3, 2, 1).astype(bool)


def _is_device_tensor(x) -> bool:
    return hasattr(x, "is_cuda") and hasattr(x, "data_ptr")


def act_(env: SingleRoom, action) -> None:
    """`RCW.act!(env, action)` SR:333-340 (without the top view) for every agent.

    `action`: one int in 1..4 applied to all agents, a length-B sequence / numpy array, or
    a CUDA uint8 torch tensor of length B (stays on the device).  Anything outside 1..4
    raises AssertionError and steps no agent (`@assert` SR:140).
    """
    if _is_device_tensor(action):
        if not action.is_cuda:
            action = action.numpy()
        else:
            import torch

            if action.dtype != torch.uint8 or action.numel() != env.batch or not action.is_contiguous():
                raise ValueError("device actions must be a contiguous uint8 tensor of length batch")
            # The actions were produced on torch's current stream; unless the engine runs on that very
            # stream, make the engine's stream wait for them (an event record + wait, no host sync) and
            # hold the tensor until the cast kernel has read it (the caller may drop a temporary at once).
            cross = env._order_behind_torch()
            env._check(env._lib.rcw_step_device(env._h, C.c_void_p(action.data_ptr())))
            if cross:
                env._release_after_use(action)
            return None
    if isinstance(action, (int, np.integer)) and not isinstance(action, bool):
        # one action for every agent (what the reference's own loop passes: env(rand(1:4)), test/runtests.jl:28): a
        # device-resident constant per action, made once — no staging copy, no array to validate
        if not 1 <= int(action) <= NUM_ACTIONS:
            raise AssertionError(f"Invalid action: {action}")
        buf = env._constant_actions(int(action))
        if buf is not None:
            env._check(env._lib.rcw_step_device(env._h, C.c_void_p(buf.data_ptr())))
            return None
    a = host_actions(env.batch, action)
    env._check(env._lib.rcw_step(env._h, _as_ptr(a)))
    return None


def cast_rays_(env: SingleRoom, first: int = 0, count: Optional[int] = None):
    """`RCW.cast_rays!(world)` SR:195-231: recasts every agent's rays from the current state (rcw_cast_rays:
    the compact column descriptors are refreshed, no pixel is written) and returns the ray buffers
    (stop tile, hit dimension, distance, direction) of agents [first, first+count)."""
    env._check(env._lib.rcw_cast_rays(env._h))
    return env.world.rays(first, count)


def update_camera_view_(env: SingleRoom) -> None:
    """`RCW.update_camera_view!(env)` SR:374-444: refills `camera_view` from the stored ray results without
    casting, as the reference's does (rcw_update_camera_view: the fill kernel alone)."""
    env._check(env._lib.rcw_update_camera_view(env._h))


def update_top_view_(env: SingleRoom) -> None:
    """`RCW.update_top_view!(env)` SR:446-483 (env built with render_top_view=True): redraws `top_view`."""
    env._check(env._lib.rcw_update_top_view(env._h))


def get_action_keys(env: SingleRoom):
    """`RCW.get_action_keys(env)` SR:485: the keys of actions 1..4 — MiniFB's KB_KEY_W / S / A / D there, their letters here
    (what `viewer.play_keys` takes)."""
    return ("w", "s", "a", "d")


def get_action_names(env: SingleRoom):
    """`RCW.get_action_names(env)` SR:486."""
    return ("MOVE_FORWARD", "MOVE_BACKWARD", "TURN_LEFT", "TURN_RIGHT")


def wu_to_tu(x_wu) -> int:
    """utils.jl:5: the (1-based) tile of a world-unit coordinate."""
    return int(np.floor(x_wu)) + 1


def wu_to_pu(x_wu, pu_per_wu) -> int:
    """utils.jl:6: the (1-based) pixel of a world-unit coordinate; the product is formed in x_wu's own type, as Julia does."""
    return int(np.floor(x_wu * pu_per_wu)) + 1


def pu_to_tu(i_pu: int, pu_per_tu: int) -> int:
    """utils.jl:7: the (1-based) tile of a (1-based) pixel, `(i_pu - 1) ÷ pu_per_tu + 1`.  Julia's `÷` truncates toward zero
    (Python's `//` floors): for a pixel off the image's low edge, i_pu <= 0, the two differ — pixel 0 at 32 pixels a tile is
    tile 1 in the reference, not tile 0."""
    n, d = int(i_pu) - 1, int(pu_per_tu)
    q = abs(n) // abs(d)
    return (q if (n < 0) == (d < 0) else -q) + 1
