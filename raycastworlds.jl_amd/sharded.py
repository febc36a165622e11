"""Multi-GPU: agents shard across ranks, one process per GPU, no data-path collective.

Agents are independent (nothing in SingleRoomWorld, SR:21-40, refers to another world), so a
global batch of B agents on G GPUs is G engines of B/G agents; rank r owns the contiguous
global ids [r*B/G, (r+1)*B/G).  `agent_id_offset` keys the reset generator by GLOBAL id, so
the states do not depend on how the batch is sharded.  Stepping needs no communication.

The only exchange is the optional observation gather BASELINE.json's north_star names.
Over xGMI a full-frame gather costs ~50x what producing the frames costs, so
the default gathers the compact per-column descriptors (5 bytes per column instead of
4*H_cam) with RCCL and expands them to pixels on the receiving GPU (rcw_expand_columns);
`mode="frames"` gathers the pixels themselves for API fidelity.

Two transports, same result:
  * `gather_*`      torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests); the
                    collective is issued from the engine's stream, so it is ordered behind the step
                    that produced the frames without any host synchronisation;
  * `gather_*_abi`  the library's own entry points (rcw_comm_init / rcw_gather_*), which call RCCL
                    directly on the engine's stream — what a host without torch (the Julia binding)
                    uses; here torch.distributed only carries the 128-byte ncclUniqueId to the ranks.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """(first global agent id, count) of `rank`'s contiguous shard; shards must be equal so
    that all_gather needs no padding."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    per = global_batch // world
    return rank * per, per


def default_device() -> int:
    """The HIP device of this process: LOCAL_RANK under torch.distributed.run (the global rank is
    not a device index on a multi-node job), else torch's current device."""
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    try:
        import torch

        if torch.cuda.is_available():
            return int(torch.cuda.current_device())
    except ImportError:
        pass
    return 0


def make_unique_id(library: Optional[str] = None) -> bytes:
    """rcw_comm_unique_id: the 128 bytes ONE rank makes and every rank of the gather group hands to
    `ShardedSingleRoom.comm_init_abi(unique_id=...)` / rcw_comm_init."""
    from . import _capi

    _capi.preload_rccl()
    lib = _capi.load(library)
    uid = (C.c_uint8 * _capi.RCW_UNIQUE_ID_BYTES)()
    _capi.check(lib.rcw_comm_unique_id(uid), lib)
    return bytes(uid)


class ShardedSingleRoom:
    """B agents over `world` ranks.  `env_factory(batch=, agent_id_offset=, device=, **kw)` builds
    the local engine: SingleRoomModule.SingleRoom (HIP) unless a test injects another.

    `collective="auto"` skips the all-gather on a world of one rank (there is nothing to exchange);
    `"always"` issues it regardless — how the RCCL path is exercised on a one-GPU box."""

    def __init__(self, global_batch: int, *, rank: Optional[int] = None, world: Optional[int] = None,
                 group=None, device: Optional[int] = None, env_factory: Optional[Callable] = None,
                 collective: str = "auto", **kwargs):
        if collective not in ("auto", "always"):
            raise ValueError(f"collective must be 'auto' or 'always', not {collective!r}")
        self.group = group
        self.collective = collective
        # torch.distributed is asked only for what the caller did not say: with `rank` and `world` given (and the library
        # transport's unique id brought by the caller, comm_init_abi) a host without torch shards and gathers all the same
        if world is None or rank is None:
            dist = self._dist
            if world is None:
                world = dist.get_world_size(group) if dist.is_initialized() else 1
            if rank is None:
                rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world, self.rank = int(world), int(rank)
        self.global_batch = int(global_batch)
        self.first, self.count = shard_range(self.global_batch, self.world, self.rank)
        if env_factory is None:
            from .single_room import SingleRoom as env_factory
        # the reference's `rng` keyword (SR:49,265) for a sharded batch: generators are GLOBAL — one per global agent, or one
        # for all, which every rank advances through all global agents' draws (in the same initial state on every rank) —
        # so that the states do not depend on the sharding, as with the device generator (agent_id_offset)
        self.rng = kwargs.pop("rng", None)
        self.env = env_factory(batch=self.count, agent_id_offset=self.first,
                               device=default_device() if device is None else device, **kwargs)
        self._abi_comm = False
        if self.rng is not None:
            from .single_room import _reset_from_rng

            _reset_from_rng(self.env, self.rng, None, construction=True, first=self.first, global_batch=self.global_batch)

    @property
    def _dist(self):
        import torch.distributed as dist                  # (on first use: the torch transport, or a rank / world left to it)

        return dist

    # ---- stepping: purely local ------------------------------------------------------
    def local_slice(self, global_array):
        """This rank's rows of an array indexed by global agent id."""
        return global_array[self.first:self.first + self.count]

    def act_(self, local_actions) -> None:
        from .single_room import act_

        act_(self.env, local_actions)

    def reset_(self, local_mask=None, seed: Optional[int] = None, rng=None, global_mask=None) -> None:
        """`RCW.reset!` of this rank's agents: on the device (`local_mask`, `seed`), or — `rng`, or a batch constructed with
        one — from the caller's GLOBAL generator(s), with the GLOBAL mask (`global_mask`: a single generator's draws for the
        other ranks' unmasked agents have to be made too)."""
        from .single_room import _reset_from_rng, reset_

        if rng is None and seed is None:
            rng = self.rng
        if rng is not None:
            if local_mask is not None and global_mask is None:
                if hasattr(rng, "integers") and self.world > 1:
                    raise ValueError("a masked reset from ONE generator needs the GLOBAL mask (global_mask=...): the other ranks' draws are made too")
                global_mask = np.ones(self.global_batch, dtype=np.uint8)
                global_mask[self.first:self.first + self.count] = np.asarray(local_mask, dtype=np.uint8).reshape(self.count)
            _reset_from_rng(self.env, rng, global_mask, first=self.first, global_batch=self.global_batch)
            return
        reset_(self.env, local_mask, seed)

    # ---- the observation gather over torch.distributed -------------------------------------
    def _as_tensor(self, x):
        import torch

        if hasattr(x, "__cuda_array_interface__"):
            return x.torch(sync=False)   # ordered on the engine's stream below, not by a host sync
        return torch.as_tensor(x)

    def _skip(self) -> bool:
        return self.world == 1 and self.collective == "auto"

    def _engine_stream(self, tensor):
        """Context that makes the engine's stream torch's current stream (device tensors), so the
        collective is enqueued behind the step and its result is ordered on that same stream."""
        import contextlib

        import torch

        if getattr(tensor, "is_cuda", False) and hasattr(self.env, "torch_stream"):
            return torch.cuda.stream(self.env.torch_stream())
        return contextlib.nullcontext()

    def _all_gather(self, local, out):
        """all_gather_into_tensor of `local` (this rank's rows) into `out` (global rows).  uint32 frames go
        through the process group as int32 (same bytes) so nothing depends on unsigned support in the backend."""
        import torch

        a, b = local.contiguous(), out
        if a.dtype == torch.uint32:
            a, b = a.view(torch.int32), b.view(torch.int32)
        self._dist.all_gather_into_tensor(b, a, group=self.group)
        return out

    def gather_columns(self):
        """all_gather of the compact descriptors: (height_line_pu int32 (B, N), colour id uint8 (B, N))
        for the GLOBAL batch, on every rank (device tensors on GPU), ordered on the engine's stream."""
        import torch

        h, c = self.env.columns_device()
        h, c = self._as_tensor(h), self._as_tensor(c)
        if self._skip():
            return h, c
        with self._engine_stream(h):
            gh = torch.empty((self.global_batch,) + tuple(h.shape[1:]), dtype=h.dtype, device=h.device)
            gc = torch.empty((self.global_batch,) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device)
            self._all_gather(h, gh)
            self._all_gather(c, gc)
        return gh, gc

    def gather_observations(self, mode: str = "columns"):
        """The GLOBAL observation batch (B, N, H_cam) on every rank, ordered on the engine's stream
        (a consumer on another stream waits for `env.torch_stream()`; `env.sync()` waits on the host).

        mode="columns": gather descriptors (5 B/column), expand to pixels locally;
        mode="frames":  gather the pixels (4*H_cam B/column) — ~50x the bytes over xGMI."""
        import torch

        if mode == "frames":
            local = self._as_tensor(self.env.camera_view)
            if self._skip():
                return local
            with self._engine_stream(local):
                out = torch.empty((self.global_batch,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
                self._all_gather(local, out)
            return out
        if mode != "columns":
            raise ValueError(f"unknown gather mode {mode!r}")
        gh, gc = self.gather_columns()
        with self._engine_stream(gh):
            return self.env.expand_columns(gh, gc)

    # ---- the same gather through the C ABI (RCCL called by the library) -----------------------
    def comm_init_abi(self, unique_id: Optional[bytes] = None) -> None:
        """rcw_comm_init on this rank's engine.  Rank 0 makes the ncclUniqueId (rcw_comm_unique_id) and
        torch.distributed carries its 128 bytes to the other ranks; a world of one needs no transport.

        `unique_id`: the 128 bytes of `make_unique_id()` brought here by the CALLER's own transport (MPI, a file, a socket:
        what a host without torch.distributed does — and how several ranks can live in one process, one engine each,
        where torch.distributed's ranks are processes).  That path imports no torch."""
        from . import _capi

        if self._abi_comm:
            return
        _capi.preload_rccl()
        lib = self.env._lib
        if unique_id is not None:
            if len(unique_id) != _capi.RCW_UNIQUE_ID_BYTES:
                raise ValueError(f"a unique id is {_capi.RCW_UNIQUE_ID_BYTES} bytes, got {len(unique_id)}")
            uid = (C.c_uint8 * _capi.RCW_UNIQUE_ID_BYTES)(*bytes(unique_id))
            self.env._check(lib.rcw_comm_init(self.env._h, uid, self.rank, self.world))
            self._abi_comm = True
            return
        uid = (C.c_uint8 * _capi.RCW_UNIQUE_ID_BYTES)()
        if self.rank == 0:
            self.env._check(lib.rcw_comm_unique_id(uid))
        if self.world > 1:
            import torch                                  # (only the torch.distributed hand-over needs it)

            t = torch.tensor(list(uid), dtype=torch.uint8)
            backend = self._dist.get_backend(self.group)
            if backend == "nccl":
                t = t.cuda(self.env.device)
            self._dist.broadcast(t, src=0, group=self.group)
            uid = (C.c_uint8 * _capi.RCW_UNIQUE_ID_BYTES)(*t.cpu().tolist())
        self.env._check(lib.rcw_comm_init(self.env._h, uid, self.rank, self.world))
        self._abi_comm = True

    def gather_columns_abi(self):
        """rcw_gather_columns: (height_line_pu int32 (B, N), colour id uint8 (B, N)) of the global batch as
        torch tensors filled on the engine's stream."""
        import torch

        self.comm_init_abi()
        env = self.env
        dev = f"cuda:{env.device}"
        gh = torch.empty((self.global_batch, env.cfg.num_rays), dtype=torch.int32, device=dev)
        gc = torch.empty((self.global_batch, env.cfg.num_rays), dtype=torch.uint8, device=dev)
        cross = env._order_behind_torch()
        from . import _capi

        env._check(env._lib.rcw_gather_columns(env._h, C.c_void_p(gh.data_ptr()), C.c_void_p(gc.data_ptr())))
        if cross:
            env._release_after_use(gh, gc)
        return gh, gc

    def gather_observations_abi(self, mode: str = "columns", out=None):
        """rcw_gather_observations: the global observation batch (B, N, H_cam) uint32, filled on the
        engine's stream by the library (RCCL all-gather + on-device expansion)."""
        import torch

        from . import _capi

        modes = {"columns": _capi.RCW_GATHER_COLUMNS, "frames": _capi.RCW_GATHER_FRAMES}
        if mode not in modes:
            raise ValueError(f"unknown gather mode {mode!r}")
        self.comm_init_abi()
        env = self.env
        if out is None:
            out = torch.empty((self.global_batch, env.cfg.num_rays, env.cfg.height_camera_view_pu), dtype=torch.uint32,
                              device=f"cuda:{env.device}")
        cross = env._order_behind_torch()
        env._check(env._lib.rcw_gather_observations(env._h, modes[mode], C.c_void_p(out.data_ptr())))
        if cross:
            env._release_after_use(out)
        return out

    def close(self):
        if self._abi_comm:
            self.env._lib.rcw_comm_destroy(self.env._h)
            self._abi_comm = False
        self.env.close()
