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
"""
from __future__ import annotations

import ctypes as C
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


class ShardedSingleRoom:
    """B agents over `world` ranks.  `env_factory(batch=, agent_id_offset=, device=, **kw)` builds
    the local engine: SingleRoomModule.SingleRoom (HIP) unless a test injects another."""

    def __init__(self, global_batch: int, *, rank: Optional[int] = None, world: Optional[int] = None,
                 group=None, device: Optional[int] = None, env_factory: Optional[Callable] = None, **kwargs):
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world, self.rank = int(world), int(rank)
        self.global_batch = int(global_batch)
        self.first, self.count = shard_range(self.global_batch, self.world, self.rank)
        if env_factory is None:
            from .single_room import SingleRoom as env_factory
        self.env = env_factory(batch=self.count, agent_id_offset=self.first,
                               device=self.rank if device is None else device, **kwargs)

    # ---- stepping: purely local ------------------------------------------------------
    def local_slice(self, global_array):
        """This rank's rows of an array indexed by global agent id."""
        return global_array[self.first:self.first + self.count]

    def act_(self, local_actions) -> None:
        from .single_room import act_

        act_(self.env, local_actions)

    def reset_(self, local_mask=None, seed: Optional[int] = None) -> None:
        from .single_room import reset_

        reset_(self.env, local_mask, seed)

    # ---- the observation gather --------------------------------------------------------
    def _as_tensor(self, x):
        import torch

        if hasattr(x, "__cuda_array_interface__"):
            return x.torch()
        return torch.as_tensor(x)

    def gather_columns(self):
        """all_gather of the compact descriptors: (height_line_pu int32 (B, N), colour id uint8 (B, N))
        for the GLOBAL batch, on every rank (device tensors on GPU)."""
        import torch

        h, c = self.env.columns_device()
        h, c = self._as_tensor(h), self._as_tensor(c)
        if self.world == 1:
            return h, c
        gh = torch.empty((self.global_batch,) + tuple(h.shape[1:]), dtype=h.dtype, device=h.device)
        gc = torch.empty((self.global_batch,) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device)
        self._dist.all_gather_into_tensor(gh, h.contiguous(), group=self.group)
        self._dist.all_gather_into_tensor(gc, c.contiguous(), group=self.group)
        return gh, gc

    def gather_observations(self, mode: str = "columns"):
        """The GLOBAL observation batch (B, N, H_cam) on every rank.

        mode="columns": gather descriptors (5 B/column), expand to pixels locally;
        mode="frames":  gather the pixels (4*H_cam B/column) — ~50x the bytes over xGMI."""
        import torch

        if mode == "frames":
            local = self._as_tensor(self.env.camera_view)
            if self.world == 1:
                return local
            out = torch.empty((self.global_batch,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            self._dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
            return out
        if mode != "columns":
            raise ValueError(f"unknown gather mode {mode!r}")
        gh, gc = self.gather_columns()
        return self.env.expand_columns(gh, gc)

    def close(self):
        self.env.close()
