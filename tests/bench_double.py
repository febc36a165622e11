#!/usr/bin/env python3
"""bench.main() — the real function, every line of its control flow — on a CPU, with doubles for what it cannot have here: a GPU,
a process group, the engine.  Run by tests/test_bench_logic.py as rank 0 of a world of WORLD_SIZE ranks (the other ranks are
imagined: every collective returns what `world` identical ranks would have contributed).

The `dist` double is STRICTER than any one backend: all_gather_into_tensor takes only an output that is the concatenation of `world`
inputs along dim 0 — same number of dimensions, same trailing shape, same dtype, same device, both contiguous — which is the form both
RCCL and gloo accept (gloo rejects the stacked [world, n] output that RCCL takes: round 5's first gather_rows).  It also notes
whether any collective is issued while one of bench.py's clocks is running (the engine double sees timer_start / timer_stop, the
perf_counter pairs bracket them).

No product code runs here and nothing is measured: the figures in the line are whatever the doubles return.
"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench

CALLS = {"all_gather": 0, "all_reduce": 0, "barrier": 0, "destroyed": 0, "steps": 0, "collectives_inside_timed_regions": 0,
         "expand_columns": 0}
CLOCK = {"running": False}


class DistDouble:
    class ReduceOp:
        MAX = "max"

    def __init__(self, world, rank):
        self.world, self.rank = world, rank

    def _collective(self):
        if CLOCK["running"]:
            CALLS["collectives_inside_timed_regions"] += 1

    def barrier(self):
        self._collective()
        CALLS["barrier"] += 1

    def all_gather_into_tensor(self, out, inp):
        self._collective()
        CALLS["all_gather"] += 1
        if out.dtype != inp.dtype or out.device != inp.device:
            raise RuntimeError(f"all_gather_into_tensor: dtype / device mismatch {out.dtype}/{out.device} vs {inp.dtype}/{inp.device}")
        if not (out.is_contiguous() and inp.is_contiguous()):
            raise RuntimeError("all_gather_into_tensor: tensors must be contiguous")
        if out.dim() != inp.dim() or out.dim() < 1 or tuple(out.shape[1:]) != tuple(inp.shape[1:]) or out.shape[0] != self.world * inp.shape[0]:
            raise RuntimeError(f"all_gather_into_tensor: invalid tensor size: output {tuple(out.shape)} is not {self.world} x input {tuple(inp.shape)} along dim 0")
        out.copy_(torch.cat([inp] * self.world, dim=0))

    def all_reduce(self, t, op=None):
        self._collective()
        CALLS["all_reduce"] += 1

    def destroy_process_group(self):
        CALLS["destroyed"] += 1


class DeviceArrayDouble:
    def __init__(self, t):
        self._t = t

    def torch(self, sync=True):
        return self._t

    @property
    def shape(self):
        return tuple(self._t.shape)


class EnvDouble:
    host_syncs = 0

    def __init__(self, batch, num_rays, **kw):
        self.batch, self.N, self.Hc = batch, num_rays, 256
        self.cfg = types.SimpleNamespace(pu_per_tu=32)
        self.world = types.SimpleNamespace(status=np.zeros(batch, np.int32))
        self._form = "one-launch"
        self._h = torch.zeros((batch, num_rays), dtype=torch.int32)
        self._c = torch.zeros((batch, num_rays), dtype=torch.uint8)
        self._obs = torch.zeros((batch, num_rays, 256), dtype=torch.uint32)
        self._reward = torch.zeros(batch, dtype=torch.float32)
        self._done = torch.zeros(batch, dtype=torch.bool)

    def set_stream(self, s): pass
    def set_step_form(self, f): self._form = f if f not in (None, "auto") else "one-launch"
    def step_form(self): return self._form
    def sync(self): pass
    def clear_error(self): pass
    def timer_start(self): CLOCK["running"] = True
    def timer_stop(self): CLOCK["running"] = False; return 1.0
    def profile(self, on): pass
    def profile_read(self): return 0.0, 0.0, 0.15, 6
    def fill_kernel_name(self): return "rcw_fill256_cast_kernel"
    def top_view_form(self): return "none"
    def columns_device(self): return DeviceArrayDouble(self._h), DeviceArrayDouble(self._c)
    @property
    def camera_view(self): return DeviceArrayDouble(self._obs)
    def expand_columns(self, h, c, out=None): CALLS["expand_columns"] += 1
    def close(self): pass


class RLBaseDouble:
    @staticmethod
    def state(rl): return DeviceArrayDouble(rl.env._obs)
    @staticmethod
    def reward(rl): return DeviceArrayDouble(rl.env._reward)
    @staticmethod
    def is_terminated(rl): return DeviceArrayDouble(rl.env._done)


class RLBaseEnvDouble:
    def __init__(self, env): self.env = env
    def __call__(self, a): CALLS["steps"] += 1


def act_(env, a):
    CALLS["steps"] += 1


ENGINE = types.SimpleNamespace(SingleRoomModule=types.SimpleNamespace(SingleRoom=lambda batch, seed, device, auto_reset, agent_id_offset, render_top_view, **kw: EnvDouble(batch, **kw)),
                               act_=act_, RLBase=RLBaseDouble, RLBaseEnv=RLBaseEnvDouble)


class RuntimeDouble(bench.Runtime):
    device = "cpu"

    def __init__(self, world, rank):
        self.torch = torch
        self.dist = None
        self._world, self._rank = world, rank

    def gpu_available(self): return True
    def set_device(self, local_rank): pass
    def init_dist(self, backend, local_rank): assert backend == "nccl", backend; self.dist = DistDouble(self._world, self._rank)
    def engine(self): return ENGINE
    def ensure_built(self, rank): pass
    def make_actions(self, total, B, rank): return torch.randint(1, 5, (total, B), dtype=torch.uint8)
    def share_stream(self, env): pass
    def synchronize(self): pass


if __name__ == "__main__":
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    try:
        bench.main(sys.argv[1:], rt=RuntimeDouble(world, rank))
    finally:
        log = os.environ.get("BENCH_DOUBLE_LOG")
        if log:
            json.dump(CALLS, open(log, "w"))
