#!/usr/bin/env python3
"""Run by tests/test_gpu_rccl.py in a child process (the rendezvous variables have to be in the environment
before anything touches the GPU): a torch.distributed "nccl" (= RCCL) process group with ONE rank on the one GPU
of the box, the real HIP engine behind ShardedSingleRoom, and the observation gather forced through the
collective (collective="always") on both transports:

  torch    ShardedSingleRoom.gather_columns / gather_observations  -> dist.all_gather_into_tensor over RCCL
  abi      ShardedSingleRoom.gather_*_abi -> rcw_comm_init + rcw_gather_columns / rcw_gather_observations
           (librcw_hip calls ncclAllGather itself, on the engine's stream)

It checks each result against the engine's own camera_view / descriptors and then times the gather at the
per-GPU shard of BASELINE.json configs[3] (8192 agents x 256 columns, 16x16 map) with HIP events on the
engine's stream.  Prints one JSON line.  The 1 -> 8 GPU curve is NOT measured here: one rank moves no bytes
over xGMI; what this run establishes is that every RCCL call of the path executes on hardware.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi_census

abi_census.install_if_asked()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--time-batch", type=int, default=0, help="0: skip the timing part")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    assert os.environ.get("WORLD_SIZE") == "1" and os.environ.get("RANK") == "0"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))

    import raycastworlds_jl_amd as RCW

    CFG4 = dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=256)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}

    # ---- correctness: both transports, both modes, against the engine's own buffers ----
    sh = RCW.ShardedSingleRoom(args.batch, collective="always", device=0, seed=5, out_of_bounds=1, **CFG4)
    env = sh.env
    rng = np.random.default_rng(3)
    for _ in range(12):
        sh.act_(rng.integers(1, 5, args.batch).astype(np.uint8))
    want = env.camera_view_host()
    want_h, want_c = env.columns()
    for name, fn in (("torch", sh.gather_columns), ("abi", sh.gather_columns_abi)):
        gh, gc = fn()
        env.sync()
        assert gh.data_ptr() != env.columns_device()[0].ptr, f"{name}: no collective ran (aliased the local buffer)"
        np.testing.assert_array_equal(gh.cpu().numpy(), want_h, err_msg=f"{name} height_line_pu")
        np.testing.assert_array_equal(gc.cpu().numpy(), want_c, err_msg=f"{name} colour id")
    for name, fn in (("torch", sh.gather_observations), ("abi", sh.gather_observations_abi)):
        for mode in ("columns", "frames"):
            frames = fn(mode)
            env.sync()
            assert frames.data_ptr() != env.camera_view.ptr, f"{name}/{mode}: no collective ran"
            np.testing.assert_array_equal(frames.cpu().numpy().view(np.uint32), want, err_msg=f"{name} {mode}")
    # a step enqueued right behind a gather must not disturb it, and vice versa (stream order, no host sync)
    a = rng.integers(1, 5, args.batch).astype(np.uint8)
    sh.act_(a)
    f1 = sh.gather_observations_abi("columns")
    sh.act_(3)
    f2 = sh.gather_observations("frames")
    env.sync()
    after = env.camera_view_host()
    np.testing.assert_array_equal(f2.cpu().numpy().view(np.uint32), after)
    assert not np.array_equal(f1.cpu().numpy().view(np.uint32), after)
    out["parity"] = "ok"
    sh.close()
    del sh, env, frames, f1, f2, gh, gc

    # ---- timing at the cfg-4 shard size ----
    if args.time_batch:
        B = args.time_batch
        sh = RCW.ShardedSingleRoom(B, collective="always", device=0, seed=6, out_of_bounds=1, auto_reset=True, **CFG4)
        env = sh.env
        es = env.torch_stream()
        acts = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        N, Hc = 256, 256
        frames_buf = torch.empty((B, N, Hc), dtype=torch.uint32, device="cuda")
        torch.cuda.synchronize()

        def timed(fn):
            for _ in range(2):
                fn()
            env.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(es)
            for _ in range(args.reps):
                fn()
            e1.record(es)
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e3 / args.reps      # us per call

        t = {}
        t["step_us"] = timed(lambda: sh.act_(acts))
        t["abi_columns_gather_only_us"] = timed(lambda: sh.gather_columns_abi())
        t["abi_columns_gather_plus_expand_us"] = timed(lambda: sh.gather_observations_abi("columns", out=frames_buf))
        t["abi_frames_us"] = timed(lambda: sh.gather_observations_abi("frames", out=frames_buf))
        t["torch_columns_gather_only_us"] = timed(lambda: sh.gather_columns())
        t["torch_columns_gather_plus_expand_us"] = timed(lambda: sh.gather_observations("columns"))
        t["torch_frames_us"] = timed(lambda: sh.gather_observations("frames"))
        out["timing"] = {"agents": B, "columns": N, "H_cam": Hc, "reps": args.reps,
                         "descriptor_bytes": 5 * N * B, "frame_bytes": 4 * N * Hc * B,
                         **{k: round(v, 1) for k, v in t.items()},
                         "note": "one rank: RCCL's all-gather is a device-local copy, no xGMI traffic; "
                                 "the 1/2/4/8-GPU curve is unmeasured"}
        sh.close()
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
