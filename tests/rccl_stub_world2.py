#!/usr/bin/env python3
"""One rank of tests/test_gpu_rccl.py::test_library_transport_with_two_ranks (run twice, RANK = 0 and 1, both on GPU 0).

The library's own transport — rcw_comm_unique_id on rank 0, the 128 bytes carried to rank 1 by torch.distributed
(gloo; sharded.py's broadcast branch), rcw_comm_init(rank, world = 2), rcw_gather_columns / rcw_gather_observations in
both modes — with the real engine and real frames; only the collective library underneath is tests/stub_rccl.c (real
RCCL refuses two ranks on one device).  Each rank checks the gathered GLOBAL batch against the CPU oracle of the
unsharded batch and prints one JSON line."""
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
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 2 and "RCW_RCCL_LIBRARY" in os.environ
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import raycastworlds_jl_amd as RCW
    from oracle import oracle as O

    cfg = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    B, steps = 24, 20
    sh = RCW.ShardedSingleRoom(B, rank=rank, world=world, device=0, seed=11, out_of_bounds=1, **cfg)
    assert (sh.first, sh.count) == (rank * B // 2, B // 2)
    ref = O.OracleBatch(B, seed=11, out_of_bounds=1, **cfg)                   # the unsharded truth (same generator, global ids)
    rng = np.random.default_rng(3)
    for _ in range(steps):
        a = rng.integers(1, 5, B).astype(np.uint8)
        sh.act_(sh.local_slice(a))
        assert ref.step(a) == 0
    gh, gc = sh.gather_columns_abi()                                          # comm_init_abi: uid broadcast + rcw_comm_init(rank, 2)
    sh.env.sync()
    np.testing.assert_array_equal(gh.cpu().numpy(), ref.col_height)
    np.testing.assert_array_equal(gc.cpu().numpy(), ref.col_colour)
    info_rank, info_world = __import__("ctypes").c_int32(), __import__("ctypes").c_int32()
    sh.env._check(sh.env._lib.rcw_comm_info(sh.env._h, info_rank, info_world))
    assert (info_rank.value, info_world.value) == (rank, world)
    for mode in ("columns", "frames"):
        frames = sh.gather_observations_abi(mode)
        sh.env.sync()
        np.testing.assert_array_equal(frames.cpu().numpy().astype(np.uint32), ref.camera_view, err_msg=f"mode {mode}")
    # a step right behind a gather, a gather right behind a step
    a = rng.integers(1, 5, B).astype(np.uint8)
    sh.act_(sh.local_slice(a)); ref.step(a)
    frames = sh.gather_observations_abi("columns")
    sh.env.sync()
    np.testing.assert_array_equal(frames.cpu().numpy().astype(np.uint32), ref.camera_view)
    sh.close()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"rank": rank, "world": world, "parity": "ok", "global_batch": B}))


if __name__ == "__main__":
    main()
