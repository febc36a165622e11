#!/usr/bin/env python3
"""One PROCESS of tests/test_gpu_rccl.py::test_cfg4_as_stated_eight_ranks_with_the_gather — BASELINE.json configs[3] in
its stated shape: the 16x16 room, 256 view columns, 65,536 agents sharded over EIGHT ranks, the observation gather.

A box has one GPU and admits at most six processes on it, so the eight ranks are four processes (PROC = 0..3 of 4, a gloo
group that only carries the 128-byte unique id) with TWO ranks each, one thread and one engine per rank: rank r =
ShardedSingleRoom(65536, rank=r, world=8, **CFG4) = 8,192 agents with agent_id_offset = 8192 r, rcw_comm_init(r, 8) on
its own handle.  The collective library underneath is tests/stub_rccl.c (real RCCL refuses two ranks on one device);
everything above it — the engine, its frames, rcw_comm_* / rcw_gather_* — is the product.

Every rank: a few steps under the reference's BoundsError policy, then rcw_gather_columns of the GLOBAL batch (84 MB)
against the non-rendering CPU oracle of the unsharded batch; rcw_comm_info == (r, 8).  Rank 0 then assembles the
global OBSERVATION batch (rcw_gather_observations, descriptor transport: 17 GB of frames) while the others take part in
the same two all-gathers; every frame is checked against the expansion of its descriptors AND, pixel for pixel, against the
rendering oracle given the unsharded states.  Prints one JSON line per process."""
import ctypes
import json
import os
import sys
import threading
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi_census

abi_census.install_if_asked()

WORLD, GLOBAL_BATCH, STEPS = 8, 65536, 5
CFG4 = dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=256)
COLOURS = [0x808080, 0xC0C0C0, 0x800000, 0xC00000]


def torch_expand(h, c, Hc=256):
    """(B, N) descriptors -> (B, N, Hc) pixels, straight from SR:431-440 (as tests/test_gpu_full_size.py)."""
    h = h.to(torch.int64)
    pad = torch.where(h >= Hc - 1, torch.zeros_like(h), (Hc - h) // 2).unsqueeze(-1)
    rows = torch.arange(Hc, device=h.device).view(1, 1, Hc)
    col = torch.tensor(COLOURS, device=h.device, dtype=torch.int64)[c.to(torch.int64)].unsqueeze(-1)
    return torch.where(rows < pad, 0xFFFFFF, torch.where(rows < Hc - pad, col, 0x404040))


def main():
    proc, nproc = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert nproc * 2 == WORLD and "RCW_RCCL_LIBRARY" in os.environ
    dist.init_process_group("gloo", rank=proc, world_size=nproc)
    torch.cuda.set_device(0)
    import raycastworlds_jl_amd as RCW
    from oracle import oracle as O

    # the 128-byte id: made once (on the process that holds rank 0), carried by the caller's transport
    uid = torch.zeros(128, dtype=torch.uint8)
    if proc == 0:
        uid = torch.tensor(list(RCW.make_unique_id()), dtype=torch.uint8)
    dist.broadcast(uid, src=0)
    uid = bytes(uid.tolist())

    # the unsharded truth: same generator keyed by GLOBAL agent id, same actions
    O.set_num_threads(4)
    ref = O.OracleBatch(GLOBAL_BATCH, seed=2024, render=False, out_of_bounds=0, **CFG4)
    rng = np.random.default_rng(1)
    actions = rng.integers(1, 5, (STEPS, GLOBAL_BATCH)).astype(np.uint8)
    for s in range(STEPS):
        assert ref.step(actions[s]) == 0

    results, errors = {}, []

    def run_rank(r):
        try:
            torch.cuda.set_device(0)
            sh = RCW.ShardedSingleRoom(GLOBAL_BATCH, rank=r, world=WORLD, device=0, seed=2024, out_of_bounds=0, **CFG4)
            assert (sh.first, sh.count) == (8192 * r, 8192) and sh.env.cfg.agent_id_offset == 8192 * r
            for s in range(STEPS):
                sh.act_(sh.local_slice(actions[s]))
            try:                                               # the reference policy: a reachable BoundsError is an error code
                sh.env.sync()
                raised = False
            except IndexError:
                raised = True
            status = sh.env.world.status
            np.testing.assert_array_equal(status, sh.local_slice(ref.status), err_msg=f"rank {r}: per-agent status")
            assert raised == bool((status != 0).any())
            if raised:
                sh.env.clear_error()
            w = sh.env.world
            np.testing.assert_array_equal(w.player_position_wu.view(np.uint32), sh.local_slice(ref.position).view(np.uint32))
            np.testing.assert_array_equal(w.player_direction_au, sh.local_slice(ref.direction))
            np.testing.assert_array_equal(w.goal_position, sh.local_slice(ref.goal))
            sh.comm_init_abi(unique_id=uid)                    # rcw_comm_init(handle, uid, r, 8)
            ir, iw = ctypes.c_int32(), ctypes.c_int32()
            sh.env._check(sh.env._lib.rcw_comm_info(sh.env._h, ir, iw))
            assert (ir.value, iw.value) == (r, WORLD)
            gh, gc = sh.gather_columns_abi()                   # the global batch's descriptors, on every rank
            sh.env.sync()
            np.testing.assert_array_equal(gh.cpu().numpy(), ref.col_height, err_msg=f"rank {r}: gathered height_line_pu")
            np.testing.assert_array_equal(gc.cpu().numpy(), ref.col_colour, err_msg=f"rank {r}: gathered colour ids")
            out = {"rank": r, "bounds_errors": int((status != 0).sum())}
            if r == 0:
                frames = sh.gather_observations_abi("columns")  # the same two all-gathers + 17 GB of pixels on this rank
                sh.env.sync()
                assert tuple(frames.shape) == (GLOBAL_BATCH, 256, 256)
                for a0 in range(0, GLOBAL_BATCH, 512):
                    want = torch_expand(gh[a0:a0 + 512], gc[a0:a0 + 512]).to(torch.int32)
                    assert torch.equal(frames[a0:a0 + 512].view(torch.int32), want), f"global frames {a0}.. differ from their descriptors"
                # ... and EVERY pixel of the gathered global batch against the oracle's own rendering of the unsharded states
                O.set_num_threads(8)
                for a0 in range(0, GLOBAL_BATCH, 4096):
                    small = O.OracleBatch(4096, seed=0, **CFG4)
                    small.set_state(ref.goal[a0:a0 + 4096], ref.position[a0:a0 + 4096], ref.direction[a0:a0 + 4096])
                    got = frames[a0:a0 + 4096].cpu().numpy().view(np.uint32)
                    assert np.array_equal(got, small.camera_view), f"gathered frames of agents {a0}..{a0 + 4096} differ from the oracle's rendering"
                    small.close()
                out["global_frames_bytes"] = int(frames.numel()) * 4
                del frames
            else:
                gh2, gc2 = sh.gather_columns_abi()             # the other ranks' part in rank 0's gather
                sh.env.sync()
                assert torch.equal(gh2, gh) and torch.equal(gc2, gc)
            results[r] = (sh, out)
        except BaseException as e:                              # noqa: BLE001
            traceback.print_exc()
            errors.append(f"rank {r}: {type(e).__name__}: {e}")

    ranks = [2 * proc, 2 * proc + 1]
    threads = [threading.Thread(target=run_rank, args=(r,)) for r in ranks]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        print("\n".join(errors), file=sys.stderr)
        os._exit(1)                                            # (other ranks may be waiting in a collective: do not linger)
    # rcw_comm_destroy is collective in the stand-in (as ncclCommDestroy may be): both ranks of this process at once
    closers = [threading.Thread(target=results[r][0].close) for r in ranks]
    for t in closers:
        t.start()
    for t in closers:
        t.join()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"process": proc, "ranks": ranks, "world": WORLD, "parity": "ok", "global_batch": GLOBAL_BATCH,
                      "per_rank": [results[r][1] for r in ranks]}))


if __name__ == "__main__":
    main()
