"""N > 1 path on CPU: two and eight ranks over gloo.  The host-side sharding logic
(raycastworlds.jl_amd/sharded.py) runs for real; the engine underneath is swapped for a test
double backed by the CPU oracle (injected through `env_factory` — the product default is the
HIP engine and has no CPU fallback)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

CFG = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=32)
GLOBAL_B = 16
STEPS = 25


class OracleEngine:
    """Test double with the slice of the SingleRoom interface ShardedSingleRoom uses."""

    def __init__(self, batch, agent_id_offset, device, seed=0, **kw):
        from oracle import oracle as O

        self.o = O.OracleBatch(batch, seed=seed, agent_id_offset=agent_id_offset, out_of_bounds=1, **kw)
        self.batch = batch

    def columns_device(self):
        return torch.from_numpy(self.o.col_height.copy()), torch.from_numpy(self.o.col_colour.copy())

    @property
    def camera_view(self):
        return torch.from_numpy(self.o.camera_view.copy())          # torch.uint32, as the engine's alias is

    def expand_columns(self, h, c):
        colours = np.array([0x808080, 0xC0C0C0, 0x800000, 0xC00000], dtype=np.int64)
        h, c = h.numpy().astype(np.int64), c.numpy()
        Hc = 256
        pad = np.where(h >= Hc - 1, 0, (Hc - h) // 2)[..., None]
        rows = np.arange(Hc)[None, None, :]
        col = colours[c][..., None]
        return torch.from_numpy(np.where(rows < pad, 0xFFFFFF, np.where(rows < Hc - pad, col, 0x404040)).astype(np.uint32))

    def close(self):
        self.o.close()


def _patch_act():
    # route the module-level act_/reset_ used by ShardedSingleRoom to the test double
    from raycastworlds_jl_amd import single_room

    single_room.act_ = lambda env, a: env.o.step(np.asarray(a, dtype=np.uint8))
    single_room.reset_ = lambda env, mask=None, seed=None: env.o.reset(mask, 0 if seed is None else seed)


def _worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raycastworlds_jl_amd as RCW

        _patch_act()
        sh = RCW.ShardedSingleRoom(GLOBAL_B, env_factory=OracleEngine, seed=77, **CFG)
        assert (sh.first, sh.count) == (rank * GLOBAL_B // world, GLOBAL_B // world)
        rng = np.random.default_rng(5)
        actions = rng.integers(1, 5, (STEPS, GLOBAL_B)).astype(np.uint8)
        for s in range(STEPS):
            sh.act_(sh.local_slice(actions[s]))
        gh, gc = sh.gather_columns()
        frames_c = sh.gather_observations("columns")
        frames_f = sh.gather_observations("frames")
        assert frames_f.dtype == torch.uint32            # travelled through the process group as int32 (same bytes)
        sh.reset_(seed=5)
        gh2, _ = sh.gather_columns()
        q.put((rank, gh.numpy(), gc.numpy(), frames_c.numpy(), frames_f.numpy(), gh2.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8], ids=["two_ranks", "eight_ranks"])
def test_sharded_ranks_equal_one_unsharded_batch(oracle, world):
    """... with two ranks, and with the eight of BASELINE's cfg-4 (a node's GPU count): rank r owns agents [r B / 8, (r + 1) B / 8)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the unsharded truth
    ref = oracle.OracleBatch(GLOBAL_B, seed=77, out_of_bounds=1, **CFG)
    rng = np.random.default_rng(5)
    actions = rng.integers(1, 5, (STEPS, GLOBAL_B)).astype(np.uint8)
    for s in range(STEPS):
        ref.step(actions[s])
    for rank, gh, gc, frames_c, frames_f, gh2 in results:
        np.testing.assert_array_equal(gh, ref.col_height, err_msg=f"rank {rank} gathered heights")
        np.testing.assert_array_equal(gc, ref.col_colour)
        np.testing.assert_array_equal(frames_c.astype(np.uint32), ref.camera_view, err_msg="columns-mode gather")
        np.testing.assert_array_equal(frames_f.astype(np.uint32), ref.camera_view, err_msg="frames-mode gather")
    ref.reset(seed=5)
    np.testing.assert_array_equal(results[0][5], ref.col_height, err_msg="after sharded reset")


class _RecordingLib:
    """What comm_init_abi needs of the library: rank 0's unique id (a recognisable pattern) and a record of the
    rcw_comm_init arguments."""

    def __init__(self, rank):
        self.rank, self.calls = rank, []

    def rcw_comm_unique_id(self, uid):
        for k in range(len(uid)):
            uid[k] = (7 * k + 3) & 0xFF
        self.calls.append(("unique_id",))
        return 0

    def rcw_comm_init(self, h, uid, rank, world):
        self.calls.append(("init", bytes(uid), rank, world))
        return 0

    def rcw_comm_destroy(self, h):
        return 0


class _AbiEngine(OracleEngine):
    def __init__(self, batch, agent_id_offset, device, seed=0, **kw):
        super().__init__(batch, agent_id_offset, device, seed=seed, **kw)
        self.device, self._h = 0, None
        self._lib = _RecordingLib(dist.get_rank())

    def _check(self, rc):
        assert rc == 0


def _uid_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raycastworlds_jl_amd as RCW

        sh = RCW.ShardedSingleRoom(GLOBAL_B, env_factory=_AbiEngine, seed=1, **CFG)
        sh.comm_init_abi()
        sh.comm_init_abi()                                                   # idempotent
        q.put((rank, sh.env._lib.calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_unique_id_hand_over_between_two_ranks():
    """ShardedSingleRoom.comm_init_abi with two ranks on the CPU (gloo): only rank 0 asks the library for the
    ncclUniqueId, its 128 bytes reach rank 1 through the process group unchanged, and each rank calls rcw_comm_init once
    with its own (rank, world = 2).  (The library side of the same call runs on the GPU with two ranks:
    tests/test_gpu_rccl.py::test_library_transport_with_two_ranks.)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = bytes((7 * k + 3) & 0xFF for k in range(128))
    assert results[0] == [("unique_id",), ("init", want, 0, 2)]
    assert results[1] == [("init", want, 1, 2)]


def _rows_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    rows = bench.gather_rank_rows(dist, world, [0.1 * (rank + 1), 10.0 + rank, 0.011, 0.156 + 0.01 * rank, 0.0])
    warm = bench.gather_rank_rows(dist, world, [0.0])                 # (the warm-up collective in front of the timed region)
    q.put((rank, rows, warm))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_gathers_every_ranks_own_clock():
    """bench.py, N > 1: the ranks stop their own clocks and the figures are gathered AFTERWARDS — one
    all-gather in the flat form both gloo and RCCL take.  Three ranks over gloo on the CPU: every rank sees every rank's row in rank
    order, and the reduction keeps the slowest rank's time and the spread."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    world = 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [[0.1 * (r + 1), 10.0 + r, 0.011, 0.156 + 0.01 * r, 0.0] for r in range(world)]
    for rank, rows, warm in got:
        assert np.allclose(rows, want) and warm == [[0.0]] * world, (rank, rows)
    red = bench.reduce_rank_times(got[0][1], steps=20)
    assert abs(red["dt"] - 0.3) < 1e-12 and abs(red["fill_ms"] - 0.176) < 1e-12 and len(red["per_rank"]) == 3
    assert abs(red["ms_per_step_min"] - 5.0) < 1e-9 and abs(red["ms_per_step_max"] - 15.0) < 1e-9
