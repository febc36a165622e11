"""Parity at BASELINE.json's FULL sizes (cfg-2 4096 x 256, cfg-3 16384 x 512, cfg-4's per-GPU shard 8192 x 256 of
the 65536-agent batch, cfg-5 8192 x 1024),
where stepping the rendering oracle would take minutes:

* per-agent state and per-column descriptors (height_line_pu, colour id) of EVERY agent against
  the oracle run without the pixel fill (cheap: 5 bytes per column);
* the full observation batch against an independent expansion of those descriptors written
  with torch ops on the device (size-independent property: a frame is a pure function of its
  column descriptors, SR:431-440), AND every pixel of every agent against frames the oracle renders from
  the same states (round 4: the oracle renders 2048 agents at a time; rounds 1-3 compared a sample);
* sharding invariance: two engines with agent_id_offset reproduce one engine;
* determinism / idempotence: turn left then right restores the frame bit for bit.
"""
import os

import numpy as np
import pytest

from helpers import CFG1, CFG2, CFG3, CFG4, CFG5

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

COLOURS = [0x808080, 0xC0C0C0, 0x800000, 0xC00000]


def torch_expand(h, c, Hc=256):
    """(B, N) int32 / uint8 descriptors -> (B, N, Hc) int64 pixels, straight from SR:431-440."""
    h = h.to(torch.int64)
    pad = torch.where(h >= Hc - 1, torch.zeros_like(h), (Hc - h) // 2).unsqueeze(-1)
    rows = torch.arange(Hc, device=h.device).view(1, 1, Hc)
    col = torch.tensor(COLOURS, device=h.device, dtype=torch.int64)[c.to(torch.int64)].unsqueeze(-1)
    return torch.where(rows < pad, 0xFFFFFF, torch.where(rows < Hc - pad, col, 0x404040))


def check_every_pixel_against_the_oracle(env, oracle, orc, cfg, chunk=2048, **okw):
    """EVERY frame of the batch against frames rendered by the oracle (not the expansion of the engine's own descriptors): the
    agents' states go, a few thousand at a time, into a rendering oracle (`update_camera_view!` SR:374-444 on the CPU), and the
    device frames of the same agents are compared pixel for pixel.  1-8 GiB of pixels at the BASELINE sizes; the oracle renders
    a chunk in well under a second on the box's cores."""
    for a0 in range(0, env.batch, chunk):
        n = min(chunk, env.batch - a0)
        small = oracle.OracleBatch(n, seed=0, **okw, **cfg)
        small.set_state(orc.goal[a0:a0 + n], orc.position[a0:a0 + n], orc.direction[a0:a0 + n])
        got = env.camera_view_host(a0, n)
        assert np.array_equal(got, small.camera_view), \
            f"frames of agents {a0}..{a0 + n}: first differing agent {a0 + int(np.flatnonzero((got != small.camera_view).reshape(n, -1).any(axis=1))[0])}"
        small.close()


def check_frames_against_descriptors(env, chunk=512):
    obs = env.camera_view.torch()
    h, c = env.columns_device()
    h, c = h.torch(), c.torch()
    for a0 in range(0, env.batch, chunk):
        want = torch_expand(h[a0:a0 + chunk], c[a0:a0 + chunk], Hc=obs.shape[-1]).to(torch.int32)
        got = obs[a0:a0 + chunk].view(torch.int32)
        assert torch.equal(got, want), f"frames of agents {a0}..{a0 + chunk} differ from their descriptors"


def settle_bounds_errors(env, orc, where=""):
    """Wait for the engine under the REFERENCE policy (out_of_bounds = 0: a forward move that lands exactly on x = H-1 or
    y = W-1 indexes tile H+1 in is_player_colliding, CD:30-35 — a BoundsError in the reference): the agents it happened
    to must be the oracle's, their status words equal; then clear both sides (the error is sticky).  Returns how many."""
    try:
        env.sync()
        raised = False
    except IndexError:
        raised = True
    status = env.world.status
    np.testing.assert_array_equal(status, orc.status, err_msg=f"per-agent status {where}")
    assert raised == bool((status != 0).any())
    if raised:
        env.clear_error()
        orc.clear_status()
    return int((status != 0).sum())


@pytest.mark.parametrize("cfg,batch,steps,oob", [(CFG2, 4096, 24, 1), (CFG3, 16384, 12, 0), (CFG4, 8192, 6, 1), (CFG5, 8192, 4, 1)],
                         ids=["cfg2_4096x256", "cfg3_16384x512_reference_policy", "cfg4_shard_8192x256", "cfg5_8192x1024"])
def test_full_size_parity(rcw, oracle, cfg, batch, steps, oob):
    env = rcw.SingleRoomModule.SingleRoom(batch=batch, seed=2024, out_of_bounds=oob, **cfg)
    orc = oracle.OracleBatch(batch, seed=2024, render=False, out_of_bounds=oob, **cfg)
    rng = np.random.default_rng(1)
    sample = rng.choice(batch, 16, replace=False)
    for s in range(steps):
        a = rng.integers(1, 5, batch).astype(np.uint8)
        rcw.act_(env, a)
        assert orc.step(a) == 0
    if oob == 0:
        settle_bounds_errors(env, orc, "after the rollout")
    w = env.world
    np.testing.assert_array_equal(w.player_position_wu.view(np.uint32), orc.position.view(np.uint32))
    np.testing.assert_array_equal(w.player_direction_au, orc.direction)
    np.testing.assert_array_equal(w.reward, orc.reward)
    np.testing.assert_array_equal(w.done.astype(np.uint8), orc.done)
    np.testing.assert_array_equal(w.tile_map_chunks, orc.tile_map_chunks())
    h, c = env.columns()
    np.testing.assert_array_equal(h, orc.col_height)
    np.testing.assert_array_equal(c, orc.col_colour)
    check_frames_against_descriptors(env)
    # ... and every pixel of every agent against the oracle's own rendering of the same states (round 4; rounds 1-3: a sample of 16)
    check_every_pixel_against_the_oracle(env, oracle, orc, cfg)
    # a masked reset (the fill kernel's mask path: whole chunks of masked-out agents are skipped) and one more step
    mask = (rng.random(batch) < 0.35).astype(np.uint8)
    rcw.reset_(env, mask=mask, seed=321); orc.reset(mask=mask, seed=321)
    h, c = env.columns()
    np.testing.assert_array_equal(h, orc.col_height)
    np.testing.assert_array_equal(c, orc.col_colour)
    check_frames_against_descriptors(env)
    a = rng.integers(1, 5, batch).astype(np.uint8)
    rcw.act_(env, a)
    assert orc.step(a) == 0
    if oob == 0:
        settle_bounds_errors(env, orc, "after the masked reset's step")
    np.testing.assert_array_equal(env.world.player_position_wu.view(np.uint32), orc.position.view(np.uint32))
    h, c = env.columns()
    np.testing.assert_array_equal(h, orc.col_height)
    check_frames_against_descriptors(env)
    env.close()


def test_bench_configuration_at_full_size(rcw, oracle):
    """bench.py's exact configuration, parity-checked at its full size: cfg-2 (8x8, 256 columns), 4096 agents, the
    REFERENCE policy for the reachable BoundsError (out_of_bounds = 0, CD:30-35), auto-reset on, uniform 1..4 actions
    made on the device as bench.py makes them (bench.py:171-177), 400 steps with no host synchronisation in between.
    Against the non-rendering oracle given the same actions: position bits, heading, goal, episode count, reward, done,
    tile map and the descriptors of every agent — and the per-agent STATUS words: the agents that hit the reference's
    BoundsError must be the same ones, left untouched by that action.  Frames: all of them against the expansion of
    their descriptors, and all of them — 1 GiB, every pixel — against the rendering oracle given the final states."""
    B, steps = 4096, 400
    env = rcw.SingleRoomModule.SingleRoom(batch=B, seed=0, auto_reset=True, out_of_bounds=0, **CFG2)
    orc = oracle.OracleBatch(B, seed=0, render=False, auto_reset=1, out_of_bounds=0, **CFG2)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234)
    actions = torch.randint(1, 5, (steps, B), dtype=torch.uint8, device="cuda", generator=gen)
    host = actions.cpu().numpy()
    hit = np.zeros(B, dtype=bool)
    for half in range(2):
        for s in range(half * steps // 2, (half + 1) * steps // 2):
            rcw.act_(env, actions[s])                     # device actions: rcw_step_device, nothing waits
            assert orc.step(host[s]) == 0
        hit |= orc.status != 0
        n = settle_bounds_errors(env, orc, f"after {(half + 1) * steps // 2} steps")
        w = env.world
        np.testing.assert_array_equal(w.player_position_wu.view(np.uint32), orc.position.view(np.uint32))
        np.testing.assert_array_equal(w.player_direction_au, orc.direction)
        np.testing.assert_array_equal(w.goal_position, orc.goal)
        np.testing.assert_array_equal(w.episode, orc.episode)
        np.testing.assert_array_equal(w.reward, orc.reward)
        np.testing.assert_array_equal(w.done.astype(np.uint8), orc.done)
        np.testing.assert_array_equal(w.tile_map_chunks, orc.tile_map_chunks())
        h, c = env.columns()
        np.testing.assert_array_equal(h, orc.col_height)
        np.testing.assert_array_equal(c, orc.col_colour)
        check_frames_against_descriptors(env)
    assert hit.any(), "no agent reached the reference's BoundsError in 400 steps x 4096 agents: the policy went untested"
    assert (orc.episode >= 3).any(), "no agent got into a third episode"
    check_every_pixel_against_the_oracle(env, oracle, orc, CFG2)           # all 4096 frames, incl. the agents that hit the error
    env.close()


@pytest.mark.parametrize("hc,cfg,batch", [(100, CFG2, 8192), (84, CFG2, 9001), (250, CFG3, 2048), (40, CFG5, 6000), (27, CFG2, 30000)],
                         ids=["100_rows_8192x256", "84_rows_9001x256", "250_rows_2048x512", "40_rows_6000x1024", "27_rows_30000x256"])
def test_full_size_other_camera_heights(rcw, oracle, hc, cfg, batch):
    """rcw_fill_flat_kernel at 0.8–1 GiB of pixels a step (height_camera_view_pu other than a power-of-two multiple of 64,
    SR:271): 100 and 84 rows (16-byte groups inside one column; 9001 agents: the batch ends inside a chunk), 250 and 40 rows
    (groups that straddle columns; up to eight columns a chunk).  State and descriptors of every agent against the
    non-rendering oracle, ALL frames against the torch expansion of the descriptors, sampled frames against the oracle."""
    env = rcw.SingleRoomModule.SingleRoom(batch=batch, seed=11, out_of_bounds=1, height_camera_view_pu=hc, **cfg)
    assert env.fill_kernel_name() == "rcw_fill_flat_kernel"
    orc = oracle.OracleBatch(batch, seed=11, render=False, out_of_bounds=1, height_camera_view_pu=hc, **cfg)
    rng = np.random.default_rng(6)
    for s in range(5):
        a = rng.integers(1, 5, batch).astype(np.uint8)
        rcw.act_(env, a)
        assert orc.step(a) == 0
    h, c = env.columns()
    np.testing.assert_array_equal(h, orc.col_height)
    np.testing.assert_array_equal(c, orc.col_colour)
    check_frames_against_descriptors(env, chunk=256)
    mask = (rng.random(batch) < 0.3).astype(np.uint8)
    rcw.reset_(env, mask=mask, seed=5); orc.reset(mask=mask, seed=5)
    a = rng.integers(1, 5, batch).astype(np.uint8)
    rcw.act_(env, a); orc.step(a)
    check_frames_against_descriptors(env, chunk=256)
    sample = np.unique(np.concatenate([[0, 1, batch // 2, batch - 2, batch - 1], rng.choice(batch, 11, replace=False)]))
    small = oracle.OracleBatch(len(sample), seed=0, height_camera_view_pu=hc, **cfg)
    small.set_state(orc.goal[sample], orc.position[sample], orc.direction[sample])
    got = np.stack([env.camera_view_host(int(i), 1)[0] for i in sample])
    np.testing.assert_array_equal(got, small.camera_view)
    env.close()


def _skip_unless_free(nbytes):
    free, _ = torch.cuda.mem_get_info()
    if free < nbytes * 1.15 + (4 << 30):
        pytest.skip(f"needs {nbytes / 2**30:.0f} GiB of free device memory, {free / 2**30:.0f} GiB are free")


@pytest.mark.parametrize("hc,cfg,batch", [(256, CFG2, 70000), (100, CFG2, 170000), (250, CFG2, 70001), (256, CFG3, 400000)],
                         ids=["256_rows_18GB", "100_rows_17GB", "250_rows_18GB", "256_rows_210GB"])
def test_batches_beyond_2_32_pixels(rcw, oracle, hc, cfg, batch):
    """A frame batch of more than 2^32 pixels (17–18 GB of the card's 288, and once 210 GB — the 16×16 room's 512-column frames
    of 400,000 agents: byte offsets and pixel offsets need 64 bits, chunk and column ids still fit 32) through rcw_fill256_kernel and rcw_fill_flat_kernel: descriptors of every agent against the
    oracle, every frame against the expansion of its descriptors, the frames either side of the 2^32-pixel and
    2^32-byte marks and the last ones against the rendering oracle."""
    px_agent = hc * cfg["num_rays"]
    _skip_unless_free(4 * px_agent * batch)
    env = rcw.SingleRoomModule.SingleRoom(batch=batch, seed=13, out_of_bounds=1, height_camera_view_pu=hc, **cfg)
    assert batch * px_agent > 1 << 32
    orc = oracle.OracleBatch(batch, seed=13, render=False, out_of_bounds=1, height_camera_view_pu=hc, **cfg)
    rng = np.random.default_rng(8)
    for s in range(2):
        a = rng.integers(1, 5, batch).astype(np.uint8)
        rcw.act_(env, a)
        assert orc.step(a) == 0
    h, c = env.columns()
    np.testing.assert_array_equal(h, orc.col_height)
    np.testing.assert_array_equal(c, orc.col_colour)
    check_frames_against_descriptors(env, chunk=1024)
    marks = [(1 << 32) // px_agent, (1 << 30) // px_agent, (1 << 31) // px_agent]
    sample = np.unique(np.clip(np.concatenate([[m - 1, m, m + 1] for m in marks] + [[0, batch - 2, batch - 1]]), 0, batch - 1))
    small = oracle.OracleBatch(len(sample), seed=0, height_camera_view_pu=hc, **cfg)
    small.set_state(orc.goal[sample], orc.position[sample], orc.direction[sample])
    got = np.stack([env.camera_view_host(int(i), 1)[0] for i in sample])
    np.testing.assert_array_equal(got, small.camera_view)
    env.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("cfg,batch,form,pu", [(CFG2, 4096, "two-kernels", 32), (dict(height_tile_map_tu=8, width_tile_map_tu=16, num_rays=512), 2048, "two-kernels", 32),
                                               (CFG4, 1024, "two-kernels", 32), (CFG4, 4100, "two-kernels", 32),
                                               (CFG2, 24576, "two-kernels", 13), (CFG2, 3000, "two-kernels", 20),
                                               (dict(height_tile_map_tu=24, width_tile_map_tu=24, num_rays=256), 1999, "two-kernels", 12),
                                               (dict(height_tile_map_tu=24, width_tile_map_tu=24, num_rays=256), 455, "two-kernels", 32), (CFG5, 256, "two-kernels", 32),
                                               (CFG5, 64, "two-kernels", 32), (dict(height_tile_map_tu=24, width_tile_map_tu=24, num_rays=256), 130, "two-kernels", 32),
                                               (dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64), 70000, "two-kernels", 32),
                                               (dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=32), 420000, "two-kernels", 13)],
                         ids=["cfg2_4096", "reference_default_2048", "cfg4_1024", "cfg4_4100_in_runs",
                              "cfg2_24576_of_104x104", "cfg2_3000_of_320x320", "room24_1999_of_288x288", "room24_455_of_768x768", "cfg5_256_of_1024x1024",
                              "cfg5_64_in_four_parts", "room24_130_in_two_parts",
                              "beyond_2_32_pixels_of_256x256", "beyond_2_32_pixels_of_104x104"])
def test_full_size_top_view(rcw, oracle, cfg, batch, form, pu):
    """The opt-in top view at full batch sizes (1 GiB of pixels a step: the two-kernel form's store kernel sweeps its
    window 16 times; 4 GiB of 512² px images: the batch goes in four runs of agents, 4100 does not divide evenly).  State of every agent against the non-rendering oracle; both images of a sample of agents (the
    first and last, around the middle, random ones) against a small rendering oracle given the same states; and over
    ALL images a size-independent property: no pixel outside the six colours update_top_view! can write (SR:288-290,
    SR:364-367) — a chunk the store kernel skipped or wrote twice with stale descriptors would show.  The last three
    cases go through rcw_top_store_flat_kernel (13-, 20- and 12-pixel tiles; ≈ 1 GiB, 1.1 GiB and 0.6 GiB of pixels in images whose
    size is no multiple of a chunk, so nearly every chunk holds a border between columns or images)."""
    _skip_unless_free(4 * batch * (cfg["height_tile_map_tu"] * cfg["width_tile_map_tu"] * pu * pu + 256 * cfg["num_rays"]))
    env = rcw.SingleRoomModule.SingleRoom(batch=batch, seed=77, out_of_bounds=1, render_top_view=True, pu_per_tu=pu, **cfg)
    assert env.top_view_form() == form
    orc = oracle.OracleBatch(batch, seed=77, render=False, out_of_bounds=1, **cfg)
    rng = np.random.default_rng(3)
    for s in range(5):
        a = rng.integers(1, 5, batch).astype(np.uint8)
        rcw.act_(env, a)
        assert orc.step(a) == 0
    def verify(sample_seed):
        rng_s = np.random.default_rng(sample_seed)
        w = env.world
        np.testing.assert_array_equal(w.player_position_wu.view(np.uint32), orc.position.view(np.uint32))
        np.testing.assert_array_equal(w.player_direction_au, orc.direction)
        sample = np.unique(np.concatenate([[0, 1, 3, 4, batch // 2 - 1, batch // 2, batch - 2, batch - 1], rng_s.choice(batch, 16, replace=False)]))
        small = oracle.OracleBatch(len(sample), seed=0, render_top_view=1, pu_per_tu=pu, **cfg)
        small.set_state(orc.goal[sample], orc.position[sample], orc.direction[sample])
        got = np.stack([env.top_view_host(int(i), 1)[0] for i in sample])
        np.testing.assert_array_equal(got, small.top_view)
        cam = np.stack([env.camera_view_host(int(i), 1)[0] for i in sample])
        np.testing.assert_array_equal(cam, small.camera_view)
        tv = env.top_view.torch().view(torch.int32)
        palette = torch.tensor([0x000000, 0xFFFFFF, 0xFF0000, 0xCCCCCC, 0x808080, 0xC0C0C0], dtype=torch.int32, device=tv.device)
        for a0 in range(0, batch, 256):
            assert bool(torch.isin(tv[a0:a0 + 256], palette).all()), f"a pixel outside the palette in agents {a0}.."
        # ... and every pixel that is neither ray grey nor circle grey is the tile layer's, rebuilt here with torch ops from
        # the goal positions alone (SR:348-369: walls on the room's border, the goal tile, 0xCCCCCC tile borders)
        H, W = cfg["height_tile_map_tu"], cfg["width_tile_map_tu"]
        goal = torch.from_numpy(np.ascontiguousarray(w.goal_position)).to(tv.device).to(torch.int64)        # 1-based (i, j)
        ii = torch.arange(H, device=tv.device).view(1, 1, H)
        jj = torch.arange(W, device=tv.device).view(1, W, 1)
        wall = (ii == 0) | (ii == H - 1) | (jj == 0) | (jj == W - 1)
        edge_i = torch.arange(H * pu, device=tv.device) % pu
        edge_j = torch.arange(W * pu, device=tv.device) % pu
        edge = ((edge_j == 0) | (edge_j == pu - 1)).view(1, W * pu, 1) | ((edge_i == 0) | (edge_i == pu - 1)).view(1, 1, H * pu)
        step = max(1, (64 << 20) // (H * W * pu * pu * 4))
        rays_seen = 0
        for a0 in range(0, batch, step):
            g = goal[a0:a0 + step]
            is_goal = (ii == (g[:, 0] - 1).view(-1, 1, 1)) & (jj == (g[:, 1] - 1).view(-1, 1, 1))
            tiles = torch.where(wall, 0xFFFFFF, torch.where(is_goal, 0xFF0000, 0)).to(torch.int32)
            layer = tiles.repeat_interleave(pu, dim=1).repeat_interleave(pu, dim=2)
            layer = torch.where(edge, torch.tensor(0xCCCCCC, dtype=torch.int32, device=tv.device), layer)
            got_px = tv[a0:a0 + step].reshape(layer.shape)
            drawn = (got_px == 0x808080) | (got_px == 0xC0C0C0)
            assert bool(((got_px == layer) | drawn).all()), f"tile layer differs in agents {a0}.."
            per_agent = drawn.flatten(1).sum(1)
            assert int(per_agent.min()) >= 4, "an image without a ray or circle pixel"
            rays_seen += int(per_agent.sum())
        assert rays_seen > batch * pu

    def every_pixel():
        """... and, where the batch's top views are within 4 GiB, EVERY pixel of both images of EVERY agent against the oracle's
        rendering of the same states (SimpleDraw's rasterisers restated: 2048 agents at a time)."""
        px = cfg["height_tile_map_tu"] * cfg["width_tile_map_tu"] * pu * pu
        if 4 * px * batch > (4 << 30):
            return
        for a0 in range(0, batch, 2048):
            n = min(2048, batch - a0)
            small = oracle.OracleBatch(n, seed=0, render_top_view=1, pu_per_tu=pu, **cfg)
            small.set_state(orc.goal[a0:a0 + n], orc.position[a0:a0 + n], orc.direction[a0:a0 + n])
            assert np.array_equal(env.top_view_host(a0, n), small.top_view), f"top views of agents {a0}..{a0 + n} differ from the oracle's"
            assert np.array_equal(env.camera_view_host(a0, n), small.camera_view), f"camera views of agents {a0}..{a0 + n} differ from the oracle's"
            small.close()

    verify(1)
    every_pixel()
    # a MASKED reset next (the draw kernel then runs for the masked agents only, the store kernel writes their pixels only —
    # chunk by chunk, per pixel where an image border or a run's border falls inside a chunk), one more step, and the same again
    mask = (rng.random(batch) < 0.4).astype(np.uint8)
    mask[[0, batch - 1]] = 1; mask[[1, batch - 2]] = 0
    rcw.reset_(env, mask=mask, seed=123); orc.reset(mask=mask, seed=123)
    verify(2)
    a = rng.integers(1, 5, batch).astype(np.uint8)
    rcw.act_(env, a)
    assert orc.step(a) == 0
    verify(3)
    every_pixel()
    if batch >= 12288 and 4 * cfg["height_tile_map_tu"] * cfg["width_tile_map_tu"] * pu * pu * batch <= (4 << 30):
        # update_top_view!(env) ALONE on a batch of tens of thousands of small images: the draw kernel in workgroups of 64 / 128 threads
        env.sync()
        env.top_view.torch().zero_()
        torch.cuda.synchronize()
        rcw.update_top_view_(env)
        verify(4)                                                            # (a sample of agents pixel by pixel, every image's palette and tile layer)
    env.close()


def test_sharding_invariance_on_device(rcw):
    kw = dict(seed=9, out_of_bounds=1, **CFG2)
    whole = rcw.SingleRoomModule.SingleRoom(batch=512, **kw)
    lo = rcw.SingleRoomModule.SingleRoom(batch=256, agent_id_offset=0, **kw)
    hi = rcw.SingleRoomModule.SingleRoom(batch=256, agent_id_offset=256, **kw)
    rng = np.random.default_rng(2)
    for s in range(40):
        a = rng.integers(1, 5, 512).astype(np.uint8)
        rcw.act_(whole, a); rcw.act_(lo, a[:256]); rcw.act_(hi, a[256:])
    for e in (whole, lo, hi):
        rcw.reset_(e, seed=31)                      # exercises the on-device generator again
    np.testing.assert_array_equal(whole.world.goal_position,
                                  np.concatenate([lo.world.goal_position, hi.world.goal_position]))
    np.testing.assert_array_equal(whole.camera_view_host(),
                                  np.concatenate([lo.camera_view_host(), hi.camera_view_host()]))
    for e in (whole, lo, hi):
        e.close()


def test_turn_left_then_right_restores_the_frame(rcw):
    env = rcw.SingleRoomModule.SingleRoom(batch=1024, seed=4, **CFG2)
    before = env.camera_view.torch().clone()
    rcw.act_(env, 3)
    middle = env.camera_view.torch().clone()
    rcw.act_(env, 4)
    after = env.camera_view.torch()
    assert torch.equal(before.view(torch.int32), after.view(torch.int32))
    assert not torch.equal(before.view(torch.int32), middle.view(torch.int32))
    env.close()


def test_sharded_wrapper_world_of_one(rcw, oracle):
    """ShardedSingleRoom on one rank: stepping is local, and the compact gather + on-device expansion
    (rcw_expand_columns) reproduces the engine's own frames."""
    sh = rcw.ShardedSingleRoom(256, rank=0, world=1, device=0, seed=6, out_of_bounds=1, **CFG2)
    orc = oracle.OracleBatch(256, seed=6, out_of_bounds=1, **CFG2)
    rng = np.random.default_rng(8)
    for s in range(30):
        a = rng.integers(1, 5, 256).astype(np.uint8)
        sh.act_(sh.local_slice(a))
        orc.step(a)
    frames_c = sh.gather_observations("columns")
    frames_f = sh.gather_observations("frames")
    assert torch.equal(frames_c.view(torch.int32), frames_f.view(torch.int32))
    np.testing.assert_array_equal(frames_c.cpu().numpy().view(np.uint32), orc.camera_view)
    sh.close()


def test_long_rollout_with_auto_reset_stays_in_parity(rcw, oracle):
    """5000 steps x 512 agents with auto-reset: thousands of episodes end and restart on the device;
    state, episode counters and descriptors are compared with the oracle every 500 steps."""
    kw = dict(seed=12, auto_reset=True, out_of_bounds=1, **CFG1)
    env = rcw.SingleRoomModule.SingleRoom(batch=512, **kw)
    orc = oracle.OracleBatch(512, seed=12, render=False, auto_reset=1, out_of_bounds=1, **CFG1)
    rng = np.random.default_rng(3)
    for s in range(5000):
        a = rng.integers(1, 5, 512).astype(np.uint8)
        rcw.act_(env, a)
        orc.step(a)
        if s % 500 == 499:
            w = env.world
            np.testing.assert_array_equal(w.episode, orc.episode, err_msg=f"episodes at step {s}")
            np.testing.assert_array_equal(w.player_position_wu.view(np.uint32), orc.position.view(np.uint32))
            np.testing.assert_array_equal(w.player_direction_au, orc.direction)
            np.testing.assert_array_equal(w.goal_position, orc.goal)
            h, c = env.columns()
            np.testing.assert_array_equal(h, orc.col_height)
            np.testing.assert_array_equal(c, orc.col_colour)
    assert int(env.world.episode.sum()) > 512 + 100, "expected hundreds of episode restarts"
    check_frames_against_descriptors(env)
    env.close()
