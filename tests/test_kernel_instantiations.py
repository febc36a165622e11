"""Every kernel instantiation the shipped library carries has a parity case on the GPU.

The library picks a template instantiation from the geometry (camera height -> rcw_fill_flat_kernel<ALIGNED, K>; top-view image
height and pixel scale -> rcw_top_store_flat_kernel<STRADDLE, NARROW, K>; world-unit type and the two unpinned cast_ray switches ->
<T, TIE, DIST> of the casting / drawing kernels).  The rules are restated here, the case lists below are checked against the
kernels of the shipped build's ISA (CPU), and each case runs against the oracle (GPU).  tools/gpu_round.sh then counts, under
rocprofv3, which instantiations the suite really launched (profiles/r04_kernel_census.txt)."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import CFG1, CFG2, assert_state_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# camera heights (height_camera_view_pu, SR:271) that take rcw_fill_flat_kernel: one for every (H_cam % 4 == 0, K), K = 254 // H_cam + 2
FILL_HEIGHTS = (255, 260, 129, 132, 85, 88, 65, 68, 51, 52, 43, 44, 37, 40, 33, 36, 29, 26, 28, 25, 24)
# (map rows, map columns, pixels a tile) that take rcw_top_store_flat_kernel: one for every reachable (pu % 4 != 0, pu >= 19, K), K = 251 // (H pu) + 2
TOP_FLAT = ((21, 4, 12), (11, 4, 12), (7, 4, 12), (6, 4, 12), (5, 4, 12), (4, 4, 12), (13, 4, 20), (7, 4, 20), (5, 4, 20), (4, 4, 20), (3, 5, 20),
            (28, 4, 9), (16, 4, 9), (8, 4, 11), (8, 4, 9), (6, 4, 10), (4, 4, 11), (12, 4, 21), (8, 4, 19), (4, 4, 21), (4, 4, 19))


def _b(x):
    return "true" if x else "false"


def fill_flat_label(hc):
    return f"rcw_fill_flat_kernel<{_b(hc % 4 == 0)}, {254 // hc + 2}, false>"      # (the third parameter: two wavefronts to a slot — the development build's only)


def top_flat_label(H, W, pu):
    return f"rcw_top_store_flat_kernel<{_b(pu % 4 != 0)}, {_b(pu >= 19)}, {251 // (H * pu) + 2}>"


def shipped_kernels():
    csrc = os.path.join(ROOT, "raycastworlds.jl_amd", "csrc")
    res = subprocess.run(["make", "-C", csrc, "asm"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    text = open(os.path.join(ROOT, "raycastworlds.jl_amd", "lib", "asm", "rcw_kernels.s")).read()
    mangled = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, re.M)
    out = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
    names = set()
    for n in out:
        n = n.replace("(anonymous namespace)::", "")
        n = re.sub(r"^void ", "", n)
        names.add(re.sub(r"\(.*$", "", n))
    return names


def test_the_case_lists_cover_every_shipped_instantiation_of_the_flat_kernels():
    """(CPU) the instantiations of the two geometry-selected families in the shipped build's ISA are exactly the ones the case
    lists aim at: none the GPU test below would miss, none compiled in that no geometry reaches."""
    names = shipped_kernels()
    fill = {n for n in names if n.startswith("rcw_fill_flat_kernel<")}
    top = {n for n in names if n.startswith("rcw_top_store_flat_kernel<")}
    assert fill == {fill_flat_label(hc) for hc in FILL_HEIGHTS}, fill ^ {fill_flat_label(hc) for hc in FILL_HEIGHTS}
    assert top == {top_flat_label(*g) for g in TOP_FLAT}, top ^ {top_flat_label(*g) for g in TOP_FLAT}
    assert len(FILL_HEIGHTS) == len(fill) and len(TOP_FLAT) == len(top)
    # no plain-store variant, no 128-row units kernel: the development build's only
    assert not [n for n in names if re.match(r"rcw_(fill256|top_store|top_store_units)_kernel<true", n)], names
    assert "rcw_top_store_units_kernel<false, 2>" not in names
    assert not [n for n in names if re.match(r"rcw_fill_flat_kernel<\w+, \d+, true>", n)], names
    # the step's kernels: <T, TIE, DIST> of the cast kernel, <T, TIE, DIST, WAVE> of the one-launch step and of what primes its slots
    for family, count in (("rcw_cast_kernel<", 8), ("rcw_fill256_cast_kernel<", 16), ("rcw_fill_window_cast_kernel<", 16), ("rcw_cast_successors_kernel<", 16)):
        assert len([n for n in names if n.startswith(family)]) == count, (family, sorted(n for n in names if n.startswith(family)))


def _steps(rcw, env, orc, rng, n, top):
    for _ in range(n):
        a = rng.integers(1, 5, env.batch).astype(np.uint8)
        rcw.act_(env, a)
        assert orc.step(a) == 0
    assert_state_equal(env, orc, where="rollout")
    if top:
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)


def _make(rcw, oracle, batch, seed, **kw):
    env = rcw.SingleRoomModule.SingleRoom(batch=batch, seed=seed, **kw)
    okw = {k: v for k, v in kw.items() if k not in ("T", "auto_reset", "library")}
    okw["auto_reset"] = 1 if kw.get("auto_reset") else 0
    if kw.get("T") == "Float64":
        okw["world_unit_bits"] = 64
    return env, oracle.OracleBatch(batch, seed=seed, **okw)


@pytest.mark.gpu
def test_every_flat_fill_instantiation(rcw, oracle):
    """rcw_fill_flat_kernel<ALIGNED, K> for K = 2 .. 12 and both alignments (29-31 rows, K = 10, are never a multiple of 4): 33 view
    columns, 5 agents — a batch that is not a whole number of 256-pixel chunks — with a masked reset in between."""
    rng = np.random.default_rng(5)
    for hc in FILL_HEIGHTS:
        env, orc = _make(rcw, oracle, 5, 11, auto_reset=True, out_of_bounds=1, height_camera_view_pu=hc, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=33)
        assert env.fill_kernel_name() == "rcw_fill_flat_kernel", hc
        _steps(rcw, env, orc, rng, 4, False)
        mask = np.array([1, 0, 1, 1, 0], dtype=np.uint8)
        rcw.reset_(env, mask=mask, seed=3); orc.reset(mask=mask, seed=3)
        _steps(rcw, env, orc, rng, 3, False)
        env.close()


@pytest.mark.gpu
def test_every_flat_top_store_instantiation(rcw, oracle):
    """rcw_top_store_flat_kernel<STRADDLE, NARROW, K>: the 21 combinations a geometry can reach (pixel scales that are / are not a
    multiple of 4, below / from 19 pixels a tile, 2 .. 7 image columns a chunk), two-kernel form, masked reset in between."""
    rng = np.random.default_rng(6)
    for H, W, pu in TOP_FLAT:
        env, orc = _make(rcw, oracle, 5, 12, auto_reset=True, out_of_bounds=1, render_top_view=True, pu_per_tu=pu, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=40)
        env.set_top_view_form("two-kernels")
        assert env.top_view_form() == "two-kernels", (H, W, pu)
        _steps(rcw, env, orc, rng, 4, True)
        mask = np.array([0, 1, 1, 0, 1], dtype=np.uint8)
        rcw.reset_(env, mask=mask, seed=4); orc.reset(mask=mask, seed=4)
        _steps(rcw, env, orc, rng, 3, True)
        rcw.update_top_view_(env)                                            # (stand-alone: draw -> store back to back where pu % 4 != 0)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        env.close()


@pytest.mark.gpu
def test_grid_stride_fill_kernel(rcw, oracle):
    """rcw_fill_any_kernel<ALIGNED>: what is left to it — camera views below 24 rows with more than 8192 view columns."""
    rng = np.random.default_rng(7)
    for hc in (8, 10):
        env, orc = _make(rcw, oracle, 2, 13, out_of_bounds=1, height_camera_view_pu=hc, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=8200)
        assert env.fill_kernel_name() == "rcw_fill_any_kernel", hc
        _steps(rcw, env, orc, rng, 3, False)
        mask = np.array([0, 1], dtype=np.uint8)
        rcw.reset_(env, mask=mask, seed=5); orc.reset(mask=mask, seed=5)
        _steps(rcw, env, orc, rng, 2, False)
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T", ["Float32", "Float64"])
def test_top_view_forms_under_every_unpinned_switch(rcw, oracle, T):
    """<T, TIE, DIST> of the kernels that draw the top view — the in-place kernel, the one-kernel ring, the draw kernel alone and
    inside the camera fill's launch (rcw_fill256_draw_kernel) — under all four settings of the two unpinned cast_ray switches."""
    rng = np.random.default_rng(8)
    for tie in (0, 1):
        for dist in (0, 1):
            for form in ("in-place", "one-kernel", "two-kernels"):
                env, orc = _make(rcw, oracle, 4, 14, T=T, out_of_bounds=1, render_top_view=True, pu_per_tu=32, dda_tie_break=tie, dda_distance=dist, **CFG2)
                env.set_top_view_form(form)
                assert env.top_view_form() == form
                _steps(rcw, env, orc, rng, 3, True)
                rcw.update_top_view_(env)
                np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
                env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T", ["Float32", "Float64"])
def test_step_kernels_under_every_unpinned_switch(rcw, oracle, T):
    """<T, TIE, DIST> of rcw_cast_kernel (the two-launch step) and <T, TIE, DIST, WAVE> of rcw_fill256_cast_kernel / rcw_fill_window_cast_kernel /
    rcw_cast_successors_kernel (the one-launch step at 256 rows, at the window's other heights, and what primes its slots; WAVE: a wavefront per agent up to 256 view columns, a workgroup per agent beyond): all four
    settings of the two unpinned cast_ray switches, both casting shapes, both forms, a masked reset in between, auto_reset on."""
    rng = np.random.default_rng(12)
    for tie in (0, 1):
        for dist in (0, 1):
            for N in (96, 300):
                for form, hc in (("one-launch", 256), ("one-launch", 128), ("two-launches", 256)):   # (128 rows: rcw_fill_window_cast_kernel)
                    env, orc = _make(rcw, oracle, 7, 18, T=T, auto_reset=True, out_of_bounds=1, dda_tie_break=tie, dda_distance=dist,
                                     height_tile_map_tu=7, width_tile_map_tu=9, num_rays=N, height_camera_view_pu=hc)
                    env.set_step_form(form)
                    assert env.step_form() == form
                    if form == "one-launch":
                        assert env.fill_kernel_name() == ("rcw_fill256_cast_kernel" if hc == 256 else "rcw_fill_window_cast_kernel")
                    _steps(rcw, env, orc, rng, 5, False)
                    mask = np.array([1, 0, 1, 1, 0, 0, 1], dtype=np.uint8)
                    rcw.reset_(env, mask=mask, seed=8); orc.reset(mask=mask, seed=8)
                    _steps(rcw, env, orc, rng, 4, False)
                    env.close()


@pytest.mark.gpu
def test_top_view_planner_sweep(rcw, oracle):
    """Every pixel scale from 8 to 40 on maps of 3, 4, 5, 7, 8 and 10 tile rows (198 geometries; whatever form the library's rule picks for each —
    in-place, one kernel, units, flat, 256-row chunks — and the fused or side-stream drawing): two steps and a stand-alone redraw
    against the oracle.  A systematic walk along the rule's boundaries, where the random fuzzers only sample."""
    rng = np.random.default_rng(9)
    forms = {}
    for H in (3, 4, 5, 7, 8, 10):
        for pu in range(8, 41):
            env, orc = _make(rcw, oracle, 3, 15, out_of_bounds=1, render_top_view=True, pu_per_tu=pu, height_tile_map_tu=H, width_tile_map_tu=4, num_rays=24)
            forms[env.top_view_form()] = forms.get(env.top_view_form(), 0) + 1
            _steps(rcw, env, orc, rng, 2, True)
            rcw.update_top_view_(env)
            np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"H {H} pu {pu}")
            env.close()
    assert forms.get("two-kernels", 0) >= 80 and forms.get("one-kernel", 0) >= 80 and sum(forms.values()) == 198, forms


@pytest.mark.gpu
def test_camera_height_sweep(rcw, oracle):
    """Every camera height from 1 to 72 rows and around the window kernels' heights (126-130, 254-258, 510-514): whatever kernel
    the rule picks (frame per workgroup below 24 rows, the flat kernel with up to twelve columns a chunk, the moving window), two
    steps and a masked reset against the oracle, at a column count (25) that leaves the batch's last chunk short."""
    rng = np.random.default_rng(10)
    names = {}
    for hc in list(range(1, 73)) + [126, 127, 128, 129, 130, 254, 255, 256, 257, 258, 510, 511, 512, 513, 514]:
        env, orc = _make(rcw, oracle, 3, 16, out_of_bounds=1, height_camera_view_pu=hc, height_tile_map_tu=6, width_tile_map_tu=7, num_rays=25)
        names[env.fill_kernel_name()] = names.get(env.fill_kernel_name(), 0) + 1
        _steps(rcw, env, orc, rng, 2, False)
        mask = np.array([1, 0, 1], dtype=np.uint8)
        rcw.reset_(env, mask=mask, seed=6); orc.reset(mask=mask, seed=6)
        _steps(rcw, env, orc, rng, 1, False)
        env.close()
    assert set(names) == {"rcw_fill_frame_kernel", "rcw_fill_flat_kernel", "rcw_fill_window_kernel", "rcw_fill256_kernel"}, names   # (256 rows x 3 agents: too small a batch for the one-launch step)


@pytest.mark.gpu
def test_turning_assignment_under_any_grid(rcw, oracle, monkeypatch):
    """rcw_top_store_flat_kernel's wavefront -> chunk assignment turns by 33 slots a group (docs/top_view.md); the walk through a group's
    slots wraps at the number of wavefronts, which follows the device's CU count: other grids (development build: RCW_TOP_STORE_GRID,
    1 .. 257 workgroups, i.e. turns of 33 mod 4 .. 33 mod 1028) and other turns (RCW_TOP_ROTATE 0 / 5) write the same pixels."""
    rng = np.random.default_rng(11)
    for grid in ("1", "3", "33", "257"):
        monkeypatch.setenv("RCW_TOP_STORE_GRID", grid)
        for rot in ("33", "5", "0"):
            monkeypatch.setenv("RCW_TOP_ROTATE", rot)
            for (H, W, pu), batch in (((8, 8, 16), 300), ((8, 8, 13), 40), ((5, 4, 20), 700)):
                env, orc = _make(rcw, oracle, batch, 17, auto_reset=True, out_of_bounds=1, render_top_view=True, pu_per_tu=pu,
                                 height_tile_map_tu=H, width_tile_map_tu=W, num_rays=40, library="dev")
                env.set_top_view_form("two-kernels")
                _steps(rcw, env, orc, rng, 2, True)
                mask = (rng.random(batch) < 0.5).astype(np.uint8)
                rcw.reset_(env, mask=mask, seed=7); orc.reset(mask=mask, seed=7)
                np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"grid {grid} turn {rot} {(H, W, pu)}")
                env.close()
