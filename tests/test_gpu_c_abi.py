"""The C ABI used from plain C (what a Julia `ccall` does), without Python in the call path:
tests/c_abi_harness.c is compiled with gcc against include/rcw.h, run on the GPU, and its frame
checksum is compared with the CPU oracle fed the same seed and action stream."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lcg_actions(steps, B, seed=99):
    out = np.zeros((steps, B), dtype=np.uint8)
    s = seed
    for t in range(steps):
        for a in range(B):
            s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            out[t, a] = 1 + (s >> 33) % 4
    return out


def test_plain_c_caller(oracle, tmp_path):
    exe = str(tmp_path / "harness")
    lib = os.path.join(ROOT, "raycastworlds.jl_amd", "lib")
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_harness.c"),
                    "-o", exe, "-L", lib, "-lrcw_hip", f"-Wl,-rpath,{lib}"], check=True)
    steps = 120
    res = subprocess.run([exe, str(steps)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    m = re.search(r"terminal_events=(\d+) checksum=([0-9a-f]{16}) pos0=([-\d.e+]+),([-\d.e+]+) dir0=(\d+)", res.stdout)
    assert m, res.stdout
    orc = oracle.OracleBatch(64, seed=2024, out_of_bounds=1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    acts = _lcg_actions(steps, 64)
    terminal = 0
    for t in range(steps):
        assert orc.step(acts[t]) == 0
        terminal += int(orc.done.sum())
    words = orc.camera_view.reshape(-1).astype(np.uint64)
    h = 1469598103934665603
    for wv in words.tolist():                      # FNV-1a over the words, as in the harness
        h = ((h ^ wv) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(m.group(1)) == terminal
    assert m.group(2) == f"{h:016x}"
    assert np.float32(m.group(3)) == orc.position[0, 0] and np.float32(m.group(4)) == orc.position[0, 1]
    assert int(m.group(5)) == orc.direction[0]
    # rcw_comm_init / rcw_gather_observations (both modes) with a world of one rank, RCCL called by the library
    assert "gather=ok" in res.stdout, res.stdout
    # the top view through the ABI alone: the two-kernel form (RCW_TOP_VIEW_TWO_KERNELS = 3), same pixels as the oracle's
    mt = re.search(r"top_view_form=(\d+) top_checksum=([0-9a-f]{16})", res.stdout)
    assert mt, res.stdout
    assert int(mt.group(1)) == 3
    assert "step_form=1" in res.stdout                                       # RCW_STEP_TWO_LAUNCHES by the rule at 64 agents; the harness steps a third of its rollout in the one-launch form
    assert "fill_kernel=rcw_fill256_draw_kernel" in res.stdout           # the handle renders the top view: camera fill + drawing in one launch
    assert "top_view_form_alone=3" in res.stdout                         # 256 x 256 px: the stand-alone call takes draw -> store too
    ort = oracle.OracleBatch(64, seed=2024, out_of_bounds=1, render_top_view=1, pu_per_tu=32,
                             height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    ort.set_state(orc.goal, orc.position, orc.direction)                     # (the image is a function of the state: rendered once, from the rollout's last)
    top = ort.top_view.reshape(-1).astype(np.uint64)
    weights = np.arange(top.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    with np.errstate(over="ignore"):
        ht_ = int((top * weights).sum(dtype=np.uint64))                      # position-weighted sum mod 2^64, as in the harness
    assert mt.group(2) == f"{ht_:016x}"
