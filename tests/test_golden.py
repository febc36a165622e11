"""Committed golden rollouts (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle still reproduces them (the fixtures pin the oracle against drift).
GPU: the HIP path reproduces them through the C ABI without the oracle in the loop."""
import glob
import os
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _cfg(z):
    return {str(k): int(v) for k, v in zip(z["config_keys"], z["config_vals"])}


def _crcs(frames):
    return np.array([zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in frames], dtype=np.uint32)


def _check(z, s, pos, d, reward, done, col_h, col_c, frames):
    np.testing.assert_array_equal(pos.view(np.uint32), z["pos"][s].view(np.uint32), err_msg=f"pos step {s}")
    np.testing.assert_array_equal(d, z["dir"][s])
    np.testing.assert_array_equal(reward, z["reward"][s])
    np.testing.assert_array_equal(np.asarray(done, dtype=np.uint8), z["done"][s])
    np.testing.assert_array_equal(col_h, z["col_h"][s], err_msg=f"height_line_pu step {s}")
    np.testing.assert_array_equal(col_c, z["col_c"][s], err_msg=f"colour step {s}")
    np.testing.assert_array_equal(_crcs(frames), z["crc"][s], err_msg=f"frame crc step {s}")


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    z = np.load(path)
    B = z["actions"].shape[1]
    orc = oracle.OracleBatch(B, seed=int(z["seed"]), out_of_bounds=1, **_cfg(z))
    np.testing.assert_array_equal(orc.goal, z["goal0"])
    np.testing.assert_array_equal(orc.position, z["pos0"])
    np.testing.assert_array_equal(orc.direction, z["dir0"])
    np.testing.assert_array_equal(orc.tile_map_chunks(), z["tile_map0"])
    np.testing.assert_array_equal(_crcs(orc.camera_view), z["crc0"])
    for s in range(z["actions"].shape[0]):
        assert orc.step(z["actions"][s]) == 0
        _check(z, s, orc.position, orc.direction, orc.reward, orc.done, orc.col_height, orc.col_colour,
               orc.camera_view)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_reproduces_golden(rcw, path):
    z = np.load(path)
    B = z["actions"].shape[1]
    env = rcw.SingleRoomModule.SingleRoom(batch=B, seed=int(z["seed"]), out_of_bounds=1, **_cfg(z))
    w = env.world
    np.testing.assert_array_equal(w.goal_position, z["goal0"])
    np.testing.assert_array_equal(w.player_position_wu, z["pos0"])
    np.testing.assert_array_equal(w.player_direction_au, z["dir0"])
    np.testing.assert_array_equal(w.tile_map_chunks, z["tile_map0"])
    np.testing.assert_array_equal(_crcs(env.camera_view_host()), z["crc0"])
    for s in range(z["actions"].shape[0]):
        rcw.act_(env, z["actions"][s])
        h, c = env.columns()
        _check(z, s, w.player_position_wu, w.player_direction_au, w.reward, w.done, h, c, env.camera_view_host())
    env.close()
