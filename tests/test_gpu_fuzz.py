"""Differential fuzzing in the suite: tools/fuzz_parity.py over 120 random configurations with a
fixed seed (map size, columns, headings, field of view, radius, step, camera/image heights, Float32
and Float64 world units, the unpinned switches, both BoundsError policies, auto-reset, top view)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_stay_in_parity():
    res = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "fuzz_parity.py"), "120", "2025"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "120 random configurations" in res.stdout and ", 0 mismatches" in res.stdout


def test_random_flat_kernel_geometries_stay_in_parity():
    """The same over the geometries only the flat kernels take: top views at 9..60 pixels a tile (two-kernel form asked
    for) and camera heights anywhere from 24 rows."""
    res = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "fuzz_parity.py"), "80", "77", "flat"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "80 random configurations" in res.stdout and ", 0 mismatches" in res.stdout


def test_random_sequences_of_api_calls_stay_in_parity():
    """tools/api_fuzz.py: 12 handles x 50 random calls — steps with host / device / scalar actions, masked and full resets,
    injected states, rejected actions, another stream, another output buffer, another top-view form, another form of the step (one launch / two), stand-alone
    re-renders, rays, descriptor expansion, profiling — every observable compared with the oracle after every call."""
    res = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "api_fuzz.py"), "12", "5", "50"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "12 runs x 50 calls, every observable equal" in res.stdout


def test_random_sequences_with_the_rccl_gather_stay_in_parity():
    """The same behind ShardedSingleRoom in a one-rank "nccl" group with the collective forced: the observation gather —
    torch.distributed and the library's own ncclAllGather, columns + expansion and frames — between the other calls
    (another stream, another output buffer, resets ...)."""
    res = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "api_fuzz.py"), "8", "9", "40", "sharded"],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "8 runs x 40 calls, every observable equal" in res.stdout and "gather_obs_abi" in res.stdout
