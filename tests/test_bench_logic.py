"""bench.py's host logic that needs no GPU."""
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_watchdog_makes_a_hang_look_like_one():
    """a collective that hangs inside the OPTIONAL gather must not end the run with exit code 0.
    Every rank names the collective it was stuck in on stderr and leaves with code 3; rank 0 first writes the headline it
    had ready, with the gather reported as timed out."""
    bench = _bench()
    for rank in (0, 1):
        emitted, codes, err = [], [], io.StringIO()
        ready = {"line": {"metric": "m", "value": 1.0}} if rank == 0 else {"line": None}
        pending = {"what": "all_gather_into_tensor of frames (uint32 as int32): 1073741824 B per rank -> 8589934592 B, world 8"}
        wd = bench.make_watchdog(rank, emitted.append, ready, pending, 120.0, exit_fn=codes.append, err=err)
        wd()
        assert codes == [3]
        assert f"rank {rank}" in err.getvalue() and "all_gather_into_tensor of frames" in err.getvalue() and "world 8" in err.getvalue()
        if rank == 0:
            assert len(emitted) == 1 and emitted[0]["value"] == 1.0
            assert "did not finish" in emitted[0]["gather"]["error"] and "all_gather_into_tensor of frames" in emitted[0]["gather"]["error"]
            json.dumps(emitted[0])
        else:
            assert emitted == []


def test_bench_refuses_to_run_without_a_gpu_or_with_a_wrong_world():
    """No CPU fallback in the measured path: without a GPU bench.py exits 3 with a message, and a --gpus that does not
    match WORLD_SIZE exits 2 before anything is initialised."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 2 and "WORLD_SIZE" in res.stderr and res.stdout.strip() == ""
    import torch

    if not torch.cuda.is_available():
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 3 and "no GPU" in res.stderr and res.stdout.strip() == ""
