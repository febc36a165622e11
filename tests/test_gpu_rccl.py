"""The RCCL path on hardware: a one-rank "nccl" process group on the box's GPU, the real engine, the observation
gather forced through the collective on both transports (torch.distributed and the library's rcw_gather_*).
Runs tests/rccl_world1.py in a child process because MASTER_ADDR / RANK / WORLD_SIZE must be set before the
process first touches the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_gather_world_of_one(rcw, tmp_path):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1.py"), "--batch", "256",
                          "--time-batch", "8192", "--reps", "10"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + "\n" + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["backend"] == "nccl" and out["world"] == 1 and out["parity"] == "ok"
    t = out["timing"]
    assert t["agents"] == 8192 and t["abi_columns_gather_only_us"] > 0 and t["abi_frames_us"] > 0
    dst = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "rccl_world1.json"), "w") as f:
        json.dump(out, f, indent=1)
