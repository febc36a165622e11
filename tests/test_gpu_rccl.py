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


def test_library_transport_with_two_ranks(rcw, oracle, tmp_path):
    """rcw_comm_init(rank = 1, world = 2), the uid hand-over of sharded.py and the library's gather with TWO ranks: both on
    the box's one GPU, the real engine, tests/stub_rccl.c in place of librccl (RCW_RCCL_LIBRARY).  Each rank compares the
    gathered global batch with the unsharded CPU oracle; the stub's log shows who initialised what."""
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-fvisibility=hidden", "-o", stub, os.path.join(ROOT, "tests", "stub_rccl.c"),
                    "-ldl", "-lrt"], check=True)
    log = str(tmp_path / "stub.log")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                   RCW_RCCL_LIBRARY=stub, RCW_STUB_LOG=log)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_stub_world2.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, so[-2000:] + "\n" + se[-4000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("{")][-1]))
    assert sorted(o["rank"] for o in outs) == [0, 1] and all(o["parity"] == "ok" for o in outs)
    text = open(log).read()
    assert text.count("unique_id") == 1                                       # made once, on rank 0 ...
    uid = [l.split("uid=")[1].split()[0] for l in text.splitlines() if l.startswith("comm_init")]
    assert len(uid) == 2 and uid[0] == uid[1]                                 # ... and the same 128 bytes reached both ranks
    assert "comm_init rank=1 world=2" in text and "comm_init rank=0 world=2" in text
    assert text.count("all_gather rank=1") >= 4
