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
    stub, log, port = _build_stub(tmp_path), str(tmp_path / "stub.log"), _free_port()
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


def _build_stub(tmp_path):
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-fvisibility=hidden", "-o", stub, os.path.join(ROOT, "tests", "stub_rccl.c"),
                    "-ldl", "-lrt"], check=True)
    return stub


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def test_cfg4_as_stated_eight_ranks_with_the_gather(rcw, oracle, tmp_path):
    """BASELINE.json configs[3] in its own shape: 16x16, 256 columns, 65,536 agents over EIGHT ranks with the observation
    gather — on the one GPU a box has.  Eight ranks = four processes of two rank-threads (a box admits six processes on
    its card, this one included); tests/rccl_stub_cfg4_world8.py says what each rank checks.  Here: every process ends
    green, the stand-in's log shows ONE unique id and eight rcw_comm_init calls with it, ranks 0..7 of world 8."""
    torch = pytest.importorskip("torch")
    free, _ = torch.cuda.mem_get_info()
    if free < (60 << 30):
        pytest.skip(f"needs about 45 GiB of device memory, {free / 2**30:.0f} GiB are free")
    stub, log, port = _build_stub(tmp_path), str(tmp_path / "stub.log"), _free_port()
    procs = []
    for p in range(4):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(p), WORLD_SIZE="4", LOCAL_RANK="0",
                   RCW_RCCL_LIBRARY=stub, RCW_STUB_LOG=log, OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_stub_cfg4_world8.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, so[-2000:] + "\n" + se[-6000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("{")][-1]))
    assert sorted(r for o in outs for r in o["ranks"]) == list(range(8)) and all(o["parity"] == "ok" and o["world"] == 8 for o in outs)
    rank0 = [q for o in outs for q in o["per_rank"] if q["rank"] == 0][0]
    assert rank0["global_frames_bytes"] == 65536 * 256 * 256 * 4                      # 17 GB: the whole observation batch
    text = open(log).read()
    assert text.count("unique_id") == 1
    inits = [l for l in text.splitlines() if l.startswith("comm_init")]
    assert len(inits) == 8 and len({l.split("uid=")[1].split()[0] for l in inits}) == 1
    assert sorted(int(l.split("rank=")[1].split()[0]) for l in inits) == list(range(8)) and all("world=8" in l for l in inits)
    for r in range(8):
        assert text.count(f"all_gather rank={r} ") == 4, f"rank {r}: two gathers of two all-gathers each"
    dst = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "cfg4_world8.json"), "w") as f:
        json.dump(outs, f, indent=1)


def test_bench_rehearsal_with_four_ranks(rcw, tmp_path):
    """bench.py as the driver launches it for N > 1 — `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
    — rehearsed with every rank on the box's one GPU (--rehearse-on-one-gpu: gloo rendezvous, the gather block staged
    through the host).  N = 4: a box admits six processes on its card and this test process is one of them.  ONE JSON
    line on stdout, n_gpus 4, global batch 4 x 4096, a gather block without an error, exit code 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "2",
                          "--rehearse-on-one-gpu", "--cpu-baseline-seconds", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + "\n" + res.stderr[-6000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["global_batch"] == 4 * 4096 and out["steps"] == 5 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0               # rank 0's, at any N (round 6)
    assert out["roofline"]["traffic"] is None or "not re-measured" in out["roofline"]["traffic_source"]
    # every rank's own clock, reduced afterwards: the slowest rank is the job's time, the spread and every rank's fill-kernel
    # launch time are in the line
    assert out["ms_per_step"] == out["ms_per_step_max"] >= out["ms_per_step_min"] > 0 and "no collective inside the timed region" in out["timing"]
    per = out["roofline"]["per_rank"]
    assert [p["rank"] for p in per] == [0, 1, 2, 3] and all(p["launch_ms"] > 0 and p["ms_per_step"] > 0 for p in per)
    assert max(p["ms_per_step"] for p in per) == out["ms_per_step_max"] and min(p["ms_per_step"] for p in per) == out["ms_per_step_min"]
    assert out["roofline"]["launch_ms"] == max(p["launch_ms"] for p in per)
    g = out["gather"]
    assert "error" not in g and g["ranks"] == 4 and g["columns_us"] > 0 and g["frames_us"] > 0 and "rehearsal" in g


def test_a_hung_gather_looks_hung(rcw):
    """VERDICT round 3, weak #8, end to end: two ranks on the box's GPU (rehearsal mode), rank 1 never enters the gather's first
    collective, so rank 0 waits in it for ever — until the watchdog (here: 4 s).  The launcher must see a FAILURE (bench.py's
    ranks leave with exit code 3), rank 0's stdout still carries the ONE bench line, its gather reported as timed out with the
    collective it was stuck in, and both ranks name theirs on stderr."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--batch", "256", "--rehearse-on-one-gpu", "--watchdog-seconds", "4", "--hang-rank", "1"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode != 0, "a hung gather ended with exit code 0"
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0                                    # the headline is unaffected
    assert "did not finish within 4 s" in out["gather"]["error"] and "all_gather_into_tensor of height_line_pu" in out["gather"]["error"]
    assert "rank 0" in res.stderr and "rank 1" in res.stderr and "exiting with code 3" in res.stderr
    assert "told to hang" in res.stderr                                               # rank 1 says where it was, too


def test_communicator_argument_checks_and_a_library_that_cannot_be_loaded(rcw):
    """The refusals around the communicator that no other test provoked (a census of the error returns taken, round 4), in child
    processes because RCCL is loaded once per process: RCW_RCCL_LIBRARY naming a file that does not exist, or a library without
    the RCCL symbols, is an error (never a silent fall back to another copy); with the real library and a world of one: a second
    rcw_comm_init, NULL / misaligned gather buffers, an unknown gather mode."""

    prog = r"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import raycastworlds_jl_amd as RCW
from raycastworlds_jl_amd import _capi
mode = sys.argv[1]
env = RCW.SingleRoomModule.SingleRoom(batch=8, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
lib, h = env._lib, env._h
uid = (C.c_uint8 * _capi.RCW_UNIQUE_ID_BYTES)()
if mode in ("missing", "not-rccl"):
    rc = lib.rcw_comm_unique_id(uid)
    print("RC", rc, _capi.last_error(lib))
    rc = lib.rcw_comm_init(h, uid, 0, 1)
    print("RC", rc, _capi.last_error(lib))
else:
    _capi.preload_rccl()
    assert lib.rcw_comm_unique_id(uid) == 0
    assert lib.rcw_comm_init(h, uid, 0, 1) == 0
    print("RC", lib.rcw_comm_init(h, uid, 0, 1), _capi.last_error(lib))
    buf = C.c_void_p()
    assert lib.rcw_device_malloc(h, 8 * 64 * 256 * 4 + 64, C.byref(buf)) == 0
    print("RC", lib.rcw_gather_columns(h, None, buf), _capi.last_error(lib))
    print("RC", lib.rcw_gather_observations(h, 0, None), _capi.last_error(lib))
    print("RC", lib.rcw_gather_observations(h, 0, C.c_void_p(buf.value + 4)), _capi.last_error(lib))
    print("RC", lib.rcw_gather_observations(h, 7, buf), _capi.last_error(lib))
    assert lib.rcw_gather_observations(h, 0, buf) == 0 and lib.rcw_sync(h) == 0
    assert lib.rcw_comm_destroy(h) == 0 and lib.rcw_device_free(h, buf) == 0
env.close()
print("DONE")
"""

    def child(mode, **extra_env):
        env = {k: v for k, v in os.environ.items() if k != "RCW_RCCL_LIBRARY"}
        env.update(extra_env)
        res = subprocess.run([sys.executable, "-c", prog, mode], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and "DONE" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
        return [l for l in res.stdout.splitlines() if l.startswith("RC ")]

    out = child("missing", RCW_RCCL_LIBRARY="/nonexistent/librccl.so.1")
    assert len(out) == 2 and all(l.startswith("RC -7 RCW_RCCL_LIBRARY=/nonexistent/librccl.so.1 could not be loaded") for l in out), out
    import ctypes.util

    libm = ctypes.util.find_library("m")
    out = child("not-rccl", RCW_RCCL_LIBRARY=libm)
    assert len(out) == 2 and all(l.startswith("RC -7 librccl lacks ncclGetUniqueId") for l in out), out
    out = child("real")
    assert len(out) == 5, out
    assert out[0].startswith("RC -1 the handle already has a communicator")
    assert out[1].startswith("RC -1 NULL argument") and out[2].startswith("RC -1 NULL argument")
    assert out[3].startswith("RC -1 frames must be 16-byte aligned") and out[4].startswith("RC -1 unknown gather mode 7")
