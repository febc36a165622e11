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


def test_pmc_parser_sums_xcds_and_averages_dispatches(tmp_path):
    """roofline.traffic is measured live (two rocprofv3 --pmc passes of bench.py as child processes): the parser of their
    *counter_collection.csv — a row per XCD and dispatch — sums a dispatch's rows and averages over the kernel's dispatches."""
    bench = _bench()
    d = tmp_path / "run" / "box"
    d.mkdir(parents=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    fill = '"void (anonymous namespace)::rcw_fill256_kernel<false>(RcwDev, int const*, unsigned char const*, unsigned int __vector(4)*, long long, unsigned char const*)"'
    cast = '"void (anonymous namespace)::rcw_cast_kernel<float, false, false>(RcwDev, unsigned char const*, unsigned char const*, int)"'
    for disp, per_xcd in ((1, 131072.0), (2, 131074.0)):
        for xcd in range(8):
            rows.append(f"{disp},{disp},4,1,100,100,65536,7,{fill},256,0,0,24,0,48,WRITE_SIZE,{per_xcd},1,2")
        rows.append(f"{disp + 10},{disp + 10},4,1,100,100,262144,8,{cast},64,112,0,44,0,84,WRITE_SIZE,5.0,1,2")
    (d / "123_counter_collection.csv").write_text("\n".join(rows) + "\n")
    assert bench.parse_pmc(str(tmp_path), "WRITE_SIZE", "rcw_fill256_kernel") == (8 * 131072.0 + 8 * 131074.0) / 2
    assert bench.parse_pmc(str(tmp_path), "WRITE_SIZE", "rcw_cast") == 5.0
    assert bench.parse_pmc(str(tmp_path), "FETCH_SIZE", "rcw_fill256_kernel") is None
    assert bench.parse_pmc(str(tmp_path / "nowhere"), "WRITE_SIZE", "rcw_") is None


def test_rank_times_are_reduced_after_the_clock_and_keep_the_spread():
    """with N > 1 every rank stops its own clock behind its own synchronisation; the job's time is the
    slowest rank's (MAX) and the line keeps the spread and every rank's fill-kernel launch time, so a straggling GPU shows."""
    bench = _bench()
    rows = [(3.4e-3, 3.30, 0.011, 0.156, 0.0), (3.9e-3, 3.80, 0.012, 0.181, 0.0), (3.5e-3, 3.35, 0.011, 0.157, 0.0)]
    r = bench.reduce_rank_times(rows, steps=20)
    assert r["dt"] == 3.9e-3 and r["kernel_ms"] == 3.80 and r["fill_ms"] == 0.181 and r["cast_ms"] == 0.012
    assert abs(r["ms_per_step_min"] - 0.17) < 1e-12 and abs(r["ms_per_step_max"] - 0.195) < 1e-12
    assert [p["rank"] for p in r["per_rank"]] == [0, 1, 2]
    assert [p["launch_ms"] for p in r["per_rank"]] == [0.156, 0.181, 0.157]            # rank 1 is the slow GPU, and it shows
    json.dumps(r)
    one = bench.reduce_rank_times([(1.0, 2.0, 3.0, 4.0, 5.0)], steps=10)
    assert one["ms_per_step_min"] == one["ms_per_step_max"] == 100.0 and len(one["per_rank"]) == 1


def test_no_collective_inside_the_timed_regions():
    """The source of bench.py between a clock's start and its stop holds no barrier, all-reduce or all-gather (the closing
    barrier comes AFTER the clock has stopped; one warm-up collective comes before it starts)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for start, stop in (("t0 = time.perf_counter()", "dt = time.perf_counter() - t0  # this rank's own clock"), ("t0a = time.perf_counter()", "dta = time.perf_counter() - t0a")):
        b = src.index(stop)
        a = src.rindex(start, 0, b)
        region = src[a:b]
        for word in ("barrier(", "all_reduce", "all_gather", "gather_rows", "dist."):
            assert word not in region, (word, start)
        assert "rt.synchronize()" in region                          # the rank's own synchronisation is inside (Runtime.synchronize = torch.cuda.synchronize)
        assert "barrier()" in src[b:b + 400]                          # ... and the barrier right behind the stop
    warm = src.index("gather_rows([0.0])")
    assert warm < src.index("t0 = time.perf_counter()", warm) < src.index("dt = time.perf_counter() - t0  # this rank's own clock")


def test_top_view_scratch_bytes_of_the_fused_launch():
    """bench.py --top-view: the fused launch writes the frames AND the drawing's scratch; 8x8 tiles of 32 px, 4096 agents: 8 KiB of
    plane + 8 B + 8 tile columns x 8 B an agent."""
    bench = _bench()
    assert bench.top_view_scratch_bytes(8, 8, 32, 4096) == 4096 * (8192 + 8 + 64)
    assert bench.top_view_scratch_bytes(8, 16, 32, 1) == 256 * 512 // 8 + 8 + 8 * 16


def test_the_multi_rank_control_flow_runs_on_doubles(tmp_path):
    """bench.py's N > 1 path has only ever run over gloo and a stand-in; its first run over RCCL is the
    driver's.  tests/bench_double.py runs main() itself — rank 0 of a world of 2 — with doubles for torch.distributed and the engine:
    the `dist` refuses any all_gather_into_tensor whose output is not the concatenation of `world` inputs along dim 0 with the same
    dtype, device and contiguity (the rule RCCL and gloo share; round 5's first gather_rows broke it and only a GPU rehearsal caught
    it), counts barriers, and the engine records its calls.  The line must come out whole: n_gpus 2, the slowest rank's clock, both
    ranks in per_rank, the gather block without an error, and — new in round 6 — rank 0's cpu_baseline at N > 1 and a traffic figure
    that is either null or says it was not re-measured at this N."""
    env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_RANK",)}
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", BENCH_DOUBLE_LOG=str(tmp_path / "calls.json"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_double.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                          "--batch", "8", "--cpu-baseline-seconds", "0.3", "--api", "rlbase"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert abs(d["value"] - 2 * 8 * 6 / (d["ms_per_step"] * 6 / 1e3)) < 1e-6 * d["value"]
    assert d["ms_per_step"] == d["ms_per_step_max"] >= d["ms_per_step_min"] > 0
    assert [r["rank"] for r in d["roofline"]["per_rank"]] == [0, 1]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "caches" in d["cpu_baseline"]["sample"]
    assert d["roofline"]["traffic"] is None or "not re-measured" in d["roofline"]["traffic_source"]
    assert "error" not in d["gather"], d["gather"]
    assert d["gather"]["ranks"] == 2 and d["api_loop"]["host_syncs_per_step"] == 0
    calls = json.load(open(tmp_path / "calls.json"))
    assert calls["all_gather"] >= 4 and calls["barrier"] >= 4 and calls["destroyed"] == 1
    assert calls["steps"] == 2 + 6 + 6 + 2 + 6                       # warm-up + timed + per-kernel events + the API loop's two
    assert calls["collectives_inside_timed_regions"] == 0
