#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched SingleRoom step/render path on MI355X.

A "step" is one pass of the hot path (dynamics -> cast -> project -> frame fill,
RCW.act!(env, a) SR:333-340 minus the top view) over one batch of agents.  Workload at
every N: BASELINE.json configs[1] per GPU (SingleRoom 8x8, 256 view columns, 4096 agents,
H_cam = 256), agents sharded by rank with no data-path collective (weak scaling).
State, actions and observations are resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (HBM write
roofline of the step kernel, HIP-event timed on the kernel's own stream; `traffic` = HBM bytes
per launch from the PMC counters, measured by the run itself at N = 1: two short
`rocprofv3 --pmc` passes of this script as child processes, after the timed region) and
`cpu_baseline` (the CPU oracle — a C restatement of the reference, kind "port" — timed on
this box's host cores on a bounded sample of the same workload).  A default run takes 20-40 s:
a few seconds for the GPU part, ~6-12 s each for the two counter passes (bounded at 60 s each; a pass
that fails or runs out of time says so on stderr and the committed figure is used, labelled), ~15 s for
the CPU baseline.

N > 1: every rank stops its own clock after its own synchronisation — no collective inside the timed
region —; the per-rank times are gathered afterwards, `value` uses the slowest rank (MAX), and the line
carries `ms_per_step_min` / `ms_per_step_max` over ranks and `roofline.per_rank` (every rank's fill-kernel
launch time), so that a straggling GPU shows.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

WORKLOADS = {
    # name: (kwargs, agents per GPU)
    "cfg2": (dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256), 4096),
    "cfg3": (dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=512), 16384),
    "cfg4": (dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=256), 8192),
    "cfg5": (dict(height_tile_map_tu=32, width_tile_map_tu=32, num_rays=1024), 8192),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants a 16-CPU share per GPU)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _time_oracle(O, kw, agents, threads, target_seconds):
    """env-steps/s of the CPU oracle with `threads` OpenMP threads on `agents` agents, for about target_seconds."""
    O.set_num_threads(threads)
    orc = O.OracleBatch(agents, seed=0, auto_reset=1, out_of_bounds=0, **kw)
    rng = np.random.default_rng(0)
    acts = rng.integers(1, 5, (64, agents)).astype(np.uint8)
    for s in range(8):                       # warm up threads and pages
        orc.step(acts[s])
    t0 = time.perf_counter()
    for s in range(16):
        orc.step(acts[s])
    per_step = (time.perf_counter() - t0) / 16
    steps = int(max(16, min(200000, target_seconds / max(per_step, 1e-7))))
    t0 = time.perf_counter()
    for s in range(steps):
        orc.step(acts[s & 63])
    dt = time.perf_counter() - t0
    orc.close()
    return agents * steps / dt, steps, dt


def cpu_baseline(kw, batch_hint, target_seconds=10.0):
    """Time the CPU oracle on a bounded sample of the same workload: (i) ONE thread — what the single-threaded
    reference's own loop corresponds to — and (ii) OpenMP over agents on all usable host cores."""
    from oracle import oracle as O

    threads = usable_cores()
    frame_kib = 4 * 256 * kw["num_rays"] // 1024
    one, steps1, dt1 = _time_oracle(O, kw, min(batch_hint, 32), 1, 0.6 * target_seconds)
    agents = min(batch_hint, 32 * threads)
    allc, steps, dt = _time_oracle(O, kw, agents, threads, target_seconds)
    return {
        "value": allc,
        "unit": "env-steps/s",
        "cores": threads,
        "kind": "port",
        "single_thread": one,
        "all_cores": allc,
        "sample": f"all cores: {agents} agents x {steps} steps ({dt:.1f} s, OpenMP over agents, {threads} threads); "
                  f"single thread: {min(batch_hint, 32)} agents x {steps1} steps ({dt1:.1f} s); same workload, C restatement "
                  f"of the reference camera path incl. the frame fill; the sample's frames ({agents * frame_kib // 1024} MiB, {frame_kib} KiB an agent) "
                  f"stay in the host's caches where the GPU's batch does not fit any: a baseline that flatters the CPU",
    }


def parse_pmc(directory, counter, kernel_substring):
    """Mean of `counter` per dispatch of the kernels whose name contains `kernel_substring`, from the *counter_collection.csv of a
    rocprofv3 --pmc run (rows are per XCD: summed per dispatch first).  None when nothing matches."""
    import csv
    import glob
    from collections import defaultdict

    per = defaultdict(float)
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter and kernel_substring in r.get("Kernel_Name", ""):
                    per[(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    return sum(per.values()) / len(per) if per else None


def being_profiled() -> bool:
    """This process already runs under a profiler (tools/kstats.sh, tools/gpu_round.sh wrap bench.py in rocprofv3): no nested one."""
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def reduce_rank_times(rows, steps):
    """rows[r] = (dt seconds, kernel_ms, cast_ms, fill_ms, top_ms) as rank r measured them with its OWN clock (stopped after
    its own synchronisation).  The job's figures are the slowest rank's (MAX, what whole-job throughput is bound by); the
    spread and every rank's fill-kernel launch time go into the line so that one slow GPU is visible."""
    rows = [[float(v) for v in r] for r in rows]
    mx = [max(r[k] for r in rows) for k in range(5)]
    return {
        "dt": mx[0], "kernel_ms": mx[1], "cast_ms": mx[2], "fill_ms": mx[3], "top_ms": mx[4],
        "ms_per_step_min": min(r[0] for r in rows) * 1e3 / steps,
        "ms_per_step_max": mx[0] * 1e3 / steps,
        "per_rank": [{"rank": i, "ms_per_step": r[0] * 1e3 / steps, "launch_ms": r[3], "cast_ms": r[2],
                      "step_kernels_ms": r[1] / steps} for i, r in enumerate(rows)],
    }


def gather_rank_rows(dist, world, values, device="cpu"):
    """Every rank's row of figures on every rank, [world][len(values)]: ONE all-gather (flat output, the form gloo and RCCL both
    take), always OUTSIDE a timed region.  `dist` None (a single rank without a process group): the one row."""
    if dist is None:
        return [[float(v) for v in values]]
    import torch

    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    allt = torch.empty(world * len(values), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(allt, t)
    return allt.view(world, len(values)).cpu().tolist()


def top_view_scratch_bytes(H, W, pu, B):
    """What the top view's drawing leaves in HBM for the store kernel, per launch: the ray-line bit plane (one bit a
    pixel), the player's pixel (8 B an agent) and the 2-bit tile codes (8 B per tile column and 256-row run)."""
    Ht, Wt = H * pu, W * pu
    return B * (Ht * Wt // 8 + 8 + 8 * W * ((Ht + 255) // 256))


def live_traffic(workload, batch, kernel, seconds=60.0, extra_args=()):
    """HBM bytes per launch of the dominant kernel from the PMC counters, measured NOW: this script again, a few steps, under
    rocprofv3 — WRITE_SIZE and FETCH_SIZE in separate passes (they do not fit one), `--pmc` with `--kernel-trace` only, as the
    guide's HBM section prescribes; KiB units; FETCH_SIZE doubled (gfx950 reports half of wide streaming reads: an upper bound
    for this kernel's narrow descriptor gathers).  Child processes, outside every timed region; None if the profiler is not
    there or a pass fails — the caller then falls back to the committed figure and says so."""
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        print("bench.py: roofline.traffic: no rocprofv3 on PATH; the committed figure (if any) is used", file=sys.stderr)
        return None
    if being_profiled():
        return None                                      # (a profiler's own run of this script: it measures what it came for)
    out = {}
    for counter in ("WRITE_SIZE", "FETCH_SIZE"):
        d = None
        try:
            d = tempfile.mkdtemp(prefix="rcw_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--traffic", "off",
                   "--workload", workload, "--batch", str(batch)] + list(extra_args)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
            env["TMPDIR"] = "/tmp"
            # its own session: on a timeout the WHOLE group goes (rocprofv3 runs the program as its child; killing the
            # profiler alone would leave that child on the GPU)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=seconds)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, 9)
                except OSError:
                    pass
                proc.wait()
                rc = -9
            v = parse_pmc(d, counter, kernel) if rc == 0 else None
            if v is None:
                why = (f"ran out of its {seconds:.0f} s" if rc == -9 else f"ended with exit code {rc}" if rc != 0
                       else f"recorded no dispatch of {kernel}")
                print(f"bench.py: roofline.traffic: the live rocprofv3 --pmc {counter} pass {why}; "
                      f"the committed figure (if any) is used and labelled in traffic_source", file=sys.stderr)
        except Exception as e:   # noqa: BLE001 — a reported extra: never costs the bench line
            print(f"bench.py: roofline.traffic: the live rocprofv3 --pmc {counter} pass failed ({type(e).__name__}: {e}); "
                  f"the committed figure (if any) is used and labelled in traffic_source", file=sys.stderr)
            v = None
        finally:
            if d:
                shutil.rmtree(d, ignore_errors=True)
        if v is None:
            return None
        out[counter] = v * 1024.0                        # KiB -> bytes
    return {"write_bytes": out["WRITE_SIZE"], "fetch_bytes_raw": out["FETCH_SIZE"], "traffic": out["WRITE_SIZE"] + 2.0 * out["FETCH_SIZE"]}


def make_watchdog(rank, emit, headline_ready, pending, seconds, exit_fn=os._exit, err=None):
    """What runs when the optional gather has not finished `seconds` after it began (a rank that died or hangs inside a
    collective leaves the others waiting for ever).  The bench line must still go out — rank 0 writes the headline it
    has ready, with the gather reported as timed out — but a hang must LOOK like one to whoever launched the run: every
    rank says on stderr which collective it was in (and the sizes), and leaves with exit code 3, not 0."""
    def watchdog():
        stream = err if err is not None else sys.stderr
        print(f"bench.py: rank {rank}: the optional observation gather did not finish within {seconds:.0f} s; "
              f"stuck in: {pending.get('what', 'unknown')}; exiting with code 3", file=stream, flush=True)
        if rank == 0 and headline_ready.get("line") is not None:
            out_t = dict(headline_ready["line"])
            out_t["gather"] = {"error": f"the optional observation gather did not finish within {seconds:.0f} s "
                                        f"(rank 0 was in: {pending.get('what', 'unknown')}); headline unaffected; exit code 3"}
            emit(out_t)
        exit_fn(3)
    return watchdog


class Runtime:
    """What main() needs from torch, torch.distributed and the engine, in one place — so that tests/test_bench_logic.py can run main()'s
    whole control flow (warm-up collective, timed region, gather of the ranks' clocks, the gather block, the line) on a CPU with doubles
    for the three: a `dist` whose all_gather_into_tensor enforces the flat-shape rule RCCL and gloo share, a recording engine."""
    device = "cuda"

    def __init__(self):
        import torch

        self.torch = torch
        self.dist = None

    def gpu_available(self):
        return self.torch.cuda.is_available()

    def set_device(self, local_rank):
        self.torch.cuda.set_device(local_rank)

    def init_dist(self, backend, local_rank):
        import torch.distributed as dist

        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=self.torch.device("cuda", local_rank))
        self.dist = dist

    def engine(self):
        import raycastworlds_jl_amd as RCW

        return RCW

    def ensure_built(self, rank):
        from raycastworlds_jl_amd import _capi

        if not os.path.exists(_capi.LIB_PATH):       # a checkout without the (git-ignored) build artefact
            if rank == 0:
                from raycastworlds_jl_amd import build as _build

                _build.build()
            if self.dist is not None:
                self.dist.barrier()

    def make_actions(self, total, B, rank):
        gen = self.torch.Generator(device="cuda")
        gen.manual_seed(1234 + rank)
        return self.torch.randint(1, 5, (total, B), dtype=self.torch.uint8, device="cuda", generator=gen)

    def share_stream(self, env):
        # engine and torch share ONE stream (the deployment shape: policy kernels and env kernels in order on
        # a single queue, no cross-stream waits)
        stream = self.torch.cuda.Stream()
        env.set_stream(stream.cuda_stream)
        self.torch.cuda.set_stream(stream)

    def synchronize(self):
        self.torch.cuda.synchronize()


def main(argv=None, rt=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="agents per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0, help="about how long the all-cores leg of the CPU baseline runs")
    ap.add_argument("--traffic", default="live", choices=["live", "committed", "off"],
                    help="roofline.traffic (HBM bytes per launch of the dominant kernel, PMC counters): live = measured by this run (rank 0 "
                         "at N = 1: two short rocprofv3 --pmc passes of this script as child processes, after the timed region; falls back to "
                         "the committed figure if the profiler is not available), committed = profiles/pmc_traffic.json, off = null")
    ap.add_argument("--no-auto-reset", action="store_true")
    ap.add_argument("--step-form", default="auto", choices=["auto", "one-launch", "two-launches"],
                    help="rcw_set_step_form: auto = the library's rule (one launch a step wherever the geometry allows: every BASELINE "
                         "configuration), two-launches = the cast kernel followed by the fill kernel (rounds 1-5), for the comparison")
    ap.add_argument("--top-view", action="store_true",
                    help="also render the reference's top view every step (update_top_view! SR:446-483, opt-in in the "
                         "engine) and report that kernel's own roofline block; NOT the headline workload")
    ap.add_argument("--gather", action="store_true",
                    help="also time the optional observation gather (RCCL all-gather of descriptors / frames) after the "
                         "timed region; on by default when --gpus > 1, with one GPU it runs a one-rank process group")
    ap.add_argument("--api", default="raw", choices=["raw", "rlbase"],
                    help="raw: time RCW.act!(env, actions) alone (the headline).  rlbase: ALSO time the loop of the reference's own "
                         "test (test/runtests.jl:26-33: RLBase.state -> env(action) -> RLBase.reward -> RLBase.is_terminated every step, "
                         "with a device-resident consumer of reward / done) and report it as `api_loop` beside the headline")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: all ranks share GPU 0 and rendezvous over gloo (checks the N>1 code path on a 1-GPU box; "
                         "the gather block runs too, its collectives staged through host memory because gloo does not all-gather "
                         "device tensors, and its frame gather on the first 256 agents of every rank)")
    ap.add_argument("--watchdog-seconds", type=float, default=120.0,
                    help="how long the optional gather may take before every rank gives up with exit code 3")
    ap.add_argument("--hang-rank", type=int, default=-1,
                    help="test only: this rank never enters the gather's first collective (tests/test_gpu_rccl.py drives the watchdog with it)")
    args = ap.parse_args(argv)

    # Rank 0 must print ONE JSON line on stdout and nothing else.  Libraries do not know that (RCCL prints a version
    # banner on stdout when a communicator is created): from here on file descriptor 1 is stderr, and the JSON line is
    # written to the saved, real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rt = rt or Runtime()
    torch = rt.torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not rt.gpu_available():
        print("bench.py: no GPU visible; the product path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if args.rehearse_on_one_gpu:
        local_rank = 0
    rt.set_device(local_rank)
    if world > 1 or args.gather:
        if world == 1:                               # --gather on one GPU: a process group of one rank
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        rt.init_dist("gloo" if args.rehearse_on_one_gpu else "nccl", local_rank)
    dist = rt.dist
    dev = rt.device

    RCW = rt.engine()
    rt.ensure_built(rank)

    kw, per_gpu = WORKLOADS[args.workload]
    B = args.batch or per_gpu
    N, Hc = kw["num_rays"], 256
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=0, device=local_rank, auto_reset=not args.no_auto_reset,
                                          agent_id_offset=rank * B, render_top_view=args.top_view, **kw)
    if args.step_form != "auto":
        env.set_step_form(args.step_form)
    step_form = env.step_form()
    # U{1..4} actions per agent per step (test/runtests.jl:28), pre-generated on the device
    total = args.warmup + args.steps
    actions = rt.make_actions(total, B, rank)
    rt.synchronize()
    rt.share_stream(env)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync_counting_bounds_errors():
        # The reference raises BoundsError when a forward move lands exactly on x = H-1 or
        # y = W-1 (include/rcw.h, RCW_OOB_ERROR).  The engine reports it and leaves that agent
        # untouched for that action; the batch goes on.  Count the agents it happened to.
        try:
            env.sync()
            return 0
        except IndexError:
            n = int((env.world.status != 0).sum())
            env.clear_error()
            return n

    cpu_coll = args.rehearse_on_one_gpu              # gloo moves host tensors only

    def gather_rows(values):
        return gather_rank_rows(dist, world, values, "cpu" if cpu_coll else dev)

    for s in range(args.warmup):
        RCW.act_(env, actions[s])
    sync_counting_bounds_errors()
    rt.synchronize()
    gather_rows([0.0])                               # one warm-up collective: the communicator is up before the clock starts
    rt.synchronize()
    barrier()
    t0 = time.perf_counter()
    env.timer_start()
    for s in range(args.warmup, total):
        RCW.act_(env, actions[s])
    kernel_ms = env.timer_stop()   # HIP events on the stream the kernels run on
    bounds_errors = sync_counting_bounds_errors()
    rt.synchronize()
    dt = time.perf_counter() - t0  # this rank's own clock, stopped behind its own synchronisation: NO collective inside dt
    barrier()
    # Outside the timed region: the same steps again with HIP events around each kernel, to
    # read the dominant (fill) kernel's own launch duration for the roofline block.
    nprof = min(args.steps, 200)
    env.profile(True)
    for s in range(args.warmup, args.warmup + nprof):
        RCW.act_(env, actions[s])
    cast_ms, top_ms, fill_ms, nrec = env.profile_read()
    env.profile(False)
    sync_counting_bounds_errors()
    # the ranks' own figures, gathered; the job's are the slowest rank's (MAX), the spread goes into the line
    ranks = reduce_rank_times(gather_rows([dt, kernel_ms, cast_ms, fill_ms, top_ms]), args.steps)
    dt, kernel_ms, cast_ms, fill_ms, top_ms = (ranks[k] for k in ("dt", "kernel_ms", "cast_ms", "fill_ms", "top_ms"))
    # Outside the timed region too (--api rlbase): the same steps through the verbs the reference's test drives
    # (test/runtests.jl:26-33), with a GPU-resident consumer: state(env) -> env(action) -> reward(env) -> is_terminated(env)
    # every step.  reward / is_terminated return the engine's own device arrays (refreshed in stream order), so the loop
    # issues no host synchronisation; `host_syncs_per_step` counts what the host layer issued.
    api_loop = None
    if args.api == "rlbase":
        RLBase = RCW.RLBase
        rl = RCW.RLBaseEnv(env)
        returns = torch.zeros(B, dtype=torch.float32, device=dev)
        episodes = torch.zeros(B, dtype=torch.int32, device=dev)
        def api_step(s):
            state = RLBase.state(rl)                                  # runtests.jl:27 (aliased, device-resident)
            rl(actions[s])                                            # runtests.jl:29
            r = RLBase.reward(rl).torch(sync=False)                   # runtests.jl:30
            d = RLBase.is_terminated(rl).torch(sync=False)            # runtests.jl:31
            returns.add_(r)                                           # the consumer: on the stream the engine runs on
            episodes.add_(d)
            return state

        for s in range(args.warmup):                                  # (the whole loop body: torch loads its kernels on first use)
            api_step(s)
        sync_counting_bounds_errors()
        rt.synchronize()
        barrier()
        returns.zero_(); episodes.zero_()
        syncs0 = env.host_syncs
        t0a = time.perf_counter()
        for s in range(args.warmup, total):
            state = api_step(s)
        syncs = env.host_syncs - syncs0                               # (before the closing synchronisation below)
        bounds_api = sync_counting_bounds_errors()
        rt.synchronize()
        dta = time.perf_counter() - t0a                               # (this rank's own clock, as above)
        barrier()
        dta = max(r[0] for r in gather_rows([dta]))
        api_loop = {"loop": "RLBase.state -> env(action) -> RLBase.reward -> RLBase.is_terminated per step (test/runtests.jl:26-33), "
                            "reward / done consumed on the device (returns += reward; episodes += done)",
                    "value": world * B * args.steps / dta, "unit": "env-steps/s", "ms_per_step": dta * 1e3 / args.steps,
                    "fraction_of_raw_act": (world * B * args.steps / dta) / (world * B * args.steps / dt),
                    "host_syncs_per_step": syncs / args.steps, "state_shape": list(state.shape),
                    "goals_reached": int(episodes.sum().item()), "agents_that_hit_reference_BoundsError": bounds_api}
    # Outside the timed region too: the OPTIONAL observation gather north_star names (stepping itself has no
    # collective).  RCCL all-gather over the process group, issued on the stream the engine runs on: the compact
    # descriptors (5 B per column) with the pixel expansion on the receiver, and the frames themselves.
    fill_kernel = env.fill_kernel_name()
    top_pu, top_form = env.cfg.pu_per_tu, env.top_view_form()

    def build_line(gather):
        frame_bytes = 4 * Hc * N                          # bytes per env-step
        bytes_per_launch = frame_bytes * B                # one fill launch = B agents
        launch_note = None
        if fill_kernel == "rcw_fill256_draw_kernel":
            # --top-view, fused launch: the kernel between the two events is the camera fill AND the top view's drawing; what it
            # writes is the frames plus the drawing's scratch for the store kernel
            bytes_per_launch += top_view_scratch_bytes(kw["height_tile_map_tu"], kw["width_tile_map_tu"], top_pu, B)
            launch_note = ("one launch = the camera fill (workgroups 0..255) + the top view's drawing (one workgroup per agent): bytes_per_launch = "
                           "frames + bit planes + player pixels + tile codes; launch_ms covers both")
        elif args.top_view and top_form == "two-kernels":
            launch_note = "the top view's draw kernel runs on a side stream beside this launch and inside its launch_ms"
        fill_s = fill_ms / 1e3                            # dominant kernel, HIP events around it
        achieved = bytes_per_launch / fill_s / 1e9
        step_s = kernel_ms / 1e3 / args.steps             # cast + fill, events around the region
        step_achieved = bytes_per_launch / step_s / 1e9
        traffic, traffic_source = None, None
        if traffic_live is not None:
            traffic = traffic_live["traffic"]
            traffic_source = (f"measured by this run: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes, --kernel-trace only) over 8 steps of "
                              f"this command; WRITE_SIZE {traffic_live['write_bytes']:.0f} B + 2 x FETCH_SIZE {traffic_live['fetch_bytes_raw']:.0f} B "
                              f"(gfx950 FETCH correction, an upper bound for narrow gathers) per launch of {fill_kernel}")
        elif args.traffic != "off" and not args.top_view:      # (the committed figure is the plain camera fill's: never another kernel's)
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    with open(tpath) as f:
                        t = json.load(f)
                    if t.get("workload") == args.workload and t.get("batch") == B and t.get("kernel", "rcw_fill256_kernel") == fill_kernel:
                        traffic = t.get("traffic_bytes_per_launch")
                        traffic_source = (f"profiles/pmc_traffic.json: WRITE_SIZE + 2 x FETCH_SIZE of {fill_kernel} from separate rocprofv3 --pmc "
                                          "passes of this command at N = 1, committed; not re-measured in this run"
                                          + (f" ({world} ranks: each rank runs the same per-GPU workload)" if world > 1 else ""))
                except Exception:
                    traffic = None
        out = {
            "metric": "env-steps/sec (whole node) + frames/sec, SingleRoom batch=4096 cols=256",
            "value": world * B * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "ms_per_step_min": ranks["ms_per_step_min"],
            "ms_per_step_max": ranks["ms_per_step_max"],
            "timing": "per rank: barrier | clock | K steps | stream + device synchronised | clock; MAX over ranks afterwards (no collective inside the timed region)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"SingleRoom {kw['height_tile_map_tu']}x{kw['width_tile_map_tu']}, {N} view columns, "
                            f"H_cam {Hc}, {B} agents per GPU ({args.workload}; BASELINE.json configs[1] is cfg2)",
                "agents_per_gpu": B,
                "global_batch": world * B,
                "auto_reset": not args.no_auto_reset,
                "agents_that_hit_reference_BoundsError": bounds_errors,
                "actions": "uniform 1..4 per agent per step, device resident",
                "sharding": f"agents by rank x{world}, no data-path collective",
                "step_form": step_form,
            },
            "frames_per_s": world * B * args.steps / dt,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": fill_kernel,
                "bytes_per_launch": bytes_per_launch,
                "launch_ms": fill_ms,
                "launches_timed": nrec,
                "per_rank": ranks["per_rank"],
                "whole_step": {
                    "kernels": ("rcw_fill256_cast_kernel alone: the fill workgroups write the frames the actions select among the successors the "
                                "previous launch cast, the casting workgroups commit the actions and cast the next successors beside them"
                                if step_form == "one-launch" else "rcw_cast_kernel + fill kernel"),
                    "launch_ms": step_s * 1e3,
                    "cast_ms": cast_ms,
                    "achieved": step_achieved,
                    "frac": step_achieved / HBM_PEAK_GBS,
                },
            },
        }
        if launch_note:
            out["roofline"]["launch_note"] = launch_note
        if gather is not None:
            out["gather"] = gather
        if args.top_view:
            pu = top_pu
            top_bytes = 4 * (kw["height_tile_map_tu"] * pu) * (kw["width_tile_map_tu"] * pu) * B
            top_gbs = top_bytes / (top_ms / 1e3) / 1e9
            out["config"]["render_top_view"] = True
            form = top_form
            out["top_view"] = {
                "form": form,
                "kernel": ("rcw_top_store_kernel (or its _units / _flat sibling for this geometry)" if form == "two-kernels" else "rcw_top_view_kernel"),
                "bound": "hbm", "bytes_per_launch": top_bytes, "launch_ms": top_ms,
                "achieved": top_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": top_gbs / HBM_PEAK_GBS,
                "note": "update_top_view! SR:446-483, 4*(H*pu)*(W*pu) bytes per agent written once; opt-in, not in `value` of the headline run"
                        + ("; two-kernel form: launch_ms is the store kernel (the one that writes the image); the drawing (VALU/LDS work, "
                           "planes of 1/32 of the image) " + ("rides in the camera fill's launch (rcw_fill256_draw_kernel, roofline.kernel)"
                                                              if fill_kernel == "rcw_fill256_draw_kernel" else
                                                              "is rcw_top_draw_kernel on a side stream beside the camera fill")
                           + " and is inside roofline.launch_ms — see profiles/ for its own duration; where the rule keeps the DRAWING on the handle's stream and sends "
                             "the camera fill to the side stream (big images: include/rcw.h, rcw_profile), launch_ms here runs from the fill's end to the step's end and "
                             "may hold the drawing's tail: `achieved` is then a lower bound of the store kernel's own rate" if form == "two-kernels" else ""),
            }
        if api_loop is not None:
            out["api_loop"] = api_loop
        return out

    gather = None
    traffic_live = None
    headline_ready = {"line": None}
    pending = {"what": "nothing yet"}

    def emit(line_dict):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line_dict) + "\n").encode())

    if dist is not None:
        import threading

        rehearsal = args.rehearse_on_one_gpu
        if rank == 0:
            headline_ready["line"] = build_line(None)
        timer = threading.Timer(args.watchdog_seconds, make_watchdog(rank, emit, headline_ready, pending, args.watchdog_seconds))
        timer.daemon = True
        timer.start()
        ok = 1
        Bf = min(B, 256) if rehearsal else B             # agents per rank in the FRAME gather (all of them on a real node)

        def all_gather(out, inp, what):
            pending["what"] = (f"all_gather_into_tensor of {what}: {inp.numel() * inp.element_size()} B per rank -> "
                               f"{out.numel() * out.element_size()} B, world {world}")
            if rehearsal:                                # gloo all-gathers host tensors only
                host = torch.empty(out.shape, dtype=out.dtype)
                dist.all_gather_into_tensor(host, inp.cpu())
                out.copy_(host)
            else:
                dist.all_gather_into_tensor(out, inp)

        def all_reduce_max(values, what):
            pending["what"] = f"all_reduce(MAX) of {what}, world {world}"
            t = torch.tensor(values, dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return [float(v) for v in t]

        try:
            if args.hang_rank == rank:
                pending["what"] = "nothing: this rank was told to hang (--hang-rank, test only)"
                time.sleep(10 ** 6)
            h_loc, c_loc = env.columns_device()
            h_loc, c_loc = h_loc.torch(sync=False), c_loc.torch(sync=False)
            obs_loc = env.camera_view.torch(sync=False).view(torch.int32)[:Bf]
            gh = torch.empty((world * B, N), dtype=torch.int32, device=dev)
            gc = torch.empty((world * B, N), dtype=torch.uint8, device=dev)
            frames_all = torch.empty((world * B, N, Hc), dtype=torch.uint32, device=dev)
            frames_part = frames_all.view(torch.int32)[:world * Bf]

            def timed(fn, reps):
                fn()
                rt.synchronize(); pending["what"] = "barrier inside the gather timing"; barrier()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                rt.synchronize(); pending["what"] = "barrier inside the gather timing"; barrier()
                return (time.perf_counter() - t0) / reps * 1e6

            def cols():
                all_gather(gh, h_loc, "height_line_pu (int32)")
                all_gather(gc, c_loc, "colour ids (uint8)")

            def cols_expand():
                cols()
                env.expand_columns(gh, gc, out=frames_all)

            t_cols, t_cols_expand = timed(cols, 2 if rehearsal else 10), timed(cols_expand, 1 if rehearsal else 5)
            t_frames = timed(lambda: all_gather(frames_part, obs_loc, "frames (uint32 as int32)"), 1 if rehearsal else 3)
            if rehearsal:
                # what a rehearsal CAN check: the gathered global batch is this rank's own shard in the right place
                rt.synchronize()
                assert torch.equal(gh[rank * B:(rank + 1) * B], h_loc) and torch.equal(gc[rank * B:(rank + 1) * B], c_loc)
                assert torch.equal(frames_part[rank * Bf:(rank + 1) * Bf], obs_loc)
        except Exception as e:   # noqa: BLE001 — a reported extra must never cost the bench line
            ok = 0
            gather = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            print(f"bench.py: rank {rank}: the optional gather failed in: {pending['what']}: {type(e).__name__}: {e}", file=sys.stderr)
            t_cols = t_cols_expand = t_frames = 0.0
        try:
            # collective-safe: every rank reaches this all-reduce, whatever happened above on it
            tt = all_reduce_max([t_cols, t_cols_expand, t_frames, -float(ok)], "the gather timings and the ok flag")
            if tt[3] > -1.0:                              # some rank failed
                gather = gather or {"error": "the gather failed on another rank"}
            else:
                gather = {"ranks": world, "agents_per_rank": B, "descriptor_bytes_per_rank": 5 * N * B,
                          "frame_bytes_per_rank": 4 * N * Hc * Bf, "columns_us": tt[0],
                          "columns_plus_expand_us": tt[1], "frames_us": tt[2],
                          "note": "max over ranks, host-timed between barriers; not part of `value` (the gather is optional)"}
                if rehearsal:
                    gather["rehearsal"] = ("all ranks on ONE GPU, collectives over gloo staged through host memory, frames of the "
                                           f"first {Bf} agents of every rank: the code path, not a measurement")
        except Exception as e:   # noqa: BLE001
            gather = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        timer.cancel()
        pending["what"] = "nothing (the gather is over)"

    if rank == 0:
        if world == 1 and args.traffic == "live":
            traffic_live = live_traffic(args.workload, B, fill_kernel, extra_args=(["--top-view"] if args.top_view else []) + ["--step-form", args.step_form])
        out = build_line(gather)
        if not args.no_cpu_baseline:
            # rank 0's own host cores, outside every timed region, at any N (the other ranks wait at the closing barrier)
            out["cpu_baseline"] = cpu_baseline(kw, B, args.cpu_baseline_seconds)
        emit(out)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
