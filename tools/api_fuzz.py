#!/usr/bin/env python3
"""Dev tool (GPU box): stateful fuzzing of the host API.  Random SEQUENCES of calls — steps with host / device / scalar
actions, masked and full resets, injected states, rejected actions, another stream, another output buffer
(rcw_bind_obs), another top-view form, another form of the step (one launch / two launches), stand-alone re-renders, ray materialisation, descriptor expansion, profiling
on / off — against the CPU oracle driven by the same sequence; every observable is compared after every call.

    python tools/api_fuzz.py [runs] [seed] [ops per run] [sharded|pairs]

"sharded": the engine sits behind ShardedSingleRoom in a torch.distributed "nccl" (= RCCL) group of ONE rank, with the
collective forced, and the observation gather — both transports (torch.distributed / the library's own ncclAllGather),
both modes (columns + expansion / frames) — joins the calls.
"pairs": two handles of different geometry live together on the device and the calls alternate between them at random
(what is per FUNCTION or per DEVICE in the library — kernel attributes, the loaded RCCL — is shared by them).
"""
import os
import socket
import sys

SHARDED = len(sys.argv) > 4 and sys.argv[4] == "sharded"
PAIRS = len(sys.argv) > 4 and sys.argv[4] == "pairs"
if SHARDED:                                                  # (the rendezvous variables before anything touches the GPU)
    with socket.socket() as _s:
        _s.bind(("127.0.0.1", 0))
        _port = _s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import raycastworlds_jl_amd as RCW
from helpers import assert_state_equal
from oracle import oracle as O

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 60
O.set_num_threads(8)
if SHARDED:
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
GEOMETRIES = (
    dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64),
    dict(height_tile_map_tu=8, width_tile_map_tu=16, num_rays=100, height_camera_view_pu=100),
    dict(height_tile_map_tu=12, width_tile_map_tu=7, num_rays=33, height_camera_view_pu=37),
    dict(height_tile_map_tu=6, width_tile_map_tu=9, num_rays=256, height_camera_view_pu=64),
    dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=128, height_camera_view_pu=27),
    dict(height_tile_map_tu=5, width_tile_map_tu=5, num_rays=7, height_camera_view_pu=512),
    dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256),
    dict(height_tile_map_tu=9, width_tile_map_tu=11, num_rays=150, height_camera_view_pu=250),
    dict(height_tile_map_tu=4, width_tile_map_tu=20, num_rays=64, height_camera_view_pu=24, player_radius_wu=0.3, position_increment_wu=0.2),
    dict(height_tile_map_tu=32, width_tile_map_tu=32, num_rays=64, height_camera_view_pu=128),
    # 256-row camera views: without a top view these take the one-launch step (a wavefront per agent / a workgroup per agent / the table's tail)
    dict(height_tile_map_tu=7, width_tile_map_tu=9, num_rays=300),
    dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=512),
    dict(height_tile_map_tu=6, width_tile_map_tu=6, num_rays=1100, num_directions=32),
    dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=200, num_directions=96),
)
counts = {}
OPS = ["act_host", "act_host", "act_device", "act_device", "act_scalar", "reset_mask", "reset_all", "reset_rng", "set_state", "bad_action", "stream",
       "bind_obs", "form", "step_form", "step_form", "rerender", "rays", "expand", "profile"] + (["gather_columns", "gather_columns_abi", "gather_obs", "gather_obs_abi"] * 2 if SHARDED else [])


class Ctx:
    """One handle, its oracle, and what the calls need to know of its geometry."""

    def __init__(self, tag):
        self.tag = tag
        kw = dict(GEOMETRIES[int(rng.integers(len(GEOMETRIES)))])
        self.B = int(rng.choice([1, 7, 64, 300]))
        self.top = bool(rng.integers(0, 2))
        if self.top:
            kw.update(render_top_view=True, pu_per_tu=int(rng.choice([8, 10, 12, 13, 16, 20, 24, 32])))
        if rng.integers(0, 4) == 0:
            kw["T"] = "Float64"
        self.auto = bool(rng.integers(0, 2))
        seed = int(rng.integers(1, 1 << 30))
        self.sh = None
        if SHARDED:
            self.sh = RCW.ShardedSingleRoom(self.B, collective="always", device=0, seed=seed, auto_reset=self.auto, out_of_bounds=1, **kw)
            self.env = self.sh.env
        else:
            self.env = RCW.SingleRoomModule.SingleRoom(batch=self.B, seed=seed, auto_reset=self.auto, out_of_bounds=1, **kw)
        okw = {k: v for k, v in kw.items() if k not in ("T",)}
        if kw.get("T") == "Float64":                          # convert(Float64, .) of the same kwargs (SR:263-270)
            okw["world_unit_bits"] = 64
            for key in ("player_radius_wu", "position_increment_wu", "semi_field_of_view_wu", "camera_height_tile_wu"):
                if key in kw:
                    okw[key + "_f64"] = float(kw[key])
        if self.top:
            okw["render_top_view"] = 1
        self.orc = O.OracleBatch(self.B, seed=seed, auto_reset=1 if self.auto else 0, out_of_bounds=1, **okw)
        self.kw = kw
        self.H, self.W, self.N = kw["height_tile_map_tu"], kw["width_tile_map_tu"], kw["num_rays"]
        self.Hc = kw.get("height_camera_view_pu", 256)
        self.keep = []                                       # buffers / streams handed to the engine stay alive for the run
        self.log, self.last, self.last_form = [], {}, None

    def check(self, where, rays=False):
        env, orc = self.env, self.orc
        assert_state_equal(env, orc, rays=rays, where=where)
        np.testing.assert_array_equal(env.world.episode, orc.episode, err_msg=f"episode {where}")
        if self.top:
            np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"top_view {where}")

    def call(self, op, where):
        env, orc, B, H, W, N, Hc, top, kw = self.env, self.orc, self.B, self.H, self.W, self.N, self.Hc, self.top, self.kw
        self.log.append(op)
        if op in ("act_host", "act_device"):
            a = rng.integers(1, 5, B).astype(np.uint8)
            RCW.act_(env, a if op == "act_host" else torch.from_numpy(a).cuda())
            assert orc.step(a) == 0
        elif op == "act_scalar":
            a = int(rng.integers(1, 5))
            RCW.act_(env, a)
            assert orc.step(np.full(B, a, dtype=np.uint8)) == 0
        elif op in ("reset_mask", "reset_all"):
            mask = (rng.random(B) < 0.5).astype(np.uint8) if op == "reset_mask" else None
            s = int(rng.integers(0, 1 << 30))
            RCW.reset_(env, mask=mask, seed=s)
            orc.reset(mask=mask, seed=s)
        elif op == "reset_rng":
            # the reference's `rng` keyword (SR:49): draws from the caller's generator(s) on the host, in the reference's order;
            # the oracle gets the same draws made here from generators in the same state
            from raycastworlds_jl_amd.single_room import reference_reset_draws
            mask = (rng.random(B) < 0.5).astype(np.uint8) if rng.integers(0, 2) else None
            s = int(rng.integers(0, 1 << 30))
            per_agent = bool(rng.integers(0, 2))
            mine = [np.random.default_rng([s, a]) for a in range(B)] if per_agent else np.random.default_rng(s)
            theirs = [np.random.default_rng([s, a]) for a in range(B)] if per_agent else np.random.default_rng(s)
            nd = kw.get("num_directions", 128)
            goal, pos, d = np.ones((B, 2), np.int32), np.ones((B, 2), np.float64 if kw.get("T") == "Float64" else np.float32), np.zeros(B, np.int32)
            for a in range(B):
                if mask is None or mask[a]:
                    gi, gj, ti, tj, da = reference_reset_draws(theirs[a] if per_agent else theirs, H, W, nd)
                    goal[a], pos[a], d[a] = (gi, gj), (ti - 0.5, tj - 0.5), da
            self.last = dict(op=op, per_agent=per_agent, mask=None if mask is None else mask.tolist())
            if self.sh is not None:
                self.sh.reset_(local_mask=mask, rng=mine)
            else:
                RCW.reset_(env, mask=mask, rng=mine)
            orc.set_state(goal, pos, d, mask=mask)
        elif op == "set_state":
            if H < 4:
                return                                       # (no second free interior row to move the player to)
            goal = np.stack([rng.integers(2, H, B), rng.integers(2, W, B)], axis=1).astype(np.int32)
            tile = np.stack([rng.integers(2, H, B), rng.integers(2, W, B)], axis=1)
            same = (tile == goal).all(axis=1)                # the player never starts on the goal tile (SR:124)
            tile[same, 0] = np.where(goal[same, 0] > 2, goal[same, 0] - 1, goal[same, 0] + 1)
            pos = (tile - 0.5).astype(np.float64 if kw.get("T") == "Float64" else np.float32)
            d = rng.integers(0, kw.get("num_directions", 128), B).astype(np.int32)
            mask = (rng.random(B) < 0.6).astype(np.uint8) if rng.integers(0, 2) else None
            self.last = dict(op=op, mask=None if mask is None else mask.tolist())
            env.set_state(goal, pos, d, mask=mask)
            orc.set_state(goal, pos, d, mask=mask)
        elif op == "bad_action":
            a = rng.integers(1, 5, B).astype(np.uint8)
            a[int(rng.integers(0, B))] = int(rng.choice([0, 5, 255]))
            try:
                RCW.act_(env, a)
                raise SystemExit("an invalid action was accepted")
            except AssertionError:
                pass
            assert orc.step(a) == -2
        elif op == "stream":
            choice = int(rng.integers(0, 3))
            st = torch.cuda.Stream() if choice else None
            self.keep.append(st)
            env.sync()
            env.set_stream(st if choice != 2 else st.cuda_stream)
        elif op == "bind_obs":
            if rng.integers(0, 3) == 0:
                env.sync(); env.bind_obs(None)
            else:
                buf = torch.empty(B * N * Hc, dtype=torch.int32, device="cuda")
                self.keep.append(buf)
                env.sync(); env.bind_obs(buf.data_ptr())
            RCW.update_camera_view_(env)                     # (the new buffer holds nothing yet: SR:374's call fills it)
        elif op == "form":
            if not top:
                return
            form = [None, "one-kernel", "two-kernels", "in-place"][int(rng.integers(0, 4))]
            try:
                nr = int(rng.integers(0, 4))
                self.last_form = (form, nr)
                env.set_top_view_form(form, runs=nr)
            except (RuntimeError, ValueError, AssertionError, NotImplementedError):
                pass                                         # (a form this geometry cannot take is refused, nothing changes)
            RCW.update_top_view_(env)
        elif op == "step_form":
            form = [None, "one-launch", "two-launches"][int(rng.integers(0, 3))]
            try:
                env.set_step_form(form)
            except RuntimeError:
                assert form == "one-launch" and (top or Hc != 256), (form, top, Hc)   # (refused where the geometry cannot take it: nothing changes)
        elif op == "rerender":
            RCW.update_camera_view_(env)
            if top:
                RCW.update_top_view_(env)
        elif op == "rays":
            self.check(where + " rays", rays=True)
        elif op == "expand":
            h, c = env.columns_device()
            out = env.expand_columns(h.torch(), c.torch())
            env.sync()
            np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32).reshape(orc.camera_view.shape), orc.camera_view)
        elif op == "profile":
            env.profile(bool(rng.integers(0, 2)))
        elif op in ("gather_columns", "gather_columns_abi"):
            gh, gc = self.sh.gather_columns() if op == "gather_columns" else self.sh.gather_columns_abi()
            env.sync()
            np.testing.assert_array_equal(gh.cpu().numpy(), orc.col_height, err_msg=op)
            np.testing.assert_array_equal(gc.cpu().numpy(), orc.col_colour, err_msg=op)
        elif op in ("gather_obs", "gather_obs_abi"):
            mode = str(rng.choice(["columns", "frames"]))
            frames = self.sh.gather_observations(mode) if op == "gather_obs" else self.sh.gather_observations_abi(mode)
            env.sync()
            np.testing.assert_array_equal(frames.cpu().numpy().view(np.uint32).reshape(orc.camera_view.shape), orc.camera_view, err_msg=f"{op} {mode}")
        self.check(where)

    def diagnose(self):
        env, orc, B = self.env, self.orc, self.B
        print(f"FAILED at handle {self.tag}: B={B} top={self.top} auto_reset={self.auto} kw={self.kw}\n  ops: {self.log}\n"
              f"  last arguments: {self.last}, last form asked: {self.last_form}", flush=True)
        if not self.top:
            return
        print(f"  top view form now: {env.top_view_form()}")
        try:
            env.clear_error()
            diff = (env.top_view_host() != orc.top_view).reshape(B, -1)
            bad = np.nonzero(diff.any(axis=1))[0]
            print(f"  agents with wrong top-view pixels: {bad.tolist()} ({diff.sum(axis=1)[bad].tolist()} pixels)")
            if len(bad):
                a0 = int(bad[0]); q = np.nonzero(diff[a0])[0]
                print(f"  agent {a0}: flat pixel offsets {q[:40].tolist()} got {env.top_view_host(a0, 1).reshape(-1)[q[:8]].tolist()} want {orc.top_view[a0].reshape(-1)[q[:8]].tolist()}")
            RCW.update_top_view_(env)
            print(f"  after one more rcw_update_top_view: {int((env.top_view_host() != orc.top_view).sum())} wrong pixels")
        except Exception as e:                               # noqa: BLE001
            print("  (diagnosis failed:", e, ")")

    def close(self):
        if self.sh is not None:
            self.sh.close()
        self.env.close(); self.orc.close()


for run in range(runs):
    ctxs = [Ctx(f"{run}.{i}") for i in range(2 if PAIRS else 1)]
    for c in ctxs:
        c.check(f"run {c.tag} after create")
    for k in range(n_ops):
        c = ctxs[int(rng.integers(len(ctxs)))]
        op = str(rng.choice(OPS))
        counts[op] = counts.get(op, 0) + 1
        try:
            c.call(op, f"run {c.tag} op {k} ({op})")
        except BaseException:
            c.diagnose()
            raise
    if PAIRS:                                                # (the other handle was not disturbed by the last calls of this one)
        for c in ctxs:
            c.check(f"run {c.tag} at the end")
    for c in ctxs:
        c.close()
    if run % 5 == 4:
        print(f"run {run + 1} ok", flush=True)
print(f"{runs} runs x {n_ops} calls, every observable equal to the oracle's after every call; calls made: {dict(sorted(counts.items()))}")
if SHARDED:
    dist.destroy_process_group()
