"""Reader and judge for the files julia/make_reference_fixtures.jl writes (the real RayCastWorlds.jl's outputs on
the discriminating inputs).  Test infrastructure: used by tests/test_reference_fixtures.py, and by the self-test of
the kit, which feeds it files in the same format written from the CPU oracle under a known switch setting.

A "backend" is any callable  run(case_inputs, switches, directions) -> dict of arrays  (oracle or HIP engine).
"""
import itertools
import os

import numpy as np

SWITCHES = ("dda_tie_break", "dda_distance", "normalize_mode")
RAY_FIELDS = ("ray_direction_bits", "ray_stop_position_tu", "ray_hit_dimension", "ray_distance_bits")


SEEDED_NAME = "seeded_rng_mt1_resets"       # julia/make_reference_fixtures.jl: seeded_case


def lcg_actions(n, seed=99):
    """The action stream of make_reference_fixtures.jl / tests/c_abi_harness.c."""
    s, out = seed, []
    for _ in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out.append(1 + (s >> 33) % 4)
    return out


def read_case(directory, name):
    """One <name>.txt (+ the raw images) as a dict of numpy arrays."""
    rec = {}
    with open(os.path.join(directory, name + ".txt")) as f:
        for line in f:
            key, _, rest = line.rstrip("\n").partition(" ")
            rec[key] = rest
    H, W, N, nd, Hc, Ht, Wt = (int(v) for v in rec["shape"].split())
    ints = lambda k: np.array([int(v) for v in rec[k].split()], dtype=np.int64)   # noqa: E731
    out = dict(name=name, versions=rec.get("versions", ""), H=H, W=W, N=N, nd=nd, Hc=Hc, Ht=Ht, Wt=Wt)
    for k in ("directions_wu_bits", "ray_direction_bits", "ray_stop_position_tu", "ray_hit_dimension",
              "ray_distance_bits", "rollout_actions", "rollout_position_bits", "rollout_direction_au",
              "rollout_reward_bits", "rollout_done"):
        out[k] = ints(k)
    out["rollout_error_step"] = int(rec["rollout_error_step"])
    # Julia column-major (H_cam, N) == C-order (N, H_cam)
    out["camera_view"] = np.fromfile(os.path.join(directory, name + ".camera_view.u32"), dtype="<u4").reshape(N, Hc)
    out["top_view"] = np.fromfile(os.path.join(directory, name + ".top_view.u32"), dtype="<u4").reshape(Wt, Ht)
    after = os.path.join(directory, name + ".camera_view_after_rollout.u32")
    if os.path.exists(after):
        out["camera_view_after_rollout"] = np.fromfile(after, dtype="<u4").reshape(N, Hc)
    return out


def read_manifest(directory):
    names = [l.strip() for l in open(os.path.join(directory, "manifest.tsv")) if l.strip() and not l.startswith("#")]
    return [read_case(directory, n) for n in names]


def directions_from_bits(ref):
    return ref["directions_wu_bits"].astype(np.uint32).view(np.float32).reshape(ref["nd"], 2)


def compare(ref, got):
    """Names of the fields where `got` (a backend's output) differs from the reference record."""
    bad = []
    for k in RAY_FIELDS + ("camera_view", "top_view"):
        if k in got and not np.array_equal(np.asarray(got[k]).reshape(-1).astype(np.int64),
                                           np.asarray(ref[k]).reshape(-1).astype(np.int64)):
            bad.append(k)
    if "rollout" in got:
        r = got["rollout"]
        n = len(ref["rollout_direction_au"])
        if (r["error_step"] != ref["rollout_error_step"] or
                not np.array_equal(r["position_bits"][:2 * n], ref["rollout_position_bits"]) or
                not np.array_equal(r["direction_au"][:n], ref["rollout_direction_au"]) or
                not np.array_equal(r["reward_bits"][:n], ref["rollout_reward_bits"]) or
                not np.array_equal(r["done"][:n], ref["rollout_done"])):
            bad.append("rollout")
        if "camera_view_after_rollout" in ref and not np.array_equal(r["camera_view"], ref["camera_view_after_rollout"]):
            bad.append("camera_view_after_rollout")
    return bad


def judge(refs, cases_by_name, run):
    """Try all 8 switch settings.  Returns (table, table_matches_libm) where table maps each setting to the list of
    (case, differing fields) it fails on; an empty list = that setting reproduces the reference bit for bit.  The
    reference's own directions_wu is installed in the backend first, so a cos/sin difference between Julia and the C
    library cannot masquerade as a switch mismatch; table_matches_libm says whether it was needed at all."""
    table = {}
    for setting in itertools.product((0, 1), repeat=3):
        sw = dict(zip(SWITCHES, setting))
        failures = []
        for ref in refs:
            got = run(cases_by_name[ref["name"]], sw, directions_from_bits(ref))
            bad = compare(ref, got)
            if bad:
                failures.append((ref["name"], bad))
        table[setting] = failures
    return table


# ---- the seeded case: SingleRoom(; rng = MersenneTwister(1)) with two reset!(env) inside a 64-step rollout ----------
def read_seeded(directory, name=SEEDED_NAME):
    """<name>.txt of the seeded case, or None when the directory has none (fixtures made before round 4)."""
    path = os.path.join(directory, name + ".txt")
    if not os.path.exists(path):
        return None
    rec = {}
    with open(path) as f:
        for line in f:
            key, _, rest = line.rstrip("\n").partition(" ")
            rec[key] = rest
    H, W, N, nd, Hc, Ht, Wt = (int(v) for v in rec["shape"].split())
    ints = lambda k: np.array([int(v) for v in rec[k].split()], dtype=np.int64)   # noqa: E731
    out = dict(name=name, H=H, W=W, N=N, nd=nd, Hc=Hc, Ht=Ht, Wt=Wt, rollout_error_step=int(rec["rollout_error_step"]))
    for k in ("directions_wu_bits", "reset_steps", "reset_states", "rollout_actions", "rollout_position_bits", "rollout_direction_au",
              "rollout_reward_bits", "rollout_done", "rollout_goal"):
        out[k] = ints(k)
    out["reset_states"] = out["reset_states"].reshape(-1, 5)                     # goal_i goal_j x_bits y_bits heading
    assert len(out["reset_states"]) == 1 + len(out["reset_steps"])
    out["camera_view"] = np.fromfile(os.path.join(directory, name + ".camera_view.u32"), dtype="<u4").reshape(N, Hc)
    out["top_view"] = np.fromfile(os.path.join(directory, name + ".top_view.u32"), dtype="<u4").reshape(Wt, Ht)
    out["camera_view_after_rollout"] = np.fromfile(os.path.join(directory, name + ".camera_view_after_rollout.u32"), dtype="<u4").reshape(N, Hc)
    return out


def replay_seeded(ref, backend):
    """Replays the seeded trajectory on `backend` — an object with set_state(goal, position, heading), step(action) -> 0 or
    an error code, state() -> (position (2,) float32, heading, reward, done, goal (2,)), camera_view(), top_view() — injecting
    the dumped post-reset states where the reference called reset!(env), and returns the names of what differs."""
    def inject(row):
        gi, gj, xb, yb, d = (int(v) for v in row)
        pos = np.array([xb, yb], dtype=np.uint32).view(np.float32)
        backend.set_state([gi, gj], pos, d)

    bad = []
    inject(ref["reset_states"][0])
    if not np.array_equal(backend.camera_view(), ref["camera_view"]):
        bad.append("camera_view after construction")
    if not np.array_equal(backend.top_view(), ref["top_view"]):
        bad.append("top_view after construction")
    resets = {int(k): i + 1 for i, k in enumerate(ref["reset_steps"])}
    n = len(ref["rollout_direction_au"])
    error_step = 0
    for k, a in enumerate(ref["rollout_actions"].tolist(), start=1):
        if k in resets:
            inject(ref["reset_states"][resets[k]])
            p, d, r, dn, g = backend.state()
            if float(r) != 0.0 or bool(dn):
                bad.append(f"reward / done after the reset before step {k}")
        if backend.step(int(a)) != 0:
            error_step = k
            break
        if k > n:
            bad.append(f"the reference stopped after {n} steps"); break
        p, d, r, dn, g = backend.state()
        i = k - 1
        if (np.asarray(p, dtype=np.float32).view(np.uint32).tolist() != ref["rollout_position_bits"][2 * i:2 * i + 2].tolist()
                or int(d) != int(ref["rollout_direction_au"][i]) or int(np.float32(r).view(np.uint32)) != int(ref["rollout_reward_bits"][i])
                or int(dn) != int(ref["rollout_done"][i]) or [int(g[0]), int(g[1])] != ref["rollout_goal"][2 * i:2 * i + 2].tolist()):
            bad.append(f"state after step {k}"); break
    if error_step != ref["rollout_error_step"]:
        bad.append(f"BoundsError step {error_step} vs {ref['rollout_error_step']}")
    if not bad and not np.array_equal(backend.camera_view(), ref["camera_view_after_rollout"]):
        bad.append("camera_view after the rollout")
    return bad
