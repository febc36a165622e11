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
